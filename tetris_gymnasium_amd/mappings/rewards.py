"""Rewards of the NumPy rule-set (reference mappings/rewards.py:5-15).  Per-handle data of the step kernel since ABI 4:
`Tetris(rewards_mapping=RewardsMapping(alife=0, game_over=-5))` works as in the reference.  `clear_line` is a field the
reference never reads (score is rows^2 * width, envs/tetris.py:621-630)."""
from dataclasses import dataclass


@dataclass
class RewardsMapping:
    alife: float = 1
    clear_line: float = 1
    game_over: float = 0
    invalid_action: float = -0.1
