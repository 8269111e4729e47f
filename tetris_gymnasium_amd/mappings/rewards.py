"""Reward constants of the NumPy rule-set (reference mappings/rewards.py:5-15).  The engine implements
exactly these values; other mappings are rejected by `Tetris`."""
from dataclasses import dataclass


@dataclass
class RewardsMapping:
    alife: float = 1
    clear_line: float = 1
    game_over: float = 0
    invalid_action: float = -0.1
