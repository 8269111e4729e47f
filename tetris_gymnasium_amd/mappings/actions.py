"""Action ids of the NumPy rule-set (reference mappings/actions.py:5-19).  Per-handle data of the step kernel since ABI 4: any
assignment of ids works as in the reference's elif chain (envs/tetris.py:222-259), except hard_drop sharing its value with a
field tested before it (refused by tg_create)."""
from dataclasses import dataclass


@dataclass
class ActionsMapping:
    move_left: int = 0
    move_right: int = 1
    move_down: int = 2
    rotate_clockwise: int = 3
    rotate_counterclockwise: int = 4
    hard_drop: int = 5
    swap: int = 6
    no_op: int = 7
