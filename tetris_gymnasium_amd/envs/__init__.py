"""Environment surfaces (mirror of tetris_gymnasium/envs).  Like the reference (envs/__init__.py:6-14) the package registers its
env with gymnasium when gymnasium is importable: `gym.make("tetris_gymnasium_amd/Tetris", ...)` builds the drop-in `Tetris` the way
the reference's examples build theirs with `gym.make("tetris_gymnasium/Tetris", ...)` — under its own namespace, so both packages
can be installed side by side."""
from .._gym_compat import HAVE_GYMNASIUM
from .tetris import Tetris  # noqa: F401

if HAVE_GYMNASIUM:  # pragma: no cover - depends on the environment (UNVERIFIED: neither image of this project has gymnasium)
    try:
        from gymnasium import register

        register(id="tetris_gymnasium_amd/Tetris", entry_point="tetris_gymnasium_amd.envs:Tetris")
    except Exception:  # (registered twice, or a gymnasium without `register` at the top level)
        pass
