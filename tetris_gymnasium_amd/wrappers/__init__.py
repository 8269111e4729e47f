"""The reference's wrappers (tetris_gymnasium/wrappers/) over the drop-in `Tetris` env, computed by the HIP engine."""
from .grouped import GroupedActionsObservations  # noqa: F401
from .observation import FeatureVectorObservation, RgbObservation  # noqa: F401
