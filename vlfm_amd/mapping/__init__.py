from .base_map import BaseMap  # noqa: F401
from .obstacle_map import ObstacleMap, ObstacleMapBatch  # noqa: F401
from .value_map import ValueMap, ValueMapBatch  # noqa: F401
