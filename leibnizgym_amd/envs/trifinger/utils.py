"""Sizes of the TriFinger platform and of the manipulated cuboid.

Numbers are pinned against the reference by tests/golden/constants.npz (generated from
leibnizgym/envs/trifinger/utils.py of the reference): for the 65 mm cube radius_3d = 0.05629165,
max_com_distance_to_center = 0.13870835, min_height = 0.0325, max_height = 0.1; arena radius 0.195 m."""
import enum
import math
from typing import Sequence, Tuple, Union

ARENA_RADIUS = 0.195          # metres: the cube's centre is sampled so that its bounding sphere stays inside


class TrifingerDimensions(enum.Enum):
    """Vector sizes used by the observation / state / action specs (members compare by value, so several names
    share one value exactly as in the reference enum)."""
    StateDim = 13             # pose (7) + twist (6)
    PoseDim = 7               # the reference's `PoseDim = 7,` is a 1-tuple by accident; nothing reads it
    VelocityDim = 6
    WrenchDim = 6
    NumFingers = 3
    JointPositionDim = 9
    JointVelocityDim = 9
    JointTorqueDim = 9
    GeneralizedCoordinatesDim = 9
    GeneralizedVelocityDim = 9
    ObjectPoseDim = 7
    ObjectVelocityDim = 6


class CuboidalObject:
    """Derived sizes of a cuboid with edge lengths `size` (a float means a cube)."""
    max_height = 0.1          # highest goal position of the object's centre

    def __init__(self, size: Union[float, Sequence[float]]):
        self.size = size

    @property
    def size(self) -> Tuple[float, float, float]:
        return self._size

    @size.setter
    def size(self, value: Union[float, Sequence[float]]):
        edges = (value,) * 3 if isinstance(value, (int, float)) else tuple(value)
        self._size = edges
        self.radius_3d = 0.5 * math.sqrt(3.0) * max(edges)               # bounding-sphere radius of the largest edge
        self.max_com_distance_to_center = ARENA_RADIUS - self.radius_3d   # sampling radius on the table
        self.min_height = 0.5 * edges[2]                                  # resting on the table
