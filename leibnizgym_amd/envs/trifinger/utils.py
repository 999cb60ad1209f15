"""Dimensions of the TriFinger platform and of the manipulated cuboid.

Counterpart of reference leibnizgym/envs/trifinger/utils.py; the numbers are pinned against the
reference by tests/golden/constants.npz (CuboidalObject(0.065): radius_3d 0.05629165,
max_com_distance_to_center 0.13870835, min_height 0.0325, max_height 0.1)."""
import enum
import math
from typing import Tuple, Union


class TrifingerDimensions(enum.Enum):
    # the reference writes `PoseDim = 7,` (a tuple, utils.py:26); nothing reads it - kept as an int here
    PoseDim = 7
    VelocityDim = 6
    StateDim = 13
    WrenchDim = 6
    NumFingers = 3
    JointPositionDim = 9
    JointVelocityDim = 9
    JointTorqueDim = 9
    GeneralizedCoordinatesDim = JointPositionDim
    GeneralizedVelocityDim = JointVelocityDim
    ObjectPoseDim = 7
    ObjectVelocityDim = 6


# radius of the arena in which the cube may be placed
ARENA_RADIUS = 0.195


class CuboidalObject:
    """Sizes derived from the cuboid's edge lengths; used for sampling poses inside the arena."""
    max_height = 0.1

    def __init__(self, size: Union[float, Tuple[float, float, float]]):
        self.size = size

    @property
    def size(self) -> Tuple[float, float, float]:
        return self._size

    @size.setter
    def size(self, size: Union[float, Tuple[float, float, float]]):
        self._size = (size, size, size) if isinstance(size, float) else tuple(size)
        self.radius_3d = max(self._size) * math.sqrt(3) / 2
        self.max_com_distance_to_center = ARENA_RADIUS - self.radius_3d
        self.min_height = self._size[2] / 2
