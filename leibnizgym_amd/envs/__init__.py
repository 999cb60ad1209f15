from .env_base import ISAACGYM_DEFAULT_CONFIG_DICT, IsaacEnvBase  # noqa: F401
from .trifinger import TrifingerEnv  # noqa: F401
