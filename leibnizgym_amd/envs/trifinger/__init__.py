from .trifinger_env import TRIFINGER_DEFAULT_CONFIG_DICT, TrifingerEnv  # noqa: F401
