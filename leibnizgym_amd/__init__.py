"""MI355X-native TriFinger vectorised environment behind leibnizgym's VecTask / RL-Games API."""
__all__ = ["TrifingerEnv", "IsaacEnvBase", "VecTaskPython"]


def __getattr__(name):
    if name in ("TrifingerEnv", "IsaacEnvBase"):
        from . import envs
        return getattr(envs, name)
    if name == "VecTaskPython":
        from .wrappers import VecTaskPython
        return VecTaskPython
    raise AttributeError(name)
