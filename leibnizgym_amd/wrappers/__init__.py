from .vec_task import VecTask, VecTaskPython  # noqa: F401
