"""Custom errors (counterpart of reference leibnizgym/utils/errors.py:9-24)."""


class InvalidTaskNameError(Exception):
    def __init__(self, task_name):
        valid_tasks = ["Trifinger"]
        super().__init__(f"Unrecognized task: `{task_name}`. Task should be in: {valid_tasks}")
