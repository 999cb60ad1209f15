"""Exception types raised at the task-selection boundary.

`InvalidTaskNameError` keeps the reference's message format (leibnizgym/utils/errors.py:9-24) because launch scripts
and logs of downstream users match on it."""

VALID_TASKS = ("Trifinger",)


class InvalidTaskNameError(Exception):
    """Raised by `parse_vec_task` when `args.task` names no environment of this package."""

    def __init__(self, task_name: str):
        self.task_name = task_name
        super().__init__("Unrecognized task: `{}`. Task should be in: {}".format(task_name, list(VALID_TASKS)))
