"""Host-side utilities; `from leibnizgym_amd.utils import *` gives the message and dict helpers like the reference's
`leibnizgym/utils/__init__.py` (message.py + helpers.py star-imports)."""
from .helpers import (merged, print_debug, print_dict, print_error, print_info, print_notify, print_warn,  # noqa: F401
                      update_dict)

__all__ = ["update_dict", "merged", "print_info", "print_debug", "print_notify", "print_warn", "print_error", "print_dict"]
