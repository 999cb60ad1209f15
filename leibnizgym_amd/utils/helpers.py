"""Small host-side helpers (counterparts of reference leibnizgym/utils/helpers.py and message.py)."""
import collections.abc
import copy


def update_dict(orig_dict: dict, new_dict: collections.abc.Mapping) -> dict:
    """Nested dict.update().  Like the reference (helpers.py:25-45) it updates `orig_dict` in place and
    returns it; callers in this package pass a deep copy of the defaults, so module-level defaults are never
    mutated (the reference mutates them, which limits it to one env instance per process)."""
    for key, value in new_dict.items():
        if isinstance(value, collections.abc.Mapping):
            orig_dict[key] = update_dict(orig_dict.get(key, {}), value)
        else:
            orig_dict[key] = value
    return orig_dict


def merged(defaults: dict, overrides) -> dict:
    out = copy.deepcopy(defaults)
    if overrides is not None:
        update_dict(out, overrides)
    return out


def print_info(msg):
    print(f"[INFO] {msg}")


def print_debug(msg):
    print(f"[DEBUG] {msg}")


def print_notify(msg):
    print(f"[NOTIFY] {msg}")


def print_warn(msg):
    print(f"[WARN] {msg}")


def print_error(msg):
    print(f"[ERROR] {msg}")


def print_dict(d, nesting=0):
    for k, v in d.items():
        if isinstance(v, dict):
            print(" " * nesting + f"{k}:")
            print_dict(v, nesting + 4)
        else:
            print(" " * nesting + f"{k}: {v}")
