"""Load the reference's importable leaf modules BY PATH (this container only).

/root/reference cannot travel to the GPU box; this loader is used only by
tests/golden/make_golden.py to generate the committed .npz fixtures.
The real package __init__ files pull `termcolor` / `isaacgym`, so empty parent
packages are registered and the six leaf files are executed unmodified
(SURVEY.md section 8c).
"""
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("TF_REFERENCE_ROOT", "/root/reference")

_LEAVES = [
    ("leibnizgym.utils.torch_utils", "leibnizgym/utils/torch_utils.py"),
    ("leibnizgym.utils.mdp", "leibnizgym/utils/mdp.py"),
    ("leibnizgym.utils.helpers", "leibnizgym/utils/helpers.py"),
    ("leibnizgym.envs.trifinger.utils", "leibnizgym/envs/trifinger/utils.py"),
    ("leibnizgym.envs.trifinger.rewards", "leibnizgym/envs/trifinger/rewards.py"),
    ("leibnizgym.envs.trifinger.sample", "leibnizgym/envs/trifinger/sample.py"),
]


def load_reference():
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    for pkg in ("leibnizgym", "leibnizgym.utils", "leibnizgym.envs", "leibnizgym.envs.trifinger"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []  # mark as package
            sys.modules[pkg] = m
    mods = {}
    for name, rel in _LEAVES:
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        mods[name.rsplit(".", 1)[-1] if name.count(".") == 2 else name.split(".")[-1]] = mod
        parent, _, child = name.rpartition(".")
        setattr(sys.modules[parent], child, mod)
    return types.SimpleNamespace(
        torch_utils=sys.modules["leibnizgym.utils.torch_utils"],
        mdp=sys.modules["leibnizgym.utils.mdp"],
        helpers=sys.modules["leibnizgym.utils.helpers"],
        tf_utils=sys.modules["leibnizgym.envs.trifinger.utils"],
        rewards=sys.modules["leibnizgym.envs.trifinger.rewards"],
        sample=sys.modules["leibnizgym.envs.trifinger.sample"],
    )
