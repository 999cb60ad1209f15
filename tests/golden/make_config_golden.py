#!/usr/bin/env python3
"""Generate tests/golden/config.json: the configuration schema the reference launcher composes, read from its files.

Build container only.  Sources (paths relative to the reference checkout):
  scripts/rlg_hydra.py:15-249     the structured configs (dataclasses SimConfig, EnvConfig, Trifinger, TrifingerDifficulty1..4, Args):
                                  parsed with `ast` - the file itself cannot be imported here (hydra, omegaconf, isaacgym are absent).
                                  Per class: annotated fields with their defaults (`field(default_factory=lambda: X)` -> X, `MISSING` ->
                                  "???", a nested `SimConfig()` -> that class's resolved fields) and, separately, the UNannotated
                                  assignments (`task_difficulty = 4`, `episode_length = 750`: class attributes, not dataclass fields).
  resources/config/rlg/asymm.yaml the agent tree, as PyYAML loads it
  resources/config/config.yaml    the defaults list (gym / rlg group choices)
Numbers and key names only - no text of a reference file is kept.

    python tests/golden/make_config_golden.py            # writes tests/golden/config.json
"""
import ast
import json
import os

import yaml

REF = os.environ.get("TF_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config.json")


def literal(node, classes):
    """value of a default expression"""
    if isinstance(node, ast.Name) and node.id == "MISSING":
        return "???"
    if isinstance(node, ast.Call):
        fn = node.func.id if isinstance(node.func, ast.Name) else getattr(node.func, "attr", "")
        if fn == "field":
            for kw in node.keywords:
                if kw.arg == "default_factory" and isinstance(kw.value, ast.Lambda):
                    return literal(kw.value.body, classes)
                if kw.arg == "default":
                    return literal(kw.value, classes)
        if fn in classes and not node.args and not node.keywords:      # nested structured config with its defaults
            return resolved(fn, classes)["fields"]
        raise ValueError(f"unsupported default: {ast.dump(node)}")
    return ast.literal_eval(node)


def parse_classes(tree):
    classes = {}
    for node in tree.body:
        if not isinstance(node, ast.ClassDef):
            continue
        if not any((isinstance(d, ast.Name) and d.id == "dataclass") for d in node.decorator_list):
            continue
        classes[node.name] = node
    return classes


def resolved(name, classes):
    """fields (annotated, in definition order, parents first, a redefinition replaces) and unannotated class attributes of a dataclass"""
    node = classes[name]
    fields, plain = {}, {}
    for b in node.bases:
        if isinstance(b, ast.Name) and b.id in classes:
            r = resolved(b.id, classes)
            fields.update(r["fields"])
            plain.update(r["unannotated"])
    for st in node.body:
        if isinstance(st, ast.AnnAssign) and isinstance(st.target, ast.Name) and st.value is not None:
            fields[st.target.id] = literal(st.value, classes)
        elif isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Name):
            plain[st.targets[0].id] = literal(st.value, classes)
    return {"bases": [b.id for b in node.bases if isinstance(b, ast.Name)], "fields": fields, "unannotated": plain}


def main():
    src = open(os.path.join(REF, "scripts", "rlg_hydra.py")).read()
    classes = parse_classes(ast.parse(src))
    out = {"classes": {n: resolved(n, classes) for n in classes}}
    # the names the structured configs are registered under (cs.store(group=..., name=..., node=...))
    stores = []
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "store":
            kw = {k.arg: k.value for k in node.keywords}
            if "name" in kw and "node" in kw:
                stores.append({"group": ast.literal_eval(kw["group"]) if "group" in kw else None, "name": ast.literal_eval(kw["name"]),
                               "node": kw["node"].id if isinstance(kw["node"], ast.Name) else None})
    out["stores"] = stores
    out["rlg_asymm"] = yaml.safe_load(open(os.path.join(REF, "resources", "config", "rlg", "asymm.yaml")))
    top = yaml.safe_load(open(os.path.join(REF, "resources", "config", "config.yaml")))
    out["config_yaml"] = {"defaults": top["defaults"], "output_root": top["output_root"]}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", OUT, "classes:", sorted(out["classes"]), "stores:", stores)


if __name__ == "__main__":
    main()
