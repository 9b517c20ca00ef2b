"""Extract the CONSTANTS the reference task definition holds into tests/golden/ref_constants.json.

Run in the build container (the reference tree does not exist on the GPU box):

    python tests/golden/make_ref_constants.py [/root/reference]

The reference cannot be imported offline, but its files can be read: this script parses train.py /
convert.py with `ast` (nothing is executed) and records VALUES only - joint tables, the config dataclass defaults, every
constant keyword argument of the task wiring methods (actuators, randomisers, events, resets, observations, commands, rewards,
terminations, curriculum, model constructor), the launch block, and convert.py's command names. Each entry keeps the line
number it was read from. tests/test_ref_constants.py asserts this build's tables (spec/constants.py, layout.default_config,
host/task.py launch_config and the reward table of kbj_config) against the JSON.

This pins what the in-tree reference files pin. It does NOT pin the un-vendored ksim / mujoco-mjx semantics.
"""
from __future__ import annotations

import ast
import json
import math
import os
import sys

WIRING_METHODS = ("get_actuators", "get_physics_randomizers", "get_events", "get_resets", "get_observations", "get_commands",
                  "get_rewards", "get_terminations", "get_curriculum", "get_model", "get_optimizer", "get_mujoco_model",
                  "get_mujoco_model_metadata")


class NotConstant(Exception):
    pass


def const(node):
    """Evaluate a literal expression: numbers, strings, tuples/lists/dicts of them, unary minus, + - * /, math.radians / math.pi."""
    if isinstance(node, ast.Constant):
        return node.value
    if isinstance(node, (ast.Tuple, ast.List)):
        return [const(e) for e in node.elts]
    if isinstance(node, ast.Dict):
        return {const(k): const(v) for k, v in zip(node.keys, node.values)}
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
        v = const(node.operand)
        return -v if isinstance(node.op, ast.USub) else v
    if isinstance(node, ast.BinOp) and isinstance(node.op, (ast.Add, ast.Sub, ast.Mult, ast.Div)):
        a, b = const(node.left), const(node.right)
        if isinstance(a, (int, float)) and isinstance(b, (int, float)):
            return {ast.Add: a + b, ast.Sub: a - b, ast.Mult: a * b, ast.Div: a / b if b else float("nan")}[type(node.op)]
        raise NotConstant
    if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id == "math" and node.attr == "pi":
        return math.pi
    if isinstance(node, ast.Call) and dotted(node.func) == "math.radians" and len(node.args) == 1:
        return math.radians(const(node.args[0]))
    raise NotConstant


def dotted(node) -> str:
    if isinstance(node, ast.Name):
        return node.id
    if isinstance(node, ast.Attribute):
        return dotted(node.value) + "." + node.attr
    return "?"


def call_record(call: ast.Call) -> dict:
    rec = {"call": dotted(call.func), "line": call.lineno, "kwargs": {}, "args": [], "symbolic": {}}
    for a in call.args:
        try:
            rec["args"].append(const(a))
        except NotConstant:
            rec["args"].append(None)
    for kw in call.keywords:
        if kw.arg is None:
            continue
        try:
            rec["kwargs"][kw.arg] = const(kw.value)
        except NotConstant:
            rec["symbolic"][kw.arg] = ast.unparse(kw.value)[:80]     # e.g. "self.config.ctrl_dt / 5": an expression, not a literal
    return rec


def method_calls(fn: ast.FunctionDef) -> dict:
    """Constant-argument calls of one wiring method. Dict-literal returns keep their keys; other calls are listed in order."""
    out = {"line": fn.lineno, "entries": {}, "order": [], "calls": []}
    keyed = set()
    for node in ast.walk(fn):
        if isinstance(node, ast.Dict):
            for k, v in zip(node.keys, node.values):
                if isinstance(k, ast.Constant) and isinstance(k.value, str) and isinstance(v, ast.Call):
                    out["entries"][k.value] = call_record(v)
                    out["order"].append(k.value)        # source order (the JSON itself is written with sorted keys)
                    keyed.add(id(v))
    for node in ast.walk(fn):
        if isinstance(node, ast.Call) and id(node) not in keyed:
            name = dotted(node.func)
            if name.split(".")[0] in ("ksim", "optax", "mujoco_scenes") or name[:1].isupper():
                rec = call_record(node)
                # nested helper calls such as ksim.AdditiveUniformNoise(mag=...) are kept as their own records
                out["calls"].append(rec)
    out["calls"].sort(key=lambda r: r["line"])
    return out


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    with open(os.path.join(ref, "train.py")) as f:
        tree = ast.parse(f.read())
    doc = {"_about": "values extracted by tests/golden/make_ref_constants.py (ast, nothing executed) from the reference's train.py / convert.py",
           "tables": {}, "config_defaults": {}, "wiring": {}, "launch": {}, "class_defaults": {}}
    for node in tree.body:
        if isinstance(node, ast.AnnAssign) and isinstance(node.target, ast.Name) and node.target.id in ("JOINT_BIASES", "JOINT_LIMITS"):
            d = const(node.value)
            doc["tables"][node.target.id] = {"line": node.lineno, "names": list(d), "values": list(d.values())}
        if isinstance(node, ast.ClassDef) and node.name == "HumanoidWalkingTaskConfig":
            for st in node.body:
                if isinstance(st, ast.AnnAssign) and isinstance(st.value, ast.Call) and dotted(st.value.func) == "xax.field":
                    for kw in st.value.keywords:
                        if kw.arg == "value":
                            doc["config_defaults"][st.target.id] = {"line": st.lineno, "value": const(kw.value)}
        if isinstance(node, ast.ClassDef) and node.name == "HumanoidWalkingTask":
            for st in node.body:
                if isinstance(st, ast.FunctionDef) and st.name in WIRING_METHODS:
                    doc["wiring"][st.name] = method_calls(st)
        if isinstance(node, ast.ClassDef):
            # attrs / dataclass field defaults of the in-tree reward / observation / command classes (e.g. error_scale defaults)
            fields = {}
            for st in node.body:
                if isinstance(st, ast.AnnAssign) and isinstance(st.target, ast.Name) and st.value is not None:
                    v = st.value
                    if isinstance(v, ast.Call) and dotted(v.func) in ("attrs.field", "xax.field"):
                        for kw in v.keywords:
                            if kw.arg in ("default", "value"):
                                try:
                                    fields[st.target.id] = const(kw.value)
                                except NotConstant:
                                    pass
                    else:
                        try:
                            fields[st.target.id] = const(v)
                        except NotConstant:
                            pass
            if fields and node.name not in ("HumanoidWalkingTaskConfig",):
                doc["class_defaults"][node.name] = {"line": node.lineno, "fields": fields}
        if isinstance(node, ast.If) and "__main__" in ast.unparse(node.test):
            for sub in ast.walk(node):
                if isinstance(sub, ast.Call) and dotted(sub.func) == "HumanoidWalkingTaskConfig":
                    rec = call_record(sub)
                    doc["launch"] = {"line": sub.lineno, "kwargs": rec["kwargs"], "symbolic": rec["symbolic"]}
    # convert.py: the command names of the deployment contract and the carry-size expression
    with open(os.path.join(ref, "convert.py")) as f:
        ctree = ast.parse(f.read())
    for node in ast.walk(ctree):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            if node.targets[0].id == "command_names":
                doc["convert"] = {"line": node.lineno, "command_names": const(node.value)}
        if isinstance(node, ast.keyword) and node.arg == "carry_size":
            doc.setdefault("convert", {})["carry_size_expr"] = ast.unparse(node.value)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_constants.json")
    with open(out, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
        f.write("\n")
    n = sum(len(m["entries"]) + len(m["calls"]) for m in doc["wiring"].values())
    print(f"wrote {out}: {len(doc['tables'])} tables, {len(doc['config_defaults'])} config defaults, {n} wiring records, "
          f"{len(doc['launch'].get('kwargs', {}))} launch overrides, {len(doc['class_defaults'])} classes with defaults")


if __name__ == "__main__":
    main()
