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


# ---- structure pins: observation packing order, mirror sign / slice tables, convert.py's step_fn (values only) ----------------------
# A tiny symbolic evaluator over the AST of the reference's array-shuffling methods: a value is a list of columns, each column a
# (sign, source key, source column) triple; `x[..., a:b]`, unary minus, `jnp.concatenate([...], axis=-1)`, `.reshape(-1, w)` (rows of
# width w: the pattern then holds for every row), `self.mirror_joints(x)` and local names are understood, nothing is executed.
OBS_WIDTH = {"left_foot_touch": 1, "right_foot_touch": 1, "feet_position": 6, "base_position": 3, "base_orientation": 4, "base_height": 1,
             "center_of_mass_inertia": 10, "center_of_mass_velocity": 6}   # per ROW of the reshape the reference applies (train.py:1650, 1668)


class Sym:
    def __init__(self, cols, per_row=None):
        self.cols = cols          # [(sign, key, col)], col = None: "the whole array of `key`" (width not needed)
        self.per_row = per_row


def _slice_bounds(sub):
    sl = sub.slice
    if isinstance(sl, ast.Tuple):      # x[..., a:b]
        sl = sl.elts[-1]
    if isinstance(sl, ast.Slice):
        lo = const(sl.lower) if sl.lower is not None else 0
        hi = const(sl.upper) if sl.upper is not None else None
        return lo, hi
    raise NotConstant


def sym_eval(node, env, width_of, joints_fn):
    if isinstance(node, ast.Name):
        return env[node.id]
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub):
        v = sym_eval(node.operand, env, width_of, joints_fn)
        return Sym([(-s, k, c) for s, k, c in v.cols], v.per_row)
    if isinstance(node, ast.Subscript):
        if isinstance(node.value, ast.Name) and node.value.id in ("obs", "cmd", "observations", "commands") and isinstance(node.slice, ast.Constant):
            key = node.slice.value
            w = width_of(key)
            return Sym([(1, key, c) for c in range(w)] if w else [(1, key, None)])
        v = sym_eval(node.value, env, width_of, joints_fn)
        lo, hi = _slice_bounds(node)
        hi = len(v.cols) if hi is None else hi
        return Sym(v.cols[lo:hi], v.per_row)
    if isinstance(node, ast.Call):
        fn = dotted(node.func)
        if fn == "jnp.concatenate":
            parts = [sym_eval(e, env, width_of, joints_fn) for e in node.args[0].elts]
            return Sym([c for q in parts for c in q.cols], parts[0].per_row)
        if fn == "jnp.zeros":
            n = [kw for kw in node.keywords if kw.arg == "shape"]
            n = const(n[0].value)[0] if n else const(node.args[0])[0]
            return Sym([(0, None, None)] * n)
        if fn == "self.mirror_joints":
            return joints_fn(sym_eval(node.args[0], env, width_of, joints_fn))
        if isinstance(node.func, ast.Attribute) and node.func.attr == "reshape":
            v = sym_eval(node.func.value, env, width_of, joints_fn)
            try:
                shape = [const(a) for a in node.args]
            except NotConstant:
                return v              # .reshape(obs[...].shape): back to the flat layout, the per-row pattern stays
            return Sym(v.cols, per_row=shape[-1])
    raise NotConstant(ast.dump(node)[:120])


def cols_json(v):
    return {"per_row": v.per_row, "cols": [[s, k, c] for s, k, c in v.cols]}


def extract_mirror(cls: ast.ClassDef) -> dict:
    fns = {st.name: st for st in cls.body if isinstance(st, ast.FunctionDef)}
    out = {}
    # mirror_joints: j -> -concat(slices of j)
    mj = fns["mirror_joints"]
    env = {"j": Sym([(1, "j", c) for c in range(20)])}
    ret = None
    for st in mj.body:
        if isinstance(st, ast.Assign):
            env[st.targets[0].id] = sym_eval(st.value, env, lambda k: 0, None)
        if isinstance(st, ast.Return):
            ret = sym_eval(st.value, env, lambda k: 0, None)
    out["mirror_joints"] = {"line": mj.lineno, "perm": [c for _, _, c in ret.cols], "sign": [s for s, _, _ in ret.cols]}

    def joints_fn(v):
        assert len(v.cols) == 20, "mirror_joints takes 20 columns"
        return Sym([(ret.cols[i][0] * v.cols[ret.cols[i][2]][0], v.cols[ret.cols[i][2]][1], v.cols[ret.cols[i][2]][2]) for i in range(20)])

    def width_of(key):
        if "joint" in key or key == "actuator_force":
            return 20
        if "gyro" in key or "gravity" in key or key.endswith("velocity"):
            return OBS_WIDTH.get(key, 3)
        return OBS_WIDTH.get(key, 0)

    for name, var in (("mirror_obs", "obs"), ("mirror_cmd", "cmd")):
        fn = fns[name]
        env = {}
        res = {}
        for st in fn.body:
            if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name):
                if name == "mirror_cmd" and isinstance(st.value, ast.Subscript) and isinstance(st.value.value, ast.Name) and st.value.value.id == "cmd":
                    env[st.targets[0].id] = Sym([(1, st.value.slice.value, c) for c in range(16)])
                else:
                    env[st.targets[0].id] = sym_eval(st.value, env, width_of, joints_fn)
            if isinstance(st, ast.Return):
                d = st.value.args[0]          # xax.FrozenDict({...})
                for k, v in zip(d.keys, d.values):
                    res[k.value] = cols_json(sym_eval(v, env, width_of, joints_fn))
        out[name] = {"line": fn.lineno, "keys": res}
    return out


def extract_packing(cls: ast.ClassDef) -> dict:
    """Order, source key, wrapper and divisor of every entry of the concatenated observation vectors (train.py:1351-1433)."""
    fns = {st.name: st for st in cls.body if isinstance(st, ast.FunctionDef)}
    out = {}
    for name in ("run_actor", "run_critic"):
        fn = fns[name]
        src = {}
        entries = None
        for st in ast.walk(fn):
            if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name):
                v = st.value
                if isinstance(v, ast.Subscript) and isinstance(v.value, ast.Name) and v.value.id in ("observations", "commands") and isinstance(v.slice, ast.Constant):
                    src[st.targets[0].id] = v.slice.value
                elif isinstance(v, ast.List) and st.targets[0].id == "obs":
                    entries = v.elts
            if isinstance(st, ast.Call) and dotted(st.func) == "jnp.concatenate" and isinstance(st.args[0], ast.List) and len(st.args[0].elts) > 3:
                entries = st.args[0].elts
        recs = []
        for e in entries:
            rec = {"wrapper": None, "divisor": None}
            if isinstance(e, ast.BinOp) and isinstance(e.op, ast.Div):
                rec["divisor"] = const(e.right)
                e = e.left
            if isinstance(e, ast.Call):
                rec["wrapper"] = dotted(e.func).replace("self.", "")
                e = e.args[0]
            rec["var"] = e.id
            rec["key"] = src.get(e.id, e.id)      # zero_cmd is derived from the command (threshold recorded below)
            recs.append(rec)
        out[name] = {"line": fn.lineno, "entries": recs}
    # constants of the wrappers
    for st in ast.walk(fns["normalize_joint_vel"]):
        if isinstance(st, ast.BinOp) and isinstance(st.op, ast.Div):
            out["normalize_joint_vel_divisor"] = const(st.right)
    for st in ast.walk(fns["run_actor"]):
        if isinstance(st, ast.Compare) and isinstance(st.ops[0], ast.Lt):
            out["zero_cmd"] = {"threshold": const(st.comparators[0]), "expr": ast.unparse(st.left)[:80]}
    enc = fns["encode_projected_gravity"]
    for st in ast.walk(enc):
        if isinstance(st, ast.Call) and dotted(st.func) == "jnp.concatenate":
            out["encode_projected_gravity_order"] = [ast.unparse(e).split("[")[0] for e in st.args[0].elts]
    for st in enc.body:
        if isinstance(st, ast.Assign):
            out.setdefault("encode_projected_gravity_exprs", {})[st.targets[0].id] = ast.unparse(st.value)[:120]
    return out


def extract_step_fn(ctree) -> dict:
    for node in ast.walk(ctree):
        if isinstance(node, ast.FunctionDef) and node.name == "step_fn":
            args = [a.arg for a in node.args.args]
            entries = []
            for st in ast.walk(node):
                if isinstance(st, ast.Call) and dotted(st.func) == "jnp.concatenate" and isinstance(st.args[0], ast.List):
                    for e in st.args[0].elts:
                        rec = {"wrapper": None}
                        if isinstance(e, ast.Call):
                            rec["wrapper"] = dotted(e.func).replace("task.", "")
                            e = e.args[0]
                        rec["var"] = e.id
                        entries.append(rec)
            ret = [ast.unparse(e)[:40] for st in node.body if isinstance(st, ast.Return) for e in st.value.elts]
            return {"line": node.lineno, "args": args, "entries": entries, "returns": ret}
    return {}


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
            doc["packing"] = extract_packing(node)      # train.py:1329-1433
            doc["mirror"] = extract_mirror(node)        # train.py:1574-1756
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
    doc.setdefault("convert", {})["step_fn"] = extract_step_fn(ctree)      # convert.py:84-119
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_constants.json")
    with open(out, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
        f.write("\n")
    n = sum(len(m["entries"]) + len(m["calls"]) for m in doc["wiring"].values())
    print(f"wrote {out}: {len(doc['tables'])} tables, {len(doc['config_defaults'])} config defaults, {n} wiring records, "
          f"{len(doc['launch'].get('kwargs', {}))} launch overrides, {len(doc['class_defaults'])} classes with defaults")


if __name__ == "__main__":
    main()
