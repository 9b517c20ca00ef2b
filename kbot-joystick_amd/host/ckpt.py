"""Checkpoint container in the layout xax writes for the reference (`humanoid_walking_task/run_N/checkpoints/ckpt.bin`,
convert.sh:4; saved every `save_every_n_seconds`, train.py:1788; read back by `load_ckpt(path, part="model")`, convert.py:39).

FORMAT PROVENANCE: the container format lives in the un-vendored xax 0.4.2 / equinox 0.12.2 packages, which are not in the
reference tree and cannot be installed here. What this module writes is the layout recorded in SURVEY.md appendix B.7 from
upstream memory - UNVALIDATED against a real ckpt.bin:
    gzip'd tar with members
      model_0      equinox `tree_serialise_leaves`: the array leaves of Model(actor, critic) in pytree-flatten order, written
                   back to back with numpy.save (one .npy blob per leaf)
      opt_state_0  the optax state the same way: for adamw = (ScaleByAdamState(count:int32, mu, nu), ...) -> count, mu leaves, nu leaves
      state        JSON: training counters
      config       YAML: the config dataclass
Everything this build needs beyond that to resume bit-exactly (env rows, reward carries, model carries, the pending observation
rows) goes into extra members prefixed `kbj_`, which a reader of the upstream layout ignores.

Leaf order of `model_0` (equinox flattens dataclass fields in declaration order: train.py:847-1046; convert.py:44-46 takes
`model.actor`): actor.input_proj.{weight,bias}, actor.rnns[l].{weight_ih,weight_hh,bias}, actor.output_proj.{weight,bias},
then the critic likewise == spec/layout.param_leaves == the flat parameter vector of include/kbj.h.
"""
from __future__ import annotations

import io
import json
import tarfile
import time
from typing import Dict, Optional

import numpy as np

from ..spec import layout as L


def _npy_blobs(arrays) -> bytes:
    buf = io.BytesIO()
    for a in arrays:
        np.save(buf, np.ascontiguousarray(a), allow_pickle=False)
    return buf.getvalue()


def _read_blobs(data: bytes, count: Optional[int] = None):
    buf, out = io.BytesIO(data), []
    while buf.tell() < len(data) and (count is None or len(out) < count):
        out.append(np.load(buf, allow_pickle=False))
    return out


def split_leaves(flat: np.ndarray, hidden_size: int, depth: int = 2, extra_obs=(0, 0)):
    """Flat fp32 parameter vector -> list of (name, array) in equinox leaf order. extra_obs: user observation columns (actor, critic)."""
    out, off = [], 0
    for name, shape in L.param_leaves(hidden_size, depth, extra_obs):
        n = int(np.prod(shape))
        out.append((name, np.asarray(flat[off:off + n], np.float32).reshape(shape)))
        off += n
    if off != flat.size:
        raise ValueError(f"parameter vector has {flat.size} floats, the layout for hidden_size {hidden_size} needs {off}")
    return out


def join_leaves(leaves, hidden_size: int, depth: int = 2, extra_obs=(0, 0)) -> np.ndarray:
    want = L.param_leaves(hidden_size, depth, extra_obs)
    if len(leaves) != len(want):
        raise ValueError(f"model_0 holds {len(leaves)} leaves, the layout for hidden_size {hidden_size} has {len(want)}")
    for a, (name, shape) in zip(leaves, want):
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"leaf {name}: shape {tuple(a.shape)} != {tuple(shape)}")
    return np.concatenate([np.asarray(a, np.float32).ravel() for a in leaves])


def _yaml(d: dict) -> str:
    try:
        import yaml
        return yaml.safe_dump(d, sort_keys=False)
    except Exception:   # yaml is present in this image; JSON is valid YAML if it ever is not
        return json.dumps(d, indent=1)


def _add(tar: tarfile.TarFile, name: str, data: bytes):
    info = tarfile.TarInfo(name)
    info.size, info.mtime = len(data), int(time.time())
    tar.addfile(info, io.BytesIO(data))


def save_ckpt(path: str, params: np.ndarray, opt_m: np.ndarray, opt_v: np.ndarray, opt_count: int, hidden_size: int, depth: int,
              state: dict, config: dict, extras: Optional[Dict[str, np.ndarray]] = None, schedule_count: Optional[int] = None, extra_obs=(0, 0),
              compresslevel: int = 1) -> None:
    """Write `ckpt.bin`. `extras` (name -> array) are this build's resume payload (kbj_* members). `schedule_count`: with a learning-rate
    schedule (train.py:1067-1077, either branch) optax's state tree ends in a ScaleByScheduleState(count) leaf - (count, mu.., nu.., count);
    pass the optimizer-step count to write that trailing leaf so that the reference's optimizer tree has as many leaves as the file.
    `compresslevel`: gzip level of the container (any level reads back the same). The payload is ~150 MB of fp32 at 8192 envs, which deflate
    barely shrinks: level 9 (tarfile's default) costs 6-10 s per save, level 1 about a second - and a training loop saves every minute."""
    model = [a for _, a in split_leaves(np.asarray(params), hidden_size, depth, extra_obs)]
    mu = [a for _, a in split_leaves(np.asarray(opt_m), hidden_size, depth, extra_obs)]
    nu = [a for _, a in split_leaves(np.asarray(opt_v), hidden_size, depth, extra_obs)]
    import os
    tmp = path + ".tmp"        # never leave a truncated ckpt.bin behind: write beside it, flush to disk, then rename over it
    with open(tmp, "wb") as fh, tarfile.open(fileobj=fh, mode="w:gz", compresslevel=compresslevel) as tar:
        _add(tar, "model_0", _npy_blobs(model))
        tail = [] if schedule_count is None else [np.asarray(schedule_count, np.int32)]
        _add(tar, "opt_state_0", _npy_blobs([np.asarray(opt_count, np.int32)] + mu + nu + tail))
        _add(tar, "state", json.dumps(state).encode())
        _add(tar, "config", _yaml(config).encode())
        for k, v in (extras or {}).items():
            _add(tar, "kbj_" + k, _npy_blobs([v]))
        tar.close()
        fh.flush()
        os.fsync(fh.fileno())
    os.replace(tmp, path)


def load_ckpt(path: str, part: str = "all", hidden_size: Optional[int] = None, depth: int = 2):
    """convert.py:39 `load_ckpt(path, part="model")`: part in {"model", "opt_state", "state", "config", "all"}.
    "model" -> the flat fp32 parameter vector (equinox leaf order); "all" -> dict with every member decoded."""
    import os
    if not os.path.exists(path):
        raise FileNotFoundError(path)            # convert.py:33-34 error behaviour
    with tarfile.open(path, "r:gz") as tar:
        members = {m.name: tar.extractfile(m).read() for m in tar.getmembers() if m.isfile()}
    config = None
    if "config" in members:
        try:
            import yaml
            config = yaml.safe_load(members["config"].decode())
        except Exception:
            config = json.loads(members["config"].decode())
    if hidden_size is None:
        hidden_size = int((config or {}).get("hidden_size", 0)) or None
        depth = int((config or {}).get("depth", depth))
    if hidden_size is None:
        raise ValueError("hidden_size is neither given nor stored in the checkpoint's config")
    # user observation columns (widened input projections): stored in the config member by the task that wrote the file
    extra_obs = (int((config or {}).get("extra_actor_obs", 0) or 0), int((config or {}).get("extra_critic_obs", 0) or 0))
    nleaf = len(L.param_leaves(hidden_size, depth, extra_obs))

    def model():
        return join_leaves(_read_blobs(members["model_0"]), hidden_size, depth, extra_obs)

    def opt_state():
        """optax state leaves: adam / adamw = ScaleByAdamState(count, mu, nu) [+ ...]; with a schedule (train.py:1067-1077) optax adds a
        second integer `count` leaf (ScaleByScheduleState). Integer scalars are counters wherever they sit; the float leaves are mu then
        nu in parameter order. Returns None when the member is absent or does not have that shape (model-only use still works)."""
        if "opt_state_0" not in members:
            return None
        blobs = _read_blobs(members["opt_state_0"])
        is_count = lambda b: np.asarray(b).size == 1 and np.issubdtype(np.asarray(b).dtype, np.integer)   # optax counts are 0-d int32 (saved here as shape (1,))
        counts = [int(np.asarray(b).reshape(-1)[0]) for b in blobs if is_count(b)]
        arrays = [b for b in blobs if not is_count(b)]
        if len(arrays) != 2 * nleaf:
            return None
        try:
            return dict(count=counts[0] if counts else 0, counts=counts, mu=join_leaves(arrays[:nleaf], hidden_size, depth, extra_obs), nu=join_leaves(arrays[nleaf:], hidden_size, depth, extra_obs))
        except ValueError:
            return None

    if part == "model":
        return model()
    if part == "opt_state":
        return opt_state()
    state = json.loads(members["state"].decode()) if "state" in members else {}
    if part == "state":
        return state
    if part == "config":
        return config
    if part != "all":
        raise ValueError(f"unknown part {part!r}")
    extras = {k[4:]: _read_blobs(v, 1)[0] for k, v in members.items() if k.startswith("kbj_")}
    return dict(model=model(), opt_state=opt_state(), state=state, config=config, extras=extras)


def has_member(path: str, name: str) -> bool:
    with tarfile.open(path, "r:gz") as tar:
        return any(m.name == name for m in tar.getmembers())
