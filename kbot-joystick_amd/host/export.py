"""Deployment hand-off of the trained actor (the reference's convert.py side path).

convert.py:43-119 builds its kinfer model from `model.actor`, the 16 command names, the joint order and a flat carry of
`depth * 2 * hidden + 20` floats (LSTM h, c per layer, then the low-pass state). The xax `ckpt.bin` container itself cannot be
reproduced offline (format not in the tree), so the actor is written as a numpy archive with exactly those pieces, leaf names as
equinox prints them."""
from __future__ import annotations

import numpy as np

from ..spec import constants, layout as L


def actor_leaves(params: np.ndarray, hidden_size: int, depth: int = 2, extra_obs=(0, 0)) -> dict:
    """Slice the flat fp32 parameter vector into the actor's named leaves."""
    out, off = {}, 0
    for name, shape in L.param_leaves(hidden_size, depth, extra_obs):
        n = int(np.prod(shape))
        if name.startswith("actor."):
            out[name] = np.asarray(params[off:off + n], np.float32).reshape(shape)
        off += n
    if off != params.size:
        raise ValueError(f"parameter vector has {params.size} floats, layout needs {off}")
    return out


def export_actor(path: str, params: np.ndarray, hidden_size: int, depth: int, ctrl_dt: float, cutoff_frequency: float,
                 min_std: float, max_std: float, var_scale: float, joint_biases, extra_obs=(0, 0)) -> None:
    """Write the deployable actor: leaves + the constants its forward needs (train.py:913-941) + the I/O contract of convert.py."""
    leaves = actor_leaves(np.asarray(params), hidden_size, depth, extra_obs)
    meta = dict(
        joint_names=np.array(constants.JOINT_NAMES), command_names=np.array(constants.COMMAND_NAMES), step_fn_inputs=np.array(constants.STEP_FN_INPUTS),
        joint_biases=np.asarray(joint_biases, np.float32), hidden_size=hidden_size, depth=depth,
        carry_size=depth * 2 * hidden_size + L.NU, num_inputs=L.NOBS_ACTOR + extra_obs[0], num_outputs=2 * L.NU, ctrl_dt=ctrl_dt,
        cutoff_frequency=cutoff_frequency, min_std=min_std, max_std=max_std, var_scale=var_scale)
    np.savez(path, **leaves, **{f"meta.{k}": v for k, v in meta.items()})
