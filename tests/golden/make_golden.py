"""Regenerates tests/golden/oracle_rollout.npz: a small teacher-forced oracle rollout (fp64) used as a regression pin.

The reference ships no golden vectors and cannot be imported offline, so these vectors come from the
build's own CPU oracle: they pin the oracle (and through the parity tests the HIP path) against silent drift, they do NOT
pin parity with the JAX reference (unpinned)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kbot_joystick_amd.spec import compiler, layout as L   # noqa: E402
from oracle import oracle as O   # noqa: E402


def make():
    model = compiler.load_model("kbot-headless")
    cfg = L.default_config(num_envs=4, batch_size=4)
    o = O.Oracle(model, cfg, seed=0, precision="f64")
    a, c, x = o.reset_all()
    rng = np.random.default_rng(123)
    T = 16
    acts = (np.tile(np.array(model.joint_bias, np.float32), (T, 4, 1)) + rng.normal(size=(T, 4, 20)).astype(np.float32) * 0.2)
    aux = np.zeros((T + 1, 4, L.AUX["SIZE"]), np.float32)
    actor = np.zeros((T + 1, 4, L.LD_ACTOR), np.float32)
    critic = np.zeros((T + 1, 4, L.LD_CRITIC), np.float32)
    actor[0], critic[0], aux[0] = a, c, x
    for t in range(T):
        actor[t + 1], critic[t + 1], aux[t + 1] = o.step(acts[t], aux[t])
    rew, comps = o.rewards(aux[:T])
    return dict(actions=acts, actor=actor, critic=critic, aux=aux, reward=rew, comps=comps, es=o.es.copy(), ep0=o.ep.copy())


if __name__ == "__main__":
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_rollout.npz"), **make())
