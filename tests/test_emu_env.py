"""CPU tests of the HIP kernel BODY compiled as a host emulation (tests/emu), against the oracle.

The emulation executes the same phases, the same arrow-matrix Cholesky and the same butterfly reduction order
as the GPU kernel, so this pins the lane-parallel algorithm without a GPU; the GPU run of the real kernel is
tests/test_gpu_env.py.
"""
import ctypes as C

import numpy as np
import pytest

from kbot_joystick_amd.spec import layout as L
from oracle import oracle as O
from tests import helpers as H


@pytest.mark.parametrize("which", ["kbot-headless", "kbot"])
def test_reset_matches_oracle(which):
    from kbot_joystick_amd.spec import compiler
    m = compiler.load_model(which)
    emu = H.emu_lib()
    N = 16
    cfg = L.default_config(num_envs=N)
    o = O.Oracle(m, cfg, seed=5, precision="f32")
    a0, c0, x0 = o.reset_all()
    ep, es = np.zeros_like(o.ep), np.zeros_like(o.es)
    a1, c1, x1 = o.new_obs()
    emu.kbj_emu_reset_all(C.byref(m), C.byref(cfg), C.c_uint32(5), H.fptr(ep), H.fptr(es), H.fptr(a1), H.fptr(c1), H.fptr(x1))
    assert np.array_equal(o.ep, ep)                                  # every draw is a single fp32 rounding
    assert np.array_equal(o.es[:, :54], es[:, :54])                  # qpos, qvel bit-exact
    assert np.array_equal(o.es[:, 80:125].view(np.uint32), es[:, 80:125].view(np.uint32))
    assert np.array_equal(o.es[:, 128:].view(np.uint32), es[:, 128:].view(np.uint32))
    assert np.abs(o.es[:, 125:128] - es[:, 125:128]).max() < 1e-6      # lagged projected gravity: the kinematics compose rotations in another order
    assert np.abs(a0 - a1).max() < 1e-5
    assert (np.abs(c0 - c1) / (1 + np.abs(c0))).max() < 1e-4
    assert np.abs(x0 - x1).max() < 1e-5


@pytest.mark.parametrize("command_mode", [0, 2])
def test_teacher_forced_steps_match_oracle(model, command_mode):
    """command_mode 2: the sampler (and PlaneXYPositionReset) on jax.random's key handling (a25) - the kernel body's integer key derivation and
    two-rounding uniforms against the oracle's, bit for bit (the command block below), with a switch probability that exercises the branch."""
    emu = H.emu_lib()
    N = 48
    cfg = L.default_config(num_envs=N, command_mode=command_mode, **(dict(switch_prob=0.2) if command_mode == 2 else {}))
    o = O.Oracle(model, cfg, seed=11, precision="f32")
    a0, c0, x0 = o.reset_all()
    ep, es = np.zeros_like(o.ep), np.zeros_like(o.es)
    a1, c1, x1 = o.new_obs()
    rng = np.random.default_rng(0)
    errs = {k: [] for k in H.TOL}
    ndone = 0
    for t in range(40):
        act = H.random_actions(model, rng, N)
        auxo, auxe = x0.copy(), x0.copy()
        ep[:], es[:] = o.ep, o.es                                    # teacher forcing: start from the oracle's state
        a0, c0, x0 = o.step(act, auxo)
        emu.kbj_emu_env_step(C.byref(model), C.byref(cfg), C.c_uint32(11), H.fptr(ep), H.fptr(es), H.fptr(act), H.fptr(auxe),
                             H.fptr(a1), H.fptr(c1), H.fptr(x1))
        assert np.array_equal(auxo[:, L.AUX["DONE"]], auxe[:, L.AUX["DONE"]])
        ndone += int((auxo[:, L.AUX["DONE"]] != 0).sum())
        # integer bookkeeping is exact
        assert np.array_equal(o.es[:, 122:125], es[:, 122:125])
        assert np.array_equal(o.es[:, 128:130].view(np.uint32), es[:, 128:130].view(np.uint32))
        assert np.array_equal(o.es[:, 100:116], es[:, 100:116])      # commands
        for k, v in H.state_errors(o.es, es).items():
            errs[k].append(v)
        assert np.median(np.abs(a0 - a1).max(1)) < 1e-4
    assert ndone > 0                                                 # the reset path was exercised
    H.check_error_distribution(errs, label="emu vs oracle ")


def test_yardstick_oracle_fp32_vs_fp64(model):
    """The tolerance table is the oracle's own fp32-vs-fp64 spread: check the yardstick itself."""
    N = 48
    cfg = L.default_config(num_envs=N)
    o = O.Oracle(model, cfg, seed=11, precision="f32")
    o64 = O.Oracle(model, cfg, seed=11, precision="f64")
    a0, c0, x0 = o.reset_all()
    o64.reset_all()
    rng = np.random.default_rng(0)
    errs = {k: [] for k in H.TOL}
    for t in range(40):
        act = H.random_actions(model, rng, N)
        o64.ep[:], o64.es[:] = o.ep, o.es
        a0, c0, x0 = o.step(act, x0.copy())
        o64.step(act, x0.copy())
        for k, v in H.state_errors(o.es, o64.es).items():
            errs[k].append(v)
    H.check_error_distribution(errs, label="oracle f32 vs f64 ")
