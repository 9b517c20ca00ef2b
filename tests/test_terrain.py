"""Sine terrain (BASELINE configs[4]): z = A sin(2 pi x / L) sin(2 pi y / L), capsule ends against the
tangent plane of the surface below their centres, contact frame = (surface normal, world-x tangent, n x t1).
The surface is this build's own definition (the reference's "sine" scene lives in the un-vendored mujoco-scenes fork);
these tests pin the oracle's terrain path by independent checks and the kernel body / GPU kernel against the oracle."""
import ctypes as C

import numpy as np
import pytest

from kbot_joystick_amd.spec import compiler, layout as L
from oracle import oracle as O
from tests import helpers as H

AMP, WL = 0.05, 2.0


def _h(x, y, amp=AMP, wl=WL):
    k = 2 * np.pi / wl
    return amp * np.sin(k * x) * np.sin(k * y)


def _cfg(**kw):
    return L.default_config(terrain_amp=AMP, terrain_wavelength=WL, **kw)


def test_reset_places_robot_on_the_surface():
    m = compiler.load_model("kbot")
    cfg = _cfg(num_envs=64, reset_xy_range=3.0)          # spread the robots over several terrain periods
    o = O.Oracle(m, cfg, seed=3, precision="f64")
    o.reset_all()
    x, y, z = o.es[:, 0], o.es[:, 1], o.es[:, 2]
    stencil = np.stack([_h(x + dx, y + dy) for dx, dy in ((0, 0), (0.15, 0), (-0.15, 0), (0, 0.15), (0, -0.15))])
    assert np.abs(z - (m.qpos0[2] + stencil.max(0))).max() < 1e-6
    assert np.ptp(z) > 0.5 * AMP                           # the terrain really moves the spawn height


def test_robot_is_carried_by_the_terrain(quiet_cfg):
    """PD-held neutral pose dropped on a slope: the feet stop at the surface and the touch sensors carry about the weight."""
    m = compiler.load_model("kbot")
    quiet_cfg.terrain_amp, quiet_cfg.terrain_wavelength = AMP, WL
    quiet_cfg.num_envs = 8
    quiet_cfg.reset_joint_vel_scale = quiet_cfg.reset_joint_pos_scale = quiet_cfg.reset_base_vel_xy_scale = 0.0
    quiet_cfg.reset_xy_range = 2.0
    o = O.Oracle(m, quiet_cfg, seed=1, precision="f64")
    a, c, x = o.reset_all()
    act = np.tile(np.array(m.joint_bias, np.float32), (8, 1))
    tot = []
    for _ in range(30):
        aux = x.copy()
        a, c, x = o.step(act, aux)
        tot.append(c[:, 65] + c[:, 66])
    alive = x[:, L.AUX["DONE"]] == 0
    weight = m.total_mass * 9.81
    load = np.mean(tot[18:28], axis=0)
    assert np.isfinite(o.es).all()
    # feet positions (critic obs 67:73 are base-relative, yaw-rotated): use base height above the local surface instead
    zrel = o.es[:, 2] - _h(o.es[:, 0], o.es[:, 1])
    assert (zrel[alive] > 0.6).all() and (zrel[alive] < m.qpos0[2] + 0.05).all()      # neither fell through nor floats
    assert np.median(np.abs(load - weight) / weight) < 0.15                             # normal forces ~ weight on a <= 9 deg slope


def test_flat_limit_is_continuous(model):
    """terrain_amp -> 0 through the terrain code path reproduces the plane path."""
    N = 16
    o0 = O.Oracle(model, L.default_config(num_envs=N), seed=2, precision="f64")
    o1 = O.Oracle(model, L.default_config(num_envs=N, terrain_amp=1e-9, terrain_wavelength=WL), seed=2, precision="f64")
    _, _, x0 = o0.reset_all()
    _, _, x1 = o1.reset_all()
    rng = np.random.default_rng(0)
    for _ in range(5):
        act = H.random_actions(model, rng, N)
        a0, c0, x0n = o0.step(act, x0.copy())
        a1, c1, x1n = o1.step(act, x1.copy())
        x0, x1 = x0n, x1n
    assert np.abs(o0.es[:, :54] - o1.es[:, :54]).max() < 1e-5


def test_terrain_yardstick_fp32_vs_fp64():
    m = compiler.load_model("kbot")
    N = 32
    cfg = _cfg(num_envs=N, reset_xy_range=2.0)
    o, o64 = O.Oracle(m, cfg, seed=11, precision="f32"), O.Oracle(m, cfg, seed=11, precision="f64")
    a0, c0, x0 = o.reset_all()
    o64.reset_all()
    rng = np.random.default_rng(0)
    errs = {k: [] for k in H.TOL}
    for t in range(25):
        act = H.random_actions(m, rng, N)
        o64.ep[:], o64.es[:] = o.ep, o.es
        a0, c0, x0n = o.step(act, x0.copy())
        o64.step(act, x0.copy())
        x0 = x0n
        for k, v in H.state_errors(o64.es, o.es).items():
            errs[k].append(v)
    H.check_error_distribution(errs, label="terrain f32 vs f64 ")


def test_terrain_kernel_body_matches_oracle():
    """Host emulation of the HIP kernel body on the terrain config (full kbot model, teacher forced)."""
    m = compiler.load_model("kbot")
    emu = H.emu_lib()
    N = 32
    cfg = _cfg(num_envs=N, reset_xy_range=2.0)
    o = O.Oracle(m, cfg, seed=7, precision="f32")
    a0, c0, x0 = o.reset_all()
    ep, es = np.zeros_like(o.ep), np.zeros_like(o.es)
    a1, c1, x1 = o.new_obs()
    emu.kbj_emu_reset_all(C.byref(m), C.byref(cfg), C.c_uint32(7), H.fptr(ep), H.fptr(es), H.fptr(a1), H.fptr(c1), H.fptr(x1))
    assert np.array_equal(o.ep, ep) and np.array_equal(o.es[:, :54], es[:, :54])
    rng = np.random.default_rng(0)
    errs = {k: [] for k in H.TOL}
    for t in range(25):
        act = H.random_actions(m, rng, N)
        auxo, auxe = x0.copy(), x0.copy()
        ep[:], es[:] = o.ep, o.es
        a0, c0, x0 = o.step(act, auxo)
        emu.kbj_emu_env_step(C.byref(m), C.byref(cfg), C.c_uint32(7), H.fptr(ep), H.fptr(es), H.fptr(act), H.fptr(auxe),
                             H.fptr(a1), H.fptr(c1), H.fptr(x1))
        assert np.array_equal(auxo[:, L.AUX["DONE"]], auxe[:, L.AUX["DONE"]])
        for k, v in H.state_errors(o.es, es).items():
            errs[k].append(v)
    H.check_error_distribution(errs, label="terrain emu vs oracle ")


@pytest.mark.gpu
@pytest.mark.parametrize("N,steps", [(64, 25), (8192, 6)])     # the second case = the per-GPU env count of BASELINE configs[4]
def test_gpu_terrain_steps_match_oracle(N, steps):
    """BASELINE configs[4] shape: full kbot, sine terrain, UnifiedCommand sampler, randomisers + pushes on — HIP env kernel vs
    the fp32 oracle, teacher forced from the oracle's state each step, through the C ABI."""
    import torch
    from kbot_joystick_amd.host import binding as Bd
    m = compiler.load_model("kbot")
    cfg = _cfg(num_envs=N, batch_size=32, reset_xy_range=2.0)
    ctx = Bd.Context(m, cfg, 0, torch.cuda.current_stream().cuda_stream)
    o = O.Oracle(m, cfg, seed=9, precision="f32")
    a0, c0, x0 = o.reset_all()
    dev = "cuda:0"
    act_o, crit_o, aux_o = (torch.zeros(N, L.LD_ACTOR, device=dev), torch.zeros(N, L.LD_CRITIC, device=dev), torch.zeros(N, L.AUX["SIZE"], device=dev))
    ctx.env_reset_all(9, act_o, crit_o, aux_o)
    ctx.synchronize()
    ep, es = ctx.env_get_state()
    assert np.array_equal(o.ep, ep)
    assert np.abs(o.es[:, :54] - es[:, :54]).max() < 1e-6      # sinf/cosf differ by an ulp between host and device
    rng = np.random.default_rng(0)
    errs = {k: [] for k in H.TOL}
    aux_t = torch.zeros(N, L.AUX["SIZE"], device=dev)
    obs = []
    for t in range(steps):
        act = H.random_actions(m, rng, N)
        auxo = x0.copy()
        ctx.env_set_state(o.ep, o.es)
        aux_t.copy_(torch.from_numpy(x0))
        a0, c0, x0 = o.step(act, auxo)
        ctx.env_step(torch.from_numpy(act).to(dev), aux_t, act_o, crit_o, aux_o)
        ctx.synchronize()
        ep, es = ctx.env_get_state()
        assert np.array_equal(auxo[:, L.AUX["DONE"]], aux_t.cpu().numpy()[:, L.AUX["DONE"]])
        assert np.array_equal(o.ep, ep) and np.array_equal(o.es[:, 80:125], es[:, 80:125])     # parameters, commands, pushes, counters: exact
        run = auxo[:, L.AUX["DONE"]] == 0
        for k, v in H.state_errors(o.es, es).items():
            errs[k].append(v[run])
        obs.append(np.abs(a0 - act_o.cpu().numpy()).max(1)[run])
    obs = np.concatenate(obs)
    assert np.median(obs) < 5e-6 and np.quantile(obs, 0.99) < 5e-5        # measured vs fp64: median 1.4e-6, p99 9.3e-6
    # terrain + full kbot, measured (profiles/parity_r02.json): p99 qpos 1.1e-6 / qvel 5.7e-6 / qacc 1.9e-5, oracle fp32 1.25e-6 / 6.5e-6 / 2.2e-5
    tol = dict(qpos=(5e-7, 2.5e-6, 5e-6, 0.1), qvel=(2e-6, 1.3e-5, 3e-5, 1.0), qacc=(8e-6, 4.5e-5, 1e-4, 4.0))
    H.check_error_distribution(errs, tol=tol, label="terrain gpu vs oracle ")
    ctx.close()
