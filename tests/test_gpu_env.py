"""GPU parity tests of the env kernels (kbj_env_reset_all / kbj_env_step / kbj_rewards) through the C ABI,
against the CPU oracle on identical seeds. Run with `pytest -m gpu` on an MI355X."""
import numpy as np
import pytest

from kbot_joystick_amd.spec import compiler, layout as L
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _ctx(model, cfg):
    import torch
    from kbot_joystick_amd.host import binding as B
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return B.Context(model, cfg, device=0, stream=torch.cuda.current_stream().cuda_stream), torch


def _obs(torch, N):
    dev = "cuda:0"
    return (torch.zeros(N, L.LD_ACTOR, device=dev), torch.zeros(N, L.LD_CRITIC, device=dev), torch.zeros(N, L.AUX["SIZE"], device=dev))


@pytest.mark.parametrize("which", ["kbot-headless", "kbot"])
def test_reset_matches_oracle(which):
    from oracle import oracle as O
    m = compiler.load_model(which)
    N = 64
    cfg = L.default_config(num_envs=N, batch_size=min(512, N))
    ctx, torch = _ctx(m, cfg)
    a, c, x = _obs(torch, N)
    ctx.env_reset_all(5, a, c, x)
    ctx.synchronize()
    ep, es = ctx.env_get_state()
    o = O.Oracle(m, cfg, seed=5, precision="f32")
    a0, c0, x0 = o.reset_all()
    assert np.array_equal(o.ep, ep)
    assert np.array_equal(o.es[:, 0:3], es[:, 0:3]) and np.array_equal(o.es[:, 7:27], es[:, 7:27])
    assert np.abs(o.es[:, 3:7] - es[:, 3:7]).max() < 3e-7           # cosf/sinf(yaw/2): device libm vs glibc, 1-2 ulp
    assert np.abs(o.es[:, 28:54] - es[:, 28:54]).max() == 0
    assert np.array_equal(o.es[:, 80:125], es[:, 80:125])
    assert np.array_equal(o.es[:, 128:130].view(np.uint32), es[:, 128:130].view(np.uint32))
    assert np.abs(a0 - a.cpu().numpy()).max() < 1e-4
    assert (np.abs(c0 - c.cpu().numpy()) / (1 + np.abs(c0))).max() < 1e-3
    assert np.abs(x0 - x.cpu().numpy()).max() < 1e-4
    ctx.close()


@pytest.mark.parametrize("N,steps,command", [(128, 30, "sampler"), (8192, 12, "sampler"), (8192, 6, "fixed"), (256, 20, "sampler on jax.random keys")])
def test_teacher_forced_steps_match_oracle(model, N, steps, command):
    """One control step from the identical state, HIP vs the oracle: discrete results exact, continuous state within the measured
    tolerances (tests/helpers.TOL) and, at the BASELINE env count, within 2x the oracle's OWN fp32-vs-fp64 spread on the same
    env-steps. command = "fixed": BASELINE configs[1] (command_mode 1, (0.5, 0, 0)) - what bench.py times."""
    from oracle import oracle as O
    kw = dict(command_mode=1, fixed_command=[0.5] + [0.0] * 15) if command == "fixed" else {}
    if command == "sampler on jax.random keys":      # a25: kbj_config.command_mode = 2 (the command block and the reset's x, y are compared bit for bit below)
        kw = dict(command_mode=2, switch_prob=0.2)
    cfg = L.default_config(num_envs=N, batch_size=min(512, N), **kw)
    ctx, torch = _ctx(model, cfg)
    a, c, x = _obs(torch, N)
    a2, c2, x2 = _obs(torch, N)
    ctx.env_reset_all(11, a, c, x)
    full = N >= 4096
    o = O.Oracle(model, cfg, seed=11, precision="f32")
    o64 = O.Oracle(model, cfg, seed=11, precision="f64") if full else None
    a0, c0, x0 = o.reset_all()
    assert np.abs(a0 - a.cpu().numpy()).max() < 1e-4
    if command == "fixed":
        assert np.array_equal(o.es[:, 100:116], np.tile(np.array([0.5] + [0.0] * 15, np.float32), (N, 1)))
    rng = np.random.default_rng(0)
    errs, errs_o32, errs64, switch = {k: [] for k in H.TOL}, {k: [] for k in H.TOL}, {k: [] for k in H.TOL}, []
    obs_a, obs_c, ndone = [], [], 0
    cd_err, touch_err = [], []
    for t in range(steps):
        act = H.random_actions(model, rng, N)
        ep0, es0 = o.ep.copy(), o.es.copy()
        ctx.env_set_state(ep0, es0)                                  # teacher forcing
        aux_t = torch.from_numpy(x0.copy()).cuda()
        auxo, aux64 = x0.copy(), x0.copy()
        if full:
            o64.ep[:], o64.es[:] = ep0, es0
            _, _, _, d64 = o64.step_diag(act, aux64)
            a0, c0, x0, d32 = o.step_diag(act, auxo)
        else:
            a0, c0, x0 = o.step(act, auxo)
        ctx.env_step(torch.from_numpy(act).cuda(), aux_t, a2, c2, x2)
        ctx.synchronize()
        ep, es = ctx.env_get_state()
        auxe = aux_t.cpu().numpy()
        assert np.array_equal(auxo[:, L.AUX["DONE"]], auxe[:, L.AUX["DONE"]])
        ndone += int((auxo[:, L.AUX["DONE"]] != 0).sum())
        assert np.array_equal(o.es[:, 122:125], es[:, 122:125])                                       # push / time counters
        assert np.array_equal(o.es[:, 128:130].view(np.uint32), es[:, 128:130].view(np.uint32))       # episode / step counters
        assert np.array_equal(o.es[:, 100:116], es[:, 100:116])                                       # command
        assert np.array_equal(o.es[:, 80:100], es[:, 80:100])                                         # ACT_PREV: latency / drop (a3)
        assert np.array_equal(o.es[:, 116:122], es[:, 116:122])                                       # push wrench (a22)
        assert np.array_equal(o.ep, ep)
        run = auxo[:, L.AUX["DONE"]] == 0                    # a reset re-draws the state from the RNG: compare the running envs
        for k, v in H.state_errors(o.es, es).items():
            errs[k].append(v[run])
        if full:
            run64 = run & (aux64[:, L.AUX["DONE"]] == 0)
            for k, v in H.state_errors(o64.es, es).items():
                errs64[k].append(v[run64])
            for k, v in H.state_errors(o64.es, o.es).items():
                errs_o32[k].append(v[run64])
            switch.append(((d32 != d64).any(1) | (d64[:, 0] >= cfg.solver_iterations) | (d32[:, 0] >= cfg.solver_iterations))[run64])
        obs_a.append(np.abs(a0 - a2.cpu().numpy()).max(1)[run])
        obs_c.append((np.abs(c0 - c2.cpu().numpy()) / (1 + np.abs(c0))).max(1)[run])
        # com_distance (a17) and touch beyond the reset, compared directly (pre-divergence: same state in, one step)
        nx = x2.cpu().numpy()
        cd_err.append(np.abs(nx[run, L.AUX["COMDIST"]] - x0[run, L.AUX["COMDIST"]]))
        touch_err.append(np.abs(nx[run, L.AUX["TOUCH"]:L.AUX["TOUCH"] + 2] - x0[run, L.AUX["TOUCH"]:L.AUX["TOUCH"] + 2]).max(1))
    assert ndone > 0 or steps < 30                                   # the reset path is exercised in the long case
    # a17 com_distance and the foot touch sensors of the next observation. Measured at 8192 envs x 12 steps against the fp64 oracle
    # (tools/parity_quantiles.py -> profiles/parity_r03.json): com_distance p99 1.0e-7, p99.9 1.4e-7, max 1.1e-3 (the one env-step on a
    # solver switch; the fp32 oracle's own max 1.7e-4); touch (newtons) p99 3.7e-3, p99.9 5.8e-3, max 195 = a contact that closes in one
    # evaluation and not in the other (the fp32 oracle's own: 3.7e-3 / 5.5e-3 / 176). Bounds = ~10x the quantiles, a capped count of
    # outliers for the discrete events, and a hard cap on the extreme value.
    cd, tc = np.concatenate(cd_err), np.concatenate(touch_err)
    assert np.quantile(cd, 0.99) < 2e-6 and np.quantile(cd, 0.999) < 1e-5 and cd.max() < 0.02, (np.quantile(cd, 0.99), np.quantile(cd, 0.999), cd.max())
    assert (cd > 1e-4).sum() <= 2 + cd.size // 20000
    assert np.quantile(tc, 0.99) < 5e-2 and np.quantile(tc, 0.999) < 0.2, (np.quantile(tc, 0.99), np.quantile(tc, 0.999))
    assert (tc > 1.0).sum() <= 3 + tc.size // 5000 and tc.max() < 400.0, ((tc > 1.0).sum(), tc.max())      # 400 N = the robot's weight
    oa, oc = np.concatenate(obs_a), np.concatenate(obs_c)
    # observation rows (measured vs fp64: actor median 1.2e-6, p99 6.7e-6; critic (relative) median 4.5e-6, p99 2.6e-5)
    assert np.median(oa) < 5e-6 and np.quantile(oa, 0.99) < 3e-5 and np.median(oc) < 2e-5 and np.quantile(oc, 0.99) < 1e-4
    H.check_error_distribution(errs, label="hip vs oracle fp32 ")
    if full:
        H.check_against_oracle_spread(errs64, errs_o32, switch, label="hip vs oracle fp64 ")
    ctx.close()


def test_free_running_rollout_statistics(model):
    """Without teacher forcing trajectories diverge chaotically, but episode statistics must agree."""
    from oracle import oracle as O
    N = 256
    cfg = L.default_config(num_envs=N, batch_size=min(512, N))
    ctx, torch = _ctx(model, cfg)
    a, c, x = _obs(torch, N)
    a2, c2, x2 = _obs(torch, N)
    ctx.env_reset_all(3, a, c, x)
    o = O.Oracle(model, cfg, seed=3, precision="f32")
    a0, c0, x0 = o.reset_all()
    rng = np.random.default_rng(1)
    T = 60
    aux_g = torch.zeros(T + 1, N, L.AUX["SIZE"], device="cuda:0")
    aux_g[0] = x
    aux_o = np.zeros((T + 1, N, L.AUX["SIZE"]), np.float32)
    aux_o[0] = x0
    for t in range(T):
        act = H.random_actions(model, rng, N, 0.1)
        a0, c0, nx = o.step(act, aux_o[t])
        aux_o[t + 1] = nx
        ctx.env_step(torch.from_numpy(act).cuda(), aux_g[t], a2, c2, aux_g[t + 1])
    ctx.synchronize()
    g = aux_g.cpu().numpy()
    # first steps agree closely, later only statistically
    assert np.abs(g[0] - aux_o[0]).max() < 1e-4
    assert np.median(np.abs(g[3, :, 0:6] - aux_o[3, :, 0:6])) < 1e-3
    done_g, done_o = (g[:T, :, 70] != 0).mean(), (aux_o[:T, :, 70] != 0).mean()
    assert abs(done_g - done_o) < 0.3 * max(done_o, 1e-3) + 2e-3
    assert abs(g[:T, :, 51:53].mean() - aux_o[:T, :, 51:53].mean()) < 0.1 * aux_o[:T, :, 51:53].mean()
    # rewards kernel on the oracle's trajectory: bit-for-bit same inputs -> tight tolerance
    rew = torch.zeros(T, N, device="cuda:0")
    comps = torch.zeros(T, N, L.NREW, device="cuda:0")
    ctx.rewards(torch.from_numpy(aux_o[:T].copy()).cuda(), T, rew, comps)
    ctx.synchronize()
    r_o, c_o = o.rewards(aux_o[:T])
    assert np.abs(comps.cpu().numpy() - c_o).max() < 2e-4
    assert np.abs(rew.cpu().numpy() - r_o).max() < 2e-4
    ctx.close()


def test_full_size_properties(model):
    """BASELINE config size (8192 envs): invariants that do not need the oracle."""
    N = 8192
    cfg = L.default_config(num_envs=N, batch_size=min(512, N))
    ctx, torch = _ctx(model, cfg)
    a, c, x = _obs(torch, N)
    a2, c2, x2 = _obs(torch, N)
    ctx.env_reset_all(1, a, c, x)
    act = torch.from_numpy(np.tile(np.array(model.joint_bias, np.float32), (N, 1))).cuda()
    for t in range(5):
        ctx.env_step(act, x, a2, c2, x2)
        x, x2 = x2, x
    ctx.synchronize()
    ep, es = ctx.env_get_state()
    assert np.isfinite(es[:, :125]).all() and np.isfinite(a2.cpu().numpy()).all() and np.isfinite(c2.cpu().numpy()).all()
    assert np.abs(np.linalg.norm(es[:, 3:7], axis=1) - 1).max() < 1e-5
    assert (es[:, 129].view(np.uint32) == 5).all()
    # determinism: same seed, same actions -> identical bits
    ctx2, _ = _ctx(model, cfg)
    b, d, y = _obs(torch, N)
    b2, d2, y2 = _obs(torch, N)
    ctx2.env_reset_all(1, b, d, y)
    for t in range(5):
        ctx2.env_step(act, y, b2, d2, y2)
        y, y2 = y2, y
    ctx2.synchronize()
    ep2, es2 = ctx2.env_get_state()
    assert np.array_equal(es.view(np.uint32), es2.view(np.uint32))
    ctx.close(); ctx2.close()


def test_register_solver_matches_lds_formulation(model):
    """The product's constraint solver keeps the arrow-matrix LDL^T and the Newton loop in registers (DPP pivots, an INCREMENTAL Hessian
    that adds and removes +-D contributions as the active set changes); `tests/emu` runs the other formulation of the same solver (vectors
    in LDS, Hessian rebuilt every iteration) on the host. This test closes the gap on the hardware: libkbj_ldssolver.so is the same library
    built with that LDS formulation (`make -C kbot-joystick_amd/csrc ldssolver`); both builds step the same envs from IDENTICAL states
    (teacher forcing: the product's state is copied into the A/B context before every control step - 5 substeps, up to 40 Newton
    iterations each), so a drift of the incremental Hessian or a wrong pivot shows as a state difference after one control step.
    Discrete outcomes (done flags) must agree, the continuous state to rounding level on almost every env (contact-set flips at a
    threshold are legitimate and bounded)."""
    import os
    import subprocess
    import torch
    from kbot_joystick_amd.host import binding as B
    csrc = os.path.dirname(B.LIB_PATH)
    subprocess.check_call(["make", "-C", csrc, "-s", "ldssolver"])
    lds = B.load_library_at(os.path.join(csrc, "libkbj_ldssolver.so"))
    N, steps = 2048, 40
    cfg = L.default_config(num_envs=N, batch_size=512)
    stream = torch.cuda.current_stream().cuda_stream
    prod = B.Context(model, cfg, device=0, stream=stream)
    ab = B.Context(model, cfg, device=0, stream=stream, lib=lds)
    a, c, x = _obs(torch, N)
    a2, c2, x2 = _obs(torch, N)
    b2, d2, y2 = _obs(torch, N)
    prod.env_reset_all(5, a, c, x)
    ab.env_reset_all(5, b2, d2, y2)
    rng = np.random.default_rng(3)
    dq, dv, flips, dones = [], [], 0, 0
    for t in range(steps):
        ep, es = prod.env_get_state()
        ab.env_set_state(ep, es)
        act = torch.from_numpy(H.random_actions(model, rng, N)).cuda()
        xa, xb = x.clone(), x.clone()
        prod.env_step(act, xa, a2, c2, x2)
        ab.env_step(act, xb, b2, d2, y2)
        torch.cuda.synchronize()
        _, es1 = prod.env_get_state()
        _, es2 = ab.env_get_state()
        done1, done2 = xa[:, L.AUX["DONE"]].cpu().numpy(), xb[:, L.AUX["DONE"]].cpu().numpy()
        flips += int((done1 != done2).sum())
        dones += int((done1 != 0).sum())
        run = (done1 == 0) & (done2 == 0)                       # a reset env's row is the new episode's state in both builds
        dq.append(np.abs(es1[run, L.ES["QPOS"]:L.ES["QPOS"] + 27] - es2[run, L.ES["QPOS"]:L.ES["QPOS"] + 27]).max(axis=1))
        dv.append(np.abs(es1[run, L.ES["QVEL"]:L.ES["QVEL"] + 26] - es2[run, L.ES["QVEL"]:L.ES["QVEL"] + 26]).max(axis=1))
        x.copy_(x2)
    dq, dv = np.concatenate(dq), np.concatenate(dv)
    stats = dict(n=int(dq.size), dones=dones, done_flips=flips, qpos_p50=float(np.median(dq)), qpos_p99=float(np.quantile(dq, 0.99)), qpos_max=float(dq.max()),
                 qvel_p50=float(np.median(dv)), qvel_p99=float(np.quantile(dv, 0.99)), qvel_max=float(dv.max()))
    print("register vs LDS solver:", stats)
    assert dones > 50                                            # the random policy falls: contact-rich states are in the sample
    assert flips <= max(2, dones // 50)
    # measured on MI355X (81 k env-steps, 825 terminations): qpos p50 6e-8 / p99 4.5e-7 / max 2e-3, qvel p50 5e-6 / p99 4.8e-5 / max 0.5
    assert stats["qpos_p50"] < 1e-6 and stats["qpos_p99"] < 1e-5 and stats["qvel_p50"] < 1e-4 and stats["qvel_p99"] < 1e-3
    assert float((dv > 1e-2).mean()) < 2e-3 and stats["qpos_max"] < 2e-2          # the few contact-set flips at a threshold
    prod.close(); ab.close()
