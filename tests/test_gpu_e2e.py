"""End-to-end GPU parity: a whole (tiny) training iteration through the C ABI — rollout with sampled actions, rewards, GAE,
minibatch BPTT gradients, AdamW — against the CPU oracle trainer on identical seeds, plus the committed golden rollout."""
import os

import numpy as np
import pytest

from kbot_joystick_amd.spec import layout as L

pytestmark = pytest.mark.gpu


def _check_rollout_against_oracle(task, ref, T):
    """Free-running rollout with sampled actions vs the oracle trainer. Steps 0 and 1 start from (nearly) identical states: every
    sample is bounded there (max). Over the whole rollout the trajectories separate slowly (chaotic contact dynamics feed back through
    the policy), so the bulk is bounded by median AND p99, and the discrete outcomes (done flags) must agree exactly."""
    act = task.traj.action.cpu().numpy()
    logp, val, rew = task.traj.logp.cpu().numpy(), task.traj.value.cpu().numpy(), task.traj.reward.cpu().numpy()
    assert np.abs(act[0] - ref["action"][0]).max() < 1e-4 and np.abs(act[1] - ref["action"][1]).max() < 2e-3   # bounded max, steps 0..1
    assert np.abs(logp[:2] - ref["logp"][:2]).max() < 2e-2 and np.abs(val[:2] - ref["value"][:2]).max() < 1e-3
    assert np.abs(rew[:2] - ref["reward"][:2]).max() < 5e-3
    assert np.array_equal(task.traj.aux[:T, :, L.AUX["DONE"]].cpu().numpy(), ref["aux"][:T, :, L.AUX["DONE"]])
    for name, got, want, med, p99 in (("action", act, ref["action"], 1e-4, 5e-2), ("logp", logp, ref["logp"], 1e-3, 5e-1),
                                      ("value", val, ref["value"], 1e-4, 2e-2), ("reward", rew, ref["reward"], 1e-3, 1e-1)):
        d = np.abs(got - want)
        assert np.median(d) < med, (name, "median", float(np.median(d)))
        assert np.quantile(d, 0.99) < p99, (name, "p99", float(np.quantile(d, 0.99)))


def test_config0_full_iteration_matches_oracle():
    """BASELINE configs[0] (`python -m train num_envs=4`, plumbing): kbot-headless, 4 envs, a 64-step rollout, batch 4, the launch
    networks (hidden 256), 3 passes - one whole train_iteration against the oracle trainer."""
    import torch
    from kbot_joystick_amd.host.task import HumanoidWalkingTask, launch_config
    from oracle.trainer import OracleTrainer
    cfg = launch_config(num_envs=4, batch_size=4, rollout_length_seconds=64 * 0.02, robot="kbot-headless", seed=0)
    task = HumanoidWalkingTask(cfg, device=torch.device("cuda", 0))
    assert (task.T, task.H, task.kcfg.num_passes) == (64, 256, 3)
    params0 = task.params.cpu().numpy().copy()
    tr = OracleTrainer(task.model_blob, task.kcfg, seed=0, params=params0, precision="f32")
    task.rollout()
    torch.cuda.synchronize()
    ref = tr.rollout()
    _check_rollout_against_oracle(task, ref, task.T)
    for name in ("actor_obs", "critic_obs", "aux"):
        getattr(task.traj, name).copy_(torch.from_numpy(ref[name]))
    for name in ("action", "logp", "value", "reward"):
        getattr(task.traj, name).copy_(torch.from_numpy(ref[name]))
    perms = []
    for p in range(3):
        g = torch.Generator(device="cpu"); g.manual_seed((cfg.seed * 1000003 + task.iteration * 97 + p) & 0x7FFFFFFF)
        perms.append(torch.randperm(task.N, generator=g).numpy())
    task.update()
    torch.cuda.synchronize()
    tr.update(perms)
    p_gpu, p_ref = task.params.cpu().numpy(), tr.params.numpy()
    moved = np.abs(p_ref - params0).max()
    assert moved > 1e-4 and np.abs(p_gpu - p_ref).max() < 0.02 * moved + 1e-6 and task.opt_step == tr.opt_step == 3
    task.ctx.close()


@pytest.mark.parametrize("H,D", [(64, 2), (128, 1)])     # (128, 1): another point of the reference's model fields (train.py:78-85), end to end
def test_training_iteration_matches_oracle(H, D):
    import torch
    from kbot_joystick_amd.host.task import HumanoidWalkingTask, launch_config
    from oracle.trainer import OracleTrainer
    cfg = launch_config(num_envs=8, batch_size=4, hidden_size=H, depth=D, rollout_length_seconds=0.12, robot="kbot-headless", seed=5, num_passes=2)
    task = HumanoidWalkingTask(cfg, device=torch.device("cuda", 0))
    params0 = task.params.cpu().numpy().copy()
    tr = OracleTrainer(task.model_blob, task.kcfg, seed=5, params=params0, precision="f32")
    # ---- rollout (actions sampled with the same threefry Gaussian draws on both sides) ----
    task.rollout()
    torch.cuda.synchronize()
    ref = tr.rollout()
    T = task.T
    _check_rollout_against_oracle(task, ref, T)
    # ---- update on the ORACLE's trajectory (so both sides differentiate the same data) ----
    for name in ("actor_obs", "critic_obs", "aux"):
        getattr(task.traj, name).copy_(torch.from_numpy(ref[name]))
    task.traj.action.copy_(torch.from_numpy(ref["action"])); task.traj.logp.copy_(torch.from_numpy(ref["logp"]))
    task.traj.value.copy_(torch.from_numpy(ref["value"])); task.traj.reward.copy_(torch.from_numpy(ref["reward"]))
    perms = []
    for p in range(task.kcfg.num_passes):                                         # the permutations HumanoidWalkingTask.update draws
        g = torch.Generator(device="cpu"); g.manual_seed((cfg.seed * 1000003 + task.iteration * 97 + p) & 0x7FFFFFFF)
        perms.append(torch.randperm(task.N, generator=g).numpy())
    task.update()
    torch.cuda.synchronize()
    tr.update(perms)
    p_gpu, p_ref = task.params.cpu().numpy(), tr.params.numpy()
    moved = np.abs(p_ref - params0).max()
    assert moved > 1e-4                                                           # the optimizer really stepped
    assert np.abs(p_gpu - p_ref).max() < 0.02 * moved + 1e-6
    assert task.opt_step == tr.opt_step == 4


def test_hip_env_matches_golden_rollout(model):
    """The committed golden vectors (oracle fp64, tests/golden) — the HIP path from the same seed and actions."""
    import torch
    from kbot_joystick_amd.host import binding as B
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_rollout.npz"))
    cfg = L.default_config(num_envs=4, batch_size=4)
    ctx = B.Context(model, cfg, 0, torch.cuda.current_stream().cuda_stream)
    dev = "cuda:0"
    T = ref["actions"].shape[0]
    actor = torch.zeros(T + 1, 4, L.LD_ACTOR, device=dev); critic = torch.zeros(T + 1, 4, L.LD_CRITIC, device=dev)
    aux = torch.zeros(T + 1, 4, L.AUX["SIZE"], device=dev)
    ctx.env_reset_all(0, actor[0], critic[0], aux[0])
    acts = torch.from_numpy(ref["actions"]).to(dev)
    for t in range(T):
        ctx.env_step(acts[t], aux[t], actor[t + 1], critic[t + 1], aux[t + 1])
    ctx.synchronize()
    ep, es = ctx.env_get_state()
    assert np.abs(actor[0].cpu().numpy() - ref["actor"][0]).max() < 1e-5
    assert np.abs(aux[:4].cpu().numpy() - ref["aux"][:4]).max() < 5e-3          # fp32 GPU vs fp64 oracle, early steps
    assert np.median(np.abs(actor[:8].cpu().numpy() - ref["actor"][:8])) < 1e-5
    rew = torch.zeros(T, 4, device=dev)
    ctx.rewards(torch.from_numpy(ref["aux"][:T].copy()).to(dev), T, rew, None)
    ctx.synchronize()
    assert np.abs(rew.cpu().numpy() - ref["reward"]).max() < 2e-4
    ctx.close()


def test_training_with_mirror_losses_runs():
    """Dataclass-default mirror scales (train.py:115-122): rollout advances the mirror carries, update adds the aux losses."""
    import torch
    from kbot_joystick_amd.host.task import HumanoidWalkingTask, launch_config
    cfg = launch_config(num_envs=64, batch_size=32, hidden_size=64, rollout_length_seconds=0.2, robot="kbot-headless", seed=2, num_passes=1,
                        actor_mirror_loss_scale=1.0, critic_mirror_loss_scale=0.01)
    task = HumanoidWalkingTask(cfg)
    p0 = task.params.clone()
    for _ in range(2):
        task.train_iteration()
    torch.cuda.synchronize()
    m = task.metrics.cpu()
    assert torch.isfinite(m).all() and torch.isfinite(task.params).all()
    assert float(m[8]) > 0 and float(m[9]) >= 0
    assert float(task.carry.actor_mirror_hc.abs().max()) > 0 and float(task.traj.carry0_actor_mirror_hc.abs().max()) > 0
    assert not torch.equal(p0, task.params)
    task.ctx.close()


@pytest.mark.parametrize("mirror,T", [(False, 6), (True, 6), (False, 5)])
def test_pipelined_rollout_equals_stepwise_calls(mirror, T):
    """kbj_rollout runs the actor -> env chain on the caller's stream with the critic (and the mirror branches) on a side lane; the trajectory
    must be bit-identical to the strictly serial order (KBJ_ROLLOUT_PIPELINE=0) and to driving kbj_policy_step / kbj_env_step /
    kbj_carry_reset one full-batch call at a time. The rollout alternates the h planes of
    the carries between the caller's arrays and workspace partners per step (lstm_step_kernel cannot update h in place): an odd T
    ends on the partners and has to be copied home, an even one must not be."""
    import torch
    from kbot_joystick_amd.host import binding as Bd, buffers
    from kbot_joystick_amd.spec import compiler, layout as L
    N, H = 512, 64
    kw = dict(actor_mirror_loss_scale=1.0, critic_mirror_loss_scale=0.01) if mirror else {}
    m = compiler.load_model("kbot-headless")
    cfg = L.default_config(num_envs=N, batch_size=64, rollout_len=T, hidden_size=H, **kw)
    import os
    out = []
    for mode in ("serial", "side lane (default)", "stepwise"):
        os.environ.pop("KBJ_ROLLOUT_PIPELINE", None)
        if mode != "side lane (default)":
            os.environ["KBJ_ROLLOUT_PIPELINE"] = "0"
        ctx = Bd.Context(m, cfg, 0, torch.cuda.current_stream().cuda_stream)
        params = torch.zeros(ctx.param_count(), device="cuda:0")
        ctx.init_params(9, params)
        carry = buffers.CarryBuffers(N, H, 2, "cuda:0", mirror=mirror)
        tr = buffers.TrajBuffers(T, N, H, 2, "cuda:0", mirror=mirror)
        ctx.env_reset_all(3, tr.actor_obs[T], tr.critic_obs[T], tr.aux[T])
        for it in range(2):           # two rollouts: the second starts from carried state and row T -> row 0
            if mode != "stepwise":
                ctx.rollout(params, carry.c, 3, it * T, tr.c)
            else:
                tr.actor_obs[0].copy_(tr.actor_obs[T]); tr.critic_obs[0].copy_(tr.critic_obs[T]); tr.aux[0].copy_(tr.aux[T])
                for t in range(T):
                    ctx.policy_step(params, tr.actor_obs[t], tr.critic_obs[t], carry.c, 3, it * T + t, False, tr.action[t], tr.logp[t], tr.value[t])
                    ctx.env_step(tr.action[t], tr.aux[t], tr.actor_obs[t + 1], tr.critic_obs[t + 1], tr.aux[t + 1])
                    ctx.carry_reset(carry.c, tr.aux[t].data_ptr() + 4 * L.AUX["DONE"], L.AUX["SIZE"])
                ctx.rewards(tr.aux, T, tr.reward)
        ctx.synchronize()
        got = [tr.actor_obs.clone(), tr.critic_obs.clone(), tr.aux.clone(), tr.action.clone(), tr.logp.clone(), tr.value.clone(), tr.reward.clone(),
               carry.actor_hc.clone(), carry.critic_hc.clone(), carry.lpf.clone()]
        if mirror:
            got += [carry.actor_mirror_hc.clone(), carry.critic_mirror_hc.clone(), carry.lpf_mirror.clone()]
        out.append(got)
        ctx.close()
    os.environ.pop("KBJ_ROLLOUT_PIPELINE", None)
    for a, b, d in zip(*out):
        assert torch.equal(a, d) and torch.equal(b, d)


def test_reward_components_and_actor_export(tmp_path):
    """Logging / deployment side of the task: per-term reward means add up to the reward, the exported actor is the parameter prefix."""
    import numpy as np
    import torch
    from kbot_joystick_amd.host.task import HumanoidWalkingTask, launch_config
    from kbot_joystick_amd.spec import constants
    cfg = launch_config(num_envs=64, batch_size=32, hidden_size=64, rollout_length_seconds=0.2, robot="kbot-headless", seed=4, num_passes=1,
                        log_reward_components=True)
    task = HumanoidWalkingTask(cfg)
    task.train_iteration()
    torch.cuda.synchronize()
    comps = task.reward_components()
    assert list(comps) == list(constants.REWARD_NAMES) and all(np.isfinite(v) for v in comps.values())
    scales = torch.tensor([0.2, 0.1, 0.2, 0.2, 0.2, 0.1, 0.1, 1.5, 0.1, 0.05, 0.1, 0.1], device=task.device)     # train.py:1225-1256
    assert torch.allclose((task.traj.comps * scales).sum(-1), task.traj.reward, atol=1e-5)
    path = tmp_path / "actor.npz"
    task.export_actor(str(path))
    z = np.load(path)
    n = task.ctx.actor_param_count()
    flat = np.concatenate([z[k].ravel() for k in z.files if k.startswith("actor.")])
    assert flat.size == n and np.array_equal(flat, task.params[:n].cpu().numpy())
    task.ctx.close()


def test_full_size_rollout_is_independent_of_the_batch():
    """BASELINE size property (8192 envs, launch networks): an env's trajectory does not depend on how many other envs share
    the launch — RNG streams are keyed by global env id, the env kernel is per-env, and a GEMM output row is the same fp32 MFMA
    chain whatever the tile shape. The first 256 envs of an 8192-env rollout equal a 256-env rollout bit for bit."""
    import torch
    from kbot_joystick_amd.host import binding as Bd, buffers
    from kbot_joystick_amd.spec import compiler, layout as L
    T, H = 6, 256
    m = compiler.load_model("kbot-headless")
    got = {}
    for N in (8192, 256):
        cfg = L.default_config(num_envs=N, batch_size=256, rollout_len=T, hidden_size=H)
        ctx = Bd.Context(m, cfg, 0, torch.cuda.current_stream().cuda_stream)
        params = torch.zeros(ctx.param_count(), device="cuda:0")
        ctx.init_params(21, params)
        carry = buffers.CarryBuffers(N, H, 2, "cuda:0")
        tr = buffers.TrajBuffers(T, N, H, 2, "cuda:0")
        ctx.env_reset_all(8, tr.actor_obs[T], tr.critic_obs[T], tr.aux[T])
        ctx.rollout(params, carry.c, 8, 0, tr.c)
        ctx.synchronize()
        got[N] = [t[:, :256].clone() for t in (tr.actor_obs, tr.critic_obs, tr.aux, tr.action, tr.logp, tr.value, tr.reward)]
        got[N] += [carry.actor_hc[:, :, :256].clone(), carry.critic_hc[:, :, :256].clone(), carry.lpf[:256].clone()]
        assert torch.isfinite(tr.critic_obs).all() and torch.isfinite(tr.logp).all()
        ctx.close()
    for a, b in zip(got[8192], got[256]):
        assert torch.equal(a, b)
