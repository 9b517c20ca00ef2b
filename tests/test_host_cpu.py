"""Host-side pieces that need no GPU: the xax-layout checkpoint container, the scalar logger, reward / command overrides."""
import glob
import os

import numpy as np
import pytest

from kbot_joystick_amd.spec import constants, layout as L


def test_ckpt_roundtrip_and_actor_contract(tmp_path):
    """convert.py:36-46,67-78: `load_ckpt(part="model")` gives the actor's leaves in equinox order; carry = depth*2*H + 20."""
    from kbot_joystick_amd.host import ckpt
    from kbot_joystick_amd.host.task import ModelView
    H, depth = 64, 2
    pa, pc = L.param_count(H, depth)
    rng = np.random.default_rng(0)
    p, m, v = (rng.standard_normal(pa + pc).astype(np.float32) for _ in range(3))
    path = str(tmp_path / "ckpt.bin")
    ckpt.save_ckpt(path, p, m, v, 11, H, depth, dict(num_steps=3, opt_step=11), dict(hidden_size=H, depth=depth, robot="kbot"),
                   extras=dict(es=np.arange(12, dtype=np.float32).reshape(3, 4)))
    z = ckpt.load_ckpt(path)
    assert np.array_equal(z["model"], p) and np.array_equal(z["opt_state"]["mu"], m) and np.array_equal(z["opt_state"]["nu"], v)
    assert z["opt_state"]["count"] == 11 and z["state"]["num_steps"] == 3 and z["config"]["robot"] == "kbot"
    assert np.array_equal(z["extras"]["es"], np.arange(12, dtype=np.float32).reshape(3, 4))
    assert np.array_equal(ckpt.load_ckpt(path, "model"), p)                           # hidden size read from the config member
    # the container: gzip tar with the four upstream member names first
    import tarfile
    with tarfile.open(path, "r:gz") as tar:
        assert tar.getnames()[:4] == ["model_0", "opt_state_0", "state", "config"]
        blob = tar.extractfile("model_0").read()
    assert blob[:6] == b"\x93NUMPY"                                                   # back-to-back numpy.save blobs
    mv = ModelView(z["model"], H, depth)
    assert mv.actor.input_proj.weight.shape == (H, 65) and mv.actor.output_proj.weight.shape == (40, H)
    assert mv.actor.rnns[1].weight_hh.shape == (4 * H, H) and mv.critic.input_proj.weight.shape == (H, 475)
    assert mv.carry_size == depth * 2 * H + len(constants.JOINT_NAMES)
    flat_actor = np.concatenate([a.ravel() for n, a in ckpt.split_leaves(p, H, depth) if n.startswith("actor.")])
    assert np.array_equal(flat_actor, p[:pa])                                         # the actor is the prefix of the flat vector
    with pytest.raises(FileNotFoundError):
        ckpt.load_ckpt(str(tmp_path / "missing.bin"))                                 # convert.py:33-34
    with pytest.raises(ValueError):
        ckpt.load_ckpt(path, "model", hidden_size=128)


def test_scalar_logger_writes_csv_and_tensorboard_events(tmp_path):
    from kbot_joystick_amd.host import scalars as S
    assert S.crc32c(b"123456789") == 0xE3069283                                       # CRC-32C check value
    lg = S.ScalarLogger(str(tmp_path))
    lg.log(1, {"train/loss": 1.5, "reward/linvel": 0.25})
    lg.log(2, {"train/loss": 1.25, "valid/reward_per_step": 0.5})
    lg.close()
    ev = S.read_event_file(glob.glob(str(tmp_path / "events.out.tfevents.*"))[0])     # verifies both CRCs of every record
    assert ev == [(1, {"train/loss": 1.5, "reward/linvel": 0.25}), (2, {"train/loss": 1.25, "valid/reward_per_step": 0.5})]
    rows = open(tmp_path / "scalars.csv").read().strip().splitlines()
    assert rows[0] == "step,wall_time,train/loss,reward/linvel,valid/reward_per_step" and len(rows) == 3


def test_tensorboard_event_file_decodes_with_the_protobuf_runtime(tmp_path):
    """An INDEPENDENT reader for the hand-written event file: TFRecord framing walked here, every payload parsed by google.protobuf against
    the schema TensorBoard reads (tensorflow/core/util/event.proto + framework/summary.proto field numbers: Event{wall_time = 1 double,
    step = 2 int64, file_version = 3 string, summary = 5}, Summary{value = 1 repeated}, Value{tag = 1 string, simple_value = 2 float}),
    descriptors built at run time; record CRCs checked against RFC 3720's CRC-32C test vectors' implementation below, not the writer's."""
    pb = pytest.importorskip("google.protobuf")
    import struct
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    from kbot_joystick_amd.host import scalars as S
    fd = descriptor_pb2.FileDescriptorProto(name="kbj_tb_event.proto", package="tensorboard", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto
    val = fd.message_type.add(name="Value")
    val.field.add(name="tag", number=1, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    val.field.add(name="simple_value", number=2, type=T.TYPE_FLOAT, label=T.LABEL_OPTIONAL)
    summ = fd.message_type.add(name="Summary")
    summ.field.add(name="value", number=1, type=T.TYPE_MESSAGE, label=T.LABEL_REPEATED, type_name=".tensorboard.Value")
    evd = fd.message_type.add(name="Event")
    evd.field.add(name="wall_time", number=1, type=T.TYPE_DOUBLE, label=T.LABEL_OPTIONAL)
    evd.field.add(name="step", number=2, type=T.TYPE_INT64, label=T.LABEL_OPTIONAL)
    evd.field.add(name="file_version", number=3, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    evd.field.add(name="summary", number=5, type=T.TYPE_MESSAGE, label=T.LABEL_OPTIONAL, type_name=".tensorboard.Summary")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    Event = message_factory.GetMessageClass(pool.FindMessageTypeByName("tensorboard.Event"))

    def crc32c_bitwise(data):                      # the definition (reflected polynomial 0x1EDC6F41), no table
        c = 0xFFFFFFFF
        for b in data:
            c ^= b
            for _ in range(8):
                c = (c >> 1) ^ (0x82F63B78 & -(c & 1))
        return c ^ 0xFFFFFFFF
    assert crc32c_bitwise(bytes(32)) == 0x8A9136AA and crc32c_bitwise(bytes([0xFF] * 32)) == 0x62A8AB43      # RFC 3720 B.4
    mask = lambda c: (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF
    lg = S.ScalarLogger(str(tmp_path))
    lg.log(7, {"train/loss": 0.75, "reward/feet_airtime": -0.125})
    lg.log(1 << 40, {"perf/env_steps_per_s": 2.25e6})
    lg.close()
    raw = open(glob.glob(str(tmp_path / "events.out.tfevents.*"))[0], "rb").read()
    off, events = 0, []
    while off < len(raw):
        (n,) = struct.unpack_from("<Q", raw, off)
        assert struct.unpack_from("<I", raw, off + 8)[0] == mask(crc32c_bitwise(raw[off:off + 8]))
        payload = raw[off + 12:off + 12 + n]
        assert struct.unpack_from("<I", raw, off + 12 + n)[0] == mask(crc32c_bitwise(payload))
        e = Event()
        e.ParseFromString(payload)
        assert e.SerializeToString() == payload          # nothing in the payload the schema does not know, canonical field order
        events.append(e)
        off += 16 + n
    assert off == len(raw) and events[0].file_version == "brain.Event:2" and events[0].wall_time > 1e9
    got = [(e.step, {v.tag: v.simple_value for v in e.summary.value}) for e in events[1:]]
    assert got == [(7, {"train/loss": 0.75, "reward/feet_airtime": -0.125}), (1 << 40, {"perf/env_steps_per_s": 2.25e6})]
    assert all(e.wall_time > 1e9 for e in events[1:])


def test_reward_and_command_overrides_reach_kbj_config():
    from kbot_joystick_amd.host.task import launch_config
    c = launch_config(reward_scales={"torque": 0.0, "feet_airtime": 2.0}, reward_params={"base_height": {"standard_height": 0.85}},
                      command_ranges={"vx_range": (-1.0, 2.0)})
    k = c.to_kbj(4096)
    assert k.reward_scale[constants.REWARD_NAMES.index("torque")] == 0.0 and k.reward_scale[constants.REWARD_NAMES.index("feet_airtime")] == 2.0
    assert abs(k.rew_standard_height - 0.85) < 1e-6 and (k.vx_lo, k.vx_hi) == (-1.0, 2.0)
    with pytest.raises(KeyError):
        launch_config(reward_scales={"nope": 1.0}).to_kbj(4096)
    with pytest.raises(KeyError):
        launch_config(reward_params={"torque": {"standard_height": 1.0}}).to_kbj(4096)
    with pytest.raises(ValueError):
        launch_config(allreduce="sometimes").to_kbj(4096)
    # a25: the samplers on jax.random's key handling = kbj_config.command_mode 2; the library's host-side check accepts 0..2 only
    assert launch_config().to_kbj(4096).command_mode == 0 and launch_config(jax_random_keys=True).to_kbj(4096).command_mode == 2
    with pytest.raises(ValueError):
        launch_config(jax_random_keys=True, fixed_command=(0.5, 0.0, 0.0)).to_kbj(4096)


def test_oracle_rewards_follow_the_config(model):
    """The oracle reads the reward table from kbj_config: doubling a scale changes the total by exactly that term."""
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    N, T = 8, 5
    aux = rng.normal(size=(T, N, L.AUX["SIZE"])).astype(np.float32) * 0.1
    aux[:, :, L.AUX["BQUAT"]] = 1.0
    aux[:, :, L.AUX["DONE"]:] = 0
    base = L.default_config(num_envs=N, batch_size=N)
    r0, c0 = O.Oracle(model, base, 0).rewards(aux)
    i = constants.REWARD_NAMES.index("angvel")
    mod = L.default_config(num_envs=N, batch_size=N)
    mod.reward_scale[i] = 2 * base.reward_scale[i]
    r1, c1 = O.Oracle(model, mod, 0).rewards(aux)
    assert np.allclose(c0, c1) and np.allclose(r1 - r0, base.reward_scale[i] * c0[:, :, i], atol=1e-6)
    mod2 = L.default_config(num_envs=N, batch_size=N, rew_angvel_err=0.4)
    _, c2 = O.Oracle(model, mod2, 0).rewards(aux)
    assert np.allclose(c2[:, :, i], np.sqrt(c0[:, :, i]), atol=1e-6)                 # exp(-e/0.4) = sqrt(exp(-e/0.2))


def test_upstream_style_checkpoint_without_kbj_members(tmp_path):
    """A checkpoint as the reference's own run would leave it: no kbj_* members, no `opt_step` key in `state`, and an optimizer with a
    schedule (train.py:1067-1077: optax adds a second integer `count` leaf). Model and optimizer load; the counters fall back to the
    optax count; an optimizer member this build cannot map onto (mu, nu) degrades to model-only instead of failing."""
    import io
    import json
    import tarfile
    from kbot_joystick_amd.host import ckpt
    H, depth = 64, 2
    pa, pc = L.param_count(H, depth)
    rng = np.random.default_rng(1)
    p, m, v = (rng.standard_normal(pa + pc).astype(np.float32) for _ in range(3))
    leaves = lambda flat: [a for _, a in ckpt.split_leaves(flat, H, depth)]

    def write(path, opt_blobs):
        with tarfile.open(path, "w:gz") as tar:
            for name, data in (("model_0", ckpt._npy_blobs(leaves(p))), ("opt_state_0", opt_blobs), ("state", json.dumps(dict(num_steps=7, num_samples=123)).encode()),
                               ("config", ckpt._yaml(dict(hidden_size=H, depth=depth)).encode())):
                if data is None:
                    continue
                info = tarfile.TarInfo(name); info.size = len(data)
                tar.addfile(info, io.BytesIO(data))

    path = str(tmp_path / "upstream.bin")
    write(path, ckpt._npy_blobs([np.asarray(41, np.int32)] + leaves(m) + leaves(v) + [np.asarray(41, np.int32)]))   # adamw under a schedule
    z = ckpt.load_ckpt(path)
    assert np.array_equal(z["model"], p) and z["extras"] == {} and "opt_step" not in z["state"]
    assert z["opt_state"]["count"] == 41 and z["opt_state"]["counts"] == [41, 41]
    assert np.array_equal(z["opt_state"]["mu"], m) and np.array_equal(z["opt_state"]["nu"], v)
    write(path, ckpt._npy_blobs([np.asarray(3, np.int32)] + leaves(m)))          # a state this build cannot map: model-only
    z = ckpt.load_ckpt(path)
    assert z["opt_state"] is None and np.array_equal(z["model"], p)
    write(path, None)                                                             # no optimizer member at all
    assert ckpt.load_ckpt(path)["opt_state"] is None and np.array_equal(ckpt.load_ckpt(path, "model"), p)


def test_schedule_count_leaf_and_episode_length_rounding(tmp_path):
    """With use_lr_decay optax's state tree is (count, mu.., nu.., count) in either branch of train.py:1067-1077: the writer emits the
    trailing ScaleByScheduleState leaf so that the reference can restore its optimizer from the file, and the reader maps it back.
    `max_length_sec` rounds like T and substeps do (2.3 s / 0.02 s is 115 control steps, not 114)."""
    import tarfile
    from kbot_joystick_amd.host import ckpt
    from kbot_joystick_amd.host.task import launch_config
    H, depth = 64, 2
    pa, pc = L.param_count(H, depth)
    nleaf = len(L.param_leaves(H, depth))
    p = np.arange(pa + pc, dtype=np.float32)
    path = str(tmp_path / "ckpt.bin")
    for sched, want in ((None, 1 + 2 * nleaf), (17, 2 + 2 * nleaf)):
        ckpt.save_ckpt(path, p, p, p, 17, H, depth, dict(num_steps=1, opt_step=17), dict(hidden_size=H, depth=depth), schedule_count=sched)
        with tarfile.open(path, "r:gz") as tar:
            blobs = ckpt._read_blobs(tar.extractfile("opt_state_0").read())
        assert len(blobs) == want
        if sched is not None:
            assert blobs[-1].dtype == np.int32 and int(blobs[-1].reshape(-1)[0]) == 17
        z = ckpt.load_ckpt(path, "opt_state")
        assert z["count"] == 17 and z["counts"] == ([17] if sched is None else [17, 17]) and np.array_equal(z["mu"], p)
    assert ckpt.has_member(path, "opt_state_0") and not ckpt.has_member(path, "kbj_es")
    k = launch_config(termination_params={"episode_length": {"max_length_sec": 2.3}}).to_kbj(4096)
    assert k.max_episode_steps == 115


def test_lr_decay_without_weight_decay_needs_an_explicit_opt_in():
    """train.py:1074-1075 as written (scale_by_adam chained with scale_by_schedule, no sign flip) is gradient ASCENT: flipping one
    documented flag must not silently produce a diverging run."""
    from kbot_joystick_amd.host.task import launch_config
    with pytest.raises(ValueError, match="reproduce_reference_lr_sign"):
        launch_config(use_lr_decay=True, adam_weight_decay=0.0).to_kbj(4096)
    launch_config(use_lr_decay=True, adam_weight_decay=0.0, reproduce_reference_lr_sign=True).to_kbj(4096)
    launch_config(use_lr_decay=True).to_kbj(4096)                         # the adamw branch (train.py:1076-1077) descends


def test_checkpoint_write_is_atomic(tmp_path, monkeypatch):
    """save_ckpt writes ckpt.bin.tmp, fsyncs and renames: a crash in the middle of a save leaves the previous checkpoint intact."""
    from kbot_joystick_amd.host import ckpt
    H, depth = 64, 1
    pa, pc = L.param_count(H, depth)
    p = np.arange(pa + pc, dtype=np.float32)
    path = str(tmp_path / "ckpt.bin")
    ckpt.save_ckpt(path, p, p, p, 1, H, depth, dict(num_steps=1, opt_step=1), dict(hidden_size=H, depth=depth))
    assert os.listdir(tmp_path) == ["ckpt.bin"]
    calls = []
    real = ckpt._add

    def failing(tar, name, data):
        calls.append(name)
        if name == "state":
            raise OSError("disk full")
        real(tar, name, data)
    monkeypatch.setattr(ckpt, "_add", failing)
    with pytest.raises(OSError):
        ckpt.save_ckpt(path, 2 * p, p, p, 2, H, depth, dict(num_steps=2, opt_step=2), dict(hidden_size=H, depth=depth))
    monkeypatch.setattr(ckpt, "_add", real)
    assert np.array_equal(ckpt.load_ckpt(path, "model"), p) and ckpt.load_ckpt(path, "state")["num_steps"] == 1


def test_reward_error_scales_are_validated():
    from kbot_joystick_amd.host.task import launch_config
    for bad in (0.0, -0.1, float("nan"), float("inf")):
        with pytest.raises(ValueError, match="error_scale"):
            launch_config(reward_params={"linvel": {"error_scale": bad}}).to_kbj(4096)
    with pytest.raises(ValueError, match="finite"):
        launch_config(reward_params={"base_height": {"standard_height": float("nan")}}).to_kbj(4096)
    launch_config(reward_params={"roll_pitch": {"error_scale_zero_cmd": 0.02}, "feet_airtime": {"touchdown_penalty": 0.0}}).to_kbj(4096)


def test_view_kinematics_match_the_oracle_and_the_player_page_is_self_contained(tmp_path):
    """host/view.py (run_mode=view, reference README.md:66-70): body positions / orientations from the model blob's tree against the oracle's
    forward pass at random poses, the foot capsules under the envs' randomised geometry, and the written page (data embedded, no external
    reference)."""
    import json, re
    from kbot_joystick_amd.spec import compiler
    from kbot_joystick_amd.host import view as V
    from oracle import oracle as O
    m = compiler.load_model("kbot-headless")
    cfg = L.default_config(num_envs=3)
    ep = O.default_params(m, cfg)
    rng = np.random.default_rng(3)
    nb, nq = int(m.nbody), int(m.nq)
    qs = []
    for _ in range(5):
        q = np.array(m.qpos0[:nq], np.float64)
        q[7:] += rng.normal(0, 0.5, nq - 7); qq = rng.normal(0, 1, 4); q[3:7] = qq / np.linalg.norm(qq); q[:3] += rng.normal(0, 0.3, 3)
        r = O.forward(m, cfg, ep, q, np.zeros(26))
        xp, xq = V.forward_kinematics(m, q)
        sgn = np.sign((xq * r["xquat"][:nb]).sum(-1, keepdims=True))
        assert np.abs(xp - r["xpos"][:nb]).max() < 1e-12 and np.abs(xq - sgn * r["xquat"][:nb])[1:].max() < 1e-12
        qs.append(q)
    # batched call == per-pose calls; capsule end points sit half a length either side of the centre, along the body's rotated axis
    Q = np.stack(qs)[:, None, :].repeat(3, axis=1)                       # [F = 5][K = 3][nq]
    xp, xq = V.forward_kinematics(m, Q)
    assert np.array_equal(xp[2, 1], V.forward_kinematics(m, qs[2])[0])
    eps = np.tile(ep, (3, 1)); eps[1, L.EP["CAP_HALF"]:L.EP["CAP_HALF"] + 4] *= 1.1
    seg, rad = V.capsule_segments(m, xp, xq, eps[None])
    ln = np.linalg.norm(seg[..., 1, :] - seg[..., 0, :], axis=-1)
    assert np.allclose(ln[:, 0], 2 * eps[0, L.EP["CAP_HALF"]:L.EP["CAP_HALF"] + 4]) and np.allclose(ln[:, 1], 2.2 * eps[0, L.EP["CAP_HALF"]:L.EP["CAP_HALF"] + 4], rtol=1e-6)
    rec = V.Recording(m, Q, eps, np.zeros((5, 3, 16), np.float32), np.zeros((4, 3), np.float32), np.zeros((4, 3), np.float32), 0.02, 0)
    stem = str(tmp_path / "roll")
    rec.save_npz(stem + ".npz"); rec.save_html(stem + ".html")
    z = np.load(stem + ".npz")
    assert z["qpos"].shape == (5, 3, nq) and z["xpos"].shape == (5, 3, nb, 3) and z["caps"].shape == (5, 3, 4, 2, 3)
    page = open(stem + ".html").read()
    assert "http://" not in page and "https://" not in page and "src=" not in page            # plays offline, nothing fetched
    data = json.loads(re.search(r"const R = (\{.*?\});\n", page, re.S).group(1))
    assert data["F"] == 5 and data["K"] == 3 and data["nbody"] == nb and data["track"] == 1 and len(data["xpos"][0][0]) == nb
    assert np.allclose(np.array(data["xpos"]), xp, atol=1e-4)
    # the page's script runs (node with a stub canvas, when the image has node): 120 animation frames without an exception
    import shutil, subprocess
    if shutil.which("node"):
        js = re.search(r"<script>\n(.*)</script>", page, re.S).group(1)
        stub = """
const calls = {n: 0};
const ctx = new Proxy({}, {get: (t, k) => (k in t) ? t[k] : (...a) => { calls.n++; }, set: (t, k, v) => { t[k] = v; return true; }});
function el(id) { return {id, getContext: () => ctx, width: 1200, height: 520, add() {}, selectedIndex: 1, value: "1", textContent: "", onclick: null, oninput: null}; }
const els = {};
globalThis.document = {getElementById: (id) => els[id] || (els[id] = el(id)), createElement: () => ({})};
let frames = 0;
globalThis.requestAnimationFrame = (f) => { if (frames++ < 120) setImmediate(() => f(frames * 16.7)); else console.log("OK " + calls.n + " " + els.hud.textContent); };
"""
        (tmp_path / "check.js").write_text(stub + js)
        out = subprocess.run(["node", str(tmp_path / "check.js")], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0 and out.stdout.startswith("OK ") and "command vx" in out.stdout, out.stderr[-400:]


def test_reference_reward_classes_on_a_ksim_shaped_trajectory_match_the_oracle(model_full):
    """host/trajectory.py on the CPU: (a) the torch forward kinematics equal the numpy ones the view recorder uses (themselves pinned to the
    oracle's xpos / xquat above); (b) the reference's twelve reward classes, restated in torch against `Trajectory` with the reference's
    attribute names (train.py:138-506), reproduce the ORACLE's reward scan term by term on a synthetic two-rollout record - random poses,
    commands incl. zero commands and turning, contacts, terminations - whose aux columns are what the env kernel would have written for
    those states. (The GPU twin, against rewards_kernel on real rollouts: tests/test_gpu_host.py.)"""
    import types
    import torch
    from kbot_joystick_amd.host import trajectory as TJ, view as V
    from examples import reference_rewards as RR          # the reference's reward classes: example material, not product code
    from kbot_joystick_amd.spec import layout
    from oracle import oracle as O
    m = model_full
    rng = np.random.default_rng(3)
    T, N = 40, 24
    A, Q = L.AUX, L.QSTATE
    cfg = layout.default_config(num_envs=N)
    o = O.Oracle(m, cfg, precision="f64")
    terms = RR.reference_rewards(m, ctrl_dt=cfg.ctrl_dt)
    assert list(terms) == list(constants.REWARD_NAMES)
    assert [t.scale for t in terms.values()] == pytest.approx(list(cfg.reward_scale))
    carries = {}
    for rollout in range(2):
        q = np.tile(np.array(m.qpos0[:L.NQ], np.float64), (T, N, 1))
        q[..., 7:] += rng.normal(0, 0.3, (T, N, L.NQ - 7))
        qq = np.array([1.0, 0, 0, 0]) + rng.normal(0, 0.25, (T, N, 4)); q[..., 3:7] = qq / np.linalg.norm(qq, axis=-1, keepdims=True)
        q[..., :3] += rng.normal(0, 0.1, (T, N, 3))
        xp, xq = V.forward_kinematics(m, q)
        xpt, xqt = TJ.forward_kinematics(m, torch.from_numpy(q))
        assert np.abs(xpt.numpy() - xp).max() < 1e-12 and np.abs(xqt.numpy() - xq).max() < 1e-12          # (a)
        qs = np.zeros((T, N, Q["SIZE"]), np.float32)
        qs[..., Q["QPOS_KIN"]:Q["QPOS_KIN"] + L.NQ] = q
        qpos_after = q + rng.normal(0, 0.01, q.shape)
        qs[..., Q["QPOS"]:Q["QPOS"] + L.NQ] = qpos_after
        qvel = rng.normal(0, 0.5, (T, N, L.NV))
        qs[..., Q["QVEL"]:Q["QVEL"] + L.NV] = qvel
        aux = np.zeros((T + 1, N, A["SIZE"]), np.float32)
        a = aux[:T]
        a[..., A["QVEL"]:A["QVEL"] + 6] = qs[..., Q["QVEL"]:Q["QVEL"] + 6]
        a[..., A["BQUAT"]:A["BQUAT"] + 4] = xq[..., int(m.base_body), :]
        a[..., A["LFQUAT"]:A["LFQUAT"] + 4] = xq[..., int(m.lfoot_body), :]
        a[..., A["RFQUAT"]:A["RFQUAT"] + 4] = xq[..., int(m.rfoot_body), :]
        a[..., A["BASEZ"]], a[..., A["LFZ"]], a[..., A["RFZ"]] = xp[..., int(m.base_body), 2], xp[..., int(m.lfoot_body), 2], xp[..., int(m.rfoot_body), 2]
        a[..., A["ARMQ"]:A["ARMQ"] + 10] = qs[..., Q["QPOS"] + 17:Q["QPOS"] + 27]
        a[..., A["CTRL"]:A["CTRL"] + L.NU] = rng.normal(0, 8, (T, N, L.NU))
        a[..., A["TOUCH"]:A["TOUCH"] + 2] = rng.uniform(0, 300, (T, N, 2)) * (rng.uniform(size=(T, N, 2)) < 0.6)
        a[..., A["COMDIST"]] = np.where(rng.uniform(size=(T, N)) < 0.2, -1.0, rng.uniform(0, 0.2, (T, N)))
        cmd = rng.uniform(-0.5, 0.5, (N, L.NCMD)); cmd[: N // 3, :3] = 0.0; cmd[N // 3: N // 2, 2] = 0.0        # standing envs, straight walkers, turners
        a[..., A["CMD"]:A["CMD"] + L.NCMD] = cmd[None]
        a[T // 2:, N - 4:, A["CMD"]:A["CMD"] + 3] = 0.0                                                            # a switch to the zero command mid-rollout
        a[..., A["DONE"]] = np.where(rng.uniform(size=(T, N)) < 0.05, np.where(rng.uniform(size=(T, N)) < 0.5, -1.0, 1.0), 0.0)
        rew, comps = o.rewards(aux[:T])
        fake = types.SimpleNamespace(aux=torch.from_numpy(aux), qstate=torch.from_numpy(qs), action=torch.zeros(T, N, L.NU), reward=torch.zeros(T, N),
                                     actor_obs=torch.zeros(T + 1, N, L.LD_ACTOR), critic_obs=torch.zeros(T + 1, N, L.LD_CRITIC))
        tr = TJ.Trajectory(fake, T, m)
        assert tr.qpos.shape == (T, N, 27) and tr.qvel.shape == (T, N, 26) and tr.xpos.shape == (T, N, 24, 3) and tr.xquat.shape == (T, N, 24, 4)
        assert tr.done.dtype == torch.bool and tr.obs["left_foot_touch"].shape == (T, N, 1) and tr.command["unified_command"].shape == (T, N, 16)
        total = torch.zeros(T, N)
        for k, (name, term) in enumerate(terms.items()):
            if hasattr(term, "get_reward_stateful"):
                if name not in carries:
                    carries[name] = term.initial_carry(N, "cpu")
                r, carries[name] = term.get_reward_stateful(tr, carries[name])
            else:
                r = term.get_reward(tr)
            assert r.shape == (T, N), name
            err = float((r.double() - torch.from_numpy(comps[..., k]).double()).abs().max())
            assert err < 2e-4, (rollout, name, err)
            total += term.scale * r
        assert float((total - torch.from_numpy(rew)).abs().max()) < 2e-4
    # per_env: reward bodies written time-first, exactly as the reference's (train.py:316-334, 487-494: `trajectory.xquat[:, 1, :]`, `jnp.pad(..., ((1, 0), (0, 0)),
    # mode="edge")`), evaluated per env under torch.vmap as ksim evaluates them under jax.vmap - bit-equal to the batched classes
    def xy_orientation(traj):
        e = TJ.quat_to_euler(traj.xquat[:, 1, :])
        base_xy_quat = TJ.euler_to_quat(torch.cat([e[:, :2], torch.zeros_like(e[:, 2:])], dim=-1))
        cmd = traj.command["unified_command"]
        cmd_quat = TJ.euler_to_quat(torch.stack([cmd[:, 4], cmd[:, 5], torch.zeros_like(cmd[:, 5])], dim=-1))
        quat_error = 1 - (cmd_quat * base_xy_quat).sum(dim=-1) ** 2
        is_zero_cmd = torch.linalg.norm(cmd[:, :3], dim=-1) < 1e-3
        return torch.exp(-quat_error / torch.where(is_zero_cmd, 0.01, 0.03))

    def base_acceleration(traj):
        base_vel = traj.qvel[:, :6]
        padded = torch.cat([base_vel[:1], base_vel], dim=0)
        done_padded = torch.cat([traj.done[:1], traj.done], dim=0)
        acc = torch.where(done_padded[:-1, None], torch.zeros_like(base_vel), padded[1:] - padded[:-1])
        return torch.exp(-acc.abs().sum(dim=-1) / 5.0)
    assert torch.equal(TJ.per_env(xy_orientation)(tr), terms["roll_pitch"].get_reward(tr))
    assert torch.equal(TJ.per_env(base_acceleration)(tr), terms["base_accel"].get_reward(tr))
    with pytest.raises(ValueError, match="record_state"):
        TJ.Trajectory(types.SimpleNamespace(aux=torch.zeros(2, 1, A["SIZE"]), qstate=None, action=torch.zeros(1, 1, 20), actor_obs=torch.zeros(2, 1, L.LD_ACTOR),
                                            critic_obs=torch.zeros(2, 1, L.LD_CRITIC)), 1, m)
