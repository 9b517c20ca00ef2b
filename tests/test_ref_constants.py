"""Pins this build's constant tables to the values the reference's own files hold.

tests/golden/ref_constants.json is extracted from /root/reference/train.py + convert.py by tests/golden/make_ref_constants.py
(AST, nothing executed; values only). These tests compare spec/constants.py, layout.default_config (= kbj_config, which the HIP
kernels and the oracle both read - the reward table included), the config dataclass defaults and launch_config against it.
What this does NOT pin: the semantics of the un-vendored ksim / mujoco-mjx code.
"""
import json
import math
import os

import pytest

from kbot_joystick_amd.spec import constants, layout as L

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ref():
    with open(os.path.join(HERE, "golden", "ref_constants.json")) as f:
        return json.load(f)


def approx(a, b, rel=1e-6):
    return abs(a - b) <= rel * max(1.0, abs(a), abs(b))


def test_fixture_is_current_with_the_reference(ref):
    """In the build container (reference tree present) the committed JSON must equal a fresh extraction."""
    if not os.path.exists("/root/reference/train.py"):
        pytest.skip("reference tree not present (GPU box)")
    import subprocess, sys, tempfile, shutil
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(os.path.join(HERE, "golden", "make_ref_constants.py"), tmp)
        subprocess.check_call([sys.executable, os.path.join(tmp, "make_ref_constants.py"), "/root/reference"], stdout=subprocess.DEVNULL)
        with open(os.path.join(tmp, "ref_constants.json")) as f:
            assert json.load(f) == ref
    finally:
        shutil.rmtree(tmp)


def test_joint_tables(ref):
    jb, jl = ref["tables"]["JOINT_BIASES"], ref["tables"]["JOINT_LIMITS"]          # train.py:22-70
    assert tuple(jb["names"]) == constants.JOINT_NAMES == tuple(jl["names"])
    assert all(approx(a, b, 1e-12) for a, b in zip(jb["values"], constants.JOINT_BIASES))
    assert all(approx(a[0], b[0], 1e-12) and approx(a[1], b[1], 1e-12) for a, b in zip(jl["values"], constants.JOINT_LIMITS))
    assert tuple(ref["convert"]["command_names"]) == constants.COMMAND_NAMES         # convert.py:48-65


def test_model_blob_carries_the_joint_tables(ref, model, model_full):
    for m in (model, model_full):
        assert all(approx(a, b) for a, b in zip(ref["tables"]["JOINT_BIASES"]["values"], m.joint_bias))
        for (lo, hi), mlo, mhi in zip(ref["tables"]["JOINT_LIMITS"]["values"], m.joint_lo, m.joint_hi):
            assert approx(lo, mlo) and approx(hi, mhi)


def test_config_dataclass_defaults(ref):
    from kbot_joystick_amd.host.task import HumanoidWalkingTaskConfig
    c = HumanoidWalkingTaskConfig()
    for name, rec in ref["config_defaults"].items():                                  # train.py:73-122
        assert getattr(c, name) == rec["value"], name


def test_launch_block(ref):
    from kbot_joystick_amd.host.task import launch_config
    c = launch_config()
    for name, v in ref["launch"]["kwargs"].items():                                   # train.py:1761-1791
        got = getattr(c, name)
        assert (list(got) if isinstance(got, tuple) else got) == v, name
    k = c.to_kbj(c.num_envs)
    assert (k.num_envs, k.batch_size, k.num_passes, k.hidden_size) == (4096, 512, 3, 256)
    assert k.rollout_len == round(ref["launch"]["kwargs"]["rollout_length_seconds"] / ref["launch"]["kwargs"]["ctrl_dt"])
    assert k.substeps == round(ref["launch"]["kwargs"]["ctrl_dt"] / ref["launch"]["kwargs"]["dt"])
    assert (k.solver_iterations, k.ls_iterations) == (ref["launch"]["kwargs"]["iterations"], ref["launch"]["kwargs"]["ls_iterations"])
    assert approx(k.latency_lo, 0.003) and approx(k.latency_hi, 0.01) and approx(k.drop_action_prob, 0.05)
    assert approx(k.gamma, 0.94) and approx(k.lam, 0.94) and approx(k.entropy_coef, 0.004) and approx(k.learning_rate, 5e-4)
    assert k.actor_mirror_loss_scale == 0.0 and k.critic_mirror_loss_scale == 0.0


def test_reward_table(ref):
    """kbj_config.reward_scale / rew_* (read by rewards_kernel and by the oracle) == get_rewards() (train.py:1224-1256)."""
    from kbot_joystick_amd.host import wiring
    w = ref["wiring"]["get_rewards"]
    assert tuple(w["order"]) == constants.REWARD_NAMES
    k = L.default_config()
    views = wiring.rewards(k)
    for i, name in enumerate(constants.REWARD_NAMES):
        kw = dict(w["entries"][name]["kwargs"])
        assert approx(k.reward_scale[i], kw.pop("scale")), name
        assert approx(constants.REWARD_SCALES[i], k.reward_scale[i])
        for pname, v in kw.items():
            if isinstance(v, str):
                continue                      # body names
            assert approx(views[name].params[pname], v), (name, pname)
        for pname, expr in w["entries"][name]["symbolic"].items():
            if pname == "ctrl_dt":
                assert expr == "self.config.ctrl_dt" and approx(views[name].params["ctrl_dt"], k.ctrl_dt)


def test_command_ranges(ref):
    kw = ref["wiring"]["get_commands"]["entries"]["unified_command"]                  # train.py:1211-1221
    k = L.default_config()
    for short, key in (("vx", "vx_range"), ("vy", "vy_range"), ("wz", "wz_range"), ("bh", "bh_range"), ("rx", "rx_range"), ("ry", "ry_range")):
        assert approx(getattr(k, short + "_lo"), kw["kwargs"][key][0]) and approx(getattr(k, short + "_hi"), kw["kwargs"][key][1])
    assert kw["symbolic"]["switch_prob"] == "self.config.ctrl_dt / 5" and approx(k.switch_prob, k.ctrl_dt / 5)


def test_actuator_randomizer_event_reset_termination_constants(ref):
    k = L.default_config()
    a = ref["wiring"]["get_actuators"]["calls"][0]["kwargs"]                          # train.py:1097-1105
    for f in ("kp_scale", "kd_scale", "torque_limit_scale_low", "action_bias_scale", "torque_bias_scale"):
        assert approx(getattr(k, f), a[f]), f
    r = ref["wiring"]["get_physics_randomizers"]["entries"]                           # train.py:1107-1132
    assert approx(k.floor_friction_lo, r["floor_friction"]["kwargs"]["scale_lower"]) and approx(k.floor_friction_hi, r["floor_friction"]["kwargs"]["scale_upper"])
    assert approx(k.com_jitter, r["all_body_COM"]["kwargs"]["scale"]) and approx(k.inertia_scale, r["all_body_inertia"]["kwargs"]["scale"])
    cb = r["collision_body"]["kwargs"]
    assert approx(k.cap_radius_scale, cb["radius_scale"]) and approx(k.cap_length_scale, cb["length_scale"])
    assert [round(x, 6) for x in k.cap_jitter] == [cb["position_jitter_x"], cb["position_jitter_y"], cb["position_jitter_z"]]
    assert tuple(cb["geom_names"]) == constants.COLLISION_CAPSULES
    e = ref["wiring"]["get_events"]["entries"]["force_push"]["kwargs"]                # train.py:1134-1144
    assert approx(k.push_max_force, e["max_force"]) and approx(k.push_max_torque, e["max_torque"])
    assert [round(x, 6) for x in (k.push_dur_lo, k.push_dur_hi, k.push_int_lo, k.push_int_hi)] == e["duration_range"] + e["interval_range"]
    assert e["body_name"] == constants.BASE_BODY
    rs = {c["call"]: c["kwargs"] for c in ref["wiring"]["get_resets"]["calls"]}       # train.py:1146-1153
    assert approx(k.reset_joint_pos_scale, rs["ksim.RandomJointPositionReset.create"]["scale"])
    assert approx(k.reset_joint_vel_scale, rs["ksim.RandomJointVelocityReset"]["scale"])
    assert approx(k.reset_base_vel_xy_scale, rs["ksim.RandomBaseVelocityXYReset"]["scale"])
    assert approx(k.reset_xy_range, rs["PlaneXYPositionReset"]["x_range"]) and approx(k.reset_xy_range, rs["PlaneXYPositionReset"]["y_range"])
    t = ref["wiring"]["get_terminations"]["entries"]                                  # train.py:1258-1269
    assert approx(k.unhealthy_z, t["bad_z"]["kwargs"]["unhealthy_z"]) and approx(k.max_tilt_rad, t["not_upright"]["kwargs"]["max_radians"])
    assert approx(k.max_episode_steps * k.ctrl_dt, t["episode_length"]["kwargs"]["max_length_sec"])
    assert t["bad_z"]["kwargs"]["foot_left_body_name"] == constants.FOOT_LEFT_BODY and t["bad_z"]["kwargs"]["foot_right_body_name"] == constants.FOOT_RIGHT_BODY


def test_observation_noise_and_model_constants(ref):
    k = L.default_config()
    o = ref["wiring"]["get_observations"]                                             # train.py:1156-1204
    assert approx(k.jpos_bias_range, o["entries"]["biased_joint_position"]["kwargs"]["bias_range"])
    pg = o["entries"]["imu_projected_gravity"]["kwargs"]
    assert approx(k.pg_bias, pg["bias"]) and approx(k.pg_lag_lo, pg["min_lag"]) and approx(k.pg_lag_hi, pg["max_lag"])
    noises = [(c["call"], list(c["kwargs"].values())[0]) for c in o["calls"]]         # in source order: jpos, jvel (uniform), gyro, projected gravity (gaussian)
    assert [n for n, _ in noises] == ["ksim.AdditiveUniformNoise", "ksim.AdditiveUniformNoise", "ksim.AdditiveGaussianNoise", "ksim.AdditiveGaussianNoise"]
    for got, (_, want) in zip((k.jpos_noise, k.jvel_noise, k.gyro_noise_std, k.pg_noise_std), noises):
        assert approx(got, want)
    assert len(o["order"]) == 21
    mk = ref["wiring"]["get_model"]["calls"][0]["kwargs"]                             # train.py:1320-1321
    assert approx(k.min_std, mk["min_std"]) and approx(k.max_std, mk["max_std"])
    d = ref["config_defaults"]
    assert approx(k.var_scale, d["var_scale"]["value"]) and approx(k.weight_decay, d["adam_weight_decay"]["value"])
    assert approx(k.lpf_alpha, k.ctrl_dt / (k.ctrl_dt + 1 / (2 * math.pi * d["cutoff_frequency"]["value"])))
    assert ref["convert"]["carry_size_expr"] == "(depth * 2 * hidden_size + len(joint_names),)"


def test_wiring_views_mirror_the_reference_methods(ref, model_full):
    """Every get_* the reference overrides has a view, with the reference's entry names in the reference's order."""
    from kbot_joystick_amd.host import wiring
    k = L.default_config()
    assert list(wiring.physics_randomizers(k)) == ref["wiring"]["get_physics_randomizers"]["order"]
    assert list(wiring.events(k)) == ref["wiring"]["get_events"]["order"]
    assert list(wiring.observations(k)) == ref["wiring"]["get_observations"]["order"]
    assert list(wiring.commands(model_full, k)) == ref["wiring"]["get_commands"]["order"]
    assert list(wiring.rewards(k)) == ref["wiring"]["get_rewards"]["order"]
    assert list(wiring.terminations(k)) == ref["wiring"]["get_terminations"]["order"]
    assert [r.name for r in wiring.resets(k)] == [c["call"].split(".")[-2] if c["call"].endswith(".create") else c["call"].split(".")[-1]
                                                   for c in ref["wiring"]["get_resets"]["calls"]]
    cur = ref["wiring"]["get_curriculum"]["calls"][0]["kwargs"]
    c = wiring.CurriculumSpec()
    assert (c.step_size, c.step_every_n_epochs, c.min_level) == (cur["step_size"], cur["step_every_n_epochs"], cur["min_level"])
    from kbot_joystick_amd.host.task import HumanoidWalkingTask
    for name in list(ref["wiring"]) + ["get_ppo_variables", "get_initial_model_carry", "sample_action", "run_actor", "run_critic", "launch", "load_task", "load_ckpt"]:
        assert callable(getattr(HumanoidWalkingTask, name)), name


# ---- structure pins (observation packing order, mirror tables, convert.py's step function): derived from the reference's source text ----
# reference observation key -> (name in layout.OBS / kbj_model.h KBJ_OBS_*, element offset inside that piece, width of the raw observation)
NB1 = L.NBODY - 1
PIECES = {"joint_position": ("JPOS", 0, L.NU), "noisy_biased_joint_position": ("JPOS", 0, L.NU), "joint_velocity": ("JVEL", 0, L.NU),
          "noisy_joint_velocity": ("JVEL", 0, L.NU), "projected_gravity": ("PG", 0, 3), "noisy_imu_projected_gravity": ("PG", 0, 3),
          "imu_gyro": ("GYRO", 0, 3), "noisy_imu_gyro": ("GYRO", 0, 3), "zero_cmd": ("ZEROCMD", 0, 1), "unified_command": ("CMD", 0, L.NCMD),
          "left_foot_touch": ("TOUCH", 0, 1), "right_foot_touch": ("TOUCH", 1, 1), "feet_position": ("FEETPOS", 0, 6), "base_position": ("BASEPOS", 0, 3),
          "base_orientation": ("BASEQUAT", 0, 4), "center_of_mass_inertia": ("CINERT", 0, 10 * NB1), "center_of_mass_velocity": ("CVEL", 0, 6 * NB1),
          "base_linear_velocity": ("LINVEL", 0, 3), "base_angular_velocity": ("ANGVEL", 0, 3), "actuator_force": ("ACTFRC", 0, L.NU), "base_height": ("HEIGHT", 0, 1)}


def packed_width(entry):
    return 5 if entry["wrapper"] == "encode_projected_gravity" else PIECES[entry["key"]][2]     # roll, pitch, unit vector (train.py:1338-1349)


def test_observation_packing_order_matches_the_reference(ref):
    """train.py:1351-1433: the order, wrappers and divisors of the concatenated actor / critic vectors give this build's offsets."""
    pk = ref["packing"]
    assert pk["encode_projected_gravity_order"] == ["roll", "pitch", "projected_gravity_unit"]
    assert pk["normalize_joint_vel_divisor"] == L.OBS_JVEL_DIV and pk["zero_cmd"]["threshold"] == 1e-3 and ":3" in pk["zero_cmd"]["expr"]
    for fn, total in (("run_actor", L.NOBS_ACTOR), ("run_critic", L.NOBS_CRITIC)):
        off = 0
        for e in pk[fn]["entries"]:
            name, sub, _ = PIECES[e["key"]]
            assert L.OBS[name][0] + sub == off, (fn, e["key"], off)
            w = packed_width(e)
            off += w
            assert {"JPOS": "normalize_joint_pos", "JVEL": "normalize_joint_vel", "PG": "encode_projected_gravity"}.get(name) == e["wrapper"], e
            assert e["divisor"] == (L.OBS_ACTFRC_DIV if name == "ACTFRC" else None), e
        assert off == total
    # the actor row reads the NOISY keys, the critic row the clean ones (train.py:1360-1363, 1388-1391)
    assert [e["key"] for e in pk["run_actor"]["entries"][:4]] == ["noisy_biased_joint_position", "noisy_joint_velocity", "noisy_imu_projected_gravity", "noisy_imu_gyro"]
    assert [e["key"] for e in pk["run_critic"]["entries"][:4]] == ["joint_position", "joint_velocity", "projected_gravity", "imu_gyro"]
    # every piece is contiguous and the table covers the critic row exactly once
    cover = sorted(L.OBS.values())
    assert cover[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(cover, cover[1:])) and cover[-1][0] + cover[-1][1] == L.NOBS_CRITIC


def test_header_observation_offsets_equal_layout():
    import re
    hdr = open(os.path.join(os.path.dirname(HERE), "include", "kbj_model.h")).read()
    enums = {k: int(v) for k, v in re.findall(r"KBJ_OBS_([A-Z]+)\s*=\s*(\d+)", hdr)}
    assert enums == {k: v[0] for k, v in L.OBS.items()}
    assert float(re.search(r"KBJ_OBS_JVEL_DIV\s+([0-9.]+)f", hdr).group(1)) == L.OBS_JVEL_DIV
    assert float(re.search(r"KBJ_OBS_ACTFRC_DIV\s+([0-9.]+)f", hdr).group(1)) == L.OBS_ACTFRC_DIV


def reference_mirror_table(ref, m, fn):
    """The packed-row mirror out[k] = mul * in[src] + add implied by mirror_obs / mirror_cmd / mirror_joints (train.py:1574-1756)
    applied to the row `fn` packs (train.py:1351-1433)."""
    mir = ref["mirror"]
    keys = dict(mir["mirror_obs"]["keys"]); keys.update(mir["mirror_cmd"]["keys"])
    bias = list(m.joint_bias)
    rng = [max(b - lo, hi - b) for b, lo, hi in zip(m.joint_bias, m.joint_lo, m.joint_hi)]     # normalize_joint_pos, train.py:1329-1333
    table = {}
    for e in ref["packing"][fn]["entries"]:
        name, sub, raw_w = PIECES[e["key"]]
        base = L.OBS[name][0] + sub
        if e["key"] == "zero_cmd":       # |cmd[0:3]| is invariant under the sign flips of mirror_cmd
            assert all(abs(c[0]) == 1 and c[2] == i for i, c in enumerate(keys["unified_command"]["cols"][:3]))
            table[base] = (base, 1.0, 0.0)
            continue
        rec = keys[e["key"]]
        cols, per_row = rec["cols"], rec["per_row"]
        if per_row:                       # the same pattern for every body row (train.py:1650, 1668)
            cols = [[rec["cols"][i % per_row][0], rec["cols"][i % per_row][1], (i // per_row) * per_row + rec["cols"][i % per_row][2]] for i in range(raw_w)]
        assert len(cols) == raw_w, (e["key"], len(cols))
        src_of = lambda k, c: L.OBS[PIECES[k][0]][0] + PIECES[k][1] + c
        if e["wrapper"] == "encode_projected_gravity":
            assert [(s, c) for s, _, c in cols] == [(1, 0), (-1, 1), (1, 2)]      # g -> (g0, -g1, g2): roll = atan2(g1, -g2) flips, pitch and |g| do not
            for i, sg in enumerate((-1.0, 1.0, 1.0, -1.0, 1.0)):
                table[base + i] = (base + i, sg, 0.0)
            continue
        for i, (s, k, c) in enumerate(cols):
            src = src_of(k, c)
            if e["wrapper"] == "normalize_joint_pos":      # (s (x r_c + b_c) - b_i) / r_i
                table[base + i] = (src, s * rng[c] / rng[i], (s * bias[c] - bias[i]) / rng[i])
            else:                                            # plain pieces and pieces divided by the same constant on both sides
                table[base + i] = (src, float(s), 0.0)
    return table


@pytest.mark.parametrize("critic", [False, True])
def test_mirror_tables_match_the_reference(ref, model, model_full, critic):
    """The table the kernels apply (kbj_mirror_table = kbj_nn.hip build_mirror_tables) against the one derived from the reference text."""
    from kbot_joystick_amd.host import binding
    mj = ref["mirror"]["mirror_joints"]
    assert mj["perm"] == [5, 6, 7, 8, 9, 0, 1, 2, 3, 4] + list(range(10, 20)) and set(mj["sign"]) == {-1}
    for m in (model, model_full):
        src, mul, add = binding.mirror_table(m, critic)
        want = reference_mirror_table(ref, m, "run_critic" if critic else "run_actor")
        n = L.NOBS_CRITIC if critic else L.NOBS_ACTOR
        assert sorted(want) == list(range(n))
        for k in range(n):
            assert src[k] == want[k][0], (k, src[k], want[k])
            assert abs(mul[k] - want[k][1]) < 1e-6 and abs(add[k] - want[k][2]) < 1e-6, (k, mul[k], add[k], want[k])
        assert not mul[n:].any() and not add[n:].any()        # row padding mirrors to zero


def test_oracle_mirror_equals_the_table(ref, model):
    """oracle/nn.py mirrors by unpacking / mirroring / re-packing as train.py does; it must agree with the table on random rows."""
    import numpy as np
    import torch
    from kbot_joystick_amd.host import binding
    from oracle import nn as ON
    g = torch.Generator().manual_seed(0)
    for critic, fn, ld in ((False, ON.mirror_actor_obs, L.LD_ACTOR), (True, ON.mirror_critic_obs, L.LD_CRITIC)):
        x = torch.randn(7, ld, generator=g, dtype=torch.float64)
        pg = x[:, 42:45] / x[:, 42:45].norm(dim=-1, keepdim=True)        # a consistent (roll, pitch, unit gravity) block
        x[:, 40] = torch.atan2(pg[:, 1], -pg[:, 2]); x[:, 41] = torch.atan2(-pg[:, 0], torch.sqrt(pg[:, 1] ** 2 + pg[:, 2] ** 2)); x[:, 42:45] = pg
        x[:, 48] = (x[:, 49:52].norm(dim=-1) < 1e-3).double()
        src, mul, add = binding.mirror_table(model, critic)
        n = L.NOBS_CRITIC if critic else L.NOBS_ACTOR
        want = x[:, torch.from_numpy(src.astype(np.int64))] * torch.from_numpy(mul).double() + torch.from_numpy(add).double()
        got = fn(x, model)
        assert (got[:, :n] - want[:, :n]).abs().max() < 1e-6


def test_convert_step_fn_contract(ref):
    """convert.py:84-119: argument order of the deployed step function, its observation concatenation (= the actor row) and outputs."""
    sf = ref["convert"]["step_fn"]
    assert tuple(sf["args"]) == constants.STEP_FN_INPUTS
    assert [e["wrapper"] for e in sf["entries"]] == [e["wrapper"] for e in ref["packing"]["run_actor"]["entries"]]
    assert [e["var"] for e in sf["entries"]] == ["joint_angles", "joint_angular_velocities", "projected_gravity", "gyroscope", "cmd_zero", "command"]
    assert sf["returns"] == ["dist.mode()", "new_carry_flat"]
    assert ref["convert"]["carry_size_expr"].replace(" ", "") == "(depth*2*hidden_size+len(joint_names),)"
