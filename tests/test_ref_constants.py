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
