"""Pins for the CPU oracle's physics (oracle/kbj_oracle_physics.h).

The reference ships no tests and its physics (mujoco-mjx 3.3.5) is not installable offline, so the
oracle is pinned by independent known answers: the Random123 threefry vectors, a Jacobian-sum mass
matrix (model compiler, numpy), free fall, Newton's equation residual, energy conservation,
static weight on the feet, scipy rotations and fp32-vs-fp64 agreement.
"""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from kbot_joystick_amd.spec import compiler, layout as L
from oracle import oracle as O


def test_abi_sizes_match_python_mirror(model):
    import ctypes
    assert O.lib().kbj_cpu_sizeof_model() == ctypes.sizeof(L.Model)
    assert O.lib().kbj_cpu_sizeof_config() == ctypes.sizeof(L.Config)


def test_threefry_known_answers():
    # Random123 threefry2x32-20 KATs
    assert O.threefry(0, 0, 0, 0) == (0x6B200159, 0x99BA4EFE)
    assert O.threefry(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF) == (0x1CB996FC, 0xBB002BE7)
    assert O.threefry(0x13198A2E, 0x03707344, 0x243F6A88, 0x85A308D3) == (0xC4923A9C, 0x483DF7A0)


def test_total_mass(model, model_full):
    # <inertial> sums + the 4.19 g sphere MuJoCo infers for the base body
    assert abs(model.total_mass - 36.339 - 0.00419) < 2e-3
    assert abs(model_full.total_mass - 38.013 - 0.00419) < 2e-3


def _random_pose(model, rng, z=2.0):
    q = np.array(model.qpos0, np.float64)
    q[7:] = np.array(model.joint_bias) + rng.uniform(-0.4, 0.4, 20)
    q[3:7] = rng.normal(size=4)
    q[3:7] /= np.linalg.norm(q[3:7])
    q[2] = z
    return q


@pytest.mark.parametrize("which", ["kbot-headless", "kbot"])
def test_mass_matrix_vs_jacobian_sum(which, quiet_cfg):
    m = compiler.load_model(which)
    ep = O.default_params(m, quiet_cfg)
    rng = np.random.default_rng(0)
    for _ in range(5):
        q = _random_pose(m, rng)
        r = O.forward(m, quiet_cfg, ep, q, np.zeros(26))
        M = compiler.mass_matrix(m, q)
        assert np.abs(r["m"] - M).max() < 1e-6
        assert np.abs(r["m"] - r["m"].T).max() == 0
        assert np.linalg.eigvalsh(r["m"]).min() > 0


def test_kinematics_vs_compiler_fk(model, quiet_cfg):
    ep = O.default_params(model, quiet_cfg)
    rng = np.random.default_rng(1)
    q = _random_pose(model, rng)
    r = O.forward(model, quiet_cfg, ep, q, np.zeros(26))
    xpos, xquat, _ = compiler.forward_kinematics(model, q)
    assert np.abs(r["xpos"] - xpos).max() < 1e-7
    assert np.abs(r["xquat"] - xquat).max() < 1e-7


def test_free_fall(model, quiet_cfg):
    ep = O.default_params(model, quiet_cfg)
    q = np.array(model.qpos0, np.float64)
    q[2] = 3.0
    r = O.forward(model, quiet_cfg, ep, q, np.zeros(26))
    assert np.allclose(r["qacc"][:3], [0, 0, -9.81], atol=1e-6)
    assert np.abs(r["qacc"][3:]).max() < 1e-9          # frictionloss rows hold the joints, no contacts
    assert r["touch"].sum() == 0
    # z(t) = z0 - g t^2 / 2 with semi-implicit Euler: z_n = z0 - g dt^2 n(n+1)/2
    v = np.zeros(26)
    n = 50
    for _ in range(n):
        r = O.forward(model, quiet_cfg, ep, q, v, integrate=True)
        q, v = r["qpos_next"], r["qvel_next"]
    assert abs(q[2] - (3.0 - 9.81 * quiet_cfg.dt ** 2 * n * (n + 1) / 2)) < 1e-6


def test_newton_residual_with_contacts(model, quiet_cfg):
    """M qacc = qfrc_actuator - qfrc_bias + J^T f must hold for the solver's output."""
    ep = O.default_params(model, quiet_cfg)
    rng = np.random.default_rng(2)
    q = np.array(model.qpos0, np.float64)
    q[7:] = np.array(model.joint_bias)
    q[2] -= 0.05                                         # feet penetrate the floor
    v = rng.normal(size=26) * 0.3
    ctrl = rng.normal(size=20) * 5
    quiet_cfg.solver_iterations = 400                    # run CG to convergence (cold start): checks the solver maths
    r = O.forward(model, quiet_cfg, ep, q, v, ctrl=ctrl)
    assert r["efc_active"][40:].sum() == 32
    assert r["iters"][0] < 400                           # stopped on the gradient tolerance
    res = r["m"] @ r["qacc"] - (r["actfrc"] - r["bias"] + r["qfrc_con"])
    assert np.abs(res).max() < 1e-5 * np.abs(r["qfrc_con"]).max()
    assert (r["efc_force"][20:] >= 0).all()             # unilateral rows push only
    fl = np.array([ep[L.EP["FRICLOSS"] + 6 + u] for u in range(20)])
    assert (np.abs(r["efc_force"][:20]) <= fl + 1e-9).all()


def test_energy_conservation_in_flight(model, quiet_cfg):
    """Total energy is conserved up to the first-order error of semi-implicit Euler: drift halves with dt."""
    ep = O.default_params(model, quiet_cfg)
    ep[L.EP["FRICLOSS"]:L.EP["FRICLOSS"] + 26] = 0      # no dissipation
    drifts = []
    for dt, n in ((0.001, 200), (0.0005, 400)):
        rng = np.random.default_rng(3)
        q = np.array(model.qpos0, np.float64)
        q[7:] = np.array(model.joint_bias) + rng.uniform(-0.2, 0.2, 20)
        q[2] = 5.0
        v = rng.normal(size=26) * 0.5
        quiet_cfg.dt = dt
        e0 = None
        for _ in range(n):
            r = O.forward(model, quiet_cfg, ep, q, v, integrate=True)
            assert r["efc_active"].sum() == 0
            e = r["energy"].sum()
            e0 = e if e0 is None else e0
            q, v = r["qpos_next"], r["qvel_next"]
        drifts.append(e - e0)
        assert abs(e - e0) < 0.01 * r["energy"][0]
    assert abs(drifts[1] / drifts[0] - 0.5) < 0.05


def test_momentum_conservation_in_flight(model, quiet_cfg):
    """No external force but gravity: linear momentum changes by m g dt per step, whatever the joints do."""
    ep = O.default_params(model, quiet_cfg)
    rng = np.random.default_rng(4)
    q = _random_pose(model, rng, z=5.0)
    v = rng.normal(size=26)
    ctrl = rng.normal(size=20) * 10

    def momentum(r):
        mass = ep[L.EP["MASS"]:L.EP["MASS"] + 24].astype(np.float64)
        # body com velocity = cvel_lin + w x (xipos - tree com); use finite differences of subtree_com instead
        return r["subcom"][1] * mass.sum()
    quiet_cfg.dt = 1e-4                                  # the centre-of-mass path is parabolic up to O(dt)
    r0 = O.forward(model, quiet_cfg, ep, q, v, ctrl=ctrl, integrate=True)
    r1 = O.forward(model, quiet_cfg, ep, r0["qpos_next"], r0["qvel_next"], ctrl=ctrl, integrate=True)
    r2 = O.forward(model, quiet_cfg, ep, r1["qpos_next"], r1["qvel_next"], ctrl=ctrl)
    dt = quiet_cfg.dt
    acc = (momentum(r2) - 2 * momentum(r1) + momentum(r0)) / dt ** 2 / ep[L.EP["MASS"]:L.EP["MASS"] + 24].sum()
    assert np.allclose(acc, [0, 0, -9.81], atol=0.05)


def test_static_weight_on_feet(model, quiet_cfg):
    """After settling on the PD-held neutral pose the touch sensors carry the robot's weight."""
    quiet_cfg.num_envs = 1
    quiet_cfg.reset_joint_vel_scale = 0.0
    quiet_cfg.reset_joint_pos_scale = 0.0
    quiet_cfg.reset_base_vel_xy_scale = 0.0
    o = O.Oracle(model, quiet_cfg, seed=0, precision="f64")
    a, c, x = o.reset_all()
    act = np.array(model.joint_bias, np.float32)[None]
    tot = []
    for _ in range(32):
        aux = x.copy()
        a, c, x = o.step(act, aux)
        assert aux[0, L.AUX["DONE"]] == 0
        tot.append(c[0, 65] + c[0, 66])
    weight = model.total_mass * 9.81
    # the open-loop neutral pose stands still for ~0.3 s after the landing transient (then slowly tips over)
    assert abs(np.mean(tot[20:30]) - weight) < 0.03 * weight
    assert abs(o.es[0, L.ES["QVEL"] + 2]) < 0.05        # vertical velocity


def test_fp32_matches_fp64_one_step(model, quiet_cfg):
    ep = O.default_params(model, quiet_cfg)
    rng = np.random.default_rng(5)
    q = np.array(model.qpos0, np.float64)
    q[7:] = np.array(model.joint_bias) + rng.uniform(-0.1, 0.1, 20)
    q[2] -= 0.045
    v = rng.normal(size=26) * 0.2
    a = O.forward(model, quiet_cfg, ep, q, v, precision="f64", integrate=True)
    b = O.forward(model, quiet_cfg, ep, q, v, precision="f32", integrate=True)
    assert np.abs(a["qpos_next"] - b["qpos_next"]).max() < 1e-5
    assert np.abs(a["qvel_next"] - b["qvel_next"]).max() < 5e-3
    assert np.abs(a["m"] - b["m"]).max() < 1e-4


def test_gyro_and_imu_quat(model, quiet_cfg):
    ep = O.default_params(model, quiet_cfg)
    rng = np.random.default_rng(6)
    q = _random_pose(model, rng)
    v = np.zeros(26)
    v[3:6] = [0.3, -0.2, 0.5]                            # body-local angular velocity of the base
    r = O.forward(model, quiet_cfg, ep, q, v)
    # imu is welded to the torso which is welded (rotated about z by ~90 deg) to the base
    rot_base = Rotation.from_quat(np.roll(q[3:7], -1))
    rot_imu = Rotation.from_quat(np.roll(r["imuquat"], -1))
    w_world = rot_base.apply(v[3:6])
    assert np.allclose(rot_imu.inv().apply(w_world), r["gyro"], atol=1e-9)


def test_com_distance_vs_scipy_hull(model):
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(7)
    for _ in range(20):
        pts = np.zeros((8, 3))
        pts[:, :2] = rng.uniform(-0.3, 0.3, (8, 2))
        com = rng.uniform(-0.1, 0.1, 2)
        hull = ConvexHull(pts[:, :2])
        poly = pts[hull.vertices, :2]
        x, y = poly[:, 0], poly[:, 1]
        x1, y1 = np.roll(x, -1), np.roll(y, -1)
        cr = x * y1 - x1 * y
        area = cr.sum() / 2
        cx, cy = ((x + x1) * cr).sum() / (6 * area), ((y + y1) * cr).sum() / (6 * area)
        assert abs(O.com_distance(pts, com) - np.hypot(cx - com[0], cy - com[1])) < 1e-12
    # degenerate: all points collinear -> mean-point fallback (train.py:545-552)
    pts = np.zeros((8, 3))
    pts[:, 0] = np.linspace(-1, 1, 8)
    d = O.com_distance(pts, np.zeros(2))
    assert np.isfinite(d)
