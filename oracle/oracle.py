"""ctypes wrapper of the CPU oracle (oracle/_build/libkbj_oracle.so) — TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package. Parity with the JAX reference is UNPINNED.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("KBJ_ORACLE_LIB") or os.path.join(_HERE, "_build", "libkbj_oracle.so")   # override: the sanitizer build (tests/test_sanitize.py)

DBG = dict(XPOS=(0, 72), XQUAT=(72, 96), M=(168, 676), QACC=(844, 26), QACC_SMOOTH=(870, 26), BIAS=(896, 26),
           EFC_FORCE=(922, 72), TOUCH=(994, 2), GYRO=(996, 3), SUBCOM=(999, 72), CINERT=(1071, 240), CVEL=(1311, 144),
           CONPOS=(1455, 24), CONDIST=(1479, 8), ENERGY=(1487, 2), ITERS=(1489, 1), IMUQUAT=(1490, 4), QFRC_CON=(1494, 26),
           ACTFRC=(1520, 26), EFC_ACTIVE=(1546, 72), EFC_AREF=(1618, 72), EFC_D=(1690, 72))


def build(force: bool = False) -> str:
    if os.environ.get("KBJ_ORACLE_LIB"):
        return _LIB_PATH
    srcs = [os.path.join(_HERE, f) for f in ("kbj_oracle.cpp", "kbj_oracle_physics.h", "kbj_oracle_task.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "kbj_model.h"))
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.kbj_cpu_com_distance.restype = C.c_double
        _lib.kbj_cpu64_com_distance.restype = C.c_double
    return _lib


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _ref(s):
    return C.byref(s)


class Oracle:
    """Stateful oracle over N envs. precision: 'f32' or 'f64' (state records are fp32 in both)."""

    def __init__(self, model, config, seed: int = 0, precision: str = "f32"):
        from kbot_joystick_amd.spec import layout as L
        self.L = L
        self.model, self.config, self.seed = model, config, seed
        self.suf = "" if precision == "f32" else "64"
        self.N = config.num_envs
        self.ep = np.zeros((self.N, L.EP["SIZE"]), np.float32)
        self.es = np.zeros((self.N, L.ES["SIZE"]), np.float32)
        self.rc = np.zeros((self.N, L.RC["SIZE"]), np.float32)
        self.rc[:, L.RC["CONTACT"]:L.RC["CONTACT"] + 2] = 1.0

    def _fn(self, name):
        return getattr(lib(), f"kbj_cpu{self.suf}_{name}")

    def new_obs(self):
        L = self.L
        return (np.zeros((self.N, L.LD_ACTOR), np.float32), np.zeros((self.N, L.LD_CRITIC), np.float32),
                np.zeros((self.N, L.AUX["SIZE"]), np.float32))

    def reset_all(self):
        a, c, x = self.new_obs()
        self._fn("reset_all")(_ref(self.model), _ref(self.config), C.c_uint32(self.seed), _p(self.ep), _p(self.es), _p(a), _p(c), _p(x))
        return a, c, x

    def step(self, action: np.ndarray, aux_t: np.ndarray):
        """action [N,20]; aux_t [N,72] is completed in place; returns next (actor, critic, aux) rows."""
        action = np.ascontiguousarray(action, np.float32)
        a, c, x = self.new_obs()
        self._fn("env_step")(_ref(self.model), _ref(self.config), C.c_uint32(self.seed), _p(self.ep), _p(self.es), _p(action),
                             _p(aux_t), _p(a), _p(c), _p(x))
        return a, c, x

    def step_diag(self, action: np.ndarray, aux_t: np.ndarray):
        """step() + per-env int32[4] diagnostics: (max Newton iterations, total iterations, active-contact history hash,
        force-carrying-row history hash) over the substeps of this control step."""
        action = np.ascontiguousarray(action, np.float32)
        a, c, x = self.new_obs()
        diag = np.zeros((self.N, 4), np.int32)
        self._fn("env_step_diag")(_ref(self.model), _ref(self.config), C.c_uint32(self.seed), _p(self.ep), _p(self.es), _p(action),
                                  _p(aux_t), _p(a), _p(c), _p(x), _p(diag, C.c_int32))
        return a, c, x, diag

    def rewards(self, aux: np.ndarray):
        """aux [T,N,72] -> (reward [T,N], components [T,N,12]); updates the reward carry."""
        T, N = aux.shape[:2]
        aux = np.ascontiguousarray(aux, np.float32)
        rew = np.zeros((T, N), np.float32)
        comps = np.zeros((T, N, self.L.NREW), np.float32)
        self._fn("rewards")(_ref(self.model), _ref(self.config), _p(aux), T, N, _p(self.rc), _p(rew), _p(comps))
        return rew, comps


def default_params(model, config) -> np.ndarray:
    from kbot_joystick_amd.spec import layout as L
    ep = np.zeros(L.EP["SIZE"], np.float32)
    lib().kbj_cpu_default_params(_ref(model), _ref(config), _p(ep))
    return ep


def forward(model, config, ep, qpos, qvel, ctrl=None, push=None, warm=None, precision="f64", integrate=False):
    """One forward pass for one env; returns dict of named float64 arrays (+ qpos_next/qvel_next if integrate)."""
    suf = "" if precision == "f32" else "64"
    qpos = np.ascontiguousarray(qpos, np.float64)
    qvel = np.ascontiguousarray(qvel, np.float64)
    ctrl = np.zeros(20) if ctrl is None else np.ascontiguousarray(ctrl, np.float64)
    warm = np.zeros(26) if warm is None else np.ascontiguousarray(warm, np.float64)
    pp = None if push is None else np.ascontiguousarray(push, np.float64)
    out = np.zeros(lib().kbj_cpu_dbg_size(), np.float64)
    qn, vn = np.zeros(27), np.zeros(26)
    getattr(lib(), f"kbj_cpu{suf}_forward")(_ref(model), _ref(config), _p(np.ascontiguousarray(ep, np.float32)), _p(qpos, C.c_double),
                                             _p(qvel, C.c_double), _p(ctrl, C.c_double), _p(pp, C.c_double), _p(warm, C.c_double),
                                             _p(out, C.c_double), _p(qn, C.c_double) if integrate else None,
                                             _p(vn, C.c_double) if integrate else None)
    res = {k.lower(): out[o:o + n].copy() for k, (o, n) in DBG.items()}
    res["m"] = res["m"].reshape(26, 26)
    res["xpos"] = res["xpos"].reshape(24, 3)
    res["xquat"] = res["xquat"].reshape(24, 4)
    res["subcom"] = res["subcom"].reshape(24, 3)
    res["cinert"] = res["cinert"].reshape(24, 10)
    res["cvel"] = res["cvel"].reshape(24, 6)
    res["conpos"] = res["conpos"].reshape(8, 3)
    if integrate:
        res["qpos_next"], res["qvel_next"] = qn, vn
    return res


def threefry(k0, k1, c0, c1):
    out = (C.c_uint32 * 2)()
    lib().kbj_cpu_threefry(C.c_uint32(k0), C.c_uint32(k1), C.c_uint32(c0), C.c_uint32(c1), out)
    return int(out[0]), int(out[1])


def com_distance(pts, com, precision="f64"):
    pts = np.ascontiguousarray(pts, np.float64).reshape(8, 3)
    com = np.ascontiguousarray(com, np.float64)
    fn = lib().kbj_cpu64_com_distance if precision == "f64" else lib().kbj_cpu_com_distance
    return float(fn(_p(pts, C.c_double), _p(com, C.c_double)))
