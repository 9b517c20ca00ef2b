"""ctypes binding of libkbj.so (C ABI: include/kbj.h).

This is the only place the product touches native code. There is NO fallback: if the HIP library is
missing or a call fails, a KbjError is raised. torch is used only as the owner of device memory and
streams (tensor.data_ptr(), torch.cuda.current_stream()).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from ..spec import layout as L

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
LIB_PATH = os.path.join(_CSRC, os.environ.get("KBJ_LIB_NAME", "libkbj.so"))   # KBJ_LIB_NAME: A/B builds of the same ABI (diagnostics)


class KbjError(RuntimeError):
    pass


class Carry(C.Structure):
    _fields_ = [("actor_hc_d", C.c_void_p), ("critic_hc_d", C.c_void_p), ("lpf_d", C.c_void_p),
                ("actor_mirror_hc_d", C.c_void_p), ("critic_mirror_hc_d", C.c_void_p), ("lpf_mirror_d", C.c_void_p)]


class Traj(C.Structure):
    _fields_ = [("T", C.c_int32), ("N", C.c_int32),
                ("actor_obs_d", C.c_void_p), ("critic_obs_d", C.c_void_p), ("aux_d", C.c_void_p),
                ("action_d", C.c_void_p), ("logp_d", C.c_void_p), ("value_d", C.c_void_p), ("reward_d", C.c_void_p),
                ("carry0_actor_hc_d", C.c_void_p), ("carry0_critic_hc_d", C.c_void_p), ("carry0_lpf_d", C.c_void_p),
                ("carry0_actor_mirror_hc_d", C.c_void_p), ("carry0_critic_mirror_hc_d", C.c_void_p),
                ("carry0_lpf_mirror_d", C.c_void_p), ("reward_comps_d", C.c_void_p), ("qstate_d", C.c_void_p)]


class PpoVars(C.Structure):
    _fields_ = [("logp_d", C.c_void_p), ("value_d", C.c_void_p), ("entropy_d", C.c_void_p), ("action_std_d", C.c_void_p), ("action_mean_d", C.c_void_p)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("launches", C.c_int32), ("total_ms", C.c_float), ("flops", C.c_double)]


_vp, _i, _u32, _f, _sz = C.c_void_p, C.c_int, C.c_uint32, C.c_float, C.c_size_t
_cfgp = C.POINTER(L.Config)

# every symbol include/kbj.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "kbj_create": (_i, [C.POINTER(_vp), _vp, _sz, _cfgp, _i, _vp]),
    "kbj_destroy": (_i, [_vp]),
    "kbj_last_error": (C.c_char_p, [_vp]),
    "kbj_check_config": (_i, [_vp, _vp, _sz]),
    "kbj_set_advantage_sums": (_i, [_vp, _vp]),
    "kbj_ppo_prefetch": (_i, [_vp, _vp, _vp]),
    "kbj_sizeof_model": (_i, []),
    "kbj_sizeof_config": (_i, []),
    "kbj_sizeof_traj": (_i, []),
    "kbj_sizeof_carry": (_i, []),
    "kbj_synchronize": (_i, [_vp]),
    "kbj_env_reset_all": (_i, [_vp, _u32, _vp, _vp, _vp]),
    "kbj_env_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "kbj_env_record_state": (_i, [_vp, _vp]),
    "kbj_env_reset_where": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "kbj_env_set_command": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "kbj_env_get_qstate": (_i, [_vp, _vp, _vp]),
    "kbj_env_set_qstate": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kbj_env_get_state": (_i, [_vp, _vp, _vp]),
    "kbj_env_set_state": (_i, [_vp, _vp, _vp]),
    "kbj_env_get_reward_carry": (_i, [_vp, _vp]),
    "kbj_env_set_reward_carry": (_i, [_vp, _vp]),
    "kbj_rewards": (_i, [_vp, _vp, _i, _vp, _vp]),
    "kbj_param_count": (_sz, [_cfgp]),
    "kbj_actor_param_count": (_sz, [_cfgp]),
    "kbj_init_params": (_i, [_vp, _u32, _vp]),
    "kbj_mirror_table": (_i, [_vp, _sz, _i, _vp, _vp, _vp]),
    "kbj_policy_step": (_i, [_vp, _vp, _vp, _vp, C.POINTER(Carry), _u32, _u32, _i, _vp, _vp, _vp]),
    "kbj_carry_reset": (_i, [_vp, C.POINTER(Carry), _vp, _i]),
    "kbj_rollout": (_i, [_vp, _vp, C.POINTER(Carry), _u32, _u32, C.POINTER(Traj)]),
    "kbj_set_rollout_argmax": (_i, [_vp, _i]),
    "kbj_recurrence_residency": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "kbj_gae": (_i, [_vp, C.POINTER(Traj), _vp, _vp]),
    "kbj_ppo_grad": (_i, [_vp, _vp, C.POINTER(Traj), _vp, _i, _vp, _vp, _vp, _vp]),
    "kbj_ppo_forward": (_i, [_vp, _vp, C.POINTER(Traj), _vp, _i, C.POINTER(PpoVars)]),
    "kbj_stream_wait_actor_grad": (_i, [_vp, _vp]),
    "kbj_adamw_step": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _f]),
    "kbj_set_learning_rate": (_i, [_vp, _f]),
    "kbj_profile_begin": (_i, [_vp]),
    "kbj_profile_end": (_i, [_vp, C.POINTER(_f), C.POINTER(_i), C.POINTER(_f), C.POINTER(_i)]),
    "kbj_profile_kernel_stats": (_i, [_vp, C.POINTER(KernelStat), _i, C.POINTER(_i)]),
}


def build_library(force: bool = False) -> str:
    """Compile libkbj.so for gfx950 with hipcc (works without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", _CSRC, "-s", "-j4"])
    return LIB_PATH


_lib = None


def load_library_at(path: str) -> C.CDLL:
    """dlopen one build of the ABI (RTLD_LOCAL: several builds can live in one process, e.g. the A/B solver test) and type its symbols."""
    if not os.path.exists(path):
        raise KbjError(f"{path} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch bundles its own ROCm runtime: import it first so that libkbj.so binds to the SAME libamdhip64
    # (loading /opt/rocm's copy first leaves the process with two HIP runtimes and no visible device)
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the library does not match include/kbj.h
        fn.restype, fn.argtypes = res, args
    return lib


def load_library() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = load_library_at(LIB_PATH)
    return _lib


def check_config(cfg) -> str:
    """kbj_check_config (host-only): '' when the library serves this kbj_config, else the reason kbj_create would refuse it with."""
    why = C.create_string_buffer(512)
    rc = load_library().kbj_check_config(C.addressof(cfg), why, C.sizeof(why))
    return "" if rc == 0 else why.value.decode()


def mirror_table(model, critic: bool):
    """kbj_mirror_table: (src, mul, add) arrays of the packed-row mirror the kernels apply (host-only call, no device needed)."""
    import numpy as np
    n = L.LD_CRITIC if critic else L.LD_ACTOR
    src, mul, add = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    rc = load_library().kbj_mirror_table(C.addressof(model), C.sizeof(model), int(critic), src.ctypes.data, mul.ctypes.data, add.ctypes.data)
    if rc != n:
        raise KbjError(f"kbj_mirror_table returned {rc}, expected {n}")
    return src, mul, add


def _ptr(t):
    """Device/host pointer of a torch tensor or numpy array (must be contiguous)."""
    if t is None or isinstance(t, int):   # an int is a raw address (e.g. a column inside a row-major tensor)
        return t
    if hasattr(t, "data_ptr"):
        if not t.is_contiguous():
            raise KbjError("tensor passed to libkbj must be contiguous")
        return t.data_ptr()
    return t.ctypes.data


class Context:
    """RAII wrapper of kbj_ctx."""

    def __init__(self, model: L.Model, config: L.Config, device: int = 0, stream: int | None = None, lib: C.CDLL | None = None):
        self.lib = lib if lib is not None else load_library()
        if self.lib.kbj_sizeof_model() != C.sizeof(L.Model) or self.lib.kbj_sizeof_config() != C.sizeof(L.Config):
            raise KbjError("struct layout mismatch between spec/layout.py and libkbj.so")
        if self.lib.kbj_sizeof_traj() != C.sizeof(Traj) or self.lib.kbj_sizeof_carry() != C.sizeof(Carry):
            raise KbjError("struct layout mismatch between host/binding.py (Traj / Carry) and libkbj.so")
        self.model, self.config = model, config
        self._h = _vp()
        blob = C.string_at(C.addressof(model), C.sizeof(model))
        rc = self.lib.kbj_create(C.byref(self._h), blob, len(blob), C.byref(config), device, stream)
        if rc != 0:
            raise KbjError(self.lib.kbj_last_error(None).decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.kbj_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, name: str, *args):
        rc = getattr(self.lib, name)(self._h, *args)
        if rc != 0:
            raise KbjError(f"{name}: {self.lib.kbj_last_error(self._h).decode()}")

    # ---- thin typed helpers ----
    def synchronize(self):
        self.call("kbj_synchronize")

    def env_reset_all(self, seed, actor0, critic0, aux0):
        self.call("kbj_env_reset_all", seed, _ptr(actor0), _ptr(critic0), _ptr(aux0))

    def env_step(self, action, aux_t, actor_next, critic_next, aux_next, qstate_t=None):
        """qstate_t: optional [N][QSTATE SIZE] row that receives the step's state record (kbj_env_record_state, one-shot)."""
        if qstate_t is not None:
            self.call("kbj_env_record_state", _ptr(qstate_t))
        self.call("kbj_env_step", _ptr(action), _ptr(aux_t), _ptr(actor_next), _ptr(critic_next), _ptr(aux_next))

    def env_reset_where(self, mask, actor_next, critic_next, aux_next):
        self.call("kbj_env_reset_where", _ptr(mask), _ptr(actor_next), _ptr(critic_next), _ptr(aux_next))

    def env_set_command(self, mask, cmd, actor_next, critic_next, aux_next):
        """mask None = every env; cmd [N, 16] float32 on the device."""
        self.call("kbj_env_set_command", _ptr(mask) if mask is not None else None, _ptr(cmd), _ptr(actor_next), _ptr(critic_next), _ptr(aux_next))

    def env_get_qstate(self, qpos, qvel):
        """kbj_env_get_qstate: every env's qpos [N, 27] / qvel [N, 26] into device tensors (asynchronous)."""
        self.call("kbj_env_get_qstate", _ptr(qpos), _ptr(qvel))

    def env_set_qstate(self, mask, qpos, qvel, actor_next, critic_next, aux_next):
        """kbj_env_set_qstate: new qpos / qvel for the masked envs (None = all) + their next observation rows rewritten."""
        self.call("kbj_env_set_qstate", _ptr(mask) if mask is not None else None, _ptr(qpos), _ptr(qvel), _ptr(actor_next), _ptr(critic_next), _ptr(aux_next))

    def env_get_state(self):
        import numpy as np
        N = self.config.num_envs
        ep = np.zeros((N, L.EP["SIZE"]), np.float32)
        es = np.zeros((N, L.ES["SIZE"]), np.float32)
        self.call("kbj_env_get_state", _ptr(ep), _ptr(es))
        return ep, es

    def env_set_state(self, ep, es):
        import numpy as np
        ep = None if ep is None else np.ascontiguousarray(ep, np.float32)
        es = None if es is None else np.ascontiguousarray(es, np.float32)
        self.call("kbj_env_set_state", _ptr(ep), _ptr(es))

    def env_get_reward_carry(self):
        import numpy as np
        rc = np.zeros((self.config.num_envs, L.RC["SIZE"]), np.float32)
        self.call("kbj_env_get_reward_carry", _ptr(rc))
        return rc

    def env_set_reward_carry(self, rc):
        import numpy as np
        rc = np.ascontiguousarray(rc, np.float32)
        if rc.shape != (self.config.num_envs, L.RC["SIZE"]):
            raise KbjError(f"reward carry must be [{self.config.num_envs}][{L.RC['SIZE']}], got {rc.shape}")
        self.call("kbj_env_set_reward_carry", _ptr(rc))

    def rewards(self, aux, T, reward, comps=None):
        self.call("kbj_rewards", _ptr(aux), T, _ptr(reward), _ptr(comps))

    def param_count(self) -> int:
        return int(self.lib.kbj_param_count(C.byref(self.config)))

    def actor_param_count(self) -> int:
        return int(self.lib.kbj_actor_param_count(C.byref(self.config)))

    def init_params(self, seed, params):
        self.call("kbj_init_params", seed, _ptr(params))

    def policy_step(self, params, actor_obs, critic_obs, carry: Carry, seed, step_index, argmax, action, logp, value):
        self.call("kbj_policy_step", _ptr(params), _ptr(actor_obs), _ptr(critic_obs), C.byref(carry), seed, step_index,
                  int(argmax), _ptr(action), _ptr(logp), _ptr(value))

    def carry_reset(self, carry: Carry, done, stride):
        self.call("kbj_carry_reset", C.byref(carry), _ptr(done), stride)

    def rollout(self, params, carry: Carry, seed, first_step_index, traj: Traj):
        self.call("kbj_rollout", _ptr(params), C.byref(carry), seed, first_step_index, C.byref(traj))

    def set_rollout_argmax(self, argmax: bool):
        """kbj_set_rollout_argmax: the following rollout() calls act with the distribution's mode (validation rollouts)."""
        self.call("kbj_set_rollout_argmax", int(bool(argmax)))

    def recurrence_residency(self):
        """kbj_recurrence_residency: (workgroups of one persistent recurrence launch, launches in flight at once, resident workgroup slots)."""
        g, c, s = _i(0), _i(0), _i(0)
        self.call("kbj_recurrence_residency", C.byref(g), C.byref(c), C.byref(s))
        return g.value, c.value, s.value

    def gae(self, traj: Traj, adv, target):
        self.call("kbj_gae", C.byref(traj), _ptr(adv), _ptr(target))

    def ppo_grad(self, params, traj: Traj, env_idx, B, adv, target, grad, metrics):
        self.call("kbj_ppo_grad", _ptr(params), C.byref(traj), _ptr(env_idx), B, _ptr(adv), _ptr(target), _ptr(grad),
                  _ptr(metrics))

    def ppo_forward(self, params, traj: Traj, env_idx, B, logp, value, entropy=None, action_std=None, action_mean=None):
        """kbj_ppo_forward: the on-policy pass (no gradients) for the B envs `env_idx` names; outputs are [T][B](x20) in env_idx order."""
        out = PpoVars(_ptr(logp), _ptr(value), _ptr(entropy), _ptr(action_std), _ptr(action_mean))
        self.call("kbj_ppo_forward", _ptr(params), C.byref(traj), _ptr(env_idx), B, C.byref(out))

    def ppo_prefetch(self, traj: "Traj", env_idx):
        """kbj_ppo_prefetch: queue the parameter-independent gathers of the NEXT ppo_grad (same traj / env_idx pointers) now."""
        self.call("kbj_ppo_prefetch", C.byref(traj), _ptr(env_idx))

    def set_advantage_sums(self, sums):
        """kbj_set_advantage_sums: a float64 [3] device tensor (sum adv, sum adv^2, count) or None (default: each minibatch's own)."""
        self.call("kbj_set_advantage_sums", _ptr(sums))

    def stream_wait_actor_grad(self, hip_stream: int):
        self.call("kbj_stream_wait_actor_grad", hip_stream)

    def adamw_step(self, params, m, v, grad, step, grad_scale=1.0):
        self.call("kbj_adamw_step", _ptr(params), _ptr(m), _ptr(v), _ptr(grad), step, grad_scale)

    def set_learning_rate(self, lr: float):
        self.call("kbj_set_learning_rate", lr)

    def profile_begin(self):
        self.call("kbj_profile_begin")

    def profile_end(self):
        a, b, c, d = _f(), _i(), _f(), _i()
        self.call("kbj_profile_end", C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        arr, n = (KernelStat * 32)(), _i()
        self.call("kbj_profile_kernel_stats", arr, 32, C.byref(n))
        kernels = [dict(name=arr[k].name.decode(), launches=arr[k].launches, total_ms=arr[k].total_ms, flops=arr[k].flops) for k in range(n.value)]
        return dict(env_step_ms=a.value, env_step_launches=b.value, nn_ms=c.value, nn_launches=d.value, kernels=kernels)
