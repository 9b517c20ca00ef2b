"""Per-phase shader-clock profile (mean over every 32nd env) inside env_step_kernel (needs the diagnostics build libkbj_stamps.so:
hipcc ... -DKBJ_ENV_STAMPS -c kbj_env.hip)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KBJ_LIB_NAME", "libkbj_stamps.so")
import numpy as np, torch
from kbot_joystick_amd.spec import compiler, layout as L
from kbot_joystick_amd.host import binding as B
N = 8192
m = compiler.load_model("kbot-headless"); cfg = L.default_config(num_envs=N, batch_size=512)
ctx = B.Context(m, cfg, 0, torch.cuda.current_stream().cuda_stream)
dev = "cuda:0"
a, c, x = torch.zeros(N, 68, device=dev), torch.zeros(N, 476, device=dev), torch.zeros(N, 72, device=dev)
a2, c2, x2 = torch.zeros_like(a), torch.zeros_like(c), torch.zeros_like(x)
ctx.env_reset_all(1, a, c, x)
act = torch.from_numpy(np.tile(np.array(m.joint_bias, np.float32), (N, 1))).cuda()
act = act + float(os.environ.get("KBJ_ACT_NOISE", "0.3")) * torch.randn(N, 20, device=dev, generator=torch.Generator(device=dev).manual_seed(0))   # a random-init policy's spread
for _ in range(20): ctx.env_step(act, x, a2, c2, x2)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
ctx.lib.kbj_debug_env_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
ctx.lib.kbj_debug_env_stamps(buf, 1)
K = 50
for _ in range(K): ctx.env_step(act, x, a2, c2, x2)
torch.cuda.synchronize()
ctx.lib.kbj_debug_env_stamps(buf, 0)
names = {0: "pd + loop entry", 1: "kinematics", 2: "com/cinert/cdof", 3: "crb + mass matrix", 4: "collide + velocity pass", 5: "smooth forces (RNE)",
         6: "constraint rows", 7: "solve: M factor + qacc_smooth", 8: "solve: warm start choice", 9: "newton: forces+gradient", 10: "newton: hessian+factor+solve",
         11: "newton: M*search, J*search", 12: "newton: line search", 13: "newton: update", 15: "solve: exit (final forces)", 16: "sensors", 17: "integrate",
         18: "kernel entry: state load", 19: "tail: termination, reset, command, obs, store"}
tot = sum(buf[k] for k in range(32))
nenv = (N + 31) // 32   # every 32nd env adds its cycles
print(f"mean over {nenv} envs: {tot / K / nenv:.0f} cycles per control step (a wave shares its SIMD with 2 others)")
for k in range(32):
    if buf[k]:
        print(f"  [{k:2d}] {names.get(k, '?'):44s} {buf[k] / K / nenv:10.0f} cycles  {100.0 * buf[k] / tot:5.1f} %")
