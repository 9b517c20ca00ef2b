"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side native code of the test infrastructure: the host emulation of the env
kernel body (tests/emu: every LDS index of the lane-parallel phases, the emulated wave primitives of the register solver, the LDS
formulation of the solver) and the C++ oracle (fp32 and fp64). GPU AddressSanitizer is not available on the pool, so this is where
out-of-bounds indexing in the kernel body's phases would show. The sanitized libraries are loaded in a child process with libasan preloaded
(a sanitized .so cannot be dlopen'ed into an unsanitized interpreter otherwise)."""
import os
import subprocess
import sys

import pytest

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %(root)r)
from kbot_joystick_amd.spec import compiler, layout as L
from oracle import oracle as O
from tests import helpers as H
for robot, terrain in (("kbot-headless", 0.0), ("kbot", 0.05)):
    model = compiler.load_model(robot)
    N = 12
    cfg = L.default_config(num_envs=N, batch_size=N, **(dict(terrain_amp=terrain, terrain_wavelength=2.0) if terrain else {}))
    o32, o64 = O.Oracle(model, cfg, seed=3, precision="f32"), O.Oracle(model, cfg, seed=3, precision="f64")
    a0, c0, x0 = o32.reset_all(); o64.reset_all()
    emus = [C.CDLL(p) for p in %(emus)r]
    rng = np.random.default_rng(1)
    a1, c1, x1 = o32.new_obs()
    for lib in emus:                      # reset path of the kernel body
        ep, es = np.zeros_like(o32.ep), np.zeros_like(o32.es)
        lib.kbj_emu_reset_all(C.byref(model), C.byref(cfg), C.c_uint32(3), H.fptr(ep), H.fptr(es), H.fptr(a1), H.fptr(c1), H.fptr(x1))
        assert np.array_equal(ep, o32.ep)
    for t in range(30):                   # enough steps for pushes, command switches and a few terminations (big action noise)
        act = H.random_actions(model, rng, N, scale=1.0)
        for lib in emus:
            ep, es, aux = o32.ep.copy(), o32.es.copy(), x0.copy()
            lib.kbj_emu_env_step(C.byref(model), C.byref(cfg), C.c_uint32(3), H.fptr(ep), H.fptr(es), H.fptr(act), H.fptr(aux), H.fptr(a1), H.fptr(c1), H.fptr(x1))
            assert np.isfinite(es).all()
        o64.ep[:], o64.es[:] = o32.ep, o32.es
        aux32 = x0.copy()
        a0, c0, x0 = o32.step(act, aux32)
        o64.step(act, x0.copy())
    T = 4
    aux = np.zeros((T + 1, N, L.AUX["SIZE"]), np.float32); aux[:] = x0
    o32.rewards(aux[:T])
print("SANITIZE_OK")
'''


@pytest.mark.timeout(900)
def test_emulation_and_oracle_under_asan_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    emus = [H.emu_lib("reg", sanitize=True), H.emu_lib("lds", sanitize=True)]
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               KBJ_ORACLE_LIB=os.path.join(ROOT, "oracle", "_build", "libkbj_oracle_asan.so"), OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, emus=emus)], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0 and "SANITIZE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-6000:]
