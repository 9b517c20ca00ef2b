"""GPU checks that live in stand-alone programs under tools/ (built from source with hipcc on the box that runs them) and belong in the
driver's `-m gpu` run:

* tools/wave_test: every wave primitive of csrc/kbj_wave.h and the whole arrow (LDL^T) solve of the env kernel's register solver on the GPU
  against the host emulation of the SAME source - bit for bit for the primitives; the solve, whose pivots go through v_rcp_f32 on the GPU and
  through a division in the emulation, to 1e-6 of the emulation's and 2e-6 of a double-precision dense solve (measured 2.2e-7 / 4.0e-7).
  This is what lets tests/test_emu_env.py (CPU) speak for the register solver the GPU runs.
* tools/gemm_bench 10: the operand range of the bf16 x3 split GEMM (kbj_config.gemm_bf16x3): as accurate as the exact fp32-MFMA kernel for
  operands scaled anywhere in 2^-100 .. 2^100, bounded loss below 2^-110 where the split's lower pieces enter the bf16 subnormal range.
"""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "tools")


def _run(cmd, timeout=600):
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    return out.returncode, out.stdout, out.stderr


def test_wave_primitives_and_arrow_solve_bit_identical_to_the_emulation():
    rc, so, se = _run(["make", "-C", os.path.join(TOOLS, "wave_test"), "-s"])
    assert rc == 0, (so[-500:], se[-1500:])
    rc, so, se = _run([os.path.join(TOOLS, "wave_test", "wave_test")], timeout=300)
    assert rc == 0 and "WAVE TEST PASSED" in so, (so[-2000:], se[-500:])
    lines = [l for l in so.splitlines() if l.strip()]
    prim = [l for l in lines if l.split()[1:2] == ["ok"]]
    assert len(prim) == 16, so                                   # sixteen primitives, every one reported lane-exact
    assert not any("FAIL" in l for l in lines), so
    solve = [l for l in lines if l.startswith("arrow_solve_w")]
    import re
    m = re.search(r"GPU vs emulation ([0-9.eE+-]+), GPU vs double dense solve ([0-9.eE+-]+)", solve[0]) if solve else None
    assert m and float(m.group(1)) < 1e-6 and float(m.group(2)) < 2e-6, solve     # relative to max |x| over 256 random arrow systems


def test_gemm_bf16x3_operand_range():
    rc, so, se = _run(["make", "-C", TOOLS, "-s", "gemm_bench"])
    assert rc == 0, (so[-500:], se[-1500:])
    rc, so, se = _run([os.path.join(TOOLS, "gemm_bench"), "10"], timeout=600)
    assert rc == 0 and "X3 RANGE TEST PASSED" in so, (so[-3000:], se[-500:])
    rows = [l for l in so.splitlines() if l.lstrip().startswith("A x 2^")]
    assert len(rows) == 24 and all(l.rstrip().endswith("ok") or "bounded loss" in l for l in rows), so
    strict = [l for l in rows if "bounded loss" not in l]
    assert len(strict) == 16                                       # every operand scale in 2^-100 .. 2^100: as accurate as the exact kernel
