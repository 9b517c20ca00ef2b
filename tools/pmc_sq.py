"""Per-kernel SQ / TCC counter summary from several rocprofv3 --pmc passes of ONE command.

    python3 tools/pmc_sq.py OUT_DIR [--kernel SUBSTR] [--set NAME] -- python3 tools/bench_env.py 8192

Each pass is its own `rocprofv3 --pmc ... --kernel-trace` run (never combined with --stats or the tracing domains; the program
itself comes directly after `--`). Counters the device does not list (`rocprofv3 -L`) are dropped from a pass instead of
failing it. A pass that times out stops the whole collection (no further GPU step after a timeout).
Output: OUT_DIR/summary.json = {kernel: {counter: mean per launch, ..., "launches": n}} and a markdown table on stdout.

Units (MI355X_MICROARCH.md, PMC slots): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_BUSY_CYCLES is per shader engine; FETCH_SIZE / WRITE_SIZE are KiB (FETCH_SIZE x2 on gfx950 for wide coalesced reads).
"""
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

SETS = {
    # 8 SQ slots per pass
    "sq": [
        ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"],
        ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
        ["SQ_INSTS_SMEM", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "SQ_ACTIVE_INST_VMEM", "SQ_INSTS_FLAT", "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_MISC"],
        ["SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_BRANCH", "SQ_INST_LEVEL_LDS", "SQ_INST_LEVEL_VMEM", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_LDS_LOAD", "SQ_INSTS_LDS_STORE", "SQ_IFETCH"],
        ["SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT", "SQ_BUSY_CU_CYCLES", "SQ_CYCLES", "SQ_LDS_ADDR_CONFLICT"],
    ],
    "mfma": [["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY"],
             ["GRBM_GUI_ACTIVE", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_VALU_MFMA_COEXEC_CYCLES"]],
    "hbm": [["FETCH_SIZE"], ["WRITE_SIZE"]],
}


def available_counters():
    try:
        out = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, timeout=120).stdout
    except Exception:
        return None
    names = set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", out))
    return names or None


def short_name(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", n)


def collect(directory):
    tot, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short_name(row["Kernel_Name"])
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k][row["Counter_Name"]] += 1
    return tot, cnt


def main():
    argv = sys.argv[1:]
    if "--" not in argv:
        raise SystemExit(__doc__)
    i = argv.index("--")
    opts, cmd = argv[:i], argv[i + 1:]
    out = os.path.abspath(opts[0])
    kernel = opts[opts.index("--kernel") + 1] if "--kernel" in opts else None
    sets = [opts[j + 1] for j, o in enumerate(opts) if o == "--set"] or ["sq", "hbm"]
    timeout = int(opts[opts.index("--timeout") + 1]) if "--timeout" in opts else 420
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    avail = available_counters()
    summary = defaultdict(dict)
    npass = 0
    for s in sets:
        for counters in SETS[s]:
            use = [c for c in counters if avail is None or c in avail]
            dropped = [c for c in counters if c not in use]
            if dropped:
                print(f"[pmc_sq] not listed by rocprofv3 -L, dropped: {dropped}", flush=True)
            if not use:
                continue
            d = os.path.join(out, f"pass{npass}")
            npass += 1
            run = ["rocprofv3", "--pmc", *use, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + cmd
            print("[pmc_sq]", " ".join(run), flush=True)
            try:
                r = subprocess.run(run, env=env, cwd="/tmp", timeout=timeout, stdout=open(os.path.join(out, f"pass{npass - 1}.log"), "w"), stderr=subprocess.STDOUT)
            except subprocess.TimeoutExpired:
                print("[pmc_sq] pass timed out: stopping", flush=True)
                sys.exit(3)
            if r.returncode != 0:
                print(f"[pmc_sq] pass failed rc={r.returncode} (see log); continuing with the next pass", flush=True)
                continue
            tot, cnt = collect(d)
            for k in tot:
                if kernel and kernel not in k:
                    continue
                for c in tot[k]:
                    summary[k][c] = tot[k][c] / cnt[k][c]
                    summary[k]["launches"] = cnt[k][c]
    with open(os.path.join(out, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    for k, v in summary.items():
        print(f"\n### {k} ({v.get('launches')} launches, mean per launch)")
        for c in sorted(v):
            if c != "launches":
                print(f"  {c:34s} {v[c]:16.1f}")


if __name__ == "__main__":
    main()
