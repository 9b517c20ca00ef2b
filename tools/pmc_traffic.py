"""HBM traffic per kernel launch from two rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/pmc_traffic.json

FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots). Both are reported in KiB. On gfx950 FETCH_SIZE tallies the
128-byte requests of wide coalesced reads at 64 bytes, i.e. reports half the bytes: it is doubled here, as the guide
prescribes (other access widths are uncalibrated, so treat the read side as an estimate within that factor).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(directory, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {directory}")
    for path in files:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = re.sub(r"^void ", "", row["Kernel_Name"]).replace("(anonymous namespace)::", "")
                name = re.sub(r"\(.*$", "", name)
                tot[name] += float(row["Counter_Value"])
                cnt[name] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for name in sorted(set(fetch) | set(write)):
        f_kib, nf = fetch.get(name, (0.0, 0))
        w_kib, nw = write.get(name, (0.0, 0))
        rd, wr = 2.0 * f_kib * 1024.0, w_kib * 1024.0
        out[name] = dict(hbm_bytes_per_launch=round(rd + wr), read_bytes=round(rd), write_bytes=round(wr), fetch_size_kib_raw=round(f_kib, 2),
                         write_size_kib_raw=round(w_kib, 2), launches=max(nf, nw), fetch_correction="x2 (gfx950)")
    # provenance: bench.py only reports these bytes while the kernel sources are the ones profiled
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import subprocess
    from bench import source_fingerprint
    try:
        git = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        git = None
    cmd = os.environ.get("KBJ_PROFILE_CMD") or ""
    m_steps, m_warm = re.search(r"--steps\s+(\d+)", cmd), re.search(r"--warmup\s+(\d+)", cmd)
    iters = (int(m_steps.group(1)) if m_steps else 3) + (int(m_warm.group(1)) if m_warm else 1) + 1      # + bench.py's instrumented roofline iteration
    out["_meta"] = dict(source_fingerprint=source_fingerprint(), git=git, command=cmd or None, iterations=iters)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
