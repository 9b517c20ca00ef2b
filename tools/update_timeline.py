"""Kernel timeline of ONE minibatch step of the update inside bench.py (from one adamw_kernel launch to the next: AdamW, the host's
index upload, gathers, both nets forward / backward, gradient norm) from a rocprofv3 --kernel-trace run.
usage (GPU box): cd /tmp && rocprofv3 --kernel-trace -d <dir> -o run -- python3 <repo>/bench.py --steps 1 --warmup 1 --no-cpu-baseline; python3 tools/update_timeline.py <dir>/run_results.db [k]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end,queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[k], idx[k + 1]
t0 = rows[a][1]
for r in rows[a:b + 1]:
    n = re.sub(r"\(.*", "", r[0].replace("kbj::", "").replace("(anonymous namespace)::", "").replace("void ", ""))
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f} q{r[3]} {n[:80]}")
