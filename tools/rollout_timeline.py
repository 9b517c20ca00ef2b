"""Kernel timeline of one control step of kbj_rollout (between two env_step_kernel launches) from a rocprofv3 --kernel-trace run of bench.py.
usage (GPU box): cd /tmp && rocprofv3 --kernel-trace -d <dir> -o run -- python3 <repo>/bench.py --steps 1 --warmup 0 --no-cpu-baseline; python3 tools/rollout_timeline.py <dir>/run_results.db"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end,queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "env_step_kernel" in r[0]]
a, b = idx[50], idx[52]
t0 = rows[a][1]
for r in rows[a:b + 1]:
    n = re.sub(r"\(.*", "", r[0].replace("kbj::", "").replace("(anonymous namespace)::", "").replace("void ", ""))
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f} q{r[3]} {n[:70]}")
