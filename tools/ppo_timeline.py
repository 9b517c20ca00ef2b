"""Kernel timeline of ONE kbj_ppo_grad call (start / end / duration / queue per launch) from a rocprofv3 --kernel-trace run of
tools/bench_ppo.py: the view that shows which chain of a minibatch is critical.
usage (GPU box): cd /tmp && rocprofv3 --kernel-trace -d <dir> -o run -- python3 <repo>/tools/bench_ppo.py ; python3 tools/ppo_timeline.py <dir>/run_results.db"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end,queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "gather_small" in r[0]]
a, b = idx[-2], idx[-1]
t0 = rows[a][1]
for r in rows[a - 3:b - 2]:
    n = re.sub(r"\(.*", "", r[0].replace("kbj::", "").replace("(anonymous namespace)::", "").replace("void ", ""))
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f} q{r[3]} {n[:70]}")
