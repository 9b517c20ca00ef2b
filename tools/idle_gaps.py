"""GPU idle time inside the timed iterations of bench.py from a rocprofv3 --kernel-trace run: the union of all kernel intervals (any
queue) against the wall span, and the largest gaps with the kernels on either side - shows whether the host ever starves the queues.
usage (GPU box): cd /tmp && rocprofv3 --kernel-trace -d <dir> -o run -- python3 <repo>/bench.py --steps 2 --warmup 1 --no-cpu-baseline; python3 tools/idle_gaps.py <dir>/run_results.db"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
name = lambda r: re.sub(r"\(.*", "", r[0].replace("kbj::", "").replace("void ", ""))[:48]
env = [i for i, r in enumerate(rows) if "env_step_kernel" in r[0]]
a, b = env[100], env[300 - 1]          # iterations 2 and 3 of the run (100 env steps each): rollout of it. 2 .. rollout of it. 3
a_idx, b_idx = a, [i for i, r in enumerate(rows) if "adamw_kernel" in r[0] and i > b][47]
t0, t1 = rows[a_idx][1], rows[b_idx][2]
busy_end, idle, gaps = rows[a_idx][1], 0, []
for i in range(a_idx, b_idx + 1):
    s, e = rows[i][1], rows[i][2]
    if s > busy_end:
        idle += s - busy_end
        gaps.append((s - busy_end, i))
    busy_end = max(busy_end, e)
span = t1 - t0
print(f"span {span / 1e6:.2f} ms (2 iterations), no kernel running for {idle / 1e6:.3f} ms = {100.0 * idle / span:.2f} %; {len(gaps)} gaps")
for g, i in sorted(gaps, reverse=True)[:12]:
    print(f"  {g / 1e3:8.1f} us  between {name(rows[i - 1])} and {name(rows[i])}")
hist = {}
for g, _ in gaps:
    k = 1 if g < 2e3 else 2 if g < 5e3 else 5 if g < 10e3 else 10 if g < 20e3 else 20 if g < 50e3 else 50
    hist[k] = hist.get(k, 0) + g
print("  idle by gap size (us, lower bound -> total ms):", {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
