"""Shader clock each kernel actually ran at: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the dispatch's duration, from one rocprofv3 counter pass
(kernels run one at a time under --pmc). usage (GPU box): python3 tools/clock_probe.py <outdir> -- python3 tools/bench_ppo.py"""
import csv, glob, os, re, subprocess, sys
from collections import defaultdict

argv = sys.argv[1:]
i = argv.index("--")
out, cmd = os.path.abspath(argv[0]), argv[i + 1:]
os.makedirs(out, exist_ok=True)
run = ["rocprofv3", "--pmc", "GRBM_GUI_ACTIVE", "--kernel-trace", "--output-format", "csv", "-d", out, "--"] + cmd
r = subprocess.run(run, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=open(os.path.join(out, "run.log"), "w"), stderr=subprocess.STDOUT, timeout=900)
if r.returncode != 0:
    raise SystemExit(f"clock_probe: rocprofv3 failed rc={r.returncode} (see {out}/run.log)")
acc = defaultdict(lambda: [0.0, 0.0, 0])
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != "GRBM_GUI_ACTIVE":
                continue
            k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", row["Kernel_Name"]))
            dur = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            a = acc[k]; a[0] += float(row["Counter_Value"]); a[1] += dur; a[2] += 1
print(f"{'kernel':64s} {'launches':>8s} {'avg us':>9s} {'GHz':>6s}")
for k, (cyc, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if ns > 0 and ns / n > 20000:     # kernels above 20 us: the counter's start / stop latency is a few us
        print(f"{k[:64]:64s} {n:8d} {ns / n / 1e3:9.1f} {cyc / 8 / ns:6.3f}")
