#!/bin/bash
# Collects every profile the repo commits under profiles/ (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default bench  -> gpurun_out/<tag>/stats (summarised by tools/rocprof_summary.py)
#   2. PMC passes (each its own run, kernel-trace only): FETCH_SIZE, WRITE_SIZE (HBM traffic), MFMA / VALU busy of the update
#      kernels, SQ counters of env_step_kernel
# usage: tools/profile_all.sh <tag>      (then tools/profile_collect.py <tag> on the build box writes profiles/)
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
fail() { echo "profile_all.sh: step failed: $*" >&2; exit 1; }
# the diagnostics builds the stamp tools load (hipcc is on the GPU box too): never profile against a missing library
make -C $R/kbot-joystick_amd/csrc -s stamps bstamps > $O/make_diag.log 2>&1 || fail "make stamps bstamps"
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $O/stats.log 2>&1 || fail "rocprofv3 --stats"
python3 $R/tools/rocprof_summary.py $O/stats/run_results.db > $O/kernel_stats.md || fail rocprof_summary
python3 $R/tools/pmc_sq.py $O/pmc_bench --set hbm --set mfma -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants > $O/pmc_bench.txt 2>&1 || fail "pmc_sq bench"
KBJ_PROFILE_CMD="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants" python3 $R/tools/pmc_traffic.py $O/pmc_bench/pass0 $O/pmc_bench/pass1 > $O/pmc_traffic.json || fail pmc_traffic
python3 $R/tools/pmc_sq.py $O/pmc_env --kernel env_step --set sq --set hbm -- python3 $R/tools/bench_env.py 8192 > $O/pmc_env.txt 2>&1 || fail "pmc_sq env"
# the raw traces are hundreds of MB: only the summaries travel back (gpurun_out is capped at 64 MiB)
cp $O/pmc_bench/summary.json $O/pmc_bench_summary.json && cp $O/pmc_env/summary.json $O/pmc_env_summary.json || fail "copy summaries"
rm -rf $O/stats $O/pmc_bench $O/pmc_env
python3 $R/tools/bench_env.py 8192 > $O/bench_env.txt 2>&1 || fail bench_env
python3 $R/tools/env_stamps.py > $O/env_stamps.txt 2>&1 || fail env_stamps
python3 $R/tools/bench_ppo.py > $O/bench_ppo.txt 2>&1 || fail bench_ppo
# the bench line last, with this run's traffic profile in place (bench.py reports traffic only from a profile of the same sources)
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json || fail "copy pmc_traffic"
python3 $R/bench.py --steps 10 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err || fail bench
echo done
