#!/bin/bash
# samples clocks / power while the bench runs (rocm-smi reads only). usage (GPU box): bash tools/smi_probe.sh [outdir]
O=${1:-gpurun_out/smi}
mkdir -p $O
python3 bench.py --steps 120 --warmup 3 --no-cpu-baseline --no-variants > $O/bench.json 2> $O/bench.err &
BP=$!
sleep 25
for i in $(seq 1 24); do
  rocm-smi --showclocks --showpower --showtemp --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|edge)|GPU use" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.7
done > $O/smi.txt
wait $BP
tail -c 300 $O/bench.json
rocm-smi --showmaxpower --showclkfrq 2>/dev/null | head -60 > $O/smi_static.txt
