#!/bin/bash
# samples clocks / power while the bench runs (rocm-smi reads only)
mkdir -p gpurun_out/r06jj
python3 bench.py --steps 120 --warmup 3 --no-cpu-baseline --no-variants > gpurun_out/r06jj/bench.json 2> gpurun_out/r06jj/bench.err &
BP=$!
sleep 25
for i in $(seq 1 24); do
  rocm-smi --showclocks --showpower --showtemp --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|edge)|GPU use" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.7
done > gpurun_out/r06jj/smi.txt
wait $BP
tail -c 300 gpurun_out/r06jj/bench.json
rocm-smi --showmaxpower --showclkfrq 2>/dev/null | head -60 > gpurun_out/r06jj/smi_static.txt
