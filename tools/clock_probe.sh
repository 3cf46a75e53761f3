#!/bin/bash
# Samples the GPU clocks and power (rocm-smi) while the bench workload runs: are the kernels' launch times
# under concurrency a matter of shared CUs or of the chip's clock? usage: tools/clock_probe.sh
python tools/split_bench.py > gpurun_out/clock_bench.log 2>&1 &
BP=$!
sleep 6
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
  sleep 0.7
done
wait $BP
cat gpurun_out/clock_bench.log | tail -3
