#!/bin/bash
# quick A/B after a conv_gl16_kernel change: its parity tests, single launches, the whole generator
out=gpurun_out/quick; mkdir -p $out; log=$out/log.txt; : > $log
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_ops.py -x -q -k "lds_dma or staged" --timeout 120 --timeout-method thread >> $log 2>&1 || { tail -20 $log; exit 1; }
for shape in "128 11 1 16512 16" "128 7 1 16512 16" "256 11 1 4128 16" "64 11 1 66048 16" "64 7 1 66048 16"; do
  for kind in res staged_out; do
      timeout -k 10 120 python3 tools/bench_gl.py $shape $kind 400 >> $log 2>&1 || exit 1
  done
done
timeout -k 10 180 python3 tools/bench_generator.py 16 258 30 >> $log 2>&1 || exit 1
