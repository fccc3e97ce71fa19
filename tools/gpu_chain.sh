#!/bin/bash
# One GPU-box pass for the whole-ResBlock kernel: its parity tests, then chain-against-pairs timings at the C2 and C3
# shapes of the 32- and 64-channel stages.  Every step has its own time limit; the chain stops at the first failure.
set -o pipefail
out=gpurun_out/${1:-chain}
mkdir -p "$out"
timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py -k "chain or pair" -x -q --timeout 120 --timeout-method thread > "$out/tests.log" 2>&1 || { tail -30 "$out/tests.log"; exit 1; }
tail -3 "$out/tests.log"
for cfg in "f16x2 32 3 132096 16" "f16x2 64 3 66048 16" "f16x2 32 7 132096 16" "bf16 32 3 440832 64" "bf16 64 3 220416 64" "bf16 32 7 440832 64" "bf16 64 7 220416 64" "bf16 32 11 440832 64" "bf16 64 11 220416 64"; do
  timeout -k 10 120 python tools/bench_chain.py $cfg 5 >> "$out/bench.log" 2>&1 || { tail -5 "$out/bench.log"; exit 1; }
done
cat "$out/bench.log"
