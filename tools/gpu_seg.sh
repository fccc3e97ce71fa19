#!/bin/bash
# Segments of a ResBlock on the whole-ResBlock kernel (one pair, two pairs) against the pair kernel.
set -o pipefail
out=gpurun_out/${1:-seg}; mkdir -p "$out"
for cfg in "f16x2 32 7 132096 16 5 1" "f16x2 32 7 132096 16 5 5" "f16x2 32 7 132096 16 5 1,3" "f16x2 32 11 132096 16 5 1" "f16x2 32 11 132096 16 5 5" "f16x2 32 11 132096 16 5 1,3" "f16x2 32 11 132096 16 5 1,3,5" \
           "f16x2 64 7 66048 16 5 1" "f16x2 64 7 66048 16 5 5" "f16x2 64 7 66048 16 5 1,3" "f16x2 64 11 66048 16 5 1" "f16x2 64 11 66048 16 5 5" "f16x2 64 11 66048 16 5 1,3" \
           "bf16 64 11 220416 64 3 1" "bf16 64 11 220416 64 3 5" "bf16 64 11 220416 64 3 1,3"; do
  timeout -k 10 120 python tools/bench_chain.py $cfg 2>&1 | grep "^chain\|^pairs" >> "$out/bench.log" || exit 1
done
cat "$out/bench.log"
