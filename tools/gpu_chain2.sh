#!/bin/bash
# Whole-ResBlock kernel: parity tests, then timings of its builds (VITS_CHAIN_BUILD_32 / _64) at the C2 and C3 shapes.
set -o pipefail
out=gpurun_out/${1:-chain2}
mkdir -p "$out"
timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py -k "chain" -x -q --timeout 120 --timeout-method thread > "$out/tests.log" 2>&1 || { tail -30 "$out/tests.log"; exit 1; }
for b in 2 1; do
VITS_CHAIN_BUILD_32=$b VITS_CHAIN_BUILD_64=1 timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py -k "chain_fused" -x -q --timeout 120 --timeout-method thread > "$out/tests_b$b.log" 2>&1 || { tail -30 "$out/tests_b$b.log"; exit 1; }
done
tail -2 "$out/tests.log" "$out/tests_b2.log" "$out/tests_b1.log"
run() { echo "== $1" >> "$out/bench.log"; shift; env "$@" 2>&1 | grep "^chain\|^pairs" >> "$out/bench.log" || exit 1; }
for b in 0 2; do
  run "build32=$b" VITS_CHAIN_BUILD_32=$b timeout -k 10 120 python tools/bench_chain.py f16x2 32 3 132096 16 5
  run "build32=$b" VITS_CHAIN_BUILD_32=$b timeout -k 10 120 python tools/bench_chain.py f16x2 32 7 132096 16 5
done
for b in 0 1; do
  run "build64=$b" VITS_CHAIN_BUILD_64=$b timeout -k 10 120 python tools/bench_chain.py f16x2 64 3 66048 16 5
done
for b in 0 1 2; do
  run "build32=$b" VITS_CHAIN_BUILD_32=$b timeout -k 10 120 python tools/bench_chain.py bf16 32 3 440832 64 5
  run "build32=$b" VITS_CHAIN_BUILD_32=$b timeout -k 10 120 python tools/bench_chain.py bf16 32 7 440832 64 5
done
run "bf16 64" timeout -k 10 120 python tools/bench_chain.py bf16 64 3 220416 64 5
run "bf16 64" timeout -k 10 120 python tools/bench_chain.py bf16 64 7 220416 64 5
run "build32=1" VITS_CHAIN_BUILD_32=1 timeout -k 10 120 python tools/bench_chain.py bf16 32 11 440832 64 5
grep -v "^pairs.*acc=0" "$out/bench.log"
