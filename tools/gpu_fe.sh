#!/bin/bash
# Front-end f16x2 packing: the forced-path parity test, the full-size property tests, then C3 benches with and without it.
set -o pipefail
out=gpurun_out/${1:-fe}; mkdir -p "$out"
timeout -k 10 600 python -u -m pytest tests/test_gpu_infer.py -k "f16x2_match or c3_ or c4_ or rank_count" -x -q --timeout 400 --timeout-method thread > "$out/tests.log" 2>&1 || { tail -40 "$out/tests.log"; exit 1; }
tail -3 "$out/tests.log"
B="--no-cpu-baseline --no-latency --no-sub-results --steps 8 --warmup 2"
for fe in 1 0; do
  for prec in bf16 f16x2; do
    echo "== VITS_FE_F16X2=$fe $prec" >> "$out/bench.log"
    VITS_FE_F16X2=$fe timeout -k 10 300 python bench.py --batch 64 --frames 861 --phonemes 160 --precision $prec $B 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])" >> "$out/bench.log" || exit 1
  done
done
cat "$out/bench.log"
