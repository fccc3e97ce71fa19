#!/bin/bash
set -o pipefail
timeout -k 10 500 python -u -m pytest tests/test_gpu_infer.py -x -q --timeout 300 --timeout-method thread -k "golden or c2_batch or c1_full or stream or batches" 2>&1 | tail -3 || exit 1
for n in 2 3 1 2 3; do
  VITS_BATCHES_GEN_STREAMS=$n timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-results --no-latency 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gen_streams $n', d['value'], d['ms_per_step'])" || exit 1
done
