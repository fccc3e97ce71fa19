#!/bin/bash
# Batch-1 latency (bench.py's latency section) under different whole-ResBlock dispatch lists.
set -o pipefail
out=gpurun_out/${1:-b1}; mkdir -p "$out"
for list in "32x3,64x3" "32x3,64x3,32x7" "32x3,64x3,32x7,64x7" "32x3,64x3,32x7,64x7,32x11" "32x3,64x3,32x7,64x7,32x11,64x11" "none"; do
  echo "== VITS_CHAIN_F16X2=$list" >> "$out/b1.log"
  VITS_CHAIN_F16X2="$list" timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-sub-results --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('batch1_latency_ms', 'batch1_edit_latency_ms', 'stream_first_window_ms', 'value')})" >> "$out/b1.log" || exit 1
done
cat "$out/b1.log"
