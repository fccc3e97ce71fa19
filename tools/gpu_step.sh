#!/bin/bash
set -o pipefail
for g in 1 0 1 0; do
VITS_LN_FUSED=$g timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-results 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('c2 ln_fused=$g', d['value'], d['ms_per_step'], 'b1', d['batch1_latency_ms'], d['batch1_edit_latency_ms'], d['stream_first_window_ms'])" || exit 1
done
