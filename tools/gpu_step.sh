#!/bin/bash
set -o pipefail
for g in 1 0 1 0; do
VITS_GEN_GROUP=$g timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-results --no-latency --no-overlap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('c2 no-overlap group=$g', d['value'], d['ms_per_step'])" || exit 1
VITS_GEN_GROUP=$g VITS_GEN_STREAMS=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-results --no-latency --no-overlap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('c2 no-overlap 1 stream group=$g', d['value'], d['ms_per_step'])" || exit 1
done
