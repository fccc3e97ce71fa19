#!/bin/bash
set -o pipefail
for v in "3" "" "3" ""; do
VITS_PAIR64="$v" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-results --no-latency 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('c2 pair64=[$v]', d['value'], d['ms_per_step'])
for k in d['roofline']['all_conv_kernels'][:16]:
    if ',3,' in k['kernel'] or ',3>' in k['kernel']: print(k)" || exit 1
done
