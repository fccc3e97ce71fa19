#!/bin/bash
set -o pipefail
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 || exit 1
( time timeout -k 10 600 python bench.py > gpurun_out/default_bench.json 2> gpurun_out/default_bench.err ) 2>&1 | tail -4 || exit 1
python - <<'P'
import json
d=json.loads(open('gpurun_out/default_bench.json').read().strip().splitlines()[-1])
print(d['metric'], d['value'], d['unit'], d['ms_per_step'], d['steps'], d['warmup'], d['dtype'])
print(d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print([ (s['workload'][:10], s['value']) for s in d['sub_results']])
P
