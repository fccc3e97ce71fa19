#!/bin/bash
# same-box sweep of one environment knob over several values on the default benchmark: tools/gpu_sweep.sh NAME VAR v1 v2 ...
name=$1; var=$2; shift 2
out=gpurun_out/$name; mkdir -p $out; log=$out/log.txt; : > $log
for rep in 1 2; do
  for v in "$@"; do
    echo -n "$var=$v  " >> $log
    env $var=$v timeout -k 10 300 python3 bench.py --steps 40 --warmup 10 --no-sub-results --no-latency --no-cpu-baseline 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $log || exit 1
  done
done
cat $log
