#!/bin/bash
# same-box A/B of one environment knob on the default benchmark: usage tools/gpu_ab.sh NAME VAR A B [bench args]
name=$1; var=$2; A=$3; B=$4; shift 4
out=gpurun_out/$name; mkdir -p $out; log=$out/log.txt; : > $log
for rep in 1 2 3; do
  for v in $A $B; do
    echo "$var=$v" >> $log
    env $var=$v timeout -k 10 300 python3 bench.py --steps 40 --warmup 10 --no-sub-results --no-latency --no-cpu-baseline "$@" 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $log || exit 1
  done
done
cat $log
