#!/bin/bash
# same-box A/B of one environment knob on the batch-1 latencies: usage tools/gpu_ab_b1.sh NAME VAR A B
name=$1; var=$2; A=$3; B=$4
out=gpurun_out/$name; mkdir -p $out; log=$out/log.txt; : > $log
for rep in 1 2 3; do
  for v in $A $B; do
    echo "$var=$v" >> $log
    env $var=$v timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-sub-results --no-cpu-baseline 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(c.get('batch1_latency_ms'), c.get('batch1_edit_latency_ms'), c.get('stream_first_window_ms'))" >> $log || exit 1
  done
done
cat $log
