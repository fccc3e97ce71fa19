#!/bin/bash
# same-box A/B of two builds of the library on the default benchmark (throughput) and on the batch-1 latency:
# usage tools/gpu_ab_lib.sh NAME LIB_A LIB_B   ("" = the default build)
name=$1; A=$2; B=$3
out=gpurun_out/$name; mkdir -p $out; log=$out/log.txt; : > $log
for rep in 1 2 3; do
  for lib in "$A" "$B"; do
    echo "LIB=$lib" >> $log
    VITS_AMD_LIB=$lib timeout -k 10 300 python3 bench.py --steps 40 --warmup 10 --no-sub-results --no-cpu-baseline 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('batch1_latency_ms'), d['config'].get('batch1_edit_latency_ms'))" >> $log || exit 1
  done
done
cat $log
