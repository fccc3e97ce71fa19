#!/bin/bash
# Run on the GPU box from the repo root: kernel trace and traffic counter passes of the plain-bf16 generator mode at
# 64 x 10 s, raw output under gpurun_out/refresh_bf16 (summarised by tools/summarize_rocprof.py / pmc_traffic_json.py).
set -e
R="$PWD"; O="$R/gpurun_out/refresh_bf16"; rm -rf "$O"; mkdir -p "$O"
P="--batch 64 --frames 861 --phonemes 160 --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-sub-results --serial-generator --no-overlap"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o kt -- python3 "$R/bench.py" $P > "$O/kt_bench.json" 2> "$O/kt.log"
cd "$R"
bash tools/pmc_traffic.sh "$O/pmc" python3 "$R/bench.py" $P
