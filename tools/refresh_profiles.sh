#!/bin/bash
# Run on the GPU box from the repo root: regenerates the raw material for profiles/ under gpurun_out/refresh.
#   tools/refresh_profiles.sh [profile] [mfma] [workloads]     (default: all)
# profile:   kernel trace + the two counter passes over the same short bench command, and a kernel trace of six batch-1 calls
# mfma:      matrix-pipe occupancy passes (tools/pmc_mfma.sh) over the same command and over the 64 x 10 s shape in
#            both arithmetics (long rows: the matrix-core attention kernel dispatches there), traffic passes for bf16
# workloads: bench lines of the other shapes and of the plain-bf16 generator mode
# Every GPU step has its own time limit and the script stops at the first failure.
set -e
R="$PWD"; O="$R/gpurun_out/refresh"; mkdir -p "$O"
P="--steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-sub-results --serial-generator --no-overlap"
WHAT="${*:-profile mfma workloads}"
if [[ "$WHAT" == *profile* ]]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o kt -- python3 "$R/bench.py" $P > "$O/kt_bench.json" 2> "$O/kt.log"
  cd "$R"
  bash tools/pmc_traffic.sh "$O/pmc" python3 "$R/bench.py" $P
  cd /tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/b1" -o b1 -- python3 "$R/tools/b1_trace.py" > "$O/b1.log" 2>&1
  cd "$R"
fi
if [[ "$WHAT" == *mfma* ]]; then
  C3="--batch 64 --frames 861 --phonemes 160 --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-sub-results --serial-generator --no-overlap"
  bash tools/pmc_mfma.sh "$O/mf_c2" python3 "$R/bench.py" $P
  cd "$R"
  bash tools/pmc_mfma.sh "$O/mf_c3" python3 "$R/bench.py" $C3
  cd "$R"
  bash tools/pmc_mfma.sh "$O/mf_c3_bf16" python3 "$R/bench.py" $C3 --precision bf16
  cd "$R"
  bash tools/pmc_traffic.sh "$O/pmc_c3_bf16" python3 "$R/bench.py" $C3 --precision bf16
  cd "$R"
  bash tools/pmc_traffic.sh "$O/pmc_c3" python3 "$R/bench.py" $C3
  cd /tmp
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_c3_bf16" -o kt -- python3 "$R/bench.py" $C3 --precision bf16 > "$O/kt_c3_bf16_bench.json" 2> "$O/kt_c3_bf16.log"
  cd "$R"
fi
if [[ "$WHAT" == *workloads* ]]; then
  B="--no-cpu-baseline --no-latency"
  timeout -k 10 300 python3 bench.py --batch 64 --frames 861 --phonemes 160 $B > "$O/c3_f16x2.json" 2> "$O/c3_f16x2.log"
  timeout -k 10 300 python3 bench.py --batch 64 --frames 861 --phonemes 160 --precision bf16 $B > "$O/c3_bf16.json" 2> "$O/c3_bf16.log"
  timeout -k 10 300 python3 bench.py --workload c4 $B > "$O/c4_f16x2.json" 2> "$O/c4_f16x2.log"
  timeout -k 10 300 python3 bench.py --workload c4 --precision bf16 $B > "$O/c4_bf16.json" 2> "$O/c4_bf16.log"
  timeout -k 10 300 python3 bench.py --precision bf16 $B > "$O/c2_bf16.json" 2> "$O/c2_bf16.log"
fi
ls "$O"
