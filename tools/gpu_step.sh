#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c3
timeout -k 10 300 python3 bench.py --batch 64 --frames 861 --phonemes 160 --no-cpu-baseline --no-latency --steps 6 --warmup 2 > gpurun_out/c3/c3.json 2> gpurun_out/c3/c3.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c3 -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --batch 64 --frames 861 --phonemes 160 --no-cpu-baseline --no-latency --steps 3 --warmup 1 --serial-generator --no-overlap > $GRAFT_REPO_ROOT/gpurun_out/c3/kt.json 2> $GRAFT_REPO_ROOT/gpurun_out/c3/kt.err || exit 1
