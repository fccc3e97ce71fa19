#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/b1n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o b1 -- python3 "$R/tools/b1_trace.py" > "$O/b1.log" 2>&1 || exit 1
