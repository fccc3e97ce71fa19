#!/bin/bash
# usage: tools/pmc_conv.sh "<bench_conv --only filter>" <outdir>   (run on the GPU box from the repo root)
# Two separate counter passes (<= 8 SQ counters each) over the conv micro-benchmark of one layer shape.
set -e
FILTER="$1"; OUT="$2"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA \
  --output-format csv -d "$OUT" -o p1 -- python3 "$GRAFT_REPO_ROOT/tools/bench_conv.py" --only "$FILTER" --iters 3 > "$OUT/p1.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA \
  --output-format csv -d "$OUT" -o p2 -- python3 "$GRAFT_REPO_ROOT/tools/bench_conv.py" --only "$FILTER" --iters 3 > "$OUT/p2.log" 2>&1
ls "$OUT"
