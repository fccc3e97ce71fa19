#!/bin/bash
# usage: tools/pmc_mfma.sh <outdir> <program> [args...]      (run on the GPU box from the repo root)
# One counter pass (6 SQ + 1 GRBM counters) for per-kernel matrix-pipe occupancy and shader clock, and a kernel trace
# of the same command for the durations.
set -e
OUT="$1"; shift; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT" -o mf -- "$@" > "$OUT/mf.log" 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o kt -- "$@" > "$OUT/kt.log" 2>&1
ls "$OUT"
