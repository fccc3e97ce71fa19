#!/bin/bash
# usage: tools/pmc_lds.sh <outdir> <program> [args...]      (run on the GPU box from the repo root)
# One counter pass (7 SQ + 1 GRBM counters): LDS instruction counts, LDS-busy and bank-conflict cycles per kernel.
set -e
OUT="$1"; shift; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT" -o lds -- "$@" > "$OUT/lds.log" 2>&1
ls "$OUT"
