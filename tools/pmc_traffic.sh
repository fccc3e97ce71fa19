#!/bin/bash
# usage: tools/pmc_traffic.sh <outdir> <program> [args...]      (run on the GPU box from the repo root)
# Two separate counter passes (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2) over the same command.
set -e
OUT="$1"; shift; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT" -o rd -- "$@" > "$OUT/rd.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT" -o wr -- "$@" > "$OUT/wr.log" 2>&1
ls "$OUT"
