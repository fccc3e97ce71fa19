#!/bin/bash
# usage: tools/pmc_gl.sh <outdir> <bench_gl args...>   (GPU box, repo root): counter passes over one staged-conv shape
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA \
  --output-format csv -d "$OUT" -o p1 -- python3 "$GRAFT_REPO_ROOT/tools/bench_gl.py" "$@" > "$OUT/p1.log" 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU \
  --output-format csv -d "$OUT" -o p2 -- python3 "$GRAFT_REPO_ROOT/tools/bench_gl.py" "$@" > "$OUT/p2.log" 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT \
  --output-format csv -d "$OUT" -o p3 -- python3 "$GRAFT_REPO_ROOT/tools/bench_gl.py" "$@" > "$OUT/p3.log" 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o kt -- python3 "$GRAFT_REPO_ROOT/tools/bench_gl.py" "$@" > "$OUT/kt.log" 2>&1 || exit 1
ls "$OUT"
