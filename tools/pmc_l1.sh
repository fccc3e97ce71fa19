#!/bin/bash
# usage: tools/pmc_l1.sh "<bench_conv --only filter>" <outdir>   (vector-L1 / L2-latency counters, one pass)
set -e
FILTER="$1"; OUT="$2"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD TA_TA_BUSY_sum GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT" -o l1 -- python3 "$GRAFT_REPO_ROOT/tools/bench_conv.py" --only "$FILTER" --iters 3 > "$OUT/l1.log" 2>&1
ls "$OUT"
