#!/bin/bash
out=gpurun_out/overlap; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python3 $GRAFT_REPO_ROOT/tools/overlap_trace.py 12 > $GRAFT_REPO_ROOT/$out/run.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$out/run.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 tools/overlap_analyze.py $f 30 > $out/analysis.txt 2>&1
cat $out/analysis.txt; tail -2 $out/run.log
# keep the merge small
find $out/trace -type f ! -name "*kernel_trace.csv" -delete
