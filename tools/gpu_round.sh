#!/bin/bash
# One GPU-box pass: the GPU test suite, the no-alloc test alone in a fresh process, smoke(), the default benchmark line and a
# two-rank rehearsal on the one GPU.  Every step has its own time limit; the chain stops at the first failure.
set -o pipefail
out=gpurun_out/${1:-round}
mkdir -p "$out"
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > "$out/gpu_tests.log" 2>&1 \
  && timeout -k 10 200 python -u -m pytest tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > "$out/no_alloc.log" 2>&1 \
  && timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 \
  && timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err" \
  && timeout -k 10 200 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-sub-results --no-latency --no-cpu-baseline > "$out/bench_2r.json" 2> "$out/bench_2r.err"
rc=$?
tail -5 "$out/gpu_tests.log"; tail -3 "$out/no_alloc.log" 2>/dev/null; tail -c 600 "$out/bench.json" 2>/dev/null; tail -c 400 "$out/bench_2r.json" 2>/dev/null
echo "rc=$rc"
exit $rc
