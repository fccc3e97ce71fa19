#!/bin/bash
# scratch GPU-box step (edited per experiment); every command under its own time limit, chain stops at the first failure
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py -x -q --timeout 120 --timeout-method thread -k "attention" 2>&1 | tail -4 \
 && timeout -k 10 500 python -u -m pytest tests/test_gpu_infer.py tests/test_voice_conversion.py tests/test_packed_weights.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -4 \
 && timeout -k 10 200 python bench.py --steps 4 --warmup 1 --batch 64 --frames 861 --phonemes 160 --no-cpu-baseline --no-sub-results --no-latency > gpurun_out/b_att1.json 2>gpurun_out/b_att1.err \
 && VITS_ATT_MFMA=0 timeout -k 10 200 python bench.py --steps 4 --warmup 1 --batch 64 --frames 861 --phonemes 160 --no-cpu-baseline --no-sub-results --no-latency > gpurun_out/b_att0.json 2>gpurun_out/b_att0.err \
 && timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sub-results > gpurun_out/b_rs1.json 2>gpurun_out/b_rs1.err \
 && VITS_WN_RS=0 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sub-results > gpurun_out/b_rs0.json 2>gpurun_out/b_rs0.err
echo rc=$?
