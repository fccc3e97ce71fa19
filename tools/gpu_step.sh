#!/bin/bash
set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread 2>&1 | tail -3 || exit 1
B="--no-cpu-baseline --no-latency --no-sub-results"
timeout -k 10 300 python3 bench.py --precision bf16 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 bf16', d['value'], d['ms_per_step'])" || exit 1
timeout -k 10 300 python3 bench.py --batch 64 --frames 861 --phonemes 160 --precision bf16 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 bf16', d['value'], d['ms_per_step'])" || exit 1
timeout -k 10 300 python3 bench.py --precision bf16x3 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 bf16x3', d['value'], d['ms_per_step'])" || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-results 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2', d['value'], d['ms_per_step'], 'b1', d['batch1_latency_ms'], d['batch1_edit_latency_ms'])" || exit 1
