#!/bin/bash
set -o pipefail
for sh in "64 861" "16 258" "64 1400" "1 258" "16 600"; do
  for m in 256 100000; do
    VITS_ATT_MFMA=$m timeout -k 10 100 python tools/bench_attention.py $sh 2>&1 | tail -1 || exit 1
  done
done
