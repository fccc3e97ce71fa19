#!/bin/bash
# Knock-out timings of the whole-ResBlock kernel: tools/bench_chain.py with each measurement build of the library.
#   tools/gpu_ko.sh OUTNAME "<bench_chain args>" lib1 lib2 ...
set -o pipefail
out=gpurun_out/$1; mkdir -p "$out"; cfg="$2"; shift 2
for v in "$@"; do
  lib="vits_amd/lib/variants/libvits_$v.so"; [ "$v" == "default" ] && lib="vits_amd/lib/libvits_amd.so"
  echo "== $v" >> "$out/ko.log"
  VITS_AMD_LIB="$PWD/$lib" timeout -k 10 120 python tools/bench_chain.py $cfg 5 2>&1 | grep -v amdgpu.ids | grep "^chain" >> "$out/ko.log" || exit 1
done
cat "$out/ko.log"
