#!/bin/bash
# usage: tools/build_variant.sh NAME SOURCE.hip "<extra hipcc flags>"
# A measurement build of the library: SOURCE compiled with the extra flags (knock-out / tuning defines), linked with
# the other objects of the default build into vits_amd/lib/variants/libvits_NAME.so.  Load it with VITS_AMD_LIB=<path>.
set -e
NAME="$1"; SRC="$2"; EXTRA="$3"
R="$(cd "$(dirname "$0")/.." && pwd)"
L="$R/vits_amd/lib"; V="$L/variants"; mkdir -p "$V"
BASE="$(basename "$SRC" .hip)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function $EXTRA \
  -c "$R/vits_amd/csrc/$BASE.hip" -o "$V/${BASE}_$NAME.o"
OBJS=""
for o in kernels_conv kernels_gl16 kernels_chain kernels_misc model; do
  if [ "$o" == "$BASE" ]; then OBJS="$OBJS $V/${BASE}_$NAME.o"; else OBJS="$OBJS $L/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$V/libvits_$NAME.so" $OBJS
rm -f "$V/${BASE}_$NAME.o"
echo "$V/libvits_$NAME.so"
