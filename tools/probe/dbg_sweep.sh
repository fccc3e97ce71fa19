for d in 0 8 16 24 25; do
  if [ $d = 0 ]; then lib=""; else lib="$PWD/vits_amd/lib/libvits_dbg$d.so"; fi
  echo "variant $d"
  VITS_AMD_LIB=$lib timeout -k 5 100 python3 tools/bench_conv.py --only "$1" | grep -E "rb" || exit 1
done
