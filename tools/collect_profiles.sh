#!/bin/bash
# Run in the build container after `gpurun -- bash tools/refresh_profiles.sh profile mfma` has merged gpurun_out/refresh:
# turns the raw rocprofv3 output into the summaries kept under profiles/ (round tag as first argument, default r03).
set -e
R=${1:-r03}
P="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-sub-results --serial-generator --no-overlap"
C3="bench.py --batch 64 --frames 861 --phonemes 160 --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-sub-results --serial-generator --no-overlap"
python3 tools/summarize_rocprof.py gpurun_out/refresh/kt profiles/${R}_f16x2_kernel_stats.md \
  "$P (f16x2 default; LDS-DMA convs for the 64/128/256-channel ResBlock layers, whole-ResBlock kernel for the 3-tap ResBlocks at 32, 64 and 128 channels and for the 7-tap ones (first two pairs) at 32 and 64, fused pairs for the 11-tap ones at 32)"
cp gpurun_out/refresh/kt_bench.json profiles/${R}_f16x2_bench_under_rocprof.json
python3 tools/pmc_traffic_json.py gpurun_out/refresh/pmc profiles/${R}_pmc_traffic.json "python3 $P" > /dev/null
python3 tools/summarize_rocprof.py gpurun_out/refresh/b1 profiles/${R}_c1_batch1_kernel_stats.md \
  "tools/b1_trace.py: six batch-1 calls of a 3-second utterance"
if [ -d gpurun_out/refresh/mf_c2 ]; then
  python3 tools/pmc_mfma_table.py gpurun_out/refresh/mf_c2 profiles/${R}_mfma_occupancy.md "python3 $P" profiles/${R}_pmc_traffic.json > /dev/null
  python3 tools/pmc_traffic_json.py gpurun_out/refresh/pmc_c3_bf16 profiles/${R}_pmc_traffic_c3_bf16.json "python3 $C3 --precision bf16" > /dev/null
  if [ -d gpurun_out/refresh/pmc_c3 ]; then
    python3 tools/pmc_traffic_json.py gpurun_out/refresh/pmc_c3 profiles/${R}_pmc_traffic_c3.json "python3 $C3" > /dev/null
  fi
  python3 tools/pmc_mfma_table.py gpurun_out/refresh/mf_c3 profiles/${R}_mfma_occupancy_c3.md "python3 $C3" > /dev/null
  python3 tools/pmc_mfma_table.py gpurun_out/refresh/mf_c3_bf16 profiles/${R}_mfma_occupancy_c3_bf16.md "python3 $C3 --precision bf16" profiles/${R}_pmc_traffic_c3_bf16.json > /dev/null
  python3 tools/summarize_rocprof.py gpurun_out/refresh/kt_c3_bf16 profiles/${R}_c3_bf16_kernel_stats.md \
    "$C3 --precision bf16 (plain-bf16 generator: whole-ResBlock kernel and fused pairs at 32 and 64 channels, the one-plane LDS-DMA conv at 128 / 256)"
fi
ls -la profiles/${R}_*
