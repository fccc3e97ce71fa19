#!/bin/bash
# Run in the build container after `gpurun -- bash tools/refresh_profiles.sh profile` has merged gpurun_out/refresh:
# turns the raw rocprofv3 output into the summaries kept under profiles/ (round tag as first argument, default r02).
set -e
R=${1:-r02}
P="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-sub-results --serial-generator --no-overlap"
python3 tools/summarize_rocprof.py gpurun_out/refresh/kt profiles/${R}_f16x2_kernel_stats.md \
  "$P (f16x2 default; LDS-DMA convs for the 64/128/256-channel ResBlock layers, fused pairs at 32 channels and 64-channel k = 3)"
cp gpurun_out/refresh/kt_bench.json profiles/${R}_f16x2_bench_under_rocprof.json
python3 tools/pmc_traffic_json.py gpurun_out/refresh/pmc profiles/${R}_pmc_traffic.json "python3 $P" > /dev/null
python3 tools/summarize_rocprof.py gpurun_out/refresh/b1 profiles/${R}_c1_batch1_kernel_stats.md \
  "tools/b1_trace.py: six batch-1 calls of a 3-second utterance"
ls -la profiles/${R}_*
