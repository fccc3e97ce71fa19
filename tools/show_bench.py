#!/usr/bin/env python3
"""Print the headline fields of a bench.py JSON line: usage  python tools/show_bench.py gpurun_out/<dir>/bench.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step", "batch1_latency_ms", "batch1_edit_latency_ms", "whole_path_tflops") if k in d})
r = d["roofline"]
print({k: r[k] for k in r if k != "all_conv_kernels"})
for k in r.get("all_conv_kernels", [])[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print("  ", k)
print(d.get("cpu_baseline"))
for s in d.get("sub_results", []):
    print({k: s[k] for k in ("value", "ms_per_step") if k in s}, s.get("config", {}).get("workload", s.get("workload", ""))[:70],
          s.get("roofline", {}).get("kernel"), s.get("roofline", {}).get("frac"))
