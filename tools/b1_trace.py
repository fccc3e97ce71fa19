#!/usr/bin/env python3
"""Batch-1 loop for a kernel trace: a few infer calls of the 3-second utterance, nothing else.

usage (GPU box): rocprofv3 --kernel-trace -d out -o b1 -- python3 tools/b1_trace.py ; then tools/trace_sequence.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vits_amd import synthetic  # noqa: E402

dev = torch.device("cuda:0")
net = bench.build_model(dev)
b1 = bench.to_device(synthetic.config_c1(), dev)
for _ in range(6):
    bench.step(net, b1)
    torch.cuda.synchronize()
