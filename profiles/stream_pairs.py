#!/usr/bin/env python3
"""Diagnostic: which pairs of HIP streams (in creation order) run two image groups concurrently?  HIP maps streams to a small
number of hardware queues; two streams on one queue serialise.   python profiles/stream_pairs.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from cnns_slfp_quantization_amd import _lib, layer_specs
L = _lib.load(); dev = torch.device("cuda", 0)
specs = layer_specs.conv_layers("mobilenetv1_imagenet224")
gen = torch.Generator(device=dev).manual_seed(1234)
sets = [[bench.Layer(L, s, 128, dev, 0, gen, 8, False) for s in specs] for _ in range(2)]
s0 = torch.cuda.current_stream().cuda_stream
for ls in sets:
    for l in ls: l.prepare(L, s0)
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(8)]
def timed(a, b, steps=100, warm=30):
    def step():
        for i in range(len(specs)):
            sets[0][i].run(L, a.cuda_stream); sets[1][i].run(L, b.cuda_stream)
    for _ in range(warm): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
for i in range(8):
    row = []
    for j in range(8):
        row.append("   -  " if j <= i else f"{256 / timed(streams[i], streams[j]):6.1f}")
    print(f"stream {i}: " + " ".join(row), flush=True)
