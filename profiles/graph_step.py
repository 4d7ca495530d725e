#!/usr/bin/env python3
"""Diagnostic: the 27-layer float32-interface conv path of bench.py, eager launches vs ONE hipGraph replay per step
(same kernels, same buffers): what do the launch path and the inter-kernel gaps cost?   python profiles/graph_step.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from cnns_slfp_quantization_amd import _lib, layer_specs
L = _lib.load(); dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1234)
layers = [bench.Layer(L, s, 256, dev, 0, gen, 8, False) for s in layer_specs.conv_layers("mobilenetv1_imagenet224")]
s0 = torch.cuda.current_stream().cuda_stream
for l in layers: l.prepare(L, s0)
torch.cuda.synchronize()
def step(stream):
    for l in layers: l.run(L, stream)
def timed(fn, steps=200, warm=100):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3): step(side.cuda_stream)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    step(torch.cuda.current_stream().cuda_stream)
for r in range(3):
    e = timed(lambda: step(s0)); gr = timed(g.replay)
    print(f"round {r}: eager {e:.4f} ms/step = {256 / e:.1f} k images/s; hipGraph {gr:.4f} ms/step = {256 / gr:.1f} k images/s")
