#!/usr/bin/env python3
"""Diagnostic: the 27-layer float32-interface conv path at batch 256 as ONE stream of 27 launches per step vs the same batch
split into S image groups that run the 27 layers on S HIP streams concurrently (images are independent units: while one
group's kernel drains its last workgroups another group's kernel fills the CUs).   python profiles/two_streams.py [S ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from cnns_slfp_quantization_amd import _lib, layer_specs
L = _lib.load(); dev = torch.device("cuda", 0)
specs = layer_specs.conv_layers("mobilenetv1_imagenet224")
splits = [int(a) for a in sys.argv[1:]] or [1, 2, 4]

def build(batch, n):
    gen = torch.Generator(device=dev).manual_seed(1234)
    groups = []
    for g in range(n):
        layers = [bench.Layer(L, s, batch // n, dev, 0, gen, 8, False) for s in specs]
        st = torch.cuda.Stream()
        for l in layers:
            l.prepare(L, st.cuda_stream)
        groups.append((st, layers))
    torch.cuda.synchronize()
    return groups

JOIN = os.environ.get("JOIN", "0") == "1"   # fork / join per layer: what a drop-in module can do without knowing its neighbours

def timed(groups, steps=200, warm=100):
    evs = [torch.cuda.Event() for _ in groups]
    OFF = int(os.environ.get("OFFSET", "0"))   # group g runs OFF * g layers behind group 0 (pairs big layers with small ones)
    def step():
        nl = len(specs)
        for i in range(nl):            # interleave the groups' launches layer by layer
            for gi, (st, layers) in enumerate(groups):
                layers[(i - OFF * gi) % nl].run(L, st.cuda_stream)
            if JOIN and len(groups) > 1:
                for (st, _), ev in zip(groups, evs):
                    ev.record(st)
                for k, (st, _) in enumerate(groups):
                    for j, ev in enumerate(evs):
                        if j != k:
                            st.wait_event(ev)
    for _ in range(warm): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3

for r in range(2):
    for n in splits:
        g = build(256, n)
        ms = timed(g)
        print(f"round {r}: {n} stream(s) x {256 // n} images: {ms:.4f} ms/step = {256 / ms:.1f} k images/s", flush=True)
        del g; torch.cuda.empty_cache()
