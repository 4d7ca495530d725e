#!/usr/bin/env python3
"""Diagnostic: phase stamps of k_pw_xreg (conv_pw3.hip, SLFP_PW_DBG) for one 512->512 layer at 14x14, batch 256."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from cnns_slfp_quantization_amd import _lib, layer_specs
L = _lib.load(); dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1); stream = torch.cuda.current_stream().cuda_stream
specs = [s for s in layer_specs.conv_layers("mobilenetv1_imagenet224") if s.k[0] == 1 and s.c_in == int(sys.argv[1]) and s.c_out == int(sys.argv[2])][:1]
l = bench.Layer(L, specs[0], 256, dev, 1, gen, 8, False); l.prepare(L, stream)
dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
for _ in range(3): l.run(L, stream)
torch.cuda.synchronize()
os.environ["SLFP_PW_DBG"] = hex(dbg.data_ptr())
l.run(L, stream); torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 8)
d = d[d[:, 0] != 0]
t0 = d[:, 0].min()
rel = (d[:, :5] - t0) / 100.0  # us
print("waves:", len(d))
for i, name in enumerate(["start", "X phase done", "first W stage ready", "main loop done", "end"]):
    print(f"{name:22s} mean {rel[:, i].mean():7.2f}  min {rel[:, i].min():7.2f}  max {rel[:, i].max():7.2f} us")
