#!/usr/bin/env python3
"""Diagnostic (library built with -DSLFP_PW_STAMPS): the idle time BETWEEN back-to-back launches of one k_pw_tiled layer,
from in-kernel s_memrealtime stamps (100 MHz, common to all launches): last workgroup end of launch i -> first workgroup
start of launch i + 1, plus the span the workgroups of each launch cover."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from cnns_slfp_quantization_amd import _lib, layer_specs
L = _lib.load(); dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1); stream = torch.cuda.current_stream().cuda_stream
cin, cout = int(sys.argv[1]), int(sys.argv[2])
specs = [s for s in layer_specs.conv_layers("mobilenetv1_imagenet224") if s.k[0] == 1 and s.c_in == cin and s.c_out == cout][:1]
l = bench.Layer(L, specs[0], 256, dev, 1, gen, 8, False); l.prepare(L, stream)
NL = 6
bufs = [torch.zeros(4096 * 16, dtype=torch.int64, device=dev) for _ in range(NL)]
for _ in range(3): l.run(L, stream)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for b in bufs:
    os.environ["SLFP_PW_DBG"] = hex(b.data_ptr())
    l.run(L, stream)
e1.record(); torch.cuda.synchronize()
print(f"{NL} back-to-back launches: {e0.elapsed_time(e1) * 1e3 / NL:.1f} us per launch by events")
prev_end = None
for i, b in enumerate(bufs):
    d = b.cpu().numpy().reshape(-1, 16); d = d[d[:, 0] != 0]
    s, e = d[:, 0].min(), d[:, 15].max()
    line = f"launch {i}: workgroups span {(e - s) / 100.0:6.1f} us"
    if prev_end is not None: line += f", idle since the previous launch's last workgroup {(s - prev_end) / 100.0:5.1f} us"
    print(line); prev_end = e
