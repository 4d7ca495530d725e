#!/usr/bin/env python3
"""Diagnostic (needs a library built with -DSLFP_PW_STAMPS, e.g. SLFP_EXTRA_HIPCC_FLAGS=-DSLFP_PW_STAMPS): where a
k_pw_tiled workgroup's time goes on one MobileNetV1 pointwise layer at batch 256.  16 s_memrealtime stamps (100 MHz)
per workgroup: 0 start, 1 table visible, 2 prologue done (first stage in LDS), 3..10 after each stage barrier, 12 loop
done, 13 last stage done, 14 stores issued, 15 stores drained."""
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
nwg = 4096
dbg = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
for _ in range(3): l.run(L, stream)
torch.cuda.synchronize()
os.environ["SLFP_PW_DBG"] = hex(dbg.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); l.run(L, stream); e1.record(); torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16)
d = d[d[:, 0] != 0]
t0 = d[:, 0].min()
us = (d - t0) / 100.0
print(f"kernel {l.kernel}: {e0.elapsed_time(e1) * 1e3:.1f} us by events, {len(d)} workgroups, last stamp at {us[:, 15].max():.1f} us")
names = ["start", "table visible", "prologue done"] + [f"stage {i} done" for i in range(8)] + ["(unused)", "loop done", "last stage done", "stores issued", "stores drained"]
order = np.argsort(us[:, 0])
first = order[: min(512, len(order))]; rest = order[min(512, len(order)):]
for grp, idx in (("first wave of workgroups (start < median)", first), ("later workgroups", rest)):
    if len(idx) == 0: continue
    print(grp, len(idx))
    prev = us[idx, 0]
    for i, n in enumerate(names):
        if i == 11 or (3 <= i <= 10 and d[idx[0], i] == 0): continue
        cur = us[idx, i]
        print(f"  {n:18s} at {cur.mean():7.2f} us (min {cur.min():6.2f} max {cur.max():6.2f})   +{(cur - prev).mean():6.2f} since previous")
        prev = cur
dur = us[:, 15] - us[:, 0]
print(f"workgroup lifetime mean {dur.mean():.2f} us, min {dur.min():.2f}, max {dur.max():.2f}")
