#!/usr/bin/env python3
"""Diagnostic (library built with -DSLFP_PW_STAMPS -DSLFP_PW_STAMPS2): inside ONE 64-deep stage (stage 3) of k_pw_tiled,
wave 0 of every workgroup: stage start -> X(t+1) encoded and stored (includes the wait for its loads) -> first k-step's
MFMAs retired (includes the wait for its W fragments) -> second k-step -> stage barrier passed."""
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
dbg = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
for _ in range(3): l.run(L, stream)
torch.cuda.synchronize()
os.environ["SLFP_PW_DBG"] = hex(dbg.data_ptr())
l.run(L, stream); torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16); d = d[d[:, 0] != 0]
first = d[np.argsort(d[:, 0])[:512]]
for grp, a in (("first round (2 workgroups per CU)", first), ("all", d)):
    seg = np.diff(a[:, 3:8].astype(np.float64), axis=1) / 100.0
    names = ["W(k0) issue + wait X + encode + store", "k-step 0: wait W + 16 MFMAs", "k-step 1: issue/wait W + 16 MFMAs", "stage barrier"]
    print(grp, len(a), "workgroups; stage 3 total %.2f us" % seg.sum(1).mean())
    for n, c in zip(names, seg.T):
        print(f"  {n:42s} mean {c.mean():5.2f} us  (p10 {np.percentile(c, 10):5.2f}  p90 {np.percentile(c, 90):5.2f})")
