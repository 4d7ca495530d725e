#!/usr/bin/env python3
"""Per-block timing of the fused depthwise+pointwise kernel (csrc/conv_dwpw.hip) against the two BN-fused convs run
one after the other, MobileNetV1-224 blocks 1-4, batch 256, channels_last; a 1 GiB fill between launches keeps the
Infinity Cache cold.  SLFP_DWPW_* environment switches are read per launch."""
import os, sys
import numpy as np, torch, torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import utils.conv2d_func as cf
cf.options.dwpw_all = True
from cnns_slfp_quantization_amd import fusion, layer_specs
dev = torch.device("cuda", 0)
specs = layer_specs.conv_layers("mobilenetv1_imagenet224")
g = torch.Generator(device=dev).manual_seed(3)
flush = torch.empty(1 << 28, device=dev)
rows = []
for bi in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    sd, sp = specs[1 + 2 * bi], specs[2 + 2 * bi]
    mk = lambda s: cf.conv2d_Q(8, np.float64(s.Kw), np.float64(s.Ka))(s.c_in, s.c_out, s.k[0], np.float64(s.Kw), np.float64(s.Ka), s.stride[0], s.pad[0], groups=s.groups, bias=False)
    m = nn.Sequential(mk(sd), nn.BatchNorm2d(sd.c_out), nn.ReLU(), mk(sp), nn.BatchNorm2d(sp.c_out), nn.ReLU()).to(dev).eval().to(memory_format=torch.channels_last)
    x = (torch.randn((256, sd.c_in, sd.h, sd.w), generator=g, device=dev).abs() * 4 * sd.Ka).contiguous(memory_format=torch.channels_last)
    fusion.fuse_bn_relu(m)
    def timeit(mod, reps=5):
        ts = []
        with torch.no_grad():
            mod(x); torch.cuda.synchronize()
            for _ in range(reps):
                flush.fill_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); mod(x); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
        return float(np.median(ts))
    t_two = timeit(m)
    fusion.fuse_dw_pw(m)
    t_one = timeit(m)
    blk = [mod for mod in m if isinstance(mod, fusion.DwPwBlock)][0]
    print(f"block {bi + 1}: {sd.c_in:4d}@{sd.h} s{sd.stride[0]} -> {sp.c_out:4d}: two kernels {t_two:7.1f} us, one kernel {t_one:7.1f} us ({blk._last_kernel})")
