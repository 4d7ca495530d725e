#!/usr/bin/env python3
"""Which depthwise+pointwise pairs should run as ONE kernel inside the whole MobileNetV1-224 (batch 256)?  In isolation
(dwpw_bench.py) only the first block breaks even; inside the net a fused block also spares the cache its intermediate."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
import utils.conv2d_func as cf
from cnns_slfp_quantization_amd import layer_specs
dev = torch.device("cuda", 0)
specs = layer_specs.conv_layers("mobilenetv1_imagenet224")
for pairs in ({(32, 1)}, {(32, 1), (64, 2)}, {(32, 1), (128, 1)}, {(32, 1), (64, 2), (128, 1), (128, 2)}, set()):
    cf.options.dwpw_pairs = pairs
    res = []
    for _ in range(2):
        w = bench.whole_net(specs, "mobilenetv1_imagenet224", 256, dev, 20)
        res.append((w["fused_bn_relu"], w["fused_dw_pw"], w["dw_pw_blocks_one_kernel"]))
    print(sorted(pairs), res)
