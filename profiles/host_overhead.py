#!/usr/bin/env python3
"""Host-side cost of one forward through the drop-in modules: MobileNetV1 with BN/ReLU folded
(fusion.fuse_bn_relu), eval / no_grad / channels_last, at small and large batch:

  * eager, options.plan_cache = False: descriptor, output shape, workspace size and kernel name are re-derived through
    the C ABI on every call and the workspace is a fresh allocation (the round-1 host path);
  * eager, plan cache on (default);
  * the same forward replayed as one hipGraph.

Prints wall microseconds per forward (host-bound when it exceeds the GPU time, which the hipGraph replay approximates).
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import utils.conv2d_func as cf  # noqa: E402
from cnns_slfp_quantization_amd import fusion, layer_specs  # noqa: E402
from cnns_slfp_quantization_amd.conv2d_func import options  # noqa: E402


def build(net, dev, classes, pool):
    layers = []
    specs = layer_specs.conv_layers(net)
    for s in specs:
        conv = cf.conv2d_Q(q_bit=8, Kw=np.float64(s.Kw), Ka=np.float64(s.Ka))(
            s.c_in, s.c_out, s.k[0], np.float64(s.Kw), np.float64(s.Ka), s.stride[0], s.pad[0], groups=s.groups, bias=False)
        layers += [conv, nn.BatchNorm2d(s.c_out), nn.ReLU(inplace=True)]
    model = nn.Sequential(*layers, nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(specs[-1].c_out, classes)).to(dev).eval()
    model = model.to(memory_format=torch.channels_last)
    fusion.fuse_bn_relu(model)
    return model


def wall(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    dev = torch.device("cuda", 0)
    print(f"{'net':28s} {'batch':>5s} {'no plan cache':>14s} {'plan cache':>11s} {'hipGraph':>9s}   (us per forward)")
    for net, hw, classes, batches in (("mobilenetv1_imagenet224", 224, 1000, (1, 8, 64, 256)),
                                      ("mobilenetv1_cifar32", 32, 100, (1, 128, 1024))):
        try:
            model = build(net, dev, classes, None)
        except Exception as e:  # noqa: BLE001
            print(net, "skipped:", e)
            continue
        for b in batches:
            x = torch.randn(b, 3, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
            iters = 200 if b <= 64 else 30
            with torch.no_grad():
                options.plan_cache = False
                t_off = wall(lambda: model(x), iters)
                options.plan_cache = True
                t_on = wall(lambda: model(x), iters)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    model(x)
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    y = model(x)
                t_g = wall(g.replay, iters)
                same = torch.equal(y, model(x))
            print(f"{net:28s} {b:5d} {t_off:14.1f} {t_on:11.1f} {t_g:9.1f}   graph==eager: {same}")


if __name__ == "__main__":
    main()
