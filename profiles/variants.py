#!/usr/bin/env python3
"""Same-process, interleaved A/B of launch-time environment switches of libslfp_hip.so, per layer.

    python profiles/variants.py --family dw3x3 --var SLFP_DW_OLD=1 --var SLFP_DW_ABLATE=1 --var SLFP_DW_ABLATE=2

Times every layer of the chosen kernel family (MobileNetV1-224, batch 256 by default) with HIP events, for the
default library behaviour and for each --var setting, in interleaved rounds (cdna guide rule 24: one process, one
device), and prints median microseconds and algorithmic GB/s.  Within a round the layers run ONCE each, in order, like a
bench.py step: a layer that is launched several times back to back finds its 100-200 MB of input and output in the
256 MiB Infinity Cache and looks 20-25 % faster than it is in the net (--warm restores that behaviour).  The library reads its switches once at load; this tool calls
slfp_debug_reload_switches() after changing the environment (SLFP_DW_OLD, SLFP_PW_*, SLFP_LONG_ENCODE with care: tables are cached).
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cnns_slfp_quantization_amd import _lib, layer_specs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="mobilenetv1_imagenet224")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--family", default="dw3x3")
    ap.add_argument("--var", action="append", default=[])
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--passes", type=int, default=1)
    ap.add_argument("--post", action="store_true")
    ap.add_argument("--warm", action="store_true", help="3 back-to-back launches per layer (Infinity-Cache-warm numbers)")
    ap.add_argument("--layers", default="", help="comma-separated indices into the family's layers (default all)")
    args = ap.parse_args()
    L = _lib.load()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(1234)
    stream = torch.cuda.current_stream().cuda_stream
    layers = [bench.Layer(L, s, args.batch, dev, args.passes, gen, 8, args.post) for s in layer_specs.conv_layers(args.net)]
    layers = [l for l in layers if args.family in l.kernel]
    if args.layers:
        layers = [layers[int(i)] for i in args.layers.split(",")]
    for l in layers:
        l.prepare(L, stream)
    variants = [("default", {})] + [(v, dict(kv.split("=", 1) for kv in v.split(","))) for v in args.var]
    keys = {k for _, e in variants for k in e}
    times = {name: [[] for _ in layers] for name, _ in variants}
    for r in range(args.rounds + 1):
        for name, env in variants:
            for k in keys:
                os.environ.pop(k, None)
            os.environ.update(env)
            L.slfp_debug_reload_switches()   # the library reads its switches once at load; this re-reads them
            reps = 3 if args.warm else 1
            evs = []
            for i, l in enumerate(layers):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    l.run(L, stream)
                e1.record()
                evs.append((e0, e1))
                if args.warm:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            if r:
                for i, (e0, e1) in enumerate(evs):
                    times[name][i].append(e0.elapsed_time(e1) / reps * 1e3)
    print(f"{'layer':34s}" + "".join(f"{n:>26s}" for n, _ in variants))
    tot = {n: 0.0 for n, _ in variants}
    for i, l in enumerate(layers):
        s = l.spec
        row = f"{l.kernel:14s}{s.c_in:5d}->{s.c_out:<5d}s{s.stride[0]} {s.h:3d}  "
        for n, _ in variants:
            us = float(np.median(times[n][i]))
            tot[n] += us
            row += f"{us:10.1f} us {l.bytes / us / 1e3:7.0f} GB/s  "
        print(row)
    print(f"{'total us':34s}" + "".join(f"{tot[n]:26.1f}" for n, _ in variants))
    tb = sum(l.bytes for l in layers)
    print(f"{'GB/s':34s}" + "".join(f"{tb / tot[n] / 1e3:26.0f}" for n, _ in variants))


if __name__ == "__main__":
    main()
