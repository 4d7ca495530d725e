"""Per-layer times of MobileNetV1-224 (batch 256) on the float32 interface (slfp_conv2d_fwd_post) and on 1-byte codes
(slfp_conv2d_fwd_codes), each layer on its own synthetic input, layers in net order (cold caches), HIP events.

    python profiles/codes_layers.py [--batch 256] [--reps 10] [--qbits 8]
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cnns_slfp_quantization_amd import _lib, layer_specs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--qbits", type=int, default=8)
    ap.add_argument("--net", default="mobilenetv1_imagenet224")
    args = ap.parse_args()
    L = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    specs = layer_specs.conv_layers(args.net)
    gen = torch.Generator(device=dev).manual_seed(1)
    fmt = (_lib.FMT_ACT8 if args.qbits == 8 else _lib.FMT_SFP7) | _lib.FMT_EXT
    n = args.batch
    layers = []
    for i, s in enumerate(specs):
        d = _lib.ConvDesc(n=n, c_in=s.c_in, h=s.h, w=s.w, c_out=s.c_out, kh=s.k[0], kw=s.k[1], stride_h=s.stride[0],
                          stride_w=s.stride[1], pad_h=s.pad[0], pad_w=s.pad[1], dil_h=1, dil_w=1, groups=s.groups,
                          x_layout=_lib.LAYOUT_NHWC, y_layout=_lib.LAYOUT_NHWC, qbits=args.qbits, ka=float(np.float32(s.Ka)),
                          kw_scale=float(np.float32(s.Kw)), mfma_passes=_lib.MFMA_F16X1, reserved=0)
        x = torch.randn((n, s.h, s.w, s.c_in), generator=gen, device=dev)
        x = (x if s.c_in == 3 else torch.relu(x)) * (4.0 * s.Ka)
        w = torch.randn((s.c_out, s.c_in // s.groups, s.k[0], s.k[1]), generator=gen, device=dev)
        fan = (s.c_in // s.groups) * s.k[0] * s.k[1]
        w.mul_(min(5.0 * s.Kw, 3.0 * (2.0 / fan) ** 0.5 + 2.0 * s.Kw))
        blob = torch.empty(L.slfp_conv2d_wprep_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
        _lib.check(L.slfp_conv2d_prepare_weights(ctypes.byref(d), w.data_ptr(), blob.data_ptr(), None, st))
        sc = torch.rand(s.c_out, generator=gen, device=dev) + 0.5
        sh = torch.randn(s.c_out, generator=gen, device=dev) * 0.1
        xc = None
        if s.c_in != 3:
            xc = torch.empty(x.shape, dtype=torch.uint8, device=dev)
            _lib.check(L.slfp_encode_f32(x.data_ptr(), xc.data_ptr(), x.numel(), d.ka, fmt, st))
        last = i == len(specs) - 1
        ka_next = specs[i + 1].Ka if not last else 1.0
        io = _lib.ConvIo(x_codes=0 if s.c_in == 3 else 1, y_codes=0 if last else 1, y_ka=float(np.float32(ka_next)), y_qbits=args.qbits)
        ok = L.slfp_conv2d_codes_supported(ctypes.byref(d), ctypes.byref(io), 0, 1)
        y32 = torch.empty((n, s.h_out, s.w_out, s.c_out), device=dev)
        yc = torch.empty((n, s.h_out, s.w_out, s.c_out), dtype=torch.uint8, device=dev) if not last else y32
        layers.append(dict(s=s, d=d, x=x, xc=xc, blob=blob, sc=sc, sh=sh, io=io, ok=ok, y32=y32, yc=yc, last=last))

    def run32(l):
        _lib.check(L.slfp_conv2d_fwd_post(ctypes.byref(l["d"]), l["x"].data_ptr(), l["blob"].data_ptr(), None, l["sc"].data_ptr(),
                                          l["sh"].data_ptr(), 1, l["y32"].data_ptr(), None, None, st))

    def runc(l):
        if not l["ok"]:
            return run32(l)
        xin = l["x"] if l["s"].c_in == 3 else l["xc"]
        _lib.check(L.slfp_conv2d_fwd_codes(ctypes.byref(l["d"]), ctypes.byref(l["io"]), xin.data_ptr(), l["blob"].data_ptr(), None,
                                           l["sc"].data_ptr(), l["sh"].data_ptr(), 1, l["yc"].data_ptr(), st))

    res = {}
    for name, fn in (("f32", run32), ("codes", runc), ("f32b", run32), ("codesb", runc)):
        for l in layers:
            fn(l)
        torch.cuda.synchronize()
        ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in layers] for _ in range(args.reps)]
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for k in range(args.reps):
            for i, l in enumerate(layers):
                ev[k][i][0].record(); fn(l); ev[k][i][1].record()
        t1.record()
        torch.cuda.synchronize()
        res[name] = [float(np.median([ev[k][i][0].elapsed_time(ev[k][i][1]) for k in range(args.reps)])) * 1e3 for i in range(len(layers))]
        res[name + "_total"] = t0.elapsed_time(t1) * 1e3 / args.reps
    print(f"{'layer':34s} {'f32 us':>8s} {'GB/s':>7s} | {'codes us':>8s} {'GB/s(1B)':>8s} {'speedup':>7s}")
    tot32 = totc = 0.0
    for i, l in enumerate(layers):
        s = l["s"]
        a = min(res["f32"][i], res["f32b"][i]); c = min(res["codes"][i], res["codesb"][i])
        b32 = 4 * n * (s.in_elems + s.out_elems) + 4 * s.w_elems
        bc = n * (s.in_elems * (4 if s.c_in == 3 else 1) + s.out_elems * (4 if l["last"] else 1)) + 4 * s.w_elems
        tot32 += a; totc += c
        kind = "dw" if s.groups > 1 else ("pw" if s.k[0] == 1 else "stem")
        print(f"{kind:4s} {s.c_in:4d}->{s.c_out:4d} s{s.stride[0]} {s.h:3d}->{s.h_out:3d} {'' if l['ok'] else '(f32 fallback)':14s} "
              f"{a:8.1f} {b32 / a / 1e3:7.0f} | {c:8.1f} {bc / c / 1e3:8.0f} {a / c:7.2f}")
    print(f"sum of layers: f32 {tot32:.0f} us, codes {totc:.0f} us; wall per pass: f32 {min(res['f32_total'], res['f32b_total']):.0f} us, "
          f"codes {min(res['codes_total'], res['codesb_total']):.0f} us -> {n / min(res['codes_total'], res['codesb_total']) * 1e6:.0f} img/s")


if __name__ == "__main__":
    main()
