#!/usr/bin/env python3
"""Time single dense 3x3 layers through the C ABI (encode pre-pass + GEMM kernel), default vs SLFP_DENSE_NORES (the per-tile
k_dense3x3 instead of the weights-resident k_dense3x3_res).  Runs on the GPU box:  python profiles/dense_layer_time.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnns_slfp_quantization_amd import _lib as lib

L = lib.load()
dev = torch.device("cuda:0")
LAYERS = [(64, 64, 224, 128), (64, 128, 112, 128), (64, 64, 56, 128), (16, 64, 55, 256), (64, 256, 13, 256)][:int(os.environ.get("DLT_LAYERS", "5"))]
VARIANTS = (("resident", None), ("per_tile", "SLFP_DENSE_NORES"), ("fp16_copy", "SLFP_DENSE_NOENCX"), ("resident2", None))[:int(os.environ.get("DLT_VARIANTS", "4"))]   # ablation runs (profiles/ablate_dense_res.sh): 1
for ci, co, h, n in LAYERS:
    d = lib.ConvDesc(n=n, c_in=ci, h=h, w=h, c_out=co, kh=3, kw=3, stride_h=1, stride_w=1, pad_h=1, pad_w=1, dil_h=1, dil_w=1, groups=1,
                     x_layout=lib.LAYOUT_NHWC, y_layout=lib.LAYOUT_NHWC, qbits=8, ka=0.2, kw_scale=0.03, mfma_passes=1, reserved=0)
    x = torch.relu(torch.randn((n, h, h, ci), device=dev))
    w = torch.randn((co, ci, 3, 3), device=dev) * 0.1
    blob = torch.empty(L.slfp_conv2d_wprep_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
    lib.check(L.slfp_conv2d_prepare_weights(ctypes.byref(d), w.data_ptr(), blob.data_ptr(), None, None))
    ws = torch.empty(L.slfp_conv2d_workspace_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
    y = torch.empty((n, h, h, co), device=dev)
    yc = torch.empty((n, h, h, co), dtype=torch.uint8, device=dev)
    res = {}
    for tag, env in VARIANTS:
        os.environ.pop("SLFP_DENSE_NORES", None)
        os.environ.pop("SLFP_DENSE_NOENCX", None)
        if env:
            os.environ[env] = "1"
        L.slfp_debug_reload_switches()
        for codes in (False, True):
            io = lib.ConvIo(x_codes=0, y_codes=1 if codes else 0, y_ka=0.3, y_qbits=8)
            def run():
                if not codes:
                    lib.check(L.slfp_conv2d_fwd_post(ctypes.byref(d), x.data_ptr(), blob.data_ptr(), None, None, None, 1, y.data_ptr(), None,
                                                     ws.data_ptr(), None))
                    return
                lib.check(L.slfp_conv2d_fwd_codes_ws(ctypes.byref(d), ctypes.byref(io), x.data_ptr(), blob.data_ptr(), None, None, None, 1,
                                                     yc.data_ptr(), ws.data_ptr(), None))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record(); torch.cuda.synchronize()
            res[(tag, codes)] = e0.elapsed_time(e1) / 20 * 1e3
    os.environ.pop("SLFP_DENSE_NORES", None)
    os.environ.pop("SLFP_DENSE_NOENCX", None)
    L.slfp_debug_reload_switches()
    gmac = n * h * h * ci * co * 9 / 1e9
    print(f"{ci}->{co} @{h} batch {n} ({gmac:.0f} GMAC): " + "  ".join(f"{t}{'/codes' if c else '/f32'} {v:.0f} us" for (t, c), v in res.items()), flush=True)
