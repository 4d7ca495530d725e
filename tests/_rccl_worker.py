"""Worker for tests/test_gpu_rccl.py: ONE rank on the one GPU of the box, backend "nccl" (= RCCL on ROCm).
Runs exactly what bench.py runs for N > 1 -- RCCL initialisation with a device id, the one-bucket broadcast of the weights'
u8 codes on DEVICE tensors, per-rank timing, the output gather, config 4's global-batch slicing -- and then builds the
kernels' weight blobs from the broadcast bucket and compares them byte for byte with the blobs built from the weights."""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cnns_slfp_quantization_amd import _lib, layer_specs, sharding  # noqa: E402


def main():
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)   # as bench.py: RCCL, communicator bound to the device
    assert dist.get_backend() == "nccl"
    L = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    # BASELINE config 4's slicing: global batch 1024 over the ranks (here one rank owns it all; 8 ranks own 128 each)
    assert sharding.per_rank_batch(128, 1024, 0, 1) == (0, 1024, True)
    assert [sharding.shard_range(1024, r, 8) for r in (0, 7)] == [(0, 128), (896, 1024)]
    specs = layer_specs.conv_layers("resnet50_imagenet224")
    gen = torch.Generator(device=dev).manual_seed(3)
    descs, ws, codes = [], [], []
    for s in specs:
        d = _lib.ConvDesc(n=2, c_in=s.c_in, h=s.h, w=s.w, c_out=s.c_out, kh=s.k[0], kw=s.k[1], stride_h=s.stride[0], stride_w=s.stride[1],
                          pad_h=s.pad[0], pad_w=s.pad[1], dil_h=1, dil_w=1, groups=s.groups, x_layout=_lib.LAYOUT_NHWC,
                          y_layout=_lib.LAYOUT_NHWC, qbits=8, ka=float(np.float32(s.Ka)), kw_scale=float(np.float32(s.Kw)),
                          mfma_passes=0, reserved=0)
        w = torch.randn((s.c_out, s.c_in // s.groups, s.k[0], s.k[1]), generator=gen, device=dev) * (4.0 * s.Kw)
        c = torch.empty(w.numel(), dtype=torch.uint8, device=dev)
        _lib.check(L.slfp_encode_f32(w.data_ptr(), c.data_ptr(), w.numel(), d.kw_scale, _lib.FMT_W8 | _lib.FMT_EXT, stream))
        descs.append(d); ws.append(w); codes.append(c)
    torch.cuda.synchronize()
    total = sum(int(c.numel()) for c in codes)
    assert total == 23454912                                    # ResNet-50's conv weights as 1-byte codes
    sent = [c.clone() for c in codes]
    sharding.broadcast_blobs(codes, src=0, force=True)           # ONE ncclBroadcast of a 23.5 MB device bucket
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(codes, sent))
    for d, w, c in zip(descs, ws, codes):
        n = L.slfp_conv2d_wprep_bytes(ctypes.byref(d))
        b1 = torch.zeros(n, dtype=torch.uint8, device=dev)
        b2 = torch.zeros(n, dtype=torch.uint8, device=dev)
        _lib.check(L.slfp_conv2d_prepare_weights(ctypes.byref(d), w.data_ptr(), b1.data_ptr(), None, stream))
        _lib.check(L.slfp_conv2d_prepare_weights_codes(ctypes.byref(d), c.data_ptr(), b2.data_ptr(), None, stream))
        assert torch.equal(b1, b2), "blob from the broadcast codes differs from the blob from the weights"
    times = sharding.rank_times(0.125, device=dev, force=True)   # all_gather of a float64 device tensor
    assert times == [0.125]
    logits = torch.arange(128 * 10, dtype=torch.float32, device=dev).reshape(128, 10)
    full = sharding.gather_outputs(logits, force=True)           # all_gather of the per-rank outputs
    assert torch.equal(full, logits)
    t = torch.tensor([3.5], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    assert float(t) == 3.5
    dist.destroy_process_group()
    print("rccl rank 0 ok", flush=True)


if __name__ == "__main__":
    main()
