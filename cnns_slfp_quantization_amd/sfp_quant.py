"""Host-side mirror of the reference's utils/sfp_quant.py on top of the HIP C ABI.

Same names, arguments and behaviour as the reference module so that
`from utils.sfp_quant import *` keeps working in nets_cifar/* and nets_imgnet/*:

    quantize_weight(k), quantize_act(k), quantize_layerout(k)   -> callable (tensor -> tensor)
    weight_quantize_func(q_bit), act_quantize_func(q_bit), layerout_quantize_func(q_bit)  (nn.Module)

and the star-export of torch / nn / F / np that the nets rely on (the reference nets use
`np` without importing it, e.g. nets_imgnet/mobilenetv1.py:16).

quantize_weight / quantize_act run ONE HIP pass (slfp_quantize_f32) instead of the
reference's ~25 ATen passes (utils/sfp_quant.py:32-47, :80-96) and are bit-identical to
it; backward is the reference's straight-through estimator (:50-53, :99-102).  They
require ROCm ("cuda") float32 tensors: there is no CPU compute path in this package (the
CPU checker is oracle/, test-only).  k == 32 is the identity (:11-12, :60-61).

quantize_layerout (SFP<4,4> output quantizer, :105-133) runs on the device as well
(slfp_quantize_layerout_f32, bit-identical including the reference's XOR quirks) and can be
fused into the conv epilogue by fusion.fuse_bn_relu; like every quantized op it needs a ROCm tensor.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

__all__ = ["torch", "nn", "F", "np", "quantize_weight", "quantize_act", "quantize_layerout",
           "weight_quantize_func", "act_quantize_func", "layerout_quantize_func"]


def _stream_handle(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _require_gpu_f32(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: the SLFP HIP path needs a ROCm ('cuda') tensor, got device {t.device}; "
                           "there is no CPU compute path in this package")
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: expected float32, got {t.dtype}")


def hip_quantize(x, scale_div, fmt):
    """y = Q_fmt(x / float32(scale_div)) via slfp_quantize_f32; same memory layout as x."""
    _require_gpu_f32(x, "slfp quantize")
    L = _lib.load()
    dense = x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))
    src = x if dense else x.contiguous()
    y = torch.empty_like(src)  # preserves strides of a dense tensor
    with torch.cuda.device(x.device):
        _lib.check(L.slfp_quantize_f32(src.data_ptr(), y.data_ptr(), src.numel(), float(np.float32(scale_div)), fmt,
                                       _stream_handle(x)))
    return y


def hip_encode(x, scale_div, fmt):
    """uint8 extended codes of Q_fmt(x / float32(scale_div)) (slfp_encode_f32 with SLFP_FMT_EXT), same layout as x:
    decode(code) == quantize(x) bit for bit.  The 1-byte inter-layer format of fusion.link_codes."""
    _require_gpu_f32(x, "slfp encode")
    L = _lib.load()
    dense = x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))
    src = x if dense else x.contiguous()
    c = torch.empty_like(src, dtype=torch.uint8)
    with torch.cuda.device(x.device):
        _lib.check(L.slfp_encode_f32(src.data_ptr(), c.data_ptr(), src.numel(), float(np.float32(scale_div)), fmt | _lib.FMT_EXT,
                                     _stream_handle(x)))
    return c


def hip_decode(codes, fmt):
    """float32 values of extended codes (slfp_decode_f32 with SLFP_FMT_EXT), same layout as `codes`."""
    if not codes.is_cuda or codes.dtype != torch.uint8:
        raise TypeError("slfp decode: expected a ROCm ('cuda') uint8 tensor of codes")
    L = _lib.load()
    dense = codes.is_contiguous() or (codes.dim() == 4 and codes.is_contiguous(memory_format=torch.channels_last))
    src = codes if dense else codes.contiguous()
    y = torch.empty_like(src, dtype=torch.float32)
    with torch.cuda.device(codes.device):
        _lib.check(L.slfp_decode_f32(src.data_ptr(), y.data_ptr(), src.numel(), fmt | _lib.FMT_EXT, _stream_handle(codes)))
    return y


def hip_maxpool_codes(codes, kernel_size, stride, padding, q_bit):
    """nn.MaxPool2d (floor mode, dilation 1) on a channels_last tensor of extended activation codes (slfp_maxpool2d_codes):
    equal to encoding the pooled float32 tensor, bit for bit."""
    if not codes.is_cuda or codes.dtype != torch.uint8 or codes.dim() != 4:
        raise TypeError("slfp maxpool: expected a 4-d ROCm ('cuda') uint8 tensor of codes")
    if not codes.is_contiguous(memory_format=torch.channels_last):
        codes = codes.contiguous(memory_format=torch.channels_last)
    kh, kw = (kernel_size, kernel_size) if isinstance(kernel_size, int) else kernel_size
    stride = kernel_size if stride is None else stride
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    n, c, h, w = codes.shape
    ho, wo = (h + 2 * ph - kh) // sh + 1, (w + 2 * pw - kw) // sw + 1
    y = torch.empty((n, c, ho, wo), dtype=torch.uint8, device=codes.device, memory_format=torch.channels_last)
    L = _lib.load()
    with torch.cuda.device(codes.device):
        _lib.check(L.slfp_maxpool2d_codes(codes.data_ptr(), y.data_ptr(), n, h, w, c, kh, kw, sh, sw, ph, pw, int(q_bit), _stream_handle(codes)))
    return y


def _make_qfn(fmt):

    class qfn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, input):
            return hip_quantize(input, 1.0, fmt)

        @staticmethod
        def backward(ctx, grad_output):  # STE (utils/sfp_quant.py:50-53, :99-102)
            return grad_output.clone()
    return qfn.apply


def _check_k(k):
    if k not in (32, 8, 7):
        # the reference's forward leaves `out` unbound for other k (UnboundLocalError)
        raise ValueError(f"q_bit must be 32, 8 or 7, got {k}")


def quantize_weight(k):
    """utils/sfp_quant.py:7-54."""
    _check_k(k)
    if k == 32:
        return lambda x: x
    return _make_qfn(_lib.FMT_W8 if k == 8 else _lib.FMT_SFP7)


def quantize_act(k):
    """utils/sfp_quant.py:56-103."""
    _check_k(k)
    if k == 32:
        return lambda x: x
    return _make_qfn(_lib.FMT_ACT8 if k == 8 else _lib.FMT_SFP7)


class _LayeroutHip(torch.autograd.Function):
    """SFP<4,4> layer-output quantizer (utils/sfp_quant.py:112-126) as ONE HIP pass
    (slfp_quantize_layerout_f32), bit-identical to the reference including its quirks (tests/golden: all
    denormals + sampled binades): the reference writes `2^(-8)` / `2^(-7)`, which Python parses as integer
    XOR (= -6 / -5), so its two "subnormal" overrides never fire, only the `>= 248 -> 248` clamp is live and
    exact zeros come out as NaN (0 * inf).  STE backward (:129-132)."""

    @staticmethod
    def forward(ctx, x):
        L = _lib.load()
        dense = x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))
        src = x if dense else x.contiguous()
        y = torch.empty_like(src)
        with torch.cuda.device(x.device):
            _lib.check(L.slfp_quantize_layerout_f32(src.data_ptr(), y.data_ptr(), src.numel(), _stream_handle(x)))
        return y

    @staticmethod
    def backward(ctx, g):
        return g.clone()


def _layerout_dispatch(x):
    _require_gpu_f32(x, "quantize_layerout")   # no CPU compute path: the CPU restatement lives in oracle/ (tests only)
    return _LayeroutHip.apply(x)


def quantize_layerout(k):
    """utils/sfp_quant.py:105-133 (k <= 8 -> SFP<4,4>, 32 -> identity)."""
    if k == 32:
        return lambda x: x
    if k > 8:
        raise ValueError(f"q_bit must be <= 8 or 32, got {k}")
    return _layerout_dispatch


def absmax(x):
    """max |x| as a 0-dim tensor on x's device (HIP wave-shuffle reduction): the statistic the
    reference's calibration pass takes per layer (cifar100_train_eval.py:261-271)."""
    _require_gpu_f32(x, "slfp absmax")
    L = _lib.load()
    src = x if x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)) else x.contiguous()
    out = torch.empty((), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.slfp_absmax_f32(src.data_ptr(), src.numel(), out.data_ptr(), _stream_handle(x)))
    return out


class _QuantModule(nn.Module):
    _factory = None

    def __init__(self, q_bit):
        super().__init__()
        assert q_bit <= 8 or q_bit == 32  # utils/sfp_quant.py:138,152,166
        self.q_bit = q_bit
        self.quantize = type(self)._factory(q_bit) if q_bit in (32, 8, 7) else None

    def forward(self, x):
        if self.q_bit == 32:
            return x
        if self.q_bit in (8, 7):
            return self.quantize(x)
        # the reference falls off the end of its if/elif here (utils/sfp_quant.py:142-147)
        raise UnboundLocalError("q_bit must be 32, 8 or 7 for the SLFP/SFP quantizers")


class weight_quantize_func(_QuantModule):
    """utils/sfp_quant.py:135-147."""
    _factory = staticmethod(quantize_weight)


class act_quantize_func(_QuantModule):
    """utils/sfp_quant.py:149-161."""
    _factory = staticmethod(quantize_act)


class layerout_quantize_func(nn.Module):
    """utils/sfp_quant.py:163-175."""

    def __init__(self, q_bit):
        super().__init__()
        assert q_bit <= 8 or q_bit == 32
        self.q_bit = q_bit
        self.quantize = quantize_layerout(k=q_bit)

    def forward(self, x):
        if self.q_bit == 32:
            return x
        if self.q_bit in (8, 7):
            return self.quantize(x)
        raise UnboundLocalError("q_bit must be 32, 8 or 7")
