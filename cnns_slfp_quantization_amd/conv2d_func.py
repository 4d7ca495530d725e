"""Host-side mirror of the reference's utils/conv2d_func.py on top of the HIP C ABI.

Drop-in operator API (same names, positional order, attributes and state-dict keys):

    conv2d_Q(q_bit, Kw, Ka)       -> class Conv2d_Q(nn.Conv2d)   utils/conv2d_func.py:8-26
    conv2d_Q_bias(q_bit, Kw, Ka)  -> class Conv2d_Q(nn.Conv2d)   utils/conv2d_func.py:28-48
    linear_Q(q_bit, Kw, Ka)       -> class Linear_Q(nn.Linear)   utils/conv2d_func.py:50-66

Where the reference runs  x/Ka -> ~25-pass quantize_act -> w/Kw -> ~25-pass
quantize_weight (every forward) -> fp32 F.conv2d -> *Ka*Kw  (conv2d_func.py:20-25), this
module makes ONE call into libslfp_hip.so (slfp_conv2d_fwd): the SLFP encode is applied
inline on the kernels' load path; `input_q` / `weight_q` -- which the CIFAR nets read back
after every forward (nets_cifar/mobilenetv1.py:88-171) -- are materialised lazily on first
access.

Weights: the reference re-quantizes them on EVERY forward (utils/conv2d_func.py:22).  So does
this module whenever the weights can change under it: in training mode, and whenever autograd
is recording -- the reference's own optimizers update `p.data` in place (utils/optimizer.py:
58-63), which does not bump `weight._version`, so no version key can see it.  Only in
inference (`module.eval()` and `torch.no_grad()`/`inference_mode`) is the kernel-specific
weight blob cached, keyed on storage, `_version`, shape, scales and kernel; every
`module.train(...)` / `.eval()` call and `invalidate()` drop it.

Memory layout: the kernels are NHWC.  A `torch.channels_last` input is consumed and
produced in place (zero copies; BN/ReLU keep the format); an NCHW-contiguous input is
transposed inside the C ABI and, by default, the output comes back NCHW so that code
which `.view()`s NCHW strides (nets_cifar/shufflenet_v2.py:41) keeps working.  Use
`model.to(memory_format=torch.channels_last)` + a channels_last input for the fast path.

There is no CPU compute path here (q_bit 8/7 need a ROCm tensor; q_bit 32 is the
reference's passthrough).  Autograd: forward is always the HIP kernel; backward is the
reference's STE composite on the GPU (training is outside the accelerated scope).
"""
import contextlib
import ctypes

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .sfp_quant import *  # noqa: F401,F403  (the reference re-exports these: conv2d_func.py:5)
from .sfp_quant import (_require_gpu_f32, _stream_handle, hip_quantize, hip_encode, hip_decode, weight_quantize_func,
                        act_quantize_func)

__all__ = ["torch", "nn", "F", "np", "conv2d_Q", "conv2d_Q_bias", "linear_Q", "options",
           "quantize_weight", "quantize_act", "quantize_layerout",
           "weight_quantize_func", "act_quantize_func", "layerout_quantize_func"]


class _Options:
    """Process-wide knobs of the HIP path (not part of the reference API)."""
    mfma_passes = _lib.MFMA_DEFAULT  # pointwise MFMA operand precision, see include/slfp.h
    output_layout = "same"           # "same": follow the input's memory format; "nhwc": always channels_last
    eager_stash = False              # True: materialise input_q / weight_q on every forward like the reference
    plan_cache = True                # False: rebuild descriptor / shapes / workspace on every call (host-overhead A/B)
    dwpw_all = False                 # True: fusion.DwPwBlock uses the one-kernel form wherever the library supports it,
                                     # not only where it measured faster than two kernels
    dwpw_pairs = {(32, 1)}           # (depthwise channels, stride) pairs that run as one kernel by default


options = _Options()


def _f32(v):
    """float32(K) as a Python float: the cast ATen applies to the 0-dim float64 scale."""
    return float(np.float32(float(v)))


def _scalar_scale(t, name):
    if t.numel() != 1:
        raise ValueError(f"{name} must be a scalar calibration scale, got a tensor of shape {tuple(t.shape)} "
                         "(pass the per-layer value, as the reference nets do)")
    v = float(t)
    if not v > 0:
        raise ValueError(f"{name} must be > 0, got {v}")
    return v


def _scale_key(t, name):
    """The scale as a Python float for the plan key (validated like _scalar_scale, without its tensor round trip)."""
    if torch.is_tensor(t):
        if t.numel() != 1:
            _scalar_scale(t, name)
        return t.item()
    return float(t)


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


class _PreparedWeights:
    """Per-module cache of the kernel-specific weight blob (and the OIHW weight_q tensor)."""

    def __init__(self):
        self.key = None
        self.blob = None
        self.weight_q = None

    def invalidate(self):
        self.key = None
        self.blob = None
        self.weight_q = None

    def get(self, L, desc, weight, want_weight_q, cache=True, kernel=None):
        key = (weight.device, weight.data_ptr(), weight._version, tuple(weight.shape), desc.qbits,
               desc.kw_scale, kernel if kernel is not None else L.slfp_conv2d_kernel_name(ctypes.byref(desc)))
        if not cache or key != self.key or self.blob is None or (want_weight_q and self.weight_q is None):
            nbytes = L.slfp_conv2d_wprep_bytes(ctypes.byref(desc))
            w = weight.detach()
            w = w if w.is_contiguous() else w.contiguous()  # OIHW
            blob = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
            wq = torch.empty_like(w) if want_weight_q else None
            _lib.check(L.slfp_conv2d_prepare_weights(ctypes.byref(desc), w.data_ptr(), blob.data_ptr(),
                                                     wq.data_ptr() if wq is not None else None,
                                                     _stream_handle(weight)))
            self.key, self.blob, self.weight_q = key, blob, wq
        return self.blob


class _Plan:
    """What one (module, input shape, layout, scales, precision) combination resolves to in the C ABI."""
    __slots__ = ("desc", "ho", "wo", "ws_bytes", "kernel", "io", "codes_ok")

    def __init__(self, desc, ho, wo, ws_bytes, kernel):
        self.desc, self.ho, self.wo, self.ws_bytes, self.kernel = desc, ho, wo, ws_bytes, kernel


_workspaces = {}
_same_device = contextlib.nullcontext()


def _on_device(device):
    """torch.cuda.device(device) only when it is not the current one already (the guard costs ~5 us per call)."""
    return _same_device if device.index == torch.cuda.current_device() else torch.cuda.device(device)


def _workspace(device, nbytes):
    """Scratch for the kernels that need one (the dense path's pre-encoded input), reused across calls: one buffer per
    (device, stream), grown to the largest request.  Calls on one stream are ordered, so consecutive layers can share
    it; a different stream gets its own.  Under hipGraph capture a fresh tensor is taken from the graph's pool instead
    (the captured pointer must stay valid for the graph's lifetime)."""
    if torch.cuda.is_current_stream_capturing() or not options.plan_cache:
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def _hip_conv2d(mod, x, weight, bias, cache_ok=False):
    """One slfp_conv2d_fwd call for module `mod` (an nn.Conv2d subclass below).  cache_ok: the prepared weights may
    come from the module's cache (decided by the caller: grad mode is off inside autograd.Function.forward)."""
    _require_gpu_f32(x, "Conv2d_Q")
    if weight.device != x.device:
        raise RuntimeError(f"Conv2d_Q: input is on {x.device} but weight is on {weight.device}")
    squeeze = x.dim() == 3
    if squeeze:
        x = x.unsqueeze(0)
    if x.dim() != 4:
        raise RuntimeError(f"Expected 3D (unbatched) or 4D (batched) input to conv2d, but got input of size: {list(x.shape)}")
    if isinstance(mod.padding, str) or mod.padding_mode != "zeros":
        raise NotImplementedError("Conv2d_Q (HIP): only explicit zero padding is supported")
    L = _lib.load()
    nhwc_in = x.is_contiguous(memory_format=torch.channels_last)
    if not nhwc_in and not x.is_contiguous():
        x = x.contiguous()
    N, C, H, W = x.shape
    if C != mod.in_channels:
        raise RuntimeError(f"Given groups={mod.groups}, weight of size {list(weight.shape)}, expected input"
                           f"{list(x.shape)} to have {mod.in_channels} channels, but got {C} channels instead")
    nhwc_out = nhwc_in or options.output_layout == "nhwc"
    # Everything that depends only on (module geometry, input shape, layouts, scales, precision) is computed once
    # per distinct key and kept on the module: descriptor, output shape, workspace size, kernel name.
    ka, kw_ = _scale_key(mod.Ka, "Ka"), _scale_key(mod.Kw, "Kw")
    pkey = (N, H, W, nhwc_in, nhwc_out, ka, kw_, options.mfma_passes, mod.stride, mod.padding, mod.dilation, tuple(weight.shape))
    plan = mod._plans.get(pkey) if options.plan_cache else None
    if plan is None:
        sh, sw = _pair(mod.stride)
        ph, pw = _pair(mod.padding)
        dh, dw = _pair(mod.dilation)
        d = _lib.ConvDesc(n=N, c_in=C, h=H, w=W, c_out=mod.out_channels, kh=weight.shape[2], kw=weight.shape[3],
                          stride_h=sh, stride_w=sw, pad_h=ph, pad_w=pw, dil_h=dh, dil_w=dw, groups=mod.groups,
                          x_layout=_lib.LAYOUT_NHWC if nhwc_in else _lib.LAYOUT_NCHW,
                          y_layout=_lib.LAYOUT_NHWC if nhwc_out else _lib.LAYOUT_NCHW,
                          qbits=mod.q_bit, ka=_f32(_scalar_scale(mod.Ka, "Ka")), kw_scale=_f32(_scalar_scale(mod.Kw, "Kw")),
                          mfma_passes=options.mfma_passes, reserved=0)
        ho, wo = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(L.slfp_conv2d_out_shape(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)))
        plan = _Plan(d, ho.value, wo.value, L.slfp_conv2d_workspace_bytes(ctypes.byref(d)),
                     L.slfp_conv2d_kernel_name(ctypes.byref(d)).decode())
        if len(mod._plans) >= 64:   # a net fed ever-changing shapes: do not grow without bound
            mod._plans.clear()
        mod._plans[pkey] = plan
    d = plan.desc
    with _on_device(x.device):
        # cache the prepared weights only where they cannot change unseen: inference (see the module docstring)
        blob = mod._prep.get(L, d, weight, want_weight_q=options.eager_stash, cache=cache_ok, kernel=plan.kernel)
        y = torch.empty((N, mod.out_channels, plan.ho, plan.wo), dtype=torch.float32, device=x.device,
                        memory_format=torch.channels_last if nhwc_out else torch.contiguous_format)
        ws = _workspace(x.device, plan.ws_bytes) if plan.ws_bytes else None
        b = None
        if bias is not None:
            b = bias.detach()
            b = b if b.is_contiguous() else b.contiguous()
        xq = torch.empty_like(x) if options.eager_stash else None
        post = getattr(mod, "_post", None)  # (scale, shift, flags) set by fusion.fuse_bn_relu: eval-BN (+ layerout) + ReLU in the epilogue
        ps = psh = None
        relu = 0
        if post is not None:
            ps, psh, relu = post
            if ps is not None and ps.device != x.device:
                ps, psh = ps.to(x.device), psh.to(x.device)
                mod._post = (ps, psh, relu)
        _lib.check(L.slfp_conv2d_fwd_post(ctypes.byref(d), x.data_ptr(), blob.data_ptr(),
                                          b.data_ptr() if b is not None else None,
                                          ps.data_ptr() if ps is not None else None,
                                          psh.data_ptr() if psh is not None else None, int(relu), y.data_ptr(),
                                          xq.data_ptr() if xq is not None else None,
                                          ws.data_ptr() if ws is not None else None, _stream_handle(x)))
    mod._last_kernel = plan.kernel
    mod._last_input = x.detach()
    mod._input_q = xq
    return y.squeeze(0) if squeeze else y


def _act_fmt(q_bit):
    return _lib.FMT_ACT8 if q_bit == 8 else _lib.FMT_SFP7


def _hip_conv2d_codes(mod, x, weight, bias):
    """Conv2d_Q.forward inside a chain linked by fusion.link_codes: `x` is float32 or the uint8 codes the previous layer
    wrote for THIS module's Ka / q_bit; the result is uint8 codes for the next layer (mod._code_out = (Ka_next, q_bit_next))
    or float32.  One slfp_conv2d_fwd_codes call where libslfp_hip has a kernel for the combination; otherwise the same
    values through the float32 interface plus slfp_encode_f32 / slfp_decode_f32 (always correct, never faster)."""
    x_codes = x.dtype == torch.uint8
    out = mod._code_out
    if not x_codes:
        _require_gpu_f32(x, "Conv2d_Q")
    elif not x.is_cuda:
        raise RuntimeError("Conv2d_Q: code tensors live on the ROCm device")
    if x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last):
        # the code path is NHWC only; anything else takes the float32 interface
        x32 = hip_decode(x, _act_fmt(mod.q_bit)) if x_codes else x
        y = _hip_conv2d(mod, x32, weight, bias, cache_ok=True)
        return hip_encode(y, out[0], _act_fmt(out[1])) if out is not None else y
    L = _lib.load()
    N, C, H, W = x.shape
    if C != mod.in_channels:
        raise RuntimeError(f"Given groups={mod.groups}, weight of size {list(weight.shape)}, expected input"
                           f"{list(x.shape)} to have {mod.in_channels} channels, but got {C} channels instead")
    ka, kw_ = _scale_key(mod.Ka, "Ka"), _scale_key(mod.Kw, "Kw")
    pkey = ("codes", N, H, W, x_codes, out, ka, kw_, options.mfma_passes, mod.stride, mod.padding, mod.dilation, tuple(weight.shape))
    plan = mod._plans.get(pkey)
    if plan is None:
        sh, sw = _pair(mod.stride)
        ph, pw = _pair(mod.padding)
        dh, dw = _pair(mod.dilation)
        d = _lib.ConvDesc(n=N, c_in=C, h=H, w=W, c_out=mod.out_channels, kh=weight.shape[2], kw=weight.shape[3],
                          stride_h=sh, stride_w=sw, pad_h=ph, pad_w=pw, dil_h=dh, dil_w=dw, groups=mod.groups,
                          x_layout=_lib.LAYOUT_NHWC, y_layout=_lib.LAYOUT_NHWC, qbits=mod.q_bit,
                          ka=_f32(_scalar_scale(mod.Ka, "Ka")), kw_scale=_f32(_scalar_scale(mod.Kw, "Kw")),
                          mfma_passes=options.mfma_passes, reserved=0)
        ho, wo = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(L.slfp_conv2d_out_shape(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)))
        io = _lib.ConvIo(x_codes=1 if x_codes else 0, y_codes=1 if out is not None else 0,
                         y_ka=_f32(out[0]) if out is not None else 1.0, y_qbits=int(out[1]) if out is not None else 8)
        post = mod._post
        flags = int(post[2]) if post is not None else 0
        ok = bool(L.slfp_conv2d_codes_supported(ctypes.byref(d), ctypes.byref(io), 1 if bias is not None else 0, flags))
        plan = _Plan(d, ho.value, wo.value, L.slfp_conv2d_workspace_bytes(ctypes.byref(d)) if ok else 0,
                     L.slfp_conv2d_kernel_name(ctypes.byref(d)).decode())
        plan.io, plan.codes_ok = io, ok
        if len(mod._plans) >= 64:
            mod._plans.clear()
        mod._plans[pkey] = plan
    if not plan.codes_ok:
        x32 = hip_decode(x, _act_fmt(mod.q_bit)) if x_codes else x
        y = _hip_conv2d(mod, x32, weight, bias, cache_ok=True)
        return hip_encode(y, out[0], _act_fmt(out[1])) if out is not None else y
    d = plan.desc
    with _on_device(x.device):
        blob = mod._prep.get(L, d, weight, want_weight_q=False, cache=True, kernel=plan.kernel)
        y = torch.empty((N, mod.out_channels, plan.ho, plan.wo), dtype=torch.uint8 if out is not None else torch.float32,
                        device=x.device, memory_format=torch.channels_last)
        b = None
        if bias is not None:
            b = bias.detach()
            b = b if b.is_contiguous() else b.contiguous()
        ps = psh = None
        relu = 0
        if mod._post is not None:
            ps, psh, relu = mod._post
            if ps is not None and ps.device != x.device:
                ps, psh = ps.to(x.device), psh.to(x.device)
                mod._post = (ps, psh, relu)
        ws = _workspace(x.device, plan.ws_bytes) if plan.ws_bytes else None   # dense k x k layers: the fp16 operand copy
        _lib.check(L.slfp_conv2d_fwd_codes_ws(ctypes.byref(d), ctypes.byref(plan.io), x.data_ptr(), blob.data_ptr(),
                                              b.data_ptr() if b is not None else None,
                                              ps.data_ptr() if ps is not None else None,
                                              psh.data_ptr() if psh is not None else None, int(relu), y.data_ptr(),
                                              ws.data_ptr() if ws is not None else None, _stream_handle(x)))
    mod._last_kernel = plan.kernel + ("+codes_in" if x_codes else "") + ("+codes_out" if out is not None else "")
    if x_codes:
        mod._last_input, mod._last_codes = None, x.detach()
    else:
        mod._last_input, mod._last_codes = x.detach(), None
    mod._input_q = None
    return y


class _SlfpConv2dFn(torch.autograd.Function):
    """HIP forward; backward = the reference's composite (STE through both quantizers:
    utils/sfp_quant.py:50-53, :99-102; conv gradients from torch.nn.grad on the GPU)."""

    @staticmethod
    def forward(ctx, x, weight, bias, mod, scaled_bias):
        ctx.mod = mod
        ctx.scaled_bias = scaled_bias
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return _hip_conv2d(mod, x, weight, bias if scaled_bias else None)

    @staticmethod
    def backward(ctx, gy):
        mod = ctx.mod
        x, weight = ctx.saved_tensors
        ka, kw = _f32(mod.Ka), _f32(mod.Kw)
        fa = _lib.FMT_ACT8 if mod.q_bit == 8 else _lib.FMT_SFP7
        fw = _lib.FMT_W8 if mod.q_bit == 8 else _lib.FMT_SFP7
        g = (gy * kw * ka).contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wq = hip_quantize(weight.detach().contiguous(), kw, fw)
            gx = torch.nn.grad.conv2d_input(x.shape, wq, g, mod.stride, mod.padding, mod.dilation, mod.groups) / ka
        if ctx.needs_input_grad[1]:
            xq = hip_quantize(x.detach().contiguous(), ka, fa)
            gw = torch.nn.grad.conv2d_weight(xq, weight.shape, g, mod.stride, mod.padding, mod.dilation, mod.groups) / kw
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(dim=(0, 2, 3)) if ctx.scaled_bias else g.sum(dim=(0, 2, 3))
        return gx, gw, gb, None, None


def _apply_post_composite(out, post):
    """The fused epilogue written with stock ATen ops (q_bit == 32 passthrough only)."""
    if post is None:
        return out
    scale, shift, flags = post   # flags: 1 = ReLU (the layer-output quantizer is never fused onto a q_bit 32 conv)
    if scale is not None:
        out = out * scale.to(out.device).view(1, -1, 1, 1) + shift.to(out.device).view(1, -1, 1, 1)
    return torch.relu(out) if (int(flags) & 1) else out


def _conv_class(q_bit, Kw, Ka, bias_default, scaled_bias):
    class Conv2d_Q(nn.Conv2d):
        # Kw, Ka are positional arguments 4 and 5 (utils/conv2d_func.py:10-11, :30)
        def __init__(self, in_channels, out_channels, kernel_size, Kw=Kw, Ka=Ka,
                     stride=1, padding=0, dilation=1, groups=1, bias=bias_default):
            super(Conv2d_Q, self).__init__(in_channels, out_channels, kernel_size, stride,
                                           padding, dilation, groups, bias)
            self.q_bit = q_bit
            self.quantize_weight = weight_quantize_func(q_bit=q_bit)
            self.quantize_act = act_quantize_func(q_bit=q_bit)
            # plain attributes, 0-dim float64, stay on the CPU (utils/conv2d_func.py:17-18)
            self.Kw = torch.tensor(Kw)
            self.Ka = torch.tensor(Ka)
            self._prep = _PreparedWeights()
            self._plans = {}
            self._last_input = None
            self._input_q = None
            self._weight_q32 = None
            self._last_kernel = None
            self._post = None  # (scale, shift, relu): fused eval-BN + ReLU epilogue (fusion.fuse_bn_relu)
            self._code_out = None   # (Ka, q_bit) of the next Conv2d_Q: hand it 1-byte codes (fusion.link_codes)
            self._last_codes = None
            self._scaled_bias = scaled_bias
            self.output = None

        # -- the reference stores these on every forward (utils/conv2d_func.py:21-22);
        #    here they are computed on first access after a forward.
        @property
        def input_q(self):
            if self.q_bit == 32:
                return self._input_q
            if self._input_q is None and self._last_input is not None:
                fmt = _lib.FMT_ACT8 if self.q_bit == 8 else _lib.FMT_SFP7
                self._input_q = hip_quantize(self._last_input, _f32(self.Ka), fmt)
            elif self._input_q is None and self._last_codes is not None:
                # inside a code chain the input arrived already quantized: input_q = decode(codes), bit for bit
                self._input_q = hip_decode(self._last_codes, _act_fmt(self.q_bit))
            return self._input_q

        @property
        def weight_q(self):
            if self.q_bit == 32:
                return self._weight_q32
            if self._last_input is None and self._last_codes is None:
                return None
            stale = self._prep.weight_q is None or self._prep.key is None or self._prep.key[2] != self.weight._version
            if stale or self.training or torch.is_grad_enabled():   # p.data updates are invisible to _version
                fmt = _lib.FMT_W8 if self.q_bit == 8 else _lib.FMT_SFP7
                self._prep.weight_q = hip_quantize(self.weight.detach().contiguous(), _f32(self.Kw), fmt)
            return self._prep.weight_q

        def invalidate(self):
            """Drop the cached quantized weights (call after changing `weight.data` in place during inference)."""
            self._prep.invalidate()

        def train(self, mode=True):
            self._prep.invalidate()   # weights may have been stepped through `.data` since the last forward
            return super(Conv2d_Q, self).train(mode)

        def forward(self, input, order=None):
            if self.q_bit == 32:
                # identity quantizers (utils/sfp_quant.py:11-12, :60-61): stock ATen, any device
                self._input_q = input / self.Ka
                self._weight_q32 = self.weight / self.Kw
                b = self.bias
                if b is not None and scaled_bias:
                    b = b / self.Ka / self.Kw
                self.output = F.conv2d(self._input_q, self._weight_q32, b, self.stride, self.padding,
                                       self.dilation, self.groups) * self.Ka * self.Kw
                if self.bias is not None and not scaled_bias:
                    pass  # conv2d_Q hands the raw bias to F.conv2d above (utils/conv2d_func.py:23)
                self.output = _apply_post_composite(self.output, self._post)
                return self.output
            if self.q_bit not in (8, 7):
                raise UnboundLocalError("q_bit must be 32, 8 or 7 (utils/sfp_quant.py:142-147)")
            if input.dtype == torch.uint8 or self._code_out is not None:
                # a link of fusion.link_codes: 1-byte codes on one or both sides (inference only)
                if self.training or self._scaled_bias is False and self.bias is not None:
                    raise RuntimeError("Conv2d_Q: code links (fusion.link_codes) are inference-only and need the scaled bias class; "
                                       "call fusion.unlink_codes(model)")
                with torch.no_grad():
                    self.output = _hip_conv2d_codes(self, input, self.weight, self.bias)
                return self.output
            need_grad = torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad or
                                                     (self.bias is not None and self.bias.requires_grad))
            if self._post is not None and (need_grad and self.training):
                raise RuntimeError("Conv2d_Q: a fused BN/ReLU epilogue is inference-only; call fusion.unfuse(model) to train")
            if self._post is not None and self.bias is not None and not scaled_bias:
                raise NotImplementedError("fused epilogue with conv2d_Q's raw (unscaled) bias is not supported")
            if need_grad and self._post is None:
                # conv2d_Q's raw bias is added below, outside the Function: only the scaled one goes through it
                out = _SlfpConv2dFn.apply(input, self.weight, self.bias if scaled_bias else None, self, scaled_bias)
            else:
                out = _hip_conv2d(self, input, self.weight, self.bias if scaled_bias else None,
                                  cache_ok=not self.training and not torch.is_grad_enabled())
            if self.bias is not None and not scaled_bias:
                # conv2d_Q hands the raw bias to F.conv2d (utils/conv2d_func.py:23): (conv + b)*Ka*Kw
                out = out + (self.bias * self.Ka * self.Kw).to(out.dtype).view(1, -1, 1, 1)
            self.output = out
            return out

    return Conv2d_Q


def conv2d_Q(q_bit, Kw, Ka):
    """utils/conv2d_func.py:8-26: bias defaults to False and is NOT rescaled."""
    return _conv_class(q_bit, Kw, Ka, bias_default=False, scaled_bias=False)


def conv2d_Q_bias(q_bit, Kw, Ka):
    """utils/conv2d_func.py:28-48: bias defaults to True, bias_q = bias / Ka / Kw."""
    return _conv_class(q_bit, Kw, Ka, bias_default=True, scaled_bias=True)


def _hip_linear(mod, x, weight, bias, cache_ok=False):
    _require_gpu_f32(x, "Linear_Q")
    L = _lib.load()
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    B, I = x2.shape
    O = weight.shape[0]
    if I != weight.shape[1]:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({B}x{I} and {weight.shape[1]}x{O})")
    w = weight.detach()
    w = w if w.is_contiguous() else w.contiguous()
    b = None
    if bias is not None:
        b = bias.detach()
        b = b if b.is_contiguous() else b.contiguous()
    y = torch.empty((B, O), dtype=torch.float32, device=x.device)
    ka, kw = _f32(_scalar_scale(mod.Ka, "Ka")), _f32(_scalar_scale(mod.Kw, "Kw"))
    with torch.cuda.device(x.device):
        # quantize the weights once per weight version (the reference re-quantizes on every forward,
        # utils/conv2d_func.py:62: AlexNet's 9216x4096 / VGG-16's 25088x4096 layers make that the
        # dominant cost of their classifiers)
        key = (w.device, w.data_ptr(), weight._version, tuple(w.shape), mod.q_bit, kw, options.mfma_passes)
        cache = mod.__dict__.get("_lin_prep")
        if cache is None or cache[0] != key or not cache_ok:   # see the module docstring
            blob = torch.empty(max(L.slfp_linear_workspace_bytes(1, I, O), 16), dtype=torch.uint8, device=x.device)
            _lib.check(L.slfp_linear_prepare_weights(w.data_ptr(), blob.data_ptr(), I, O, kw, mod.q_bit,
                                                     options.mfma_passes, _stream_handle(x)))
            cache = (key, blob)
            mod.__dict__["_lin_prep"] = cache
        _lib.check(L.slfp_linear_fwd_prepared(x2.data_ptr(), cache[1].data_ptr(), b.data_ptr() if b is not None else None,
                                              y.data_ptr(), B, I, O, ka, kw, mod.q_bit, options.mfma_passes,
                                              _stream_handle(x)))
    return y.reshape(*lead, O)


class _SlfpLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, mod):
        ctx.mod = mod
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return _hip_linear(mod, x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        mod = ctx.mod
        x, weight = ctx.saved_tensors
        ka, kw = _f32(mod.Ka), _f32(mod.Kw)
        fa = _lib.FMT_ACT8 if mod.q_bit == 8 else _lib.FMT_SFP7
        fw = _lib.FMT_W8 if mod.q_bit == 8 else _lib.FMT_SFP7
        g = gy * kw * ka
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = (g @ hip_quantize(weight.detach().contiguous(), kw, fw)) / ka
        if ctx.needs_input_grad[1]:
            xq = hip_quantize(x.detach().contiguous(), ka, fa)
            gw = (g.reshape(-1, g.shape[-1]).t() @ xq.reshape(-1, xq.shape[-1])) / kw
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.reshape(-1, gy.shape[-1]).sum(0)
        return gx, gw, gb, None


def linear_Q(q_bit, Kw, Ka):
    """utils/conv2d_func.py:50-66 (next-row component: same C ABI, pointwise kernel family)."""
    class Linear_Q(nn.Linear):
        def __init__(self, in_features, out_features, Kw=Kw, Ka=Ka, bias=True):
            super(Linear_Q, self).__init__(in_features, out_features, bias)
            self.q_bit = q_bit
            self.quantize_weight = weight_quantize_func(q_bit=q_bit)
            self.quantize_act = act_quantize_func(q_bit=q_bit)
            self.Kw = torch.tensor(Kw)
            self.Ka = torch.tensor(Ka)
            self._last_input = None
            self._input_q = None
            self._weight_q = None
            self.bias_q = None

        # The reference stores input_q / weight_q / bias_q on every forward (utils/conv2d_func.py:60-64) and its nets
        # read them back afterwards (nets_cifar/mobilenetv1.py:169-170, resnet50.py:353-354, alexnet.py:107-114);
        # here the two quantized tensors are computed on first access after a forward.
        @property
        def input_q(self):
            if self.q_bit != 32 and self._input_q is None and self._last_input is not None:
                fmt = _lib.FMT_ACT8 if self.q_bit == 8 else _lib.FMT_SFP7
                self._input_q = hip_quantize(self._last_input, _f32(self.Ka), fmt)
            return self._input_q

        @property
        def weight_q(self):
            if self.q_bit != 32 and self._weight_q is None and self._last_input is not None:
                fmt = _lib.FMT_W8 if self.q_bit == 8 else _lib.FMT_SFP7
                self._weight_q = hip_quantize(self.weight.detach().contiguous(), _f32(self.Kw), fmt)
            return self._weight_q

        def invalidate(self):
            self.__dict__.pop("_lin_prep", None)

        def train(self, mode=True):
            self.invalidate()
            return super(Linear_Q, self).train(mode)

        def forward(self, input):
            if self.q_bit == 32:
                self._input_q = input / self.Ka
                self._weight_q = self.weight / self.Kw
                # the reference dereferences self.bias unconditionally (utils/conv2d_func.py:63)
                self.bias_q = self.bias / self.Kw / self.Ka
                return F.linear(self._input_q, self._weight_q, self.bias_q) * self.Kw * self.Ka
            if self.q_bit not in (8, 7):
                raise UnboundLocalError("q_bit must be 32, 8 or 7 (utils/sfp_quant.py:142-147)")
            if self.bias is None:
                raise TypeError("unsupported operand type(s) for /: 'NoneType' and 'Tensor'")  # as the reference
            self._last_input = input.detach()
            self._input_q = None
            self._weight_q = None
            self.bias_q = self.bias / self.Kw / self.Ka      # utils/conv2d_func.py:63 (the kernel applies the same two divisions)
            need_grad = torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad or
                                                     self.bias.requires_grad)
            if need_grad:
                return _SlfpLinearFn.apply(input, self.weight, self.bias, self)
            return _hip_linear(self, input, self.weight, self.bias, cache_ok=not self.training and not torch.is_grad_enabled())

    return Linear_Q
