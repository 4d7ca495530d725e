"""Inference-time fusion of the eval-mode BatchNorm2d (+ ReLU) that follows every Conv2d_Q in the
reference nets into the conv kernels' epilogue.

The reference runs  Conv2d_Q -> nn.BatchNorm2d -> nn.ReLU(inplace)  as three modules
(nets_imgnet/mobilenetv1.py:24-41); stock BN + ReLU move 16 B per conv output element, twice the
hot path's own traffic.  `fuse_bn_relu(model)` rewrites every such run inside an nn.Sequential:
the BatchNorm2d's running statistics and affine are folded into a per-channel (scale, shift) that
the HIP epilogue applies after the reference's (out*Ka)*Kw roundings (slfp_conv2d_fwd_post), and
the BN / ReLU modules are replaced by nn.Identity.  Call it AFTER load_state_dict and model.eval().
Nets that wire conv->bn by hand in forward() (ResNet-50 blocks) use `fuse_named_bn(model, example_input)` (folds every
conv<k>/bn<k> pair by name and verifies itself against the example input) or `fuse_pair(conv, bn, relu)` per pair.

`fuse_dw_pw(model)` goes one step further for MobileNet blocks (second half): an adjacent
[depthwise Conv2d_Q + BN + ReLU] -> [pointwise Conv2d_Q (+ BN + ReLU)] pair that libslfp_hip can run as ONE kernel
(slfp_dwpw_fwd: the depthwise result is quantized for the pointwise layer where it is produced and never goes to HBM)
is replaced by a `DwPwBlock`; the result is bit-identical to the two fused convs run separately.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn


def _is_conv_q(m):
    return isinstance(m, nn.Conv2d) and hasattr(m, "q_bit") and hasattr(m, "Ka") and hasattr(m, "_post")


def fold_bn(bn):
    """(scale, shift) float32 with  bn(x) == x * scale + shift  in eval mode."""
    if bn.training or not bn.track_running_stats or bn.running_mean is None:
        raise RuntimeError("fuse_bn_relu: the BatchNorm2d must be in eval mode with running statistics")
    var = bn.running_var.detach().double()
    mean = bn.running_mean.detach().double()
    gamma = bn.weight.detach().double() if bn.affine else torch.ones_like(var)
    beta = bn.bias.detach().double() if bn.affine else torch.zeros_like(var)
    scale = gamma / torch.sqrt(var + bn.eps)
    shift = beta - mean * scale
    return scale.float().contiguous(), shift.float().contiguous()


def fuse_pair(conv, bn=None, relu=False, layerout=False):
    """Attach bn (eval-mode nn.BatchNorm2d or None), an optional SFP<4,4> layer-output quantizer
    (utils/sfp_quant.py:105-133; needs bn) and an optional ReLU to `conv`'s epilogue, in that order."""
    if not _is_conv_q(conv):
        raise TypeError("fuse_pair: conv must be a Conv2d_Q module of this package")
    if layerout and bn is None:
        raise ValueError("fuse_pair: the layer-output quantizer is fused only together with a BatchNorm2d")
    flags = (1 if relu else 0) | (2 if layerout else 0)   # SLFP_POST_RELU | SLFP_POST_LAYEROUT
    if bn is not None:
        if bn.num_features != conv.out_channels:
            raise ValueError("fuse_pair: BatchNorm2d width does not match the conv's out_channels")
        scale, shift = fold_bn(bn)
        dev = conv.weight.device
        conv._post = (scale.to(dev), shift.to(dev), flags)
    else:
        conv._post = (None, None, flags)
    return conv


def _is_layerout(m):
    """layerout_quantize_func(q_bit <= 8): the SFP<4,4> quantizer module of the `*_swish` / `*_gelu` / ShuffleNetV2 nets."""
    return type(m).__name__ == "layerout_quantize_func" and getattr(m, "q_bit", 32) in (8, 7)


def fuse_bn_relu(model, dw_pw=False):
    """Fuse every [Conv2d_Q, BatchNorm2d(eval), (layerout_quantize_func), (ReLU)] run found in nn.Sequential
    containers (nets_imgnet/mobilenetv1.py:24-41; nets_cifar/mobilenetv1.py:196-231 for the layerout form; a
    Swish / GELU after the quantizer stays a module).  Returns the number of fused convolutions.
    dw_pw=True additionally pairs depthwise and pointwise convs into one kernel where supported (fuse_dw_pw)."""
    n = 0
    for seq in [m for m in model.modules() if isinstance(m, nn.Sequential)]:
        names = list(seq._modules.keys())
        i = 0
        while i < len(names):
            conv = seq._modules[names[i]]
            if _is_conv_q(conv) and conv._post is None and i + 1 < len(names):
                bn = seq._modules[names[i + 1]]
                ok = (isinstance(bn, nn.BatchNorm2d) and not bn.training and bn.num_features == conv.out_channels
                      and (conv.bias is None or getattr(conv, "_scaled_bias", False)))
                if ok:
                    j = i + 2
                    lo = (j < len(names) and _is_layerout(seq._modules[names[j]]) and conv.q_bit in (8, 7))
                    j += 1 if lo else 0
                    relu = j < len(names) and isinstance(seq._modules[names[j]], nn.ReLU)
                    j += 1 if relu else 0
                    fuse_pair(conv, bn, relu, lo)
                    conv._fused_modules = [(names[k], seq._modules[names[k]]) for k in range(i + 1, j)]
                    for k in range(i + 1, j):
                        seq._modules[names[k]] = nn.Identity()
                    n += 1
                    i = j
                    continue
            i += 1
    if dw_pw:
        fuse_dw_pw(model)
    return n


def unfuse(model):
    """Undo fuse_bn_relu (restores the original BatchNorm2d / ReLU modules; un-pairs DwPwBlocks first)."""
    unfuse_dw_pw(model)
    n = 0
    for seq in [m for m in model.modules() if isinstance(m, nn.Sequential)]:
        names = list(seq._modules.keys())
        for i, name in enumerate(names):
            conv = seq._modules[name]
            if _is_conv_q(conv) and conv._post is not None and hasattr(conv, "_fused_modules"):
                for key, mod in conv._fused_modules:
                    seq._modules[key] = mod
                conv._post = None
                del conv._fused_modules
                n += 1
    return n


def fuse_named_bn(model, example_input=None, rtol=None):
    """Blocks that wire conv -> bn by hand in their forward() (the reference's ResNet-50 Bottleneck,
    nets_imgnet/resnet50.py:24-100: self.conv1 / self.bn1 / self.relu ...) cannot be rewritten by position, but they
    follow the torchvision naming: every Conv2d_Q child called `<prefix>conv<suffix>` whose sibling `<prefix>bn<suffix>`
    is an eval-mode BatchNorm2d of matching width gets that BatchNorm folded into its epilogue (the shared ReLU module
    stays where it is: it is also applied after the residual add), and the BatchNorm child becomes nn.Identity.
    Because a name is only a convention, pass `example_input`: one forward records what every BatchNorm is actually fed,
    and a pair is folded only if bn<k>'s input IS conv<k>'s output tensor (the wiring itself is checked, not a numerical
    consequence of it: in a deep quantized net a one-ulp change of an activation flips codes downstream and moves the
    logits by percents -- SURVEY section 7 -- so an output tolerance cannot tell a wrong pairing from rounding).
    `rtol`: optionally ALSO require the model's output to move by at most this much (tensor-relative); on failure
    everything is rolled back and a RuntimeError says so.  Returns the number of folded pairs; `unfuse_named_bn(model)`
    restores the modules."""
    y0 = None
    conv_out, bn_in = {}, {}
    if example_input is not None:
        hooks = []
        for m in model.modules():
            if _is_conv_q(m):
                hooks.append(m.register_forward_hook(lambda mod, inp, out: conv_out.__setitem__(mod, out)))
            elif isinstance(m, nn.BatchNorm2d):
                hooks.append(m.register_forward_pre_hook(lambda mod, inp: bn_in.__setitem__(mod, inp[0])))
        try:
            with torch.no_grad():
                y0 = model(example_input)
        finally:
            for h in hooks:
                h.remove()
    done = []
    for parent in model.modules():
        if isinstance(parent, nn.Sequential):
            continue   # positional runs are fuse_bn_relu's job
        for name, conv in list(parent._modules.items()):
            if not _is_conv_q(conv) or conv._post is not None or "conv" not in name:
                continue
            bn_name = name.replace("conv", "bn", 1)
            bn = parent._modules.get(bn_name)
            if not isinstance(bn, nn.BatchNorm2d) or bn.training or bn.num_features != conv.out_channels:
                continue
            if conv.bias is not None and not getattr(conv, "_scaled_bias", False):
                continue
            if example_input is not None and (conv not in conv_out or bn_in.get(bn) is not conv_out[conv]):
                continue   # forward() does not apply this bn to this conv's output
            fuse_pair(conv, bn, relu=False)
            conv._named_bn = (parent, bn_name, bn)
            parent._modules[bn_name] = nn.Identity()
            done.append(conv)
    conv_out.clear()
    bn_in.clear()
    if y0 is not None and done and rtol is not None:
        with torch.no_grad():
            y1 = model(example_input)
        err = float((y1 - y0).abs().max() / y0.abs().max().clamp_min(1e-30))
        if not err <= rtol:
            unfuse_named_bn(model)
            raise RuntimeError(f"fuse_named_bn: the model's output moved by {err:.3e} (> {rtol}); nothing was changed")
    return len(done)


def unfuse_named_bn(model):
    """Undo fuse_named_bn."""
    n = 0
    for conv in model.modules():
        if _is_conv_q(conv) and hasattr(conv, "_named_bn"):
            parent, bn_name, bn = conv._named_bn
            parent._modules[bn_name] = bn
            conv._post = None
            del conv._named_bn
            n += 1
    return n


# ------------------------------------------------------------------ depthwise + pointwise in one kernel
class DwPwBlock(nn.Module):
    """[Conv2d_Q 3x3 depthwise, BN, ReLU, Conv2d_Q 1x1, (BN), (ReLU)] as one launch (csrc/conv_dwpw.hip).  Holds the two
    original conv modules (parameters, state-dict keys and scales unchanged: `dw`, `pw`); inference only.  Falls back
    to running them one after the other when the input is not a channels_last ROCm tensor of a supported size."""

    def __init__(self, dw, pw):
        super().__init__()
        self.dw = dw
        self.pw = pw
        self._last_kernel = None

    def forward(self, x):
        from . import _lib
        from .conv2d_func import _f32, _scalar_scale, options
        from .sfp_quant import _stream_handle
        dw, pw = self.dw, self.pw
        ok = (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
              and not self.training and not torch.is_grad_enabled() and options.mfma_passes in (_lib.MFMA_DEFAULT, _lib.MFMA_F16X1)
              and dw.q_bit in (8, 7) and dw._post is not None and dw._post[0] is not None and not (int(dw._post[2]) & 2)
              and pw._post is not None and not (int(pw._post[2]) & 2))
        # measured (profiles/dwpw_bench.py): the one-kernel form only pays on the 32-channel stride-1
        # block; the wider ones run faster as two kernels until the kernel's next version.  options.dwpw_all forces it.
        if ok and not getattr(options, "dwpw_all", False):
            ok = (dw.in_channels, int(dw.stride[0])) in getattr(options, "dwpw_pairs", {(32, 1)})
        if not ok:
            self._last_kernel = None
            return pw(dw(x))
        L = _lib.load()
        N, C, H, W = x.shape

        def desc(m, n, c, h, w):
            return _lib.ConvDesc(n=n, c_in=c, h=h, w=w, c_out=m.out_channels, kh=m.weight.shape[2], kw=m.weight.shape[3],
                                 stride_h=m.stride[0], stride_w=m.stride[1], pad_h=m.padding[0], pad_w=m.padding[1], dil_h=1, dil_w=1,
                                 groups=m.groups, x_layout=_lib.LAYOUT_NHWC, y_layout=_lib.LAYOUT_NHWC, qbits=m.q_bit,
                                 ka=_f32(_scalar_scale(m.Ka, "Ka")), kw_scale=_f32(_scalar_scale(m.Kw, "Kw")),
                                 mfma_passes=options.mfma_passes, reserved=0)

        d1 = desc(dw, N, C, H, W)
        ho, wo = ctypes.c_int64(), ctypes.c_int64()
        with torch.cuda.device(x.device):
            _lib.check(L.slfp_conv2d_out_shape(ctypes.byref(d1), ctypes.byref(ho), ctypes.byref(wo)))
            d2 = desc(pw, N, dw.out_channels, ho.value, wo.value)
            if not L.slfp_dwpw_supported(ctypes.byref(d1), ctypes.byref(d2)):
                self._last_kernel = None
                return pw(dw(x))
            b1 = dw._prep.get(L, d1, dw.weight, want_weight_q=False, cache=True)
            b2 = pw._prep.get(L, d2, pw.weight, want_weight_q=False, cache=True)
            s1, h1, f1 = dw._post
            s2, h2, f2 = pw._post
            if s1.device != x.device:
                s1, h1 = s1.to(x.device), h1.to(x.device)
                dw._post = (s1, h1, f1)
            if s2 is not None and s2.device != x.device:
                s2, h2 = s2.to(x.device), h2.to(x.device)
                pw._post = (s2, h2, f2)
            bias2 = pw.bias.detach().contiguous() if (pw.bias is not None and getattr(pw, "_scaled_bias", False)) else None
            if pw.bias is not None and bias2 is None:
                return pw(dw(x))   # conv2d_Q's raw bias is added outside the kernels
            y = torch.empty((N, pw.out_channels, ho.value, wo.value), dtype=torch.float32, device=x.device,
                            memory_format=torch.channels_last)
            _lib.check(L.slfp_dwpw_fwd(ctypes.byref(d1), ctypes.byref(d2), x.data_ptr(), b1.data_ptr(), s1.data_ptr(), h1.data_ptr(),
                                       int(f1) & 1, b2.data_ptr(), bias2.data_ptr() if bias2 is not None else None,
                                       s2.data_ptr() if s2 is not None else None, h2.data_ptr() if s2 is not None else None,
                                       int(f2) & 1, y.data_ptr(), _stream_handle(x)))
        self._last_kernel = "dwpw_fused_f16x1" if dw.q_bit == 8 else "dwpw_fused_f16_exact"
        dw._last_input, dw._input_q = x.detach(), None
        pw._last_input, pw._input_q = None, None   # the pointwise input never exists as a tensor
        return y


def _is_dw(m):
    return _is_conv_q(m) and m.groups == m.in_channels == m.out_channels and tuple(m.kernel_size) == (3, 3)


def _is_pw(m):
    return _is_conv_q(m) and m.groups == 1 and tuple(m.kernel_size) == (1, 1) and tuple(m.stride) == (1, 1)


def fuse_dw_pw(model):
    """After fuse_bn_relu: replace every [depthwise Conv2d_Q (+BN+ReLU folded), Identity..., pointwise Conv2d_Q (+BN+ReLU
    folded)] run of an nn.Sequential by one DwPwBlock (the pointwise slot; the depthwise slot becomes nn.Identity).
    Whether a pair really runs as one kernel is decided per call (input layout, size, libslfp_hip's
    slfp_dwpw_supported); otherwise the block runs its two convs as before.  Returns the number of blocks formed."""
    n = 0
    for seq in [m for m in model.modules() if isinstance(m, nn.Sequential)]:
        names = list(seq._modules.keys())
        i = 0
        while i < len(names):
            dw = seq._modules[names[i]]
            if _is_dw(dw) and dw._post is not None and dw._post[0] is not None:
                j = i + 1
                while j < len(names) and isinstance(seq._modules[names[j]], nn.Identity):
                    j += 1
                if j < len(names) and _is_pw(seq._modules[names[j]]) and seq._modules[names[j]]._post is not None \
                        and seq._modules[names[j]].in_channels == dw.out_channels:
                    pw = seq._modules[names[j]]
                    blk = DwPwBlock(dw, pw)
                    blk.train(dw.training)   # a new module starts in training mode; follow the convs
                    blk._dw_slot = names[i]
                    seq._modules[names[j]] = blk
                    seq._modules[names[i]] = nn.Identity()
                    n += 1
                    i = j + 1
                    continue
            i += 1
    return n


def unfuse_dw_pw(model):
    """Undo fuse_dw_pw."""
    n = 0
    for seq in [m for m in model.modules() if isinstance(m, nn.Sequential)]:
        names = list(seq._modules.keys())
        for j, name in enumerate(names):
            blk = seq._modules[name]
            if isinstance(blk, DwPwBlock):
                seq._modules[blk._dw_slot] = blk.dw
                seq._modules[name] = blk.pw
                n += 1
    return n


# ------------------------------------------------------------------ 1-byte activation codes between layers
class CodeMaxPool2d(nn.Module):
    """An nn.MaxPool2d inside a code chain (link_codes): uint8 codes are pooled as codes (slfp_maxpool2d_codes: the class of a
    window's largest input -- bit-identical to pooling the float32 tensor and encoding it), anything else goes to the original
    module, which this wrapper keeps (`pool`; it has no parameters, the state dict does not change)."""

    def __init__(self, pool, q_bit):
        super().__init__()
        self.pool = pool
        self.q_bit = int(q_bit)

    def forward(self, x):
        if x.dtype == torch.uint8:
            from .sfp_quant import hip_maxpool_codes
            p = self.pool
            return hip_maxpool_codes(x, p.kernel_size, p.stride, p.padding, self.q_bit)
        return self.pool(x)


def _poolable(m):
    return (isinstance(m, nn.MaxPool2d) and not m.ceil_mode and not m.return_indices
            and (m.dilation == 1 or m.dilation == (1, 1)))


def link_codes(model, example_input=None):
    """After fuse_bn_relu: wherever a Conv2d_Q's (fused BN + ReLU) output feeds the next Conv2d_Q of an nn.Sequential
    directly (only nn.Identity in between -- nets_imgnet/mobilenetv1.py:24-33 after fusion), link the two: the producer's
    epilogue applies the CONSUMER's quantize_act(. / Ka) (utils/conv2d_func.py:21) and stores 1-byte codes, the consumer
    decodes them (libslfp_hip: slfp_conv2d_fwd_codes).  Same classes, same values: the net's output is bit-identical to
    the unlinked fused net (single-pass MFMA mode / SFP<3,3>), activations cross HBM as 1 B per element instead of 4.
    With `example_input` (a channels_last ROCm batch) only links for which both kernels exist are made (one forward
    records the shapes); without it every candidate is linked and combinations without a kernel run through the float32
    interface plus an encode / decode pass (correct, slower).  Inference only.  Returns the number of links."""
    from . import _lib
    from .conv2d_func import _f32, _scalar_scale, options
    shapes = {}
    if example_input is not None:
        hooks = []
        for m in model.modules():
            if _is_conv_q(m):
                hooks.append(m.register_forward_pre_hook(lambda mod, inp: shapes.__setitem__(mod, tuple(inp[0].shape))))
        was = [(m, m._code_out) for m in model.modules() if _is_conv_q(m)]
        try:
            with torch.no_grad():
                model(example_input)
        finally:
            for h in hooks:
                h.remove()

    def supported(m, x_codes, out):
        if example_input is None:
            return True
        shp = shapes.get(m)
        if shp is None or len(shp) != 4:
            return False
        L = _lib.load()
        n, c, h, w = shp
        d = _lib.ConvDesc(n=n, c_in=c, h=h, w=w, c_out=m.out_channels, kh=m.weight.shape[2], kw=m.weight.shape[3],
                          stride_h=m.stride[0], stride_w=m.stride[1], pad_h=m.padding[0], pad_w=m.padding[1], dil_h=m.dilation[0],
                          dil_w=m.dilation[1], groups=m.groups, x_layout=_lib.LAYOUT_NHWC, y_layout=_lib.LAYOUT_NHWC, qbits=m.q_bit,
                          ka=_f32(_scalar_scale(m.Ka, "Ka")), kw_scale=_f32(_scalar_scale(m.Kw, "Kw")),
                          mfma_passes=options.mfma_passes, reserved=0)
        io = _lib.ConvIo(x_codes=1 if x_codes else 0, y_codes=1 if out is not None else 0,
                         y_ka=_f32(out[0]) if out is not None else 1.0, y_qbits=int(out[1]) if out is not None else 8)
        flags = int(m._post[2]) if m._post is not None else 0
        return bool(L.slfp_conv2d_codes_supported(ctypes.byref(d), ctypes.byref(io), 1 if m.bias is not None else 0, flags))

    def eligible(m):
        return (_is_conv_q(m) and m.q_bit in (8, 7) and not m.training and isinstance(m.padding, tuple)
                and (m.bias is None or getattr(m, "_scaled_bias", False)) and not (m._post is not None and (int(m._post[2]) & 2)))

    def flat(seq):   # an nn.Sequential of nn.Sequentials runs its leaves in order (conv_bn / conv_dw blocks of the reference nets)
        for m in seq._modules.values():
            if isinstance(m, nn.Sequential):
                yield from flat(m)
            else:
                yield m

    nested = {c for m in model.modules() if isinstance(m, nn.Sequential) for c in m._modules.values() if isinstance(c, nn.Sequential)}
    n_links = 0
    for seq in [m for m in model.modules() if isinstance(m, nn.Sequential) and m not in nested]:
        allm = [m for m in flat(seq) if not isinstance(m, nn.Identity)]
        # nn.MaxPool2d modules between two convs stay inside the chain (CodeMaxPool2d): drop them from the adjacency list and
        # remember which ones sit behind each conv
        mods, pools_after = [], {}
        for m in allm:
            if _poolable(m) and mods and eligible(mods[-1]):
                pools_after.setdefault(len(mods) - 1, []).append(m)
            else:
                mods.append(m)
        # candidate links: consecutive eligible convs
        cand = [i for i in range(len(mods) - 1) if eligible(mods[i]) and eligible(mods[i + 1])]
        # a conv can consume codes only if its producer link exists; walk left to right and keep links whose two sides have kernels
        linked_in = set()
        for i in cand:
            a, b = mods[i], mods[i + 1]
            out = (float(_scalar_scale(b.Ka, "Ka")), int(b.q_bit))
            a_in = i in linked_in                       # does `a` itself read codes?
            if not supported(a, a_in, out):
                # head of a run without a float32 -> codes kernel: enter the chain through one slfp_encode_f32 pass if at least
                # two more layers then run on codes
                if a_in or not (i + 1 in cand and supported(b, True, (float(_scalar_scale(mods[i + 2].Ka, "Ka")), int(mods[i + 2].q_bit)))):
                    continue
            # b with codes in: it may or may not write codes itself; require the float32-out form here (the code-out form is
            # checked when its own link is made; if that fails b keeps float32 out)
            if not supported(b, True, None):
                continue
            if pools_after.get(i) and a.out_channels % 4:
                continue   # slfp_maxpool2d_codes needs C % 4 == 0
            a._code_out = out
            linked_in.add(i + 1)
            n_links += 1
            for pm in pools_after.get(i, []):   # the pools between a and b now see codes
                for parent in [m for m in model.modules() if not isinstance(m, CodeMaxPool2d)]:
                    for name, child in list(parent._modules.items()):
                        if child is pm:
                            parent._modules[name] = CodeMaxPool2d(pm, b.q_bit)
    return n_links


def link_codes_traced(model, example_input):
    """link_codes for blocks that wire their layers by hand in forward() (the reference's ResNet-50 Bottleneck,
    nets_imgnet/resnet50.py:74-100: conv1 -> bn1 -> relu -> conv2 -> bn2 -> relu -> conv3 -> bn3 -> (+ identity) -> relu, with ONE
    shared nn.ReLU).  One forward records the tensors: a Conv2d_Q `b` whose input IS the output of a Conv2d_Q `a` -- directly or
    through nn.Identity (a folded BatchNorm, fuse_named_bn) and / or an nn.ReLU module -- is a candidate; `a` gets the ReLU
    folded into its epilogue (the module's own relu() then runs on the uint8 codes, where it is the identity) and writes `b`'s
    codes.  Links are made only where libslfp_hip has both kernels (slfp_conv2d_codes_supported), never for a producer with
    two consumers, and the whole set is VERIFIED: the linked model must reproduce the unlinked output bit for bit on
    `example_input`, otherwise (a functional use of the tensor that hooks cannot see, e.g. a torch.cat) everything is rolled
    back and 0 is returned.  Inference only; unlink_codes undoes it.  Returns the number of links."""
    import ctypes as _ct
    from . import _lib
    from .conv2d_func import _f32, _scalar_scale, options
    conv_io, relu_io, pool_io, order, keep = {}, [], [], [], []
    hooks = []
    for m in model.modules():
        if _is_conv_q(m):
            def _rec(mod, inp, out):
                conv_io[mod] = (inp[0], out)
                order.append(mod)
                keep.append((inp[0], out))
            hooks.append(m.register_forward_hook(_rec))
        elif isinstance(m, nn.ReLU):
            def _rec_relu(mod, inp, out):   # (returns None: a hook's return value would replace the module's output)
                relu_io.append((inp[0], out))
                keep.append((inp[0], out))
            hooks.append(m.register_forward_hook(_rec_relu))
        elif _poolable(m):
            def _rec_pool(mod, inp, out):
                pool_io.append((inp[0], out, mod))
                keep.append((inp[0], out))
            hooks.append(m.register_forward_hook(_rec_pool))
    try:
        with torch.no_grad():
            y0 = model(example_input)
    finally:
        for h in hooks:
            h.remove()
    if len(order) != len(set(order)):
        return 0   # a module that runs twice per forward has no single producer / consumer
    producer = {id(out): c for c, (_, out) in conv_io.items()}
    relu_src = {id(out): inp for inp, out in relu_io}
    pool_src = {id(out): (inp, mod) for inp, out, mod in pool_io}
    pool_uses = {}
    for _, _, mod in pool_io:
        pool_uses[mod] = pool_uses.get(mod, 0) + 1

    def eligible(m, producer_side=False):
        return (m.q_bit in (8, 7) and not m.training and isinstance(m.padding, tuple) and (m._code_out is None or not producer_side)
                and (m.bias is None or getattr(m, "_scaled_bias", False)) and not (m._post is not None and (int(m._post[2]) & 2)))

    def supported(m, x_codes, out, flags):
        xin = conv_io[m][0]
        if xin.dim() != 4:
            return False
        n, c, h, w = xin.shape
        d = _lib.ConvDesc(n=n, c_in=c, h=h, w=w, c_out=m.out_channels, kh=m.weight.shape[2], kw=m.weight.shape[3],
                          stride_h=m.stride[0], stride_w=m.stride[1], pad_h=m.padding[0], pad_w=m.padding[1], dil_h=m.dilation[0],
                          dil_w=m.dilation[1], groups=m.groups, x_layout=_lib.LAYOUT_NHWC, y_layout=_lib.LAYOUT_NHWC, qbits=m.q_bit,
                          ka=_f32(_scalar_scale(m.Ka, "Ka")), kw_scale=_f32(_scalar_scale(m.Kw, "Kw")),
                          mfma_passes=options.mfma_passes, reserved=0)
        io = _lib.ConvIo(x_codes=1 if x_codes else 0, y_codes=1 if out is not None else 0,
                         y_ka=_f32(out[0]) if out is not None else 1.0, y_qbits=int(out[1]) if out is not None else 8)
        return bool(_lib.load().slfp_conv2d_codes_supported(_ct.byref(d), _ct.byref(io), 1 if m.bias is not None else 0, flags))

    cand = {}
    for b in order:
        t, via_relu, pools = conv_io[b][0], False, []
        for _ in range(4):   # look back through nn.ReLU / nn.MaxPool2d modules (a ReLU'd tensor pools the same either way)
            if id(t) in relu_src:
                t, via_relu = relu_src[id(t)], True
            elif id(t) in pool_src and pool_uses[pool_src[id(t)][1]] == 1:
                t, pm = pool_src[id(t)]
                pools.append(pm)
            else:
                break
        a = producer.get(id(t))
        if a is None or a is b or not eligible(a, producer_side=True) or not eligible(b):
            continue
        if conv_io[b][0].dtype == torch.uint8:
            continue   # already linked
        if pools and a.out_channels % 4:
            continue   # slfp_maxpool2d_codes needs C % 4 == 0
        cand.setdefault(a, []).append((b, via_relu, pools))
    made, reads_codes, wrapped = [], set(), []
    for a in order:   # execution order: whether `a` itself reads codes is known when its own link is decided
        if a not in cand or len(cand[a]) != 1:
            continue
        b, via_relu, pools = cand[a][0]
        flags = (int(a._post[2]) if a._post is not None else 0) | (1 if via_relu else 0)
        out = (float(_scalar_scale(b.Ka, "Ka")), int(b.q_bit))
        bflags = int(b._post[2]) if b._post is not None else 0
        a_in = a in reads_codes or conv_io[a][0].dtype == torch.uint8   # does `a` itself read codes (an earlier link)?
        if not supported(a, a_in, out, flags) or not supported(b, True, b._code_out, bflags):
            continue
        made.append((a, a._post))
        a._post = ((a._post[0], a._post[1]) if a._post is not None else (None, None)) + (flags,)
        a._code_out = out
        a._pre_link_post = made[-1][1]
        reads_codes.add(b)
        for pm in pools:   # the pools between a and b now see codes
            for parent in [m for m in model.modules() if not isinstance(m, CodeMaxPool2d)]:
                for name, child in list(parent._modules.items()):
                    if child is pm:
                        parent._modules[name] = CodeMaxPool2d(pm, b.q_bit)
                        wrapped.append((parent, name, pm))
    if not made:
        return 0
    ok = False
    try:
        with torch.no_grad():
            y1 = model(example_input)
        ok = y1.dtype == y0.dtype and torch.equal(y1, y0)
    except Exception:
        ok = False
    if not ok:
        for a, post in made:
            a._post, a._code_out = post, None
            del a._pre_link_post
        for parent, name, pm in wrapped:
            parent._modules[name] = pm
        return 0
    return len(made)


def unlink_codes(model):
    """Undo link_codes / link_codes_traced."""
    n = 0
    for parent in model.modules():
        for name, child in list(parent._modules.items()):
            if isinstance(child, CodeMaxPool2d):
                parent._modules[name] = child.pool
    for m in model.modules():
        if _is_conv_q(m) and m._code_out is not None:
            m._code_out = None
            if hasattr(m, "_pre_link_post"):
                m._post = m._pre_link_post
                del m._pre_link_post
            n += 1
    return n
