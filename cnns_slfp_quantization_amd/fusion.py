"""Inference-time fusion of the eval-mode BatchNorm2d (+ ReLU) that follows every Conv2d_Q in the
reference nets into the conv kernels' epilogue.

The reference runs  Conv2d_Q -> nn.BatchNorm2d -> nn.ReLU(inplace)  as three modules
(nets_imgnet/mobilenetv1.py:24-41); stock BN + ReLU move 16 B per conv output element, twice the
hot path's own traffic.  `fuse_bn_relu(model)` rewrites every such run inside an nn.Sequential:
the BatchNorm2d's running statistics and affine are folded into a per-channel (scale, shift) that
the HIP epilogue applies after the reference's (out*Ka)*Kw roundings (slfp_conv2d_fwd_post), and
the BN / ReLU modules are replaced by nn.Identity.  Call it AFTER load_state_dict and model.eval().
Nets that wire conv->bn by hand (ResNet-50 blocks) can use `fuse_pair(conv, bn, relu)`.
"""
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


def fuse_bn_relu(model):
    """Fuse every [Conv2d_Q, BatchNorm2d(eval), (layerout_quantize_func), (ReLU)] run found in nn.Sequential
    containers (nets_imgnet/mobilenetv1.py:24-41; nets_cifar/mobilenetv1.py:196-231 for the layerout form; a
    Swish / GELU after the quantizer stays a module).  Returns the number of fused convolutions."""
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
    return n


def unfuse(model):
    """Undo fuse_bn_relu (restores the original BatchNorm2d / ReLU modules)."""
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
