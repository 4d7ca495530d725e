#!/usr/bin/env python3
"""Round-3 fixtures from the IMPORTED reference (build container only; needs /root/reference, read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r3.py

  nets_r3_golden.npz   BASELINE configs 3-5 end to end, small batch, 224x224: the reference's own ResNet50 (Qbits 8),
                       VGG16_Q (Qbits 8), SqueezeNet (Qbits 7) and ShuffleNetV2 (Qbits 7; `layerout` quantizers live) with
                       name-seeded parameters (netgen_r3.param_by_name), BatchNorm statistics calibrated once with identity
                       quantizers (stored), for SqueezeNet -- no BatchNorm -- a per-layer weight gain that keeps every
                       layer's input inside its quantizer's range (stored).  Stored: logits, one strided mid-network
                       activation per net, the {module name: (Ka, Kw)} manifest read off the reference modules.
                       While generating, THIS repo's topology builders (netgen_r3.py) are run with the REFERENCE's
                       operator classes and must reproduce the reference net's logits exactly -- the wiring of the
                       test-side nets is thereby checked against the reference itself.
  rawbias_golden.npz   conv2d_Q(bias=True): the class whose bias is handed to F.conv2d unscaled (utils/conv2d_func.py:
                       8-26), three small geometries, Qbits 8 and 7.
Only data is stored."""
import json
import os
import sys
import types
import warnings

import numpy as np

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, HERE)
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

sys.modules.setdefault("torchsummary", types.ModuleType("torchsummary"))
sys.modules["torchsummary"].summary = lambda *a, **k: None
import netgen_r3 as ng  # noqa: E402
import utils.conv2d_func as ref_cf  # noqa: E402
import utils.sfp_quant as ref_sq  # noqa: E402

assert "/root/reference" in ref_cf.__file__ and "/root/reference" in ref_sq.__file__, "must import the REFERENCE utils"

BATCH = 4
SEED = 3030


def ref_model(net, q):
    if net == "resnet50":
        from nets_imgnet.resnet50 import ResNet50
        return ResNet50(qbit=q)
    if net == "squeezenet":
        from nets_imgnet.squeezenet1_0 import SqueezeNet
        return SqueezeNet(qbit=q)
    if net == "vgg16":
        from nets_cifar.vgg16 import VGG16_Q
        return VGG16_Q(qbit=q)
    from nets_cifar.shufflenet_v2 import ShuffleNetV2
    m = ShuffleNetV2(qbit=q)
    return m


def run(m, x):
    for fn in ("reset_layer_inputs_outputs", "reset_layer_weights"):
        if hasattr(m, fn):
            getattr(m, fn)()   # the CIFAR nets fill these dictionaries in forward (shufflenet_v2.py:197-309 needs them to exist)
    with torch.no_grad():
        return m.eval()(x)


def manifest_of(m):
    return {name: (float(mod.Ka), float(mod.Kw)) for name, mod in m.named_modules()
            if hasattr(mod, "Ka") and hasattr(mod, "Kw") and isinstance(mod, (nn.Conv2d, nn.Linear))}


def calibrate_bn_by_name_(m, x):
    """One identity-quantizer forward during which every BatchNorm2d takes the statistics of its own input."""
    hooks = []

    def pre(mod, args):
        h = args[0].detach()
        mod.running_mean.copy_(h.mean(dim=(0, 2, 3)))
        mod.running_var.copy_(h.var(dim=(0, 2, 3), unbiased=False) + 1e-3)

    for mod in m.modules():
        if isinstance(mod, nn.BatchNorm2d):
            hooks.append(mod.register_forward_pre_hook(pre))
    run(m, x)
    for h in hooks:
        h.remove()
    out = {}
    for name, mod in m.named_modules():
        if isinstance(mod, nn.BatchNorm2d):
            out[f"bn:{name}:mean"] = mod.running_mean.clone().numpy()
            out[f"bn:{name}:var"] = mod.running_var.clone().numpy()
    return out


def lsuv_(m, x, manifest):
    """Nets without BatchNorm (SqueezeNet): scale each conv's weights, in execution order, so that the RMS of its output is
    3 x the Ka of the layer that runs next (identity quantizers): every layer then sees inputs inside its code range."""
    order = []
    hs = [mod.register_forward_hook(lambda mod, i, o: order.append(mod)) for mod in m.modules() if isinstance(mod, nn.Conv2d)]
    run(m, x)
    for h in hs:
        h.remove()
    names = {mod: name for name, mod in m.named_modules()}
    gains = {}
    for i, mod in enumerate(order):
        target = 3.0 * (manifest[names[order[i + 1]]][0] if i + 1 < len(order) else 1.0)
        got = {}
        h = mod.register_forward_hook(lambda mod, inp, o: got.__setitem__("rms", float(o.detach().pow(2).mean().sqrt())))
        run(m, x)
        h.remove()
        g = target / max(got["rms"], 1e-12)
        with torch.no_grad():
            mod.weight.mul_(g)
        gains[names[mod] + ".weight"] = g
    return gains


def make_net(net, q):
    torch.set_num_threads(8)
    x = ng.net_input224(BATCH, seed=700 + len(net))
    m32 = ng.fill_parameters_by_name(ref_model(net, 32).eval(), SEED)
    manifest = manifest_of(m32)
    gains = lsuv_(m32, x, manifest) if net == "squeezenet" else {}
    bn = calibrate_bn_by_name_(m32, x)
    ref = ng.load_bn_stats_by_name_(ng.fill_parameters_by_name(ref_model(net, q).eval(), SEED, gains), bn)
    feats = {}
    tap = {"resnet50": "layer2", "squeezenet": "features.5", "vgg16": "layer3", "shufflenetv2": "stage3"}[net]
    h = dict(ref.named_modules())[tap].register_forward_hook(lambda mod, i, o: feats.__setitem__("mid", o.detach()[:2, ::8, ::4, ::4].numpy().copy()))
    logits = run(ref, x).numpy()
    h.remove()
    assert np.isfinite(logits).all(), net
    # this repo's topology code, on the REFERENCE's operators: must give the reference net's logits bit for bit
    f = ng.Factories(ref_cf, q, manifest, layerout=ref_sq.layerout_quantize_func)
    mine = ng.BUILDERS[net](f).eval()
    assert [(k, tuple(v.shape)) for k, v in sorted(mine.state_dict().items())] == \
           [(k, tuple(v.shape)) for k, v in sorted(ref.state_dict().items())], f"{net}: state-dict keys differ from the reference's"
    ng.load_bn_stats_by_name_(ng.fill_parameters_by_name(mine, SEED, gains), bn)
    mine_logits = run(mine, x).numpy()
    assert np.array_equal(mine_logits.view(np.uint32), logits.view(np.uint32)), (net, float(np.abs(mine_logits - logits).max()))
    print(f"{net} q{q}: logits range [{logits.min():.3f}, {logits.max():.3f}], top1 {logits.argmax(1).tolist()}, "
          f"mid {feats['mid'].shape} absmax {np.abs(feats['mid']).max():.3f}; test-side topology == reference: exact", flush=True)
    out = {f"{net}:logits": logits.astype(np.float32), f"{net}:mid": feats["mid"].astype(np.float32),
           f"{net}:manifest": np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8),
           f"{net}:gains": np.frombuffer(json.dumps(gains).encode(), dtype=np.uint8),
           f"{net}:meta": np.array([q, BATCH, 700 + len(net), SEED], dtype=np.int64)}
    out.update({f"{net}:{k}": v.astype(np.float32) for k, v in bn.items()})
    return out


def make_rawbias():
    out = {}
    rng = np.random.default_rng(17)
    cases = {"dw": (2, 32, 10, 10, 32, 3, 1, 1, 32), "pw": (2, 64, 9, 9, 96, 1, 1, 0, 1), "dense": (2, 16, 8, 8, 32, 3, 2, 1, 1)}
    for name, (N, C, H, W, O, k, s, p, g) in cases.items():
        Ka, Kw = np.float64(0.23), np.float64(0.041)
        x = (np.abs(rng.standard_normal((N, C, H, W))) * 4 * Ka).astype(np.float32)
        w = (rng.standard_normal((O, C // g, k, k)) * 3 * Kw).astype(np.float32)
        b = (rng.standard_normal(O) * 2.0).astype(np.float32)   # in units of the quantized domain: the bias is NOT rescaled
        out[f"{name}_meta"] = np.array([N, C, H, W, O, k, s, p, g], dtype=np.int64)
        out[f"{name}_scales"] = np.array([Ka, Kw])
        out[f"{name}_x"], out[f"{name}_w"], out[f"{name}_b"] = x, w, b
        for q in (8, 7):
            m = ref_cf.conv2d_Q(q_bit=q, Kw=Kw, Ka=Ka)(C, O, k, Kw, Ka, s, p, groups=g, bias=True).eval()
            with torch.no_grad():
                m.weight.copy_(torch.from_numpy(w))
                m.bias.copy_(torch.from_numpy(b))
                out[f"{name}_y_q{q}"] = m(torch.from_numpy(x)).numpy().astype(np.float32)
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "rawbias_golden.npz"), **make_rawbias())
    allout = {}
    for net, q in (("squeezenet", 7), ("shufflenetv2", 7), ("vgg16", 8), ("resnet50", 8)):
        allout.update(make_net(net, q))
    np.savez_compressed(os.path.join(HERE, "nets_r3_golden.npz"), **allout)
    for f in ("rawbias_golden.npz", "nets_r3_golden.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
