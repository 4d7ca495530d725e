#!/usr/bin/env python3
"""Extract the Conv2d_Q / Linear_Q layer tables of the reference nets (shapes, strides,
groups, bias flag, calibration scales Ka / Kw) into
cnns_slfp_quantization_amd/data/layer_specs.json.

Runs only in the build container (imports /root/reference; `torchsummary`, which the nets
import but never call on the forward path, is stubbed in-process).  The JSON is DATA:
layer geometry observed with forward hooks + the per-layer scale constants the nets carry
(e.g. nets_imgnet/mobilenetv1.py:15-19); no reference code is stored.
"""
import json
import os
import sys
import types
import warnings

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.modules["torchsummary"] = types.ModuleType("torchsummary")
sys.modules["torchsummary"].summary = lambda *a, **k: None

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402


def trace(model, res, qbit_name):
    rows = []

    def hook(mod, inp, out):
        x = inp[0]
        if isinstance(mod, nn.Conv2d):
            rows.append(dict(kind="conv", c_in=mod.in_channels, c_out=mod.out_channels, k=list(mod.kernel_size),
                             stride=list(mod.stride), pad=list(mod.padding), dil=list(mod.dilation), groups=mod.groups,
                             bias=mod.bias is not None, h=int(x.shape[2]), w=int(x.shape[3]),
                             h_out=int(out.shape[2]), w_out=int(out.shape[3]),
                             Ka=float(mod.Ka), Kw=float(mod.Kw)))
        else:
            rows.append(dict(kind="linear", c_in=mod.in_features, c_out=mod.out_features, bias=mod.bias is not None,
                             Ka=float(mod.Ka), Kw=float(mod.Kw)))

    hs = [m.register_forward_hook(hook) for m in model.modules()
          if hasattr(m, "Ka") and hasattr(m, "Kw") and isinstance(m, (nn.Conv2d, nn.Linear))]
    model.eval()
    if hasattr(model, "reset_layer_inputs_outputs"):
        model.reset_layer_inputs_outputs()
    if hasattr(model, "reset_layer_weights"):
        model.reset_layer_weights()
    with torch.no_grad():
        model(torch.randn(1, 3, res, res))
    for h in hs:
        h.remove()
    return rows


def main():
    from nets_imgnet.mobilenetv1 import MobileNetV1_Q as MBV1_IMG
    from nets_cifar.mobilenetv1 import MobileNetV1_Q as MBV1_CIFAR
    from nets_cifar.vgg16 import VGG16_Q
    from nets_cifar.shufflenet_v2 import ShuffleNetV2
    from nets_imgnet.resnet50 import ResNet50
    from nets_imgnet.squeezenet1_0 import SqueezeNet
    from nets_imgnet.alexnet import AlexNet
    specs = {}
    # q_bit=32: the geometry and the scales do not depend on it, and the trace stays NaN-free
    specs["mobilenetv1_imagenet224"] = dict(res=224, source="nets_imgnet/mobilenetv1.py:43-59",
                                            layers=trace(MBV1_IMG(ch_in=3, qbit=32), 224, "qbit"))
    specs["mobilenetv1_cifar32"] = dict(res=32, source="nets_cifar/mobilenetv1.py:43-64",
                                        layers=trace(MBV1_CIFAR(ch_in=3, qbit=32), 32, "qbit"))
    specs["vgg16_224"] = dict(res=224, source="nets_cifar/vgg16.py (CIFAR topology fed 224x224)",
                              layers=trace(VGG16_Q(qbit=32), 224, "qbit"))
    specs["resnet50_imagenet224"] = dict(res=224, source="nets_imgnet/resnet50.py",
                                         layers=trace(ResNet50(qbit=32), 224, "qbit"))
    specs["squeezenet1_0_imagenet224"] = dict(res=224, source="nets_imgnet/squeezenet1_0.py",
                                              layers=trace(SqueezeNet(qbit=32), 224, "qbit"))
    specs["shufflenetv2_224"] = dict(res=224, source="nets_cifar/shufflenet_v2.py (CIFAR topology fed 224x224)",
                                     layers=trace(ShuffleNetV2(qbit=32, ratio=1, class_num=100), 224, "qbit"))
    specs["alexnet_imagenet224"] = dict(res=224, source="nets_imgnet/alexnet.py",
                                        layers=trace(AlexNet(qbit=32), 224, "qbit"))
    out = os.path.join(ROOT, "cnns_slfp_quantization_amd", "data", "layer_specs.json")
    json.dump(specs, open(out, "w"), indent=0)
    for k, v in specs.items():
        convs = [r for r in v["layers"] if r["kind"] == "conv"]
        ine = sum(r["c_in"] * r["h"] * r["w"] for r in convs)
        oute = sum(r["c_out"] * r["h_out"] * r["w_out"] for r in convs)
        we = sum(r["c_out"] * (r["c_in"] // r["groups"]) * r["k"][0] * r["k"][1] for r in convs)
        macs = sum(r["c_out"] * r["h_out"] * r["w_out"] * (r["c_in"] // r["groups"]) * r["k"][0] * r["k"][1] for r in convs)
        print(f"{k}: {len(convs)} conv, in_elems {ine}, out_elems {oute}, w_elems {we}, MACs {macs/1e6:.1f} M")


if __name__ == "__main__":
    main()
