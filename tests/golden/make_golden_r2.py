#!/usr/bin/env python3
"""Round-2 fixtures from the IMPORTED reference (build container only; needs /root/reference, read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r2.py

  net224_golden.npz   BASELINE config 2: nets_imgnet/mobilenetv1.py MobileNetV1_Q(qbit=8) at 224x224 with
                      netgen.fill_parameters and BatchNorm statistics calibrated on 16 of the images (netgen.calibrate_bn_; stored), 64 seeded images: logits [64, 1000] (float32) and strided samples of two
                      intermediate activations (after block 1 and block 7) for the first 4 images.
  calib_golden.json   the statistics the reference's calibration pass get_scale_factor
                      (cifar100_train_eval.py:213-277) produces on the reference's CIFAR MobileNetV1_Q: the loop's
                      arithmetic (reset stashes, forward, collect input_q / weight_q / outputs, max|cat|) is restated
                      here because the harness itself imports torchvision / tensorboardX, which are not installed; the
                      NET, its stash read-out and the quantizers are the reference's own code.  Qbits 32 and 8,
                      two batches of 8 seeded images; plus the text of the two files the harness writes (:287-301).
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

sys.modules.setdefault("torchsummary", types.ModuleType("torchsummary"))
sys.modules["torchsummary"].summary = lambda *a, **k: None
import netgen  # noqa: E402
import utils.sfp_quant  # noqa: E402

assert "/root/reference" in sys.modules["utils.sfp_quant"].__file__, "must import the REFERENCE utils"


def make_net224():
    from nets_imgnet.mobilenetv1 import MobileNetV1_Q
    torch.set_num_threads(8)
    x = netgen.net_input224(64)
    # BatchNorm statistics of a "trained" net: calibrated once on 16 images with identity quantizers (see netgen)
    m32 = netgen.fill_parameters(MobileNetV1_Q(ch_in=3, qbit=32).eval())
    stats = netgen.calibrate_bn_(m32, x[:16].clone())
    bn = {}
    for i, (mu, var) in stats.items():
        bn[f"bn_mean_{i}"] = mu.astype(np.float32)
        bn[f"bn_var_{i}"] = var.astype(np.float32)
    m = netgen.load_bn_stats_(netgen.fill_parameters(MobileNetV1_Q(ch_in=3, qbit=8).eval()), bn)
    outs = []
    feats = {}
    with torch.no_grad():
        for i in range(0, 64, 8):
            xb = x[i:i + 8].clone()
            if i == 0:
                h = xb
                for bi, blk in enumerate(list(m.model)[:8]):
                    h = blk(h)
                    if bi == 1:
                        feats["block1"] = h[:4, :, ::16, ::16].numpy().copy()
                    if bi == 7:
                        feats["block7"] = h[:4, ::8, ::2, ::2].numpy().copy()
            outs.append(m(xb).numpy())
            print("net224 batch", i // 8, flush=True)
    logits = np.concatenate(outs)
    print("logits range", logits.min(), logits.max(), "top1 of first 8:", logits[:8].argmax(1).tolist())
    return dict(logits_q8=logits.astype(np.float32), **{k + "_q8": v.astype(np.float32) for k, v in feats.items()}, **bn)


def get_scale_factor_arith(model, batches):
    """cifar100_train_eval.py:213-277, accuracy bookkeeping left out."""
    layer_inputs, layer_outputs, layer_weights = {}, {}, {}
    model.eval()
    for inputs in batches:
        model.reset_layer_inputs_outputs()
        model.reset_layer_weights()
        model(inputs)
        for idx, t in model.get_layer_inputs().items():
            layer_inputs.setdefault(idx, []).append(t.detach().cpu())
        for idx, t in model.get_layer_outputs().items():
            layer_outputs.setdefault(idx, []).append(t.detach().cpu())
        for idx, t in model.get_layer_weights().items():
            layer_weights.setdefault(idx, []).append(t.detach().cpu())
    mx = lambda d: {idx: torch.max(torch.abs(torch.cat(v, dim=0))).item() for idx, v in d.items()}  # noqa: E731
    return mx(layer_inputs), mx(layer_outputs), mx(layer_weights)


def files_text(net, max_in, max_out, max_w):
    """cifar100_train_eval.py:287-301."""
    a = ""
    for idx, v in max_in.items():
        a += f"Layer {idx} Max Absolute Input:\n" + str(v) + "\n\n"
    for idx, v in max_out.items():
        a += f"Layer {idx} Max Absolute Output:\n" + str(v) + "\n\n"
    b = ""
    for idx, v in max_w.items():
        b += f"Layer {idx} Max Absolute weight:\n" + str(v) + "\n\n"
    return {f"max_inout_{net}.txt": a, f"max_weight_{net}.txt": b}


def make_calib():
    from nets_cifar.mobilenetv1 import MobileNetV1_Q
    out = {}
    x = netgen.net_input(16, seed=321)
    for q in (32, 8):
        m = netgen.fill_parameters(MobileNetV1_Q(ch_in=3, qbit=q))
        with torch.no_grad():
            mi, mo, mw = get_scale_factor_arith(m, [x[:8].clone(), x[8:].clone()])
        out[f"q{q}"] = {"max_in": {str(k): v for k, v in mi.items()}, "max_out": {str(k): v for k, v in mo.items()},
                        "max_w": {str(k): v for k, v in mw.items()}, "files": files_text("MobileNetV1", mi, mo, mw)}
        print(f"calib q={q}: inputs {len(mi)}, outputs {len(mo)}, weights {len(mw)}; max_in[0]={mi[0]:.6f} max_in[27]={mi[27]:.6f}")
    return out


if __name__ == "__main__":
    json.dump(make_calib(), open(os.path.join(HERE, "calib_golden.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "net224_golden.npz"), **make_net224())
    for f in ("calib_golden.json", "net224_golden.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
