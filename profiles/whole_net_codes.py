#!/usr/bin/env python3
"""Diagnostic: whole VGG16_Q / ResNet50 (the round-3 fixture nets, tests/golden/netgen_r3.py topologies on the drop-in modules)
through the drop-in modules: stock BatchNorm / ReLU modules vs fused epilogues vs fused + 1-byte code hand-overs
(fusion.link_codes / link_codes_traced).   python profiles/whole_net_codes.py [batch]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import netgen_r3 as ng
import utils.conv2d_func as cf
import utils.sfp_quant as sq
from cnns_slfp_quantization_amd import fusion
dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
gold = np.load(os.path.join(ROOT, "tests", "golden", "nets_r3_golden.npz"))

def build(net):
    q, _, in_seed, seed = [int(v) for v in gold[f"{net}:meta"]]
    manifest = json.loads(bytes(gold[f"{net}:manifest"]).decode())
    gains = json.loads(bytes(gold[f"{net}:gains"]).decode())
    m = ng.BUILDERS[net](ng.Factories(cf, q, manifest, layerout=sq.layerout_quantize_func))
    ng.fill_parameters_by_name(m, seed, gains)
    ng.load_bn_stats_by_name_(m, {k[len(net) + 1:]: gold[k] for k in gold.files if k.startswith(f"{net}:bn:")})
    x = ng.net_input224(4, in_seed).repeat((batch + 3) // 4, 1, 1, 1)[:batch]
    return m.to(dev).eval().to(memory_format=torch.channels_last), x.to(dev).contiguous(memory_format=torch.channels_last)

def rate(m, x, steps=20, warm=5):
    with torch.no_grad():
        for _ in range(warm): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): m(x)
        torch.cuda.synchronize()
    return batch * steps / (time.perf_counter() - t0)

for net in ("vgg16", "resnet50"):
    m, x = build(net)
    r_stock = rate(m, x)
    with torch.no_grad():
        n_f = fusion.fuse_bn_relu(m) + fusion.fuse_named_bn(m, example_input=x)
        y_f = m(x)
    r_fused = rate(m, x)
    with torch.no_grad():
        n_l = fusion.link_codes(m, x) + fusion.link_codes_traced(m, x)
        same = bool(torch.equal(m(x), y_f))
    r_codes = rate(m, x)
    print(f"{net} batch {batch}: stock modules {r_stock:.0f} images/s; {n_f} conv+bn pairs fused {r_fused:.0f}; + {n_l} code hand-overs {r_codes:.0f} "
          f"(logits bit-identical: {same})")
