#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the IMPORTED reference.

Runs only in the build container (needs /root/reference, read-only; never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--skip-sweep]

What it does
  1. PINS THE ORACLE: sweeps every float32 in [2^-6, 2^5) (both signs folded: the codec is
     odd) through the reference's quantize_act / quantize_weight for Qbits 8 and 7 and checks
     oracle/slfp_oracle.c and oracle/torch_port.py bit-for-bit; checks edge cases; checks
     the oracle's conv2d/linear against the reference's Conv2d_Q / Linear_Q modules.
  2. WRITES FIXTURES (inputs + the reference's outputs, data only):
       codec_golden.npz   known-answer vectors for the three codecs (+ scaled division)
       conv_golden.npz    per-layer Conv2d_Q / Conv2d_Q(bias) / Linear_Q cases, Qbits 8/7/32
       sweep_report.json  the sweep verdicts (counts, mismatches = 0)
The inputs come from this repo's own numpy generator (seeded); only data is stored.
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")  # `utils` must be the REFERENCE package here
sys.path.insert(1, ROOT)

import torch  # noqa: E402

from oracle import slfp_oracle as so  # noqa: E402
from oracle import torch_port as tp  # noqa: E402
from utils.conv2d_func import conv2d_Q, conv2d_Q_bias, linear_Q  # noqa: E402  (reference)
from utils.sfp_quant import quantize_act, quantize_layerout, quantize_weight  # noqa: E402  (reference)

assert "/root/reference" in sys.modules["utils.sfp_quant"].__file__, "must import the REFERENCE utils"

FMTS = {  # name -> (oracle fmt, reference callable, torch_port args)
    "act8": (so.FMT_ACT8, quantize_act(8), (8, "act")),
    "w8": (so.FMT_W8, quantize_weight(8), (8, "weight")),
    "act7": (so.FMT_SFP7, quantize_act(7), (7, "act")),
    "w7": (so.FMT_SFP7, quantize_weight(7), (7, "weight")),
}


def ref_bits(fn, bits_u32):
    x = torch.from_numpy(bits_u32.view(np.float32).copy())
    return fn(x).numpy().view(np.uint32)


def same_bits(a, b):
    """bit-equal, treating every NaN as equal to every NaN."""
    na = (a & 0x7FFFFFFF) > 0x7F800000
    nb = (b & 0x7FFFFFFF) > 0x7F800000
    return np.array_equal(na, nb) and np.array_equal(a[~na], b[~nb])


def sweep():
    """Exhaustive [2^-6, 2^5): 11 binades x 2^23 patterns per format."""
    report = {}
    lo, hi = 0x3C800000, 0x42000000
    chunk = 1 << 23
    for name, (fmt, fn, tpa) in FMTS.items():
        t0 = time.time()
        bad_c = bad_t = 0
        flips = []
        prev_last = None
        for start in range(lo, hi, chunk):
            bits = np.arange(start, start + chunk, dtype=np.uint32)
            r = ref_bits(fn, bits)
            c = so.quantize(bits.view(np.float32), 1.0, fmt).view(np.uint32)
            bad_c += int((r != c).sum())
            if start < lo + 3 * chunk:  # torch_port shares the ATen kernels: 3 binades suffice
                t = tp.fake_quant(torch.from_numpy(bits.view(np.float32).copy()), *tpa).numpy().view(np.uint32)
                bad_t += int((r != t).sum())
            # negative twin on a thinned subset
            nb = bits[::257] | np.uint32(0x80000000)
            rn = ref_bits(fn, nb)
            cn = so.quantize(nb.view(np.float32), 1.0, fmt).view(np.uint32)
            bad_c += int((rn != cn).sum())
            d = np.nonzero(r[1:] != r[:-1])[0]
            flips.extend((bits[d + 1]).tolist())
            if prev_last is not None and prev_last != r[0]:
                flips.append(int(bits[0]))
            prev_last = r[-1]
        report[name] = {"patterns": hi - lo, "oracle_mismatch": bad_c, "torch_port_mismatch": bad_t,
                        "n_flip_points": len(flips), "seconds": round(time.time() - t0, 1)}
        report[name + "_flips"] = flips
        print(name, report[name], flush=True)
        assert bad_c == 0 and bad_t == 0, f"oracle does not match the reference for {name}"
    return report


def edge_inputs():
    e = [0.0, -0.0, 1e-12, -1e-12, 1e-45, 0.01, 0.06251, 0.125, 0.1, 0.2, 1.0, 15.0,  # + the reference KAT
         0.0625, np.nextafter(np.float32(0.0625), np.float32(0)), np.nextafter(np.float32(0.125), np.float32(0)),
         15.32165, np.nextafter(np.float32(15.32165), np.float32(16)), 15.3216524, 14.75, 14.5, 15.5, 16.0, 31.9,
         32.0, 100.0, 3e38, np.inf, -np.inf, np.nan, -0.3, -15.4, -0.07, 7.999999, 8.0, 0.24999999, 0.25]
    return np.array(e, dtype=np.float32)


def neighbourhoods(flips):
    """+-2 ULP around every flip point of the sweep (they cover all 11 binades already)."""
    f = np.array(sorted(set(flips)), dtype=np.int64)
    n = (f[:, None] + np.arange(-2, 3)[None, :]).reshape(-1).astype(np.uint32)
    return np.concatenate([n, n | np.uint32(0x80000000)])


def make_codec(report, rng):
    out = {}
    kat = np.array([0.01, 0.06251, 0.125, 0.1, 0.2, 1, 15], dtype=np.float32)  # sfp_quant.py:179
    out["kat_in"] = kat
    out["kat_act8"] = quantize_act(8)(torch.from_numpy(kat.copy())).numpy()
    for name, (fmt, fn, _) in FMTS.items():
        flips = report.get(name + "_flips")
        parts = [edge_inputs().view(np.uint32)]
        if flips:
            parts.append(neighbourhoods(flips))
        # log-uniform random magnitudes over [2^-8, 2^6) with random signs + raw random bit patterns
        mag = np.exp2(rng.uniform(-8, 6, 40000)).astype(np.float32) * rng.choice([-1.0, 1.0], 40000).astype(np.float32)
        parts.append(mag.view(np.uint32))
        parts.append(rng.integers(0, 1 << 32, 20000, dtype=np.uint64).astype(np.uint32))
        bits = np.concatenate(parts)
        r = ref_bits(fn, bits)
        c = so.quantize(bits.view(np.float32), 1.0, fmt).view(np.uint32)
        assert same_bits(r, c), f"oracle mismatch on fixture inputs for {name}"
        out[name + "_in_bits"] = bits
        out[name + "_out_bits"] = r
    # SFP<4,4> layer-output quantizer (sfp_quant.py:105-133): pin the oracle on every denormal, on a
    # full binade around 1 and 248, and on 2^14 samples of every other binade; then store a fixture
    lo = quantize_layerout(8)
    bad = 0
    starts = list(range(0, 0x00800000, 1 << 22)) + [0x3F800000, 0x43000000]
    for st in starts:
        bits = np.arange(st, st + (1 << 22 if st < 0x00800000 else 1 << 23), dtype=np.uint32)
        bad += int(not same_bits(ref_bits(lo, bits), so.layerout(bits.view(np.float32)).view(np.uint32)))
    for e in range(1, 255):
        bits = (np.uint32(e << 23) + rng.integers(0, 1 << 23, 1 << 14, dtype=np.uint64).astype(np.uint32))
        bits = np.concatenate([bits, bits | np.uint32(0x80000000)])
        bad += int(not same_bits(ref_bits(lo, bits), so.layerout(bits.view(np.float32)).view(np.uint32)))
    assert bad == 0, "oracle layerout does not match the reference"
    mids = (np.arange(0x3F800000, 0x43800000, 1 << 19, dtype=np.int64)[:, None] + 0x40000 + np.arange(-2, 3)[None, :]).reshape(-1)
    lbits = np.concatenate([edge_inputs().view(np.uint32), mids.astype(np.uint32), (mids.astype(np.uint32) | np.uint32(0x80000000)),
                            np.arange(1, 300, dtype=np.uint32), rng.integers(0, 1 << 32, 30000, dtype=np.uint64).astype(np.uint32),
                            (np.exp2(rng.uniform(-12, 9, 30000)).astype(np.float32) * rng.choice([-1.0, 1.0], 30000).astype(np.float32)).view(np.uint32)])
    out["layerout_in_bits"] = lbits
    out["layerout_out_bits"] = ref_bits(lo, lbits)
    assert same_bits(out["layerout_out_bits"], so.layerout(lbits.view(np.float32)).view(np.uint32))
    # scaled division: the operator divides by a float64 0-dim tensor (conv2d_func.py:21-22)
    ka64 = np.array([2.6023073196411133, 13.16812801361084, 1.7093303203582764, 9.842595100402832]) / 15.5
    xs = (rng.standard_normal(30000) * 3.0).astype(np.float32)
    out["div_scales_f64"] = ka64
    out["div_in"] = xs
    for i, k in enumerate(ka64):
        K = torch.tensor(k)  # float64 0-dim, as Conv2d_Q.__init__ makes it (conv2d_func.py:17-18)
        for name in ("act8", "w8", "act7"):
            fmt, fn, _ = FMTS[name]
            r = fn(torch.from_numpy(xs.copy()) / K).numpy().view(np.uint32)
            c = so.quantize(xs, np.float32(k), fmt).view(np.uint32)
            assert same_bits(r, c), f"float32(K) division does not reproduce the reference ({name}, K={k})"
            out[f"div{i}_{name}_out_bits"] = r
    return out


# (name, N, C, H, W, O, k, stride, pad, groups, bias, relu_input)
CONV_CASES = [
    ("dw3_s1", 2, 32, 12, 12, 32, 3, 1, 1, 32, False, True),
    ("dw3_s2_odd", 2, 64, 13, 13, 64, 3, 2, 1, 64, False, True),
    ("dw3_c24", 1, 24, 9, 9, 24, 3, 1, 1, 24, False, True),
    ("dw3_c58_s2", 1, 58, 10, 10, 58, 3, 2, 1, 58, False, True),
    ("dw3_c512_7", 2, 512, 7, 7, 512, 3, 1, 1, 512, False, True),
    ("pw_32_64", 2, 32, 8, 8, 64, 1, 1, 0, 1, False, True),
    ("pw_128_256", 1, 128, 7, 7, 256, 1, 1, 0, 1, False, True),
    ("pw_512_512", 1, 512, 4, 4, 512, 1, 1, 0, 1, False, True),
    ("pw_24_58", 1, 24, 6, 6, 58, 1, 1, 0, 1, False, True),
    ("pw_58_58", 1, 58, 5, 5, 58, 1, 1, 0, 1, False, True),
    ("pw_s2_down", 1, 64, 8, 8, 128, 1, 2, 0, 1, False, True),
    ("pw_bias", 1, 16, 6, 6, 8, 1, 1, 0, 1, True, True),
    ("stem3_s2", 2, 3, 16, 16, 32, 3, 2, 1, 1, False, False),
    ("dense3_bias", 1, 16, 10, 10, 32, 3, 1, 1, 1, True, True),
    ("dense3_64", 1, 64, 6, 6, 64, 3, 1, 1, 1, False, True),
    ("stem7_s2_p3", 1, 3, 20, 20, 16, 7, 2, 3, 1, False, False),
    ("stem7_s2_p0_bias", 1, 3, 21, 21, 8, 7, 2, 0, 1, True, False),
    ("alex11_s4_bias", 1, 3, 35, 35, 8, 11, 4, 2, 1, True, False),
    ("alex5_p2_bias", 1, 8, 9, 9, 16, 5, 1, 2, 1, True, True),
]
CONV_SCALES = [(2.6023073196411133 / 15.5, 1.9635683298110962 / 15.5),
               (6.629735469818115 / 15.5, 0.5438900589942932 / 15.5),
               (1.7093303203582764 / 15.5, 0.21044661104679108 / 15.5)]


def gen_case(idx, case):
    """Deterministic inputs for one conv case: the repo's own generator (numpy PCG64)."""
    name, N, C, H, W, O, k, s, p, g, has_bias, relu = case
    rng = np.random.default_rng(1000 + idx)
    Ka, Kw = CONV_SCALES[idx % len(CONV_SCALES)]
    x = rng.standard_normal((N, C, H, W)).astype(np.float32) * np.float32(6.0 * Ka)
    if relu:
        x = np.maximum(x, 0)  # post-ReLU-like: ~50 % exact zeros
    w = rng.standard_normal((O, C // g, k, k)).astype(np.float32) * np.float32(5.0 * Kw)
    w[rng.random(w.shape) < 0.02] = 0.0  # a few exactly-zero (pruned) weights
    b = (rng.standard_normal(O).astype(np.float32) * np.float32(0.5)) if has_bias else None
    return x, w, b, Ka, Kw


def make_conv():
    out = {}
    names = []
    worst = 0.0
    for idx, case in enumerate(CONV_CASES):
        name, N, C, H, W, O, k, s, p, g, has_bias, relu = case
        x, w, b, Ka, Kw = gen_case(idx, case)
        for q in (8, 7, 32):
            if q == 32 and idx % 5:
                continue
            factory = conv2d_Q_bias if has_bias else conv2d_Q
            Conv = factory(q_bit=q, Kw=np.float64(Kw), Ka=np.float64(Ka))
            m = Conv(C, O, k, np.float64(Kw), np.float64(Ka), s, p, groups=g, bias=has_bias).eval()
            with torch.no_grad():
                m.weight.copy_(torch.from_numpy(w))
                if has_bias:
                    m.bias.copy_(torch.from_numpy(b))
                y = m(torch.from_numpy(x.copy())).numpy()
                xq, wq = m.input_q.numpy(), m.weight_q.numpy()
            # pin the oracle + the torch port on this case
            yo, xqo, wqo = so.conv2d(x, w, b, s, p, 1, g, Ka, Kw, q, want_q=True)
            assert same_bits(xq.view(np.uint32), xqo.view(np.uint32)), (name, q, "input_q")
            assert same_bits(wq.view(np.uint32), wqo.view(np.uint32)), (name, q, "weight_q")
            err = float(np.abs(yo - y).max() / np.abs(y).max())
            worst = max(worst, err)
            assert err < 2e-6, (name, q, err)
            yt, _, _ = tp.conv2d_q(torch.from_numpy(x.copy()), torch.from_numpy(w), None if b is None else torch.from_numpy(b),
                                   s, p, 1, g, np.float64(Ka), np.float64(Kw), q)
            assert np.array_equal(yt.numpy(), y), (name, q, "torch_port differs from the reference")
            key = f"{name}_q{q}"
            names.append(key)
            out[key + "_y"] = y
            if q == 8:
                out[name + "_x"] = x
                out[name + "_w"] = w
                if has_bias:
                    out[name + "_b"] = b
                out[name + "_meta"] = np.array([N, C, H, W, O, k, s, p, g, int(has_bias)], dtype=np.int64)
                out[name + "_scales"] = np.array([Ka, Kw], dtype=np.float64)
            if q != 32 and H * W * C * N <= 8192:
                out[key + "_xq"] = xq
    # Linear_Q (conv2d_func.py:50-66)
    rng = np.random.default_rng(77)
    Ka, Kw = CONV_SCALES[0]
    xl = np.maximum(rng.standard_normal((4, 64)).astype(np.float32) * np.float32(6 * Ka), 0)
    wl = rng.standard_normal((10, 64)).astype(np.float32) * np.float32(5 * Kw)
    bl = rng.standard_normal(10).astype(np.float32)
    out["linear_x"], out["linear_w"], out["linear_b"] = xl, wl, bl
    out["linear_scales"] = np.array([Ka, Kw])
    for q in (8, 7):
        Lin = linear_Q(q_bit=q, Kw=np.float64(Kw), Ka=np.float64(Ka))
        m = Lin(64, 10).eval()
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(wl))
            m.bias.copy_(torch.from_numpy(bl))
            y = m(torch.from_numpy(xl.copy())).numpy()
        yo = so.linear(xl, wl, bl, Ka, Kw, q)
        assert np.abs(yo - y).max() / np.abs(y).max() < 2e-6
        out[f"linear_q{q}_y"] = y
    out["case_keys"] = np.array(names)
    print(f"conv cases: {len(names)}; worst oracle-vs-reference max-rel error {worst:.2e}")
    return out


def make_net():
    """BASELINE config 1: the reference's CIFAR MobileNetV1 (nets_cifar/mobilenetv1.py), batch 8,
    Qbits 8 and 7, deterministic parameters -> golden logits (+ the first block's activations)."""
    import types
    sys.modules.setdefault("torchsummary", types.ModuleType("torchsummary"))
    sys.modules["torchsummary"].summary = lambda *a, **k: None
    from nets_cifar.mobilenetv1 import MobileNetV1_Q  # the reference net, unmodified
    sys.path.insert(2, HERE)
    import netgen
    out = {}
    x = netgen.net_input()
    for q in (8, 7, 32):
        m = MobileNetV1_Q(ch_in=3, qbit=q).eval()
        netgen.fill_parameters(m)
        with torch.no_grad():
            logits = m(x.clone())
            h = m.model[0](x.clone())
        out[f"logits_q{q}"] = logits.numpy()
        out[f"block0_q{q}"] = h.numpy()
        print(f"net q={q}: logits range [{float(logits.min()):.3f}, {float(logits.max()):.3f}], "
              f"finite={bool(torch.isfinite(logits).all())}, top1={logits.argmax(1).tolist()}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-sweep", action="store_true")
    args = ap.parse_args()
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    rep_path = os.path.join(HERE, "sweep_report.json")
    if args.skip_sweep and os.path.exists(rep_path):
        report = json.load(open(rep_path))
    else:
        report = sweep()
        report["_torch"] = torch.__version__
        json.dump(report, open(rep_path, "w"))
    np.savez_compressed(os.path.join(HERE, "codec_golden.npz"), **make_codec(report, rng))
    np.savez_compressed(os.path.join(HERE, "conv_golden.npz"), **make_conv())
    np.savez_compressed(os.path.join(HERE, "net_golden.npz"), **make_net())
    for f in ("codec_golden.npz", "conv_golden.npz", "net_golden.npz", "sweep_report.json"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
