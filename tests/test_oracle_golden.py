"""CPU: the oracle (C restatement + torch port) against the committed golden vectors that
tests/golden/make_golden.py produced from the imported reference."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_errors, same_bits
from oracle import slfp_oracle as so
from oracle import torch_port as tp

FMT = {"act8": so.FMT_ACT8, "w8": so.FMT_W8, "act7": so.FMT_SFP7, "w7": so.FMT_SFP7}
TPA = {"act8": (8, "act"), "w8": (8, "weight"), "act7": (7, "act"), "w7": (7, "weight")}


def test_sweep_report_pins_the_oracle():
    rep = json.load(open(os.path.join(GOLDEN, "sweep_report.json")))
    for name in FMT:
        assert rep[name]["patterns"] == 92274688  # every float32 in [2^-6, 2^5)
        assert rep[name]["oracle_mismatch"] == 0 and rep[name]["torch_port_mismatch"] == 0
    # distinct outputs: 7 binades x 16 (8) mantissas + specials
    assert rep["w8"]["n_flip_points"] == 113 and rep["act8"]["n_flip_points"] == 106 and rep["act7"]["n_flip_points"] == 56


def test_reference_kat(codec_golden):
    # the only known-answer vector the reference itself carries (utils/sfp_quant.py:177-182)
    y = so.quantize(codec_golden["kat_in"], 1.0, so.FMT_ACT8)
    assert same_bits(y, codec_golden["kat_act8"])
    np.testing.assert_allclose(y, [1e-10, 0.125, 0.125, 0.125, 0.20131129, 1.0, 15.3216524], rtol=1e-7)


@pytest.mark.parametrize("name", list(FMT))
def test_codec_golden_c_oracle(codec_golden, name):
    x = codec_golden[name + "_in_bits"].view(np.float32)
    assert same_bits(so.quantize(x, 1.0, FMT[name]), codec_golden[name + "_out_bits"])


@pytest.mark.parametrize("name", list(FMT))
def test_codec_golden_torch_port(codec_golden, name):
    x = torch.from_numpy(codec_golden[name + "_in_bits"].view(np.float32).copy())
    y = tp.fake_quant(x, *TPA[name]).numpy()
    assert same_bits(y, codec_golden[name + "_out_bits"])


def test_scaled_division_is_float32_division(codec_golden):
    x = codec_golden["div_in"]
    for i, k in enumerate(codec_golden["div_scales_f64"]):
        for name in ("act8", "w8", "act7"):
            assert same_bits(so.quantize(x, np.float32(k), FMT[name]), codec_golden[f"div{i}_{name}_out_bits"])


def test_reciprocal_multiply_is_not_bit_exact():
    """x * (1/Ka) flips codes at the quantizer thresholds (measured: ~2e-7 of N(0,3) inputs);
    this is why the kernels keep the IEEE division.  Build inputs that sit on thresholds."""
    k = np.float32(2.6023073196411133 / 15.5)
    t = np.float32(1.0) + np.arange(1, 32, 2, dtype=np.float32) / np.float32(32)   # act8 linear-RNE midpoints
    base = (t[None, :] * np.exp2(np.arange(-3, 4, dtype=np.float32))[:, None]).ravel()
    x0 = (base * k).astype(np.float32).view(np.uint32)
    x = (x0[:, None] + np.arange(-8, 9, dtype=np.int64)[None, :]).astype(np.uint32).ravel().view(np.float32)
    right = so.quantize(x, k, so.FMT_ACT8)
    wrong = so.quantize(x * (np.float32(1) / k), 1.0, so.FMT_ACT8)
    assert (wrong.view(np.uint32) != right.view(np.uint32)).any()


@pytest.mark.parametrize("fmt", [so.FMT_ACT8, so.FMT_W8, so.FMT_SFP7])
def test_code_roundtrip_and_idempotence(fmt):
    rng = np.random.default_rng(5)
    x = np.concatenate([np.exp2(rng.uniform(-9, 6, 200000)).astype(np.float32) * rng.choice([-1, 1], 200000).astype(np.float32),
                        np.array([0.0, -0.0, 1e-20, 15.32165, 15.3216524, 20.0, -20.0, np.inf], np.float32)])
    q = so.quantize(x, 1.0, fmt)
    # extended codes make decode(encode(x)) == quantize(x) bit for bit
    assert same_bits(so.decode(so.encode(x, 1.0, fmt | so.FMT_EXT), fmt | so.FMT_EXT), q)
    # canonical codes differ from it only on exact zeros and on the clamp literal
    d = so.decode(so.encode(x, 1.0, fmt), fmt)
    neq = d.view(np.uint32) != q.view(np.uint32)
    assert np.all((x[neq] == 0) | (np.abs(x[neq]) > 15.3))
    # code layout: sign | (E+4) | m  (utils/sfp_quant.py:95 "0 111 1111")
    c = so.encode(np.array([15.0, -15.0, 1.0, 0.125, 0.01], np.float32), 1.0, fmt)
    if fmt == so.FMT_SFP7:
        assert list(c) == [0x3F, 0x7F, 0x20, 0x08, 0x00]
    else:
        assert list(c) == [0x7F, 0xFF, 0x40, 0x10, 0x00]
    # idempotent except that the two spellings of the top code map to each other
    qq = so.quantize(q, 1.0, fmt)
    neq = qq.view(np.uint32) != q.view(np.uint32)
    assert np.all(np.abs(q[neq]) > 15.3)
    # monotone (non-decreasing) on positive inputs up to the 3-ULP wobble at the clamp
    xs = np.sort(np.abs(x[np.isfinite(x)]))
    qs = so.quantize(xs, 1.0, fmt)
    assert np.all(np.diff(qs.astype(np.float64)) >= -3e-6)  # 15.3216524 -> 15.3216496


def _cases(conv_golden):
    return [str(k) for k in conv_golden["case_keys"]]


def test_conv_golden_c_oracle_and_port(conv_golden):
    worst = 0.0
    for key in _cases(conv_golden):
        name, q = key.rsplit("_q", 1)
        q = int(q)
        N, C, H, W, O, k, s, p, g, has_b = [int(v) for v in conv_golden[name + "_meta"]]
        Ka, Kw = conv_golden[name + "_scales"]
        x, w = conv_golden[name + "_x"], conv_golden[name + "_w"]
        b = conv_golden[name + "_b"] if has_b else None
        y, xq, wq = so.conv2d(x, w, b, s, p, 1, g, Ka, Kw, q, want_q=True)
        ref = conv_golden[key + "_y"]
        emax, el2 = rel_errors(y, ref)
        worst = max(worst, emax)
        assert emax < 2e-6 and el2 < 2e-6, (key, emax, el2)
        if key + "_xq" in conv_golden.files:
            assert same_bits(xq, conv_golden[key + "_xq"]), key
        yt, _, _ = tp.conv2d_q(torch.from_numpy(x.copy()), torch.from_numpy(w), None if b is None else torch.from_numpy(b),
                               s, p, 1, g, np.float64(Ka), np.float64(Kw), q)
        assert rel_errors(yt.numpy(), ref)[0] < 1e-6, key  # same ATen kernels as the reference
    assert worst < 2e-6


def test_linear_golden(conv_golden):
    Ka, Kw = conv_golden["linear_scales"]
    for q in (8, 7):
        y = so.linear(conv_golden["linear_x"], conv_golden["linear_w"], conv_golden["linear_b"], Ka, Kw, q)
        assert rel_errors(y, conv_golden[f"linear_q{q}_y"])[0] < 2e-6


def test_oracle_rejects_bad_geometry():
    x = np.zeros((1, 6, 4, 4), np.float32)
    w = np.zeros((4, 2, 3, 3), np.float32)
    with pytest.raises((ValueError, AssertionError)):
        so.conv2d(x, w, None, 1, 1, 1, 4, 0.1, 0.1, 8)  # groups does not divide C_in


@pytest.mark.skipif(not os.path.isdir("/root/reference/utils"), reason="reference not mounted (GPU box)")
def test_live_reference_agrees_with_oracle():
    """In the build container the reference itself is importable: spot-check live."""
    import subprocess
    import sys
    code = (
        "import sys, warnings; warnings.filterwarnings('ignore'); sys.path.insert(0, '/root/reference'); sys.path.insert(1, %r)\n"
        "import numpy as np, torch\n"
        "from utils.sfp_quant import quantize_act, quantize_weight\n"
        "from oracle import slfp_oracle as so\n"
        "x = (torch.randn(1 << 18) * 5)\n"
        "for b, fa, fw in ((8, so.FMT_ACT8, so.FMT_W8), (7, so.FMT_SFP7, so.FMT_SFP7)):\n"
        "    assert np.array_equal(quantize_act(b)(x).numpy().view(np.uint32), so.quantize(x.numpy(), 1.0, fa).view(np.uint32))\n"
        "    assert np.array_equal(quantize_weight(b)(x).numpy().view(np.uint32), so.quantize(x.numpy(), 1.0, fw).view(np.uint32))\n"
        "print('ok')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_layerout_golden(codec_golden):
    """SFP<4,4> layer-output quantizer (utils/sfp_quant.py:105-133) incl. its quirks: the reference's
    2^(-8) is an XOR, so exact zeros come out NaN and tiny values are rounded, not flushed."""
    x = codec_golden["layerout_in_bits"].view(np.float32)
    assert same_bits(so.layerout(x), codec_golden["layerout_out_bits"])
    y = so.layerout(np.array([0.3, 1.03, 300.0, -500.0, 0.0, 247.9, 1e-40], np.float32))
    assert y[0] == np.float32(0.296875) and y[1] == 1.0 and y[2] == 248.0 and y[3] == -248.0 and np.isnan(y[4]) and y[5] == 248.0
    assert 0 < y[6] < 1.1e-40
