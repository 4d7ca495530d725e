"""torch-CPU port of the reference hot path -- TEST INFRASTRUCTURE, not product code.

The reference computes its fake-quantizers as a chain of ~25 full-tensor float32 ATen
passes and then calls F.conv2d (utils/sfp_quant.py:10-48, :59-97; utils/conv2d_func.py:20-25,
:41-47).  This module re-states that chain with the same ATen primitives so that
  * bench.py can time "the reference's CPU path" on the GPU box's host cores
    (cpu_baseline.kind == "port"; the reference's Python itself cannot travel), and
  * tests have a second, independent checker beside oracle/slfp_oracle.c.
tests/golden/make_golden.py proves it bit-identical to the imported reference on the
codec and on conv outputs (same ATen kernels, same order).
"""
import torch
import torch.nn.functional as F

_TINY = 1e-10


def _split(x):
    """sign, |x|, floor(log2|x|) and the mantissa in [1,2) (sfp_quant.py:17-22 / :34-39)."""
    sgn = torch.sign(x)
    mag = torch.abs(x)
    expo = torch.floor(torch.log2(mag))
    mant = mag / torch.pow(2, expo)
    return sgn, mag, expo, mant


def _finish(sgn, mag, out_mag, top, top_inclusive):
    """The ordered masked overrides and the sign product (sfp_quant.py:26-30 / :43-47)."""
    out_mag[mag < 0.0625] = _TINY
    out_mag[(mag >= 0.0625) & (mag < 0.125)] = 0.125
    if top_inclusive:
        out_mag[mag >= top] = top
    else:
        out_mag[mag > top] = top
    return torch.mul(sgn, out_mag)


def fake_quant(x, bits, kind):
    """kind 'act' -> quantize_act(bits).forward, 'weight' -> quantize_weight(bits).forward."""
    if bits == 32:
        return x
    sgn, mag, expo, mant = _split(x)
    if bits == 7:  # SFP<3,3>, identical for weights and activations
        mant_q = torch.round(mant * 8) / 8
        return _finish(sgn, mag, torch.mul(mant_q, torch.pow(2, expo)), 15, True)
    if bits != 8:
        raise ValueError("bits must be 32, 8 or 7")
    if kind == "act":  # linear RNE first, then the log converter (sfp_quant.py:88-89)
        mant = torch.round(mant * 16) / 16
    mant_log = torch.round(torch.log2(mant) * 16) / 16
    return _finish(sgn, mag, torch.pow(2, expo + mant_log), 15.32165, False)


def conv2d_q(x, weight, bias, stride, padding, dilation, groups, Ka, Kw, bits):
    """Conv2d_Q.forward: returns (output, input_q, weight_q).  Ka/Kw are 0-dim float64
    tensors exactly as the reference module holds them (conv2d_func.py:17-18)."""
    Ka = torch.as_tensor(Ka, dtype=torch.float64)
    Kw = torch.as_tensor(Kw, dtype=torch.float64)
    xq = fake_quant(x / Ka, bits, "act")
    wq = fake_quant(weight / Kw, bits, "weight")
    bq = None if bias is None else bias / Ka / Kw
    out = F.conv2d(xq, wq, bq, stride, padding, dilation, groups) * Ka * Kw
    return out, xq, wq


def linear_q(x, weight, bias, Ka, Kw, bits):
    """Linear_Q.forward (conv2d_func.py:60-65)."""
    Ka = torch.as_tensor(Ka, dtype=torch.float64)
    Kw = torch.as_tensor(Kw, dtype=torch.float64)
    xq = fake_quant(x / Ka, bits, "act")
    wq = fake_quant(weight / Kw, bits, "weight")
    bq = None if bias is None else bias / Kw / Ka
    return F.linear(xq, wq, bq) * Kw * Ka


def layerout(x):
    """quantize_layerout(k <= 8).forward (utils/sfp_quant.py:112-126), the reference's op sequence with its
    quirks: `2^(-8)` / `2^(-7)` are integer XORs there (= -6 / -5), so the two "subnormal" overrides never
    fire, only the `>= 248 -> 248` clamp is live and exact zeros come out as NaN (0 * inf)."""
    mag = x.abs()
    e = torch.floor(torch.log2(mag))
    scale = torch.pow(2, e)
    out = torch.round(mag / scale * 16) / 16 * scale
    out = torch.where(mag >= 248, torch.full_like(out, 248.0), out)
    return torch.sign(x) * out
