"""Conv2d_Q layer tables of the reference nets (data/layer_specs.json).

The reference's model files cannot travel to the GPU box, so the shapes, strides, groups,
bias flags and per-layer calibration scales (Ka, Kw) of every Conv2d_Q / Linear_Q layer
were extracted once from the imported reference (tests/golden/make_layer_specs.py) and are
kept here as data.  bench.py and the tests build their workloads from these tables.
"""
import json
import os
from dataclasses import dataclass

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "layer_specs.json")
_cache = None


@dataclass(frozen=True)
class ConvSpec:
    c_in: int
    c_out: int
    k: tuple
    stride: tuple
    pad: tuple
    groups: int
    bias: bool
    h: int
    w: int
    h_out: int
    w_out: int
    Ka: float
    Kw: float

    @property
    def in_elems(self):
        return self.c_in * self.h * self.w

    @property
    def out_elems(self):
        return self.c_out * self.h_out * self.w_out

    @property
    def w_elems(self):
        return self.c_out * (self.c_in // self.groups) * self.k[0] * self.k[1]

    @property
    def macs(self):
        """Multiply-accumulates per image."""
        return self.h_out * self.w_out * self.c_out * (self.c_in // self.groups) * self.k[0] * self.k[1]

    def algorithmic_bytes(self, batch):
        """fp32 activations read once, fp32 outputs written once, fp32 weights
        read once per batch."""
        return 4 * batch * (self.in_elems + self.out_elems) + 4 * self.w_elems


def nets():
    global _cache
    if _cache is None:
        _cache = json.load(open(_PATH))
    return _cache


def conv_layers(net):
    """List of ConvSpec for `net` (a key of data/layer_specs.json), in execution order."""
    rows = nets()[net]["layers"]
    out = []
    for r in rows:
        if r["kind"] != "conv":
            continue
        out.append(ConvSpec(c_in=r["c_in"], c_out=r["c_out"], k=tuple(r["k"]), stride=tuple(r["stride"]),
                            pad=tuple(r["pad"]), groups=r["groups"], bias=r["bias"], h=r["h"], w=r["w"],
                            h_out=r["h_out"], w_out=r["w_out"], Ka=r["Ka"], Kw=r["Kw"]))
    return out


def algorithmic_bytes_per_image(net, batch):
    """bytes/img = sum 4*(in+out) + (1/B) * sum 4*w   (BASELINE.md section 3)."""
    ls = conv_layers(net)
    return sum(4 * (l.in_elems + l.out_elems) for l in ls) + sum(4 * l.w_elems for l in ls) / batch
