"""CPU: the C-ABI library loads and exports every symbol include/slfp.h declares, the
host-only entry points (descriptor validation, kernel selection, sizes) behave, and the
Python mirror of the reference operator API keeps the reference's surface.  No compute
call is made here (no GPU in this tier)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import ROOT
from cnns_slfp_quantization_amd import _lib


def _desc(**kw):
    base = dict(n=2, c_in=32, h=16, w=16, c_out=32, kh=3, kw=3, stride_h=1, stride_w=1, pad_h=1, pad_w=1,
                dil_h=1, dil_w=1, groups=32, x_layout=_lib.LAYOUT_NHWC, y_layout=_lib.LAYOUT_NHWC, qbits=8,
                ka=0.17, kw_scale=0.12, mfma_passes=0, reserved=0)
    base.update(kw)
    return _lib.ConvDesc(**base)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "slfp.h")).read()
    declared = set(re.findall(r"\b(slfp_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.load()
    for name in declared:
        assert hasattr(L, name), f"libslfp_hip.so does not export {name}"
    assert L.slfp_version() == 1
    assert ctypes.sizeof(_lib.ConvDesc) == 7 * 8 + 14 * 4


def test_kernel_selection_and_sizes():
    L = _lib.load()
    name = lambda d: L.slfp_conv2d_kernel_name(ctypes.byref(d)).decode()
    assert name(_desc()) == "dw3x3_nhwc"
    assert name(_desc(stride_h=2, stride_w=2)) == "dw3x3_nhwc"
    assert name(_desc(c_in=58, c_out=58, groups=58)) == "dw3x3_nhwc"           # ShuffleNetV2 width 58: 8-byte lanes
    odd = _desc(c_in=57, c_out=57, groups=57)                                    # odd width:
    assert name(odd) == "repad+dw3x3_nhwc"                                       # same kernel on channel-padded copies
    assert L.slfp_conv2d_wprep_bytes(ctypes.byref(odd)) == 2304                  # 9*60*4 rounded up to 256
    assert L.slfp_conv2d_workspace_bytes(ctypes.byref(odd)) == 2 * (2 * 16 * 16 * 60 * 4) + 3 * 256
    assert name(_desc(c_in=58, c_out=58, groups=1, kh=1, kw=1, pad_h=0, pad_w=0)) == "pw_mfma_f16x1"   # even widths: 8-byte accesses
    assert name(_desc(c_in=57, c_out=58, groups=1, kh=1, kw=1, pad_h=0, pad_w=0)) == "repad+pw_mfma_f16x1"  # odd: padded copies
    assert name(_desc(c_in=58, c_out=1026, groups=1, kh=1, kw=1, pad_h=0, pad_w=0)) == "repad+pw_mfma_f16x1"  # W too big for LDS
    assert name(_desc(c_in=58, c_out=58, groups=2)) == "direct_nhwc"             # grouped, not depthwise
    pw = _desc(kh=1, kw=1, pad_h=0, pad_w=0, groups=1, c_in=128, c_out=256)
    assert name(pw) == "pw_mfma_f16x1"                                           # default: one fp16 pass
    pw.mfma_passes = _lib.MFMA_F16X3
    assert name(pw) == "pw_mfma_f16x3"
    pw.qbits = 7
    assert name(pw) == "pw_mfma_f16_exact"                                       # SFP<3,3> is exact in fp16
    assert name(_desc(c_in=3, c_out=32, groups=1, stride_h=2, stride_w=2)) == "stem_nhwc"   # MobileNetV1 stem: specialised fp32 kernel
    stem = _desc(c_in=3, c_out=64, groups=1)
    assert name(stem) == "stem_small_mfma_f16x1"                                 # K = 27: one MFMA k-step
    assert L.slfp_conv2d_wprep_bytes(ctypes.byref(stem)) == 4 * 1024            # 4 channel tiles x 1 KiB fragment
    stem.mfma_passes = _lib.MFMA_F16X3
    assert name(stem) == "stem_nhwc"                                             # float32-equivalent mode: fp32 stem kernel
    dense = _desc(c_in=16, c_out=32, groups=1)
    assert name(dense) == "dense_mfma_f16x1"                                     # dense k x k: implicit GEMM on MFMA
    assert L.slfp_conv2d_wprep_bytes(ctypes.byref(dense)) == 9 * 64 * 32 * 2     # [tap][C_in pad 64][C_out pad 16] fp16
    dense.mfma_passes = _lib.MFMA_F16X3
    assert name(dense) == "dense_mfma_f16x3"                                     # float32-equivalent mode: hi + lo planes
    assert L.slfp_conv2d_wprep_bytes(ctypes.byref(dense)) == 2 * 9 * 64 * 32 * 2
    assert L.slfp_conv2d_workspace_bytes(ctypes.byref(dense)) == 256 + 2 * (2 * 16 * 16 * 64 * 2)   # zero page + 2 fp16 planes, C padded to whole 64-channel chunks
    dense.stride_h = dense.stride_w = 2
    assert name(dense) == "direct_nhwc"                                          # stride-2 halo tiles do not fit twice
    dense.qbits = 7
    assert name(dense) == "dense_mfma_f16_exact"
    assert name(_desc(c_in=8, c_out=32, groups=1)) == "direct_nhwc"               # too few channels for a k-step
    big_stem = _desc(c_in=3, c_out=64, groups=1, kh=7, kw=7, stride_h=2, stride_w=2, pad_h=3, pad_w=3, h=64, w=64)
    assert name(big_stem) == "stem_mfma_f16x1"                                   # 7x7 image stem: im2row + MFMA
    assert L.slfp_conv2d_wprep_bytes(ctypes.byref(big_stem)) == 7 * 1 * 4 * 1024  # [kh][32-k step][4 channel tiles][1 KiB]
    assert L.slfp_conv2d_workspace_bytes(ctypes.byref(big_stem)) == 2 * 64 * 32 * 32 * 2  # N*H*Wo*Rp fp16
    big_stem.mfma_passes = _lib.MFMA_F16X3
    assert name(big_stem) == "stem_nhwc"
    assert name(_desc(c_in=16, c_out=32, groups=1, dil_h=2, dil_w=2)) == "direct_nhwc"
    ho, wo = ctypes.c_int64(), ctypes.c_int64()
    assert L.slfp_conv2d_out_shape(ctypes.byref(_desc(h=224, w=224, c_in=3, c_out=32, groups=1, stride_h=2, stride_w=2)),
                                   ctypes.byref(ho), ctypes.byref(wo)) == 0
    assert (ho.value, wo.value) == (112, 112)
    assert L.slfp_conv2d_wprep_bytes(ctypes.byref(_desc())) == 1280  # 9*32*4 rounded up to 256
    assert L.slfp_conv2d_wprep_bytes(ctypes.byref(pw)) == 2 * 128 * 256 * 2
    assert L.slfp_conv2d_workspace_bytes(ctypes.byref(_desc())) == 0
    assert L.slfp_conv2d_workspace_bytes(ctypes.byref(_desc(x_layout=0, y_layout=0))) == 2 * 2 * 32 * 16 * 16 * 4


@pytest.mark.parametrize("bad, code", [
    (dict(groups=3), _lib.ERR_SHAPE), (dict(qbits=6), _lib.ERR_BAD_ARG), (dict(qbits=32), _lib.ERR_BAD_ARG),
    (dict(ka=0.0), _lib.ERR_BAD_ARG), (dict(h=1, w=1, pad_h=0, pad_w=0), _lib.ERR_SHAPE),
    (dict(x_layout=7), _lib.ERR_BAD_ARG), (dict(stride_h=0), _lib.ERR_SHAPE), (dict(mfma_passes=2), _lib.ERR_BAD_ARG),
    (dict(ka=1e-35), _lib.ERR_UNSUPPORTED), (dict(kw_scale=float("inf")), _lib.ERR_UNSUPPORTED),
])
def test_bad_descriptors_return_status_not_crash(bad, code):
    L = _lib.load()
    d = _desc(**bad)
    assert L.slfp_conv2d_out_shape(ctypes.byref(d), None, None) == code
    assert _lib.last_error() != ""
    assert L.slfp_conv2d_wprep_bytes(ctypes.byref(d)) == 0
    with pytest.raises(_lib.SlfpError):
        _lib.check(code)
    # null pointers are rejected before anything touches a device
    assert L.slfp_conv2d_fwd(ctypes.byref(_desc()), None, None, None, None, None, None, None) == _lib.ERR_BAD_ARG
    assert L.slfp_quantize_f32(None, None, 16, 1.0, 0, None) == _lib.ERR_BAD_ARG
    assert L.slfp_quantize_f32(None, None, 0, 1.0, 0, None) == 0  # empty input is a no-op


def test_operator_surface_matches_reference():
    import utils.conv2d_func as cf  # the drop-in shim
    ns = {}
    exec("from utils.sfp_quant import *\nfrom utils.activation_func import *\nfrom utils.conv2d_func import *", ns)
    for name in ("torch", "nn", "F", "np", "conv2d_Q", "conv2d_Q_bias", "linear_Q", "quantize_weight", "quantize_act",
                 "quantize_layerout", "weight_quantize_func", "act_quantize_func", "layerout_quantize_func",
                 "STL", "Swish", "Sigmoid"):
        assert name in ns, name
    Ka, Kw = np.array([2.64, 2.60]) / 15.5, np.array([0.86, 1.96]) / 15.5
    Conv2d = cf.conv2d_Q(q_bit=8, Kw=Kw, Ka=Ka)               # class defaults are whole arrays, as in the nets
    m = Conv2d(32, 32, 3, Kw[1], Ka[1], 2, 1, groups=32, bias=False)  # Kw, Ka positional 4 and 5
    assert isinstance(m, nn.Conv2d) and m.stride == (2, 2) and m.padding == (1, 1) and m.groups == 32
    assert list(m.state_dict().keys()) == ["weight"]
    assert m.Ka.dtype == torch.float64 and m.Ka.dim() == 0 and float(m.Ka) == Ka[1]
    assert isinstance(m.quantize_weight, cf.weight_quantize_func) and isinstance(m.quantize_act, cf.act_quantize_func)
    assert m.q_bit == 8 and m.input_q is None and m.weight_q is None
    mb = cf.conv2d_Q_bias(8, Kw[0], Ka[0])(3, 8, 7, stride=2)
    assert sorted(mb.state_dict().keys()) == ["bias", "weight"]
    lin = cf.linear_Q(8, Kw[0], Ka[0])(64, 10)
    assert isinstance(lin, nn.Linear) and sorted(lin.state_dict().keys()) == ["bias", "weight"]
    # non-strict state-dict loading as the harness does (cifar100_train_eval.py:158-159)
    m.load_state_dict({"weight": torch.zeros_like(m.weight), "extra": torch.zeros(1)}, strict=False)
    with pytest.raises(AssertionError):
        cf.weight_quantize_func(16)  # utils/sfp_quant.py:138


def test_no_cpu_compute_path_and_qbit32_passthrough():
    import utils.conv2d_func as cf
    from oracle import torch_port as tp
    x = torch.randn(2, 8, 6, 6)
    m8 = cf.conv2d_Q(8, 0.1, 0.2)(8, 16, 3, padding=1)
    with pytest.raises(RuntimeError, match="ROCm"):
        m8(x)  # q_bit 8 on a CPU tensor: loud failure, never a silent fallback
    with pytest.raises(RuntimeError, match="ROCm"):
        cf.quantize_act(8)(x)
    m32 = cf.conv2d_Q_bias(32, 0.1, 0.2)(8, 16, 3, padding=1).eval()
    with torch.no_grad():
        y = m32(x)
        ref, xq, wq = tp.conv2d_q(x, m32.weight, m32.bias, 1, 1, 1, 1, 0.2, 0.1, 32)
    assert torch.allclose(y, ref.float(), rtol=1e-6, atol=1e-6)
    assert torch.equal(m32.input_q, x / m32.Ka) and m32.output is y
    assert cf.quantize_act(32)(x) is x and cf.layerout_quantize_func(32)(x) is x


def test_layerout_restatement_keeps_reference_quirks_and_product_has_no_cpu_path():
    from oracle import torch_port
    from oracle import slfp_oracle as so
    from utils.sfp_quant import layerout_quantize_func
    x = torch.tensor([0.3, 1.03, 300.0, -500.0, 0.0])
    y = torch_port.layerout(x)                       # the reference's op sequence (test infrastructure)
    assert y[0].item() == pytest.approx(0.296875) and y[1].item() == pytest.approx(1.0)
    assert y[2].item() == 248.0 and y[3].item() == -248.0
    assert torch.isnan(y[4])  # exact zero -> NaN in the reference too (2^(-8) is XOR there)
    yo = so.layerout(x.numpy())                      # the integer restatement agrees
    assert np.array_equal(np.isnan(yo), np.isnan(y.numpy())) and np.array_equal(yo[:4], y.numpy()[:4])
    with pytest.raises(RuntimeError):                # the product refuses CPU tensors (no CPU compute path)
        layerout_quantize_func(8)(x)


@pytest.mark.skipif(not os.path.isdir("/root/reference/nets_cifar"), reason="reference not mounted (GPU box)")
def test_reference_nets_run_unmodified_on_the_shim():
    """In the build container: the reference's own net files import `utils.*` from THIS repo (path
    order) and construct/run unmodified.  CPU has no quantized compute path, so the forward is
    checked with q_bit=32 (the passthrough) against the reference running on its own utils."""
    import subprocess
    import sys
    code = r'''
import sys, types, warnings
warnings.filterwarnings("ignore")
first, second = sys.argv[1], sys.argv[2]
sys.path.insert(0, first); sys.path.insert(1, second)
sys.modules["torchsummary"] = types.ModuleType("torchsummary"); sys.modules["torchsummary"].summary = lambda *a, **k: None
import torch, numpy as np
import utils.conv2d_func as cf
from nets_cifar.mobilenetv1 import MobileNetV1_Q
from nets_imgnet.squeezenet1_0 import SqueezeNet
torch.manual_seed(0)
m = MobileNetV1_Q(ch_in=3, qbit=32).eval()
x = torch.randn(2, 3, 32, 32)
with torch.no_grad():
    y = m(x)
q8 = MobileNetV1_Q(ch_in=3, qbit=8)       # constructs (27 Conv2d_Q + Linear_Q) with the array-valued defaults
sq = SqueezeNet(qbit=7)                   # conv2d_Q_bias users
n = sum(1 for mod in q8.modules() if isinstance(mod, torch.nn.Conv2d))
print("FILE", cf.__file__); print("N", n); print("SUM", float(y.double().abs().sum()))
'''
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    outs = []
    for first, second in ((ROOT, "/root/reference"), ("/root/reference", ROOT)):
        r = subprocess.run([sys.executable, "-c", code, first, second], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(dict(l.split(" ", 1) for l in r.stdout.strip().splitlines()))
    assert outs[0]["FILE"].startswith(ROOT) and outs[1]["FILE"].startswith("/root/reference")
    assert outs[0]["N"] == outs[1]["N"] == "27"
    assert abs(float(outs[0]["SUM"]) - float(outs[1]["SUM"])) <= 1e-4 * abs(float(outs[1]["SUM"]))


def test_fuse_bn_relu_host_logic():
    """fusion.fuse_bn_relu on the q_bit=32 passthrough (CPU): BN/ReLU folded into the conv module's
    post-op, modules replaced by Identity, unfuse restores them."""
    import utils.conv2d_func as cf
    from cnns_slfp_quantization_amd import fusion
    C = cf.conv2d_Q(32, 0.1, 0.2)
    Cb = cf.conv2d_Q_bias(32, 0.1, 0.2)
    m = nn.Sequential(C(8, 16, 3, 0.1, 0.2, 1, 1), nn.BatchNorm2d(16), nn.ReLU(inplace=True),
                      Cb(16, 16, 1, 0.1, 0.2), nn.BatchNorm2d(16),
                      C(16, 6, 1, 0.1, 0.2), nn.BatchNorm2d(6), nn.ReLU()).eval()   # 6 channels: any width fuses
    for b in (m[1], m[4], m[6]):
        b.running_mean.normal_(); b.running_var.uniform_(0.5, 1.5); b.weight.data.uniform_(0.5, 1.5); b.bias.data.normal_()
    x = torch.randn(2, 8, 6, 6)
    with torch.no_grad():
        y0 = m(x)
        assert fusion.fuse_bn_relu(m) == 3
        assert [type(c).__name__ for c in m] == ["Conv2d_Q", "Identity", "Identity", "Conv2d_Q", "Identity", "Conv2d_Q", "Identity", "Identity"]
        assert m[0]._post[2] == 1 and m[3]._post[2] == 0   # flags: SLFP_POST_RELU
        y1 = m(x)
        assert torch.allclose(y0, y1, rtol=1e-5, atol=1e-5)
        assert fusion.unfuse(m) == 3 and isinstance(m[1], nn.BatchNorm2d) and isinstance(m[6], nn.BatchNorm2d) and m[0]._post is None
        assert torch.equal(m(x), y0)
    m.train()
    with pytest.raises(RuntimeError):
        fusion.fold_bn(m[1])  # training-mode BN cannot be folded


def test_round2_host_logic_on_cpu():
    """Host-side pieces added in round 2 that need no GPU: the depthwise+pointwise pairing (q_bit 32 passthrough blocks run
    their two convs), unfuse() undoing it, the calibration files' text format (cifar100_train_eval.py:287-301), the
    hipGraph wrapper refusing CPU inputs and training mode, and the plan key helper rejecting vector scales."""
    import utils.conv2d_func as cf
    from cnns_slfp_quantization_amd import calibration, fusion
    from cnns_slfp_quantization_amd.conv2d_func import _scale_key
    from cnns_slfp_quantization_amd.graph import GraphedModule
    C = cf.conv2d_Q(32, 0.1, 0.2)
    m = nn.Sequential(C(8, 8, 3, 0.1, 0.2, 1, 1, groups=8), nn.BatchNorm2d(8), nn.ReLU(inplace=True),
                      C(8, 16, 1, 0.1, 0.2), nn.BatchNorm2d(16), nn.ReLU(inplace=True),
                      C(16, 16, 3, 0.1, 0.2, 2, 1, groups=16), nn.BatchNorm2d(16), nn.ReLU(inplace=True),
                      C(16, 4, 1, 0.1, 0.2), nn.BatchNorm2d(4)).eval()
    for b in m:
        if isinstance(b, nn.BatchNorm2d):
            b.running_mean.normal_(); b.running_var.uniform_(0.5, 1.5); b.weight.data.uniform_(0.5, 1.5); b.bias.data.normal_()
    x = torch.randn(2, 8, 9, 9)
    with torch.no_grad():
        y0 = m(x)
        assert fusion.fuse_bn_relu(m, dw_pw=True) == 4
        blocks = [b for b in m if isinstance(b, fusion.DwPwBlock)]
        assert len(blocks) == 2
        y1 = m(x)
        assert torch.allclose(y0, y1, rtol=1e-5, atol=1e-5)
        assert all(b._last_kernel is None for b in blocks)            # CPU / q_bit 32: the pair ran as its two convs
        assert fusion.unfuse(m) == 4 and not any(isinstance(b, fusion.DwPwBlock) for b in m)
        assert torch.equal(m(x), y0)
    files = calibration.scale_files_text("mobilenetv1", {1: 2.5, 2: 3.0}, {28: 9.75}, {1: 0.5})
    assert files["max_inout_mobilenetv1.txt"] == ("Layer 1 Max Absolute Input:\n2.5\n\nLayer 2 Max Absolute Input:\n3.0\n\n"
                                                   "Layer 28 Max Absolute Output:\n9.75\n\n")
    assert files["max_weight_mobilenetv1.txt"] == "Layer 1 Max Absolute weight:\n0.5\n\n"
    assert calibration.scales_from_max({2: 31.0, 1: 15.5}) == [1.0, 2.0]
    g = GraphedModule(m)
    with pytest.raises(RuntimeError):
        g(x)                                                          # CPU tensor: needs a ROCm device
    assert _scale_key(torch.tensor(0.25), "Ka") == 0.25 and _scale_key(0.5, "Kw") == 0.5
    with pytest.raises(ValueError):
        _scale_key(torch.tensor([0.1, 0.2]), "Ka")


def test_fuse_named_bn_on_hand_wired_blocks():
    """fusion.fuse_named_bn: a torchvision-style residual block (conv1/bn1/relu, conv2/bn2, + identity, relu: the wiring of
    nets_imgnet/resnet50.py:24-100) folds its BatchNorms by name; with an example input it checks the WIRING (bn<k> must be
    fed conv<k>'s output tensor) and leaves a pair alone when the naming convention lies (here: a block whose bn2 is
    applied BEFORE conv2); an optional output tolerance rolls everything back when exceeded."""
    import utils.conv2d_func as cf
    from cnns_slfp_quantization_amd import fusion
    C = cf.conv2d_Q(32, 0.1, 0.2)

    class Block(nn.Module):
        def __init__(self, honest=True):
            super().__init__()
            self.conv1 = C(8, 8, 3, 0.1, 0.2, 1, 1); self.bn1 = nn.BatchNorm2d(8)
            self.conv2 = C(8, 8, 1, 0.1, 0.2); self.bn2 = nn.BatchNorm2d(8)
            self.relu = nn.ReLU()
            self.honest = honest

        def forward(self, x):
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out)) if self.honest else self.conv2(self.bn2(out))
            return self.relu(out + x)

    x = torch.randn(2, 8, 6, 6)
    for honest in (True, False):
        m = nn.Sequential(Block(honest), Block(honest)).eval()
        for b in m.modules():
            if isinstance(b, nn.BatchNorm2d):
                b.running_mean.normal_(); b.running_var.uniform_(0.5, 1.5); b.weight.data.uniform_(0.5, 1.5); b.bias.data.normal_()
        with torch.no_grad():
            y0 = m(x)
            if honest:
                assert fusion.fuse_named_bn(m, x) == 4
                assert isinstance(m[0].bn1, nn.Identity) and m[0].conv1._post is not None and m[0].conv1._post[2] == 0
                assert torch.allclose(m(x), y0, rtol=1e-5, atol=1e-5)
                assert fusion.unfuse_named_bn(m) == 4 and isinstance(m[1].bn2, nn.BatchNorm2d)
                with pytest.raises(RuntimeError):
                    fusion.fuse_named_bn(m, x, rtol=1e-12)   # folding changes the rounding by ~1e-7: an impossible bar rolls back
                assert isinstance(m[0].bn1, nn.BatchNorm2d) and m[0].conv1._post is None
            else:
                assert fusion.fuse_named_bn(m, x) == 2     # conv1/bn1 of each block; the mis-named conv2/bn2 pairs are left alone
                assert isinstance(m[0].bn2, nn.BatchNorm2d) and m[0].conv2._post is None and isinstance(m[0].bn1, nn.Identity)
                assert torch.allclose(m(x), y0, rtol=1e-5, atol=1e-5)
                assert fusion.unfuse_named_bn(m) == 2
            assert torch.equal(m(x), y0)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__ (build / smoke) and bench.py's cpu_baseline leg
    may import it.  A product path that routed through it would void every parity claim."""
    import glob
    offenders = []
    for path in glob.glob(os.path.join(ROOT, "cnns_slfp_quantization_amd", "*.py")) + glob.glob(os.path.join(ROOT, "utils", "*.py")):
        for line in open(path):
            if re.match(r"\s*(from|import)\s+oracle\b", line):
                offenders.append((path, line.strip()))
    assert not offenders, offenders
    bench = open(os.path.join(ROOT, "bench.py")).read()
    # bench.py: exactly one import, inside cpu_baseline()
    assert bench.count("from oracle") == 1 and bench.split("from oracle")[0].rsplit("\ndef ", 1)[-1].startswith("cpu_baseline(")


def test_no_getenv_on_the_launch_path():
    """experiment switches are read once at load (csrc/codec.hip: read_switches), never per launch.
    The only other getenv sits inside the SLFP_PW_STAMPS diagnostic build."""
    csrc = os.path.join(ROOT, "cnns_slfp_quantization_amd", "csrc")
    hits = []
    for f in sorted(os.listdir(csrc)):
        text = open(os.path.join(csrc, f)).read()
        allowed = (0, 0)
        if f == "codec.hip":
            allowed = (text.index("static Switches read_switches"), text.index("static Switches g_switches"))
        for m in re.finditer(r"getenv\s*\(", text):
            if allowed[0] < m.start() < allowed[1]:
                continue
            if "SLFP_PW_STAMPS" in text[max(0, m.start() - 200):m.start()] or text[:m.start()].rsplit("\n", 1)[-1].lstrip().startswith("//"):
                continue   # the diagnostic build, or a comment that documents the rule
            hits.append((f, text[:m.start()].count("\n") + 1))
    assert not hits, hits
