"""GPU (MI355X): the 1-byte inter-layer activation format (second half; include/slfp.h
slfp_conv2d_fwd_codes; csrc/slfp_codes.hpp), through the C ABI.

What the reference does between two quantized convolutions (nets_imgnet/mobilenetv1.py:24-33 + utils/conv2d_func.py:21):
    y = relu(bn(conv_q(x)));   input_q = quantize_act(y / Ka_next)
The code path stores slfp_encode_f32(y, Ka_next, fmt | EXT) in the producer and decodes in the consumer.  The bars:
  * PRODUCER: the bytes layer i writes are equal to slfp_encode_f32(float32 output of the fused layer i, Ka_{i+1}, ..);
  * CONSUMER: layer i+1's output from the bytes is BIT-IDENTICAL to its output from the float32 tensor;
  * the code tables are exhaustively checked on the device (all 2^32 inputs) against the long form.
The float32-interface kernels themselves are pinned to the oracle / the reference's golden vectors in test_gpu_parity.py.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import slfp_oracle as so

pytestmark = pytest.mark.gpu
SOAK = int(os.environ.get("SLFP_TEST_SOAK", "0") or 0)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def lib():
    from cnns_slfp_quantization_amd import _lib
    L = _lib.load()  # raises if libslfp_hip.so is missing: no fallback
    assert L.slfp_device_count() >= 1
    return _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _desc(lib, s, n, qbits):
    return lib.ConvDesc(n=n, c_in=s.c_in, h=s.h, w=s.w, c_out=s.c_out, kh=s.k[0], kw=s.k[1], stride_h=s.stride[0],
                        stride_w=s.stride[1], pad_h=s.pad[0], pad_w=s.pad[1], dil_h=1, dil_w=1, groups=s.groups,
                        x_layout=lib.LAYOUT_NHWC, y_layout=lib.LAYOUT_NHWC, qbits=qbits, ka=float(np.float32(s.Ka)),
                        kw_scale=float(np.float32(s.Kw)), mfma_passes=lib.MFMA_F16X1, reserved=0)


def _encode(lib, x, ka, fmt):
    c = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    lib.check(lib.load().slfp_encode_f32(x.data_ptr(), c.data_ptr(), x.numel(), float(np.float32(ka)), fmt | lib.FMT_EXT, _stream()))
    return c


class _Layer:
    """One Conv2d_Q layer of a layer table with random weights, folded-BN vectors and prepared weights."""

    def __init__(self, lib, s, n, qbits, dev, gen, post=True, relu=True):
        L = lib.load()
        self.s, self.n, self.qbits, self.relu = s, n, qbits, relu
        self.d = _desc(lib, s, n, qbits)
        fan = (s.c_in // s.groups) * s.k[0] * s.k[1]
        self.w = torch.randn((s.c_out, s.c_in // s.groups, s.k[0], s.k[1]), generator=gen, device=dev)
        self.w.mul_(min(5.0 * s.Kw, 3.0 * (2.0 / fan) ** 0.5 + 2.0 * s.Kw))
        self.blob = torch.empty(L.slfp_conv2d_wprep_bytes(ctypes.byref(self.d)), dtype=torch.uint8, device=dev)
        lib.check(L.slfp_conv2d_prepare_weights(ctypes.byref(self.d), self.w.data_ptr(), self.blob.data_ptr(), None, _stream()))
        self.scale = (torch.rand(s.c_out, generator=gen, device=dev) + 0.5) if post else None
        self.shift = (torch.randn(s.c_out, generator=gen, device=dev) * 0.3) if post else None
        self.kernel = L.slfp_conv2d_kernel_name(ctypes.byref(self.d)).decode()

    def out_shape(self):
        return (self.n, self.s.h_out, self.s.w_out, self.s.c_out)

    def fwd_f32(self, lib, x):
        """the float32 interface: slfp_conv2d_fwd_post"""
        L = lib.load()
        y = torch.empty(self.out_shape(), device=x.device)
        ws_n = L.slfp_conv2d_workspace_bytes(ctypes.byref(self.d))
        ws = torch.empty(ws_n, dtype=torch.uint8, device=x.device) if ws_n else None
        lib.check(L.slfp_conv2d_fwd_post(ctypes.byref(self.d), x.data_ptr(), self.blob.data_ptr(), None,
                                         self.scale.data_ptr() if self.scale is not None else None,
                                         self.shift.data_ptr() if self.shift is not None else None,
                                         1 if self.relu else 0, y.data_ptr(), None, ws.data_ptr() if ws is not None else None, _stream()))
        return y

    def fwd_codes(self, lib, x, x_codes, y_ka=None, y_qbits=8):
        """slfp_conv2d_fwd_codes: y_ka None -> float32 out, else the consumer's codes"""
        L = lib.load()
        io = lib.ConvIo(x_codes=1 if x_codes else 0, y_codes=0 if y_ka is None else 1,
                        y_ka=float(np.float32(y_ka if y_ka is not None else 1.0)), y_qbits=y_qbits)
        assert L.slfp_conv2d_codes_supported(ctypes.byref(self.d), ctypes.byref(io), 0, 1 if self.relu else 0) == 1, \
            (self.kernel, self.s, x_codes, y_ka)
        y = torch.empty(self.out_shape(), dtype=torch.float32 if y_ka is None else torch.uint8, device=x.device)
        ws_n = L.slfp_conv2d_workspace_bytes(ctypes.byref(self.d))
        if ws_n:   # dense k x k layers: the fp16 operand copy (slfp_conv2d_fwd_codes_ws)
            ws = torch.empty(ws_n, dtype=torch.uint8, device=x.device)
            lib.check(L.slfp_conv2d_fwd_codes_ws(ctypes.byref(self.d), ctypes.byref(io), x.data_ptr(), self.blob.data_ptr(), None,
                                                 self.scale.data_ptr() if self.scale is not None else None,
                                                 self.shift.data_ptr() if self.shift is not None else None,
                                                 1 if self.relu else 0, y.data_ptr(), ws.data_ptr(), _stream()))
            return y
        lib.check(L.slfp_conv2d_fwd_codes(ctypes.byref(self.d), ctypes.byref(io), x.data_ptr(), self.blob.data_ptr(), None,
                                          self.scale.data_ptr() if self.scale is not None else None,
                                          self.shift.data_ptr() if self.shift is not None else None,
                                          1 if self.relu else 0, y.data_ptr(), _stream()))
        return y


def _synthetic_input(s, n, dev, gen, signed=False):
    """post-ReLU-like activations spanning all binades and both clamps, with exact zeros (half of a ReLU's outputs)"""
    x = torch.randn((n, s.h, s.w, s.c_in), generator=gen, device=dev)
    if not signed:
        x = torch.relu(x)
    x = x * (6.0 * s.Ka)
    x.view(-1)[::97] = 17.0 * s.Ka          # beyond the clamp
    x.view(-1)[5::193] = 0.05 * s.Ka        # the "tiny" class
    return x


def test_code_tables_equal_long_form_for_all_2_32_inputs(lib, dev):
    """Producer table (signed and unsigned variants) vs the long form behind slfp_encode_f32, decode tables vs
    slfp_decode_f32: every float32 pattern, on the device."""
    from cnns_slfp_quantization_amd import layer_specs
    L = lib.load()
    scales = [1.0, 0.171, 15.5 / 3.0, 2.0 ** -7, 0.4277248690205236, 1e-3, 37.25]
    if SOAK:
        scales += sorted({float(np.float32(s.Ka)) for net in layer_specs.nets() for s in layer_specs.conv_layers(net)})
    out = torch.zeros(3, dtype=torch.int64, device=dev)
    for fmt in (lib.FMT_ACT8, lib.FMT_SFP7):
        for k in scales:
            lib.check(L.slfp_debug_code_mismatches(float(np.float32(k)), fmt, out.data_ptr(), _stream()))
            torch.cuda.synchronize()
            assert out.tolist() == [0, 0, 0], (fmt, k, out.tolist())


def _mobilenet_specs():
    from cnns_slfp_quantization_amd import layer_specs
    return layer_specs.conv_layers("mobilenetv1_imagenet224")


def _distinct(specs):
    seen, out = set(), []
    for i, s in enumerate(specs):
        key = (s.c_in, s.c_out, s.k, s.stride, s.h)
        if key not in seen:
            seen.add(key)
            out.append(i)
    return out


@pytest.mark.parametrize("qbits", [8, 7])
def test_each_mobilenet_layer_on_codes_is_bit_identical_to_the_float32_interface(lib, dev, qbits):
    """All 19 distinct MobileNetV1-224 layer shapes (nets_imgnet/mobilenetv1.py:43-57), small batch, both formats:
    consumer from codes == consumer from float32 (bit for bit); producer's bytes == slfp_encode_f32(float32 output)."""
    specs = _mobilenet_specs()
    gen = torch.Generator(device=dev).manual_seed(2024 + qbits)
    fmt = lib.FMT_ACT8 if qbits == 8 else lib.FMT_SFP7
    n = 3
    for i in _distinct(specs):
        s = specs[i]
        ka_next = specs[i + 1].Ka if i + 1 < len(specs) else 0.2345
        lay = _Layer(lib, s, n, qbits, dev, gen)
        x = _synthetic_input(s, n, dev, gen, signed=(s.c_in == 3))
        y_ref = lay.fwd_f32(lib, x)
        codes_ref = _encode(lib, y_ref, ka_next, fmt)
        if s.c_in == 3:
            if qbits != 8:
                continue   # the specialised stem kernel is SLFP<3,4> only; SFP<3,3> chains start after the stem
            yc = lay.fwd_codes(lib, x, False, ka_next, qbits)
            assert torch.equal(yc, codes_ref), ("stem", i)
            continue
        xc = _encode(lib, x, s.Ka, fmt)
        y = lay.fwd_codes(lib, xc, True)                       # codes in, float32 out
        assert torch.equal(y.view(torch.int32), y_ref.view(torch.int32)), (lay.kernel, i, "consumer", float((y - y_ref).abs().max()))
        yc = lay.fwd_codes(lib, xc, True, ka_next, qbits)      # codes in, codes out
        bad = int((yc != codes_ref).sum())
        assert bad == 0, (lay.kernel, i, "producer", bad, yc.numel())


@pytest.mark.parametrize("qbits", [8, 7])
def test_dense_layers_on_codes_are_bit_identical_to_the_float32_interface(lib, dev, qbits):
    """The dense k x k family (VGG-16 / ResNet-50 3x3, SqueezeNet expand3x3, a stride-2 and a 5x5 case) through
    slfp_conv2d_fwd_codes_ws: codes in -> the same float32 output bit for bit (the decode pre-pass builds the same fp16 operand
    copy as the encode pre-pass); float32 or codes in -> codes out == slfp_encode_f32(float32 output, Ka_next); with and
    without a ReLU in front of the output quantizer; a workspace is required."""
    from cnns_slfp_quantization_amd.layer_specs import ConvSpec
    gen = torch.Generator(device=dev).manual_seed(77 + qbits)
    fmt = lib.FMT_ACT8 if qbits == 8 else lib.FMT_SFP7
    L = lib.load()
    shapes = [  # c_in, c_out, k, stride, pad, h
        (64, 64, 3, 1, 1, 28), (128, 256, 3, 1, 1, 14), (256, 256, 3, 1, 1, 9), (64, 256, 3, 1, 1, 13),
        (128, 128, 3, 2, 1, 15), (48, 192, 5, 1, 2, 13), (512, 512, 3, 1, 1, 7), (96, 80, 3, 1, 1, 11)]
    for ci, co, k, st, pd, h in shapes:
        ho = (h + 2 * pd - k) // st + 1
        s = ConvSpec(c_in=ci, c_out=co, k=(k, k), stride=(st, st), pad=(pd, pd), groups=1, bias=False, h=h, w=h, h_out=ho, w_out=ho,
                     Ka=0.37, Kw=0.021)
        for relu in (True, False):
            lay = _Layer(lib, s, 3, qbits, dev, gen, post=True, relu=relu)
            assert lay.kernel.startswith("dense_mfma"), lay.kernel
            x = _synthetic_input(s, 3, dev, gen, signed=not relu)
            y_ref = lay.fwd_f32(lib, x)
            ka_next = 0.2345
            codes_ref = _encode(lib, y_ref, ka_next, fmt)
            xc = _encode(lib, x, s.Ka, fmt)
            y = lay.fwd_codes(lib, xc, True)
            assert torch.equal(y.view(torch.int32), y_ref.view(torch.int32)), (ci, co, k, st, relu, "consumer", float((y - y_ref).abs().max()))
            for src, is_codes in ((x, False), (xc, True)):
                yc = lay.fwd_codes(lib, src, is_codes, ka_next, qbits)
                bad = int((yc != codes_ref).sum())
                assert bad == 0, (ci, co, k, st, relu, "producer", is_codes, bad, yc.numel())
    # no workspace -> refused, nothing launched
    s = ConvSpec(c_in=64, c_out=64, k=(3, 3), stride=(1, 1), pad=(1, 1), groups=1, bias=False, h=8, w=8, h_out=8, w_out=8, Ka=0.37, Kw=0.021)
    lay = _Layer(lib, s, 2, qbits, dev, gen)
    io = lib.ConvIo(x_codes=1, y_codes=0, y_ka=1.0, y_qbits=8)
    xc = torch.zeros((2, 8, 8, 64), dtype=torch.uint8, device=dev)
    y = torch.empty((2, 8, 8, 64), device=dev)
    rc = L.slfp_conv2d_fwd_codes(ctypes.byref(lay.d), ctypes.byref(io), xc.data_ptr(), lay.blob.data_ptr(), None, None, None, 0,
                                 y.data_ptr(), _stream())
    assert rc == lib.ERR_BAD_ARG, rc
    # code output needs C_out % 16 == 0
    s = ConvSpec(c_in=64, c_out=72, k=(3, 3), stride=(1, 1), pad=(1, 1), groups=1, bias=False, h=8, w=8, h_out=8, w_out=8, Ka=0.37, Kw=0.021)
    d = _desc(lib, s, 2, qbits)
    io = lib.ConvIo(x_codes=1, y_codes=1, y_ka=0.5, y_qbits=qbits)
    assert L.slfp_conv2d_codes_supported(ctypes.byref(d), ctypes.byref(io), 0, 1) == 0
    io = lib.ConvIo(x_codes=1, y_codes=0, y_ka=1.0, y_qbits=qbits)
    assert L.slfp_conv2d_codes_supported(ctypes.byref(d), ctypes.byref(io), 0, 1) == 1


def test_resident_weight_dense_kernel_writes_the_same_codes_on_long_walks(lib, dev):
    """k_dense3x3_res with code output where every persistent workgroup walks over many tiles (>= 16 per walker; the small
    cases of the test above give each walker one): VGG-16's conv1_2 at batch 24 and conv2_1 (two channel slices) at batch 48;
    codes == slfp_encode_f32(float32 output), float32 and code input, with and without the ReLU."""
    from cnns_slfp_quantization_amd.layer_specs import ConvSpec
    gen = torch.Generator(device=dev).manual_seed(2025)
    for ci, co, h, n in ((64, 64, 224, 24), (64, 128, 112, 48)):
        s = ConvSpec(c_in=ci, c_out=co, k=(3, 3), stride=(1, 1), pad=(1, 1), groups=1, bias=False, h=h, w=h, h_out=h, w_out=h, Ka=0.37, Kw=0.021)
        for relu in (True, False):
            lay = _Layer(lib, s, n, 8, dev, gen, post=True, relu=relu)
            assert lay.kernel.startswith("dense_mfma"), lay.kernel
            x = _synthetic_input(s, n, dev, gen, signed=not relu)
            y_ref = lay.fwd_f32(lib, x)
            codes_ref = _encode(lib, y_ref, 0.2345, lib.FMT_ACT8)
            del y_ref
            xc = _encode(lib, x, s.Ka, lib.FMT_ACT8)
            for src, is_codes in ((x, False), (xc, True)):
                yc = lay.fwd_codes(lib, src, is_codes, 0.2345, 8)
                bad = int((yc != codes_ref).sum())
                assert bad == 0, (ci, co, h, n, relu, is_codes, bad, yc.numel())
                del yc


@pytest.mark.parametrize("qbits", [8, 7])
def test_small_k_stem_writes_the_next_layers_codes(lib, dev, qbits):
    """VGG-16's first layer (3x3 s1 3 -> 64 on the one-k-step MFMA stem, nets_cifar/vgg16.py:31): float32 image in, the
    consumer's codes out == slfp_encode_f32(float32 output of the fused layer, Ka_next); with and without a ReLU; ragged size."""
    from cnns_slfp_quantization_amd.layer_specs import ConvSpec
    gen = torch.Generator(device=dev).manual_seed(31 + qbits)
    fmt = lib.FMT_ACT8 if qbits == 8 else lib.FMT_SFP7
    L = lib.load()
    for h, st in ((37, 1), (32, 2)):
        ho = (h + 2 - 3) // st + 1
        s = ConvSpec(c_in=3, c_out=64, k=(3, 3), stride=(st, st), pad=(1, 1), groups=1, bias=False, h=h, w=h, h_out=ho, w_out=ho, Ka=0.17, Kw=0.05)
        for relu in (True, False):
            lay = _Layer(lib, s, 3, qbits, dev, gen, post=True, relu=relu)
            assert lay.kernel.startswith("stem_small_mfma"), lay.kernel
            x = _synthetic_input(s, 3, dev, gen, signed=True)
            y_ref = lay.fwd_f32(lib, x)
            yc = lay.fwd_codes(lib, x, False, 0.31, qbits)
            bad = int((yc != _encode(lib, y_ref, 0.31, fmt)).sum())
            assert bad == 0, (h, st, relu, bad, yc.numel())
    s = ConvSpec(c_in=3, c_out=24, k=(3, 3), stride=(1, 1), pad=(1, 1), groups=1, bias=False, h=16, w=16, h_out=16, w_out=16, Ka=0.17, Kw=0.05)
    io = lib.ConvIo(x_codes=0, y_codes=1, y_ka=0.31, y_qbits=qbits)
    assert L.slfp_conv2d_codes_supported(ctypes.byref(_desc(lib, s, 2, qbits)), ctypes.byref(io), 0, 1) == 0   # ShuffleNetV2's 24-channel stem: no


@pytest.mark.parametrize("qbits", [8, 7])
def test_maxpool_on_codes_equals_encoding_the_pooled_tensor(lib, dev, qbits):
    """slfp_maxpool2d_codes: pooling the codes == slfp_encode_f32(max_pool2d(float32 tensor)) bit for bit -- post-ReLU and
    signed tensors with every special class in them (exact zeros, the 1e-10 class, values beyond the clamp, whose Qbits-8
    literal is 3 ulp BELOW the top regular value), VGG-16's 2x2 / 2 pools, ResNet-50's 3x3 / 2 pad 1, 16- and 4-channel lanes."""
    import torch.nn.functional as F
    from cnns_slfp_quantization_amd.sfp_quant import hip_maxpool_codes
    fmt = lib.FMT_ACT8 if qbits == 8 else lib.FMT_SFP7
    gen = torch.Generator(device=dev).manual_seed(11 + qbits)
    ka = 0.23
    for (n, c, h, w), (k, st, pd), signed in (((3, 64, 30, 30), (2, 2, 0), False), ((2, 12, 17, 19), (2, 2, 0), True),
                                               ((2, 32, 23, 23), (3, 2, 1), False), ((2, 16, 9, 9), (3, 1, 1), True)):
        x = torch.randn((n, c, h, w), generator=gen, device=dev) * (5.0 * ka)
        x = x if signed else torch.relu(x)
        x.view(-1)[::7] = 15.4 * ka            # the top regular class
        x.view(-1)[3::11] = 40.0 * ka          # beyond the clamp
        x.view(-1)[5::13] = 0.03 * ka          # the 1e-10 class
        x.view(-1)[6::17] = 0.0
        if signed:
            x.view(-1)[8::19] = -40.0 * ka
        x = x.contiguous(memory_format=torch.channels_last)
        def enc(t):   # elementwise in memory order, keeping the channels_last strides
            t = t.contiguous(memory_format=torch.channels_last)
            c = torch.empty_like(t, dtype=torch.uint8)
            lib.check(lib.load().slfp_encode_f32(t.data_ptr(), c.data_ptr(), t.numel(), float(np.float32(ka)), fmt | lib.FMT_EXT, _stream()))
            return c
        want = enc(F.max_pool2d(x, k, st, pd))
        got = hip_maxpool_codes(enc(x), k, st, pd, qbits)
        assert got.shape == want.shape and torch.equal(got, want), ((n, c, h, w), (k, st, pd), signed, int((got != want).sum()))
    L = lib.load()
    z = torch.zeros((1, 6, 4, 4), dtype=torch.uint8, device=dev).contiguous(memory_format=torch.channels_last)
    assert L.slfp_maxpool2d_codes(z.data_ptr(), z.data_ptr(), 1, 4, 4, 6, 2, 2, 2, 2, 0, 0, qbits, _stream()) == lib.ERR_UNSUPPORTED


def test_dense_layer_on_codes_in_the_float32_equivalent_mode(lib, dev):
    """SLFP_MFMA_F16X3 on a dense layer: the decode pre-pass also writes the residual plane, so codes in / codes out stay
    bit-identical to the float32 interface in that mode too (the pointwise code kernels exist for the single-pass mode only)."""
    from cnns_slfp_quantization_amd.layer_specs import ConvSpec
    gen = torch.Generator(device=dev).manual_seed(909)
    s = ConvSpec(c_in=64, c_out=128, k=(3, 3), stride=(1, 1), pad=(1, 1), groups=1, bias=False, h=14, w=14, h_out=14, w_out=14, Ka=0.37, Kw=0.021)
    lay = _Layer(lib, s, 3, 8, dev, gen)
    lay.d.mfma_passes = lib.MFMA_F16X3
    L = lib.load()
    lay.blob = torch.empty(L.slfp_conv2d_wprep_bytes(ctypes.byref(lay.d)), dtype=torch.uint8, device=dev)
    lib.check(L.slfp_conv2d_prepare_weights(ctypes.byref(lay.d), lay.w.data_ptr(), lay.blob.data_ptr(), None, _stream()))
    assert L.slfp_conv2d_kernel_name(ctypes.byref(lay.d)).decode() == "dense_mfma_f16x3"
    x = _synthetic_input(s, 3, dev, gen)
    y_ref = lay.fwd_f32(lib, x)
    xc = _encode(lib, x, s.Ka, lib.FMT_ACT8)
    y = lay.fwd_codes(lib, xc, True)
    assert torch.equal(y.view(torch.int32), y_ref.view(torch.int32)), float((y - y_ref).abs().max())
    yc = lay.fwd_codes(lib, xc, True, 0.2345, 8)
    assert torch.equal(yc, _encode(lib, y_ref, 0.2345, lib.FMT_ACT8))


def test_codes_without_relu_carry_the_sign(lib, dev):
    """conv -> BN -> conv without a ReLU in between (ShuffleNetV2 branches): negative values keep their sign bit."""
    specs = _mobilenet_specs()
    gen = torch.Generator(device=dev).manual_seed(7)
    for qbits, fmt in ((8, lib.FMT_ACT8), (7, lib.FMT_SFP7)):
        for i in (5, 6, 13, 14):   # dw 128@56, pw 128->128@56, dw 512@14, pw 512->512@14
            s = specs[i]
            lay = _Layer(lib, s, 2, qbits, dev, gen, relu=False)
            lay.shift -= 0.2
            x = _synthetic_input(s, 2, dev, gen)
            xc = _encode(lib, x, s.Ka, fmt)
            y_ref = lay.fwd_f32(lib, x)
            assert float((y_ref < 0).float().mean()) > 0.05
            yc = lay.fwd_codes(lib, xc, True, specs[i + 1].Ka, qbits)
            assert torch.equal(yc, _encode(lib, y_ref, specs[i + 1].Ka, fmt)), (qbits, i)


def test_full_batch_chain_matches_the_float32_interface(lib, dev):
    """BASELINE config 2 size: batch 256 through all 27 layers as ONE chain of codes (stem float32 -> codes ... -> last
    pointwise codes -> float32), against the same layers on the float32 interface.  Images are independent, the chain is
    bit-identical layer by layer, so the final float32 tensors must be EQUAL; intermediate code tensors are compared with
    slfp_encode_f32 of the float32 chain's tensors."""
    specs = _mobilenet_specs()
    n = 256
    gen = torch.Generator(device=dev).manual_seed(99)
    layers = [_Layer(lib, s, n, 8, dev, gen) for s in specs]
    for lay in layers:   # keep activations in range: BN scale so that outputs span the next layer's code range
        lay.scale.mul_(1.0)
    x = torch.randn((n, 224, 224, 3), generator=gen, device=dev) * (4.0 * specs[0].Ka)
    # float32 interface
    a = x
    f32_codes_sample = {}
    for i, lay in enumerate(layers):
        a = lay.fwd_f32(lib, a)
        if i + 1 < len(layers) and i in (0, 1, 2, 7, 12, 13, 24, 25):
            f32_codes_sample[i] = _encode(lib, a, specs[i + 1].Ka, lib.FMT_ACT8)
    y_ref = a
    # code chain
    c = layers[0].fwd_codes(lib, x, False, specs[1].Ka, 8)
    for i in range(1, len(layers)):
        if i - 1 in f32_codes_sample:
            assert torch.equal(c, f32_codes_sample[i - 1]), ("codes after layer", i - 1)
        last = i == len(layers) - 1
        c = layers[i].fwd_codes(lib, c, True, None if last else specs[i + 1].Ka, 8)
    assert torch.equal(c.view(torch.int32), y_ref.view(torch.int32))
    assert float(y_ref.abs().max()) > 0 and bool(torch.isfinite(y_ref).all())


def test_codes_vs_oracle_direct(lib, dev):
    """One depthwise and one pointwise layer on codes against the CPU oracle itself (not only against the float32-interface
    kernels): decode with the oracle, convolve in double, compare under the per-family bars of test_gpu_parity.py."""
    specs = _mobilenet_specs()
    gen = torch.Generator(device=dev).manual_seed(5)
    for i, tol in ((7, 1e-5), (8, 1e-3)):   # dw 128@56 s2 (float32 FMA), pw 128->256@28 (single-pass fp16 MFMA)
        s = specs[i]
        lay = _Layer(lib, s, 2, 8, dev, gen, post=False, relu=False)
        x = _synthetic_input(s, 2, dev, gen)
        xc = _encode(lib, x, s.Ka, lib.FMT_ACT8)
        y = lay.fwd_codes(lib, xc, True).cpu().numpy().transpose(0, 3, 1, 2)
        xq = so.decode(xc.cpu().numpy(), so.FMT_ACT8 | so.FMT_EXT).transpose(0, 3, 1, 2)
        assert np.array_equal(xq.view(np.uint32), so.quantize(x.cpu().numpy().transpose(0, 3, 1, 2), np.float32(s.Ka), so.FMT_ACT8).view(np.uint32))
        ref = so.conv2d(np.ascontiguousarray(x.cpu().numpy().transpose(0, 3, 1, 2)), lay.w.cpu().numpy(), None, s.stride, s.pad, (1, 1),
                        s.groups, np.float64(s.Ka), np.float64(s.Kw), 8)
        d = np.abs(y.astype(np.float64) - ref)
        assert d.max() <= tol * np.abs(ref).max(), (i, d.max(), np.abs(ref).max())


def test_unsupported_combinations_are_refused(lib, dev):
    L = lib.load()
    specs = _mobilenet_specs()
    d = _desc(lib, specs[2], 2, 8)
    io = lib.ConvIo(x_codes=1, y_codes=1, y_ka=0.5, y_qbits=5)
    assert L.slfp_conv2d_codes_supported(ctypes.byref(d), ctypes.byref(io), 0, 1) == 0       # bad consumer format
    io = lib.ConvIo(x_codes=0, y_codes=1, y_ka=0.5, y_qbits=8)
    assert L.slfp_conv2d_codes_supported(ctypes.byref(d), ctypes.byref(io), 0, 1) == 0       # float32 -> codes pointwise: not built
    d3 = _desc(lib, specs[2], 2, 8)
    d3.mfma_passes = lib.MFMA_F16X3
    io = lib.ConvIo(x_codes=1, y_codes=0, y_ka=1.0, y_qbits=8)
    assert L.slfp_conv2d_codes_supported(ctypes.byref(d3), ctypes.byref(io), 0, 1) == 0      # three-pass mode: not built
    x = torch.zeros((2, 112, 112, 32), dtype=torch.uint8, device=dev)
    y = torch.empty((2, 112, 112, 64), device=dev)
    blob = torch.empty(L.slfp_conv2d_wprep_bytes(ctypes.byref(d3)), dtype=torch.uint8, device=dev)
    rc = L.slfp_conv2d_fwd_codes(ctypes.byref(d3), ctypes.byref(io), x.data_ptr(), blob.data_ptr(), None, None, None, 1, y.data_ptr(), _stream())
    assert rc == lib.ERR_UNSUPPORTED and "code-path" in lib.last_error()


# ------------------------------------------------------------------ host side: fusion.link_codes on the drop-in modules
def _mobilenet224(dev, qbits=8, batch=6):
    """nets_imgnet/mobilenetv1.py:24-61 from the drop-in modules with the parameters and BatchNorm statistics of the
    committed config-2 fixture (tests/golden/net224_golden.npz: generated by the imported reference)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import netgen
    import utils.conv2d_func as cf
    from cnns_slfp_quantization_amd import layer_specs
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "net224_golden.npz"))
    rows = [r for r in layer_specs.nets()["mobilenetv1_imagenet224"]["layers"] if r["kind"] == "conv"]
    scales = [(r["Ka"], r["Kw"]) for r in rows]
    net = netgen.load_bn_stats_(netgen.fill_parameters(netgen.build_mobilenetv1_imagenet(cf.conv2d_Q, qbits, scales)), gold)
    net = net.to(dev).eval().to(memory_format=torch.channels_last)
    x = netgen.net_input224(64)[:batch].to(dev).contiguous(memory_format=torch.channels_last)
    return net, x, gold


@pytest.mark.parametrize("qbits", [8, 7])
def test_link_codes_whole_net_is_bit_identical_to_the_fused_net(dev, qbits):
    """nets_imgnet/mobilenetv1.py:43-61 built from the drop-in modules: fuse_bn_relu, then link_codes.  All 26 hand-overs
    between the 27 convs become 1-byte codes (the SFP<3,3> stem has no code-output kernel: its link runs the float32 kernel
    plus one slfp_encode_f32 pass) and the logits do not change by a single bit; input_q read back inside the chain equals
    the fused net's."""
    from cnns_slfp_quantization_amd import fusion
    net, x, gold = _mobilenet224(dev, qbits, 16)
    with torch.no_grad():
        assert fusion.fuse_bn_relu(net) == 27
        y_fused = net(x)
        conv5 = net.model[3][0]
        xq_fused = conv5.input_q.clone()
        n = fusion.link_codes(net, example_input=x)
        assert n == 26, n
        y_codes = net(x)
        assert y_codes.dtype == torch.float32
        assert torch.equal(y_codes.view(torch.int32), y_fused.view(torch.int32))
        if qbits == 8:   # the reference's own logits for these images (config-2 fixture), same bars as the float32 interface
            G = gold["logits_q8"][:16]
            Lg = y_codes.cpu().numpy()
            d = np.abs(Lg.astype(np.float64) - G)
            assert d.max() <= 5e-2 * np.abs(G).max() and np.linalg.norm(d) <= 4e-2 * np.linalg.norm(G)
            assert float((Lg.argmax(1) == G.argmax(1)).mean()) >= 0.95
        assert "codes_in" in conv5._last_kernel and "codes_out" in conv5._last_kernel
        assert torch.equal(conv5.input_q.view(torch.int32), xq_fused.view(torch.int32))   # decode(codes) == QA(x / Ka)
        # optimistic linking (no example input) gives the same links and the same bits
        assert fusion.unlink_codes(net) == n
        assert fusion.link_codes(net) >= n
        assert torch.equal(net(x).view(torch.int32), y_fused.view(torch.int32))
        assert fusion.unlink_codes(net) > 0
        assert torch.equal(net(x).view(torch.int32), y_fused.view(torch.int32))
        # as one hipGraph
        fusion.link_codes(net, example_input=x)
        from cnns_slfp_quantization_amd.graph import GraphedModule
        g = GraphedModule(net)
        assert torch.equal(g(x).view(torch.int32), y_fused.view(torch.int32))
        assert torch.equal(g(x).view(torch.int32), y_fused.view(torch.int32))


def test_image_groups_on_two_streams_return_the_same_logits(dev):
    """streams.forward_image_groups: the batch as two independent halves on two HIP streams through the whole fused / linked
    MobileNetV1 == the single forward, bit for bit (images are independent units; the modules' plan / workspace caches are
    keyed by stream)."""
    from cnns_slfp_quantization_amd import fusion, streams
    net, x, gold = _mobilenet224(dev, 8, 16)
    with torch.no_grad():
        fusion.fuse_bn_relu(net)
        y1 = net(x)
        for g in (2, 3):
            yg = streams.forward_image_groups(net, x, groups=g)
            torch.cuda.synchronize()
            assert torch.equal(yg.view(torch.int32), y1.view(torch.int32)), g
        assert fusion.link_codes(net, x) == 26
        yc = streams.forward_image_groups(net, x, groups=2)
        torch.cuda.synchronize()
        assert torch.equal(yc.view(torch.int32), y1.view(torch.int32))


def test_linked_modules_refuse_training(dev):
    from cnns_slfp_quantization_amd import fusion
    net, x, _ = _mobilenet224(dev, 8)
    with torch.no_grad():
        fusion.fuse_bn_relu(net)
        fusion.link_codes(net)
    net.train()
    with pytest.raises(RuntimeError, match="inference-only"):
        net.model[1][0](torch.zeros((1, 32, 8, 8), device=dev))
