"""Shared by tests/golden/make_golden.py (reference side) and the tests (this repo's side):
deterministic parameters for the CIFAR MobileNetV1 whole-net check, and a builder of the same
topology (nets_cifar/mobilenetv1.py:43-64) out of ANY Conv2d_Q / Linear_Q factories, with the
reference's state-dict keys (model.<i>.<j>.weight, fc.weight, ...).  Test infrastructure."""
import numpy as np
import torch
import torch.nn as nn

# (c_in, c_out, stride) of the 13 depthwise-separable blocks (nets_cifar/mobilenetv1.py:44-57)
BLOCKS = [(32, 64, 1), (64, 128, 2), (128, 128, 1), (128, 256, 2), (256, 256, 1), (256, 512, 2),
          (512, 512, 1), (512, 512, 1), (512, 512, 1), (512, 512, 1), (512, 512, 1), (512, 1024, 2), (1024, 1024, 1)]


def build_mobilenetv1_cifar(conv2d_Q, linear_Q, qbit, scales):
    """scales: list of (Ka, Kw) for the 27 convs + the fc (from data/layer_specs.json)."""
    it = iter(scales)

    def conv(inp, oup, k, stride, pad, groups=1):
        Ka, Kw = next(it)
        return conv2d_Q(q_bit=qbit, Kw=np.float64(Kw), Ka=np.float64(Ka))(inp, oup, k, np.float64(Kw), np.float64(Ka),
                                                                     stride, pad, groups=groups, bias=False)

    layers = [nn.Sequential(conv(3, 32, 3, 2, 1), nn.BatchNorm2d(32), nn.ReLU(inplace=True))]
    for inp, oup, s in BLOCKS:
        layers.append(nn.Sequential(conv(inp, inp, 3, s, 1, groups=inp), nn.BatchNorm2d(inp), nn.ReLU(inplace=True),
                                    conv(inp, oup, 1, 1, 0), nn.BatchNorm2d(oup), nn.ReLU(inplace=True)))
    layers.append(nn.AdaptiveAvgPool2d(1))
    Ka, Kw = next(it)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = nn.Sequential(*layers)
            self.fc = linear_Q(q_bit=qbit, Kw=np.float64(Kw), Ka=np.float64(Ka))(1024, 100)

        def forward(self, x):
            x = self.model(x)
            return self.fc(x.reshape(-1, 1024))

    return Net()


def fill_parameters(model, seed=2024):
    """Deterministic, activation-friendly parameters written through the state dict (same
    values for the reference net and for this repo's net: identical keys)."""
    rng = np.random.default_rng(seed)
    sd = model.state_dict()
    new = {}
    for k, v in sd.items():
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            new[k] = v
        elif k.endswith("running_mean"):
            new[k] = torch.from_numpy(rng.normal(0.0, 0.05, shape).astype(np.float32))
        elif k.endswith("running_var"):
            new[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shape).astype(np.float32))
        elif v.dim() == 1 and ".weight" in k:  # BN gamma
            new[k] = torch.from_numpy(rng.uniform(0.8, 1.6, shape).astype(np.float32))
        elif v.dim() == 1:  # BN beta / fc bias
            new[k] = torch.from_numpy(rng.normal(0.1, 0.1, shape).astype(np.float32))
        else:  # conv / fc weights: He-like so that activations neither die nor explode
            fan_in = int(np.prod(shape[1:]))
            new[k] = torch.from_numpy((rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32))
    model.load_state_dict(new)
    return model


def net_input(batch=8, seed=99):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal((batch, 3, 32, 32)).astype(np.float32))


def build_mobilenetv1_imagenet(conv2d_Q, qbit, scales):
    """nets_imgnet/mobilenetv1.py:24-61 out of ANY Conv2d_Q factory: conv_bn + 13 conv_dw blocks, AvgPool2d(7), a plain
    nn.Linear(1024, 1000) head; state-dict keys as the reference's (model.<i>.<j>.weight, fc.weight).
    scales: list of (Ka, Kw) for the 27 convs (data/layer_specs.json: mobilenetv1_imagenet224)."""
    it = iter(scales)

    def conv(inp, oup, k, stride, pad, groups=1):
        Ka, Kw = next(it)
        return conv2d_Q(q_bit=qbit, Kw=np.float64(Kw), Ka=np.float64(Ka))(inp, oup, k, np.float64(Kw), np.float64(Ka),
                                                                     stride, pad, groups=groups, bias=False)

    layers = [nn.Sequential(conv(3, 32, 3, 2, 1), nn.BatchNorm2d(32), nn.ReLU(inplace=True))]
    for inp, oup, s in BLOCKS:
        layers.append(nn.Sequential(conv(inp, inp, 3, s, 1, groups=inp), nn.BatchNorm2d(inp), nn.ReLU(inplace=True),
                                    conv(inp, oup, 1, 1, 0), nn.BatchNorm2d(oup), nn.ReLU(inplace=True)))
    layers.append(nn.AvgPool2d(7))

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = nn.Sequential(*layers)
            self.fc = nn.Linear(1024, 1000)

        def forward(self, x):
            x = self.model(x)
            return self.fc(x.view(-1, 1024))

    return Net()


def net_input224(batch, seed=199):
    """Seeded images with per-image contrast, per-channel offsets and a smooth per-image pattern: with random weights a
    plain N(0,1) image barely moves the logits (every image lands on the same class), which would make top-k
    agreement meaningless."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((batch, 3, 224, 224)).astype(np.float32)
    scale = rng.uniform(0.2, 3.0, (batch, 1, 1, 1)).astype(np.float32)
    offs = rng.uniform(-2.0, 2.0, (batch, 3, 1, 1)).astype(np.float32)
    yy, xx = np.meshgrid(np.linspace(-1, 1, 224, dtype=np.float32), np.linspace(-1, 1, 224, dtype=np.float32), indexing="ij")
    fx = rng.uniform(0.5, 6.0, (batch, 3, 1, 1)).astype(np.float32)
    fy = rng.uniform(0.5, 6.0, (batch, 3, 1, 1)).astype(np.float32)
    pattern = 1.5 * np.sin(fx * xx[None, None] * np.pi + fy * yy[None, None] * np.pi)
    return torch.from_numpy((x * scale + offs + pattern).astype(np.float32))


class StashNet(nn.Module):
    """The calibration protocol of the reference's CIFAR nets (nets_cifar/mobilenetv1.py:66-171) around a
    build_mobilenetv1_cifar net: after every forward, layer_inputs[i] / layer_weights[i] hold the i-th quantized
    layer's input_q / weight_q stash (27 convs, then the fc as 27) and layer_outputs[27] the logits."""

    def __init__(self, net):
        super().__init__()
        self.net = net
        self.model = net.model
        self.fc = net.fc
        self.layer_inputs, self.layer_outputs, self.layer_weights = {}, {}, {}

    def get_layer_inputs(self):
        return self.layer_inputs

    def get_layer_outputs(self):
        return self.layer_outputs

    def get_layer_weights(self):
        return self.layer_weights

    def reset_layer_inputs_outputs(self):
        self.layer_inputs, self.layer_outputs = {}, {}

    def reset_layer_weights(self):
        self.layer_weights = {}

    def forward(self, x):
        y = self.net(x)
        convs = [self.model[0][0]] + [m for blk in list(self.model)[1:14] for m in (blk[0], blk[3])]
        for i, m in enumerate(convs + [self.fc]):
            self.layer_inputs[i] = m.input_q
            self.layer_weights[i] = m.weight_q
        self.layer_outputs[27] = y
        return y


def calibrate_bn_(model, x):
    """Give a random-weight net the activation statistics of a trained one: one forward of `x` (identity
    quantizers, q_bit 32) during which every BatchNorm2d takes the mean / variance of ITS OWN input as running
    statistics.  Without this the deterministic weights drive every deep layer into the SLFP clamp and the logits
    stop depending on the image.  Returns {bn index: (mean, var)} (stored in the fixture, so that both sides
    load bit-identical parameters)."""
    bns = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
    hooks = []

    def make(bn):
        def pre(mod, args):
            h = args[0].detach()
            mod.running_mean.copy_(h.mean(dim=(0, 2, 3)))
            mod.running_var.copy_(h.var(dim=(0, 2, 3), unbiased=False) + 1e-3)
        return pre

    for bn in bns:
        hooks.append(bn.register_forward_pre_hook(make(bn)))
    with torch.no_grad():
        model.eval()(x)
    for h in hooks:
        h.remove()
    return {i: (bn.running_mean.clone().numpy(), bn.running_var.clone().numpy()) for i, bn in enumerate(bns)}


def load_bn_stats_(model, stats):
    bns = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
    with torch.no_grad():
        for i, bn in enumerate(bns):
            bn.running_mean.copy_(torch.from_numpy(np.asarray(stats[f"bn_mean_{i}"])))
            bn.running_var.copy_(torch.from_numpy(np.asarray(stats[f"bn_var_{i}"])))
    return model
