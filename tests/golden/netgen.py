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
