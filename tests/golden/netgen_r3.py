"""Round-3 test infrastructure shared by tests/golden/make_golden_r3.py (reference side, build container) and the GPU
tests: the topologies of BASELINE configs 3-5 built out of ANY Conv2d_Q / Linear_Q / layerout factories, with the
reference's state-dict key NAMES, so that name-seeded parameters are identical on both sides:

    ResNet50      nets_imgnet/resnet50.py:24-147      (Bottleneck wiring :76-100, torchvision naming)
    SqueezeNet    nets_imgnet/squeezenet1_0.py:21-95  (Fire concat :41-46)
    VGG16_Q       nets_cifar/vgg16.py:13-135          (the CIFAR topology; fully convolutional up to the adaptive pool)
    ShuffleNetV2  nets_cifar/shufflenet_v2.py:22-167, :312-320 (channel split / shuffle :22-46, units :48-118)

These are this repo's own definitions: layer geometry follows the cited lines, the per-layer calibration scales are NOT
re-derived here (the reference indexes its Ka / Kw tables in net-specific ways) but come from the fixture's manifest
{module name: (Ka, Kw)}, which make_golden_r3.py reads off the reference's own modules.  Data only travels."""
import zlib

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------ parameters seeded by NAME (order-independent)
def param_by_name(key, shape, seed):
    """The value fill_parameters_by_name gives state-dict entry `key`: a pure function of (key, shape, seed)."""
    rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
    if key.endswith("running_mean"):
        return rng.normal(0.0, 0.05, shape).astype(np.float32)
    if key.endswith("running_var"):
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if len(shape) == 1 and ".weight" in key:   # BN gamma
        return rng.uniform(0.8, 1.6, shape).astype(np.float32)
    if len(shape) == 1:                        # BN beta / conv and fc bias
        return rng.normal(0.1, 0.1, shape).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))           # conv / fc weights: He-like
    return (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)


def fill_parameters_by_name(model, seed=3030, weight_gain=None):
    """weight_gain: {state-dict key: factor} applied to conv weights (nets without BatchNorm: see make_golden_r3.lsuv_)."""
    new = {}
    for k, v in model.state_dict().items():
        if k.endswith("num_batches_tracked"):
            new[k] = v
            continue
        a = param_by_name(k, tuple(v.shape), seed)
        if weight_gain and k in weight_gain:
            a = (a * np.float32(weight_gain[k])).astype(np.float32)
        new[k] = torch.from_numpy(a)
    model.load_state_dict(new)
    return model


def load_bn_stats_by_name_(model, stats):
    """stats: {"bn:<module name>:mean" / ":var": array} (the fixture)."""
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.from_numpy(np.asarray(stats[f"bn:{name}:mean"])))
                m.running_var.copy_(torch.from_numpy(np.asarray(stats[f"bn:{name}:var"])))
    return model


class Factories:
    """conv(name, ...) / conv_bias(name, ...) / linear(name, ...) with the scales of manifest[name]."""

    def __init__(self, cf, qbit, manifest, layerout=None):
        self.cf, self.q, self.m, self.layerout_cls = cf, qbit, manifest, layerout
        self.prefix = []

    def _k(self, name):
        Ka, Kw = self.m[name]
        return np.float64(Ka), np.float64(Kw)

    def conv(self, name, cin, cout, k, stride=1, pad=0, groups=1):
        Ka, Kw = self._k(name)
        return self.cf.conv2d_Q(q_bit=self.q, Kw=Kw, Ka=Ka)(cin, cout, k, Kw, Ka, stride, pad, groups=groups, bias=False)

    def conv_bias(self, name, cin, cout, k, stride=1, pad=0):
        Ka, Kw = self._k(name)
        return self.cf.conv2d_Q_bias(q_bit=self.q, Kw=Kw, Ka=Ka)(cin, cout, k, Kw, Ka, stride, pad)

    def linear(self, name, cin, cout):
        Ka, Kw = self._k(name)
        return self.cf.linear_Q(q_bit=self.q, Kw=Kw, Ka=Ka)(cin, cout, Kw, Ka)

    def layerout(self):
        return self.layerout_cls(q_bit=self.q)


# ------------------------------------------------------------------ ResNet-50 (nets_imgnet/resnet50.py)
class _Bottleneck(nn.Module):
    def __init__(self, f, name, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = f.conv(name + ".conv1", inplanes, planes, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = f.conv(name + ".conv2", planes, planes, 3, stride, 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = f.conv(name + ".conv3", planes, planes * 4, 1)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU()
        self.downsample = downsample

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        identity = x if self.downsample is None else self.downsample(x)
        return self.relu(out + identity)


class ResNet50(nn.Module):
    def __init__(self, f, num_classes=1000):
        super().__init__()
        self.conv1 = f.conv("conv1", 3, 64, 7, 2, 3)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = 64
        for li, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), start=1):
            layers = []
            for b in range(blocks):
                name = f"layer{li}.{b}"
                ds = None
                if b == 0:
                    ds = nn.Sequential(f.conv(name + ".downsample.0", inplanes, planes * 4, 1, stride), nn.BatchNorm2d(planes * 4))
                layers.append(_Bottleneck(f, name, inplanes, planes, stride if b == 0 else 1, ds))
                inplanes = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*layers))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = f.linear("fc", 2048, num_classes)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


# ------------------------------------------------------------------ SqueezeNet 1.0 (nets_imgnet/squeezenet1_0.py)
class _Fire(nn.Module):
    def __init__(self, f, name, inplanes, s, e1, e3):
        super().__init__()
        self.squeeze = f.conv_bias(name + ".squeeze", inplanes, s, 1)
        self.squeeze_activation = nn.ReLU(inplace=True)
        self.expand1x1 = f.conv_bias(name + ".expand1x1", s, e1, 1)
        self.expand1x1_activation = nn.ReLU(inplace=True)
        self.expand3x3 = f.conv_bias(name + ".expand3x3", s, e3, 3, 1, 1)
        self.expand3x3_activation = nn.ReLU(inplace=True)

    def forward(self, x):
        x = self.squeeze_activation(self.squeeze(x))
        return torch.cat([self.expand1x1_activation(self.expand1x1(x)), self.expand3x3_activation(self.expand3x3(x))], 1)


class SqueezeNet(nn.Module):
    def __init__(self, f, num_classes=1000):
        super().__init__()
        fire = lambda i, *a: _Fire(f, f"features.{i}", *a)  # noqa: E731
        self.features = nn.Sequential(
            f.conv_bias("features.0", 3, 96, 7, 2, 0), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, ceil_mode=True),
            fire(3, 96, 16, 64, 64), fire(4, 128, 16, 64, 64), fire(5, 128, 32, 128, 128), nn.MaxPool2d(3, 2, ceil_mode=True),
            fire(7, 256, 32, 128, 128), fire(8, 256, 48, 192, 192), fire(9, 384, 48, 192, 192), fire(10, 384, 64, 256, 256),
            nn.MaxPool2d(3, 2, ceil_mode=True), fire(12, 512, 64, 256, 256))
        self.classifier = nn.Sequential(nn.Dropout(p=0.5), f.conv_bias("classifier.1", 512, num_classes, 1), nn.ReLU(inplace=True),
                                        nn.AdaptiveAvgPool2d((1, 1)))

    def forward(self, x):
        return torch.flatten(self.classifier(self.features(x)), 1)


# ------------------------------------------------------------------ VGG16_Q (nets_cifar/vgg16.py)
class VGG16_Q(nn.Module):
    def __init__(self, f):
        super().__init__()

        def block(li, chans):
            mods = []
            for j, (ci, co) in enumerate(chans):
                mods += [f.conv_bias(f"layer{li}.{3 * j}", ci, co, 3, 1, 1), nn.BatchNorm2d(co), nn.ReLU()]
            return nn.Sequential(*mods, nn.MaxPool2d(2, 2))

        self.layer1 = block(1, [(3, 64), (64, 64)])
        self.layer2 = block(2, [(64, 128), (128, 128)])
        self.layer3 = block(3, [(128, 256), (256, 256), (256, 256)])
        self.layer4 = block(4, [(256, 512), (512, 512), (512, 512)])
        self.layer5 = block(5, [(512, 512), (512, 512), (512, 512)])
        self.fc1 = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(), f.linear("fc1.2", 512, 512), nn.ReLU(), nn.Dropout())
        self.fc2 = nn.Sequential(f.linear("fc2.0", 512, 256), nn.ReLU(), nn.Dropout())
        self.fc3 = f.linear("fc3", 256, 100)

    def forward(self, x):
        x = self.layer5(self.layer4(self.layer3(self.layer2(self.layer1(x)))))
        return self.fc3(self.fc2(self.fc1(x)))


# ------------------------------------------------------------------ ShuffleNetV2 (nets_cifar/shufflenet_v2.py)
def channel_shuffle(x, groups):
    """(b, g, c/g, h, w) -> transpose -> flatten; written with reshape so that channels_last inputs work too."""
    b, c, h, w = x.shape
    return x.reshape(b, groups, c // groups, h, w).transpose(1, 2).reshape(b, c, h, w)


class _ShuffleUnit(nn.Module):
    def __init__(self, f, name, cin, cout, stride):
        super().__init__()
        self.stride, self.cin, self.cout = stride, cin, cout
        r, s = name + ".residual", name + ".shortcut"
        if stride != 1 or cin != cout:
            self.residual = nn.Sequential(
                f.conv(r + ".0", cin, cin, 1), nn.BatchNorm2d(cin), f.layerout(), nn.ReLU(),
                f.conv(r + ".4", cin, cin, 3, stride, 1, groups=cin), nn.BatchNorm2d(cin),
                f.conv(r + ".6", cin, cout // 2, 1), nn.BatchNorm2d(cout // 2), f.layerout(), nn.ReLU())
            self.shortcut = nn.Sequential(
                f.conv(s + ".0", cin, cin, 3, stride, 1, groups=cin), nn.BatchNorm2d(cin),
                f.conv(s + ".2", cin, cout // 2, 1), nn.BatchNorm2d(cout // 2), f.layerout(), nn.ReLU())
        else:
            self.shortcut = nn.Sequential()
            c = cin // 2
            self.residual = nn.Sequential(
                f.conv(r + ".0", c, c, 1), nn.BatchNorm2d(c), f.layerout(), nn.ReLU(),
                f.conv(r + ".4", c, c, 3, stride, 1, groups=c), nn.BatchNorm2d(c),
                f.conv(r + ".6", c, c, 1), nn.BatchNorm2d(c), f.layerout(), nn.ReLU())

    def forward(self, x):
        if self.stride == 1 and self.cout == self.cin:
            shortcut, residual = torch.split(x, self.cin // 2, dim=1)
        else:
            shortcut = residual = x
        return channel_shuffle(torch.cat([self.shortcut(shortcut), self.residual(residual)], dim=1), 2)


class ShuffleNetV2(nn.Module):
    def __init__(self, f, class_num=100):
        super().__init__()
        oc = [116, 232, 464, 1024]
        self.pre = nn.Sequential(f.conv("pre.0", 3, 24, 3, 1, 1), nn.BatchNorm2d(24))

        def stage(name, cin, cout, repeat):
            units = [_ShuffleUnit(f, f"{name}.0", cin, cout, 2)]
            units += [_ShuffleUnit(f, f"{name}.{i + 1}", cout, cout, 1) for i in range(repeat)]
            return nn.Sequential(*units)

        self.stage2 = stage("stage2", 24, oc[0], 3)
        self.stage3 = stage("stage3", oc[0], oc[1], 7)
        self.stage4 = stage("stage4", oc[1], oc[2], 3)
        self.conv5 = nn.Sequential(f.conv("conv5.0", oc[2], oc[3], 1), nn.BatchNorm2d(oc[3]), f.layerout(), nn.ReLU())
        self.fc = f.linear("fc", oc[3], class_num)

    def forward(self, x):
        x = self.conv5(self.stage4(self.stage3(self.stage2(self.pre(x)))))
        x = F.adaptive_avg_pool2d(x, 1)
        return self.fc(x.reshape(x.size(0), -1))


BUILDERS = {"resnet50": ResNet50, "squeezenet": SqueezeNet, "vgg16": VGG16_Q, "shufflenetv2": ShuffleNetV2}


def net_input224(batch, seed):
    """Seeded 224x224 images with per-image contrast and a smooth pattern (as netgen.net_input224)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((batch, 3, 224, 224)).astype(np.float32)
    scale = rng.uniform(0.4, 2.0, (batch, 1, 1, 1)).astype(np.float32)
    offs = rng.uniform(-1.0, 1.0, (batch, 3, 1, 1)).astype(np.float32)
    yy, xx = np.meshgrid(np.linspace(-1, 1, 224, dtype=np.float32), np.linspace(-1, 1, 224, dtype=np.float32), indexing="ij")
    fx = rng.uniform(0.5, 6.0, (batch, 3, 1, 1)).astype(np.float32)
    fy = rng.uniform(0.5, 6.0, (batch, 3, 1, 1)).astype(np.float32)
    return torch.from_numpy((x * scale + offs + 1.2 * np.sin(fx * xx[None, None] * np.pi + fy * yy[None, None] * np.pi)).astype(np.float32))
