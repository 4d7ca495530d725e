"""Device-side version of the reference's calibration pass.

The reference's `get_scale_factor` (cifar100_train_eval.py:213-277) copies every quantized layer's
`input_q` / `weight_q` stash to the CPU for every batch and takes max|.| over the concatenation at
the end; a human then pastes the maxima into the net file and divides by 15.5 to get Ka / Kw
(nets_cifar/mobilenetv1.py:14-18).  Here forward pre-hooks keep a running max|input| per quantized
layer on the device (slfp_absmax_f32: wave-shuffle reduction + one atomicMax per wave) and the
weights' max|w| is taken once: no host copies, one float per layer comes back at the end.
"""
import torch
import torch.nn as nn

from .sfp_quant import absmax


def quantized_layers(model):
    """Conv2d_Q / Linear_Q modules of `model` in registration order (= the reference's stash order
    for its nn.Sequential nets)."""
    return [m for m in model.modules() if isinstance(m, (nn.Conv2d, nn.Linear)) and hasattr(m, "Ka") and hasattr(m, "Kw")]


@torch.no_grad()
def collect_max_abs(model, batches, total_images=1000):
    """Run `batches` (iterable of input tensors already on the model's device) through the model and
    return (max_abs_inputs, max_abs_weights): dict layer index -> float, the statistics
    get_scale_factor returns.  Calibrate with q_bit = 32 (identity quantizers), as the reference does."""
    layers = quantized_layers(model)
    running = {}
    hooks = []

    def make_hook(i):
        def hook(mod, args):
            x = args[0]
            m = absmax(x) if x.is_cuda and x.dtype == torch.float32 else x.detach().abs().max()
            running[i] = m if i not in running else torch.maximum(running[i], m)
        return hook

    for i, mod in enumerate(layers):
        hooks.append(mod.register_forward_pre_hook(make_hook(i)))
    was_training = model.training
    model.eval()
    seen = 0
    try:
        for x in batches:
            model(x)
            seen += int(x.shape[0])
            if seen >= total_images:
                break
    finally:
        for h in hooks:
            h.remove()
        model.train(was_training)
    max_in = {i: float(v) for i, v in running.items()}
    max_w = {}
    for i, mod in enumerate(layers):
        w = mod.weight.detach()
        max_w[i] = float(absmax(w) if w.is_cuda and w.dtype == torch.float32 else w.abs().max())
    return max_in, max_w


def scales_from_max(max_abs, denom=15.5):
    """Ka / Kw lists as the reference nets build them: np.array(ka) / 15.5 (15 for ShuffleNetV2)."""
    return [max_abs[i] / denom for i in sorted(max_abs)]
