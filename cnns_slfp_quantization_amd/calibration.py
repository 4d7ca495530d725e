"""The reference's calibration pass on the device.

`get_scale_factor(model, data_loader, total_images)` mirrors cifar100_train_eval.py:213-277: per batch it
resets the net's stashes, runs the forward, and reads `model.get_layer_inputs()` / `get_layer_outputs()` /
`get_layer_weights()` -- the dictionaries the reference nets fill from every quantized layer's `input_q`
(= QA(x / Ka)) and `weight_q` (= QW(w / Kw)) stash and from the logits (nets_cifar/mobilenetv1.py:88-171) --
and returns `(acc, max_abs_layer_inputs, max_abs_layer_outputs, max_abs_layer_weights)` with the net's own
layer indices.  Where the reference copies every stash to the CPU and takes max|cat(...)| at the end
(:234-271), this keeps one running maximum per layer on the device (slfp_absmax_f32: wave-shuffle
reduction + one atomicMax per wave); max is associative, so the statistics are identical.

`write_scale_files(net, ...)` writes `max_inout_{net}.txt` / `max_weight_{net}.txt` in the reference's format
(:287-301); `scales_from_max` is the step a human does by hand in the reference (paste the maxima into the net
file and divide by 15.5: nets_cifar/mobilenetv1.py:14-18).

`collect_max_abs` is the hook-based variant for nets WITHOUT the stash protocol (nets_imgnet/*, whose read-out is
commented out, nets_imgnet/mobilenetv1.py:84-169): it applies the same definition -- max |input / Ka| after the
activation quantizer, max |QW(weight / Kw)|, max |module output| -- to every Conv2d_Q / Linear_Q in registration
order.
"""
import torch
import torch.nn as nn

from .sfp_quant import absmax


def _amax(t):
    t = t.detach()
    if t.is_cuda and t.dtype == torch.float32:
        return absmax(t if t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)) else t.contiguous())
    return t.abs().max()


def _update(running, idx, t):
    m = _amax(t)
    running[idx] = m if idx not in running else torch.maximum(running[idx], m.to(running[idx].device))


def get_scale_factor(model, data_loader, total_images):
    """cifar100_train_eval.py:213-277.  `data_loader` yields (inputs, targets); inputs are moved to the model's
    device.  Returns (acc, max_abs_layer_inputs, max_abs_layer_outputs, max_abs_layer_weights)."""
    model.eval()
    dev = next(model.parameters()).device
    correct = count = 0
    run_in, run_out, run_w = {}, {}, {}
    with torch.no_grad():
        for inputs, targets in data_loader:
            inputs = inputs.to(dev)
            targets = targets.to(dev) if targets is not None else None
            model.reset_layer_inputs_outputs()
            model.reset_layer_weights()
            outputs = model(inputs)
            for idx, t in model.get_layer_inputs().items():
                _update(run_in, idx, t)
            for idx, t in model.get_layer_outputs().items():
                _update(run_out, idx, t)
            for idx, t in model.get_layer_weights().items():
                _update(run_w, idx, t)
            if targets is not None:
                correct += int(outputs.argmax(1).eq(targets).sum().item())
            count += len(inputs)
            if count >= total_images:
                break
    acc = 100.0 * correct / total_images
    to_py = lambda d: {idx: float(v) for idx, v in d.items()}  # noqa: E731
    return acc, to_py(run_in), to_py(run_out), to_py(run_w)


def scale_files_text(net, max_in, max_out, max_w):
    """The two text files of cifar100_train_eval.py:287-301 as {filename: content}."""
    a = ""
    for idx, v in max_in.items():
        a += f"Layer {idx} Max Absolute Input:\n" + str(v) + "\n\n"
    for idx, v in max_out.items():
        a += f"Layer {idx} Max Absolute Output:\n" + str(v) + "\n\n"
    b = ""
    for idx, v in max_w.items():
        b += f"Layer {idx} Max Absolute weight:\n" + str(v) + "\n\n"
    return {f"max_inout_{net}.txt": a, f"max_weight_{net}.txt": b}


def write_scale_files(net, max_in, max_out, max_w, directory="."):
    import os
    paths = []
    for name, text in scale_files_text(net, max_in, max_out, max_w).items():
        path = os.path.join(directory, name)
        with open(path, "w") as f:
            f.write(text)
        paths.append(path)
    return paths


def quantized_layers(model):
    """Conv2d_Q / Linear_Q modules of `model` in registration order."""
    return [m for m in model.modules() if isinstance(m, (nn.Conv2d, nn.Linear)) and hasattr(m, "Ka") and hasattr(m, "Kw")]


@torch.no_grad()
def collect_max_abs(model, batches, total_images=1000):
    """For nets without the stash protocol: run `batches` (input tensors on the model's device) and return
    (max_abs_inputs, max_abs_outputs, max_abs_weights), dict layer index -> float over the quantized layers in
    registration order, with the reference's definitions: input_q, the layer's output, weight_q."""
    layers = quantized_layers(model)
    run_in, run_out, run_w = {}, {}, {}
    def make_hook(i):
        def hook(m, args, out):   # returns None: a forward hook's return value would replace the module's output
            _update(run_in, i, m.input_q)
            _update(run_out, i, out)
        return hook

    hooks = [mod.register_forward_hook(make_hook(i)) for i, mod in enumerate(layers)]
    was_training = model.training
    model.eval()
    seen = 0
    try:
        for x in batches:
            model(x)
            seen += int(x.shape[0])
            if seen >= total_images:
                break
        for i, mod in enumerate(layers):
            _update(run_w, i, mod.weight_q)
    finally:
        for h in hooks:
            h.remove()
        model.train(was_training)
    to_py = lambda d: {idx: float(v) for idx, v in d.items()}  # noqa: E731
    return to_py(run_in), to_py(run_out), to_py(run_w)


def scales_from_max(max_abs, denom=15.5):
    """Ka / Kw lists as the reference nets build them: np.array(ka) / 15.5 (15 for ShuffleNetV2)."""
    return [max_abs[i] / denom for i in sorted(max_abs)]
