"""Independent image groups on separate HIP streams.

Images are independent units of the quantized forward path (eval-mode BatchNorm, no cross-sample op: SURVEY 8e), so a batch
can run as G groups of N / G images, each group through the WHOLE model on its own stream, with no synchronisation between
the groups until the outputs are needed.  On the MI355X that is worth 6-8 % on MobileNetV1-224 at batch 256 (bench.py:
118 k -> 127-128 k images/s on the 27 conv layers): the short launches of the deep layers fill the chip one and a half
times, and while one group's kernel drains its last workgroups -- or sits in the ~2 us gap between two launches -- the other
group's kernel has the CUs.  Splitting INSIDE every layer (fork / join per Conv2d_Q) loses instead: 104-108 k
(profiles/two_streams.py); the groups have to stay independent across layers, which only the caller of the model can arrange.

    from cnns_slfp_quantization_amd import streams
    logits = streams.forward_image_groups(model, x, groups=2)      # the conv layers' outputs == model(x)'s, bit for bit

It pays where the launch path is cheap -- the C ABI called in a loop (bench.py), a hipGraph replay per group: through EAGER
Python modules the host issues every launch twice and becomes the bottleneck (whole MobileNetV1, linked, batch 256: 175 k
images/s in two groups vs 212 k in one).  Note that torch's own reductions (adaptive_avg_pool2d) may round differently at
N / G images than at N: logits can differ in the last bits although every Conv2d_Q output is identical.

The reference has nothing comparable (one CUDA stream, utils/conv2d_func.py:20-25 is called layer by layer on the whole batch)."""
import torch

_side = {}


def _streams(device, n):
    key = (device.index, n)
    st = _side.get(key)
    if st is None:
        st = [torch.cuda.Stream(device=device) for _ in range(n)]
        _side[key] = st
    return st


def forward_image_groups(model, x, groups=2, streams=None):
    """model(x) computed as `groups` independent slices of the batch, one HIP stream each (`streams`: the torch.cuda.Stream
    objects to use; default: created once per device and group count).  HIP maps streams to a handful of hardware queues and two
    streams that share a queue serialise (profiles/stream_pairs.py: of the 28 pairs among a process's first eight streams five
    do -- 107 k instead of 126 k images/s); the first two streams a process creates have always landed on different queues.  `x`: a ROCm ('cuda') tensor with the
    batch in dimension 0; the model must treat images independently (inference: eval-mode BatchNorm).  The caller's current
    stream waits for all groups before the result is returned.  Returns the concatenated outputs (a tensor, or a tuple of
    tensors if the model returns a tuple)."""
    if not x.is_cuda:
        raise TypeError("forward_image_groups: expected a ROCm ('cuda') tensor")
    n = x.shape[0]
    groups = max(1, min(int(groups), n))
    if groups == 1:
        return model(x)
    cur = torch.cuda.current_stream(x.device)
    ready = torch.cuda.Event()
    ready.record(cur)
    bounds = [(n * g) // groups for g in range(groups + 1)]
    outs, done = [], []
    for g, st in enumerate(streams if streams is not None else _streams(x.device, groups)):
        st.wait_event(ready)                       # x (and the weights) are complete on the caller's stream
        with torch.cuda.stream(st):
            xg = x[bounds[g]:bounds[g + 1]]
            xg.record_stream(st)
            outs.append(model(xg))
            ev = torch.cuda.Event()
            ev.record(st)
            done.append(ev)
    for ev in done:
        cur.wait_event(ev)
    for o in outs:
        for t in (o if isinstance(o, (tuple, list)) else (o,)):
            t.record_stream(cur)
    if isinstance(outs[0], (tuple, list)):
        return tuple(torch.cat([o[i] for o in outs], 0) for i in range(len(outs[0])))
    return torch.cat(outs, 0)
