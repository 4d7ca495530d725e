"""Small-batch inference as one hipGraph (host path).

At small batch a forward through the drop-in modules is launch- and host-bound (MobileNetV1-224 at batch 1-8: ~0.62 ms of
host work for ~0.3 ms of GPU work, profiles/host_overhead.py).  Every launch of the library goes to the stream it is
handed, so a whole forward captures into a hipGraph; `GraphedModule` does that once per input signature and replays it:

    model = MobileNetV1_Q(...).cuda().eval().to(memory_format=torch.channels_last)
    fusion.fuse_bn_relu(model)
    fast = GraphedModule(model)
    logits = fast(images)            # first call per (shape, dtype, layout): eager warm-up + capture; later calls: replay

Inference only (the module is put in eval mode and run under no_grad; the prepared weights are the cached ones).  After
changing weights or scales call `reset()`.  The reference has no counterpart: its forward is eager PyTorch
(nets_imgnet/mobilenetv1.py:84-169).
"""
import torch
import torch.nn as nn


class _Entry:
    __slots__ = ("graph", "static_in", "static_out")


class GraphedModule(nn.Module):
    def __init__(self, model, max_graphs=8, clone_output=True):
        super().__init__()
        self.model = model.eval()
        self.max_graphs = int(max_graphs)
        self.clone_output = bool(clone_output)
        self._entries = {}

    def reset(self):
        """Drop every captured graph (after a weight / scale / fusion change)."""
        self._entries.clear()

    @staticmethod
    def _signature(x):
        return (tuple(x.shape), x.dtype, x.device, tuple(x.stride()))

    def _capture(self, x):
        e = _Entry()
        e.static_in = torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device=x.device)
        e.static_in.copy_(x)
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):   # plans, prepared weights and workspaces exist before the capture
                self.model(e.static_in)
        torch.cuda.current_stream(x.device).wait_stream(side)
        e.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(e.graph):
            e.static_out = self.model(e.static_in)
        return e

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("GraphedModule needs a ROCm ('cuda') input tensor")
        if self.model.training:
            raise RuntimeError("GraphedModule is inference-only: the wrapped model went back to training mode")
        key = self._signature(x)
        e = self._entries.get(key)
        if e is None:
            if len(self._entries) >= self.max_graphs:
                self._entries.pop(next(iter(self._entries)))
            with torch.cuda.device(x.device):
                e = self._capture(x)
            self._entries[key] = e
        else:
            e.static_in.copy_(x)
        e.graph.replay()
        out = e.static_out
        if self.clone_output:
            out = tuple(o.clone() for o in out) if isinstance(out, (tuple, list)) else out.clone()
        return out
