"""HIP-graph replay of a fixed-shape MoE training step.

One forward + backward of the layer is ~50 kernel launches of 5-200 us each; issued one by one from
Python the host needs longer to enqueue them than the MI355X needs to run them.  `GraphedTrainStep`
captures the whole step (Gumbel draw, gate, dispatch, expert FFNs, combine and the full backward) once
into a HIP graph and replays it with a single launch.  Nothing is skipped or cached: every replay draws
fresh noise from the device generator (the philox offset advances per replay) and recomputes routing,
outputs and all gradients into the same static buffers.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import Tensor, nn


class GraphedTrainStep:
    """step(h, dout) -> (out, l_aux); gradients land in `p.grad` of the module's parameters and in `.dx`.

    `module(h, mode=...)` must return `(out, l_aux, ...)` like `eve_amd.MoE`.  Shapes, dtypes and `mode`
    are fixed at construction.  The loss is `sum(out * dout) + aux_weight * l_aux`."""

    def __init__(self, module: nn.Module, h: Tensor, dout: Tensor, mode: str = "vl", aux_weight: float = 1.0,
                 warmup: int = 3) -> None:
        if not h.is_cuda:
            raise RuntimeError("GraphedTrainStep needs CUDA/HIP tensors")
        self.module, self.mode = module, mode
        with torch.no_grad():  # settle lazy state first (fp32 gate weights, packed expert storage)
            module(h, mode=mode)
        self.names = [n for n, p in module.named_parameters() if p.requires_grad]
        self.params: List[nn.Parameter] = [p for _, p in module.named_parameters() if p.requires_grad]
        self.static_h = h.detach().clone()
        self.static_dout = dout.detach().clone()
        self._aux_grad = torch.full((), float(aux_weight), dtype=torch.float32, device=h.device)
        # The step runs on fresh leaves that share the parameters' storage: an autograd graph of an
        # earlier eager step that the caller still holds keeps the parameters' gradient accumulators
        # bound to the stream of that step, and a capture may not touch another stream.
        self._leaves = {n: p.detach().requires_grad_(True) for n, p in zip(self.names, self.params)}
        # warm up on a side stream (lazy initialisations, kernel attribute calls, allocator) as
        # torch.cuda.graph requires, then capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for leaf in self._leaves.values():
            leaf.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out, self.l_aux, self.dx = self._eager()
        self.grads: List[Optional[Tensor]] = [self._leaves[n].grad for n in self.names]
        self._publish_grads()

    def _publish_grads(self) -> None:
        for p, g in zip(self.params, self.grads):
            p.grad = g  # static buffers, rewritten by every replay

    def _eager(self):
        x = self.static_h.detach().requires_grad_(True)
        res = torch.func.functional_call(self.module, self._leaves, (x,), {"mode": self.mode})
        out, l_aux = res[0], res[1]
        torch.autograd.backward([out, l_aux], [self.static_dout, self._aux_grad.to(l_aux.dtype)])
        return out.detach(), l_aux.detach(), x.grad

    def __call__(self, h: Optional[Tensor] = None, dout: Optional[Tensor] = None):
        if h is not None:
            self.static_h.copy_(h)
        if dout is not None:
            self.static_dout.copy_(dout)
        self.graph.replay()
        self._publish_grads()
        return self.out, self.l_aux
