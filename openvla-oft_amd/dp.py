"""Data-parallel gradient exchange for the LoRA fine-tune (replaces the four DDP wrappers of vla-scripts/finetune.py:212-224,
891-932): the only collective on the path is the mean all-reduce of the trainable gradients once per optimizer step.

All trainable gradients already live in one flat fp32 buffer per parameter dtype (engine.ParamStore), laid out in the
order the backward produces them, so the exchange is a handful of large RCCL all-reduces (torch.distributed's "nccl"
backend IS RCCL on ROCm) instead of DDP's ~25 MB buckets -- xGMI is point-to-point, ring collectives are per-link bound,
so few large messages beat many small ones.  The division by world size is folded into the fused AdamW (grad_scale).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradReducer:
    """Sum-reduces the flat gradient buffers across ranks in buckets, on a communication stream.

    `all_reduce()` ships whatever has not been shipped yet and makes the compute stream wait: call it once after the
    backward.  For overlap, `notify(store, frontier)` may be called DURING the backward with the number of leading
    elements of `store`'s flat buffers whose gradients are final (the buffers are laid out in backward completion
    order): every bucket entirely below the frontier is launched immediately behind an event on the compute stream.
    The mean's 1/world lives in the optimizer's grad_scale.
    """

    def __init__(self, stores, world: int, bucket_bytes: int = 128 << 20):
        self.stores, self.world = list(stores), world
        self.bucket_elems = bucket_bytes // 4
        self.use_stream = torch.cuda.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
        self.comm_stream = torch.cuda.Stream() if self.use_stream else None
        self.reset()

    def reset(self):
        self.sent = {(id(st), dt): 0 for st in self.stores for dt in st.flat_grad}

    def buckets(self):
        for store in self.stores:
            for g in store.flat_grad.values():
                n = g.numel()
                for off in range(0, n, self.bucket_elems):
                    yield g[off: min(n, off + self.bucket_elems)]

    def _reduce(self, t):
        if t.is_cuda and dist.get_backend() != "nccl":   # gloo rehearsal of the multi-process path on a 1-GPU box
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def _ship(self, store, dtype, upto):
        g = store.flat_grad[dtype]
        key = (id(store), dtype)
        lo = self.sent[key]
        if upto <= lo:
            return
        if self.use_stream:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                for off in range(lo, upto, self.bucket_elems):
                    self._reduce(g[off: min(upto, off + self.bucket_elems)])
        else:
            for off in range(lo, upto, self.bucket_elems):
                self._reduce(g[off: min(upto, off + self.bucket_elems)])
        self.sent[key] = upto

    def notify(self, store, dtype, frontier_elems: int):
        """Gradients [0, frontier_elems) of store.flat_grad[dtype] are final; ship the whole buckets below the frontier."""
        if self.world <= 1:
            return
        self._ship(store, dtype, (frontier_elems // self.bucket_elems) * self.bucket_elems)

    def all_reduce(self):
        if self.world <= 1:
            return
        for store in self.stores:
            for dtype, g in store.flat_grad.items():
                self._ship(store, dtype, g.numel())
        if self.use_stream:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        # `sent` stays at "everything" until reset() (called from zero_grad at the start of the next step), so a repeated
        # call in the same step cannot reduce the same bytes twice
