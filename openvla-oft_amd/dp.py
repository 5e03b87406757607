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
    def __init__(self, stores, world: int, bucket_bytes: int = 256 << 20):
        self.stores, self.world = list(stores), world
        self.bucket_elems = bucket_bytes // 4
        self.comm_stream = torch.cuda.Stream() if torch.cuda.is_available() else None

    def buckets(self):
        for store in self.stores:
            for g in store.flat_grad.values():
                n = g.numel()
                for off in range(0, n, self.bucket_elems):
                    yield g[off: min(n, off + self.bucket_elems)]

    def all_reduce(self):
        """Sum-reduces every gradient bucket across ranks on the communication stream and makes the compute stream wait
        for it (the mean's 1/world lives in the optimizer's grad_scale)."""
        if self.world <= 1:
            return
        if self.comm_stream is None:       # CPU / gloo (tests)
            for b in self.buckets():
                dist.all_reduce(b, op=dist.ReduceOp.SUM)
            return
        cur = torch.cuda.current_stream()
        self.comm_stream.wait_stream(cur)
        with torch.cuda.stream(self.comm_stream):
            for b in self.buckets():
                dist.all_reduce(b, op=dist.ReduceOp.SUM)
        cur.wait_stream(self.comm_stream)
