"""Data-parallel gradient exchange for the LoRA fine-tune (replaces the four DDP wrappers of vla-scripts/finetune.py:212-224,
891-932): the only collective on the path is the mean all-reduce of the trainable gradients once per optimizer step.

All trainable gradients already live in one flat fp32 buffer per parameter dtype (engine.ParamStore), laid out in the
order the backward produces them, so the exchange is a handful of large RCCL all-reduces (torch.distributed's "nccl"
backend IS RCCL on ROCm) instead of DDP's ~25 MB buckets -- xGMI is point-to-point, ring collectives are per-link bound,
so few large messages beat many small ones.  The division by world size is folded into the fused AdamW (grad_scale).

Wire dtype.  The reference's DDP reduces every gradient in its PARAMETER dtype (`gradient_as_bucket_view=True`, finetune.py:224: bf16 for
the LoRA factors and the action head, fp32 for the proprio / noisy-action projectors and FiLM).  `comm_dtype="param"` (default) does the
same: the fp32 accumulators of bf16 parameters are rounded to bf16 on the communication stream (one HBM-bound kernel per bucket), summed
over the ranks in bf16 and widened back -- 0.52 GB per step on the wire instead of 1.04 GB for the LIBERO recipe.
`comm_dtype="fp32"` (env OVLA_DP_COMM_DTYPE=fp32) ships the accumulators as they are: one rounding fewer per gradient, twice the xGMI
bytes.  Nothing here has been measured on more than one GPU yet (no multi-GPU node in the build container): see DESIGN.md 2(e).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

BF16 = torch.bfloat16


class GradReducer:
    """Sum-reduces the flat gradient buffers across ranks in buckets, on a communication stream.

    `all_reduce()` ships whatever has not been shipped yet and makes the compute stream wait: call it once after the
    backward.  For overlap, `notify(store, frontier)` may be called DURING the backward with the number of leading
    elements of `store`'s flat buffers whose gradients are final (the buffers are laid out in backward completion
    order): every bucket entirely below the frontier is launched immediately behind an event on the compute stream.
    The mean's 1/world lives in the optimizer's grad_scale.
    """

    def __init__(self, stores, world: int, bucket_bytes: int = 128 << 20, comm_dtype: str = None):
        self.stores, self.world = list(stores), world
        self.bucket_elems = bucket_bytes // 4
        self.comm_dtype = comm_dtype or os.environ.get("OVLA_DP_COMM_DTYPE", "param")
        if self.comm_dtype not in ("param", "fp32"):
            raise ValueError(f"comm_dtype must be 'param' or 'fp32', got {self.comm_dtype!r}")
        self._stage = {}     # device -> bf16 staging bucket (reused in comm-stream order)
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

    def _all_reduce(self, t):
        if t.is_cuda and dist.get_backend() != "nccl":   # gloo rehearsal of the multi-process path on a 1-GPU box
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def _reduce(self, t, param_dtype=torch.float32):
        """Sums the fp32 accumulator slice `t` over the ranks, on the wire in fp32 or (bf16 parameters, comm_dtype 'param') in bf16."""
        if param_dtype != BF16 or self.comm_dtype == "fp32":
            return self._all_reduce(t)
        n = t.numel()
        if t.is_cuda:
            from . import ops   # the HIP library: required on a GPU (no torch-math fallback in the product path)

            stage = self._stage.get(t.device)
            if stage is None or stage.numel() < n:
                stage = self._stage[t.device] = torch.empty(max(n, self.bucket_elems), dtype=BF16, device=t.device)
            ops.cvt_f32_to_bf16(t, stage[:n])
            self._all_reduce(stage[:n])
            ops.cvt_bf16_to_f32(stage[:n], t)
        else:                # CPU tensors exist only in the gloo unit tests of this class
            w = t.to(BF16)
            self._all_reduce(w)
            t.copy_(w)

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
                    self._reduce(g[off: min(upto, off + self.bucket_elems)], dtype)
        else:
            for off in range(lo, upto, self.bucket_elems):
                self._reduce(g[off: min(upto, off + self.bucket_elems)], dtype)
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
