"""Execution engine of the OpenVLA-OFT action-chunk path on MI355X.

Host-side orchestration only: every tensor op below is a hand-written gfx950 kernel reached through the C-ABI
(`ops.py` -> libovla_hip.so).  PyTorch supplies device memory and the HIP stream.  Forward and backward are written
out explicitly per block (no autograd tape, no tracing compiler): each `fwd` returns the activations its `bwd` needs.

HBM layout (sized for 288 GB):
  * frozen base weights are resident twice, as W [out, in] and W^T [in, out], so forward AND data-gradient GEMMs are
    the same K-contiguous "NT" kernel (no transposed-operand kernel variants, no per-step transposes);
  * Llama q|k|v and gate|up are fused along N; their LoRA pairs are stacked the same way and ride in the GEMM's
    K-extension (one launch per fused linear);
  * all trainable tensors (LoRA A/B, action head, projectors) are views into two flat buffers (bf16 / fp32) with matching
    flat fp32 gradient buffers: one fused AdamW launch and one RCCL all-reduce per dtype bucket;
  * activations for the backward stay resident in bf16 (no recompute at 288 GB).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import os

import torch

from . import _lib, ops
from .config import VLAConfig, VitConfig

BF16 = torch.bfloat16
F32 = torch.float32
# Which activation derivatives ride in the data-gradient GEMM's epilogue (ovla.h "backward epilogues"): none | act | swiglu | all.
# Bit-identical to the separate act_bwd / swiglu_bwd kernels and it saves the dh round trip, but measured on MI355X it is NOT faster
# (B = 8: none 177.1-177.6 ms/step, act 177.5-178.0, swiglu 178.0, all 178.1-178.4): a 256x256 tile owns its CU, so the extra epilogue
# traffic is exposed, while the separate elementwise kernels already run at the HBM roofline.  Round 2 moved SwiGLU' into the UNROLLED epilogue
# (gemm_nt.hip fast_swiglu_bwd): still no gain (170.0 / 170.4 with vs 170.5 / 170.3 without).  Default: separate kernels.
_FUSE = os.environ.get("OVLA_FUSE_DACT", "none")
# The forward RoPE rides in the q|k|v projection's epilogue (bit-identical to the separate rope pass).  Round 1 had it only in the 4x2-wave
# layout's ROLLED epilogue, where it cost more than the 33 us pass it replaced (177.8-178.0 vs 176.7-177.0 ms/step) and was left off; round 2
# put it into the default 2x4 layout's unrolled read-back (a wave takes its columns together with their rotation partners: no extra LDS
# traffic beyond one more slab read, no barrier) -- see DESIGN.md section 6 for the A/B.  OVLA_FUSE_ROPE_FWD=0 restores the separate pass.
# The INVERSE RoPE of the backward, which needs no loads beyond the tables and no LDS (attention-backward epilogue), is on: -0.6 ms.
_FUSE_ROPE_FWD = os.environ.get("OVLA_FUSE_ROPE_FWD", "1") == "1"
_GROUP_TN = os.environ.get("OVLA_GROUP_TN", "1") == "1"   # ViT blocks: two neighbouring linears' LoRA weight-gradient problems in one launch (A/B switch)
_FUSE_ACT, _FUSE_SWIGLU = _FUSE in ("act", "all"), _FUSE in ("swiglu", "all")
# The action head's tail (two MLPResNet blocks, LayerNorm 2, fc2, L1 / MSE loss) as ONE launch for up to 64 rows (ovla_head_tail_fwd),
# bit-identical to the unfused eight-launch sequence.  OPT-IN (OVLA_FUSE_HEAD=1) since round 3: it never measured faster -- round 2: equal at 64
# rows, 50 us slower per replayed inference chunk; round 3, after an 8-deep register ring on its weight / activation streams (kernel 148.8 ->
# 114.6 us): whole head forward 219.8 vs 209.6 us unfused at 64 rows, 186.1 vs 169.6 us at 8 rows (tools/head_bench.py,
# profiles/r03_head_bench.txt) -- four software grid barriers cost more than the eight kernel boundaries they replace, and a software grid
# barrier needs every workgroup resident at once: the launch is refused otherwise, a timed-out spin poisons every output with NaN and sets a
# sticky word that ActionHead.check_fused_tail() turns into an exception at the step's host sync.
_FUSE_HEAD = {"0": False, "1": True}.get(os.environ.get("OVLA_FUSE_HEAD", "auto"), None)
# OVLA_ASYNC_UPLOAD=0: upload ids / labels / lengths with synchronous pageable copies as before (A/B switch of VLAEngine.forward's pinned upload)
_ASYNC_UPLOAD = os.environ.get("OVLA_ASYNC_UPLOAD", "1") != "0"
# OVLA_FOLD_RMSNORM=0: keep the decoder's RMSNorms as their own launches on the merged inference path (A/B switch of LlamaStack.fold_norms).
_FOLD_RMSNORM = os.environ.get("OVLA_FOLD_RMSNORM", "1") != "0"
# OVLA_INFER_W4=0: the folded inference path's projections take the planner's tile instead of the 4-wave 128x256 configuration (A/B switch of LlamaStack._infer_tile)
_INFER_W4 = os.environ.get("OVLA_INFER_W4", "1") != "0"
# OVLA_FUSE_SWIGLU_FWD=1: the decoder's gate|up projection and SwiGLU become one launch in the unfolded forward too (OVLA_ACT_SWIGLU on the 256x256 tile, C_pre keeps
# the [M, 2F] projection output for the backward).  Bit-identical; measured neutral on the fine-tune step (158.1 / 158.5 vs 158.0 / 158.0 ms: the read-back that saves
# the swiglu_fwd launch is itself bound by the tile's stores), so it stays off there; the folded batch-1 path always fuses (LlamaStack._infer_tile).
_FUSE_SWIGLU_FWD = os.environ.get("OVLA_FUSE_SWIGLU_FWD", "0") == "1"
# OVLA_LORA_BWD=1: the LoRA backward's dt and dB from ONE pass over dy (csrc/lora_bwd.hip) instead of a skinny NT GEMM (dt) + TN GEMMs (dB, dA).
# Built, parity-tested and measured in round 3 (tools/lora_bwd_bench.py, cold operands, M = 4864): 293.6 vs 272.4 us per decoder layer for the
# three-kernel path -- reading dy once saves 0.4 GB per layer, but a 2-D (rows x 256-column) decomposition pays it back as fp32 partial-dt slabs
# (+25 % of dy's bytes, written and read) and dB atomics (+21 %, at a quarter of the streaming rate); left OFF.
_LORA_BWD_FUSED = os.environ.get("OVLA_LORA_BWD", "0") == "1"


# ======================================================================================================================
# trainable-parameter store
# ======================================================================================================================
class Param:
    """Handle to one trainable tensor living in a flat buffer (`data`), with its fp32 gradient view (`grad`)."""

    def __init__(self, name, init: torch.Tensor):
        self.name, self.shape, self.dtype = name, tuple(init.shape), init.dtype
        self._init = init
        self.data: Optional[torch.Tensor] = None
        self.grad: Optional[torch.Tensor] = None
        self.offset = 0

    @property
    def numel(self):
        return math.prod(self.shape)


class ParamStore:
    ALIGN = 64  # elements; keeps every view 128-byte aligned

    def __init__(self, device):
        self.device = device
        self.params: List[Param] = []
        self.flat: Dict[torch.dtype, torch.Tensor] = {}
        self.flat_grad: Dict[torch.dtype, torch.Tensor] = {}
        self.exp_avg: Dict[torch.dtype, torch.Tensor] = {}
        self.exp_avg_sq: Dict[torch.dtype, torch.Tensor] = {}
        self.step = 0
        self.finalized = False

    def add(self, name: str, init: torch.Tensor) -> Param:
        assert not self.finalized and init.dtype in (BF16, F32)
        p = Param(name, init.to(self.device))
        self.params.append(p)
        return p

    def finalize(self):
        """Lays the tensors out in REVERSE registration order (= the order the backward produces their gradients), so
        gradient buckets complete front to back."""
        for dtype in (BF16, F32):
            group = [p for p in reversed(self.params) if p.dtype == dtype]
            total = 0
            for p in group:
                p.offset = total
                total += (p.numel + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            if total == 0:
                continue
            flat = torch.zeros(total, dtype=dtype, device=self.device)
            grad = torch.zeros(total, dtype=F32, device=self.device)
            for p in group:
                p.data = flat[p.offset: p.offset + p.numel].view(p.shape)
                p.data.copy_(p._init)
                p.grad = grad[p.offset: p.offset + p.numel].view(p.shape)
                p._init = None
            self.flat[dtype], self.flat_grad[dtype] = flat, grad
        self.finalized = True

    def init_optimizer(self):
        for dtype, flat in self.flat.items():
            self.exp_avg[dtype] = torch.zeros_like(flat)
            self.exp_avg_sq[dtype] = torch.zeros_like(flat)
        self.step = 0

    def zero_grad(self):
        for g in self.flat_grad.values():
            g.zero_()

    def adamw_step(self, lr: float, *, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, grad_scale=1.0):
        """torch.optim.AdamW(trainable_params, lr) of vla-scripts/finetune.py:952 as one fused launch per dtype."""
        if not self.exp_avg:
            self.init_optimizer()
        self.step += 1
        for dtype, flat in self.flat.items():
            ops.adamw(flat, self.exp_avg[dtype], self.exp_avg_sq[dtype], self.flat_grad[dtype], step=self.step, lr=lr, beta1=beta1,
                      beta2=beta2, eps=eps, weight_decay=weight_decay, grad_scale=grad_scale)

    def num_trainable(self):
        return sum(p.numel for p in self.params)

    def named(self) -> Dict[str, Param]:
        return {p.name: p for p in self.params}


# ======================================================================================================================
# linears
# ======================================================================================================================
class LoraLinear:
    """Frozen nn.Linear (+ optional bias) with peft-style LoRA adapters; `groups` fused sub-linears share the input.

    forward:  t_s = s * x A^T ;  y = x W^T (+ bias) + t_s B^T                     (one K-extended GEMM)
    backward: dt  = s * dy_g B_g   (per group);  dx = dy W + dt A                  (one K-extended GEMM on W^T / A^T)
              dB_g += dy_g^T t_s,g ;  dA += dt^T x                                 (TN GEMMs, fp32 accumulate)
    """

    def __init__(self, store: Optional[ParamStore], name: str, W: torch.Tensor, bias: Optional[torch.Tensor], lora_A: Optional[torch.Tensor],
                 lora_B: Optional[torch.Tensor], groups: int, scale: float, need_dgrad: bool = True, ref_names=None):
        self.name, self.W, self.bias, self.groups, self.scale = name, W.contiguous(), bias, groups, scale
        self.ref_names = ref_names or [name]   # the reference's nn.Linear module path of every fused group
        self.out_f, self.in_f = W.shape
        self.group_n = self.out_f // groups
        self.WT = ops.transpose(self.W) if need_dgrad else None
        self.has_lora = lora_A is not None
        if self.has_lora:
            self.r = lora_A.shape[0] // groups
            assert lora_A.shape == (groups * self.r, self.in_f) and lora_B.shape == (self.out_f, self.r)
            self.A = store.add(name + ".lora_A", lora_A.to(BF16))
            self.B = store.add(name + ".lora_B", lora_B.to(BF16))
            self.AT = None   # [in, G*r]   derived
            self.BT = None   # list of [r, group_n] derived

    def derived_pairs(self):
        """(src, dst) transposes that re-derive A^T [in, G*r] and the stacked B_g^T [G*r, group_n] from the trainable factors."""
        if not self.has_lora:
            return []
        dev = self.W.device
        if self.AT is None:
            self.AT = torch.empty((self.in_f, self.groups * self.r), dtype=BF16, device=dev)
            self.BT = torch.empty((self.groups * self.r, self.group_n), dtype=BF16, device=dev)
        pairs = [(self.A.data, self.AT)]
        for g in range(self.groups):
            pairs.append((self.B.data[g * self.group_n:(g + 1) * self.group_n], self.BT[g * self.r:(g + 1) * self.r]))
        return pairs

    def refresh_derived(self):
        for src, dst in self.derived_pairs():
            ops.transpose(src, dst)

    def export(self, kind: str = "data") -> Dict[str, torch.Tensor]:
        """Trainable tensors (kind='data') or their fp32 gradients (kind='grad') under the reference/peft-style names."""
        out = {}
        if self.has_lora:
            A, Bm = getattr(self.A, kind), getattr(self.B, kind)
            for g, rn in enumerate(self.ref_names):
                out[rn + ".lora_A.weight"] = A[g * self.r:(g + 1) * self.r]
                out[rn + ".lora_B.weight"] = Bm[g * self.group_n:(g + 1) * self.group_n]
        return out

    def merge(self):
        """W += s * B A in place (peft `merge_and_unload`, vla-scripts/merge_lora_weights_and_save.py:60-67): per fused group
        one rank-r GEMM whose epilogue adds the frozen weight (delta and sum each rounded to bf16, as peft's
        `weight.data += delta` on bf16 tensors).  The adapters stay allocated but are no longer applied: inference only."""
        if not self.has_lora or getattr(self, "merged", False):
            return
        if self.AT is None:
            self.refresh_derived()
        gn, r = self.group_n, self.r
        for g in range(self.groups):
            Wg = self.W[g * gn:(g + 1) * gn]
            ops.gemm(self.B.data[g * gn:(g + 1) * gn], self.AT[:, g * r:(g + 1) * r], out=Wg, residual=Wg, alpha=self.scale)
        if self.WT is not None:
            ops.transpose(self.W, self.WT)
        self.merged = True

    def fwd(self, x, *, act=0, residual=None, colscale=None, c_pre=None, film=None, out=None, rope=None):
        """x [M, in] -> (y [M, out], saved).  `rope` = (cos, sin, S, cols): RoPE on the first `cols` output columns in the GEMM epilogue."""
        t_s = None
        if self.has_lora and not getattr(self, "merged", False):
            t_s = ops.gemm(x, self.A.data, alpha=self.scale)
            y = ops.gemm(x, self.W, out=out, bias=self.bias, act=act, residual=residual, colscale=colscale, c_pre=c_pre, film=film,
                         a2=t_s, b2=self.B.data, k2_group_n=self.group_n if self.groups > 1 else 0, rope=rope)
        else:
            y = ops.gemm(x, self.W, out=out, bias=self.bias, act=act, residual=residual, colscale=colscale, c_pre=c_pre, film=film, rope=rope)
        return y, (x, t_s)

    def bwd(self, dy, saved, need_dx=True, dact=None, tn_queue=None):
        """dy [M, out] (gradient w.r.t. the pre-activation linear output) -> dx [M, in]; accumulates LoRA grads.
        `dact` = ("act", z, act_id) / ("swiglu", gu): the data-gradient GEMM's epilogue also applies the derivative of the
        activation that PRODUCED this linear's input (returns dz / d(gate|up) instead of dx: ovla.h backward epilogues).
        `tn_queue`: a list that collects this linear's weight-gradient TN problems instead of launching them; the caller launches the
        list (ops.gemm_tn_grouped) BEFORE anything overwrites `dy` -- two neighbouring linears' four small problems share one launch."""
        x, t_s = saved
        dx = None
        if getattr(self, "merged", False):
            raise RuntimeError(f"{self.name}: LoRA adapters were merged into the base weight; this engine is inference-only")
        if self.has_lora:
            r, G, gn = self.r, self.groups, self.group_n
            if _LORA_BWD_FUSED and r == 32 and dy.shape[0] >= 512:
                # ONE pass over dy: dt = s * dy_g . B_g and dB_g += dy_g^T t_g together (csrc/lora_bwd.hip); only dA += dt^T x is left to the TN GEMM
                dt = ops.lora_bwd(dy, self.BT, t_s, self.B.grad, gn=gn, G=G, scale=self.scale)
                probs = [(dt, x, self.A.grad)]
            else:
                # dt[:, g] = s * dy_g . B_g for every fused group in ONE block-diagonal skinny GEMM
                dt = ops.gemm(dy, self.BT, alpha=self.scale, a_group_n=r if G > 1 else 0)
                # dB_g += dy_g^T t_g ; dA += dt^T x : one grouped launch
                probs = [(dy[:, g * gn:(g + 1) * gn], t_s[:, g * r:(g + 1) * r], self.B.grad[g * gn:(g + 1) * gn]) for g in range(G)]
                probs.append((dt, x, self.A.grad))
            if tn_queue is not None:
                tn_queue.extend(probs)
            else:
                ops.gemm_tn_grouped(probs)
            if need_dx:
                dx = ops.gemm(dy, self.WT, a2=dt, b2=self.AT, dact=dact)
        elif need_dx:
            dx = ops.gemm(dy, self.WT, dact=dact)
        return dx


class FullLinear:
    """Fully trainable nn.Linear (action head, proprio / noisy-action projector, FiLM).  `master_fp32`: parameters live
    in fp32 (the reference never casts these modules, finetune.py:895-901) and a bf16 compute copy is refreshed after
    each optimizer step -- what autocast does on every call.  The input width is zero-padded to a multiple of 8 (the
    GEMM's K granularity; proprio_dim 8, noisy-action dim 1): the padded columns see zero inputs, get zero gradients
    and stay zero, and are sliced off on export."""

    def __init__(self, store: ParamStore, name: str, W: torch.Tensor, bias: Optional[torch.Tensor], master_fp32: bool = False):
        dt = F32 if master_fp32 else BF16
        self.name, self.master_fp32 = name, master_fp32
        self.out_f, self.in_f = W.shape
        self.in_pad = (self.in_f + 7) // 8 * 8
        Wp = torch.zeros((self.out_f, self.in_pad), dtype=dt, device=W.device)
        Wp[:, : self.in_f] = W.to(dt)
        self.W = store.add(name + ".weight", Wp)
        self.b = store.add(name + ".bias", bias.to(dt)) if bias is not None else None
        self.Wc = self.bc = None

    def derived_pairs(self):
        return []   # nothing to transpose; the bf16 compute copy is refreshed by refresh_derived()

    def refresh_derived(self):
        if self.master_fp32:
            self.Wc = ops.cvt_f32_to_bf16(self.W.data, self.Wc)
            if self.b is not None:
                self.bc = ops.cvt_f32_to_bf16(self.b.data, self.bc)
        else:
            self.Wc, self.bc = self.W.data, (None if self.b is None else self.b.data)

    def export(self, kind: str = "data") -> Dict[str, torch.Tensor]:
        out = {self.name + ".weight": getattr(self.W, kind)[:, : self.in_f]}
        if self.b is not None:
            out[self.name + ".bias"] = getattr(self.b, kind)
        return out

    def fwd(self, x, *, act=0, residual=None, c_pre=None, split_k=1):
        y = ops.gemm(x, self.Wc, bias=self.bc, act=act, residual=residual, c_pre=c_pre, split_k=split_k)
        return y, (x,)

    def bwd(self, dy, saved, need_dx=True):
        (x,) = saved
        M = dy.shape[0]
        ops.gemm_tn(dy, x, out=self.W.grad)
        if self.b is not None:
            ops.colsum(dy, self.b.grad)
        if not need_dx:
            return None
        if M % 8 != 0:
            raise RuntimeError(f"{self.name}: row count {M} must be a multiple of 8 for the data gradient")
        dyT = ops.transpose(dy)                                               # [out, M]
        return ops.gemm_tn(dyT, self.Wc, accumulate=False, out_dtype=BF16)   # [M, in] = dy . W  (contract over `out`)


# ======================================================================================================================
# ViT (timm VisionTransformer blocks; reference call site modeling_prismatic.py:127-139,186-227)
# ======================================================================================================================
ACT_ID = {"gelu": ops.ACT_GELU, "gelu_tanh": ops.ACT_GELU_TANH}


def lora_predicate(lora, has):
    """`lora` True / False: every adapted Linear / none.  "auto": exactly the Linears whose adapter tensors are in the state dict -- a merged
    checkpoint whose vision backbone still carries its adapters (the reference's FiLM evaluation path re-attaches LoRA to the towers only:
    experiments/robot/openvla_utils.py:311-349) loads with adapters on the towers and none on the decoder."""
    if lora == "auto":
        if has is None:
            raise ValueError('lora="auto" needs the state dict\'s `has`')

        def pred(names):
            flags = [has(n + ".lora_A.weight") for n in names]
            if any(flags) != all(flags):
                raise ValueError(f"LoRA adapters present for only some of the fused linears {names}")
            return flags[0]

        return pred
    return lambda names: bool(lora)


class VitTower:
    def __init__(self, store, prefix: str, vc: VitConfig, get, cfg: VLAConfig, lora, film: bool = False):
        self.vc, self.prefix = vc, prefix
        lora = lora if callable(lora) else lora_predicate(lora, None)
        w = get(prefix + "patch_embed.proj.weight").reshape(vc.dim, -1)                 # [dim, 3*p*p]
        self.patch_w = torch.zeros((vc.dim, vc.patch_k), dtype=BF16, device=w.device)
        self.patch_w[:, : w.shape[1]] = w
        self.patch_b = get(prefix + "patch_embed.proj.bias")
        self.pos = get(prefix + "pos_embed").reshape(vc.n_patches, vc.dim).contiguous()
        pre = []
        if vc.n_prefix > 0:
            pre.append(get(prefix + "cls_token").reshape(1, vc.dim))
            if vc.n_prefix > 1:
                pre.append(get(prefix + "reg_token").reshape(vc.n_prefix - 1, vc.dim))
        self.prefix_tokens = torch.cat(pre, 0).contiguous() if pre else None
        self.act = ACT_ID[vc.act]
        self.blocks = []
        s = cfg.lora_scale

        def L(name, groups=1):
            A = get(name + ".lora_A.weight") if lora([name]) else None
            Bm = get(name + ".lora_B.weight") if lora([name]) else None
            return LoraLinear(store, name, get(name + ".weight"), get(name + ".bias"), A, Bm, groups, s)

        # only blocks 0 .. depth-2 are ever used: the forward returns the output of block depth-2
        # (get_intermediate_layers(n={depth-2})); the last block's output is discarded (film_vit_wrapper.py:124-137)
        self.film = film
        for i in range(vc.depth - 1):
            p = f"{prefix}blocks.{i}."
            fl = None
            if film:   # film_vit_wrapper.py:53-54: fp32 nn.Linear(llm_dim, vision_dim) pairs, trained in full
                fl = (FullLinear(store, p + "scale", get(p + "scale.weight").float(), get(p + "scale.bias").float(), master_fp32=True),
                      FullLinear(store, p + "shift", get(p + "shift.weight").float(), get(p + "shift.bias").float(), master_fp32=True))
            blk = dict(film=fl, ln1_w=get(p + "norm1.weight"), ln1_b=get(p + "norm1.bias"), ln2_w=get(p + "norm2.weight"), ln2_b=get(p + "norm2.bias"),
                       qkv=L(p + "attn.qkv"), proj=L(p + "attn.proj"), fc1=L(p + "mlp.fc1"), fc2=L(p + "mlp.fc2"),
                       ls1=get(p + "ls1.scale_factor") if vc.layerscale else None, ls2=get(p + "ls2.scale_factor") if vc.layerscale else None)
            self.blocks.append(blk)

    def linears(self):
        for b in self.blocks:
            yield from (b["qkv"], b["proj"], b["fc1"], b["fc2"])
            if b["film"] is not None:
                yield from b["film"]

    def export_frozen(self) -> Dict[str, torch.Tensor]:
        """The tower's frozen tensors under their checkpoint names (views, no copies): what `vision_backbone--{step}_checkpoint.pt` carries
        beside the adapters and the FiLM Linears (finetune.py:640-655 saves the whole wrapped backbone)."""
        vc, p = self.vc, self.prefix
        out = {p + "patch_embed.proj.weight": self.patch_w[:, : 3 * vc.patch * vc.patch].reshape(vc.dim, 3, vc.patch, vc.patch),
               p + "patch_embed.proj.bias": self.patch_b, p + "pos_embed": self.pos.reshape(1, vc.n_patches, vc.dim)}
        if self.prefix_tokens is not None:
            out[p + "cls_token"] = self.prefix_tokens[:1].reshape(1, 1, vc.dim)
            if vc.n_prefix > 1:
                out[p + "reg_token"] = self.prefix_tokens[1:].reshape(1, vc.n_prefix - 1, vc.dim)
        for i, b in enumerate(self.blocks):
            q = f"{p}blocks.{i}."
            out.update({q + "norm1.weight": b["ln1_w"], q + "norm1.bias": b["ln1_b"], q + "norm2.weight": b["ln2_w"], q + "norm2.bias": b["ln2_b"]})
            for nm, key in (("attn.qkv", "qkv"), ("attn.proj", "proj"), ("mlp.fc1", "fc1"), ("mlp.fc2", "fc2")):
                if getattr(b[key], "merged", False):
                    raise RuntimeError("the adapters were merged into the base weights: there is no training-time backbone state to export")
                out[q + nm + ".weight"], out[q + nm + ".bias"] = b[key].W, b[key].bias
            if b["ls1"] is not None:
                out[q + "ls1.scale_factor"], out[q + "ls2.scale_factor"] = b["ls1"], b["ls2"]
        return out

    def fwd(self, pixels, c0: int, n_img: int, train: bool, film_avg=None):
        """pixels bf16 [B, 6*n_img, H, W]; image i uses channels [c0 + 6 i, +3).  All images go through the tower as one
        batch of B*n_img (ordered (b, img)).  Returns (tokens [B*n_img*T, dim] after block depth-2, saved)."""
        vc = self.vc
        B = pixels.shape[0] * n_img
        cols = ops.im2col(pixels, c0, vc.patch, vc.patch_k, n_img=n_img, img_cstride=6)
        patches = ops.gemm(cols, self.patch_w, bias=self.patch_b)
        x = ops.vit_embed(patches, self.pos, self.prefix_tokens, B, vc.n_patches, vc.dim)
        T = vc.n_patches + vc.n_prefix
        H, hd = vc.heads, vc.head_dim
        saved = []
        for blk in self.blocks:
            h1, mean1, rstd1 = ops.norm_fwd(x, blk["ln1_w"], blk["ln1_b"], eps=vc.eps, rms=False, save_stats=train)
            qkv, s_qkv = blk["qkv"].fwd(h1)
            o, lse = ops.attn_fwd(qkv[:, : vc.dim], qkv[:, vc.dim: 2 * vc.dim], qkv[:, 2 * vc.dim:], B, T, H, hd)
            fsv = None
            if blk["film"] is not None:
                # x = (x + ls1 * attn(...)) * (1 + gamma) + beta, gamma/beta = Linear(mean language embedding) per SAMPLE
                # (both images of a sample share them): fused into the proj GEMM's epilogue
                gamma, s_sc = blk["film"][0].fwd(film_avg)
                beta, s_sh = blk["film"][1].fwd(film_avg)
                xpre = torch.empty_like(x) if train else None
                x2, s_proj = blk["proj"].fwd(o, residual=x, colscale=blk["ls1"], film=(gamma, beta, n_img * T), c_pre=xpre)
                fsv = (gamma, s_sc, s_sh, xpre, n_img * T)
            else:
                x2, s_proj = blk["proj"].fwd(o, residual=x, colscale=blk["ls1"])
            h2, mean2, rstd2 = ops.norm_fwd(x2, blk["ln2_w"], blk["ln2_b"], eps=vc.eps, rms=False, save_stats=train)
            z = torch.empty((h2.shape[0], vc.mlp_hidden), dtype=BF16, device=h2.device) if train else None
            hmid, s_fc1 = blk["fc1"].fwd(h2, act=self.act, c_pre=z)
            x3, s_fc2 = blk["fc2"].fwd(hmid, residual=x2, colscale=blk["ls2"])
            if train:
                saved.append((x, mean1, rstd1, s_qkv, qkv, o, lse, s_proj, x2, mean2, rstd2, s_fc1, z, s_fc2, fsv))
            x = x3
        return x, (saved, B)

    def bwd(self, dx, saved_all):
        """dx [B*T, dim]: gradient w.r.t. the tower output (prefix rows already zero).  Pixels need no gradient."""
        vc = self.vc
        saved, B = saved_all
        T = vc.n_patches + vc.n_prefix
        H, hd = vc.heads, vc.head_dim
        for bi, (blk, sv) in enumerate(zip(reversed(self.blocks), reversed(saved))):
            first_block = bi == len(saved) - 1      # block 0: nothing upstream is trainable (patch / position embeddings are frozen)
            x, mean1, rstd1, s_qkv, qkv, o, lse, s_proj, x2, mean2, rstd2, s_fc1, z, s_fc2, fsv = sv
            # x3 = x2 + ls2 * fc2(act(fc1(ln2(x2))))
            d = ops.colscale(dx, blk["ls2"]) if blk["ls2"] is not None else dx
            tnq = [] if _GROUP_TN else None   # fc2's and fc1's LoRA weight gradients: four ~9 MB problems in ONE launch, before norm_bwd rewrites dx
            if _FUSE_ACT:
                dz = blk["fc2"].bwd(d, s_fc2, dact=("act", z, self.act), tn_queue=tnq)  # dh * act'(z) in the dgrad GEMM's epilogue
            else:
                dz = ops.act_bwd(z, blk["fc2"].bwd(d, s_fc2, tn_queue=tnq), self.act)
            dh2 = blk["fc1"].bwd(dz, s_fc1, tn_queue=tnq)
            if tnq:
                ops.gemm_tn_grouped(tnq)
            ops.norm_bwd(x2, dh2, blk["ln2_w"], mean2, rstd2, rms=False, dx=dx, dx_accum=True)       # dx now = d x2
            if fsv is not None:   # through the FiLM modulation: dgamma, dbeta, dx <- dx * (1 + gamma)
                gamma, s_sc, s_sh, xpre, frows = fsv
                nb = gamma.shape[0]
                dg = torch.zeros((nb, vc.dim), dtype=F32, device=dx.device)
                db = torch.zeros((nb, vc.dim), dtype=F32, device=dx.device)
                ops.film_bwd(dx, xpre, gamma, dg, db, dx.shape[0] // frows, frows)
                blk["film"][0].bwd(ops.cvt_f32_to_bf16(dg), s_sc, need_dx=False)
                blk["film"][1].bwd(ops.cvt_f32_to_bf16(db), s_sh, need_dx=False)
            # x2 = x + ls1 * proj(attn(qkv(ln1(x))))
            d = ops.colscale(dx, blk["ls1"]) if blk["ls1"] is not None else dx
            tnq = [] if _GROUP_TN else None   # likewise proj + qkv
            do = blk["proj"].bwd(d, s_proj, tn_queue=tnq)
            dqkv = torch.empty_like(qkv)
            D = vc.dim
            ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, do, lse, B, T, H, hd, dq=dqkv[:, :D], dk=dqkv[:, D:2 * D],
                         dv=dqkv[:, 2 * D:])
            dh1 = blk["qkv"].bwd(dqkv, s_qkv, need_dx=not first_block, tn_queue=tnq)   # block 0 still needs its LoRA gradients, not d x
            if tnq:
                ops.gemm_tn_grouped(tnq)
            if not first_block:
                ops.norm_bwd(x, dh1, blk["ln1_w"], mean1, rstd1, rms=False, dx=dx, dx_accum=True)    # dx now = d x
        return None  # patch embedding / position embedding are frozen and the pixels need no gradient


# ======================================================================================================================
# Llama decoder stack (transformers LlamaModel; reference call site modeling_prismatic.py:632-643)
# ======================================================================================================================
class LlamaStack:
    def __init__(self, store, cfg: VLAConfig, get, lora):
        self.cfg = cfg
        lora = lora if callable(lora) else lora_predicate(lora, None)
        D, F, s = cfg.llm_dim, cfg.llm_ff, cfg.lora_scale
        self.layers = []
        for i in range(cfg.llm_layers):
            p = f"language_model.model.layers.{i}."

            def fused(names, sub):
                W = torch.cat([get(p + sub + n + ".weight") for n in names], 0)
                on = lora([p + sub + n for n in names])
                A = torch.cat([get(p + sub + n + ".lora_A.weight") for n in names], 0) if on else None
                Bm = torch.cat([get(p + sub + n + ".lora_B.weight") for n in names], 0) if on else None
                return LoraLinear(store, p + sub + "+".join(names), W, None, A, Bm, len(names), s, ref_names=[p + sub + n for n in names])

            self.layers.append(dict(
                qkv=fused(["q_proj", "k_proj", "v_proj"], "self_attn."), o=fused(["o_proj"], "self_attn."),
                gu=fused(["gate_proj", "up_proj"], "mlp."), down=fused(["down_proj"], "mlp."),
                n1=get(p + "input_layernorm.weight"), n2=get(p + "post_attention_layernorm.weight")))
        self.norm_w = get("language_model.model.norm.weight")
        self.hd = D // cfg.llm_heads
        self.cos = self.sin = None
        self.on_grads_ready = None

    def linears(self):
        for l in self.layers:
            yield from (l["qkv"], l["o"], l["gu"], l["down"])

    # -- RMSNorm folded around the projections (inference on adapter-free weights) ------------------------------------------------------
    def fold_norms(self):
        """`north_star`'s "fused RMSNorm + RoPE + QKV" on the inference path (HF LlamaDecoderLayer: input_layernorm -> q|k|v, post_attention_layernorm
        -> gate|up; call site modeling_prismatic.py:901-912).  y = (w * x * rstd) W^T = rstd * (x (W w)^T): the norm WEIGHT is folded into a second
        copy of the frozen projection weight (columns scaled, rounded to bf16 once, offline), the per-row sum of squares comes out of the PRODUCING
        GEMM's epilogue (o_proj / down_proj with the residual add: ovla_gemm_args.rowsq_out; the first layer's from ovla_row_sumsq) and rstd scales
        the accumulator in the consuming GEMM's epilogue, before the RoPE rotation (rowscale_part) -- no norm kernel, no normalised activations in
        HBM.  Needs weights without live adapters (merged or adapter-free) and costs a second copy of the q|k|v and gate|up weights (+9 GB at 7B).
        The rounding points move (the reference rounds x * rstd and w * (.) to bf16 before the GEMM): tests/test_fullsize_e2e_gpu.py bounds the delta."""
        for l in self.layers:
            for key, nk in (("qkv", "n1"), ("gu", "n2")):
                lin = l[key]
                if lin.has_lora and not getattr(lin, "merged", False):
                    raise RuntimeError("fold_norms: the decoder still carries live LoRA adapters (merge_lora() first)")
                l[key + "_n"] = ops.colscale(lin.W, l[nk])          # bf16(W[n, k] * w[k])
        self.folded = True
        self._fold_plan = {}

    def _swiglu_pair_ok(self, lin) -> bool:
        """OVLA_ACT_SWIGLU's shape rules (ovla.h): F a multiple of 128, K of 64, LoRA rank 32 grouped by F (or merged / absent), no bias."""
        F, D = self.cfg.llm_ff, self.cfg.llm_dim
        live = lin.has_lora and not getattr(lin, "merged", False)
        return F % 128 == 0 and D % 64 == 0 and lin.bias is None and (not live or (lin.r == 32 and lin.groups == 2))

    def _infer_tile(self, M: int) -> int:
        """Tile configuration of the folded inference path's four projections.  A few hundred rows (the batch-1 chunk: M = 608 = 4.75 row tiles of 128): the
        4-wave 128x256 configuration with the hand-scheduled K loop (ovla.h tile 122: -9...-16 % per projection on cold weights, tools/cold_gemm_probe.py);
        otherwise 0 = the planner's choice.  OVLA_INFER_W4=0 switches it off."""
        D, F = self.cfg.llm_dim, self.cfg.llm_ff
        return 122 if _INFER_W4 and 256 < M <= 1024 and D % 256 == 0 and F % 128 == 0 and D <= 4096 and self.hd == 128 else 0

    def _fold_ok(self, M: int) -> bool:
        """The fold runs in the 128x128 and the 4-wave 128x256 GEMM configurations only (ovla.h): usable when all four projections of an M-row forward
        resolve to the former or take the latter."""
        ok = self._fold_plan.get(M)
        if ok is None:
            D, F = self.cfg.llm_dim, self.cfg.llm_ff
            ok = D % 512 == 0 and self.hd == 128 and (self._infer_tile(M) == 122 or all(ops.gemm_plan(M, n, k)[0] == 1 for n, k in ((3 * D, D), (D, D), (2 * F, D), (D, F))))
            self._fold_plan[M] = ok
        return ok

    def _tables(self, S, device):
        if self.cos is None or self.cos.shape[0] < S:
            n = max(S, self.cfg.max_positions)
            self.cos, self.sin = ops.rope_table(n, self.hd, self.cfg.rope_theta, device)
        return self.cos, self.sin

    def fwd(self, x, B: int, S: int, kv_len, train: bool, sel=None):
        """x bf16 [B*S, D] (inputs_embeds) -> (hidden_states[-1] [B*S, D] (post final norm), saved).
        `sel` (int32 [n], flattened (b, s) rows, n % 8 == 0): only these rows of hidden_states[-1] are wanted (the rows that predict
        the action tokens: everything else of the last layer's output is discarded by run_forward_pass / predict_action).  The last
        layer then runs its attention-output projection, MLP and the final norm on those n rows only (K / V still come from every
        row) and the result is [n, D] in `sel` order -- the same numbers, 9 % of the rows for three of its four GEMMs."""
        cfg = self.cfg
        D, H, hd, F = cfg.llm_dim, cfg.llm_heads, self.hd, cfg.llm_ff
        cos, sin = self._tables(S, x.device)
        causal = cfg.mask_mode == "causal"
        saved = []
        if sel is not None and (sel.numel() % 8 != 0 or sel.numel() == 0):
            raise ValueError("LlamaStack.fwd: the number of selected rows must be a positive multiple of 8")
        M = x.shape[0]
        fold = (not train) and getattr(self, "folded", False) and _FOLD_RMSNORM and self._fold_ok(M)
        part = rbuf = None
        itile = 0
        if fold:   # sums of squares of the first layer's input rows; every later layer's come out of the down projection's epilogue
            part = ops.row_sumsq(x)
            rbuf = torch.empty(M, dtype=F32, device=x.device)
            itile = self._infer_tile(M)
        for li, l in enumerate(self.layers):
            last_sel = sel is not None and li == len(self.layers) - 1
            if fold:   # RMSNorm + RoPE + q|k|v in ONE launch: rstd from `part`, folded weight, rotation in the epilogue
                qkv = ops.gemm(x, l["qkv_n"], rope=(cos, sin, S, 2 * D), rowscale=(part, cfg.rms_eps, rbuf), tile=itile)
                s_qkv = r1 = None
            else:
                h1, _, r1 = ops.norm_fwd(x, l["n1"], eps=cfg.rms_eps, rms=True, save_stats=train)
                if _FUSE_ROPE_FWD and hd == 128:    # RoPE on the q | k heads in the projection's epilogue (ovla_gemm_args.rope_*)
                    qkv, s_qkv = l["qkv"].fwd(h1, rope=(cos, sin, S, 2 * D))
                else:
                    qkv, s_qkv = l["qkv"].fwd(h1)
                    ops.rope_(qkv, S, 2 * H, hd, cos, sin)
            o, lse = ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, S, H, hd, kv_len=kv_len, causal=causal)
            if fold and not last_sel:
                part2 = torch.empty((M, D // 64), dtype=F32, device=x.device)
                x2 = ops.gemm(o, l["o"].W, residual=x, rowsq_out=part2, tile=itile)
                if itile == 122:   # RMSNorm + gate|up + SwiGLU in ONE launch (ovla.h: OVLA_ACT_SWIGLU on the 128x256 configuration): the [M, 2F] intermediate never exists
                    hm = ops.gemm(x2, l["gu_n"], rowscale=(part2, cfg.rms_eps, rbuf), tile=22, act=ops.ACT_SWIGLU)
                else:
                    gu = ops.gemm(x2, l["gu_n"], rowscale=(part2, cfg.rms_eps, rbuf), tile=itile)
                    hm = ops.swiglu_fwd(gu)
                part = torch.empty((M, D // 64), dtype=F32, device=x.device)
                x = ops.gemm(hm, l["down"].W, residual=x2, rowsq_out=part, tile=itile)
                continue
            if last_sel:
                x2, s_o = l["o"].fwd(ops.gather_rows(o, sel, D), residual=ops.gather_rows(x, sel, D))
            else:
                x2, s_o = l["o"].fwd(o, residual=x)
            h2, _, r2 = ops.norm_fwd(x2, l["n2"], eps=cfg.rms_eps, rms=True, save_stats=train)
            if _FUSE_SWIGLU_FWD and x2.shape[0] >= 256 and self._swiglu_pair_ok(l["gu"]):
                # gate|up + SwiGLU in ONE launch (ovla.h OVLA_ACT_SWIGLU on the 4-wave 256x256 configuration): h comes out of the projection's read-back;
                # the [M, 2F] projection output is still written (C_pre) when the backward needs it, but never read again in the forward
                gu = torch.empty((x2.shape[0], 2 * F), dtype=BF16, device=x.device) if train else None
                hm, s_gu = l["gu"].fwd(h2, act=ops.ACT_SWIGLU, c_pre=gu)
            else:
                gu, s_gu = l["gu"].fwd(h2)
                hm = ops.swiglu_fwd(gu)
            x3, s_d = l["down"].fwd(hm, residual=x2)
            if train:
                saved.append((x, r1, s_qkv, qkv, o, lse, s_o, x2, r2, s_gu, gu, s_d))
            x = x3
        out, _, rf = ops.norm_fwd(x, self.norm_w, eps=cfg.rms_eps, rms=True, save_stats=train)
        return out, (saved, x, rf, B, S, kv_len, sel)

    def bwd(self, dout, saved_all):
        """dout [B*S, D] gradient of hidden_states[-1] ([n, D] for the selected rows if the forward ran with `sel`) -> gradient of
        inputs_embeds (in a fresh buffer)."""
        cfg = self.cfg
        saved, x_last, rf, B, S, kv_len, sel = saved_all
        D, H, hd, F = cfg.llm_dim, cfg.llm_heads, self.hd, cfg.llm_ff
        causal = cfg.mask_mode == "causal"
        dx = ops.norm_bwd(x_last, dout, self.norm_w, None, rf, rms=True)
        for li, (l, sv) in enumerate(zip(reversed(self.layers), reversed(saved))):
            x, r1, s_qkv, qkv, o, lse, s_o, x2, r2, s_gu, gu, s_d = sv
            if _FUSE_SWIGLU:
                dgu = l["down"].bwd(dx, s_d, dact=("swiglu", gu))        # SwiGLU' in the dgrad GEMM's epilogue: dh never hits HBM
            else:
                dgu = ops.swiglu_bwd(gu, l["down"].bwd(dx, s_d))
            dh2 = l["gu"].bwd(dgu, s_gu)
            ops.norm_bwd(x2, dh2, l["n2"], None, r2, rms=True, dx=dx, dx_accum=True)                   # dx = d x2
            do = l["o"].bwd(dx, s_o)
            if sel is not None and li == 0:   # last layer on the selected rows: back to all rows (zero gradient everywhere else)
                do_full = torch.zeros((B * S, D), dtype=BF16, device=dx.device)
                dx_full = torch.zeros((B * S, D), dtype=BF16, device=dx.device)
                ops.gather_rows(do, sel, D, dst=do_full, scatter_add=True)
                ops.gather_rows(dx, sel, D, dst=dx_full, scatter_add=True)                             # the residual x -> x2 at those rows
                do, dx = do_full, dx_full
            dqkv = torch.empty_like(qkv)
            # dq / dk leave the attention backward already rotated back (inverse RoPE in its epilogues: no separate pass)
            ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, do, lse, B, S, H, hd, kv_len=kv_len, causal=causal,
                         dq=dqkv[:, :D], dk=dqkv[:, D:2 * D], dv=dqkv[:, 2 * D:], rope=(self.cos, self.sin))
            dh1 = l["qkv"].bwd(dqkv, s_qkv)
            ops.norm_bwd(x, dh1, l["n1"], None, r1, rms=True, dx=dx, dx_accum=True)                    # dx = d x
            if self.on_grads_ready is not None:   # this layer's LoRA gradients are final: the reducer may ship them
                self.on_grads_ready(l["qkv"].A)
        return dx


# ======================================================================================================================
# MLPResNet action head (prismatic/models/action_heads.py:38-107; L1 loss finetune.py:400, MSE :407)
# ======================================================================================================================
class ActionHead:
    def __init__(self, store, get, prefix: str, cfg: VLAConfig):
        D, A = cfg.llm_dim, cfg.action_dim
        self.cfg, self.prefix = cfg, prefix
        self.ln1_w = store.add(prefix + "layer_norm1.weight", get(prefix + "layer_norm1.weight"))
        self.ln1_b = store.add(prefix + "layer_norm1.bias", get(prefix + "layer_norm1.bias"))
        self.fc1 = FullLinear(store, prefix + "fc1", get(prefix + "fc1.weight"), get(prefix + "fc1.bias"))
        self.blocks = []
        for b in range(2):
            q = f"{prefix}mlp_resnet_blocks.{b}.ffn."
            self.blocks.append(dict(ln_w=store.add(q + "0.weight", get(q + "0.weight")), ln_b=store.add(q + "0.bias", get(q + "0.bias")),
                                    fc=FullLinear(store, q + "1", get(q + "1.weight"), get(q + "1.bias"))))
        self.ln2_w = store.add(prefix + "layer_norm2.weight", get(prefix + "layer_norm2.weight"))
        self.ln2_b = store.add(prefix + "layer_norm2.bias", get(prefix + "layer_norm2.bias"))
        self.out_w = store.add(prefix + "fc2.weight", get(prefix + "fc2.weight"))
        self.out_b = store.add(prefix + "fc2.bias", get(prefix + "fc2.bias"))

    def linears(self):
        yield self.fc1
        for b in self.blocks:
            yield b["fc"]

    def plain_params(self):
        yield from (self.ln1_w, self.ln1_b, self.ln2_w, self.ln2_b, self.out_w, self.out_b)
        for b in self.blocks:
            yield from (b["ln_w"], b["ln_b"])

    def fwd(self, actions_hidden, target=None, mse=False, train=False):
        """actions_hidden bf16 [B*A_tokens, D] (row-major (b, token)) == reshape(B, chunk, action_dim*D).
        Returns (pred [B*chunk, action_dim], loss_sum fp32[1] or None, saved)."""
        cfg = self.cfg
        rows = actions_hidden.shape[0] // cfg.action_dim
        x0 = actions_hidden.view(rows, cfg.action_dim * cfg.llm_dim)
        rows_real = rows
        if rows % 8:   # e.g. ALOHA: batch 4 x chunk 25; the GEMM-shaped data gradients want M % 8 == 0: zero-pad the rows
            rows = (rows + 7) // 8 * 8
            xp = torch.zeros((rows, x0.shape[1]), dtype=BF16, device=x0.device)
            xp[:rows_real] = x0
            x0 = xp
        h0, m0, r0 = ops.norm_fwd(x0, self.ln1_w.data, self.ln1_b.data, eps=1e-5, rms=False, save_stats=train)
        z1 = torch.empty((rows, cfg.llm_dim), dtype=BF16, device=x0.device) if train else None
        split = 8 if rows <= 256 else 1      # small-M weight stream: split K over the chip
        x, s1 = self.fc1.fwd(h0, act=ops.ACT_RELU, c_pre=z1, split_k=split)
        D = cfg.llm_dim
        if bool(_FUSE_HEAD) and rows <= 64 and D % 64 == 0 and D // 16 <= 256 and (D // 16) % 4 == 0 and \
                _lib.lib().ovla_head_tail_resident_blocks() >= D // 16:      # software grid barriers: only when the whole grid is co-resident
            # everything after fc1 -- both MLPResNet blocks, LayerNorm 2, fc2 and the loss -- is ONE launch (ovla_head_tail_fwd), bit-identical
            # to the unfused sequence below; its backward is the unfused one, fed from the tensors the kernel saves
            R = (rows + 15) // 16 * 16
            if R != rows:
                xp = torch.zeros((R, D), dtype=BF16, device=x.device)
                xp[:rows] = x
                x_in = xp
            else:
                x_in = x
            if getattr(self, "_sync", None) is None:
                self._sync = torch.zeros(2, dtype=torch.int32, device=x.device)
            loss_sum = torch.zeros(1, dtype=F32, device=x0.device) if target is not None else None
            o = ops.head_tail_fwd(x_in, [(b["ln_w"].data, b["ln_b"].data, b["fc"].Wc, b["fc"].bc) for b in self.blocks], (self.ln2_w.data, self.ln2_b.data),
                                  self.out_w.data, self.out_b.data, rows_real=rows_real, target=target, loss_sum=loss_sum, mse=mse, train=train, sync=self._sync)
            saved = None
            if train:
                xs = [x_in[:rows], o["xo"][0][:rows]]
                blocks_saved = [(xs[b], o["mean"][b][:rows], o["rstd"][b][:rows], o["zb"][b][:rows], (o["hb"][b][:rows],)) for b in range(2)]
                saved = (x0, m0, r0, z1, s1, blocks_saved, o["xo"][1][:rows], o["mean2"][:rows], o["rstd2"][:rows], o["h2"][:rows], o["pred"], target, mse, rows_real)
            return o["pred"], loss_sum, saved
        blocks_saved = []
        for b in self.blocks:
            hb, mb, rb = ops.norm_fwd(x, b["ln_w"].data, b["ln_b"].data, eps=1e-5, rms=False, save_stats=train)
            # x_new = relu(fc(ln(x))) + x : epilogue applies the activation before the residual add
            zb = torch.empty_like(x) if train else None
            xn, sb = b["fc"].fwd(hb, act=ops.ACT_RELU, residual=x, c_pre=zb, split_k=2 if rows <= 256 else 1)
            blocks_saved.append((x, mb, rb, zb, sb))
            x = xn
        h2, m2, r2 = ops.norm_fwd(x, self.ln2_w.data, self.ln2_b.data, eps=1e-5, rms=False, save_stats=train)
        loss_sum = torch.zeros(1, dtype=F32, device=x0.device) if target is not None else None
        pred = ops.head_out_fwd(h2[:rows_real], self.out_w.data, self.out_b.data, target, loss_sum, mse=mse)
        saved = (x0, m0, r0, z1, s1, blocks_saved, x, m2, r2, h2, pred, target, mse, rows_real) if train else None
        return pred, loss_sum, saved

    def check_fused_tail(self):
        """Host-side check of the fused tail's sticky timeout word (ovla_head_tail_args.sync[1]): call where the step already synchronises
        (loss.item()).  A timed-out grid barrier has already turned the loss, the predictions and every saved buffer into NaN; this turns it
        into an exception and clears the word."""
        sync = getattr(self, "_sync", None)
        if sync is not None and int(sync[1].item()) != 0:
            sync[1] = 0
            raise RuntimeError("ovla_head_tail_fwd: a grid barrier timed out (the workgroups were not co-resident); the step's head outputs are NaN. "
                               "Set OVLA_FUSE_HEAD=0 to use the unfused sequence.")

    def bwd(self, saved, dloss: float = 1.0, dpred=None):
        """Returns d(actions_hidden) [B*A_tokens, D]; accumulates the head's gradients.  With `dpred` (bf16
        [rows, action_dim]) the upstream gradient is taken as given (loss computed by the caller); otherwise the fused
        L1 / MSE gradient of mean-reduced loss * dloss is used."""
        cfg = self.cfg
        x0, m0, r0, z1, s1, blocks_saved, xl, m2, r2, h2, pred, target, mse, rows_real = saved
        rows = x0.shape[0]
        dh2 = ops.head_out_bwd(h2[:rows_real], self.out_w.data, pred, target, dloss / (rows_real * cfg.action_dim), self.out_w.grad, self.out_b.grad,
                               mse=mse, dpred=dpred)
        if rows != rows_real:
            dp = torch.zeros((rows, h2.shape[1]), dtype=BF16, device=h2.device)
            dp[:rows_real] = dh2
            dh2 = dp
        dx = ops.norm_bwd(xl, dh2, self.ln2_w.data, m2, r2, rms=False, dweight=self.ln2_w.grad, dbias=self.ln2_b.grad)
        for b, (xin, mb, rb, zb, sb) in zip(reversed(self.blocks), reversed(blocks_saved)):
            dz = ops.act_bwd(zb, dx, ops.ACT_RELU)
            dhb = b["fc"].bwd(dz, sb)
            ops.norm_bwd(xin, dhb, b["ln_w"].data, mb, rb, rms=False, dx=dx, dx_accum=True, dweight=b["ln_w"].grad, dbias=b["ln_b"].grad)
        dz1 = ops.act_bwd(z1, dx, ops.ACT_RELU)
        dh0 = self.fc1.bwd(dz1, s1)
        dx0 = ops.norm_bwd(x0, dh0, self.ln1_w.data, m0, r0, rms=False, dweight=self.ln1_w.grad, dbias=self.ln1_b.grad)
        return dx0[:rows_real].reshape(rows_real * cfg.action_dim, cfg.llm_dim)


class MlpProjector:
    """fc1 -> GELU -> fc2 with fully trainable fp32 parameters (ProprioProjector / NoisyActionProjector,
    prismatic/models/projectors.py:6-49)."""

    def __init__(self, store, get, prefix: str):
        self.prefix = prefix
        self.fc1 = FullLinear(store, prefix + "fc1", get(prefix + "fc1.weight").float(), get(prefix + "fc1.bias").float(), master_fp32=True)
        self.fc2 = FullLinear(store, prefix + "fc2", get(prefix + "fc2.weight").float(), get(prefix + "fc2.bias").float(), master_fp32=True)

    def linears(self):
        return (self.fc1, self.fc2)

    def fwd(self, x, train: bool):
        """x bf16 [rows, in] (rows padded to a multiple of 8 by the caller)."""
        if x.shape[1] != self.fc1.in_pad:
            xp = torch.zeros((x.shape[0], self.fc1.in_pad), dtype=BF16, device=x.device)
            xp[:, : x.shape[1]] = x
            x = xp
        z = torch.empty((x.shape[0], self.fc1.out_f), dtype=BF16, device=x.device) if train else None
        h, s1 = self.fc1.fwd(x, act=ops.ACT_GELU, c_pre=z)
        y, s2 = self.fc2.fwd(h)
        return y, (s1, z, s2)

    def bwd(self, dy, saved):
        s1, z, s2 = saved
        dh = self.fc2.bwd(dy, s2)
        dz = ops.act_bwd(z, dh, ops.ACT_GELU)
        self.fc1.bwd(dz, s1, need_dx=False)


def build_component(cls, device, get, prefix, **kw):
    """Builds a stand-alone component (its own ParamStore) from reference-named tensors."""
    st = ParamStore(device)
    comp = cls(st, get=get, prefix=prefix, **kw)
    st.finalize()
    comp.store = st
    for lin in comp.linears():
        lin.refresh_derived()
    return comp


# ======================================================================================================================
# the whole path
# ======================================================================================================================
class VLAEngine:
    """Prismatic VLM stack + heads.  `get(name)` returns the bf16 device tensor of a reference-named parameter
    (see weights.py for the state-dict naming)."""

    def __init__(self, cfg: VLAConfig, get, device, *, lora: bool = True, use_proprio: bool = True, head: str = "l1", use_film: bool = False,
                 has=None):
        if head not in ("l1", "diffusion", "none"):
            raise ValueError(head)
        ops.check_device(device.index or 0)
        self.cfg, self.device, self.lora = cfg, device, lora
        lora = lora_predicate(lora, has)
        st = self.store = ParamStore(device)
        # registration order = forward order (the store reverses it into backward order)
        self.use_film = use_film
        self.dino = VitTower(st, "vision_backbone.featurizer.", cfg.dino, get, cfg, lora, film=use_film)
        self.siglip = VitTower(st, "vision_backbone.fused_featurizer.", cfg.siglip, get, cfg, lora, film=use_film)
        s = cfg.lora_scale

        def L(name):
            return LoraLinear(st, name, get(name + ".weight"), get(name + ".bias"), get(name + ".lora_A.weight") if lora([name]) else None,
                              get(name + ".lora_B.weight") if lora([name]) else None, 1, s)

        self.proj = [L("projector.fc1"), L("projector.fc2"), L("projector.fc3")]
        self.embed = get("language_model.model.embed_tokens.weight")
        self.llm = LlamaStack(st, cfg, get, lora)
        self.lm_head = get("language_model.lm_head.weight") if (has is None or has("language_model.lm_head.weight")) else None
        st.finalize()
        # components are separate modules with their own parameter stores, like the reference's DDP-wrapped
        # ProprioProjector / action head / NoisyActionProjector (finetune.py:894-932); they can also be built standalone
        # (modeling.py) and passed in per call.
        self.proprio = self.noisy = self.head = None
        self.head_kind = head
        if use_proprio:
            self.proprio = build_component(MlpProjector, device, get, "proprio_projector.")
        if head == "diffusion":
            self.noisy = build_component(MlpProjector, device, get, "noisy_action_projector.")
        if head != "none":
            self.head = build_component(ActionHead, device, get, "action_head.noise_predictor.mlp_resnet." if head == "diffusion" else "action_head.model.",
                                        cfg=cfg)
        self.refresh_derived()
        if _FOLD_RMSNORM and not any(l.has_lora for l in self.llm.linears()):
            self.llm.fold_norms()       # an adapter-free decoder (merged checkpoint): inference-only model, fold the RMSNorms (LlamaStack.fold_norms)

    @property
    def stores(self):
        return [self.store] + [m.store for m in (self.proprio, self.noisy, self.head) if m is not None]

    def zero_grad(self):
        for st in self.stores:
            st.zero_grad()
        if getattr(self, "reducer", None) is not None:
            self.reducer.reset()

    def adamw_step(self, lr: float, **kw):
        for st in self.stores:
            st.adamw_step(lr, **kw)

    def num_trainable(self):
        return sum(st.num_trainable() for st in self.stores)

    # -- bookkeeping -----------------------------------------------------------------------------------------------------
    def all_linears(self):
        yield from self.dino.linears()
        yield from self.siglip.linears()
        yield from self.proj
        for m in (self.proprio, self.noisy, self.head):
            if m is not None:
                yield from m.linears()
        yield from self.llm.linears()

    def merge_lora(self):
        """Folds every LoRA adapter of the VLM into its base weight (what the reference does before deployment:
        merge_lora_weights_and_save.py).  Afterwards forward() runs plain GEMMs; training entry points raise."""
        for lin in self.vlm_linears():
            if isinstance(lin, LoraLinear):
                lin.merge()
        self.lora_merged = True
        if _FOLD_RMSNORM:
            self.llm.fold_norms()       # adapter-free decoder: RMSNorm + RoPE + q|k|v (and RMSNorm + gate|up) become one launch each at inference

    def merged_state_dict(self) -> Dict[str, torch.Tensor]:
        """Base VLM weights under the reference's HF key layout (un-fused q/k/v, gate/up), after merge_lora(): what
        merge_lora_weights_and_save.py writes with `save_pretrained`."""
        out: Dict[str, torch.Tensor] = {}
        for lin in self.vlm_linears():
            if not isinstance(lin, LoraLinear):
                continue
            gn = lin.group_n
            for g, rn in enumerate(lin.ref_names):
                out[rn + ".weight"] = lin.W[g * gn:(g + 1) * gn]
                if lin.bias is not None:
                    out[rn + ".bias"] = lin.bias[g * gn:(g + 1) * gn]
        return out

    def vlm_linears(self):
        yield from self.dino.linears()
        yield from self.siglip.linears()
        yield from self.proj
        yield from self.llm.linears()

    def refresh_derived(self):
        """Re-derives A^T / B^T / bf16 compute copies from the trainable tensors (after every optimizer step): all LoRA
        transposes of the VLM go in ONE batched launch (their addresses never change, so the descriptor table is built once)."""
        if getattr(self, "_ttab", None) is None:
            pairs = [pr for lin in self.vlm_linears() for pr in lin.derived_pairs()]
            self._ttab = ops.transpose_table(pairs, self.device) if pairs else False
        if self._ttab:
            ops.transpose_batched(self._ttab)
        for lin in self.vlm_linears():
            if isinstance(lin, FullLinear):   # FiLM scale / shift: refresh the bf16 compute copies of the fp32 masters
                lin.refresh_derived()
        for m in (self.proprio, self.noisy, self.head):
            if m is not None:
                for lin in m.linears():
                    lin.refresh_derived()

    def vision_backbone_state_dict(self) -> Dict[str, torch.Tensor]:
        """State dict of the (FiLM-wrapped, LoRA-adapted) vision backbone in the reference's key layout (weights.vision_backbone_keys_to_reference):
        frozen tower tensors + the towers' adapters + the FiLM scale / shift Linears -- the content of `vision_backbone--{step}_checkpoint.pt`
        (vla-scripts/finetune.py:640-655), which `get_vla(cfg)` with `cfg.use_film` loads back (experiments/robot/openvla_utils.py:311-349).
        The discarded last block of each tower is not held by this engine and is absent."""
        from .weights import vision_backbone_keys_to_reference

        sd = {**self.dino.export_frozen(), **self.siglip.export_frozen()}
        sd.update({k: v for k, v in self.export_trainable("data").items() if k.startswith("vision_backbone.")})
        return vision_backbone_keys_to_reference(sd)

    def export_trainable(self, kind: str = "data") -> Dict[str, torch.Tensor]:
        """Every trainable tensor (or its fp32 gradient) keyed by the reference's parameter names: LoRA adapters as
        `<linear>.lora_A.weight` / `.lora_B.weight` (un-fused per q/k/v, gate/up), components as in their state dicts
        (prismatic/models/action_heads.py, projectors.py; finetune.py:584-675)."""
        out: Dict[str, torch.Tensor] = {}
        for lin in self.all_linears():
            out.update(lin.export(kind))
        if self.head is not None:
            for p in self.head.plain_params():
                out[p.name] = getattr(p, kind)
        return out

    def load_trainable(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> List[str]:
        """Inverse of export_trainable("data"): copies reference-named tensors (lora adapters, action head, projectors, FiLM)
        into the flat parameter buffers and re-derives the transposes / bf16 compute copies.  Returns the names not found in
        `sd` (raises instead when strict).  finetune.py:134-156,193-209 (resume)."""
        missing = []
        for name, view in self.export_trainable("data").items():
            if name in sd:
                src = sd[name]
                if tuple(src.shape) != tuple(view.shape):
                    raise ValueError(f"{name}: checkpoint shape {tuple(src.shape)} != model shape {tuple(view.shape)}")
                view.copy_(src.to(view.device, view.dtype))
            else:
                missing.append(name)
        if missing and strict:
            raise KeyError(f"{len(missing)} trainable tensors missing from the checkpoint, e.g. {missing[:3]}")
        self.refresh_derived()
        return missing

    def optimizer_state_dict(self) -> Dict[str, torch.Tensor]:
        """AdamW moments + step of every parameter store as flat tensors (the layout is a pure function of the config).  The
        reference does not save optimizer state (finetune.py:584-675): this is the SURVEY.md 8f(3) extension."""
        out: Dict[str, torch.Tensor] = {}
        for i, st in enumerate(self.stores):
            if not st.exp_avg:
                st.init_optimizer()
            out[f"store{i}.step"] = torch.tensor([st.step], dtype=torch.int64)
            for dt in st.flat:
                tag = "bf16" if dt == BF16 else "f32"
                out[f"store{i}.{tag}.exp_avg"] = st.exp_avg[dt]
                out[f"store{i}.{tag}.exp_avg_sq"] = st.exp_avg_sq[dt]
        return out

    def load_optimizer_state_dict(self, sd: Dict[str, torch.Tensor]):
        for i, st in enumerate(self.stores):
            if not st.exp_avg:
                st.init_optimizer()
            st.step = int(sd[f"store{i}.step"].item())
            for dt in st.flat:
                tag = "bf16" if dt == BF16 else "f32"
                for kind, buf in (("exp_avg", st.exp_avg[dt]), ("exp_avg_sq", st.exp_avg_sq[dt])):
                    src = sd[f"store{i}.{tag}.{kind}"]
                    if src.numel() != buf.numel():
                        raise ValueError(f"optimizer state store{i}.{tag}.{kind}: {src.numel()} elements, model has {buf.numel()}")
                    buf.copy_(src.to(buf.device, buf.dtype))

    def num_patches_total(self, num_images: int, use_proprio: bool, use_diffusion: bool = False) -> int:
        """NUM_PATCHES of finetune.py:935-941."""
        return self.cfg.dino.n_patches * num_images + int(use_proprio) + int(use_diffusion)

    # -- forward pieces ----------------------------------------------------------------------------------------------------
    def vision_fwd(self, pixel_values, train: bool, film_avg=None):
        """pixel_values bf16 [B, 6*I, H, W] -> projected patches bf16 [B, I*Np, D] rows, saved.
        modeling_prismatic.py:186-227 + :250-262."""
        cfg = self.cfg
        B, C = pixel_values.shape[0], pixel_values.shape[1]
        I = C // 6
        Np, vd = cfg.dino.n_patches, cfg.vision_dim
        feats = torch.empty((B, I * Np, vd), dtype=BF16, device=self.device)   # == [(B*I), Np, vd]: (b, img) is the tower batch
        tower_saved = [None, None]

        def run_tower(k, tower, c0, col0):
            vc = tower.vc
            T = vc.n_patches + vc.n_prefix
            tok, sv = tower.fwd(pixel_values, c0, I, train, film_avg)
            # drop prefix tokens, concat features on dim 2 and images on dim 1 (modeling_prismatic.py:221-227)
            ops.copy_rows(tok, feats, B * I, Np, vc.dim, src_batch_stride=T * vc.dim, src_row0=vc.n_prefix, src_ld=vc.dim,
                          dst_batch_stride=Np * vd, dst_row0=0, dst_ld=vd, dst_col0=col0)
            tower_saved[k] = sv

        # The towers are independent until the feature concat: SigLIP runs on a second HIP stream beside DINOv2, so the
        # tails of one tower's mid-sized GEMMs and its latency-bound small kernels overlap with the other's work.
        side = self._side_stream()
        if side is not None:
            main = torch.cuda.current_stream(self.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                run_tower(1, self.siglip, 3, cfg.dino.dim)
            run_tower(0, self.dino, 0, 0)
            main.wait_stream(side)
        else:
            run_tower(0, self.dino, 0, 0)
            run_tower(1, self.siglip, 3, cfg.dino.dim)
        f2 = feats.view(B * I * Np, vd)
        z1 = torch.empty((f2.shape[0], 4 * vd), dtype=BF16, device=self.device) if train else None
        h1, s1 = self.proj[0].fwd(f2, act=ops.ACT_GELU, c_pre=z1)
        z2 = torch.empty((f2.shape[0], cfg.llm_dim), dtype=BF16, device=self.device) if train else None
        h2, s2 = self.proj[1].fwd(h1, act=ops.ACT_GELU, c_pre=z2)
        out, s3 = self.proj[2].fwd(h2)
        return out.view(B, I * Np, cfg.llm_dim), (tower_saved, s1, z1, s2, z2, s3, B, I)

    def vision_bwd(self, dpatches, saved):
        """dpatches bf16 [B*I*Np, D] contiguous."""
        cfg = self.cfg
        tower_saved, s1, z1, s2, z2, s3, B, I = saved
        Np, vd = cfg.dino.n_patches, cfg.vision_dim
        if _FUSE_ACT:
            d = self.proj[2].bwd(dpatches, s3, dact=("act", z2, ops.ACT_GELU))    # d z2 = d h2 * gelu'(z2), in the dgrad epilogue
            d = self.proj[1].bwd(d, s2, dact=("act", z1, ops.ACT_GELU))           # d z1
        else:
            d = ops.act_bwd(z2, self.proj[2].bwd(dpatches, s3), ops.ACT_GELU)
            d = ops.act_bwd(z1, self.proj[1].bwd(d, s2), ops.ACT_GELU)
        dfeat = self.proj[0].bwd(d, s1)                                           # [B*I*Np, vd]
        def run_tower(k, tower, col0):
            vc = tower.vc
            T = vc.n_patches + vc.n_prefix
            dtok = torch.zeros((B * I * T, vc.dim), dtype=BF16, device=self.device)
            # inverse of the feature concat: rows [n_prefix, T) of tower batch (b, img) <- columns [col0, col0+dim)
            ops.copy_rows(dfeat[:, col0: col0 + vc.dim], dtok, B * I, Np, vc.dim, src_batch_stride=Np * vd, src_row0=0, src_ld=vd,
                          dst_batch_stride=T * vc.dim, dst_row0=vc.n_prefix, dst_ld=vc.dim)
            tower.bwd(dtok, tower_saved[k])

        side = self._side_stream()
        if side is not None:   # same two-stream split as the forward; the towers' gradients live in disjoint buffer ranges
            main = torch.cuda.current_stream(self.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                run_tower(1, self.siglip, cfg.dino.dim)
            run_tower(0, self.dino, 0)
            main.wait_stream(side)   # before dfeat is released and before the optimizer / gradient reducer read the buffers
        else:
            run_tower(1, self.siglip, cfg.dino.dim)
            run_tower(0, self.dino, 0)

    def _side_stream(self):
        """Second HIP stream for the SigLIP tower (OVLA_VIT_STREAMS=1 disables the split)."""
        if os.environ.get("OVLA_VIT_STREAMS", "2") == "1":
            return None
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    # -- the training step pieces -------------------------------------------------------------------------------------------
    def language_average(self, ids_dev, lab_dev):
        """FiLM conditioning vector: mean of the token embeddings at every NON-action position of the text (BOS, prompt,
        stop AND pad tokens: modeling_prismatic.py:581-583 averages `input_embeddings[~all_actions_mask]`;
        film_vit_wrapper.py:243).  One kernel on device tensors (ids / labels int64 [B, L]): no host round trip, so the FiLM forward
        can be captured in a hipGraph.  Returned padded to a multiple of 8 rows (GEMM M granularity of the FiLM Linears)."""
        B = ids_dev.shape[0]
        avg = torch.zeros(((B + 7) // 8 * 8, self.cfg.llm_dim), dtype=BF16, device=self.device)
        return ops.language_average(ids_dev, lab_dev, self.embed, avg)

    def forward(self, input_ids, attention_mask, pixel_values, labels, proprio=None, noisy_actions=None, timestep_emb=None, train=False,
                proprio_projector=None, noisy_action_projector=None, cached_patches=None, sel=None):
        """Multimodal forward (modeling_prismatic.py:571-643 without the discarded lm_head/CE in L1/diffusion mode).
        Returns dict(hidden [B,S,D], P, action_rows [B,A], patches, saved).  `cached_patches` (the `patches` of a previous
        call) skips the vision towers / projector / proprio projector: the DDIM sampler reuses them across its steps
        (modeling_prismatic.py:810).  `sel="actions"` (or an int32 tensor of flattened rows): only those rows of the last hidden state
        are computed (LlamaStack.fwd) and returned as `action_hidden` [n, D]; `hidden` is then None."""
        dev = self.device
        B, L = input_ids.shape
        lens = self.check_right_padding(attention_mask)
        if input_ids.is_cuda or labels.is_cuda or not _ASYNC_UPLOAD:
            ids = input_ids.to(dev, torch.int64).contiguous()
            lab = labels.to(dev, torch.int64).contiguous()
            lens_dev = lens.to(torch.int32).to(dev)
        else:
            # ONE asynchronous upload per step from pinned memory (ids | labels | lengths packed): a pageable `.to(device)` is a synchronous copy
            # queued behind the whole previous step, so the host re-joined the GPU at every step boundary and started the ~3500 launches of the next
            # step from zero lead -- during the towers' 10-25 us kernels the GPU then ran at the host's launch rate.  PyTorch's caching host
            # allocator keeps the pinned block alive until the copy has completed.
            packed = torch.cat([input_ids.reshape(-1).to(torch.int64), labels.reshape(-1).to(torch.int64), lens.reshape(-1).to(torch.int64)]).pin_memory()
            packed_dev = packed.to(dev, non_blocking=True)
            ids, lab = packed_dev[: B * L].view(B, L), packed_dev[B * L: 2 * B * L].view(B, L)
            lens_dev = packed_dev[2 * B * L:].to(torch.int32)
        return self.forward_dev(ids, lab, lens_dev, pixel_values, proprio=proprio, noisy_actions=noisy_actions,
                                timestep_emb=timestep_emb, train=train, proprio_projector=proprio_projector,
                                noisy_action_projector=noisy_action_projector, cached_patches=cached_patches, sel=sel)

    @staticmethod
    def check_right_padding(attention_mask) -> torch.Tensor:
        """Host-side validation of the collator's mask; returns the per-row text lengths (int64, CPU)."""
        am = attention_mask.to("cpu").bool()
        L = am.shape[1]
        lens = am.sum(1)
        if not bool((am == (torch.arange(L)[None, :] < lens[:, None])).all()):
            raise ValueError("attention_mask must be right padding (a prefix of ones per row), as produced by the reference collator")
        return lens

    def forward_dev(self, ids, lab, text_lens, pixel_values, proprio=None, noisy_actions=None, timestep_emb=None, train=False,
                    proprio_projector=None, noisy_action_projector=None, cached_patches=None, film_avg=None, sel=None):
        """Device-only part of forward(): ids / lab int64 [B, L] and text_lens int32 [B] already on the device; launches
        kernels and allocates, never synchronises or reads host memory -- the part a hipGraph can capture (ChunkGraph)."""
        cfg = self.cfg
        dev = self.device
        B, L = ids.shape
        proprio_projector = proprio_projector if proprio_projector is not None else self.proprio
        noisy_action_projector = noisy_action_projector if noisy_action_projector is not None else self.noisy
        vsaved = psaved = nsaved = None
        if cached_patches is not None:
            base, n_vis = cached_patches
        else:
            if self.use_film and film_avg is None:
                film_avg = self.language_average(ids, lab)
            patches, vsaved = self.vision_fwd(pixel_values.to(dev, BF16).contiguous(), train, film_avg)
            n_vis = patches.shape[1]
            base = patches
            if proprio is not None and proprio_projector is not None:
                pr = torch.zeros(((B + 7) // 8 * 8, cfg.proprio_dim), dtype=BF16, device=dev)
                pr[:B] = proprio.reshape(B, -1).to(dev, BF16)
                pf, psaved = proprio_projector.fwd(pr, train)
                base = torch.empty((B, n_vis + 1, cfg.llm_dim), dtype=BF16, device=dev)   # token concat (pure data movement)
                base[:, :n_vis] = patches
                base[:, n_vis] = pf[:B]
        allp = base
        if timestep_emb is not None:
            allp = torch.empty((B, base.shape[1] + 1, cfg.llm_dim), dtype=BF16, device=dev)
            allp[:, : base.shape[1]] = base
            allp[:, base.shape[1]] = timestep_emb.to(dev, BF16).reshape(B, cfg.llm_dim)
        P = allp.shape[1]
        A = cfg.num_action_tokens
        noisy_feats = None
        if noisy_actions is not None:
            rows = (B * A + 7) // 8 * 8
            na = torch.zeros((rows, 1), dtype=BF16, device=dev)
            na[: B * A] = noisy_actions.reshape(B * A, 1).to(dev, BF16)
            nf, nsaved = noisy_action_projector.fwd(na, train)
            noisy_feats = nf[: B * A].view(B, A, cfg.llm_dim)
        mm, action_rows = ops.assemble_multimodal(ids, lab, self.embed, allp.contiguous(), A=A, noisy=noisy_feats, action_dim=cfg.action_dim)
        S = P + L
        kv_len = (text_lens + P).to(torch.int32)
        sel_rows = None
        if sel is not None and os.environ.get("OVLA_LAST_LAYER_SEL", "1") != "0":    # A/B switch
            sel_rows = action_rows.reshape(-1) if isinstance(sel, str) else sel
            if sel_rows.numel() % 8 != 0:
                sel_rows = None          # (e.g. ALOHA discrete: 4 x 351 rows) -> the full last layer
        hidden, lsaved = self.llm.fwd(mm.view(B * S, cfg.llm_dim), B, S, kv_len, train, sel=sel_rows)
        saved = (vsaved, psaved, nsaved, lsaved, B, S, P, n_vis, proprio_projector, noisy_action_projector, action_rows) if train else None
        # `all_patches` [B, P, D]: what sits between BOS and the text in the multimodal sequence (projected patches + proprio + timestep
        # tokens) = the reference's `projector_features` (modeling_prismatic.py:586-599, 674)
        if sel_rows is not None:
            return dict(hidden=None, action_hidden=hidden, sel_rows=sel_rows, P=P, S=S, action_rows=action_rows, patches=(base, n_vis), saved=saved,
                        all_patches=allp)
        return dict(hidden=hidden.view(B, S, cfg.llm_dim), action_hidden=None, sel_rows=None, P=P, S=S, action_rows=action_rows, patches=(base, n_vis),
                    saved=saved, all_patches=allp)

    def action_hidden(self, out):
        """(hidden rows that predict the action slots [B*A, D], their flattened row indices) of a forward() result, whether the last
        layer ran on the selected rows only (`sel="actions"`) or on all of them."""
        if out.get("action_hidden") is not None:
            return out["action_hidden"], out["sel_rows"]
        return self.gather_action_hidden(out["hidden"], out["action_rows"])

    def gather_action_hidden(self, hidden, action_rows):
        """Rows of hidden that predict the action slots: the hidden state at token i-1 predicts token i
        (finetune.py:385-394: text_hidden = last_hidden[:, P:-1] indexed with masks built from labels[:, 1:];
        modeling_prismatic.py:915-920).  `action_rows` (flattened (b, s) row of slot - 1) comes from the assembly kernel."""
        B, S, D = hidden.shape
        idx = action_rows.reshape(-1)
        return ops.gather_rows(hidden.view(B * S, D), idx, D), idx

    def backward_from_hidden(self, dhidden, saved):
        """dhidden bf16 [B*S, D] (gradient of hidden_states[-1]; [n, D] for the selected rows when the forward ran with `sel`) ->
        accumulates every VLM / projector gradient."""
        vsaved, psaved, nsaved, lsaved, B, S, P, n_vis, proprio_projector, noisy_action_projector, action_rows = saved
        D = self.cfg.llm_dim
        dmm_flat = self.llm.bwd(dhidden, lsaved)
        dmm = dmm_flat.view(B, S, D)
        if nsaved is not None:
            # the projected noisy-action features sit AT the action slots: one row after the row that predicts them
            slot_rows = (action_rows.reshape(-1) + 1).contiguous()
            n = slot_rows.numel()
            dn = torch.zeros(((n + 7) // 8 * 8, D), dtype=BF16, device=self.device)
            ops.gather_rows(dmm_flat, slot_rows, D, dst=dn)
            noisy_action_projector.bwd(dn, nsaved)
        if psaved is not None:
            dpr = torch.zeros(((B + 7) // 8 * 8, D), dtype=BF16, device=self.device)
            dpr[:B] = dmm[:, 1 + n_vis]
            proprio_projector.bwd(dpr, psaved)
        dpatches = dmm[:, 1: 1 + n_vis].contiguous().view(B * n_vis, D)
        self.vision_bwd(dpatches, vsaved)

    def attach_reducer(self, reducer, overlap: bool = True):
        """Overlaps the data-parallel gradient all-reduce with the backward: the head's gradients go out as soon as its
        backward is done, each Llama layer's LoRA gradients as soon as that layer is done (the flat buffers are in
        completion order), the rest (projector, ViT towers, proprio projector) at the end of the step."""
        self.reducer = reducer
        self._overlap = overlap and reducer is not None
        if not self._overlap:
            self.llm.on_grads_ready = None
            return

        def llm_layer_done(first_param):   # first-registered param of the layer = LAST in the reversed layout
            reducer.notify(self.store, BF16, first_param.offset + (first_param.numel + ParamStore.ALIGN - 1) // ParamStore.ALIGN * ParamStore.ALIGN)

        self.llm.on_grads_ready = llm_layer_done

    def train_step_fwd_bwd(self, batch: dict, loss_scale: float = 1.0, action_head=None, proprio_projector=None, diffusion=None,
                           noisy_action_projector=None):
        """One run_forward_pass (finetune.py:280-451) + backward.  L1 branch by default; with
        `diffusion = dict(noise, noisy_actions, timestep_emb)` the noise-prediction MSE branch (:402-407).
        Gradients accumulate in the flat buffers.  Returns (loss_sum fp32[1] device, element count, predictions)."""
        cfg = self.cfg
        head = action_head if action_head is not None else self.head
        kw = {}
        if diffusion is not None:
            kw = dict(noisy_actions=diffusion["noisy_actions"], timestep_emb=diffusion["timestep_emb"], noisy_action_projector=noisy_action_projector)
        out = self.forward(batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["labels"], proprio=batch.get("proprio"),
                           train=True, proprio_projector=proprio_projector, sel="actions", **kw)
        B, S, D = batch["input_ids"].shape[0], out["S"], cfg.llm_dim
        ah, idx = self.action_hidden(out)
        tgt_src = diffusion["noise"] if diffusion is not None else batch["actions"]
        target = tgt_src.to(self.device, BF16).reshape(B * cfg.chunk, cfg.action_dim).contiguous()
        pred, loss_sum, hsaved = head.fwd(ah, target=target, mse=diffusion is not None, train=True)
        # ---- backward ----
        dah = head.bwd(hsaved, dloss=loss_scale)
        if getattr(self, "_overlap", False) and hasattr(head, "store"):
            for dt, g in head.store.flat_grad.items():
                self.reducer.notify(head.store, dt, g.numel())
        if out["sel_rows"] is not None:
            self.backward_from_hidden(dah, out["saved"])            # the last layer ran on exactly these rows
        else:
            dhidden = torch.zeros((B * S, D), dtype=BF16, device=self.device)
            ops.gather_rows(dah, idx, D, dst=dhidden, scatter_add=True)
            self.backward_from_hidden(dhidden, out["saved"])
        return loss_sum, pred.numel(), pred


    def eval_step(self, batch: dict, diffusion=None, discrete: bool = False):
        """run_forward_pass under torch.no_grad() (run_validation, finetune.py:678-760): the same loss as the training step,
        no activations kept, no gradients touched.  Returns (loss_sum fp32[1] device, count, predictions)."""
        if discrete:
            return self.train_step_discrete(batch, backward=False)
        cfg = self.cfg
        kw = {}
        if diffusion is not None:
            kw = dict(noisy_actions=diffusion["noisy_actions"], timestep_emb=diffusion["timestep_emb"])
        out = self.forward(batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["labels"], proprio=batch.get("proprio"),
                           train=False, sel="actions", **kw)
        B = batch["input_ids"].shape[0]
        ah, _ = self.action_hidden(out)
        tgt_src = diffusion["noise"] if diffusion is not None else batch["actions"]
        target = tgt_src.to(self.device, BF16).reshape(B * cfg.chunk, cfg.action_dim).contiguous()
        pred, loss_sum, _ = self.head.fwd(ah, target=target, mse=diffusion is not None, train=False)
        return loss_sum, pred.numel(), pred

    def train_step_discrete(self, batch: dict, loss_scale: float = 1.0, proprio_projector=None, backward: bool = True):
        """run_forward_pass's discrete branch (finetune.py:357-378) + backward: `loss = output.loss`, the LlamaForCausalLM
        next-token cross entropy over the multimodal labels (logits.float(), shift by one, ignore_index -100, mean), with the
        frozen lm_head applied ONLY to the rows whose shifted label counts (the action tokens and the stop token: A + 1 rows per
        sample instead of all S).  Returns (loss_sum fp32[1] device, token count, predicted ids int64 [B, L - 1] on the host
        layout of `output.logits[:, num_patches:-1].argmax(2)` with -1 where no row was evaluated)."""
        if self.lm_head is None:
            raise RuntimeError("the discrete objective needs language_model.lm_head.weight in the checkpoint")
        labels = batch["labels"].to("cpu", torch.int64)
        B, L = labels.shape
        D = self.cfg.llm_dim
        use_pp = batch.get("proprio") is not None and (proprio_projector is not None or self.proprio is not None)
        P = self.num_patches_total(batch["pixel_values"].shape[1] // 6, use_pp)
        S = P + L
        # text position j (j >= 1) with labels[b, j] != -100 is predicted by the hidden state of text position j - 1, which sits at
        # multimodal row b * S + P + (j - 1) (the BOS row is 0, the patches are rows 1..P; modeling_prismatic.py:474-496)
        bb, jj = torch.nonzero(labels[:, 1:] != -100, as_tuple=True)
        n_tok = int(bb.numel())
        if n_tok == 0:
            raise ValueError("no label in the batch is different from IGNORE_INDEX")
        rows_idx = (bb * S + P + jj).to(torch.int32).to(self.device)
        targets = labels[bb, jj + 1].contiguous().to(self.device)
        out = self.forward(batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["labels"], proprio=batch.get("proprio"),
                           train=backward, proprio_projector=proprio_projector, sel=rows_idx)     # last layer on the counted rows only
        assert out["P"] == P and out["S"] == S
        n_pad = (n_tok + 7) // 8 * 8
        if out["sel_rows"] is not None:
            x = out["action_hidden"]                       # n_tok % 8 == 0: already the gathered rows
        else:
            x = torch.zeros((n_pad, D), dtype=BF16, device=self.device)
            ops.gather_rows(out["hidden"].view(B * S, D), rows_idx, D, dst=x)
        logits = ops.gemm(x, self.lm_head)                                   # bf16 [n_pad, vocab]: lm_head under autocast
        loss_rows, amax, dlogits = ops.token_ce(logits[:n_tok], targets, grad_scale=loss_scale / n_tok if backward else None)
        pred = torch.full((B, L - 1), -1, dtype=torch.int64)
        pred[bb, jj] = amax.to("cpu", torch.int64)
        if not backward:
            return loss_rows.sum().reshape(1), n_tok, pred
        if n_pad > n_tok:
            logits[n_tok:].zero_()
        if getattr(self, "_lm_head_t", None) is None:
            self._lm_head_t = ops.transpose(self.lm_head)                    # frozen: one transposed copy for the data gradient
        dx = ops.gemm(logits, self._lm_head_t)                               # d hidden rows = dlogits @ W
        if out["sel_rows"] is not None:
            self.backward_from_hidden(dx, out["saved"])
        else:
            dhidden = torch.zeros((B * S, D), dtype=BF16, device=self.device)
            ops.gather_rows(dx, rows_idx, D, dst=dhidden, scatter_add=True)
            self.backward_from_hidden(dhidden, out["saved"])
        return loss_rows.sum().reshape(1), n_tok, pred


# ======================================================================================================================
# hipGraph replay of the single-chunk inference forward (BASELINE.json configs[1])
# ======================================================================================================================
class ChunkGraph:
    """The batch-B inference forward (vision towers on their two streams -> projector -> assembly -> Llama stack -> action
    row gather -> optional L1 head) captured ONCE as a hipGraph and replayed per observation: ~1.3 k kernel launches become
    one graph launch, which is what bounds batch-1 latency.  Inputs are copied into static device buffers before each
    replay; outputs are static buffers too (clone them to keep a result across calls).

    Everything the capture allocates (activations, split-K workspaces) lives in the graph's private pool and stays valid as
    long as this object does.  Text length L is part of the captured shapes: callers pad the prompt to a bucket (right
    padding is masked exactly: padded keys contribute exact zeros) or keep one ChunkGraph per L."""

    def __init__(self, engine: "VLAEngine", B: int, L: int, pixel_shape, *, head=None, use_proprio: bool = True, proprio_projector=None):
        dev = engine.device
        self.engine, self.head, self.B, self.L = engine, head, B, L
        self.proprio_projector = proprio_projector
        self.ids = torch.zeros((B, L), dtype=torch.int64, device=dev)
        self.lab = torch.full((B, L), -100, dtype=torch.int64, device=dev)
        self.lens = torch.full((B,), L, dtype=torch.int32, device=dev)
        self.pixels = torch.zeros(tuple(pixel_shape), dtype=BF16, device=dev)
        self.proprio = torch.zeros((B, engine.cfg.proprio_dim), dtype=BF16, device=dev) if use_proprio else None
        self.graph = None
        self.out = None

    def _run(self):
        eng = self.engine
        out = eng.forward_dev(self.ids, self.lab, self.lens, self.pixels, proprio=self.proprio, train=False,
                              proprio_projector=self.proprio_projector, sel="actions")
        ah, _ = eng.action_hidden(out)
        pred = self.head.fwd(ah)[0] if self.head is not None else None
        return pred, ah

    def load(self, input_ids, attention_mask, pixel_values, labels, proprio=None):
        lens = VLAEngine.check_right_padding(attention_mask)
        assert tuple(input_ids.shape) == (self.B, self.L), f"ChunkGraph captured for ids {(self.B, self.L)}, got {tuple(input_ids.shape)}"
        self.ids.copy_(input_ids.to(torch.int64), non_blocking=True)
        self.lab.copy_(labels.to(torch.int64), non_blocking=True)
        self.lens.copy_(lens.to(torch.int32), non_blocking=True)
        self.pixels.copy_(pixel_values.reshape(self.pixels.shape), non_blocking=True)
        if self.proprio is not None:
            self.proprio.copy_(proprio.reshape(self.proprio.shape), non_blocking=True)

    def capture(self):
        """Call after load() of a representative input: one eager warm-up (lazy tables, kernel attributes), then the capture."""
        assert ops.PROFILE is None, "no per-launch event timing inside a graph capture"
        self._run()
        torch.cuda.synchronize(self.engine.device)
        outer_ws, ops._ws_cache = ops._ws_cache, {}          # workspaces allocated while capturing belong to the graph's pool
        try:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self._run()
        finally:
            self._ws, ops._ws_cache = ops._ws_cache, outer_ws
        return self

    def replay(self):
        self.graph.replay()
        return self.out

    def __call__(self, input_ids, attention_mask, pixel_values, labels, proprio=None):
        self.load(input_ids, attention_mask, pixel_values, labels, proprio)
        if self.graph is None:
            self.capture()
        return self.replay()
