"""Reference-facing model classes over the HIP engine.

Mirrors the call signatures the LIBERO fine-tune / eval glue uses:
  OpenVLAForActionPrediction.forward / .predict_action      prismatic/extern/hf/modeling_prismatic.py:499-675, 946-1060
  L1RegressionActionHead.predict_action                     prismatic/models/action_heads.py:84-107
  ProprioProjector / NoisyActionProjector                   prismatic/models/projectors.py:6-49
`forward` returns hidden states that carry a torch autograd edge into the engine's explicit backward, so reference-style
glue (`loss = L1Loss(gt, head.predict_action(h)); loss.backward()`) works unchanged; the fused training step
(engine.train_step_fwd_bwd) is what `finetune()` and bench.py use.  All arithmetic runs in libovla_hip.so.
"""
from __future__ import annotations

import math
from typing import Any, Dict, Optional, Tuple

import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .config import VLAConfig
from .diffusion import DDIMScheduler, SinusoidalPositionalEncoding
from .engine import ChunkGraph, ActionHead, MlpProjector, ParamStore, VLAEngine, build_component
from .weights import make_getter

BF16 = torch.bfloat16
IGNORE_INDEX = -100
ACTION_TOKEN_BEGIN_IDX = 31743
STOP_INDEX = 2


def _device(device=None) -> torch.device:
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("openvla-oft_amd runs on MI355X only: there is no CPU / eager fallback for the HIP path")
    return torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())


# ======================================================================================================================
# torch-facing parameter views (so `optimizer = AdamW(module.parameters())` style glue keeps working)
# ======================================================================================================================
class _StoreModule:
    """Gives a ParamStore the small part of the nn.Module protocol the reference glue touches."""

    store: ParamStore

    def _torch_params(self):
        st = self.store
        if not hasattr(st, "_tparams"):
            st._tparams = {}
            st._tgrad = {dt: torch.zeros_like(flat) for dt, flat in st.flat.items()}
            for p in st.params:
                tp = nn.Parameter(p.data, requires_grad=True)
                st._tparams[p.name] = tp
        return st._tparams

    def parameters(self):
        return list(self._torch_params().values())

    def named_parameters(self):
        return list(self._torch_params().items())

    def publish_grads(self):
        """fp32 accumulators -> `.grad` views in the parameter dtype (what autograd would have produced)."""
        st = self.store
        tps = self._torch_params()
        for dt, g32 in st.flat_grad.items():
            if dt == BF16:
                ops.cvt_f32_to_bf16(g32, st._tgrad[dt])
            else:
                st._tgrad[dt].copy_(g32)
        for p in st.params:
            tps[p.name].grad = st._tgrad[p.dtype][p.offset: p.offset + p.numel].view(p.shape)
        st._published = True

    def _reset_if_cleared(self):
        """optimizer.zero_grad() (set_to_none) cleared the published grads -> start a fresh accumulation."""
        st = self.store
        if getattr(st, "_published", False) and all(tp.grad is None for tp in self._torch_params().values()):
            st.zero_grad()
            st._published = False

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, *a, **k):
        return self

    @property
    def module(self):  # DDP-style `.module` access used by the reference glue (finetune.py:398)
        return self


# ======================================================================================================================
# components
# ======================================================================================================================
def _linear_init(out_f, in_f, gen):
    """nn.Linear default init (kaiming uniform a=sqrt(5) -> U(-1/sqrt(in), 1/sqrt(in)) for weight and bias)."""
    b = 1.0 / math.sqrt(in_f)
    return (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * b, (torch.rand(out_f, generator=gen) * 2 - 1) * b


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, head_mod, train):
        B = h.shape[0]
        comp = head_mod.comp
        pred, _, saved = comp.fwd(h.detach().to(BF16).contiguous().view(-1, h.shape[-1]), train=train)
        ctx.saved, ctx.mod, ctx.shape = saved, head_mod, h.shape
        return pred.view(B, comp.cfg.chunk, comp.cfg.action_dim)

    @staticmethod
    def backward(ctx, dpred):
        comp = ctx.mod.comp
        dah = comp.bwd(ctx.saved, dpred=dpred.to(BF16).contiguous().view(-1, comp.cfg.action_dim))
        ctx.mod.publish_grads()
        return dah.view(ctx.shape), None, None


class L1RegressionActionHead(_StoreModule):
    """prismatic/models/action_heads.py:84-107: MLPResNet(num_blocks=2, input_dim*ACTION_DIM -> hidden -> action_dim)."""

    prefix = "model."

    def __init__(self, input_dim: int = 4096, hidden_dim: int = 4096, action_dim: int = 7, *, num_actions_chunk: int = 8, device=None,
                 state_dict: Optional[Dict[str, torch.Tensor]] = None, seed: int = 0):
        if input_dim != hidden_dim:
            raise ValueError("the reference always builds the head with input_dim == hidden_dim == llm_dim")
        self.action_dim, self.llm_dim = action_dim, input_dim
        self.cfg = VLAConfig(llm_dim=input_dim, action_dim=action_dim, chunk=num_actions_chunk)
        self.device = _device(device)
        self.training = False
        sd = state_dict if state_dict is not None else self._default_init(seed)
        self._build(sd)

    def _default_init(self, seed):
        g = torch.Generator().manual_seed(seed)
        D, A, p = self.llm_dim, self.action_dim, self.prefix
        sd = {}
        for name, dim in (("layer_norm1", D * A), ("layer_norm2", D), ("mlp_resnet_blocks.0.ffn.0", D), ("mlp_resnet_blocks.1.ffn.0", D)):
            sd[p + name + ".weight"], sd[p + name + ".bias"] = torch.ones(dim), torch.zeros(dim)
        for name, o, i in (("fc1", D, D * A), ("mlp_resnet_blocks.0.ffn.1", D, D), ("mlp_resnet_blocks.1.ffn.1", D, D), ("fc2", A, D)):
            sd[p + name + ".weight"], sd[p + name + ".bias"] = _linear_init(o, i, g)
        return sd

    def _build(self, sd):
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}   # DDP prefix tolerance (finetune.py:134-156)
        get, _ = make_getter(sd, self.device)
        self.comp = build_component(ActionHead, self.device, get, self.prefix, cfg=self.cfg)
        self.store = self.comp.store

    def predict_action(self, actions_hidden_states: torch.Tensor) -> torch.Tensor:
        """(B, chunk*action_dim, D) -> (B, chunk, action_dim) bf16."""
        self._reset_if_cleared()
        if torch.is_grad_enabled() and actions_hidden_states.requires_grad:
            return _HeadFn.apply(actions_hidden_states, self, True)
        with torch.no_grad():
            return _HeadFn.apply(actions_hidden_states, self, False)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        out = {}
        for lin in self.comp.linears():
            out.update(lin.export("data"))
        for p in self.comp.plain_params():
            out[p.name] = p.data
        return {k: v.detach().clone() for k, v in out.items()}

    def load_state_dict(self, sd):
        self._build(sd)


class DiffusionActionHead(L1RegressionActionHead):
    """prismatic/models/action_heads.py:144-211: the same MLPResNet predicting the noise, plus the DDIM scheduler and the
    sinusoidal timestep encoder."""

    prefix = "noise_predictor.mlp_resnet."

    def __init__(self, input_dim: int = 4096, hidden_dim: int = 4096, action_dim: int = 7, num_diffusion_steps: int = 100, **kw):
        super().__init__(input_dim, hidden_dim, action_dim, **kw)
        self.noise_scheduler = DDIMScheduler(num_train_timesteps=num_diffusion_steps, beta_schedule="squaredcos_cap_v2")
        self.num_diffusion_steps = num_diffusion_steps
        self.time_encoder = SinusoidalPositionalEncoding(dim=hidden_dim)

    def sample_noisy_actions(self, ground_truth_actions: torch.Tensor, generator: Optional[torch.Generator] = None):
        """action_heads.py:167-197 (host-side: a chunk is a few hundred numbers)."""
        gt = ground_truth_actions.detach().to("cpu", torch.float32)
        B = gt.shape[0]
        noise = torch.randn(gt.shape, generator=generator).to(BF16).float()
        timesteps = torch.randint(0, self.noise_scheduler.config.num_train_timesteps, (B,), generator=generator)
        noisy = self.noise_scheduler.add_noise(gt, noise, timesteps).to(BF16)
        temb = self.time_encoder(timesteps.float()).to(BF16).unsqueeze(1)
        return dict(noise=noise.to(BF16), noisy_actions=noisy, diffusion_timestep_embeddings=temb, timesteps=timesteps)

    def predict_noise(self, actions_hidden_states: torch.Tensor) -> torch.Tensor:
        """action_heads.py:199-211"""
        return self.predict_action(actions_hidden_states)


class ProprioProjector(_StoreModule):
    """prismatic/models/projectors.py:6-24 (fp32 parameters, bf16 compute: finetune.py:895-901 never casts it)."""

    prefix = ""

    def __init__(self, llm_dim: int, proprio_dim: int, *, device=None, state_dict=None, seed: int = 0):
        self.llm_dim, self.in_dim = llm_dim, proprio_dim
        self.device = _device(device)
        self.training = False
        if state_dict is None:
            g = torch.Generator().manual_seed(seed)
            state_dict = {}
            state_dict["fc1.weight"], state_dict["fc1.bias"] = _linear_init(llm_dim, proprio_dim, g)
            state_dict["fc2.weight"], state_dict["fc2.bias"] = _linear_init(llm_dim, llm_dim, g)
        self.load_state_dict(state_dict)

    def load_state_dict(self, sd):
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        raw = lambda name: sd[name].detach().to(self.device, torch.float32)   # keep fp32 masters exact
        self.comp = build_component(MlpProjector, self.device, raw, self.prefix)
        self.store = self.comp.store

    def state_dict(self):
        out = {}
        for lin in self.comp.linears():
            out.update(lin.export("data"))
        return {k: v.detach().clone() for k, v in out.items()}


class NoisyActionProjector(ProprioProjector):
    """prismatic/models/projectors.py:27-49"""

    def __init__(self, llm_dim: int, *, device=None, state_dict=None, seed: int = 0):
        super().__init__(llm_dim, 1, device=device, state_dict=state_dict, seed=seed)


# ======================================================================================================================
# the VLA
# ======================================================================================================================
class _LMLossFn(torch.autograd.Function):
    """`output.loss` of the reference's forward (LlamaForCausalLM with `labels`: logits.float(), shift by one, cross entropy with
    ignore_index -100, mean over the counted labels; modeling_prismatic.py:632-643 + :486-496 for the multimodal labels), computed on
    the counted rows only: frozen lm_head GEMM on the gathered hidden rows, `ovla_token_ce` (fp32 log-sum-exp, gradient written in place),
    and in the backward one GEMM with the transposed lm_head + a row scatter into d hidden."""

    @staticmethod
    def forward(ctx, hidden, vla, labels, P):
        eng = vla.engine
        if eng.lm_head is None:
            raise RuntimeError("output.loss / output.logits need language_model.lm_head.weight, which this checkpoint was loaded without")
        B, S, D = hidden.shape
        lab = labels.to("cpu", torch.int64)
        bb, jj = torch.nonzero(lab[:, 1:] != IGNORE_INDEX, as_tuple=True)
        n_tok = int(bb.numel())
        if n_tok == 0:
            raise ValueError("no label in the batch is different from IGNORE_INDEX")
        # text position j + 1 is predicted by the hidden state of text position j, which is multimodal row P + j (BOS is row 0)
        rows_idx = (bb * S + P + jj).to(torch.int32).to(eng.device)
        targets = lab[bb, jj + 1].contiguous().to(eng.device)
        n_pad = (n_tok + 7) // 8 * 8
        x = torch.zeros((n_pad, D), dtype=BF16, device=eng.device)
        ops.gather_rows(hidden.detach().reshape(B * S, D), rows_idx, D, dst=x)
        logits = ops.gemm(x, eng.lm_head)
        need_grad = ctx.needs_input_grad[0]
        loss_rows, amax, _ = ops.token_ce(logits[:n_tok], targets, grad_scale=(1.0 / n_tok) if need_grad else None)
        if need_grad:
            if n_pad > n_tok:
                logits[n_tok:].zero_()
            ctx.dlogits, ctx.rows_idx, ctx.shape, ctx.eng = logits, rows_idx, (B, S, D), eng
        vla._last_token_argmax = (bb, jj, amax)
        return loss_rows.sum() / n_tok

    @staticmethod
    def backward(ctx, dloss):
        eng = ctx.eng
        B, S, D = ctx.shape
        if getattr(eng, "_lm_head_t", None) is None:
            eng._lm_head_t = ops.transpose(eng.lm_head)
        dx = ops.gemm(ctx.dlogits, eng._lm_head_t, alpha=float(dloss))
        dhidden = torch.zeros((B * S, D), dtype=BF16, device=eng.device)
        ops.gather_rows(dx, ctx.rows_idx, D, dst=dhidden, scatter_add=True)
        return dhidden.view(B, S, D), None, None, None


class PrismaticCausalLMOutputWithPast:
    """prismatic/extern/hf/modeling_prismatic.py:266-278 / :668-675.  `hidden_states[-1]` (post final norm) and `projector_features`
    are materialised by the forward; `loss` and `logits` -- which the reference always computes and the L1 / diffusion recipes throw
    away (finetune.py:396-407) -- are computed on first access: `loss` as an autograd node on the hidden state (`loss.backward()`
    works), `logits` as the fp32 [B, S, vocab] tensor of the frozen lm_head without a gradient edge (0.6 GB at B = 8)."""

    past_key_values = None
    attentions = None

    def __init__(self, vla, hidden, labels, P, projector_features):
        self._vla, self._labels, self._P = vla, labels, P
        self.hidden_states = (hidden,)
        self.projector_features = projector_features
        self._loss = self._logits = None

    @property
    def loss(self) -> torch.Tensor:
        if self._loss is None:
            self._loss = _LMLossFn.apply(self.hidden_states[-1], self._vla, self._labels, self._P)
        return self._loss

    @property
    def logits(self) -> torch.Tensor:
        if self._logits is None:
            h = self.hidden_states[-1].detach()
            B, S, D = h.shape
            self._logits = self._vla.logits_for(h.reshape(B * S, D)).view(B, S, -1)
        return self._logits

    def __getitem__(self, key):   # ModelOutput-style access: out["loss"], out["hidden_states"]
        return getattr(self, key)


class _VisionBackboneHandle:
    """vla.vision_backbone.{get_num_patches, get_num_images_in_input, set_num_images_in_input} (modeling_prismatic.py:159-184)"""

    def __init__(self, cfg: VLAConfig):
        self._cfg = cfg
        self.num_images_in_input = cfg.num_images

    def get_num_patches(self) -> int:
        return self._cfg.dino.n_patches

    def get_num_images_in_input(self) -> int:
        return self.num_images_in_input

    def set_num_images_in_input(self, n: int) -> None:
        self.num_images_in_input = n


class _VLMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, vla, kwargs, train):
        out = vla.engine.forward(train=train, **kwargs)
        ctx.vla, ctx.saved = vla, out["saved"]
        ctx.mark_non_differentiable(out["action_rows"])
        vla._last = out
        return out["hidden"], out["action_rows"]

    @staticmethod
    def backward(ctx, dhidden, _):
        eng = ctx.vla.engine
        B, S, D = dhidden.shape
        eng.backward_from_hidden(dhidden.to(BF16).contiguous().view(B * S, D), ctx.saved)
        ctx.vla.publish_grads()
        for comp in (ctx.saved[8], ctx.saved[9]):
            if comp is not None and hasattr(comp, "owner"):
                comp.owner.publish_grads()
        return None, None, None, None


class OpenVLAForActionPrediction(_StoreModule):
    """prismatic/extern/hf/modeling_prismatic.py:720-1087 over the HIP engine."""

    def __init__(self, cfg: VLAConfig, state_dict: Dict[str, torch.Tensor], *, device=None, lora: Optional[bool] = None,
                 norm_stats: Optional[dict] = None, use_film: Optional[bool] = None):
        self.cfg = cfg
        self.device = _device(device)
        get, has = make_getter(state_dict, self.device)
        if lora is None:   # per Linear: a merged checkpoint may still carry the towers' adapters (the reference's FiLM evaluation path)
            lora = "auto" if any(k.endswith(".lora_A.weight") for k in state_dict) else False
        if use_film is None:
            use_film = any(".scale.weight" in k for k in state_dict)
        self.engine = VLAEngine(cfg, get, self.device, lora=lora, use_proprio=False, head="none", has=has, use_film=use_film)
        self.store = self.engine.store
        self.llm_dim = cfg.llm_dim
        self.norm_stats = norm_stats or {}
        self.bins = np.linspace(-1, 1, cfg.n_action_bins)                       # :725-729
        self.bin_centers = (self.bins[:-1] + self.bins[1:]) / 2.0
        self.vocab_size = cfg.vocab - cfg.pad_to_multiple_of                    # :732
        self.vision_backbone = _VisionBackboneHandle(cfg)
        self.config = type("Cfg", (), {"image_sizes": [cfg.dino.image_size, cfg.siglip.image_size], "pad_token_id": cfg.pad_token_id})()
        self.training = False
        self._anchor = torch.zeros((), device=self.device, requires_grad=True)
        # hipGraph replay of predict_action (L1 / discrete paths).  Off by default: a captured graph pins the parameter
        # buffers it was captured with, so it is for deployment (weights frozen), not for evaluation inside a training loop.
        self.use_graph = os.environ.get("OVLA_INFER_GRAPH", "0") == "1"
        self._graphs: Dict[tuple, ChunkGraph] = {}

    def merge_and_unload(self):
        """peft `merge_and_unload()` of merge_lora_weights_and_save.py:60-67 on device: W += (alpha/r) B A for every adapted
        Linear; the model becomes inference-only.  Returns self, like peft."""
        self.engine.merge_lora()
        self._graphs.clear()
        return self

    def enable_graph_replay(self, on: bool = True):
        self.use_graph = on
        if not on:
            self._graphs.clear()
        return self

    # -- forward (:499-675) -------------------------------------------------------------------------------------------
    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    def forward(self, input_ids=None, attention_mask=None, pixel_values=None, labels=None, inputs_embeds=None, past_key_values=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, output_projector_features=None, return_dict=None,
                proprio=None, proprio_projector=None, noisy_actions=None, noisy_action_projector=None,
                diffusion_timestep_embeddings=None, use_film: bool = False):
        if input_ids is None or pixel_values is None or labels is None:
            raise ValueError("Invalid PrismaticForConditionalGeneration `forward()` call: the HIP path implements the multimodal "
                             "action-prediction branch (input_ids, pixel_values and labels are required)")
        if input_ids.shape[0] != pixel_values.shape[0]:
            raise ValueError("Non-homogenous batch of (text, image) input -- forward() does not support mixed batches!")
        if past_key_values is not None or inputs_embeds is not None:
            raise ValueError("cached generation / inputs_embeds are not part of the parallel-decoding action path")
        if use_film != self.engine.use_film:
            raise ValueError(f"use_film={use_film} but the model was built with use_film={self.engine.use_film} (FiLM adds parameters to the "
                             "vision backbone: finetune.py:874-888, openvla_utils.py:311-349)")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
        self._reset_if_cleared()
        for m in (proprio_projector, noisy_action_projector):
            if m is not None:
                m._reset_if_cleared()
                m.comp.owner = m
        kwargs = dict(input_ids=input_ids, attention_mask=attention_mask, pixel_values=pixel_values, labels=labels, proprio=proprio,
                      noisy_actions=noisy_actions, timestep_emb=diffusion_timestep_embeddings,
                      proprio_projector=None if proprio_projector is None else proprio_projector.comp,
                      noisy_action_projector=None if noisy_action_projector is None else noisy_action_projector.comp)
        if torch.is_grad_enabled():
            hidden, _ = _VLMFn.apply(self._anchor, self, kwargs, True)
        else:
            with torch.no_grad():
                hidden, _ = _VLMFn.apply(self._anchor, self, kwargs, False)
        # the reference also returns the CE loss / fp32 logits of the frozen lm_head; in L1 / diffusion mode they are discarded
        # (finetune.py:400,407), so the output object computes them on first access only
        return PrismaticCausalLMOutputWithPast(self, hidden, labels, self._last["P"], self._last["all_patches"])

    def logits_for(self, hidden_rows: torch.Tensor) -> torch.Tensor:
        """lm_head on selected hidden rows [n, D] -> fp32 logits [n, vocab] (discrete action-token path, :929-942)."""
        if self.engine.lm_head is None:
            raise RuntimeError("this checkpoint was loaded without language_model.lm_head.weight")
        n = hidden_rows.shape[0]
        pad = (n + 7) // 8 * 8
        rows = torch.zeros((pad, self.cfg.llm_dim), dtype=BF16, device=self.device)
        rows[:n] = hidden_rows
        return ops.cvt_bf16_to_f32(ops.gemm(rows, self.engine.lm_head))[:n]

    # -- predict_action (:946-1060) -------------------------------------------------------------------------------------
    @torch.no_grad()
    def predict_action(self, input_ids=None, unnorm_key=None, proprio=None, proprio_projector=None, action_head=None,
                       noisy_action_projector=None, use_film: bool = False, **kwargs):
        cfg = self.cfg
        if use_film != self.engine.use_film:
            raise ValueError(f"use_film={use_film} but the model was built with use_film={self.engine.use_film}")
        A = cfg.num_action_tokens
        assert input_ids.shape[0] == 1, "Generation is only currently supported for batch size of 1!"
        pixel_values, attention_mask = kwargs["pixel_values"], kwargs["attention_mask"]
        input_ids = input_ids.to("cpu", torch.int64)
        attention_mask = attention_mask.to("cpu")
        if not torch.all(input_ids[:, -1] == 29871):                                      # :974-977
            input_ids = torch.cat((input_ids, torch.tensor([[29871]], dtype=torch.int64)), dim=1)
            attention_mask = torch.cat((attention_mask, torch.ones((1, 1), dtype=attention_mask.dtype)), dim=1)
        ids = torch.cat([input_ids, torch.ones((1, A), dtype=torch.int64), torch.full((1, 1), STOP_INDEX, dtype=torch.int64)], dim=-1)
        mask = torch.cat([attention_mask, torch.ones((1, A + 1), dtype=attention_mask.dtype)], dim=-1)
        labels = torch.full_like(ids, IGNORE_INDEX)                                      # :983-993
        labels[:, input_ids.shape[-1]:] = ACTION_TOKEN_BEGIN_IDX + 1
        labels[:, -1] = STOP_INDEX
        use_proprio = proprio_projector is not None and proprio is not None
        prop = torch.as_tensor(np.asarray(proprio), dtype=torch.float32) if use_proprio else None
        use_diffusion = noisy_action_projector is not None and hasattr(action_head, "noise_scheduler")
        if use_diffusion:                                                                 # :793-877
            sched = action_head.noise_scheduler
            sched.set_timesteps(action_head.num_diffusion_steps)
            cur = kwargs.get("noise")
            if cur is None:
                cur = torch.randn((1, cfg.chunk, cfg.action_dim))
            cur = cur.to("cpu", torch.float32).to(BF16).float()
            cached, ah = None, None
            for t in sched.timesteps:
                temb = action_head.time_encoder(torch.tensor([float(t)])).to(BF16)
                out = self.engine.forward(ids, mask, pixel_values, labels, proprio=prop, train=False, noisy_actions=cur.to(BF16),
                                          timestep_emb=temb, proprio_projector=proprio_projector.comp if use_proprio else None,
                                          noisy_action_projector=noisy_action_projector.comp, cached_patches=cached, sel="actions")
                cached = out["patches"]                                                   # vision features reused across steps (:810)
                ah, _ = self.engine.action_hidden(out)
                eps = action_head.predict_noise(ah.view(1, A, cfg.llm_dim)).reshape(cur.shape).float().cpu()
                cur = sched.step(eps, int(t), cur).prev_sample.to(BF16).float()
            normalized = cur.reshape(cfg.chunk, cfg.action_dim).numpy()
            return self._unnormalize_actions(normalized, unnorm_key), ah.view(1, A, cfg.llm_dim)
        if self.use_graph:
            # one hipGraph per (text length, head, projector): ~1.3 k launches -> one graph launch (engine.ChunkGraph)
            head_comp = getattr(action_head, "comp", None) if action_head is not None else None
            pp_comp = proprio_projector.comp if use_proprio else None
            key = (ids.shape[1], tuple(pixel_values.shape), id(head_comp), id(pp_comp))
            g = self._graphs.get(key)
            if g is None:
                g = self._graphs[key] = ChunkGraph(self.engine, 1, ids.shape[1], pixel_values.shape, head=head_comp, use_proprio=use_proprio,
                                                   proprio_projector=pp_comp)
                g._keep = (head_comp, pp_comp)   # the captured kernels read these parameter buffers: keep them alive
            pred, ah = g(ids, mask, pixel_values, labels, prop)
            actions_hidden_states = ah.view(1, A, cfg.llm_dim).clone()
            if action_head is not None:
                normalized = pred.reshape(cfg.chunk, cfg.action_dim).float().cpu().numpy()
                return self._unnormalize_actions(normalized, unnorm_key), actions_hidden_states
            ah = actions_hidden_states.view(A, cfg.llm_dim)
        else:
            out = self.engine.forward(ids, mask, pixel_values, labels, proprio=prop, train=False,
                                      proprio_projector=proprio_projector.comp if use_proprio else None, sel="actions")
            ah, _ = self.engine.action_hidden(out)                                        # rows P+NPT .. P+NPT+A-1 (:915-920)
            actions_hidden_states = ah.view(1, A, cfg.llm_dim)
        if action_head is not None:                                                       # :923-927
            normalized = action_head.predict_action(actions_hidden_states).reshape(cfg.chunk, cfg.action_dim).float().cpu().numpy()
        else:                                                                             # :929-942
            tok = self.logits_for(ah).argmax(dim=1).cpu().numpy()
            d = np.clip(self.vocab_size - tok - 1, a_min=0, a_max=self.bin_centers.shape[0] - 1)
            normalized = self.bin_centers[d].reshape(cfg.chunk, cfg.action_dim)
        return self._unnormalize_actions(normalized, unnorm_key), actions_hidden_states

    # -- statistics (:772-791, :1062-1087) -------------------------------------------------------------------------------
    @staticmethod
    def _check_unnorm_key(norm_stats, unnorm_key):
        if unnorm_key is None:
            assert len(norm_stats) == 1, (
                f"Your model was trained on more than one dataset, please pass a `unnorm_key` from the following options to choose the "
                f"statistics used for un-normalizing actions: {norm_stats.keys()}")
            unnorm_key = next(iter(norm_stats.keys()))
        assert unnorm_key in norm_stats, (
            f"The `unnorm_key` you chose is not in the set of available dataset statistics, please choose from: {norm_stats.keys()}")
        return unnorm_key

    def get_action_dim(self, unnorm_key=None) -> int:
        return len(self.norm_stats[self._check_unnorm_key(self.norm_stats, unnorm_key)]["action"]["min"])

    def get_action_stats(self, unnorm_key=None):
        return self.norm_stats[self._check_unnorm_key(self.norm_stats, unnorm_key)]["action"]

    def _unnormalize_actions(self, normalized_actions, unnorm_key=None):
        stats = self.get_action_stats(unnorm_key)
        if self.cfg.norm_type == "bounds":
            mask = stats.get("mask", np.ones_like(stats["min"], dtype=bool))
            high, low = np.array(stats["max"]), np.array(stats["min"])
        elif self.cfg.norm_type == "bounds_q99":
            mask = stats.get("mask", np.ones_like(stats["q01"], dtype=bool))
            high, low = np.array(stats["q99"]), np.array(stats["q01"])
        else:
            raise ValueError("Unsupported action/proprio normalization type detected!")
        return np.where(mask, 0.5 * (normalized_actions + 1) * (high - low + 1e-8) + low, normalized_actions)

    # -- checkpoint surface --------------------------------------------------------------------------------------------
    def lora_state_dict(self) -> Dict[str, torch.Tensor]:
        """LoRA adapters under `<linear>.lora_A.weight` / `.lora_B.weight` (peft adds `base_model.model.` + `.default`)."""
        out = {}
        for lin in self.engine.vlm_linears():
            out.update(lin.export("data"))
        return {k: v.detach().clone() for k, v in out.items()}
