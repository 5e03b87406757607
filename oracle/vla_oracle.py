"""oracle/vla_oracle.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement (plain PyTorch, fp32) of the reference's OpenVLA-OFT parallel-decoding action-chunk forward path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only as the checker:
the product path (openvla-oft_amd/) never routes through it.

Pinning status (see DESIGN.md "Oracle"):
  * masks, action heads, projectors, time encoder, action-token decode, collator layout: pinned against the reference's own
    modules imported by path in this container (tests/golden/make_golden.py -> tests/golden/*.npz);
  * Llama decoder stack: pinned against stock transformers 5.15 LlamaForCausalLM (the reference's fork
    moojink/transformers-openvla-oft 4.40.1 is absent: bidirectional attention itself is PARITY UNPINNED);
  * ViT blocks: timm 0.9.10 is absent; `vit_block` is pinned against transformers' Dinov2Layer / SiglipEncoderLayer (independent
    implementations of the two tower architectures, G14); the tower wrapper (patch embed, prefix tokens, "block depth-2, no final norm") is
    restated from the reference call sites: PARITY UNPINNED against timm itself;
  * LoRA: pinned against plain torch autograd (G8); peft 0.11.1 itself absent.  DDIM (diffusers absent): PARITY UNPINNED, restated from the
    library's published algorithm;
  * the JPEG round trip (oracle/jpeg_oracle.py): pinned bit-exactly against libjpeg-turbo (G12);
  * since round 3 the model-level logic -- projector, FiLM block, vision backbones (which block, prefix drop, concat order, language average),
    action masks / embedding replacement / multimodal concat, `forward` (L1, diffusion inputs, FiLM, both mask modes, logits, CE loss),
    `predict_action` (L1, discrete, diffusion loop), `RLDSBatchTransform`, `run_forward_pass` (all three objectives + metrics) -- is pinned against
    the reference's OWN files EXECUTED in the build container: prismatic/extern/hf/modeling_prismatic.py, prismatic/models/film_vit_wrapper.py,
    prismatic/vla/datasets/datasets.py, vla-scripts/finetune.py (fixtures G16-G21; tests/golden/make_golden_ref_model.py,
    make_golden_batch_transform.py, make_golden_run_forward_pass.py).  Still restated from text only: experiments/robot/openvla_utils.py's TF image ops
    and processing_prismatic.py (TensorFlow / torchvision absent).

Every function cites the reference file:line (relative to the reference root) it follows.

`mode="bf16"` re-rounds to bfloat16 at every point where the reference's bf16-autocast PyTorch path materialises a
bf16 tensor, so that the HIP path can be compared against what the reference would compute in bf16 rather than against
exact arithmetic.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

IGNORE_INDEX = -100            # prismatic/vla/constants.py:11
ACTION_TOKEN_BEGIN_IDX = 31743  # prismatic/vla/constants.py:12
STOP_INDEX = 2                 # prismatic/vla/constants.py:13

# prismatic/vla/constants.py:26-52
PLATFORM_CONSTANTS = {
    "LIBERO": dict(NUM_ACTIONS_CHUNK=8, ACTION_DIM=7, PROPRIO_DIM=8, NORM="bounds_q99"),
    "UR5E": dict(NUM_ACTIONS_CHUNK=8, ACTION_DIM=7, PROPRIO_DIM=6, NORM="bounds"),
    "ALOHA": dict(NUM_ACTIONS_CHUNK=25, ACTION_DIM=14, PROPRIO_DIM=14, NORM="bounds"),
    "BRIDGE": dict(NUM_ACTIONS_CHUNK=5, ACTION_DIM=7, PROPRIO_DIM=7, NORM="bounds_q99"),
}


@dataclass
class VitConfig:
    dim: int
    depth: int
    heads: int
    mlp_hidden: int
    n_prefix: int = 0          # cls + register tokens (DINOv2 reg4: 5; SigLIP: 0)
    layerscale: bool = False   # DINOv2: True
    patch: int = 14
    image_size: int = 224
    eps: float = 1e-6
    act: str = "gelu"          # timm default nn.GELU (exact erf)

    @property
    def n_patches(self):
        return (self.image_size // self.patch) ** 2


@dataclass
class OracleConfig:
    llm_dim: int = 4096
    llm_layers: int = 32
    llm_heads: int = 32
    llm_ff: int = 11008
    vocab: int = 32064
    rms_eps: float = 1e-5      # Llama-2 config
    rope_theta: float = 10000.0
    dino: VitConfig = field(default_factory=lambda: VitConfig(1024, 24, 16, 4096, n_prefix=5, layerscale=True))
    siglip: VitConfig = field(default_factory=lambda: VitConfig(1152, 27, 16, 4304))
    num_images: int = 2
    lora_rank: int = 32
    lora_alpha: int = 16       # finetune.py:864  min(rank, 16)
    action_dim: int = 7
    chunk: int = 8
    proprio_dim: int = 8
    norm_type: str = "bounds_q99"
    n_action_bins: int = 256
    pad_to_multiple_of: int = 64

    @property
    def lora_scale(self):
        return self.lora_alpha / self.lora_rank


def tiny_config(**kw) -> OracleConfig:
    """Reduced dims with the same structural quirks (head_dim 128/64/72, 5 prefix tokens, LayerScale, odd MLP width)."""
    cfg = OracleConfig(
        llm_dim=256, llm_layers=2, llm_heads=2, llm_ff=512, vocab=32064,
        dino=VitConfig(128, 3, 2, 256, n_prefix=5, layerscale=True, image_size=56),
        siglip=VitConfig(144, 3, 2, 536, image_size=56),
        num_images=2, lora_rank=32, lora_alpha=16,
    )
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


# ======================================================================================================================
# integer / host-side pieces
# ======================================================================================================================
def current_action_mask(token_ids: torch.Tensor, action_dim: int) -> torch.Tensor:
    """prismatic/training/train_utils.py:8-22"""
    cumsum = torch.cumsum(token_ids != IGNORE_INDEX, dim=1)
    return (token_ids > ACTION_TOKEN_BEGIN_IDX) & (1 <= cumsum) & (cumsum <= action_dim)


def next_actions_mask(token_ids: torch.Tensor, action_dim: int) -> torch.Tensor:
    """prismatic/training/train_utils.py:25-39"""
    cumsum = torch.cumsum(token_ids != IGNORE_INDEX, dim=1)
    return (token_ids > ACTION_TOKEN_BEGIN_IDX) & (cumsum > action_dim)


def all_actions_mask(labels: torch.Tensor, action_dim: int) -> torch.Tensor:
    """prismatic/extern/hf/modeling_prismatic.py:431-436"""
    return current_action_mask(labels, action_dim) | next_actions_mask(labels, action_dim)


def action_bins(n_bins: int = 256):
    """prismatic/vla/action_tokenizer.py:30-32 ; modeling_prismatic.py:725-729"""
    bins = np.linspace(-1, 1, n_bins)
    return bins, (bins[:-1] + bins[1:]) / 2.0


def tokenize_actions(actions: np.ndarray, vocab_size: int = 32000, n_bins: int = 256) -> np.ndarray:
    """prismatic/vla/action_tokenizer.py:38-47 (ids only; the string decode/encode round trip is the tokenizer's)."""
    bins, _ = action_bins(n_bins)
    a = np.clip(actions, -1.0, 1.0)
    return vocab_size - np.digitize(a, bins)


def decode_token_ids_to_actions(ids: np.ndarray, vocab_size: int = 32000, n_bins: int = 256) -> np.ndarray:
    """prismatic/vla/action_tokenizer.py:49-68 ; modeling_prismatic.py:940-942"""
    _, centers = action_bins(n_bins)
    d = np.clip(vocab_size - ids - 1, a_min=0, a_max=centers.shape[0] - 1)
    return centers[d]


def _bounds(stats: dict, norm_type: str):
    if norm_type == "bounds":
        mask = stats.get("mask", np.ones_like(stats["min"], dtype=bool))
        return np.array(mask, dtype=bool), np.array(stats["max"]), np.array(stats["min"])
    if norm_type == "bounds_q99":
        mask = stats.get("mask", np.ones_like(stats["q01"], dtype=bool))
        return np.array(mask, dtype=bool), np.array(stats["q99"]), np.array(stats["q01"])
    raise ValueError("Unsupported action/proprio normalization type detected!")


def unnormalize_actions(normalized: np.ndarray, stats: dict, norm_type: str) -> np.ndarray:
    """prismatic/extern/hf/modeling_prismatic.py:772-791"""
    mask, high, low = _bounds(stats, norm_type)
    return np.where(mask, 0.5 * (normalized + 1) * (high - low + 1e-8) + low, normalized)


def normalize_proprio(proprio: np.ndarray, stats: dict, norm_type: str) -> np.ndarray:
    """experiments/robot/openvla_utils.py:645-675"""
    mask, high, low = _bounds(stats, norm_type)
    return np.clip(np.where(mask, 2 * (proprio - low) / (high - low + 1e-8) - 1, proprio), a_min=-1.0, a_max=1.0)


def crop_and_resize_center(image_u8: np.ndarray, crop_scale: float = 0.9, out_size: int = 224) -> np.ndarray:
    """experiments/robot/openvla_utils.py:542-622 restated in explicit float32, operation by operation, after the published
    algorithm of tensorflow==2.15.0 (pyproject.toml:52; absent here -> PARITY UNPINNED):
      convert_image_dtype(u8 -> f32):   x * f32(1/255)
      box (crop_and_resize, :565-579):  side = clip(sqrt(f32(crop_scale)), 0, 1); y1 = (1 - side) / 2; y2 = y1 + side       (f32)
      CropAndResize kernel:             scale = (y2 - y1) * (H - 1) / (out - 1);  in_y = y1 * (H - 1) + i * scale
                                        top = floor(in_y), bottom = ceil(in_y), lerp = in_y - top
                                        v = tl + (tr - tl) * x_lerp  (top and bottom rows), out = top + (bottom - top) * y_lerp
      clip_by_value(0, 1); convert_image_dtype(f32 -> u8, saturate): trunc(x * 255.5)."""
    f = np.float32
    img = image_u8.astype(f) * f(1.0 / 255.0)
    H, W = img.shape[:2]
    side = np.clip(np.sqrt(f(crop_scale)), f(0), f(1)).astype(f)
    o1 = ((f(1) - side) / f(2)).astype(f)
    o2 = (o1 + side).astype(f)

    def coords(n):
        scale = ((o2 - o1) * f(n - 1) / f(out_size - 1)).astype(f)
        c = (o1 * f(n - 1) + np.arange(out_size, dtype=f) * scale).astype(f)
        lo, hi = np.floor(c), np.ceil(c)
        return lo.astype(np.int64), hi.astype(np.int64), (c - lo).astype(f)

    y0, y1, wy = coords(H)
    x0, x1, wx = coords(W)
    wy, wx = wy[:, None, None], wx[None, :, None]
    tl, tr, bl, br = img[y0][:, x0], img[y0][:, x1], img[y1][:, x0], img[y1][:, x1]
    top = (tl + ((tr - tl) * wx).astype(f)).astype(f)
    bot = (bl + ((br - bl) * wx).astype(f)).astype(f)
    out = (top + ((bot - top) * wy).astype(f)).astype(f)
    return (np.clip(out, f(0), f(1)) * f(255.5)).astype(np.uint8)


def image_transform(image_u8: np.ndarray, means, stds) -> torch.Tensor:
    """prismatic/extern/hf/processing_prismatic.py:128-145 for an image that already is 224x224 (resize and center-crop
    are identities then): to_tensor (HWC u8 -> CHW float /255) and per-backbone normalise, channel-stacked (6, H, W).
    DINOv2 uses ImageNet mean/std, SigLIP 0.5/0.5 (timm data configs; stored in the checkpoint's preprocessor config)."""
    x = torch.from_numpy(image_u8.astype(np.float32) / 255.0).permute(2, 0, 1)
    outs = []
    for mean, std in zip(means, stds):
        m = torch.tensor(mean, dtype=torch.float32)[:, None, None]
        s = torch.tensor(std, dtype=torch.float32)[:, None, None]
        outs.append((x - m) / s)
    return torch.cat(outs, dim=0)


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
SIGLIP_MEAN, SIGLIP_STD = (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)


def build_prompt(task_label: str) -> str:
    """experiments/robot/openvla_utils.py:753 ; prismatic/models/backbones/llm/prompting/base_prompter.py:28-73"""
    return f"In: What action should the robot take to {task_label.lower()}?\nOut:"


# ======================================================================================================================
# DDIM (diffusers DDIMScheduler(num_train_timesteps=T, beta_schedule="squaredcos_cap_v2"); library defaults:
# clip_sample=True (range 1.0), prediction_type="epsilon", eta=0, timestep_spacing="leading", set_alpha_to_one=True,
# steps_offset=0).  PARITY UNPINNED (diffusers absent).  Call sites: prismatic/models/action_heads.py:163,185;
# modeling_prismatic.py:808-872.
# ======================================================================================================================
class DDIM:
    def __init__(self, num_train_timesteps: int):
        T = num_train_timesteps

        def alpha_bar(t):
            return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2

        betas = [min(1 - alpha_bar((i + 1) / T) / alpha_bar(i / T), 0.999) for i in range(T)]
        self.betas = torch.tensor(betas, dtype=torch.float32)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.T = T
        self.num_inference_steps = None
        self.timesteps = torch.arange(T - 1, -1, -1)

    def set_timesteps(self, n: int):
        self.num_inference_steps = n
        ratio = self.T // n
        self.timesteps = torch.from_numpy((np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64))

    def add_noise(self, x0, noise, t):
        ac = self.alphas_cumprod.to(x0.dtype)
        a = ac[t] ** 0.5
        s = (1 - ac[t]) ** 0.5
        while a.dim() < x0.dim():
            a, s = a.unsqueeze(-1), s.unsqueeze(-1)
        return a * x0 + s * noise

    def step(self, eps, t: int, x):
        prev_t = t - self.T // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        x0 = x0.clamp(-1.0, 1.0)
        eps = (x - a_t ** 0.5 * x0) / (1 - a_t) ** 0.5   # re-derived after clipping, as diffusers does
        return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps


def sinusoidal_encoding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """prismatic/models/action_heads.py:26-35"""
    half = dim // 2
    exponent = torch.arange(half, device=t.device) * -math.log(10000) / (half - 1)
    emb = t[:, None] * torch.exp(exponent)[None, :]
    return torch.cat((emb.sin(), emb.cos()), dim=-1)


# ======================================================================================================================
# the model
# ======================================================================================================================
class Oracle:
    """Functional restatement over a flat state dict `sd` that uses the reference's HF parameter names
    (vision_backbone.featurizer.*, vision_backbone.fused_featurizer.*, projector.*, language_model.*;
    action_head.*, proprio_projector.*, noisy_action_projector.*; LoRA as <linear>.lora_A.weight / .lora_B.weight;
    FiLM as <vit>.blocks.{i}.scale.* / .shift.*)."""

    def __init__(self, cfg: OracleConfig, sd: Dict[str, torch.Tensor], mode: str = "fp32", mask_mode: str = "bidirectional"):
        # mode "native": tensors stay in the dtype/device they are given in (bf16 on a GPU) and every op is the stock
        # PyTorch op -- the reference's eager execution path, used by bench.py --eager-baseline as the timing baseline
        assert mode in ("fp32", "bf16", "native") and mask_mode in ("bidirectional", "causal")
        self.cfg, self.sd, self.mode, self.mask_mode = cfg, sd, mode, mask_mode
        self.dev = next(iter(sd.values())).device

    # -- rounding points of the bf16-autocast reference ---------------------------------------------------------------
    def R(self, x):
        return x.to(torch.bfloat16).to(torch.float32) if self.mode == "bf16" else x

    def W(self, name):
        w = self.sd[name]
        if self.mode == "native":
            return w
        return w.to(torch.float32) if not w.requires_grad else w.float()

    def linear(self, x, name: str):
        """nn.Linear (+ peft LoRA: result + lora_B(lora_A(x)) * scaling; finetune.py:862-871).  Under autocast every
        matmul output is a bf16 tensor."""
        y = x @ self.W(name + ".weight").T
        if name + ".bias" in self.sd:
            y = y + self.W(name + ".bias")
        y = self.R(y)
        if name + ".lora_A.weight" in self.sd:
            t = self.R(x @ self.W(name + ".lora_A.weight").T)
            u = self.R(t @ self.W(name + ".lora_B.weight").T)
            y = self.R(y + self.R(u * self.cfg.lora_scale))
        return y

    def act(self, x, kind="gelu"):
        if kind == "gelu":
            return self.R(F.gelu(x))
        if kind == "gelu_tanh":
            return self.R(F.gelu(x, approximate="tanh"))
        raise ValueError(kind)

    # -- ViT (timm VisionTransformer; call site modeling_prismatic.py:127-139, 186-227) ---------------------------------
    def vit(self, img: torch.Tensor, prefix: str, vc: VitConfig, film_avg: Optional[torch.Tensor] = None) -> torch.Tensor:
        """img (B,3,H,W) -> (B, n_patches, dim): output of block index depth-2, prefix tokens dropped, no final norm
        (get_intermediate_layers(n={depth-2}), norm=False).  timm is absent; the tower-level wiring is pinned by G15 against transformers'
        Dinov2WithRegistersModel / SiglipVisionModel `hidden_states[-2]` (independent implementations of the two architectures)."""
        B = img.shape[0]
        w = self.W(prefix + "patch_embed.proj.weight")
        x = F.conv2d(img, w, self.W(prefix + "patch_embed.proj.bias"), stride=vc.patch)
        x = self.R(x.flatten(2).transpose(1, 2))                                      # (B, Np, dim)
        x = self.R(x + self.W(prefix + "pos_embed"))                                  # _pos_embed, no_embed_class
        toks = []
        if prefix + "cls_token" in self.sd:
            toks.append(self.W(prefix + "cls_token").expand(B, -1, -1))
        if prefix + "reg_token" in self.sd:
            toks.append(self.W(prefix + "reg_token").expand(B, -1, -1))
        if toks:
            x = torch.cat(toks + [x], dim=1)
        for i in range(vc.depth - 1):                                                 # blocks 0 .. depth-2
            x = self.vit_block(x, f"{prefix}blocks.{i}.", vc, film_avg)
        return x[:, vc.n_prefix:]

    def vit_block(self, x: torch.Tensor, p: str, vc: VitConfig, film_avg: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One pre-norm transformer block of timm's VisionTransformer (timm/models/vision_transformer.py `Block`; with FiLM:
        film_vit_wrapper.py:56-77): x += ls1(attn(norm1 x)); [x = x (1 + gamma) + beta]; x += ls2(mlp(norm2 x)).  x (B, T, dim); parameters under
        the name prefix `p`.  Pinned by G14 against transformers' Dinov2Layer (LayerScale) and SiglipEncoderLayer, independent implementations of
        the same two architectures (timm itself is absent)."""
        B = x.shape[0]
        H, hd = vc.heads, vc.dim // vc.heads
        h = F.layer_norm(x, (vc.dim,), self.W(p + "norm1.weight"), self.W(p + "norm1.bias"), vc.eps)
        qkv = self.linear(self.R(h), p + "attn.qkv").reshape(B, -1, 3, H, hd).permute(2, 0, 3, 1, 4)
        a = self.R(F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]))
        a = self.linear(a.transpose(1, 2).reshape(B, -1, vc.dim), p + "attn.proj")
        if vc.layerscale:
            a = self.R(a * self.W(p + "ls1.scale_factor"))
        x = self.R(x + a)
        if film_avg is not None:                                                      # film_vit_wrapper.py:65-75
            gamma = self.linear(film_avg, p + "scale")
            beta = self.linear(film_avg, p + "shift")
            x = self.R(self.R(x * self.R(1 + gamma[:, None, :])) + beta[:, None, :])
        h = F.layer_norm(x, (vc.dim,), self.W(p + "norm2.weight"), self.W(p + "norm2.bias"), vc.eps)
        h = self.act(self.linear(self.R(h), p + "mlp.fc1"), vc.act)
        h = self.linear(h, p + "mlp.fc2")
        if vc.layerscale:
            h = self.R(h * self.W(p + "ls2.scale_factor"))
        return self.R(x + h)

    def vision_backbone(self, pixel_values: torch.Tensor, film_avg=None) -> torch.Tensor:
        """modeling_prismatic.py:186-227 (film_vit_wrapper.py:231-276 with FiLM): channels [0:3] -> featurizer (DINOv2),
        [3:6] -> fused_featurizer (SigLIP); concat features on dim 2, images on dim 1."""
        outs = []
        for img in torch.split(pixel_values, 6, dim=1):
            a = self.vit(img[:, :3], "vision_backbone.featurizer.", self.cfg.dino, film_avg)
            b = self.vit(img[:, 3:], "vision_backbone.fused_featurizer.", self.cfg.siglip, film_avg)
            outs.append(torch.cat([a, b], dim=2))
        return torch.cat(outs, dim=1)

    def projector(self, x):
        """modeling_prismatic.py:250-262 (fused backbone branch)"""
        x = self.act(self.linear(x, "projector.fc1"))
        x = self.act(self.linear(x, "projector.fc2"))
        return self.linear(x, "projector.fc3")

    def mlp_projector(self, x, prefix):
        """prismatic/models/projectors.py:19-24 / 44-49"""
        return self.linear(self.act(self.linear(x, prefix + "fc1")), prefix + "fc2")

    # -- Llama (transformers LlamaModel; call site modeling_prismatic.py:632-643) --------------------------------------
    def rope(self, S: int, hd: int):
        inv_freq = 1.0 / (self.cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32, device=self.dev) / hd))
        freqs = torch.arange(S, dtype=torch.float32, device=self.dev)[:, None] * inv_freq[None, :]
        emb = torch.cat((freqs, freqs), dim=-1)
        if self.mode == "native":
            dt = self.sd["language_model.model.norm.weight"].dtype
            return emb.cos().to(dt), emb.sin().to(dt)
        return self.R(emb.cos()), self.R(emb.sin())

    def rmsnorm(self, x, name):
        if self.mode == "native":   # transformers LlamaRMSNorm: fp32 statistics, cast back, then scale
            xf = x.float()
            xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.cfg.rms_eps)
            return self.W(name) * xf.to(x.dtype)
        v = x.pow(2).mean(-1, keepdim=True)
        return self.R(self.W(name) * self.R(x * torch.rsqrt(v + self.cfg.rms_eps)))

    def llm(self, embeds: torch.Tensor, attention_mask: Optional[torch.Tensor]) -> torch.Tensor:
        """Returns hidden_states[-1] (post final norm).  attention_mask (B,S) bool: key padding.  mask_mode
        'bidirectional' = the reference fork's non-causal attention (pyproject.toml:50, modeling_prismatic.py:742);
        'causal' = stock HF."""
        B, S, D = embeds.shape
        H, hd = self.cfg.llm_heads, D // self.cfg.llm_heads
        cos, sin = self.rope(S, hd)
        bias = torch.zeros((B, 1, S, S), dtype=embeds.dtype, device=self.dev)
        if attention_mask is not None:
            bias = bias.masked_fill(~attention_mask.bool()[:, None, None, :], float("-inf"))
        if self.mask_mode == "causal":
            bias = bias.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device=self.dev), 1)[None, None], float("-inf"))

        def rot(x):
            return torch.cat((-x[..., hd // 2:], x[..., : hd // 2]), dim=-1)

        x = embeds
        for i in range(self.cfg.llm_layers):
            p = f"language_model.model.layers.{i}."
            h = self.rmsnorm(x, p + "input_layernorm.weight")
            q = self.linear(h, p + "self_attn.q_proj").view(B, S, H, hd).transpose(1, 2)
            k = self.linear(h, p + "self_attn.k_proj").view(B, S, H, hd).transpose(1, 2)
            v = self.linear(h, p + "self_attn.v_proj").view(B, S, H, hd).transpose(1, 2)
            q = self.R(self.R(q * cos) + self.R(rot(q) * sin))
            k = self.R(self.R(k * cos) + self.R(rot(k) * sin))
            a = self.R(F.scaled_dot_product_attention(q, k, v, attn_mask=bias))
            a = self.linear(a.transpose(1, 2).reshape(B, S, D), p + "self_attn.o_proj")
            x = self.R(x + a)
            h = self.rmsnorm(x, p + "post_attention_layernorm.weight")
            g = self.R(F.silu(self.linear(h, p + "mlp.gate_proj")))
            h = self.linear(self.R(g * self.linear(h, p + "mlp.up_proj")), p + "mlp.down_proj")
            x = self.R(x + h)
        return self.rmsnorm(x, "language_model.model.norm.weight")

    def lm_logits(self, hidden):
        return hidden @ self.W("language_model.lm_head.weight").T   # fp32 logits

    # -- heads ------------------------------------------------------------------------------------------------------------
    def mlp_resnet(self, x, prefix):
        """prismatic/models/action_heads.py:72-81 (module lives in bf16 outside autocast: LN output is bf16)"""
        d_in = x.shape[-1]
        x = self.R(F.layer_norm(x, (d_in,), self.W(prefix + "layer_norm1.weight"), self.W(prefix + "layer_norm1.bias"), 1e-5))
        x = self.R(F.relu(self.linear(x, prefix + "fc1")))
        hid = x.shape[-1]
        for b in range(2):
            q = f"{prefix}mlp_resnet_blocks.{b}.ffn."
            y = self.R(F.layer_norm(x, (hid,), self.W(q + "0.weight"), self.W(q + "0.bias"), 1e-5))
            x = self.R(self.R(F.relu(self.linear(y, q + "1"))) + x)
        x = self.R(F.layer_norm(x, (hid,), self.W(prefix + "layer_norm2.weight"), self.W(prefix + "layer_norm2.bias"), 1e-5))
        return self.linear(x, prefix + "fc2")

    def l1_head(self, actions_hidden):
        """L1RegressionActionHead.predict_action, action_heads.py:98-107"""
        B = actions_hidden.shape[0]
        return self.mlp_resnet(actions_hidden.reshape(B, self.cfg.chunk, -1), "action_head.model.")

    def noise_head(self, actions_hidden):
        """DiffusionActionHead.predict_noise, action_heads.py:199-211"""
        B = actions_hidden.shape[0]
        return self.mlp_resnet(actions_hidden.reshape(B, self.cfg.chunk, -1), "action_head.noise_predictor.mlp_resnet.")

    # -- multimodal forward (modeling_prismatic.py:571-643) -------------------------------------------------------------
    def multimodal_hidden(self, input_ids, attention_mask, pixel_values, labels, proprio=None, noisy_actions=None,
                          timestep_emb=None, use_film=False):
        cfg = self.cfg
        emb = self.W("language_model.model.embed_tokens.weight")[input_ids]            # :575
        amask = all_actions_mask(labels, cfg.action_dim)                                # :578
        film_avg = None
        if use_film:
            B = emb.shape[0]
            lang = emb[~amask].reshape(B, -1, emb.shape[2])                             # :581-583
            film_avg = self.R(lang.mean(dim=1))                                         # film_vit_wrapper.py:243
        patches = self.projector(self.vision_backbone(pixel_values, film_avg))          # :586
        if proprio is not None:                                                         # :589-591, :449-459
            pf = self.mlp_projector(self.R(proprio.reshape(patches.shape[0], -1).to(patches.dtype)), "proprio_projector.")
            patches = torch.cat((patches, pf[:, None, :]), dim=1)
        if timestep_emb is not None:                                                    # :594-599
            patches = torch.cat((patches, self.R(timestep_emb).to(patches.dtype)), dim=1)
        if noisy_actions is not None:                                                   # :602-616
            B = noisy_actions.shape[0]
            na = noisy_actions.reshape(B, -1, 1)
            na = na.to(patches.dtype) if self.mode == "native" else self.R(na.float())     # autocast runs the projector's Linears in bf16
            feats = self.mlp_projector(na, "noisy_action_projector.")
            emb = emb.clone()
            for b in range(B):
                emb[b, amask[b]] = feats[b]
        else:
            emb = emb * (~amask)[..., None]                                             # :620-621
        mm = torch.cat([emb[:, :1], patches, emb[:, 1:]], dim=1)                        # :474-476
        mm_mask = None
        if attention_mask is not None:
            ones = torch.ones((patches.shape[0], patches.shape[1]), dtype=torch.bool, device=self.dev)
            mm_mask = torch.cat([attention_mask[:, :1].bool(), ones, attention_mask[:, 1:].bool()], dim=1)
        return self.llm(mm, mm_mask), patches.shape[1]

    def train_forward(self, batch: dict, use_proprio=True, use_diffusion=False, use_film=False, noise=None, timesteps=None,
                      ddim: Optional[DDIM] = None):
        """vla-scripts/finetune.py:280-451 run_forward_pass (L1-regression or diffusion branch).  Returns
        (loss, predicted actions or noise, actions_hidden_states)."""
        cfg = self.cfg
        gt = batch["actions"] if self.mode == "native" else self.R(batch["actions"].float())   # :324 (.to(bfloat16))
        noisy, temb = None, None
        if use_diffusion:                                                               # action_heads.py:167-197
            noisy = self.R(ddim.add_noise(gt, noise, timesteps))
            temb = self.R(sinusoidal_encoding(timesteps.float(), cfg.llm_dim))[:, None, :]
        pv = batch["pixel_values"] if self.mode == "native" else self.R(batch["pixel_values"].float())
        hidden, P = self.multimodal_hidden(batch["input_ids"], batch["attention_mask"], pv,
                                           batch["labels"], batch["proprio"] if use_proprio else None, noisy, temb, use_film)
        gt_ids = batch["labels"][:, 1:]                                                 # :353
        m = current_action_mask(gt_ids, cfg.action_dim) | next_actions_mask(gt_ids, cfg.action_dim)
        text_hidden = hidden[:, P:-1]                                                   # :387
        B = hidden.shape[0]
        ah = text_hidden[m].reshape(B, cfg.chunk * cfg.action_dim, -1)                  # :389-394
        if use_diffusion:
            pred = self.noise_head(ah).reshape(noise.shape)                             # :404-407
            loss = self.R(F.mse_loss(pred, self.R(noise)))
        else:
            pred = self.l1_head(ah)                                                     # :396-400
            loss = self.R(self.R((gt - pred).abs()).mean())
        return loss, pred, ah

    def train_forward_discrete(self, batch: dict, use_proprio=True):
        """vla-scripts/finetune.py:357-359 (`loss = output.loss`; predicted ids = logits[:, P:-1].argmax(2)): the multimodal labels
        are IGNORE on the patches (modeling_prismatic.py:486-496); LlamaForCausalLM shifts by one and takes the mean cross entropy
        of the fp32-upcast logits over the labels that are not ignored.  Returns (loss, predicted ids [B, L - 1])."""
        pv = batch["pixel_values"] if self.mode == "native" else self.R(batch["pixel_values"].float())
        hidden, P = self.multimodal_hidden(batch["input_ids"], batch["attention_mask"], pv, batch["labels"],
                                           batch["proprio"] if use_proprio else None)
        logits = self.R(self.lm_logits(hidden)).float()                                 # lm_head under autocast, then .float()
        B = hidden.shape[0]
        mm_labels = torch.cat([batch["labels"][:, :1], torch.full((B, P), -100, dtype=batch["labels"].dtype), batch["labels"][:, 1:]], dim=1)
        loss = F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), mm_labels[:, 1:].reshape(-1), ignore_index=-100)
        return loss, logits[:, P:-1].argmax(dim=2)

    def predict_action(self, input_ids, attention_mask, pixel_values, proprio=None, unnorm_stats=None, use_film=False,
                       head: str = "l1", noise=None, num_diffusion_steps=None):
        """modeling_prismatic.py:946-1060.  head: 'l1' | 'discrete' | 'diffusion'.  Batch size 1."""
        cfg = self.cfg
        A = cfg.action_dim * cfg.chunk
        if not torch.all(input_ids[:, -1] == 29871):                                    # :974-977
            input_ids = torch.cat((input_ids, torch.tensor([[29871]], dtype=input_ids.dtype)), dim=1)
            attention_mask = torch.cat((attention_mask, torch.ones((1, 1), dtype=attention_mask.dtype)), dim=1)
        n_prompt = input_ids.shape[-1] - 1                                              # :987
        ids = torch.cat([input_ids, torch.ones((1, A), dtype=input_ids.dtype),
                         torch.full((1, 1), STOP_INDEX, dtype=input_ids.dtype)], dim=-1)  # :734-755
        mask = torch.cat([attention_mask, torch.ones((1, A + 1), dtype=attention_mask.dtype)], dim=-1)
        labels = torch.full_like(ids, IGNORE_INDEX)                                     # :983-984, :757-770
        labels[:, input_ids.shape[-1]:] = ACTION_TOKEN_BEGIN_IDX + 1
        labels[:, -1] = STOP_INDEX
        pv = self.R(pixel_values.float())
        prop = None if proprio is None else torch.as_tensor(np.asarray(proprio), dtype=torch.float32)
        if head == "diffusion":
            ddim = DDIM(num_diffusion_steps)
            ddim.set_timesteps(num_diffusion_steps)
            cur = self.R(noise.float())
            for t in ddim.timesteps:                                                    # :814-872
                temb = self.R(sinusoidal_encoding(torch.tensor([float(t)]), cfg.llm_dim))[:, None, :]
                hidden, P = self.multimodal_hidden(ids, mask, pv, labels, prop, cur, temb, use_film)
                ah = hidden[:, P + n_prompt: P + n_prompt + A]
                eps = self.noise_head(ah).reshape(cur.shape)
                cur = self.R(ddim.step(eps, int(t), cur))
            normalized = cur.reshape(cfg.chunk, cfg.action_dim).numpy()
        else:
            hidden, P = self.multimodal_hidden(ids, mask, pv, labels, prop, None, None, use_film)
            ah = hidden[:, P + n_prompt: P + n_prompt + A]                              # :915-920
            if head == "l1":
                normalized = self.l1_head(ah).reshape(cfg.chunk, cfg.action_dim).detach().numpy()  # :923-927
            else:                                                                       # :929-942
                logits = self.lm_logits(ah)
                tok = logits.argmax(dim=2).numpy()[0]
                normalized = decode_token_ids_to_actions(tok, cfg.vocab - cfg.pad_to_multiple_of, cfg.n_action_bins)
                normalized = normalized.reshape(cfg.chunk, cfg.action_dim)
        actions = normalized if unnorm_stats is None else unnormalize_actions(normalized, unnorm_stats, cfg.norm_type)
        return actions, ah


# ======================================================================================================================
# seeded random weights in the reference's parameter naming
# ======================================================================================================================
def random_state_dict(cfg: OracleConfig, seed: int = 0, lora: bool = True, film: bool = False, diffusion: bool = False,
                      dtype=torch.float32, std: float = 0.02) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def normal(*shape, s=std):
        return (torch.randn(*shape, generator=g) * s).to(dtype)

    def lin(name, out_f, in_f, bias=True, with_lora=lora, s=None):
        sd[name + ".weight"] = normal(out_f, in_f, s=s if s is not None else std)
        if bias:
            sd[name + ".bias"] = normal(out_f, s=0.02)
        if with_lora:
            sd[name + ".lora_A.weight"] = normal(cfg.lora_rank, in_f, s=1.0 / cfg.lora_rank)  # peft init "gaussian"
            sd[name + ".lora_B.weight"] = normal(out_f, cfg.lora_rank, s=0.01)                 # perturbed from 0

    def ln(name, dim):
        sd[name + ".weight"] = (1.0 + 0.1 * torch.randn(dim, generator=g)).to(dtype)
        sd[name + ".bias"] = normal(dim, s=0.05)

    for prefix, vc in (("vision_backbone.featurizer.", cfg.dino), ("vision_backbone.fused_featurizer.", cfg.siglip)):
        sd[prefix + "patch_embed.proj.weight"] = normal(vc.dim, 3, vc.patch, vc.patch, s=0.05)
        sd[prefix + "patch_embed.proj.bias"] = normal(vc.dim)
        sd[prefix + "pos_embed"] = normal(1, vc.n_patches, vc.dim, s=0.1)
        if vc.n_prefix > 0:
            sd[prefix + "cls_token"] = normal(1, 1, vc.dim, s=0.1)
            if vc.n_prefix > 1:
                sd[prefix + "reg_token"] = normal(1, vc.n_prefix - 1, vc.dim, s=0.1)
        for i in range(vc.depth):
            p = f"{prefix}blocks.{i}."
            ln(p + "norm1", vc.dim)
            lin(p + "attn.qkv", 3 * vc.dim, vc.dim, s=0.05)
            lin(p + "attn.proj", vc.dim, vc.dim, s=0.05)
            ln(p + "norm2", vc.dim)
            lin(p + "mlp.fc1", vc.mlp_hidden, vc.dim, s=0.05)
            lin(p + "mlp.fc2", vc.dim, vc.mlp_hidden, s=0.05)
            if vc.layerscale:
                sd[p + "ls1.scale_factor"] = (0.1 + 0.02 * torch.randn(vc.dim, generator=g)).to(dtype)
                sd[p + "ls2.scale_factor"] = (0.1 + 0.02 * torch.randn(vc.dim, generator=g)).to(dtype)
            if film:
                lin(p + "scale", vc.dim, cfg.llm_dim, with_lora=False, s=0.03)
                lin(p + "shift", vc.dim, cfg.llm_dim, with_lora=False, s=0.03)
    vis = cfg.dino.dim + cfg.siglip.dim
    lin("projector.fc1", 4 * vis, vis)
    lin("projector.fc2", cfg.llm_dim, 4 * vis)
    lin("projector.fc3", cfg.llm_dim, cfg.llm_dim)
    D = cfg.llm_dim
    sd["language_model.model.embed_tokens.weight"] = normal(cfg.vocab, D, s=0.5)
    for i in range(cfg.llm_layers):
        p = f"language_model.model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            lin(p + "self_attn." + n, D, D, bias=False, s=0.03)
        lin(p + "mlp.gate_proj", cfg.llm_ff, D, bias=False, s=0.03)
        lin(p + "mlp.up_proj", cfg.llm_ff, D, bias=False, s=0.03)
        lin(p + "mlp.down_proj", D, cfg.llm_ff, bias=False, s=0.03)
        sd[p + "input_layernorm.weight"] = (1.0 + 0.1 * torch.randn(D, generator=g)).to(dtype)
        sd[p + "post_attention_layernorm.weight"] = (1.0 + 0.1 * torch.randn(D, generator=g)).to(dtype)
    sd["language_model.model.norm.weight"] = (1.0 + 0.1 * torch.randn(D, generator=g)).to(dtype)
    lin("language_model.lm_head", cfg.vocab, D, bias=False, with_lora=False, s=0.05)
    # components (never LoRA'd: they are trained in full)
    lin("proprio_projector.fc1", D, cfg.proprio_dim, with_lora=False, s=0.3)
    lin("proprio_projector.fc2", D, D, with_lora=False, s=0.05)
    hp = "action_head.noise_predictor.mlp_resnet." if diffusion else "action_head.model."
    ln(hp + "layer_norm1", D * cfg.action_dim)
    lin(hp + "fc1", D, D * cfg.action_dim, with_lora=False, s=0.02)
    for b in range(2):
        ln(f"{hp}mlp_resnet_blocks.{b}.ffn.0", D)
        lin(f"{hp}mlp_resnet_blocks.{b}.ffn.1", D, D, with_lora=False, s=0.05)
    ln(hp + "layer_norm2", D)
    lin(hp + "fc2", cfg.action_dim, D, with_lora=False, s=0.05)
    if diffusion:
        lin("noisy_action_projector.fc1", D, 1, with_lora=False, s=0.5)
        lin("noisy_action_projector.fc2", D, D, with_lora=False, s=0.05)
    return sd
