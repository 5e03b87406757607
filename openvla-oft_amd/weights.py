"""Weights in the reference's parameter naming -> device tensors for the engine.

State-dict names follow the reference's HF checkpoint layout (vla-scripts/extern/convert_openvla_weights_to_hf.py:73-115):
  vision_backbone.featurizer.*        DINOv2 (timm names; LayerScale as `ls{1,2}.scale_factor`, modeling_prismatic.py:60-64)
  vision_backbone.fused_featurizer.*  SigLIP
  projector.fc{1,2,3}.*               language_model.model.* / language_model.lm_head.weight
plus the fine-tune components as saved by finetune.py:584-675: action_head.model.* (or action_head.noise_predictor.mlp_resnet.*),
proprio_projector.fc{1,2}.*, noisy_action_projector.fc{1,2}.*, and LoRA adapters as <linear>.lora_A.weight / .lora_B.weight.
"""
from __future__ import annotations

from typing import Callable, Dict

import torch

from .config import VLAConfig

BF16 = torch.bfloat16


def make_getter(sd: Dict[str, torch.Tensor], device) -> tuple[Callable[[str], torch.Tensor], Callable[[str], bool]]:
    def get(name: str) -> torch.Tensor:
        if name not in sd:
            raise KeyError(f"state dict has no parameter `{name}`")
        return sd[name].detach().to(device=device, dtype=BF16).contiguous()

    return get, (lambda name: name in sd)


def random_state_dict(cfg: VLAConfig, device, seed: int = 0, *, lora: bool = True, diffusion: bool = False, lm_head: bool = True,
                      film: bool = False, dtype=BF16) -> Dict[str, torch.Tensor]:
    """Seeded random weights of the configured architecture, generated ON DEVICE (no checkpoint exists offline; SURVEY.md
    section 8d config 3: N(0, 0.02)-style init, norm weights ~1, LayerScale 0.1, LoRA A ~ N(0, 1/r), B perturbed from 0)."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def normal(*shape, s=0.02):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * s).to(dtype)

    def around_one(n, s=0.1):
        return (1.0 + s * torch.randn(n, generator=g, device=device)).to(dtype)

    def lin(name, out_f, in_f, bias=True, with_lora=lora, s=0.02):
        sd[name + ".weight"] = normal(out_f, in_f, s=s)
        if bias:
            sd[name + ".bias"] = normal(out_f, s=0.02)
        if with_lora:
            sd[name + ".lora_A.weight"] = normal(cfg.lora_rank, in_f, s=1.0 / cfg.lora_rank)
            sd[name + ".lora_B.weight"] = normal(out_f, cfg.lora_rank, s=0.01)

    def ln(name, dim):
        sd[name + ".weight"] = around_one(dim)
        sd[name + ".bias"] = normal(dim, s=0.05)

    for prefix, vc in (("vision_backbone.featurizer.", cfg.dino), ("vision_backbone.fused_featurizer.", cfg.siglip)):
        sd[prefix + "patch_embed.proj.weight"] = normal(vc.dim, 3, vc.patch, vc.patch, s=0.05)
        sd[prefix + "patch_embed.proj.bias"] = normal(vc.dim)
        sd[prefix + "pos_embed"] = normal(1, vc.n_patches, vc.dim, s=0.1)
        if vc.n_prefix > 0:
            sd[prefix + "cls_token"] = normal(1, 1, vc.dim, s=0.1)
            if vc.n_prefix > 1:
                sd[prefix + "reg_token"] = normal(1, vc.n_prefix - 1, vc.dim, s=0.1)
        for i in range(vc.depth - 1):  # the last block's output is never used on this path
            p = f"{prefix}blocks.{i}."
            ln(p + "norm1", vc.dim)
            lin(p + "attn.qkv", 3 * vc.dim, vc.dim, s=0.03)
            lin(p + "attn.proj", vc.dim, vc.dim, s=0.03)
            ln(p + "norm2", vc.dim)
            lin(p + "mlp.fc1", vc.mlp_hidden, vc.dim, s=0.03)
            lin(p + "mlp.fc2", vc.dim, vc.mlp_hidden, s=0.03)
            if vc.layerscale:
                sd[p + "ls1.scale_factor"] = (0.1 + 0.02 * torch.randn(vc.dim, generator=g, device=device)).to(dtype)
                sd[p + "ls2.scale_factor"] = (0.1 + 0.02 * torch.randn(vc.dim, generator=g, device=device)).to(dtype)
            if film:
                lin(p + "scale", vc.dim, cfg.llm_dim, with_lora=False, s=0.01)
                lin(p + "shift", vc.dim, cfg.llm_dim, with_lora=False, s=0.01)
    vis, D = cfg.vision_dim, cfg.llm_dim
    lin("projector.fc1", 4 * vis, vis)
    lin("projector.fc2", D, 4 * vis)
    lin("projector.fc3", D, D)
    sd["language_model.model.embed_tokens.weight"] = normal(cfg.vocab, D, s=0.5)
    for i in range(cfg.llm_layers):
        p = f"language_model.model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            lin(p + "self_attn." + n, D, D, bias=False)
        lin(p + "mlp.gate_proj", cfg.llm_ff, D, bias=False)
        lin(p + "mlp.up_proj", cfg.llm_ff, D, bias=False)
        lin(p + "mlp.down_proj", D, cfg.llm_ff, bias=False)
        sd[p + "input_layernorm.weight"] = around_one(D)
        sd[p + "post_attention_layernorm.weight"] = around_one(D)
    sd["language_model.model.norm.weight"] = around_one(D)
    if lm_head:
        lin("language_model.lm_head", cfg.vocab, D, bias=False, with_lora=False, s=0.05)
    lin("proprio_projector.fc1", D, cfg.proprio_dim, with_lora=False, s=0.3)
    lin("proprio_projector.fc2", D, D, with_lora=False, s=0.02)
    hp = "action_head.noise_predictor.mlp_resnet." if diffusion else "action_head.model."
    ln(hp + "layer_norm1", D * cfg.action_dim)
    lin(hp + "fc1", D, D * cfg.action_dim, with_lora=False, s=0.01)
    for b in range(2):
        ln(f"{hp}mlp_resnet_blocks.{b}.ffn.0", D)
        lin(f"{hp}mlp_resnet_blocks.{b}.ffn.1", D, D, with_lora=False, s=0.02)
    ln(hp + "layer_norm2", D)
    lin(hp + "fc2", cfg.action_dim, D, with_lora=False, s=0.05)
    if diffusion:
        lin("noisy_action_projector.fc1", D, 1, with_lora=False, s=0.5)
        lin("noisy_action_projector.fc2", D, D, with_lora=False, s=0.02)
    return sd


# ----------------------------------------------------------------------------------------------------------------------
# peft LoRA adapter directory (`lora_adapter/`: finetune.py:617-619 `vla.module.save_pretrained(adapter_dir)`, read back by
# `PeftModel.from_pretrained` in merge_lora_weights_and_save.py:58-60).  peft 0.11.1 (pyproject.toml) is absent from this
# image; the layout below is its published on-disk format: `adapter_config.json` + `adapter_model.safetensors` whose keys
# are `base_model.model.<module path>.lora_{A,B}.weight` (the adapter name "default" is dropped when saving).
# ----------------------------------------------------------------------------------------------------------------------
PEFT_PREFIX = "base_model.model."


def save_lora_adapter(adapter_dir, tensors: Dict[str, torch.Tensor], *, r: int, lora_alpha: int, base_model_name_or_path: str = "openvla/openvla-7b"):
    """`tensors`: {<module path>.lora_A.weight | .lora_B.weight: tensor} (engine.export_trainable naming)."""
    import json
    from pathlib import Path

    from safetensors.torch import save_file

    d = Path(adapter_dir)
    d.mkdir(parents=True, exist_ok=True)
    lora = {PEFT_PREFIX + k: v.detach().to("cpu").contiguous() for k, v in tensors.items() if ".lora_A." in k or ".lora_B." in k}
    save_file(lora, str(d / "adapter_model.safetensors"))
    cfg = {"peft_type": "LORA", "task_type": None, "base_model_name_or_path": base_model_name_or_path, "r": r, "lora_alpha": lora_alpha,
           "lora_dropout": 0.0, "bias": "none", "target_modules": "all-linear", "init_lora_weights": "gaussian", "fan_in_fan_out": False,
           "inference_mode": True, "modules_to_save": None, "use_rslora": False, "use_dora": False}
    (d / "adapter_config.json").write_text(json.dumps(cfg, indent=2, sort_keys=True))
    return d


def load_lora_adapter(adapter_dir) -> tuple[Dict[str, torch.Tensor], dict]:
    """-> ({<module path>.lora_A.weight: tensor, ...}, adapter_config).  Accepts keys with or without peft's
    `base_model.model.` prefix and with an explicit `.default` adapter name."""
    import json
    from pathlib import Path

    from safetensors.torch import load_file

    d = Path(adapter_dir)
    f = d / "adapter_model.safetensors"
    if not f.is_file():
        raise ValueError(f"no adapter_model.safetensors in {d} (peft .bin adapters are pickles and are not loaded)")
    cfg = json.loads((d / "adapter_config.json").read_text()) if (d / "adapter_config.json").is_file() else {}
    if cfg and cfg.get("peft_type", "LORA") != "LORA":
        raise ValueError(f"unsupported peft_type {cfg.get('peft_type')!r}")
    if cfg.get("use_rslora") or cfg.get("use_dora"):
        raise ValueError("rsLoRA / DoRA adapters are not produced by the reference (finetune.py:862-871) and are not supported")
    out = {}
    for k, v in load_file(str(f)).items():
        if k.startswith(PEFT_PREFIX):
            k = k[len(PEFT_PREFIX):]
        k = k.replace(".lora_A.default.", ".lora_A.").replace(".lora_B.default.", ".lora_B.")
        out[k] = v
    return out, cfg


# ----------------------------------------------------------------------------------------------------------------------
# `vision_backbone--{step}_checkpoint.pt`: the FiLM-wrapped backbone's state dict as the reference writes and reads it
# ----------------------------------------------------------------------------------------------------------------------
def vision_backbone_keys_to_reference(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Engine / HF-checkpoint names (`vision_backbone.featurizer.blocks.3.attn.qkv.weight`, `.lora_A.weight`, `blocks.3.scale.weight`) ->
    the key layout of `FiLMedPrismaticVisionBackbone.state_dict()` after peft's `get_peft_model`, which is what
    vla-scripts/finetune.py:640-655 saves and experiments/robot/openvla_utils.py:340-342 loads: the wrapper holds the backbone as
    `.vision_backbone` (film_vit_wrapper.py:192), every block sits under `.block` of its FiLM wrapper (:49-54) with `scale` / `shift` beside it,
    and peft renames an adapted Linear's tensors to `base_layer.{weight,bias}` / `lora_{A,B}.default.weight`."""
    out = {}
    for k, v in sd.items():
        if not k.startswith("vision_backbone."):
            continue
        parts = k.split(".")
        if "blocks" in parts:
            i = parts.index("blocks")
            if parts[i + 2] not in ("scale", "shift"):
                parts.insert(i + 2, "block")
        k2 = ".".join(parts)
        if ".lora_A.weight" in k2 or ".lora_B.weight" in k2:
            k2 = k2.replace(".lora_A.weight", ".lora_A.default.weight").replace(".lora_B.weight", ".lora_B.default.weight")
        elif any(k2.endswith(f".{lin}.{t}") for lin in ("qkv", "proj", "fc1", "fc2") for t in ("weight", "bias")) and ".patch_embed." not in k2:
            stem, t = k2.rsplit(".", 1)
            if stem + ".lora_A.default.weight" in out or (k.rsplit(".", 1)[0] + ".lora_A.weight") in sd:
                k2 = f"{stem}.base_layer.{t}"
        out[k2] = v
    return out


def vision_backbone_keys_from_reference(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Inverse of `vision_backbone_keys_to_reference`, tolerant of DDP's `module.` prefix (finetune.py:134-156) and of files that hold the FiLM
    tensors only (what this repo wrote before round 3: `featurizer.blocks.3.scale.weight`)."""
    out = {}
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[len("module."):]
        if not k.startswith("vision_backbone."):
            k = "vision_backbone." + k
        k = k.replace(".block.", ".").replace(".base_layer.", ".").replace(".lora_A.default.", ".lora_A.").replace(".lora_B.default.", ".lora_B.")
        out[k] = v
    return out
