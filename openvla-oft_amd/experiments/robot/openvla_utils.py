"""Inference glue (mirror of experiments/robot/openvla_utils.py: get_vla :253, get_processor :380, get_proprio_projector
:393, get_action_head :463, normalize_proprio :645, prepare_images_for_vla :678, get_vla_action :711).

Differences forced by the offline MI355X environment, all explicit:
  * checkpoints are local directories of safetensors / .pt files in the reference layout (no HF-hub fetch);
  * the processor wraps a caller-supplied tokenizer (no tokenizer files exist offline) -- anything with
    `__call__(text) -> list[int]` works, e.g. transformers.LlamaTokenizerFast when its files are present.
"""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from ... import image_prep, ops
from ...config import OPENVLA_7B, VLAConfig
from ...modeling import DiffusionActionHead, L1RegressionActionHead, NoisyActionProjector, OpenVLAForActionPrediction, ProprioProjector
from ...prismatic.vla import constants as C

DEVICE = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")
OPENVLA_IMAGE_SIZE = 224


def _load_tensors(path: Path) -> Dict[str, torch.Tensor]:
    if path.suffix == ".safetensors":
        from safetensors.torch import load_file

        return load_file(str(path))
    return torch.load(str(path), weights_only=True, map_location="cpu")


def load_component_state_dict(checkpoint_path) -> Dict[str, torch.Tensor]:
    """openvla_utils.py:230-250 (strips DDP's `module.` prefix)."""
    sd = _load_tensors(Path(checkpoint_path))
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def find_checkpoint_file(pretrained_checkpoint: str, file_pattern: str) -> str:
    """openvla_utils.py:201-227"""
    files = [os.path.join(pretrained_checkpoint, f) for f in os.listdir(pretrained_checkpoint) if file_pattern in f and "checkpoint" in f]
    assert len(files) == 1, f"Expected exactly 1 {file_pattern} checkpoint but found {len(files)} in directory: {pretrained_checkpoint}"
    return files[0]


def platform_model_config(model_config: VLAConfig = OPENVLA_7B, num_images: Optional[int] = None) -> VLAConfig:
    """The architecture config with the robot platform's constants filled in from prismatic.vla.constants, which is where the
    reference reads them everywhere (ACTION_DIM / NUM_ACTIONS_CHUNK in modeling_prismatic.py:36-44, the normalisation type in
    :772-791 and openvla_utils.py:645-675): LIBERO 8 x 7 / proprio 8 / q01-q99 bounds, ALOHA 25 x 14 / proprio 14 / min-max bounds."""
    import dataclasses

    kw = dict(action_dim=C.ACTION_DIM, chunk=C.NUM_ACTIONS_CHUNK, proprio_dim=C.PROPRIO_DIM, norm_type=C.ACTION_PROPRIO_NORMALIZATION_TYPE.value)
    if num_images is not None:
        kw["num_images"] = int(num_images)
    return dataclasses.replace(model_config, **kw)


def get_vla(cfg: Any, model_config: VLAConfig = OPENVLA_7B) -> OpenVLAForActionPrediction:
    """Loads the HF-layout checkpoint shards of `cfg.pretrained_checkpoint` (safetensors) into the HIP engine.  Action / proprio
    dimensions and the un-normalisation rule follow prismatic.vla.constants (the active platform), like get_action_head /
    get_proprio_projector / normalize_proprio below and finetune()."""
    model_config = platform_model_config(model_config, getattr(cfg, "num_images_in_input", None))
    ckpt = Path(cfg.pretrained_checkpoint)
    if not ckpt.is_dir():
        raise ValueError(f"`{ckpt}` is not a local checkpoint directory (HF-hub checkpoints cannot be fetched offline)")
    if getattr(cfg, "load_in_8bit", False) or getattr(cfg, "load_in_4bit", False):
        raise NotImplementedError("bitsandbytes quantised loading is not part of the MI355X path")
    sd: Dict[str, torch.Tensor] = {}
    for shard in sorted(ckpt.glob("*.safetensors")):
        sd.update(_load_tensors(shard))
    if not sd:
        raise ValueError(f"no *.safetensors shards in {ckpt}")
    use_film = bool(getattr(cfg, "use_film", False))
    if use_film:
        sd.update(_film_vision_backbone(cfg, model_config))
    vla = OpenVLAForActionPrediction(model_config, sd, device=DEVICE, use_film=use_film)
    vla.vision_backbone.set_num_images_in_input(cfg.num_images_in_input)
    stats = ckpt / "dataset_statistics.json"
    if stats.is_file():                                                     # openvla_utils.py:352-377
        vla.norm_stats = json.loads(stats.read_text())
    vla.eval()
    return vla


def _film_vision_backbone(cfg: Any, model_config: VLAConfig) -> Dict[str, torch.Tensor]:
    """openvla_utils.py:311-349 (`_apply_film_to_vla`): the FiLM evaluation path re-attaches LoRA (r = 32, alpha = 16) to the model, wraps the
    vision backbone with FiLM and loads `vision_backbone--{step}_checkpoint.pt` -- the WHOLE wrapped backbone as training saved it
    (finetune.py:640-655): the towers' base weights, their adapters and the scale / shift Linears.  Those tensors replace the checkpoint
    shards' vision backbone (whose adapters, in a merged checkpoint, are already folded in), exactly as `load_state_dict` does there; the
    decoder and projector keep the shards' weights.  Returns the tensors under the engine's names."""
    from ...weights import vision_backbone_keys_from_reference

    if (model_config.lora_rank, model_config.lora_alpha) != (32, 16):
        raise ValueError("the FiLM evaluation path re-creates the adapters with r=32, lora_alpha=16 (openvla_utils.py:325-332)")
    part = vision_backbone_keys_from_reference(load_component_state_dict(find_checkpoint_file(cfg.pretrained_checkpoint, "vision_backbone")))
    if not any(".scale.weight" in k for k in part):
        raise ValueError("the vision_backbone checkpoint holds no FiLM scale / shift tensors")
    return part


def get_proprio_projector(cfg: Any, llm_dim: int, proprio_dim: int) -> ProprioProjector:
    sd = load_component_state_dict(find_checkpoint_file(cfg.pretrained_checkpoint, "proprio_projector"))
    return ProprioProjector(llm_dim, proprio_dim, device=DEVICE, state_dict=sd).eval()


def get_noisy_action_projector(cfg: Any, llm_dim: int) -> NoisyActionProjector:
    sd = load_component_state_dict(find_checkpoint_file(cfg.pretrained_checkpoint, "noisy_action_projector"))
    return NoisyActionProjector(llm_dim, device=DEVICE, state_dict=sd).eval()


def get_action_head(cfg: Any, llm_dim: int):
    assert not (cfg.use_l1_regression and cfg.use_diffusion), "Cannot use both L1 regression and diffusion action head!"
    if not (cfg.use_l1_regression or cfg.use_diffusion):
        raise ValueError("Either use_l1_regression or use_diffusion must be True")
    sd = load_component_state_dict(find_checkpoint_file(cfg.pretrained_checkpoint, "action_head"))
    if cfg.use_diffusion:
        return DiffusionActionHead(llm_dim, llm_dim, C.ACTION_DIM, num_diffusion_steps=cfg.num_diffusion_steps, num_actions_chunk=C.NUM_ACTIONS_CHUNK,
                                   device=DEVICE, state_dict=sd).eval()
    return L1RegressionActionHead(llm_dim, llm_dim, C.ACTION_DIM, num_actions_chunk=C.NUM_ACTIONS_CHUNK, device=DEVICE, state_dict=sd).eval()


class PrismaticProcessor:
    """processor(prompt, image) -> {"input_ids", "attention_mask", "pixel_values"} (processing_prismatic.py:175-252)."""

    def __init__(self, tokenizer, device_image_prep: bool = True):
        self.tokenizer = tokenizer
        self.device_image_prep = device_image_prep   # get_vla_action prepares the images on the GPU (ops.image_prep) when it can

    def tokenize(self, text: str) -> Dict[str, torch.Tensor]:
        ids = list(self.tokenizer(text))
        return {"input_ids": torch.tensor([ids], dtype=torch.int64), "attention_mask": torch.ones((1, len(ids)), dtype=torch.bool)}

    def __call__(self, text: str, image: np.ndarray):
        return {**self.tokenize(text), "pixel_values": image_prep.apply_transform(np.asarray(image))[None]}


def get_processor(cfg: Any, tokenizer=None) -> PrismaticProcessor:
    if tokenizer is None:
        from transformers import AutoTokenizer  # needs tokenizer files inside the checkpoint directory

        tok = AutoTokenizer.from_pretrained(cfg.pretrained_checkpoint)
        tokenizer = lambda text: tok(text, add_special_tokens=True).input_ids  # noqa: E731
    return PrismaticProcessor(tokenizer)


def normalize_proprio(proprio: np.ndarray, norm_stats: Dict[str, Any]) -> np.ndarray:
    """openvla_utils.py:645-675"""
    if C.ACTION_PROPRIO_NORMALIZATION_TYPE == C.NormalizationType.BOUNDS:
        mask = norm_stats.get("mask", np.ones_like(norm_stats["min"], dtype=bool))
        high, low = np.array(norm_stats["max"]), np.array(norm_stats["min"])
    elif C.ACTION_PROPRIO_NORMALIZATION_TYPE == C.NormalizationType.BOUNDS_Q99:
        mask = norm_stats.get("mask", np.ones_like(norm_stats["q01"], dtype=bool))
        high, low = np.array(norm_stats["q99"]), np.array(norm_stats["q01"])
    else:
        raise ValueError("Unsupported action/proprio normalization type detected!")
    return np.clip(np.where(mask, 2 * (proprio - low) / (high - low + 1e-8) - 1, proprio), a_min=-1.0, a_max=1.0)


prepare_images_for_vla = image_prep.prepare_images_for_vla


def device_pixel_values(images: List[np.ndarray], cfg: Any) -> torch.Tensor:
    """prepare_images_for_vla + processor(...)["pixel_values"] of every image, concatenated on dim 1, computed on the device: the
    uint8 frames go to HBM as they are (150 KB each); center crop, uint8 re-quantisation and both backbones' normalisations are ONE
    kernel (ovla_image_prep), bit-identical to image_prep.center_crop_image + apply_transform; frames of another size first go through the
    reference's resize_image_for_policy on the device (ovla_jpeg_roundtrip, then ovla_image_resize).  -> bf16 [1, 6 * n, 224, 224]."""
    side = image_prep.OPENVLA_IMAGE_SIZE
    for image in images:
        image_prep.check_image_format(image)
    if any(image.shape != images[0].shape for image in images):
        images = [image if image.shape == (side, side, 3) else image_prep.resize_image_for_policy(image, side, device=DEVICE) for image in images]
    frames = torch.from_numpy(np.ascontiguousarray(np.stack(images))).to(DEVICE, non_blocking=True)
    if frames.shape[1:3] != (side, side):      # JPEG round trip + lanczos3 antialias resize of all frames, two launches each
        frames = ops.jpeg_roundtrip(frames)
        spans = [tuple(torch.from_numpy(a).to(DEVICE) for a in image_prep.lanczos3_spans(n, side)) for n in frames.shape[1:3]]
        frames = ops.image_resize(frames, spans[0], spans[1])
    return ops.image_prep(frames, crop=bool(cfg.center_crop))


def get_vla_action(cfg: Any, vla, processor: Any, obs: Dict[str, Any], task_label: str, action_head=None, proprio_projector=None,
                   noisy_action_projector=None, use_film: bool = False) -> List[np.ndarray]:
    """openvla_utils.py:711-796.  Note: like the reference, overwrites obs["state"] with the normalised proprio."""
    with torch.inference_mode():
        all_images = [obs["full_image"]]
        if cfg.num_images_in_input > 1:
            all_images.extend([obs[k] for k in obs.keys() if "wrist" in k or "camera_gripper_image" in k])
        prompt = f"In: What action should the robot take to {task_label.lower()}?\nOut:"
        if (isinstance(processor, PrismaticProcessor) and processor.device_image_prep and torch.cuda.is_available()
                and vla.config.image_sizes[0] == image_prep.OPENVLA_IMAGE_SIZE):
            inputs = processor.tokenize(prompt)
            inputs["pixel_values"] = device_pixel_values(all_images, cfg)
        else:
            all_images = prepare_images_for_vla(all_images, cfg)
            primary = all_images.pop(0)
            inputs = processor(prompt, primary)
            if all_images:
                wrist = [processor(prompt, im)["pixel_values"] for im in all_images]
                inputs["pixel_values"] = torch.cat([inputs["pixel_values"]] + wrist, dim=1)
        proprio = None
        if cfg.use_proprio:
            obs["state"] = normalize_proprio(obs["state"], vla.norm_stats[cfg.unnorm_key]["proprio"])
            proprio = obs["state"]
        if action_head is None:
            action, _ = vla.predict_action(**inputs, unnorm_key=cfg.unnorm_key, do_sample=False)
        else:
            action, _ = vla.predict_action(**inputs, unnorm_key=cfg.unnorm_key, do_sample=False, proprio=proprio,
                                           proprio_projector=proprio_projector, noisy_action_projector=noisy_action_projector,
                                           action_head=action_head, use_film=use_film)
    return [action[i] for i in range(min(len(action), cfg.num_open_loop_steps))]
