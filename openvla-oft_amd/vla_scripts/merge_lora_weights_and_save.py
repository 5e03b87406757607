"""vla-scripts/merge_lora_weights_and_save.py of the reference, on the HIP engine.

The reference loads the base checkpoint, wraps it with `PeftModel.from_pretrained(<dir>/lora_adapter)`, calls
`merge_and_unload()` on the GPU and `save_pretrained`s the merged model next to the adapter (:44-67).  Here: the base
shards and the adapter are read from local directories (no hub), `W += (alpha / r) * B A` runs as one rank-r GEMM per
adapted Linear on the device (`LoraLinear.merge`), and the merged weights are written back as HF-layout safetensors shards
(+ `model.safetensors.index.json`) with the base checkpoint's own key names, so `get_vla` can load the result.
"""
from __future__ import annotations

import json
import time
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, Union

import torch

from ..config import OPENVLA_7B, VLAConfig
from ..modeling import OpenVLAForActionPrediction
from ..weights import load_lora_adapter


@dataclass
class ConvertConfig:
    base_checkpoint: Union[str, Path] = "openvla/openvla-7b"          # local directory with *.safetensors shards
    lora_finetuned_checkpoint_dir: Union[str, Path] = ""              # directory containing lora_adapter/
    max_shard_bytes: int = 5 << 30


def _read_shards(ckpt: Path) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file

    if not ckpt.is_dir():
        raise ValueError(f"`{ckpt}` is not a local checkpoint directory (HF-hub checkpoints cannot be fetched offline)")
    sd: Dict[str, torch.Tensor] = {}
    for shard in sorted(ckpt.glob("*.safetensors")):
        sd.update(load_file(str(shard)))
    if not sd:
        raise ValueError(f"no *.safetensors shards in {ckpt}")
    return sd


def save_sharded(sd: Dict[str, torch.Tensor], out_dir: Path, max_shard_bytes: int = 5 << 30):
    """HF `save_pretrained` layout: model-0000i-of-0000n.safetensors + model.safetensors.index.json (one file: model.safetensors)."""
    from safetensors.torch import save_file

    shards, cur, size = [], {}, 0
    for k, v in sd.items():
        nb = v.numel() * v.element_size()
        if cur and size + nb > max_shard_bytes:
            shards.append(cur)
            cur, size = {}, 0
        cur[k] = v.detach().to("cpu").contiguous()
        size += nb
    shards.append(cur)
    out_dir.mkdir(parents=True, exist_ok=True)
    if len(shards) == 1:
        save_file(shards[0], str(out_dir / "model.safetensors"), metadata={"format": "pt"})
        return [out_dir / "model.safetensors"]
    index, files = {"metadata": {"total_size": sum(v.numel() * v.element_size() for v in sd.values())}, "weight_map": {}}, []
    for i, sh in enumerate(shards):
        name = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(sh, str(out_dir / name), metadata={"format": "pt"})
        files.append(out_dir / name)
        for k in sh:
            index["weight_map"][k] = name
    (out_dir / "model.safetensors.index.json").write_text(json.dumps(index, indent=2))
    return files


def main(cfg: ConvertConfig, model_config: VLAConfig = OPENVLA_7B, device=None) -> Path:
    base = _read_shards(Path(cfg.base_checkpoint))
    out_dir = Path(cfg.lora_finetuned_checkpoint_dir)
    adapter, acfg = load_lora_adapter(out_dir / "lora_adapter")
    r, alpha = acfg.get("r", model_config.lora_rank), acfg.get("lora_alpha", model_config.lora_alpha)
    if (r, alpha) != (model_config.lora_rank, model_config.lora_alpha):
        import dataclasses

        model_config = dataclasses.replace(model_config, lora_rank=r, lora_alpha=alpha)
    missing = [k for k in adapter if k.rsplit(".lora_", 1)[0] + ".weight" not in base]
    if missing:
        raise ValueError(f"adapter tensors without a base Linear in {cfg.base_checkpoint}: {missing[:4]} ...")
    print(f"Loading base model: {cfg.base_checkpoint}")
    t0 = time.time()
    vla = OpenVLAForActionPrediction(model_config, {**base, **adapter}, device=device, lora=True)
    print("Merging LoRA weights into base model...")
    vla.merge_and_unload()
    merged = dict(base)
    for k, v in vla.engine.merged_state_dict().items():
        if k in merged:                      # adapted Linears only; norms, embeddings, biases, lm_head pass through unchanged
            merged[k] = v.to(merged[k].dtype)
    torch.cuda.synchronize()
    save_sharded(merged, out_dir, cfg.max_shard_bytes)
    print(f"\nMerging complete! Time elapsed (sec): {time.time() - t0}")
    print(f"\nSaved merged model checkpoint at:\n{out_dir}")
    return out_dir
