"""Model configuration of the OpenVLA-OFT path (the fields of the reference's OpenVLAConfig / timm model ids that the
hot path actually depends on; prismatic/extern/hf/configuration_prismatic.py:15-45,72-140, prismatic/vla/constants.py)."""
from __future__ import annotations

from dataclasses import dataclass, field, asdict


@dataclass
class VitConfig:
    dim: int
    depth: int
    heads: int
    mlp_hidden: int
    n_prefix: int = 0          # cls + register tokens (vit_large_patch14_reg4_dinov2: 1 + 4; vit_so400m_patch14_siglip_224: 0)
    layerscale: bool = False
    patch: int = 14
    image_size: int = 224
    eps: float = 1e-6
    act: str = "gelu"

    @property
    def n_patches(self) -> int:
        return (self.image_size // self.patch) ** 2

    @property
    def head_dim(self) -> int:
        return self.dim // self.heads

    @property
    def patch_k(self) -> int:          # im2col K padded to a multiple of 8 (588 -> 592)
        k = 3 * self.patch * self.patch
        return (k + 7) // 8 * 8


@dataclass
class VLAConfig:
    llm_dim: int = 4096
    llm_layers: int = 32
    llm_heads: int = 32
    llm_ff: int = 11008
    vocab: int = 32064
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0
    max_positions: int = 2048          # llm_max_length, configuration_prismatic.py:84
    dino: VitConfig = field(default_factory=lambda: VitConfig(1024, 24, 16, 4096, n_prefix=5, layerscale=True))
    siglip: VitConfig = field(default_factory=lambda: VitConfig(1152, 27, 16, 4304))
    num_images: int = 2
    lora_rank: int = 32
    lora_alpha: int = 16
    action_dim: int = 7
    chunk: int = 8
    proprio_dim: int = 8
    norm_type: str = "bounds_q99"
    n_action_bins: int = 256
    pad_to_multiple_of: int = 64
    pad_token_id: int = 32000
    mask_mode: str = "bidirectional"   # reference fork's non-causal attention | "causal"

    @property
    def lora_scale(self) -> float:
        return self.lora_alpha / self.lora_rank

    @property
    def vision_dim(self) -> int:
        return self.dino.dim + self.siglip.dim

    @property
    def num_action_tokens(self) -> int:
        return self.action_dim * self.chunk

    @classmethod
    def from_any(cls, other) -> "VLAConfig":
        """Builds from any dataclass/dict carrying the same field names (e.g. the test oracle's config)."""
        d = asdict(other) if not isinstance(other, dict) else dict(other)
        d["dino"] = VitConfig(**d["dino"])
        d["siglip"] = VitConfig(**d["siglip"])
        return cls(**{k: v for k, v in d.items() if k in cls.__dataclass_fields__})


OPENVLA_7B = VLAConfig()  # prism-dinosiglip-224px+7b (prismatic/conf/models.py:482-498), LIBERO constants
