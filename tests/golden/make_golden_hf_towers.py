"""Generates tests/golden/g15_hf_towers.npz: a whole tiny DINOv2-with-registers tower and a whole tiny SigLIP vision tower evaluated by the HuggingFace
`transformers` models (Dinov2WithRegistersModel / SiglipVisionModel, stock transformers in the build container) on seeded weights and images, fp32.
Pins the TOWER-LEVEL wiring of oracle.vla_oracle.Oracle.vit -- patch embedding, position embedding, prefix tokens [cls, 4 registers, patches], which
block's output is taken ("block index depth - 2", no final norm: `hidden_states[-2]` of the HF models == timm's get_intermediate_layers(n={depth-2}),
modeling_prismatic.py:127-139) -- against independent implementations of the same published architectures (timm 0.9.10 is not installable here; G14
pins one block, G7 the towers only against the oracle itself).

Weights are stored under the oracle's timm-style names.  One conversion is not a rename: HF adds position_embeddings[:, 0] to the cls token, timm's
`no_embed_class=True` DINOv2 models carry that sum inside `cls_token` (the timm checkpoint conversion folds it), so cls_token := cls + pos[:, 0].

  python tests/golden/make_golden_hf_towers.py
"""
from pathlib import Path

import numpy as np
import torch
from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel, SiglipVisionConfig, SiglipVisionModel


def randomise(mod, g):
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if "lambda" in n:
                p.copy_(0.1 + 0.02 * torch.randn(p.shape, generator=g))
            elif "norm" in n and n.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() >= 2 else 0.03))


def dino_layer(sd, i):
    p = f"encoder.layer.{i}."
    q = [sd[p + f"attention.attention.{n}.weight"] for n in ("query", "key", "value")]
    b = [sd[p + f"attention.attention.{n}.bias"] for n in ("query", "key", "value")]
    return {"norm1.weight": sd[p + "norm1.weight"], "norm1.bias": sd[p + "norm1.bias"], "attn.qkv.weight": torch.cat(q, 0), "attn.qkv.bias": torch.cat(b, 0),
            "attn.proj.weight": sd[p + "attention.output.dense.weight"], "attn.proj.bias": sd[p + "attention.output.dense.bias"],
            "ls1.scale_factor": sd[p + "layer_scale1.lambda1"], "norm2.weight": sd[p + "norm2.weight"], "norm2.bias": sd[p + "norm2.bias"],
            "mlp.fc1.weight": sd[p + "mlp.fc1.weight"], "mlp.fc1.bias": sd[p + "mlp.fc1.bias"], "mlp.fc2.weight": sd[p + "mlp.fc2.weight"],
            "mlp.fc2.bias": sd[p + "mlp.fc2.bias"], "ls2.scale_factor": sd[p + "layer_scale2.lambda1"]}


def siglip_layer(sd, i, root=""):
    p = f"{root}encoder.layers.{i}."
    return {"norm1.weight": sd[p + "layer_norm1.weight"], "norm1.bias": sd[p + "layer_norm1.bias"],
            "attn.qkv.weight": torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0),
            "attn.qkv.bias": torch.cat([sd[p + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0),
            "attn.proj.weight": sd[p + "self_attn.out_proj.weight"], "attn.proj.bias": sd[p + "self_attn.out_proj.bias"],
            "norm2.weight": sd[p + "layer_norm2.weight"], "norm2.bias": sd[p + "layer_norm2.bias"],
            "mlp.fc1.weight": sd[p + "mlp.fc1.weight"], "mlp.fc1.bias": sd[p + "mlp.fc1.bias"], "mlp.fc2.weight": sd[p + "mlp.fc2.weight"], "mlp.fc2.bias": sd[p + "mlp.fc2.bias"]}


def main():
    g = torch.Generator().manual_seed(15)
    out = {}
    img = torch.randn(2, 3, 42, 42, generator=g)       # 3 x 3 patches of 14
    # ---- DINOv2 reg4: depth 3, head_dim 64, taken after block index 1
    dc = Dinov2WithRegistersConfig(hidden_size=64, num_hidden_layers=3, num_attention_heads=1, mlp_ratio=2, image_size=42, patch_size=14,
                                   num_register_tokens=4, layerscale_value=0.1, hidden_act="gelu", layer_norm_eps=1e-6, use_swiglu_ffn=False,
                                   attn_implementation="eager")
    dm = Dinov2WithRegistersModel(dc).eval()
    randomise(dm, g)
    with torch.no_grad():
        hs = dm(img, output_hidden_states=True).hidden_states
    assert len(hs) == 4
    y = hs[-2][:, 5:]                                    # output of block index depth - 2, prefix tokens dropped, no final norm
    sd = dict(dm.state_dict())
    pos = sd["embeddings.position_embeddings"]
    w = {"patch_embed.proj.weight": sd["embeddings.patch_embeddings.projection.weight"], "patch_embed.proj.bias": sd["embeddings.patch_embeddings.projection.bias"],
         "pos_embed": pos[:, 1:], "cls_token": sd["embeddings.cls_token"] + pos[:, :1], "reg_token": sd["embeddings.register_tokens"]}
    for i in range(3):
        w.update({f"blocks.{i}.{k}": v for k, v in dino_layer(sd, i).items()})
    out.update({"dino__" + k: v.detach().numpy() for k, v in w.items()})
    out["dino__x"], out["dino__y"] = img.numpy(), y.numpy()
    # ---- SigLIP: depth 3, head_dim 72, no prefix tokens, learned position embedding
    sc = SiglipVisionConfig(hidden_size=72, num_hidden_layers=3, num_attention_heads=1, intermediate_size=136, image_size=42, patch_size=14,
                            hidden_act="gelu", layer_norm_eps=1e-6, attn_implementation="eager")
    sm = SiglipVisionModel(sc).eval()
    randomise(sm, g)
    with torch.no_grad():
        hs = sm(img, output_hidden_states=True).hidden_states
    assert len(hs) == 4
    y = hs[-2]
    sd = dict(sm.state_dict())
    root = "vision_model." if "vision_model.embeddings.patch_embedding.weight" in sd else ""   # (the prefix depends on the transformers release)
    w = {"patch_embed.proj.weight": sd[root + "embeddings.patch_embedding.weight"], "patch_embed.proj.bias": sd[root + "embeddings.patch_embedding.bias"],
         "pos_embed": sd[root + "embeddings.position_embedding.weight"][None]}
    for i in range(3):
        w.update({f"blocks.{i}.{k}": v for k, v in siglip_layer(sd, i, root).items()})
    out.update({"siglip__" + k: v.detach().numpy() for k, v in w.items()})
    out["siglip__x"], out["siglip__y"] = img.numpy(), y.numpy()
    np.savez_compressed(Path(__file__).resolve().parent / "g15_hf_towers.npz", **out)
    print({k: v.shape for k, v in out.items() if k.endswith(("__x", "__y"))})


if __name__ == "__main__":
    main()
