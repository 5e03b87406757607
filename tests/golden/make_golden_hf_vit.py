"""Generates tests/golden/g14_hf_vit_blocks.npz: one DINOv2 block (LayerScale) and one SigLIP block evaluated by the HuggingFace
`transformers` implementations of those architectures (Dinov2Layer / SiglipEncoderLayer, stock transformers in the build container), on
seeded weights and inputs, in fp32.  timm 0.9.10 -- what the reference calls (modeling_prismatic.py:127-139) -- is not installable here; these
are independent implementations of the same published blocks and pin oracle.vla_oracle.Oracle.vit_block (which G7 only pins against itself).
Weights are stored under the oracle's (timm-style) names: the q / k / v projections concatenated into `attn.qkv`, LayerScale as `ls{1,2}.scale_factor`.

  python tests/golden/make_golden_hf_vit.py
"""
from pathlib import Path

import numpy as np
import torch
from transformers.models.dinov2.configuration_dinov2 import Dinov2Config
from transformers.models.dinov2.modeling_dinov2 import Dinov2Layer
from transformers.models.siglip.configuration_siglip import SiglipVisionConfig
from transformers.models.siglip.modeling_siglip import SiglipEncoderLayer


def randomise(mod, g):
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if "lambda" in n:
                p.copy_(0.1 + 0.02 * torch.randn(p.shape, generator=g))
            elif n.endswith("norm1.weight") or n.endswith("norm2.weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() == 2 else 0.03))


def main():
    g = torch.Generator().manual_seed(14)
    out = {}
    # DINOv2-style block: head_dim 64, LayerScale, exact GELU, eps 1e-6, 5 prefix + 16 patch tokens
    dc = Dinov2Config(hidden_size=128, num_attention_heads=2, mlp_ratio=2, layerscale_value=0.1, hidden_act="gelu", layer_norm_eps=1e-6, use_swiglu_ffn=False,
                      attn_implementation="eager")
    dl = Dinov2Layer(dc).eval()
    randomise(dl, g)
    x = torch.randn(2, 21, 128, generator=g)
    with torch.no_grad():
        y = dl(x)
    y = y[0] if isinstance(y, tuple) else y
    sd = dict(dl.named_parameters())
    a = sd["attention.attention.query.weight"], sd["attention.attention.key.weight"], sd["attention.attention.value.weight"]
    b = sd["attention.attention.query.bias"], sd["attention.attention.key.bias"], sd["attention.attention.value.bias"]
    w = {"norm1.weight": sd["norm1.weight"], "norm1.bias": sd["norm1.bias"], "attn.qkv.weight": torch.cat(a, 0), "attn.qkv.bias": torch.cat(b, 0),
         "attn.proj.weight": sd["attention.output.dense.weight"], "attn.proj.bias": sd["attention.output.dense.bias"], "ls1.scale_factor": sd["layer_scale1.lambda1"],
         "norm2.weight": sd["norm2.weight"], "norm2.bias": sd["norm2.bias"], "mlp.fc1.weight": sd["mlp.fc1.weight"], "mlp.fc1.bias": sd["mlp.fc1.bias"],
         "mlp.fc2.weight": sd["mlp.fc2.weight"], "mlp.fc2.bias": sd["mlp.fc2.bias"], "ls2.scale_factor": sd["layer_scale2.lambda1"]}
    out.update({"dino__" + k: v.detach().numpy() for k, v in w.items()})
    out["dino__x"], out["dino__y"] = x.numpy(), y.numpy()
    # SigLIP-style block: head_dim 72, no LayerScale, odd MLP width, exact GELU (timm 0.9.10's vit_so400m_patch14_siglip_224 default) and the tanh form
    for act, tag in (("gelu", "siglip"), ("gelu_pytorch_tanh", "siglip_tanh")):
        sc = SiglipVisionConfig(hidden_size=144, num_attention_heads=2, intermediate_size=536, hidden_act=act, layer_norm_eps=1e-6, attn_implementation="eager")
        sl = SiglipEncoderLayer(sc).eval()
        randomise(sl, g)
        x = torch.randn(2, 16, 144, generator=g)
        with torch.no_grad():
            y = sl(x, attention_mask=None)
        y = y[0] if isinstance(y, tuple) else y
        sd = dict(sl.named_parameters())
        w = {"norm1.weight": sd["layer_norm1.weight"], "norm1.bias": sd["layer_norm1.bias"],
             "attn.qkv.weight": torch.cat([sd[f"self_attn.{n}_proj.weight"] for n in "qkv"], 0), "attn.qkv.bias": torch.cat([sd[f"self_attn.{n}_proj.bias"] for n in "qkv"], 0),
             "attn.proj.weight": sd["self_attn.out_proj.weight"], "attn.proj.bias": sd["self_attn.out_proj.bias"],
             "norm2.weight": sd["layer_norm2.weight"], "norm2.bias": sd["layer_norm2.bias"], "mlp.fc1.weight": sd["mlp.fc1.weight"], "mlp.fc1.bias": sd["mlp.fc1.bias"],
             "mlp.fc2.weight": sd["mlp.fc2.weight"], "mlp.fc2.bias": sd["mlp.fc2.bias"]}
        out.update({tag + "__" + k: v.detach().numpy() for k, v in w.items()})
        out[tag + "__x"], out[tag + "__y"] = x.numpy(), y.numpy()
    np.savez_compressed(Path(__file__).resolve().parent / "g14_hf_vit_blocks.npz", **out)
    print({k: v.shape for k, v in out.items() if k.endswith(("__x", "__y"))})


if __name__ == "__main__":
    main()
