"""Generates the fixtures of SURVEY.md section 8c whose source of truth is NOT a reference module (those are in make_golden.py):

  G6  full restated forward, tiny dims (assemble -> LLM -> shift-by-one gather -> L1 head + loss)     <- the oracle itself (fp32)
  G7  the two ViT block flavours (LayerScale + 5 prefix tokens + head_dim 64 | head_dim 72, GELU)      <- the oracle itself (fp32)
  G8  peft-style LoRA linear, forward + all gradients                                                  <- plain torch autograd
  G9  one torch.optim.AdamW step on bf16 parameters with bf16 gradients (finetune.py:952 defaults)     <- torch
  G11 config-1 plumbing: synthetic observation -> center crop -> 6*I-channel tensor -> tiny model -> actions  <- the oracle

G6 / G7 / G11 pin the oracle against ITSELF at the time of writing (a regression net for the checker: an edit of the oracle that
changes its numbers must be deliberate); they are cross-checked piecewise by G1-G5 / G10, which come from the reference's own modules.
The weights are the seeded `random_state_dict` of the tiny config; a checksum of them is stored next to the outputs.

    python tests/golden/make_golden_own.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import vla_oracle as vo  # noqa: E402

OUT = Path(__file__).resolve().parent


def tiny_batch(seed=21, B=2, prompt_lens=(9, 7)):
    """Collator-shaped batch (right padded), 56 x 56 images, built with numpy only (same layout as G10)."""
    rng = np.random.default_rng(seed)
    A, L = 56, max(prompt_lens) + 56 + 1
    ids = np.full((B, L), 32000, np.int64); labels = np.full((B, L), -100, np.int64); mask = np.zeros((B, L), bool)
    actions = rng.uniform(-1, 1, (B, 8, 7)).astype(np.float32)
    for b, tp in enumerate(prompt_lens):
        prompt = np.concatenate([[1], rng.integers(3, 31000, tp - 2), [29871]])
        tok = vo.tokenize_actions(actions[b]).reshape(-1)
        row = np.concatenate([prompt, tok, [2]])
        ids[b, : len(row)] = row; mask[b, : len(row)] = True
        labels[b, tp: len(row)] = row[tp:]
    pv = rng.standard_normal((B, 12, 56, 56)).astype(np.float32)
    proprio = rng.uniform(-1, 1, (B, 8)).astype(np.float32)
    return dict(input_ids=ids, attention_mask=mask, labels=labels, pixel_values=pv, proprio=proprio, actions=actions)


def sd_checksum(sd):
    return np.float64(sum(float(v.double().abs().sum()) for v in sd.values()))


def main():
    torch.manual_seed(0)
    ocfg = vo.tiny_config()
    sd = vo.random_state_dict(ocfg, seed=0)
    o = vo.Oracle(ocfg, sd, mode="fp32")

    # ---- G6 -------------------------------------------------------------------------------------------------------
    b = tiny_batch()
    tb = {k: torch.from_numpy(v) for k, v in b.items()}
    with torch.no_grad():
        hidden, P = o.multimodal_hidden(tb["input_ids"], tb["attention_mask"], tb["pixel_values"], tb["labels"], tb["proprio"])
        loss, pred, ah = o.train_forward(tb)
    np.savez(OUT / "g6_full_forward.npz", **b, P=np.int64(P), hidden_valid_row0=hidden[0, : 1 + P + 9 + 57].numpy(),
             action_hidden=ah.numpy(), pred=pred.numpy(), loss=np.float64(loss.item()), sd_checksum=sd_checksum(sd))

    # ---- G7 -------------------------------------------------------------------------------------------------------
    rng = np.random.default_rng(3)
    img = rng.standard_normal((3, 3, 56, 56)).astype(np.float32)
    with torch.no_grad():
        dino = o.vit(torch.from_numpy(img), "vision_backbone.featurizer.", ocfg.dino)
        sig = o.vit(torch.from_numpy(img), "vision_backbone.fused_featurizer.", ocfg.siglip)
    np.savez(OUT / "g7_vit_blocks.npz", img=img, dino=dino.numpy(), siglip=sig.numpy(), sd_checksum=sd_checksum(sd))

    # ---- G8: plain torch, no oracle code ------------------------------------------------------------------------------
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 48, generator=g)
    W, bias = torch.randn(40, 48, generator=g) * 0.1, torch.randn(40, generator=g) * 0.1
    A_, B_ = torch.randn(8, 48, generator=g) * 0.2, torch.randn(40, 8, generator=g) * 0.2
    dy = torch.randn(6, 40, generator=g)
    xr, Ar, Br = x.clone().requires_grad_(True), A_.clone().requires_grad_(True), B_.clone().requires_grad_(True)
    scale = 16 / 32
    y = torch.nn.functional.linear(xr, W, bias) + torch.nn.functional.linear(torch.nn.functional.linear(xr, Ar), Br) * scale
    y.backward(dy)
    np.savez(OUT / "g8_lora_linear.npz", x=x.numpy(), W=W.numpy(), bias=bias.numpy(), A=A_.numpy(), B=B_.numpy(), dy=dy.numpy(), scale=np.float64(scale),
             y=y.detach().numpy(), dx=xr.grad.numpy(), dA=Ar.grad.numpy(), dB=Br.grad.numpy())

    # ---- G9: torch.optim.AdamW on bf16 params, 3 steps ------------------------------------------------------------------
    p0 = (torch.randn(4096, generator=g) * 0.05).to(torch.bfloat16)
    grads = [(torch.randn(4096, generator=g) * 0.01).to(torch.bfloat16) for _ in range(3)]
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=5e-4)
    after = []
    for gr in grads:
        p.grad = gr.clone()
        opt.step()
        after.append(p.detach().float().numpy().copy())
    np.savez(OUT / "g9_adamw_bf16.npz", p0=p0.float().numpy(), grads=np.stack([x_.float().numpy() for x_ in grads]), after=np.stack(after),
             lr=np.float64(5e-4), betas=np.array([0.9, 0.999]), eps=np.float64(1e-8), weight_decay=np.float64(0.01))

    # ---- G11 ------------------------------------------------------------------------------------------------------------
    # same shapes / dtypes as experiments/robot/libero/sample_libero_spatial_observation.pkl (a pickle: not loaded); the tiny towers
    # take 56 x 56 inputs, so the 224 crop is subsampled 4x after the crop
    rng = np.random.default_rng(11)
    obs_full = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8); obs_wrist = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)
    state = rng.uniform(-1, 1, 8)
    stats = {"q01": [-1.0] * 7, "q99": [1.0] * 7, "mask": [True] * 6 + [False]}
    pstats = {"q01": [-2.0] * 8, "q99": [2.0] * 8}
    crops = [vo.crop_and_resize_center(im) for im in (obs_full, obs_wrist)]
    pv = torch.cat([vo.image_transform(c, (vo.IMAGENET_MEAN, vo.SIGLIP_MEAN), (vo.IMAGENET_STD, vo.SIGLIP_STD)) for c in crops], 0)[None]
    ids = torch.tensor([[1] + rng.integers(3, 31743, 36).tolist() + [29871]])
    prop = vo.normalize_proprio(state, pstats, ocfg.norm_type)
    actions, _ = o.predict_action(ids, torch.ones_like(ids, dtype=torch.bool), pv[:, :, ::4, ::4].contiguous(), proprio=prop, unnorm_stats=stats)
    np.savez(OUT / "g11_config1_plumbing.npz", full_image=obs_full, wrist_image=obs_wrist, state=state, input_ids=ids.numpy(),
             crop_full=crops[0], crop_wrist=crops[1], pixel_values_sum=np.float64(pv.double().sum().item()),
             pixel_values_probe=pv[0, :, 100, 50:54].numpy(), proprio_normalized=prop, actions=np.asarray(actions), sd_checksum=sd_checksum(sd))
    print("wrote", sorted(p_.name for p_ in OUT.glob("g*.npz")))


if __name__ == "__main__":
    main()
