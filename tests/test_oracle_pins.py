"""Pins the CPU oracle (oracle/vla_oracle.py) against the golden vectors produced by the reference's own modules and by
stock transformers (tests/golden/make_golden.py).  Runs on CPU; reads only the committed .npz fixtures."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import vla_oracle as vo

G = Path(__file__).resolve().parent / "golden"


def load(name):
    return dict(np.load(G / name, allow_pickle=False))


def sd_from(npz, prefixes):
    return {k: torch.from_numpy(v) for k, v in npz.items() if k.startswith(prefixes)}


def test_g1_action_masks_match_reference():
    g = load("g1_masks.npz")
    lab = torch.from_numpy(g["labels"])
    assert np.array_equal(vo.current_action_mask(lab, 7).numpy(), g["current"])
    assert np.array_equal(vo.next_actions_mask(lab, 7).numpy(), g["next"])
    assert np.array_equal(vo.current_action_mask(lab[:, 1:], 7).numpy(), g["current_shift"])
    assert np.array_equal(vo.next_actions_mask(lab[:, 1:], 7).numpy(), g["next_shift"])
    m = vo.all_actions_mask(lab, 7).numpy()
    assert (m.sum(1) == 56).all(), "exactly ACTION_DIM*NUM_ACTIONS_CHUNK action slots; the stop token is excluded"


def test_g2_l1_head_forward_and_backward_match_reference():
    g = load("g2_l1_head.npz")
    cfg = vo.tiny_config(llm_dim=64)
    sd = {k: v.requires_grad_(True) for k, v in sd_from(g, "action_head.").items()}
    o = vo.Oracle(cfg, sd)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    pred = o.l1_head(x)
    assert np.allclose(pred.detach().numpy(), g["pred"], atol=2e-6, rtol=1e-5)
    loss = (torch.from_numpy(g["gt"]) - pred).abs().mean()
    assert np.allclose(loss.item(), g["loss"], atol=1e-7)
    loss.backward()
    assert np.allclose(x.grad.numpy(), g["dx"], atol=1e-7, rtol=1e-4)
    for k, v in sd.items():
        ref = g["grad." + k[len("action_head."):]]
        assert np.allclose(v.grad.numpy(), ref, atol=1e-6, rtol=1e-4), k


def test_g3_projectors_and_time_encoder_match_reference():
    g = load("g3_projectors.npz")
    o = vo.Oracle(vo.tiny_config(llm_dim=64), sd_from(g, ("proprio_projector.", "noisy_action_projector.")))
    assert np.allclose(o.mlp_projector(torch.from_numpy(g["proprio"]), "proprio_projector.").numpy(), g["proprio_out"], atol=2e-6)
    assert np.allclose(o.mlp_projector(torch.from_numpy(g["noisy"]), "noisy_action_projector.").numpy(), g["noisy_out"], atol=2e-6)
    assert np.allclose(vo.sinusoidal_encoding(torch.from_numpy(g["timesteps"]), 64).numpy(), g["time_emb"], atol=1e-6)


def test_g4_action_tokenizer_matches_reference():
    g = load("g4_action_tokenizer.npz")
    assert int(g["begin_idx"]) == vo.ACTION_TOKEN_BEGIN_IDX
    assert np.array_equal(vo.tokenize_actions(g["actions"]), g["token_ids"])
    assert np.array_equal(vo.decode_token_ids_to_actions(g["all_ids"]), g["decoded"])
    # encode -> decode stays within one bin width
    a = np.clip(g["actions"], -1, 1)
    assert np.abs(vo.decode_token_ids_to_actions(vo.tokenize_actions(a)) - a).max() <= 2.0 / 255 + 1e-12


@pytest.mark.parametrize("mode", ["bidirectional", "causal"])
def test_g5_llama_stack_matches_stock_transformers(mode):
    g = load("g5_llama.npz")
    cfg = vo.tiny_config(llm_dim=128, llm_layers=2, llm_heads=2, llm_ff=256, vocab=320)
    o = vo.Oracle(cfg, sd_from(g, "language_model."), mask_mode=mode)
    mask = torch.from_numpy(g["mask"])
    h = o.llm(torch.from_numpy(g["embeds"]), mask).numpy()
    ref = g["hidden_" + mode]
    valid = g["mask"]
    err = np.abs(h - ref)[valid].max()
    assert err < 2e-5, f"{mode}: max err on non-pad positions {err}"
    if mode == "bidirectional":
        logits = o.lm_logits(torch.from_numpy(h)).numpy()
        assert np.abs(logits - g["logits_bidirectional"])[valid].max() < 1e-4
        assert np.array_equal(logits.argmax(-1)[valid], g["logits_bidirectional"].argmax(-1)[valid])


def test_bidirectional_and_causal_fixtures_are_distinct():
    g = load("g5_llama.npz")
    # position 0 sees only itself under the causal mask but the whole row under the bidirectional one
    assert np.abs(g["hidden_bidirectional"][0, 0] - g["hidden_causal"][0, 0]).max() > 1e-3


def test_unnormalize_and_proprio_normalisation():
    stats = {"q01": [-1.0, -0.5, 0, 0, 0, 0, 0], "q99": [1.0, 0.5, 2, 2, 2, 2, 1], "mask": [True] * 6 + [False],
             "min": [-2.0] * 7, "max": [2.0] * 7}
    a = np.linspace(-1, 1, 7)[None].repeat(2, 0)
    out = vo.unnormalize_actions(a, stats, "bounds_q99")
    assert np.allclose(out[0, 0], 0.5 * (a[0, 0] + 1) * (2.0 + 1e-8) - 1.0) and out[0, 6] == a[0, 6]
    out_b = vo.unnormalize_actions(a, stats, "bounds")
    assert np.allclose(out_b[0, 1], 0.5 * (a[0, 1] + 1) * (4.0 + 1e-8) - 2.0)
    p = np.array([5.0, 0.0, 1.0, 1.0, 1.0, 1.0, 0.3])
    n = vo.normalize_proprio(p, stats, "bounds_q99")
    assert n[0] == 1.0 and np.isclose(n[1], 0.0) and n[6] == 0.3 and (np.abs(n) <= 1).all()


def test_ddim_schedule_properties():
    d = vo.DDIM(50)
    assert d.alphas_cumprod.shape == (50,) and (d.alphas_cumprod[1:] < d.alphas_cumprod[:-1]).all()
    assert abs(d.alphas_cumprod[0].item() - (np.cos(0.008 / 1.008 * np.pi / 2 + (1 / 50) / 1.008 * np.pi / 2) ** 2 / np.cos(0.008 / 1.008 * np.pi / 2) ** 2)) < 1e-6
    d.set_timesteps(50)
    assert d.timesteps.tolist() == list(range(49, -1, -1))
    x0 = torch.rand(2, 8, 7) * 2 - 1
    eps = torch.randn(2, 8, 7)
    t = torch.tensor([10, 40])
    xt = d.add_noise(x0, eps, t)
    # a perfect epsilon prediction at t maps x_t back onto the clean-sample trajectory (eta = 0, deterministic)
    prev = d.step(eps[:1], 10, xt[:1])
    assert torch.allclose(prev, d.alphas_cumprod[9] ** 0.5 * x0[:1] + (1 - d.alphas_cumprod[9]) ** 0.5 * eps[:1], atol=1e-5)


def test_g10_collator_layout():
    """The synthetic batch generator of the product reproduces the reference collator's layout."""
    import importlib

    g = load("g10_collator.npz")
    synth = importlib.import_module("openvla-oft_amd.synthetic")
    inst = []
    for i in range(int(g["n"])):
        inst.append({k: g[f"inst{i}.{k}"] for k in ("input_ids", "labels", "pixel_values", "pixel_values_wrist", "actions", "proprio")})
    batch = synth.collate(inst, pad_token_id=32000)
    for k in ("input_ids", "labels", "attention_mask", "pixel_values", "actions", "proprio"):
        assert np.array_equal(batch[k].numpy(), g["batch." + k]), k
