"""End-to-end parity of the HIP engine against the CPU oracle on a reduced-size model with the same structural quirks
(head_dim 128 / 64 / 72, 5 ViT prefix tokens, LayerScale, non-multiple-of-64 MLP width, fused q|k|v and gate|up LoRA,
ragged right-padded prompts, 2 images + proprio, L1 head).  Same seeded inputs on both sides.

Tolerances (stated per check): the oracle runs in fp32 ("exact") or with bf16 re-rounding at the reference's rounding
points ("bf16 emulation"); the HIP path computes in bf16 with fp32 accumulation, so it is compared
  * to the bf16 emulation with a tight bound (same rounding points; differences = accumulation order + fusions), and
  * to the fp32 oracle with the bound the bf16 emulation itself achieves against fp32 (x2 slack).
"""
import importlib

import numpy as np
import pytest
import torch

from oracle import vla_oracle as vo

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.fixture(scope="module")
def setup(dev):
    load = importlib.import_module
    engine_mod, weights_mod, synth = load("openvla-oft_amd.engine"), load("openvla-oft_amd.weights"), load("openvla-oft_amd.synthetic")
    config_mod = load("openvla-oft_amd.config")
    ocfg = vo.tiny_config()
    sd = {k: v.to(BF).float() for k, v in vo.random_state_dict(ocfg, seed=0).items()}   # both sides see bf16-exact weights
    cfg = config_mod.VLAConfig.from_any(ocfg)
    get, has = weights_mod.make_getter(sd, dev)
    eng = engine_mod.VLAEngine(cfg, get, dev, lora=True, use_proprio=True, head="l1", has=has)
    batch = synth.make_batch(3, seed=1, prompt_lens=[10, 8, 12], image_size=56)
    batch["pixel_values"] = batch["pixel_values"].to(BF).float()
    batch["proprio"] = batch["proprio"].to(BF).float()
    batch["actions"] = batch["actions"].to(BF).float()
    return dict(eng=eng, cfg=cfg, ocfg=ocfg, sd=sd, batch=batch)


def test_forward_matches_oracle(setup):
    eng, ocfg, sd, batch = setup["eng"], setup["ocfg"], setup["sd"], setup["batch"]
    o32 = vo.Oracle(ocfg, sd, mode="fp32")
    o16 = vo.Oracle(ocfg, sd, mode="bf16")
    with torch.no_grad():
        h32, P = o32.multimodal_hidden(batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["labels"], batch["proprio"])
        h16, _ = o16.multimodal_hidden(batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["labels"], batch["proprio"])
        l32, p32, a32 = o32.train_forward(batch)
        l16, p16, a16 = o16.train_forward(batch)
    out = eng.forward(batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["labels"], proprio=batch["proprio"], train=False)
    assert out["P"] == P
    hid = out["hidden"].float().cpu()
    valid = torch.cat([torch.ones(3, 1 + P, dtype=torch.bool), batch["attention_mask"][:, 1:]], 1)   # pad rows are don't-care
    emu_vs_exact = rel(h16[valid], h32[valid])
    e16, e32 = rel(hid[valid], h16[valid]), rel(hid[valid], h32[valid])
    print(f"hidden: hip vs bf16-emu {e16:.3e}, hip vs fp32 {e32:.3e}, bf16-emu vs fp32 {emu_vs_exact:.3e}")
    assert e16 < 3e-2, "hidden states vs bf16-emulating oracle"
    assert e32 < max(2 * emu_vs_exact, 3e-2), "hidden states vs fp32 oracle"
    ah, _ = eng.gather_action_hidden(out["hidden"], out["action_rows"])
    assert rel(ah.view(3, 56, -1), a16) < 3e-2, "gathered action hidden states (shift-by-one gather)"
    tgt = batch["actions"].to(eng.device, BF).reshape(-1, 7).contiguous()
    pred, loss_sum, _ = eng.head.fwd(ah, target=tgt)
    pe16, pe32 = (pred.float().cpu().view(3, 8, 7) - p16).abs().max().item(), (pred.float().cpu().view(3, 8, 7) - p32).abs().max().item()
    print(f"pred Linf: hip vs bf16-emu {pe16:.3e}, hip vs fp32 {pe32:.3e}, bf16-emu vs fp32 {(p16 - p32).abs().max().item():.3e}")
    assert pe16 < 5e-2 and pe32 < max(2 * (p16 - p32).abs().max().item(), 5e-2)
    loss = loss_sum.item() / pred.numel()
    assert abs(loss - l16.item()) < 2e-2 * max(1.0, abs(l16.item())), f"L1 loss {loss} vs oracle {l16.item()}"


def test_backward_matches_oracle_autograd(setup):
    eng, ocfg, sd, batch = setup["eng"], setup["ocfg"], setup["sd"], dict(setup["batch"])
    names = set(eng.export_trainable("data"))
    sdg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    assert all(n in sdg for n in names), sorted(n for n in names if n not in sdg)[:5]
    o = vo.Oracle(ocfg, sdg, mode="fp32")
    # d|x|/dx = sign(x) is discontinuous: keep every residual at least 0.25 away from zero so that bf16-level
    # differences in the prediction cannot flip a sign and the comparison measures the backward arithmetic only.
    # (the action values only enter the forward through the loss: their token embeddings are zeroed, :620-621)
    with torch.no_grad():
        _, p0, _ = o.train_forward(batch)
    g = torch.Generator().manual_seed(7)
    off = (0.25 + 0.5 * torch.rand(p0.shape, generator=g)) * torch.where(torch.rand(p0.shape, generator=g) < 0.5, -1.0, 1.0)
    batch["actions"] = (p0 + off).to(BF).float()
    loss, _, _ = o.train_forward(batch)
    loss.backward()
    # second opinion: the oracle with bf16 re-rounding at the reference's rounding points (straight-through gradients).
    # ReLU gates and |.| signs are discontinuous, so ANY bf16 evaluation flips a few of them relative to fp32; the HIP
    # path is required to be as close to fp32 as this emulation of the reference's own bf16 arithmetic is.
    sde = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    le, _, _ = vo.Oracle(ocfg, sde, mode="bf16").train_forward(batch)
    le.backward()
    eng.zero_grad()
    loss_sum, count, _ = eng.train_step_fwd_bwd(batch)
    assert abs(loss_sum.item() / count - loss.item()) < 2e-2 * max(1.0, abs(loss.item()))
    grads = eng.export_trainable("grad")
    e_hip, e_emu, cos = {}, {}, {}
    for n in sorted(names):
        ref, emu = sdg[n].grad, sde[n].grad
        got = grads[n].float().cpu()
        assert got.shape == ref.shape and torch.isfinite(got).all(), n
        if ref.norm() < 1e-7:
            assert got.norm() < 1e-5, n
            continue
        e_hip[n] = ((got - ref).norm() / ref.norm()).item()
        e_emu[n] = ((emu - ref).norm() / ref.norm()).item()
        cos[n] = (torch.dot(got.flatten(), ref.flatten()) / (got.norm() * ref.norm())).item()
    w = sorted(e_hip, key=e_hip.get, reverse=True)[:5]
    print("worst rel-L2 (hip, bf16-emu):", [(n, f"{e_hip[n]:.3f}", f"{e_emu[n]:.3f}") for n in w])
    print(f"rel-L2 median hip {np.median(list(e_hip.values())):.4f} emu {np.median(list(e_emu.values())):.4f}; "
          f"max hip {max(e_hip.values()):.4f} emu {max(e_emu.values()):.4f}; cosine min {min(cos.values()):.5f}")
    assert np.median(list(e_hip.values())) <= 1.5 * np.median(list(e_emu.values())) + 5e-3
    assert max(e_hip.values()) <= 2.0 * max(e_emu.values()) + 2e-2
    assert min(cos.values()) > 0.98


def test_optimizer_step_moves_toward_lower_loss(setup):
    eng, batch = setup["eng"], setup["batch"]
    losses = []
    for _ in range(4):
        eng.zero_grad()
        loss_sum, count, _ = eng.train_step_fwd_bwd(batch)
        eng.adamw_step(lr=5e-4)
        eng.refresh_derived()
        losses.append(loss_sum.item() / count)
    print("losses", losses)
    assert losses[-1] < losses[0], losses


def test_discrete_objective_matches_oracle(setup, dev):
    """run_forward_pass without a continuous head (finetune.py:357-378): next-token cross entropy on the action + stop tokens
    through the frozen lm_head.  Loss and predicted ids against the oracle, every trainable gradient against the oracle's autograd
    (fp32) with the bf16-emulating oracle as the yardstick, as for the L1 objective above."""
    load = importlib.import_module
    engine_mod, weights_mod = load("openvla-oft_amd.engine"), load("openvla-oft_amd.weights")
    ocfg, sd, batch, cfg = setup["ocfg"], setup["sd"], setup["batch"], setup["cfg"]
    get, has = weights_mod.make_getter(sd, dev)
    eng = engine_mod.VLAEngine(cfg, get, dev, lora=True, use_proprio=True, head="none", has=has)
    names = set(eng.export_trainable("data"))
    assert not any(n.startswith("action_head.") for n in names)
    sdg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    sde = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    loss, ids32 = vo.Oracle(ocfg, sdg, mode="fp32").train_forward_discrete(batch)
    loss.backward()
    le, ids16 = vo.Oracle(ocfg, sde, mode="bf16").train_forward_discrete(batch)
    le.backward()
    eng.zero_grad()
    loss_sum, count, pred = eng.train_step_discrete(batch)
    counted = batch["labels"][:, 1:] != -100
    assert count == int(counted.sum()) == 3 * 57 and pred.shape == ids16.shape and bool((pred[~counted] == -1).all())
    got = loss_sum.item() / count
    print(f"CE loss: hip {got:.5f}, bf16-emu {le.item():.5f}, fp32 {loss.item():.5f}")
    assert abs(got - le.item()) < 2e-2 * max(1.0, abs(le.item())) and abs(got - loss.item()) < 3e-2 * max(1.0, abs(loss.item()))
    agree16, emu_agree = (pred[counted] == ids16[counted]).float().mean().item(), (ids16[counted] == ids32[counted]).float().mean().item()
    print(f"predicted ids equal to the bf16-emulating oracle's: {agree16:.3f} (that oracle vs fp32: {emu_agree:.3f})")
    assert agree16 >= min(0.9, emu_agree - 0.05)
    grads = eng.export_trainable("grad")
    e_hip, e_emu, cos = {}, {}, {}
    for n in sorted(names):
        ref, emu, g = sdg[n].grad, sde[n].grad, grads[n].float().cpu()
        assert g.shape == ref.shape and torch.isfinite(g).all(), n
        if ref.norm() < 1e-7:
            assert g.norm() < 1e-5, n
            continue
        e_hip[n] = ((g - ref).norm() / ref.norm()).item()
        e_emu[n] = ((emu - ref).norm() / ref.norm()).item()
        cos[n] = (torch.dot(g.flatten(), ref.flatten()) / (g.norm() * ref.norm())).item()
    print(f"rel-L2 median hip {np.median(list(e_hip.values())):.4f} emu {np.median(list(e_emu.values())):.4f}; "
          f"max hip {max(e_hip.values()):.4f} emu {max(e_emu.values()):.4f}; cosine min {min(cos.values()):.5f}")
    assert np.median(list(e_hip.values())) <= 1.5 * np.median(list(e_emu.values())) + 5e-3
    assert max(e_hip.values()) <= 2.0 * max(e_emu.values()) + 2e-2
    assert min(cos.values()) > 0.98


def test_last_layer_on_selected_rows_equals_the_full_forward(setup):
    """forward(sel="actions") runs the last decoder layer's attention-output projection, MLP and the final norm on the action rows
    only (everything else of hidden_states[-1] is discarded by run_forward_pass / predict_action): same numbers as gathering those
    rows from the full forward (different GEMM tile schedules at M = 168 vs 3 x S: equal to accumulation order), and the backward
    through it gives the same gradients."""
    eng, batch = setup["eng"], setup["batch"]
    args = (batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["labels"])
    full = eng.forward(*args, proprio=batch["proprio"], train=True)
    ah_full, idx = eng.gather_action_hidden(full["hidden"], full["action_rows"])
    sel = eng.forward(*args, proprio=batch["proprio"], train=True, sel="actions")
    assert sel["hidden"] is None and sel["action_hidden"].shape == ah_full.shape and torch.equal(sel["sel_rows"], idx)
    assert rel(sel["action_hidden"], ah_full) < 1e-2
    g = torch.Generator(device="cpu").manual_seed(3)
    dah = (torch.randn(ah_full.shape, generator=g) * 0.05).to(BF).to(eng.device)
    eng.zero_grad()
    dh = torch.zeros((full["hidden"].shape[0] * full["hidden"].shape[1], ah_full.shape[1]), dtype=BF, device=eng.device)
    importlib.import_module("openvla-oft_amd.ops").gather_rows(dah, idx, ah_full.shape[1], dst=dh, scatter_add=True)
    eng.backward_from_hidden(dh, full["saved"])
    g_full = {k: v.float().clone() for k, v in eng.export_trainable("grad").items()}
    eng.zero_grad()
    eng.backward_from_hidden(dah, sel["saved"])
    g_sel = eng.export_trainable("grad")
    worst = max((((g_sel[k].float() - g_full[k]).norm() / (g_full[k].norm() + 1e-12)).item(), k) for k in g_full if g_full[k].norm() > 1e-7)
    print(f"selected-rows backward vs full backward: worst rel-L2 {worst[0]:.3e} ({worst[1]})")
    assert worst[0] < 2e-2


def test_async_pinned_upload_equals_the_synchronous_one(dev):
    """VLAEngine.forward uploads ids | labels | lengths as ONE non-blocking copy from pinned memory (the host no longer re-joins the GPU at every step):
    back-to-back forwards on DIFFERENT batches, each behind a long-running kernel queue, must give exactly what the synchronous pageable copies give
    (the pinned staging block of call k must survive until its copy has run, although call k + 1 has long been enqueued)."""
    import importlib

    load = importlib.import_module
    engine_mod, weights_mod, config_mod, synth = load("openvla-oft_amd.engine"), load("openvla-oft_amd.weights"), load("openvla-oft_amd.config"), load("openvla-oft_amd.synthetic")
    ocfg = vo.tiny_config()
    cfg = config_mod.VLAConfig.from_any(ocfg)
    get, has = weights_mod.make_getter(vo.random_state_dict(ocfg, seed=0), dev)
    eng = engine_mod.VLAEngine(cfg, get, dev, lora=True, use_proprio=True, head="l1", has=has)
    batches = [synth.make_batch(2, seed=50 + i, prompt_lens=[9, 7 + (i % 3)], image_size=56) for i in range(6)]
    busy = torch.randn(4096, 4096, device=dev)

    def run(async_on):
        engine_mod._ASYNC_UPLOAD = async_on
        outs = []
        for b in batches:
            for _ in range(20):
                busy @ busy                                  # a queue of work ahead of the upload, so the copy executes long after forward() returned
            o = eng.forward(b["input_ids"], b["attention_mask"], b["pixel_values"].to(dev, torch.bfloat16), b["labels"], proprio=b["proprio"].to(dev, torch.bfloat16), train=False)
            outs.append((o["hidden"], o["action_rows"]))
        torch.cuda.synchronize()
        return [(h.clone(), r.clone()) for h, r in outs]

    try:
        a, s = run(True), run(False)
    finally:
        engine_mod._ASYNC_UPLOAD = True
    for (ha, ra), (hs, rs) in zip(a, s):
        assert torch.equal(ra, rs) and torch.equal(ha, hs)
