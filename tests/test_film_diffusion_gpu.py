"""SURVEY.md section 8 rows a2 (FiLM) and a11 (diffusion head + DDIM sampler), i.e. BASELINE.json config 5's ingredients,
against the CPU oracle on the reduced-size model.  Same comparison scheme as test_engine_gpu.py."""
import importlib

import numpy as np
import pytest
import torch

from oracle import vla_oracle as vo

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
load = importlib.import_module


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def grads_vs_oracle(eng, ocfg, sd, batch, **fw):
    names = set(eng.export_trainable("data"))
    out = {}
    for mode in ("fp32", "bf16"):
        sdg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
        loss, _, _ = vo.Oracle(ocfg, sdg, mode=mode).train_forward(batch, **fw)
        loss.backward()
        out[mode] = (loss.item(), {n: sdg[n].grad for n in names})
    return names, out


def compare_grads(names, grads, ref32, ref16):
    e_hip, e_emu = {}, {}
    for n in sorted(names):
        ref, emu, got = ref32[n], ref16[n], grads[n].float().cpu()
        assert ref is not None and got.shape == ref.shape and torch.isfinite(got).all(), n
        if ref.norm() < 1e-7:
            continue
        e_hip[n] = ((got - ref).norm() / ref.norm()).item()
        e_emu[n] = ((emu - ref).norm() / ref.norm()).item()
    w = sorted(e_hip, key=e_hip.get, reverse=True)[:4]
    print("worst rel-L2 (hip, bf16-emu):", [(n, f"{e_hip[n]:.3f}", f"{e_emu[n]:.3f}") for n in w])
    print(f"median hip {np.median(list(e_hip.values())):.4f} emu {np.median(list(e_emu.values())):.4f}")
    assert np.median(list(e_hip.values())) <= 1.5 * np.median(list(e_emu.values())) + 5e-3
    assert max(e_hip.values()) <= 2.0 * max(e_emu.values()) + 3e-2
    return e_hip


def make(dev, film=False, diffusion=False, seed=0):
    engine_mod, weights_mod, synth, config_mod = (load(f"openvla-oft_amd.{m}") for m in ("engine", "weights", "synthetic", "config"))
    ocfg = vo.tiny_config()
    sd = {k: v.to(BF).float() for k, v in vo.random_state_dict(ocfg, seed=seed, film=film, diffusion=diffusion).items()}
    cfg = config_mod.VLAConfig.from_any(ocfg)
    get, has = weights_mod.make_getter(sd, dev)
    eng = engine_mod.VLAEngine(cfg, get, dev, lora=True, use_proprio=True, head="diffusion" if diffusion else "l1", use_film=film, has=has)
    batch = synth.make_batch(3, seed=2, prompt_lens=[10, 8, 12], image_size=56)
    for k in ("pixel_values", "proprio", "actions"):
        batch[k] = batch[k].to(BF).float()
    return eng, ocfg, cfg, sd, batch


def test_film_forward_and_backward(dev):
    eng, ocfg, cfg, sd, batch = make(dev, film=True)
    with torch.no_grad():
        h16, P = vo.Oracle(ocfg, sd, mode="bf16").multimodal_hidden(batch["input_ids"], batch["attention_mask"], batch["pixel_values"],
                                                                     batch["labels"], batch["proprio"], use_film=True)
        h_nofilm, _ = vo.Oracle(ocfg, sd, mode="bf16").multimodal_hidden(batch["input_ids"], batch["attention_mask"], batch["pixel_values"],
                                                                         batch["labels"], batch["proprio"], use_film=False)
    out = eng.forward(batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["labels"], proprio=batch["proprio"])
    valid = torch.cat([torch.ones(3, 1 + P, dtype=torch.bool), batch["attention_mask"][:, 1:]], 1)
    e = rel(out["hidden"].float().cpu()[valid], h16[valid])
    print(f"FiLM hidden: hip vs bf16-emu {e:.3e}; effect of FiLM itself {rel(h_nofilm[valid], h16[valid]):.3e}")
    assert e < 3e-2 and rel(h_nofilm[valid], h16[valid]) > 3 * e, "FiLM must matter and must match"
    # gradients (sign-flip-free targets, see test_engine_gpu.py)
    with torch.no_grad():
        _, p0, _ = vo.Oracle(ocfg, sd).train_forward(batch, use_film=True)
    g = torch.Generator().manual_seed(3)
    batch["actions"] = (p0 + (0.25 + 0.5 * torch.rand(p0.shape, generator=g)) * torch.where(torch.rand(p0.shape, generator=g) < 0.5, -1.0, 1.0)).to(BF).float()
    names, ref = grads_vs_oracle(eng, ocfg, sd, batch, use_film=True)
    assert any(".scale.weight" in n for n in names) and any(".shift.bias" in n for n in names)
    eng.zero_grad()
    loss_sum, count, _ = eng.train_step_fwd_bwd(batch)
    assert abs(loss_sum.item() / count - ref["fp32"][0]) < 2e-2 * max(1.0, abs(ref["fp32"][0]))
    e_hip = compare_grads(names, eng.export_trainable("grad"), ref["fp32"][1], ref["bf16"][1])
    assert all(n in e_hip for n in names if ".scale.weight" in n or ".shift.weight" in n)


def test_diffusion_training_step(dev):
    eng, ocfg, cfg, sd, batch = make(dev, diffusion=True)
    g = torch.Generator().manual_seed(4)
    noise = torch.randn(3, 8, 7, generator=g).to(BF).float()
    timesteps = torch.tensor([3, 41, 17])
    ddim = vo.DDIM(50)
    names, ref = grads_vs_oracle(eng, ocfg, sd, batch, use_diffusion=True, noise=noise, timesteps=timesteps, ddim=ddim)
    assert any(n.startswith("noisy_action_projector.") for n in names)
    dmod = load("openvla-oft_amd.diffusion")
    sched, enc = dmod.DDIMScheduler(50), dmod.SinusoidalPositionalEncoding(cfg.llm_dim)
    noisy = sched.add_noise(batch["actions"], noise, timesteps).to(BF)
    temb = enc(timesteps.float()).to(BF)
    eng.zero_grad()
    loss_sum, count, pred = eng.train_step_fwd_bwd(batch, diffusion=dict(noise=noise, noisy_actions=noisy, timestep_emb=temb))
    loss = loss_sum.item() / count
    print(f"diffusion MSE loss hip {loss:.5f} oracle fp32 {ref['fp32'][0]:.5f} bf16-emu {ref['bf16'][0]:.5f}")
    assert abs(loss - ref["bf16"][0]) < 3e-2 * max(1.0, abs(ref["bf16"][0]))
    compare_grads(names, eng.export_trainable("grad"), ref["fp32"][1], ref["bf16"][1])


def test_ddim_sampling_matches_oracle(dev):
    modeling, config_mod = load("openvla-oft_amd.modeling"), load("openvla-oft_amd.config")
    ocfg = vo.tiny_config()
    sd = {k: v.to(BF).float() for k, v in vo.random_state_dict(ocfg, seed=1, diffusion=True).items()}
    cfg = config_mod.VLAConfig.from_any(ocfg)
    stats = {"d": {"action": {"q01": [-1.0] * 7, "q99": [1.0] * 7, "mask": [True] * 6 + [False]}}}
    vla = modeling.OpenVLAForActionPrediction(cfg, sd, device=dev, norm_stats=stats)
    sub = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}  # noqa: E731
    head = modeling.DiffusionActionHead(cfg.llm_dim, cfg.llm_dim, 7, num_diffusion_steps=5, device=dev, state_dict=sub("action_head."))
    pp = modeling.ProprioProjector(cfg.llm_dim, 8, device=dev, state_dict=sub("proprio_projector."))
    nap = modeling.NoisyActionProjector(cfg.llm_dim, device=dev, state_dict=sub("noisy_action_projector."))
    g = torch.Generator().manual_seed(9)
    ids = torch.cat([torch.tensor([[1]]), torch.randint(3, 31000, (1, 9), generator=g)], 1)
    mask = torch.ones_like(ids, dtype=torch.bool)
    pv = torch.randn(1, 12, 56, 56, generator=g).to(BF).float()
    proprio = (torch.rand(8, generator=g) * 2 - 1).to(BF).float().numpy()
    x_T = torch.randn(1, 8, 7, generator=g).to(BF).float()
    ref, _ = vo.Oracle(ocfg, sd, mode="bf16").predict_action(ids, mask, pv, proprio=proprio, unnorm_stats=stats["d"]["action"], head="diffusion",
                                                              noise=x_T, num_diffusion_steps=5)
    act, hid = vla.predict_action(input_ids=ids, unnorm_key="d", proprio=proprio, proprio_projector=pp, action_head=head,
                                  noisy_action_projector=nap, pixel_values=pv, attention_mask=mask, noise=x_T)
    err = np.abs(act - ref).max()
    print(f"5-step DDIM sample Linf vs bf16-emulating oracle: {err:.3e}")
    assert act.shape == (8, 7) and hid.shape == (1, 56, cfg.llm_dim) and err < 8e-2
    d = head.sample_noisy_actions(torch.rand(2, 8, 7) * 2 - 1, generator=torch.Generator().manual_seed(0))
    assert d["noisy_actions"].shape == (2, 8, 7) and d["diffusion_timestep_embeddings"].shape == (2, 1, cfg.llm_dim)
