"""Drop-in boundary, round-2 additions (VERDICT r01 "missing" 4 / 9):
  * forward() returns `.loss`, `.logits`, `.hidden_states`, `.projector_features` (modeling_prismatic.py:632-675);
  * run_forward_pass: the discrete branch (finetune.py:357-378) and compute_diffusion_l1 (:409-430) through the reference signature;
  * get_vla on a non-LIBERO platform (ALOHA constants: 25 x 14 chunk, BOUNDS normalisation);
  * get_action (experiments/robot/robot_utils.py:99-146);
  * hipGraph replay with FiLM (the language average is computed on the device).
All on the reduced-size model, checked against the CPU oracle."""
import importlib
import json
import types

import numpy as np
import pytest
import torch

from oracle import vla_oracle as vo

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
load = importlib.import_module


@pytest.fixture(scope="module")
def world(dev):
    modeling, config_mod, synth = load("openvla-oft_amd.modeling"), load("openvla-oft_amd.config"), load("openvla-oft_amd.synthetic")
    ocfg = vo.tiny_config()
    sd = {k: v.to(BF).float() for k, v in vo.random_state_dict(ocfg, seed=0).items()}
    cfg = config_mod.VLAConfig.from_any(ocfg)
    vla = modeling.OpenVLAForActionPrediction(cfg, sd, device=dev)
    pp = modeling.ProprioProjector(cfg.llm_dim, 8, device=dev,
                                   state_dict={k[len("proprio_projector."):]: v for k, v in sd.items() if k.startswith("proprio_projector.")})
    batch = synth.make_batch(2, seed=21, prompt_lens=[9, 12], image_size=56)
    for k in ("pixel_values", "proprio", "actions"):
        batch[k] = batch[k].to(BF).float()
    return dict(vla=vla, pp=pp, cfg=cfg, ocfg=ocfg, sd=sd, batch=batch, synth=synth, modeling=modeling)


def test_forward_returns_loss_logits_and_projector_features(world):
    vla, pp, ocfg, sd, batch = (world[k] for k in ("vla", "pp", "ocfg", "sd", "batch"))
    o16 = vo.Oracle(ocfg, sd, mode="bf16")
    with torch.no_grad():
        loss_ref, ids_ref = o16.train_forward_discrete(batch)
        hid_ref, P = o16.multimodal_hidden(batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["labels"], batch["proprio"])
        logits_ref = o16.R(o16.lm_logits(hid_ref))
    with torch.no_grad():
        out = vla(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], pixel_values=batch["pixel_values"], labels=batch["labels"],
                  output_hidden_states=True, proprio=batch["proprio"], proprio_projector=pp)
    B, L = batch["input_ids"].shape
    S = P + L
    assert out.hidden_states[-1].shape == (B, S, ocfg.llm_dim)
    assert out.projector_features.shape == (B, P, ocfg.llm_dim), "projected patches + the proprio token (modeling_prismatic.py:586-591)"
    assert out.logits.dtype == torch.float32 and out.logits.shape == (B, S, ocfg.vocab)
    assert abs(out.loss.item() - loss_ref.item()) < 2e-2 * max(1.0, abs(loss_ref.item())), (out.loss.item(), loss_ref.item())
    assert out["loss"] is out.loss and out.past_key_values is None
    counted = batch["labels"][:, 1:] != -100
    pred = out.logits[:, P:-1].argmax(dim=2).cpu()
    agree = (pred[counted] == ids_ref[counted]).float().mean().item()
    valid = torch.cat([torch.ones(B, 1 + P, dtype=torch.bool), batch["attention_mask"][:, 1:]], 1)
    err = ((out.logits.cpu()[valid] - logits_ref[valid]).abs().max() / logits_ref[valid].abs().max()).item()
    print(f"output.loss {out.loss.item():.5f} (oracle {loss_ref.item():.5f}); logits max-normalised error {err:.3e}; argmax ids equal {agree:.3f}")
    assert err < 3e-2 and agree >= 0.9


def test_loss_backward_equals_the_fused_discrete_step(world):
    """`output.loss.backward()` (reference glue) and engine.train_step_discrete (fused path) run the same kernels: same loss, same gradients."""
    vla, pp, batch = world["vla"], world["pp"], world["batch"]
    for m in (vla, pp):
        m.store.zero_grad()
    out = vla(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], pixel_values=batch["pixel_values"], labels=batch["labels"],
              output_hidden_states=True, proprio=batch["proprio"], proprio_projector=pp)
    out.loss.backward()
    g_api = {p.name: p.grad.float().clone() for m in (vla, pp) for p in m.store.params}
    for m in (vla, pp):
        m.store.zero_grad()
    loss_sum, count, _ = vla.engine.train_step_discrete(batch, proprio_projector=pp.comp)
    assert abs(loss_sum.item() / count - out.loss.item()) < 1e-3 * max(1.0, abs(out.loss.item()))
    worst = 0.0
    for m in (vla, pp):
        for p in m.store.params:
            b = p.grad.float()
            if b.abs().max() > 0:
                worst = max(worst, ((g_api[p.name] - b).abs().max() / b.abs().max()).item())
    print(f"output.loss.backward() vs fused discrete step: worst max-normalised gradient difference {worst:.3e}")
    assert worst < 2e-2


def test_run_forward_pass_discrete_branch(world):
    ft = load("openvla-oft_amd.vla_scripts.finetune")
    tok_mod = load("openvla-oft_amd.prismatic.vla.action_tokenizer")
    vla, pp, batch, ocfg, sd = world["vla"], world["pp"], world["batch"], world["ocfg"], world["sd"]
    P = vla.engine.num_patches_total(2, True)
    atok = tok_mod.ActionTokenizer(types.SimpleNamespace(vocab_size=32000))
    for m in (vla, pp):
        m.store.zero_grad()
    loss, metrics = ft.run_forward_pass(vla, None, None, pp, batch, atok, vla.device, False, False, True, False, P)
    assert set(metrics) == {"loss_value", "curr_action_accuracy", "curr_action_l1_loss", "next_actions_accuracy", "next_actions_l1_loss"}
    with torch.no_grad():
        loss_ref, _ = vo.Oracle(ocfg, sd, mode="bf16").train_forward_discrete(batch)
    assert abs(metrics["loss_value"] - loss_ref.item()) < 2e-2 * max(1.0, abs(loss_ref.item()))
    assert 0.0 <= metrics["curr_action_accuracy"] <= 1.0 and np.isfinite(metrics["next_actions_l1_loss"])
    loss.backward()
    assert any(p.grad is not None and p.grad.abs().max() > 0 for _, p in vla.named_parameters())


def test_run_forward_pass_diffusion_with_sampled_l1(dev):
    """use_diffusion + compute_diffusion_l1: MSE loss plus a full DDIM sampling for the L1 metrics (finetune.py:409-448)."""
    modeling, config_mod, synth, ft = (load("openvla-oft_amd.modeling"), load("openvla-oft_amd.config"), load("openvla-oft_amd.synthetic"),
                                       load("openvla-oft_amd.vla_scripts.finetune"))
    ocfg = vo.tiny_config()
    sd = {k: v.to(BF).float() for k, v in vo.random_state_dict(ocfg, seed=2, diffusion=True).items()}
    cfg = config_mod.VLAConfig.from_any(ocfg)
    vla = modeling.OpenVLAForActionPrediction(cfg, sd, device=dev)
    sub = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}  # noqa: E731
    head = modeling.DiffusionActionHead(cfg.llm_dim, cfg.llm_dim, 7, num_diffusion_steps=10, device=dev, state_dict=sub("action_head."))
    pp = modeling.ProprioProjector(cfg.llm_dim, 8, device=dev, state_dict=sub("proprio_projector."))
    nap = modeling.NoisyActionProjector(cfg.llm_dim, device=dev, state_dict=sub("noisy_action_projector."))
    batch = synth.make_batch(2, seed=4, prompt_lens=[9, 7], image_size=56)
    P = vla.engine.num_patches_total(2, True, True)
    torch.manual_seed(0)
    loss, metrics = ft.run_forward_pass(vla, head, nap, pp, batch, None, vla.device, False, True, True, False, P, compute_diffusion_l1=True,
                                        num_diffusion_steps=10)
    assert set(metrics) == {"loss_value", "curr_action_l1_loss", "next_actions_l1_loss"} and all(np.isfinite(v) for v in metrics.values())
    loss.backward()
    loss2, metrics2 = ft.run_forward_pass(vla, head, nap, pp, batch, None, vla.device, False, True, True, False, P, compute_diffusion_l1=False)
    assert set(metrics2) == {"loss_value"}, "without compute_diffusion_l1 no actions are sampled and no L1 is logged (:437)"


def test_get_vla_uses_the_platform_constants(dev, tmp_path):
    """get_vla on ALOHA must build a 25 x 14 / proprio-14 / BOUNDS model like get_action_head / normalize_proprio do."""
    from safetensors.torch import save_file

    utils, C, config_mod = load("openvla-oft_amd.experiments.robot.openvla_utils"), load("openvla-oft_amd.prismatic.vla.constants"), load("openvla-oft_amd.config")
    C.set_platform("aloha")
    try:
        ocfg = vo.tiny_config(action_dim=14, chunk=25, proprio_dim=14, num_images=3)
        sd = {k: v.to(BF) for k, v in vo.random_state_dict(ocfg, seed=0).items()}
        ck = tmp_path / "ckpt"
        ck.mkdir()
        save_file({k: v.contiguous() for k, v in sd.items() if not k.startswith(("action_head.", "proprio_projector.")) and ".lora_" not in k}, str(ck / "model.safetensors"))
        torch.save({k[len("action_head."):]: v for k, v in sd.items() if k.startswith("action_head.")}, ck / "action_head--10_checkpoint.pt")
        torch.save({k[len("proprio_projector."):]: v for k, v in sd.items() if k.startswith("proprio_projector.")}, ck / "proprio_projector--10_checkpoint.pt")
        stats = {"aloha_task": {"action": {"min": [-2.0] * 14, "max": [3.0] * 14, "q01": [-1.0] * 14, "q99": [1.0] * 14},
                                "proprio": {"min": [-3.0] * 14, "max": [3.0] * 14, "q01": [-1.0] * 14, "q99": [1.0] * 14}}}
        (ck / "dataset_statistics.json").write_text(json.dumps(stats))
        rcfg = types.SimpleNamespace(pretrained_checkpoint=str(ck), num_images_in_input=3, use_proprio=True, center_crop=False, unnorm_key="aloha_task",
                                     num_open_loop_steps=25, use_l1_regression=True, use_diffusion=False, model_family="openvla")
        base = config_mod.VLAConfig.from_any(vo.tiny_config())            # the architecture alone: LIBERO-shaped defaults
        vla = utils.get_vla(rcfg, model_config=base)
        assert (vla.cfg.action_dim, vla.cfg.chunk, vla.cfg.proprio_dim, vla.cfg.norm_type) == (14, 25, 14, "bounds")
        head = utils.get_action_head(rcfg, vla.llm_dim)
        pp = utils.get_proprio_projector(rcfg, vla.llm_dim, C.PROPRIO_DIM)
        rng = np.random.default_rng(3)
        obs = {k: rng.integers(0, 256, (224, 224, 3), dtype=np.uint8) for k in ("full_image", "left_wrist_image", "right_wrist_image")}
        obs["state"] = rng.uniform(-1, 1, 14)
        tok = lambda text: [1] + [3 + (ord(c) % 200) for c in text][:12]  # noqa: E731

        class Proc(utils.PrismaticProcessor):        # the tiny test towers take 56 x 56 inputs: subsample after the 224 transform
            def __call__(self, text, image):
                out = super().__call__(text, image)
                out["pixel_values"] = out["pixel_values"][:, :, ::4, ::4].contiguous()
                return out

        robot = load("openvla-oft_amd.experiments.robot.robot_utils")
        state0 = obs["state"].copy()
        acts = robot.get_action(rcfg, vla, obs, "scoop the beans", processor=Proc(tok), action_head=head, proprio_projector=pp)
        assert len(acts) == 25 and all(a.shape == (14,) for a in acts)
        # BOUNDS: proprio normalised with min / max, actions un-normalised with min / max (not the q01 / q99 of LIBERO)
        assert np.allclose(obs["state"], np.clip(2 * (state0 + 3.0) / (6.0 + 1e-8) - 1, -1, 1))
        prompt = torch.tensor([tok(vo.build_prompt("scoop the beans"))])
        pv = torch.cat([Proc(tok)("", obs[k])["pixel_values"] for k in ("full_image", "left_wrist_image", "right_wrist_image")], 1).to(BF).float()
        refs = {}
        for mode in ("fp32", "bf16"):
            o = vo.Oracle(ocfg, {k: v.float() for k, v in sd.items() if ".lora_" not in k}, mode=mode)     # the checkpoint on disk is adapter-free
            o.cfg.norm_type = "bounds"
            refs[mode], _ = o.predict_action(prompt, torch.ones_like(prompt, dtype=torch.bool), pv, proprio=obs["state"], unnorm_stats=stats["aloha_task"]["action"])
        got = np.stack(acts)
        e_hip, e_emu, e_he = np.abs(got - refs["fp32"]).max(), np.abs(refs["bf16"] - refs["fp32"]).max(), np.abs(got - refs["bf16"]).max()
        print(f"ALOHA-shaped get_action, un-normalised to [-2, 3] (x2.5): L-inf hip-fp32 {e_hip:.3e}, emu-fp32 {e_emu:.3e}, hip-emu {e_he:.3e}; max |a| {np.abs(refs['fp32']).max():.2f}")
        assert e_hip <= 1.5 * e_emu + 2.5 * 2 * 2.0 ** -6, "as close to exact arithmetic as the emulation of the reference's bf16 path (+ 2 bf16 ulp at |a| in [2, 4), x2.5)"
        with pytest.raises(ValueError, match="Unsupported model family"):
            robot.get_action(types.SimpleNamespace(model_family="other"), vla, obs, "x")
    finally:
        C.set_platform("libero")


def test_language_average_kernel_and_film_graph_replay(dev):
    """FiLM: the conditioning vector comes from one device kernel (ovla_language_average == the oracle's masked mean), so the FiLM
    forward is capturable: ChunkGraph replay == eager, bit for bit."""
    engine_mod, weights_mod, config_mod, synth, ops = (load("openvla-oft_amd.engine"), load("openvla-oft_amd.weights"), load("openvla-oft_amd.config"),
                                                       load("openvla-oft_amd.synthetic"), load("openvla-oft_amd.ops"))
    ocfg = vo.tiny_config()
    sd = {k: v.to(BF).float() for k, v in vo.random_state_dict(ocfg, seed=1, film=True).items()}
    cfg = config_mod.VLAConfig.from_any(ocfg)
    get, has = weights_mod.make_getter(sd, dev)
    eng = engine_mod.VLAEngine(cfg, get, dev, lora=True, use_proprio=True, head="l1", use_film=True, has=has)
    batch = synth.make_batch(3, seed=8, prompt_lens=[9, 12, 7], image_size=56)
    ids, lab = batch["input_ids"].to(dev), batch["labels"].to(dev)
    avg = eng.language_average(ids, lab)[:3].float().cpu()
    emb = sd["language_model.model.embed_tokens.weight"][batch["input_ids"]]
    keep = ~vo.all_actions_mask(batch["labels"], 7)
    ref = torch.stack([emb[b][keep[b]].mean(0) for b in range(3)]).to(BF).float()
    assert torch.equal(avg, ref), "fp32 mean of bf16 rows, one rounding: bit-exact against the oracle's definition"
    b1 = synth.make_batch(1, seed=9, prompt_lens=[10], image_size=56)
    pv, prop = b1["pixel_values"].to(dev, BF), b1["proprio"].to(dev, BF).reshape(1, -1)
    out = eng.forward(b1["input_ids"], b1["attention_mask"], pv, b1["labels"], proprio=prop, train=False, sel="actions")
    eager = eng.head.fwd(eng.action_hidden(out)[0])[0].clone()
    g = engine_mod.ChunkGraph(eng, 1, b1["input_ids"].shape[1], pv.shape, head=eng.head, use_proprio=True)
    pred, _ = g(b1["input_ids"], b1["attention_mask"], pv, b1["labels"], prop)
    assert torch.equal(pred, eager), "FiLM forward replayed from a hipGraph"
    # a different instruction through the SAME graph changes the conditioning vector (it is computed inside the graph)
    ids2 = b1["input_ids"].clone()
    ids2[0, 1:5] = torch.tensor([17, 900, 4000, 21000])
    pred2, _ = g(ids2, b1["attention_mask"], pv, b1["labels"], prop)
    out2 = eng.forward(ids2, b1["attention_mask"], pv, b1["labels"], proprio=prop, train=False, sel="actions")
    assert torch.equal(pred2, eng.head.fwd(eng.action_hidden(out2)[0])[0]) and not torch.equal(pred2, eager)
