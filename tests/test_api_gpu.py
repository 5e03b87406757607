"""Drop-in boundary on the GPU: the reference-facing classes and glue against the CPU oracle."""
import importlib
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
    stats = {"libero_spatial_no_noops": {"action": {"q01": [-1.0] * 7, "q99": [1.0, 0.5, 2, 1, 1, 1, 1], "min": [-1.0] * 7, "max": [1.0] * 7,
                                                    "mask": [True] * 6 + [False]},
                                         "proprio": {"q01": [-2.0] * 8, "q99": [2.0] * 8, "min": [-3.0] * 8, "max": [3.0] * 8}}}
    vla = modeling.OpenVLAForActionPrediction(cfg, sd, device=dev, norm_stats=stats)
    head = modeling.L1RegressionActionHead(cfg.llm_dim, cfg.llm_dim, 7, device=dev,
                                           state_dict={k[len("action_head."):]: v for k, v in sd.items() if k.startswith("action_head.")})
    pp = modeling.ProprioProjector(cfg.llm_dim, 8, device=dev,
                                   state_dict={"module." + k[len("proprio_projector."):]: v for k, v in sd.items() if k.startswith("proprio_projector.")})
    return dict(vla=vla, head=head, pp=pp, cfg=cfg, ocfg=ocfg, sd=sd, stats=stats, synth=synth, modeling=modeling)


def test_predict_action_l1_and_discrete(world):
    vla, head, pp, ocfg, sd, stats = (world[k] for k in ("vla", "head", "pp", "ocfg", "sd", "stats"))
    g = torch.Generator().manual_seed(5)
    ids = torch.cat([torch.tensor([[1]]), torch.randint(3, 31000, (1, 11), generator=g)], 1)
    mask = torch.ones_like(ids, dtype=torch.bool)
    pv = torch.randn(1, 12, 56, 56, generator=g).to(BF).float()
    proprio = (torch.rand(8, generator=g) * 2 - 1).to(BF).float().numpy()
    o16 = vo.Oracle(ocfg, sd, mode="bf16")
    ref_a, ref_h = o16.predict_action(ids, mask, pv, proprio=proprio, unnorm_stats=stats["libero_spatial_no_noops"]["action"])
    act, hid = vla.predict_action(input_ids=ids, unnorm_key="libero_spatial_no_noops", proprio=proprio, proprio_projector=pp, action_head=head,
                                  pixel_values=pv, attention_mask=mask)
    assert act.shape == (8, 7) and hid.shape == (1, 56, ocfg.llm_dim)
    err = np.abs(act - ref_a).max()
    print(f"continuous actions Linf vs bf16-emulating oracle: {err:.3e}")
    assert err < 5e-2
    # discrete path: argmax of the lm_head logits on the action rows -> 256-bin decode
    ref_d, ref_hd = o16.predict_action(ids, mask, pv, head="discrete")
    logits_ref = o16.lm_logits(ref_hd)[0]
    vla1 = world["vla"]
    vla1.norm_stats = {"only": stats["libero_spatial_no_noops"]}
    act_d, hid_d = vla1.predict_action(input_ids=ids, pixel_values=pv, attention_mask=mask)   # no head -> token path, unnorm_key inferred
    vla1.norm_stats = stats
    tok = vla.logits_for(hid_d[0]).argmax(1).cpu()
    tok_ref = logits_ref.argmax(1)
    exact = (tok == tok_ref).float().mean().item()
    gap = (logits_ref.max(1).values - logits_ref[torch.arange(56), tok]).max().item()
    print(f"action-token indices identical to the oracle: {100 * exact:.1f}% ; worst oracle-logit gap of a differing pick: {gap:.3e}")
    assert exact >= 0.9 and gap < 0.05 * logits_ref.abs().max().item()
    assert act_d.shape == (8, 7)


def test_get_vla_action_glue(world):
    utils = load("openvla-oft_amd.experiments.robot.openvla_utils")
    vla, head, pp = world["vla"], world["head"], world["pp"]
    rng = np.random.default_rng(0)
    # same shape/dtypes as experiments/robot/libero/sample_libero_spatial_observation.pkl (inputs synthetic: the pickle
    # is not loaded -- see DESIGN.md)
    obs = {"full_image": rng.integers(0, 256, (224, 224, 3), dtype=np.uint8), "wrist_image": rng.integers(0, 256, (224, 224, 3), dtype=np.uint8),
           "state": rng.uniform(-1, 1, 8), "task_description": "pick up the black bowl"}
    state0 = obs["state"].copy()
    cfg = types.SimpleNamespace(num_images_in_input=2, use_proprio=True, center_crop=True, unnorm_key="libero_spatial_no_noops", num_open_loop_steps=8)
    tok = lambda text: [1] + [3 + (ord(c) % 200) for c in text][:20]  # noqa: E731  (no tokenizer files offline)
    proc = utils.PrismaticProcessor(tok)
    # the tiny test towers run 56x56 images: shrink after the 224 crop by patch subsampling
    class P56(utils.PrismaticProcessor):
        def __call__(self, text, image):
            out = super().__call__(text, image)
            out["pixel_values"] = out["pixel_values"][:, :, ::4, ::4].contiguous()
            return out
    actions = utils.get_vla_action(cfg, vla, P56(tok), obs, obs["task_description"], action_head=head, proprio_projector=pp)
    assert isinstance(actions, list) and len(actions) == 8 and all(a.shape == (7,) for a in actions)
    assert not np.allclose(obs["state"], state0), "get_vla_action normalises obs['state'] in place like the reference"
    assert np.all(np.isfinite(np.stack(actions)))
    with pytest.raises(AssertionError, match="Incorrect image format"):
        utils.get_vla_action(cfg, vla, proc, {**obs, "full_image": obs["full_image"].astype(np.float32)}, "x", action_head=head, proprio_projector=pp)
    # the device image path (one ovla_image_prep launch; taken for 224-pixel models) equals the host functions bit for bit
    ip = load("openvla-oft_amd.image_prep")
    imgs = [obs["full_image"], obs["wrist_image"]]
    for crop in (True, False):
        c = types.SimpleNamespace(center_crop=crop)
        host = torch.cat([ip.apply_transform(im)[None] for im in ip.prepare_images_for_vla(list(imgs), c)], dim=1).to(BF)
        assert torch.equal(utils.device_pixel_values(imgs, c).cpu(), host)
    # simulator-sized frames are resized on the device (lanczos3 + antialias; tests/test_data_path.py checks the arithmetic)
    assert utils.device_pixel_values([np.zeros((256, 256, 3), np.uint8), obs["wrist_image"]], cfg).shape == (1, 12, 224, 224)


def test_autograd_bridge_equals_fused_step(world):
    """Reference-style glue (vla(...) -> mask-gather -> head.predict_action -> L1Loss -> backward) must produce the same
    gradients as the engine's fused training step."""
    ft = load("openvla-oft_amd.vla_scripts.finetune")
    vla, head, pp, synth, cfg = world["vla"], world["head"], world["pp"], world["synth"], world["cfg"]
    batch = synth.make_batch(2, seed=11, prompt_lens=[9, 12], image_size=56)
    P = vla.engine.num_patches_total(2, True)
    for m in (vla, head, pp):
        m.store.zero_grad()
    loss, metrics = ft.run_forward_pass(vla, head, None, pp, batch, None, vla.device, True, False, True, False, P)
    loss.backward()
    g_api = {n: p.grad.float().clone() for m in (vla, head, pp) for n, p in m.named_parameters()}
    assert all(torch.isfinite(v).all() for v in g_api.values()) and len(g_api) > 50
    for m in (vla, head, pp):
        m.store.zero_grad()
    loss_sum, count, _ = vla.engine.train_step_fwd_bwd(batch, action_head=head.comp, proprio_projector=pp.comp)
    assert abs(loss_sum.item() / count - metrics["loss_value"]) < 1e-2
    worst = 0.0
    for m in (vla, head, pp):
        for p in m.store.params:
            a, b = g_api[p.name], p.grad.float()
            if b.abs().max() > 0:
                worst = max(worst, ((a - b).abs().max() / b.abs().max()).item())
    print(f"autograd-bridge vs fused-step gradients: worst max-normalised difference {worst:.3e}")
    assert worst < 2e-2   # identical kernels; the only difference is bf16 rounding of the published .grad and of dL/dpred
    # torch optimizer on the published views moves the very parameters the engine computes with
    before = head.store.flat[BF].clone()
    torch.optim.AdamW(head.parameters(), lr=1e-3).step()
    assert not torch.equal(before, head.store.flat[BF])


def test_finetune_loop(world, tmp_path, monkeypatch):
    ft = load("openvla-oft_amd.vla_scripts.finetune")
    cfg = ft.FinetuneConfig(run_root_dir=tmp_path, dataset_name="libero_spatial_no_noops", batch_size=2, num_images_in_input=2, use_proprio=True,
                            max_steps=4, save_freq=2, wandb_log_freq=1, lr_warmup_steps=2, grad_accumulation_steps=2)
    sd = {k: v.clone() for k, v in world["sd"].items()}
    lines = []
    hist = ft.finetune(cfg, model_config=world["cfg"], state_dict=sd, log=lines.append,
                       dataset=(world["synth"].make_batch(2, seed=s, prompt_lens=[9, 8], image_size=56) for s in range(100)))
    assert len(hist["loss_value"]) == 4 and all(np.isfinite(hist["loss_value"]))
    assert hist["learning_rate"][0] == pytest.approx(5e-4 * (0.1 + 0.9 * 0.5)) and hist["learning_rate"][-1] == pytest.approx(5e-4)
    ck = list(tmp_path.glob("*--2_chkpt"))
    assert len(ck) == 1
    names = sorted(p.name for p in ck[0].iterdir())
    assert "action_head--2_checkpoint.pt" in names and "proprio_projector--2_checkpoint.pt" in names and "lora_adapter" in names
    head_sd = torch.load(ck[0] / "action_head--2_checkpoint.pt", weights_only=True)
    assert "model.fc1.weight" in head_sd and head_sd["model.fc1.weight"].shape == (world["cfg"].llm_dim, 7 * world["cfg"].llm_dim)


def test_finetune_discrete_objective(world, tmp_path):
    """finetune.py:357-378: neither L1 regression nor diffusion -> next-token cross entropy on the action tokens, with the
    reference's token-accuracy / decoded-L1 metrics; only the LoRA adapter (+ proprio projector) is trained and saved."""
    ft = load("openvla-oft_amd.vla_scripts.finetune")
    cfg = ft.FinetuneConfig(run_root_dir=tmp_path, dataset_name="libero_spatial_no_noops", batch_size=2, num_images_in_input=2, use_proprio=True,
                            use_l1_regression=False, use_diffusion=False, max_steps=6, save_freq=5, wandb_log_freq=1, learning_rate=2e-3)
    fixed = world["synth"].make_batch(2, seed=5, prompt_lens=[9, 8], image_size=56)
    hist = ft.finetune(cfg, model_config=world["cfg"], state_dict={k: v.clone() for k, v in world["sd"].items()}, log=lambda *_: None,
                       dataset=(fixed for _ in range(100)))
    assert len(hist["loss_value"]) == 6 and all(np.isfinite(hist["loss_value"])) and hist["loss_value"][-1] < hist["loss_value"][0]
    for k in ("curr_action_accuracy", "curr_action_l1_loss", "next_actions_accuracy", "next_actions_l1_loss"):
        assert len(hist[k]) == 6 and all(np.isfinite(hist[k])), k
    assert 0.0 <= hist["curr_action_accuracy"][-1] <= 1.0
    names = sorted(p.name for p in list(tmp_path.glob("*--5_chkpt"))[0].iterdir())
    assert "lora_adapter" in names and "proprio_projector--5_checkpoint.pt" in names and not any(n.startswith("action_head") for n in names)


def test_finetune_config5_shapes(world, tmp_path):
    """BASELINE.json config 5 ingredients together: ALOHA constants (chunk 25 x action dim 14, proprio 14), 3 images,
    FiLM + diffusion head, through the fine-tune driver (reduced-size model)."""
    ft, config_mod, C = load("openvla-oft_amd.vla_scripts.finetune"), load("openvla-oft_amd.config"), load("openvla-oft_amd.prismatic.vla.constants")
    C.set_platform("aloha")
    try:
        cfg = ft.FinetuneConfig(run_root_dir=tmp_path, dataset_name="aloha_scoop_x_into_bowl", batch_size=2, num_images_in_input=3, use_proprio=True,
                                use_l1_regression=False, use_diffusion=True, num_diffusion_steps=50, use_film=True, max_steps=3, wandb_log_freq=1,
                                save_freq=2)
        base = world["cfg"]
        mc = config_mod.VLAConfig(**{**base.__dict__, "num_images": 3})
        hist = ft.finetune(cfg, model_config=mc, log=lambda *_: None,
                           dataset=(world["synth"].make_batch(2, seed=s, prompt_lens=[9, 7], image_size=56, chunk=25, action_dim=14, proprio_dim=14,
                                                              num_images=3) for s in range(50)))
        assert len(hist["loss_value"]) == 3 and all(np.isfinite(hist["loss_value"]))
        ck = list(tmp_path.glob("*--2_chkpt"))[0]
        names = sorted(p.name for p in ck.iterdir())
        assert {"action_head--2_checkpoint.pt", "noisy_action_projector--2_checkpoint.pt", "proprio_projector--2_checkpoint.pt",
                "vision_backbone--2_checkpoint.pt"} <= set(names)
        vb = torch.load(ck / "vision_backbone--2_checkpoint.pt", weights_only=True)
        assert any(k.endswith("blocks.0.scale.weight") for k in vb)
    finally:
        C.set_platform("libero")


def test_resume_continues_the_same_trajectory(world, tmp_path, dev):
    """checkpoint -> fresh engine -> resume.  With the optimizer state restored, the next update of the resumed
    run lands on the parameters of the uninterrupted run (fp32 atomics in the weight-gradient GEMMs make the two agree to
    rounding, not bit for bit); without it (the reference saves no optimizer state) the AdamW moments restart and it does not."""
    ft = load("openvla-oft_amd.vla_scripts.finetune")
    engine_mod, weights_mod = load("openvla-oft_amd.engine"), load("openvla-oft_amd.weights")
    cfg = ft.FinetuneConfig(run_root_dir=tmp_path, dataset_name="libero_spatial_no_noops", batch_size=2, num_images_in_input=2, use_proprio=True,
                            save_freq=2, wandb_log_freq=1, max_steps=4)

    def fresh():
        get, has = weights_mod.make_getter({k: v.clone() for k, v in world["sd"].items()}, dev)
        return engine_mod.VLAEngine(world["cfg"], get, dev, lora=True, use_proprio=True, head="l1", has=has)

    def update(e, i):
        b = world["synth"].make_batch(2, seed=100 + i, prompt_lens=[9, 8], image_size=56)
        e.zero_grad(); e.train_step_fwd_bwd(b); e.adamw_step(ft.learning_rate_at(cfg, i)); e.refresh_derived()

    e0 = fresh()
    for i in range(3):
        update(e0, i)
    ck = ft.save_training_checkpoint(tmp_path / "run", 3, e0, world["stats"], rank=0)
    assert (ck / "optimizer_state--3_checkpoint.safetensors").is_file() and (ck / "lora_adapter" / "adapter_config.json").is_file()
    update(e0, 3)
    straight = {k: v.float().cpu().clone() for k, v in e0.export_trainable("data").items()}

    def dist_to_straight(e):
        cur = e.export_trainable("data")
        num = sum(((cur[k].float().cpu() - straight[k]) ** 2).sum().item() for k in straight)
        return (num / sum((straight[k] ** 2).sum().item() for k in straight)) ** 0.5

    e1 = fresh()
    info = ft.load_training_checkpoint(ck, 3, e1)
    assert info["optimizer"] and not info["missing"] and e1.stores[0].step == 3
    update(e1, 3)
    e2 = fresh()
    assert not ft.load_training_checkpoint(ck, 3, e2, load_optimizer=False)["optimizer"]
    update(e2, 3)
    d1, d2 = dist_to_straight(e1), dist_to_straight(e2)
    print(f"resumed vs uninterrupted (rel-L2 over all trainable tensors): with optimizer state {d1:.3e}, without {d2:.3e}")
    assert d1 < 2e-3 and d2 > 5 * d1
    # and through the reference-shaped entry point: finetune(resume=True) reads the same directory
    lines = []
    ft.finetune(ft.FinetuneConfig(run_root_dir=tmp_path, dataset_name="libero_spatial_no_noops", batch_size=2, num_images_in_input=2, use_proprio=True,
                                  max_steps=5, wandb_log_freq=1, resume=True, resume_step=3, vla_path=str(ck)),
                model_config=world["cfg"], state_dict={k: v.clone() for k, v in world["sd"].items()}, log=lines.append,
                dataset=(world["synth"].make_batch(2, seed=200 + s, prompt_lens=[9, 8], image_size=56) for s in range(10)))
    assert any("resumed step 3" in str(l) and "optimizer state: True" in str(l) for l in lines), lines[:4]


def test_deploy_server_act_endpoint(world):
    """vla-scripts/deploy.py mirror: POST /act with a json_numpy-style payload -> action chunk, equal to calling get_vla_action
    directly; the double-encoded form and the "error" answer to a malformed request behave like the reference."""
    import json

    from fastapi.testclient import TestClient

    dep, utils = load("openvla-oft_amd.vla_scripts.deploy"), load("openvla-oft_amd.experiments.robot.openvla_utils")
    rng = np.random.default_rng(1)
    obs = {"full_image": rng.integers(0, 256, (224, 224, 3), dtype=np.uint8), "wrist_image": rng.integers(0, 256, (224, 224, 3), dtype=np.uint8),
           "state": rng.uniform(-1, 1, 8), "instruction": "pick up the black bowl"}
    cfg = dep.DeployConfig(num_images_in_input=2, use_proprio=True, center_crop=True, unnorm_key="libero_spatial_no_noops", num_open_loop_steps=8)

    class P56(utils.PrismaticProcessor):       # the tiny test towers take 56 x 56 inputs
        def __call__(self, text, image):
            out = super().__call__(text, image)
            out["pixel_values"] = out["pixel_values"][:, :, ::4, ::4].contiguous()
            return out

    tok = lambda text: [1] + [3 + (ord(c) % 200) for c in text][:20]  # noqa: E731
    server = dep.OpenVLAServer(cfg, vla=world["vla"], processor=P56(tok), action_head=world["head"], proprio_projector=world["pp"])
    try:
        client = TestClient(server.build_app())
        r = client.post("/act", json=dep._encode(obs))
        assert r.status_code == 200
        acts = dep._decode(r.json())
        direct = utils.get_vla_action(cfg, world["vla"], P56(tok), {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in obs.items()},
                                      obs["instruction"], action_head=world["head"], proprio_projector=world["pp"])
        assert len(acts) == 8 and all(np.array_equal(a, b) for a, b in zip(acts, direct)), "server == direct call (graph replay == eager)"
        r2 = client.post("/act", json={"encoded": json.dumps(dep._encode(obs))})
        acts2 = dep._decode(json.loads(r2.json()))
        assert all(np.array_equal(a, b) for a, b in zip(acts2, direct))
        assert client.post("/act", json={"instruction": "no images"}).json() == "error"
    finally:
        world["vla"].enable_graph_replay(False)


def test_film_train_to_eval_handoff_through_get_vla(world, tmp_path, dev):
    """configs[4]'s train -> eval hand-off at the glue level (experiments/robot/openvla_utils.py:295-299, 311-349; finetune.py:640-655): two
    FiLM + LoRA optimisation steps -> `save_training_checkpoint` (writes `vision_backbone--2_checkpoint.pt` = the whole FiLM-wrapped backbone in
    the reference's key layout) -> merge script (merged shards next to the adapter) -> `get_vla(cfg)` with `cfg.use_film` (reads that file
    through `find_checkpoint_file`, `module.` prefix tolerated) + `get_action_head` / `get_proprio_projector` -> `get_vla_action(use_film=True)`.
    The effective evaluation model is the one the reference assembles: decoder + projector adapters MERGED, towers base + adapters UNMERGED + FiLM;
    the in-memory training engine brought to the same state reproduces its actions BIT FOR BIT."""
    from safetensors.torch import save_file

    engine_mod, weights_mod, ft, utils, merge_mod = (load("openvla-oft_amd." + m) for m in (
        "engine", "weights", "vla_scripts.finetune", "experiments.robot.openvla_utils", "vla_scripts.merge_lora_weights_and_save"))
    ocfg, cfg = world["ocfg"], world["cfg"]
    sd = {k: v.to(BF) for k, v in vo.random_state_dict(ocfg, seed=4, film=True).items()}
    get, has = weights_mod.make_getter(sd, dev)
    eng = engine_mod.VLAEngine(cfg, get, dev, lora=True, use_proprio=True, head="l1", use_film=True, has=has)
    for i in range(2):
        b = world["synth"].make_batch(2, seed=300 + i, prompt_lens=[9, 8], image_size=56)
        eng.zero_grad(); eng.train_step_fwd_bwd(b); eng.adamw_step(2e-3); eng.refresh_derived()
    ck = ft.save_training_checkpoint(tmp_path / "run", 2, eng, world["stats"], rank=0)
    vb = torch.load(ck / "vision_backbone--2_checkpoint.pt", weights_only=True)
    assert "vision_backbone.featurizer.blocks.0.block.attn.qkv.base_layer.weight" in vb and "vision_backbone.featurizer.blocks.0.scale.weight" in vb
    assert "vision_backbone.fused_featurizer.blocks.1.block.mlp.fc1.lora_A.default.weight" in vb and vb["vision_backbone.featurizer.blocks.0.scale.weight"].dtype == torch.float32
    torch.save({"module." + k: v for k, v in vb.items()}, ck / "vision_backbone--2_checkpoint.pt")        # as the reference's DDP wrapper saves it
    base_dir = tmp_path / "base"; base_dir.mkdir()
    base = {k: v.contiguous() for k, v in sd.items() if k.startswith(("vision_backbone.", "projector.", "language_model.")) and ".lora_" not in k
            and ".scale." not in k and ".shift." not in k}
    save_file(base, str(base_dir / "model.safetensors"))
    merge_mod.main(merge_mod.ConvertConfig(base_checkpoint=base_dir, lora_finetuned_checkpoint_dir=ck), model_config=cfg, device=dev)
    rcfg = types.SimpleNamespace(pretrained_checkpoint=str(ck), use_film=True, use_l1_regression=True, use_diffusion=False, num_images_in_input=2,
                                 use_proprio=True, center_crop=True, unnorm_key="libero_spatial_no_noops", num_open_loop_steps=8,
                                 load_in_8bit=False, load_in_4bit=False)
    vla = utils.get_vla(rcfg, model_config=cfg)
    assert vla.engine.use_film and vla.norm_stats == world["stats"]
    assert all(l.has_lora for l in vla.engine.dino.linears() if hasattr(l, "has_lora")) and not any(l.has_lora for l in vla.engine.llm.linears())
    head, pp = utils.get_action_head(rcfg, vla.llm_dim), utils.get_proprio_projector(rcfg, vla.llm_dim, 8)

    class P56(utils.PrismaticProcessor):       # the tiny test towers take 56 x 56 inputs
        def __call__(self, text, image):
            out = super().__call__(text, image)
            out["pixel_values"] = out["pixel_values"][:, :, ::4, ::4].contiguous()
            return out

    rng = np.random.default_rng(3)
    obs = {"full_image": rng.integers(0, 256, (224, 224, 3), dtype=np.uint8), "wrist_image": rng.integers(0, 256, (224, 224, 3), dtype=np.uint8),
           "state": rng.uniform(-1, 1, 8)}
    tok = lambda text: [1] + [3 + (ord(c) % 200) for c in text][:20]  # noqa: E731
    seen, rec = {}, {}
    orig = vla.predict_action
    vla.predict_action = lambda **kw: (rec.update(kw), orig(**kw))[1]
    acts = utils.get_vla_action(rcfg, vla, P56(tok), obs, "pick up the bowl", action_head=head, proprio_projector=pp, use_film=True)
    seen = dict(rec)                           # (the spy also records the refused call below)
    with pytest.raises(ValueError):
        utils.get_vla_action(rcfg, vla, P56(tok), dict(obs, state=rng.uniform(-1, 1, 8)), "pick up the bowl", action_head=head, proprio_projector=pp, use_film=False)
    # the training engine in the evaluation state the reference assembles: decoder + projector merged, towers as trained
    for lin in list(eng.llm.linears()) + list(eng.proj):
        lin.merge()
    ids = torch.cat([seen["input_ids"], torch.tensor([[29871]])], 1) if int(seen["input_ids"][0, -1]) != 29871 else seen["input_ids"]
    ids = torch.cat([ids, torch.ones((1, 56), dtype=torch.int64), torch.tensor([[2]])], 1)
    labels = torch.full_like(ids, -100); labels[:, -57:] = 31744; labels[:, -1] = 2
    out = eng.forward(ids, torch.ones_like(ids, dtype=torch.bool), seen["pixel_values"], labels, proprio=torch.as_tensor(seen["proprio"], dtype=torch.float32),
                      train=False, sel="actions")
    pred = eng.head.fwd(eng.action_hidden(out)[0])[0].float().cpu().numpy().reshape(8, 7)
    want = vla._unnormalize_actions(pred, "libero_spatial_no_noops")
    assert np.array_equal(np.stack(acts), want), f"get_vla(use_film) actions differ from the in-memory model's by {np.abs(np.stack(acts) - want).max():.3e}"
