"""Deployment-side rows of SURVEY.md §8f(1): LoRA merged into the base weights on device
(vla-scripts/merge_lora_weights_and_save.py:60-67 = peft merge_and_unload) and the single-chunk inference forward replayed
from a hipGraph (BASELINE.json configs[1]).  Reduced-size model, same seeded inputs as the oracle."""
import importlib

import pytest
import torch

from oracle import vla_oracle as vo

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def build(dev, lora_scale_sd=None):
    load = importlib.import_module
    engine_mod, weights_mod, synth, config_mod = (load("openvla-oft_amd.engine"), load("openvla-oft_amd.weights"), load("openvla-oft_amd.synthetic"),
                                                  load("openvla-oft_amd.config"))
    ocfg = vo.tiny_config()
    sd = {k: v.to(BF).float() for k, v in vo.random_state_dict(ocfg, seed=0).items()}
    for k in sd:   # peft initialises lora_B to zero; give the adapters a visible effect so that the merge is actually tested
        if k.endswith("lora_B.weight"):
            sd[k] = (torch.randn(sd[k].shape, generator=torch.Generator().manual_seed(hash(k) % 1000)) * 0.05).to(BF).float()
    cfg = config_mod.VLAConfig.from_any(ocfg)
    get, has = weights_mod.make_getter(sd, dev)
    eng = engine_mod.VLAEngine(cfg, get, dev, lora=True, use_proprio=True, head="l1", has=has)
    return engine_mod, synth, ocfg, cfg, sd, eng


def test_merge_lora_matches_peft_arithmetic_and_oracle(dev):
    engine_mod, synth, ocfg, cfg, sd, eng = build(dev)
    batch = synth.make_batch(2, seed=3, prompt_lens=[10, 7], image_size=56)
    for k in ("pixel_values", "proprio"):
        batch[k] = batch[k].to(BF).float()
    args = (batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["labels"])
    before = eng.forward(*args, proprio=batch["proprio"], train=False)["hidden"].float().cpu()
    lin = eng.llm.layers[0]["qkv"] if isinstance(eng.llm.layers[0], dict) else next(iter(eng.llm.linears()))
    W0, A, Bm = lin.W.float().cpu().clone(), lin.A.data.float().cpu(), lin.B.data.float().cpu()
    eng.merge_lora()
    # peft: weight.data += (B @ A) * scaling on bf16 tensors -> delta rounded to bf16, then the sum rounded to bf16
    gn, r = lin.group_n, lin.r
    exp = torch.cat([(W0[g * gn:(g + 1) * gn] + (lin.scale * (Bm[g * gn:(g + 1) * gn] @ A[g * r:(g + 1) * r])).to(BF).float()).to(BF).float()
                     for g in range(lin.groups)])
    got = lin.W.float().cpu()
    ulp = (got - exp).abs() / (exp.abs().clamp_min(1e-6) * 2.0 ** -7)
    assert (got == exp).float().mean().item() > 0.995 and ulp.max().item() <= 1.01, "merged weight = bf16(W + bf16(s*B@A)) (<= 1 bf16 ulp, fp32 sum order)"
    assert (got != W0).float().mean().item() > 0.5, "the merge changed the weight"
    if lin.WT is not None:
        assert torch.equal(lin.WT.float().cpu(), got.t()), "W^T re-derived after the merge"
    after = eng.forward(*args, proprio=batch["proprio"], train=False)["hidden"].float().cpu()
    valid = torch.cat([torch.ones(2, 1 + before.shape[1] - batch["input_ids"].shape[1], dtype=torch.bool), batch["attention_mask"][:, 1:]], 1)
    # merged and unmerged evaluate the same function; they differ by bf16 rounding of the merged weights only
    e = rel(after[valid], before[valid])
    print(f"merged vs unmerged hidden: {e:.3e}")
    assert e < 3e-2
    # and both agree with the fp32 oracle evaluated on the UNMERGED weights as well as the bf16 emulation does
    with torch.no_grad():
        h32, _ = vo.Oracle(ocfg, sd, mode="fp32").multimodal_hidden(*args, batch["proprio"])
        h16, _ = vo.Oracle(ocfg, sd, mode="bf16").multimodal_hidden(*args, batch["proprio"])
    emu = rel(h16[valid], h32[valid])
    assert rel(after[valid], h32[valid]) < max(2 * emu, 3e-2)
    with pytest.raises(RuntimeError, match="merged"):
        out = eng.forward(*args, proprio=batch["proprio"], train=True)
        eng.backward_from_hidden(torch.zeros_like(out["hidden"]).view(-1, cfg.llm_dim), out["saved"])
    msd = eng.merged_state_dict()
    assert any(k.endswith("q_proj.weight") for k in msd) and all(v.dtype == BF for v in msd.values())


def test_chunk_graph_replay_is_bit_identical_to_eager(dev):
    engine_mod, synth, ocfg, cfg, sd, eng = build(dev)
    eng.merge_lora()
    b1 = synth.make_batch(1, seed=5, prompt_lens=[9], image_size=56)
    b2 = synth.make_batch(1, seed=6, prompt_lens=[9], image_size=56)
    L = b1["input_ids"].shape[1]

    def eager(b):
        out = eng.forward(b["input_ids"], b["attention_mask"], b["pixel_values"].to(BF), b["labels"], proprio=b["proprio"].to(BF), train=False)
        ah, _ = eng.gather_action_hidden(out["hidden"], out["action_rows"])
        return eng.head.fwd(ah)[0].clone(), ah.clone()

    p1, a1 = eager(b1)
    p2, a2 = eager(b2)
    g = engine_mod.ChunkGraph(eng, 1, L, b1["pixel_values"].shape, head=eng.head, use_proprio=True)
    gp1, ga1 = (t.clone() for t in g(b1["input_ids"], b1["attention_mask"], b1["pixel_values"].to(BF), b1["labels"], b1["proprio"].to(BF)))
    gp2, ga2 = (t.clone() for t in g(b2["input_ids"], b2["attention_mask"], b2["pixel_values"].to(BF), b2["labels"], b2["proprio"].to(BF)))
    gp1b, _ = (t.clone() for t in g(b1["input_ids"], b1["attention_mask"], b1["pixel_values"].to(BF), b1["labels"], b1["proprio"].to(BF)))
    assert torch.equal(gp1, p1) and torch.equal(ga1, a1), "replay == eager (same kernels, same order) on the captured input"
    assert torch.equal(gp2, p2) and torch.equal(ga2, a2), "replay on a NEW observation == eager on it"
    assert torch.equal(gp1b, p1), "replays do not leak state"
    assert not torch.equal(p1, p2)
    # training steps in between must not disturb the graph's private buffers / workspaces
    eng2 = build(dev)[5]
    bt = synth.make_batch(2, seed=8, prompt_lens=[9, 8], image_size=56)
    eng2.zero_grad(); eng2.train_step_fwd_bwd(bt); eng2.adamw_step(lr=1e-3); eng2.refresh_derived()
    gp1c, _ = g(b1["input_ids"], b1["attention_mask"], b1["pixel_values"].to(BF), b1["labels"], b1["proprio"].to(BF))
    assert torch.equal(gp1c, p1)


def test_prompt_padded_to_a_bucket_gives_identical_actions(dev):
    """A ChunkGraph is captured per text length; deployments pad the prompt to a bucket.  Right padding must not change
    the action rows at all: masked keys contribute exact zeros and every other op is row-wise."""
    engine_mod, synth, ocfg, cfg, sd, eng = build(dev)
    b = synth.make_batch(1, seed=11, prompt_lens=[7], image_size=56)
    L = b["input_ids"].shape[1]
    pad = 5
    ids = torch.cat([b["input_ids"], torch.full((1, pad), 32000, dtype=torch.int64)], 1)
    am = torch.cat([b["attention_mask"], torch.zeros((1, pad), dtype=b["attention_mask"].dtype)], 1)
    lab = torch.cat([b["labels"], torch.full((1, pad), -100, dtype=torch.int64)], 1)

    def run(i, m, l):
        out = eng.forward(i, m, b["pixel_values"].to(BF), l, proprio=b["proprio"].to(BF), train=False)
        ah, _ = eng.gather_action_hidden(out["hidden"], out["action_rows"])
        return eng.head.fwd(ah)[0].clone()

    assert torch.equal(run(b["input_ids"], b["attention_mask"], b["labels"]), run(ids, am, lab))


def test_merge_script_end_to_end_and_graph_predict_action(dev, tmp_path):
    """merge_lora_weights_and_save.main on local directories (base shards + peft-format lora_adapter/), then the merged
    checkpoint through OpenVLAForActionPrediction.predict_action with and without hipGraph replay."""
    import numpy as np
    from safetensors.torch import load_file, save_file

    load = importlib.import_module
    merge_mod = load("openvla-oft_amd.vla_scripts.merge_lora_weights_and_save")
    weights_mod, modeling, config_mod = load("openvla-oft_amd.weights"), load("openvla-oft_amd.modeling"), load("openvla-oft_amd.config")
    ocfg = vo.tiny_config()
    cfg = config_mod.VLAConfig.from_any(ocfg)
    full = {k: v.to(BF) for k, v in vo.random_state_dict(ocfg, seed=2).items()}
    g = torch.Generator().manual_seed(7)
    adapter = {}
    for k, v in full.items():
        if ".lora_" in k:
            adapter[k] = (torch.randn(v.shape, generator=g) * 0.05).to(BF) if ".lora_B." in k else v
    vlm_prefixes = ("vision_backbone.", "projector.", "language_model.")
    base = {k: v for k, v in full.items() if ".lora_" not in k and k.startswith(vlm_prefixes)}
    base_dir, ft_dir = tmp_path / "base", tmp_path / "ft"
    base_dir.mkdir(); ft_dir.mkdir()
    save_file({k: v.contiguous() for k, v in base.items()}, str(base_dir / "model.safetensors"))
    weights_mod.save_lora_adapter(ft_dir / "lora_adapter", adapter, r=cfg.lora_rank, lora_alpha=cfg.lora_alpha)
    out = merge_mod.main(merge_mod.ConvertConfig(base_checkpoint=base_dir, lora_finetuned_checkpoint_dir=ft_dir, max_shard_bytes=1 << 20),
                         model_config=cfg, device=dev)
    files = sorted(out.glob("model-*.safetensors"))
    assert len(files) > 1 and (out / "model.safetensors.index.json").is_file(), "sharded like save_pretrained"
    merged = {}
    for f in files:
        merged.update(load_file(str(f)))
    assert set(merged) == set(base)
    name = "language_model.model.layers.1.mlp.up_proj"
    W0, A, Bm = base[name + ".weight"].float(), adapter[name + ".lora_A.weight"].float(), adapter[name + ".lora_B.weight"].float()
    exp = (W0 + (cfg.lora_scale * (Bm @ A)).to(BF).float()).to(BF)
    got = merged[name + ".weight"]
    assert (got == exp).float().mean().item() > 0.995 and not torch.equal(got, base[name + ".weight"])
    assert torch.equal(merged["language_model.model.norm.weight"], base["language_model.model.norm.weight"]), "non-Linear tensors pass through"

    stats = {"t": {"action": {"q01": [-1.0] * 7, "q99": [1.0] * 7, "mask": [True] * 6 + [False], "min": [-1.0] * 7, "max": [1.0] * 7}}}
    vla_m = modeling.OpenVLAForActionPrediction(cfg, merged, device=dev, norm_stats=stats)
    vla_l = modeling.OpenVLAForActionPrediction(cfg, {**base, **adapter}, device=dev, norm_stats=stats)
    assert not any(getattr(l, "has_lora", False) for l in vla_m.engine.vlm_linears()), "the merged checkpoint carries no adapters"
    head = modeling.L1RegressionActionHead(cfg.llm_dim, cfg.llm_dim, cfg.action_dim, device=dev, seed=3)
    pp = modeling.ProprioProjector(cfg.llm_dim, cfg.proprio_dim, device=dev, seed=4)
    rng = np.random.default_rng(0)
    ids = torch.tensor([[1] + rng.integers(3, 31000, 9).tolist()], dtype=torch.int64)
    kw = dict(input_ids=ids, unnorm_key="t", proprio=rng.uniform(-1, 1, cfg.proprio_dim).astype(np.float32), proprio_projector=pp, action_head=head,
              pixel_values=torch.randn(1, 12, 56, 56).to(BF), attention_mask=torch.ones_like(ids, dtype=torch.bool))
    a_l, _ = vla_l.predict_action(**kw)
    a_m, h_m = vla_m.predict_action(**kw)
    assert np.abs(a_l - a_m).max() < 5e-2, "merged checkpoint == adapter-carrying model up to bf16 rounding of the merged weights"
    vla_m.enable_graph_replay(True)
    a_g, h_g = vla_m.predict_action(**kw)
    a_g2, _ = vla_m.predict_action(**{**kw, "pixel_values": torch.randn(1, 12, 56, 56).to(BF)})
    a_g3, _ = vla_m.predict_action(**kw)
    assert np.array_equal(a_g, a_m) and torch.equal(h_g, h_m), "graph replay returns exactly the eager actions"
    assert np.array_equal(a_g3, a_m) and not np.array_equal(a_g2, a_m)
    assert len(vla_m._graphs) == 1
    # discrete path (no head) through the graph as well
    vla_m.enable_graph_replay(False)
    d_e, _ = vla_m.predict_action(**{**kw, "action_head": None})
    vla_m.enable_graph_replay(True)
    d_g, _ = vla_m.predict_action(**{**kw, "action_head": None})
    assert np.array_equal(d_e, d_g)
