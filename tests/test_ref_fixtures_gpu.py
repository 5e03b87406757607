"""The HIP path against G16-G19: vectors produced by the REFERENCE's own `modeling_prismatic.py` / `film_vit_wrapper.py`
(tests/golden/make_golden_ref_model.py; stock HF Llama underneath, duck-typed towers).  Nothing here computes with the oracle: the committed
fp32 numbers are the expectation, the tolerance is what bf16 weights + activations cost on the tiny model (the integer / data-movement
pieces are compared bit for bit)."""
import importlib
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import vla_oracle as vo     # seeded weights + config only (`random_state_dict`, `tiny_config`): no oracle arithmetic

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
G = Path(__file__).resolve().parent / "golden"
load = importlib.import_module


def fixture(name):
    return dict(np.load(G / name, allow_pickle=False))


def ref_sd(g, diffusion=False):
    sd = vo.random_state_dict(vo.tiny_config(), seed=int(g["sd_seed"]), lora=False, film=True, diffusion=diffusion)
    chk = sum(float(v.double().abs().sum()) for v in sd.values())
    assert np.isclose(chk, float(g["sd_diffusion_checksum" if diffusion else "sd_checksum"]), rtol=1e-12)
    return sd


def relmax(a, ref):
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-12))


def build(dev, g, mode, film, diffusion=False, stats=None):
    modeling, config_mod = load("openvla-oft_amd.modeling"), load("openvla-oft_amd.config")
    ocfg = vo.tiny_config()
    sd = ref_sd(g, diffusion)
    if not film:
        sd = {k: v for k, v in sd.items() if ".scale." not in k and ".shift." not in k}
    cfg = config_mod.VLAConfig.from_any(ocfg)
    cfg.mask_mode = mode
    vla = modeling.OpenVLAForActionPrediction(cfg, sd, device=dev, norm_stats=stats, lora=False, use_film=film)
    sub = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}  # noqa: E731
    pp = modeling.ProprioProjector(cfg.llm_dim, 8, device=dev, state_dict=sub("proprio_projector."))
    return vla, cfg, sd, sub, pp


def test_g16_projector_chain_matches_reference(dev):
    """PrismaticProjector (modeling_prismatic.py:231-262) as the engine runs it: three epilogue GEMMs forward (bias + GELU fused), data
    gradients + TN weight gradients backward, against the reference module's own outputs and autograd gradients."""
    g = fixture("g16_ref_projector.npz")
    engine_mod, ops = load("openvla-oft_amd.engine"), load("openvla-oft_amd.ops")
    t = lambda k: torch.from_numpy(g[k]).to(dev)  # noqa: E731
    store = engine_mod.ParamStore(dev)
    lins = [engine_mod.FullLinear(store, f"fc{i}", t(f"projector.fc{i}.weight").to(BF), t(f"projector.fc{i}.bias").to(BF)) for i in (1, 2, 3)]
    store.finalize()
    for lin in lins:
        lin.refresh_derived()
    store.zero_grad()
    x = t("x").reshape(64, 48).to(BF).contiguous()
    z1 = torch.empty((64, 192), dtype=BF, device=dev); z2 = torch.empty((64, 40), dtype=BF, device=dev)
    h1, s1 = lins[0].fwd(x, act=ops.ACT_GELU, c_pre=z1)
    h2, s2 = lins[1].fwd(h1, act=ops.ACT_GELU, c_pre=z2)
    y, s3 = lins[2].fwd(h2)
    assert relmax(y.float().cpu().numpy(), g["y"].reshape(64, 40)) < 2e-2
    dy = t("dy").reshape(64, 40).to(BF).contiguous()
    d = ops.act_bwd(z2, lins[2].bwd(dy, s3), ops.ACT_GELU)
    d = ops.act_bwd(z1, lins[1].bwd(d, s2), ops.ACT_GELU)
    dx = lins[0].bwd(d, s1)
    torch.cuda.synchronize()
    assert relmax(dx.float().cpu().numpy(), g["dx"].reshape(64, 48)) < 3e-2
    for i, lin in zip((1, 2, 3), lins):
        assert relmax(lin.W.grad.float().cpu().numpy().reshape(g[f"grad.fc{i}.weight"].shape), g[f"grad.fc{i}.weight"]) < 3e-2, i
        assert relmax(lin.b.grad.float().cpu().numpy(), g[f"grad.fc{i}.bias"]) < 3e-2, i


def test_g17_assembly_kernel_matches_reference_bit_for_bit(dev):
    """`ovla_assemble_multimodal` against the reference's `_process_action_masks` + zeroing / `_replace_input_embeddings` +
    `_build_multimodal_attention` (modeling_prismatic.py:395-496, 620-621) on ragged right-padded rows: pure data movement, so the bf16 image of
    the reference's output is reproduced exactly; the action-row indices equal the shift-by-one positions of the reference mask."""
    g = fixture("g17_ref_multimodal_helpers.npz")
    ops = load("openvla-oft_amd.ops")
    B, L = g["input_ids"].shape
    D, Pn = g["emb"].shape[2], g["patches"].shape[1]
    labels = torch.from_numpy(g["labels"]).to(dev)
    table = torch.from_numpy(g["emb"]).reshape(B * L, D).to(dev, BF).contiguous()      # one table row per (b, position): ids index it
    ids = torch.arange(B * L, dtype=torch.int64, device=dev).view(B, L)
    patches = torch.from_numpy(g["patches"]).to(dev, BF).contiguous()
    noisy = torch.from_numpy(g["noisy_features"]).to(dev, BF).contiguous()
    bf = lambda a: torch.from_numpy(a).to(BF)  # noqa: E731
    mask = g["all_actions_mask"]
    out, rows = ops.assemble_multimodal(ids, labels, table, patches, A=56, noisy=noisy)
    ref_replaced = np.concatenate([g["replaced"][:, :1], g["patches"], g["replaced"][:, 1:]], axis=1)
    assert torch.equal(out.cpu(), bf(ref_replaced)), "noisy-action features in the action slots (_replace_input_embeddings)"
    out0, rows0 = ops.assemble_multimodal(ids, labels, table, patches, A=56)
    zeroed = g["emb"] * ~mask[..., None]
    ref_zeroed = np.concatenate([zeroed[:, :1], g["patches"], zeroed[:, 1:]], axis=1)
    assert torch.equal(out0.cpu(), bf(ref_zeroed)), "zeroed action embeddings (:620-621)"
    S = Pn + L
    want = np.stack([b * S + Pn + np.nonzero(mask[b])[0] - 1 for b in range(B)])       # hidden at token i - 1 predicts token i
    assert np.array_equal(rows.cpu().numpy(), want) and np.array_equal(rows0.cpu().numpy(), want)
    # the multimodal attention mask of the reference is right padding again: the engine's kv_len = text length + P describes it exactly
    lens = g["attention_mask"].sum(1)
    assert np.array_equal(g["mm_mask"], np.arange(S)[None, :] < (lens + Pn)[:, None])


@pytest.mark.parametrize("n_img", [1, 2, 3])
def test_g18_vision_backbones_match_reference(dev, n_img):
    """Tower outputs through the HIP engine (both towers, block depth-2, prefix drop, feature / image concat; FiLM affine in the attention-output
    GEMM epilogue) against PrismaticVisionBackbone.forward / FiLMedPrismaticVisionBackbone.forward run by the reference."""
    g = fixture("g18_ref_film_backbone.npz")
    engine_mod, weights_mod, config_mod, ops = (load("openvla-oft_amd." + m) for m in ("engine", "weights", "config", "ops"))
    pv = torch.from_numpy(g[f"backbone.i{n_img}.pixel_values"]).to(dev, BF)
    lang = torch.from_numpy(g[f"backbone.i{n_img}.language"])
    for film in (False, True):
        ocfg = vo.tiny_config(num_images=n_img)
        sd = ref_sd(g)
        if not film:
            sd = {k: v for k, v in sd.items() if ".scale." not in k and ".shift." not in k}
        cfg = config_mod.VLAConfig.from_any(ocfg)
        get, has = weights_mod.make_getter(sd, dev)
        eng = engine_mod.VLAEngine(cfg, get, dev, lora=False, use_proprio=False, head="none", has=has, use_film=film)
        B = pv.shape[0]
        film_avg = None
        if film:
            film_avg = torch.zeros((8, cfg.llm_dim), dtype=BF, device=dev)
            film_avg[:B] = lang.mean(dim=1).to(dev, BF)                    # film_vit_wrapper.py:243
        feats = {}
        orig = eng.proj[0].fwd

        def spy(x, **kw):
            feats["x"] = x.clone()
            return orig(x, **kw)

        eng.proj[0].fwd = spy
        eng.vision_fwd(pv.contiguous(), False, film_avg)
        torch.cuda.synchronize()
        ref = g[f"backbone.i{n_img}.{'film' if film else 'plain'}"]
        got = feats["x"].float().cpu().numpy().reshape(ref.shape)
        err = relmax(got, ref)
        print(f"G18 backbone I={n_img} film={film}: rel-max err {err:.3e}")
        assert err < 3e-2


@pytest.mark.parametrize("film", [False, True])
@pytest.mark.parametrize("mode", ["causal", "bidirectional"])
def test_g19_forward_matches_reference(dev, mode, film):
    """`vla(...)` of the mirror (modeling.OpenVLAForActionPrediction.forward) against the reference's PrismaticForConditionalGeneration.forward:
    hidden_states[-1] on the valid rows, projector_features, the cross-entropy `.loss`, `.logits` (band, argmax on confident rows)."""
    g = fixture("g19_ref_forward_predict.npz")
    vla, cfg, sd, sub, pp = build(dev, g, mode, film)
    tag = f"{mode}.{'film' if film else 'plain'}"
    b = {k: torch.from_numpy(g[k]) for k in ("input_ids", "attention_mask", "labels", "pixel_values", "proprio")}
    valid = g["attention_mask"]
    with torch.no_grad():
        out = vla(input_ids=b["input_ids"], attention_mask=b["attention_mask"], pixel_values=b["pixel_values"].to(dev, BF), labels=b["labels"],
                  output_hidden_states=True, proprio=b["proprio"], proprio_projector=pp, use_film=film)
        hidden = out.hidden_states[-1].float().cpu().numpy()
        P = hidden.shape[1] - valid.shape[1]
        mm_valid = np.concatenate([valid[:, :1], np.ones((valid.shape[0], P), bool), valid[:, 1:]], axis=1)
        ref = g[tag + ".l1.hidden"]
        err = float(np.abs(hidden - ref)[mm_valid].max() / np.abs(ref[mm_valid]).max())
        print(f"G19 {tag}: hidden rel-max err {err:.3e}")
        assert err < 4e-2
        if mode == "causal":
            assert relmax(out.projector_features.float().cpu().numpy(), g[tag + ".l1.projector_features"]) < 3e-2
        loss = float(out.loss)
        assert abs(loss - float(g[tag + ".l1.loss"])) < 5e-2 * max(1.0, float(g[tag + ".l1.loss"])), (loss, float(g[tag + ".l1.loss"]))
        if not film:
            logits = out.logits.float().cpu().numpy()
            band = g[tag + ".l1.logits_band"]
            assert float(np.abs(logits[..., 31700:32064] - band)[mm_valid].max()) < 4e-2 * np.abs(band[mm_valid]).max() + 5e-2
            agree = (logits.argmax(-1) == g[tag + ".l1.logits_argmax"])[mm_valid].mean()
            assert agree > 0.9, f"argmax agreement on valid rows {agree:.3f} (random lm_head: no margin on some rows)"


@pytest.mark.parametrize("film", [False, True])
@pytest.mark.parametrize("mode", ["causal", "bidirectional"])
def test_g19_predict_action_matches_reference(dev, mode, film):
    """`vla.predict_action` of the mirror against the reference's (modeling_prismatic.py:946-1060): L1 head, discrete decode, both prompt forms."""
    g = fixture("g19_ref_forward_predict.npz")
    modeling = load("openvla-oft_amd.modeling")
    stats = {"libero": {"action": {k[len("stats."):]: v.tolist() for k, v in g.items() if k.startswith("stats.")}}}
    vla, cfg, sd, sub, pp = build(dev, g, mode, film, stats=stats)
    head = modeling.L1RegressionActionHead(cfg.llm_dim, cfg.llm_dim, 7, device=dev, state_dict=sub("action_head."))
    tag = f"{mode}.{'film' if film else 'plain'}"
    pv, prop = torch.from_numpy(g["pixel_values"][:1]).to(dev, BF), g["proprio"][0]
    scale = np.where(stats["libero"]["action"]["mask"], 0.5 * (np.array(stats["libero"]["action"]["q99"]) - np.array(stats["libero"]["action"]["q01"])), 1.0)
    for ptag, key in (("p", "prompt_ids"), ("pno", "prompt_ids_no_empty")):
        pid = torch.from_numpy(g[key])
        am = torch.ones_like(pid, dtype=torch.bool)
        act, ah = vla.predict_action(input_ids=pid, unnorm_key="libero", proprio=prop, proprio_projector=pp, action_head=head, use_film=film,
                                     pixel_values=pv, attention_mask=am)
        ref_h = g[f"{tag}.predict.{ptag}.l1.hidden"]
        assert tuple(ah.shape) == ref_h.shape and relmax(ah.float().cpu().numpy(), ref_h) < 4e-2
        e = np.abs((act - g[f"{tag}.predict.{ptag}.l1.actions"]) / scale).max()
        print(f"G19 {tag} {ptag}: L1 actions (normalised units) L-inf {e:.3e}")
        assert act.shape == (8, 7) and e < 6e-2
        act_d, _ = vla.predict_action(input_ids=pid, unnorm_key="libero", proprio=prop, proprio_projector=pp, action_head=None, use_film=film,
                                      pixel_values=pv, attention_mask=am)
        same = np.isclose(act_d, g[f"{tag}.predict.{ptag}.discrete.actions"], atol=1e-12).mean()
        assert same >= 0.85, f"discrete decode: {same:.3f} of the 56 bin centres identical (random lm_head rows leave no margin on the rest)"


@pytest.mark.parametrize("mode", ["causal", "bidirectional"])
def test_g19_diffusion_predict_action_matches_reference_loop(dev, mode):
    """The 5-step DDIM sampler of the mirror against the reference's `_run_diffusion_prediction` loop (:793-877) recorded around the same scheduler
    arithmetic: timestep token, noisy-action embeddings in the action slots, cached vision patches, row slicing."""
    g = fixture("g19_ref_forward_predict.npz")
    modeling = load("openvla-oft_amd.modeling")
    stats = {"libero": {"action": {k[len("stats."):]: v.tolist() for k, v in g.items() if k.startswith("stats.")}}}
    vla, cfg, sd, sub, pp = build(dev, g, mode, True, diffusion=True, stats=stats)
    T = int(g["diffusion.T"])
    head = modeling.DiffusionActionHead(cfg.llm_dim, cfg.llm_dim, 7, num_diffusion_steps=T, device=dev, state_dict=sub("action_head."))
    nap = modeling.NoisyActionProjector(cfg.llm_dim, device=dev, state_dict=sub("noisy_action_projector."))
    pid = torch.from_numpy(g["prompt_ids"])
    act, ah = vla.predict_action(input_ids=pid, unnorm_key="libero", proprio=g["proprio"][0], proprio_projector=pp, action_head=head,
                                 noisy_action_projector=nap, use_film=True, pixel_values=torch.from_numpy(g["pixel_values"][:1]).to(dev, BF),
                                 attention_mask=torch.ones_like(pid, dtype=torch.bool), noise=torch.from_numpy(g["diffusion.start_noise"]))
    scale = np.where(stats["libero"]["action"]["mask"], 0.5 * (np.array(stats["libero"]["action"]["q99"]) - np.array(stats["libero"]["action"]["q01"])), 1.0)
    e = np.abs((act - g[f"{mode}.film.predict.p.diffusion.actions"]) / scale).max()
    eh = relmax(ah.float().cpu().numpy(), g[f"{mode}.film.predict.p.diffusion.hidden"])
    print(f"G19 diffusion {mode}: actions L-inf {e:.3e}, last-step hidden rel-max {eh:.3e}")
    assert e < 0.1 and eh < 6e-2


@pytest.mark.parametrize("mode", ["bidirectional", "causal"])
def test_g21_run_forward_pass_matches_reference(dev, mode):
    """The mirror's `run_forward_pass` (openvla-oft_amd/vla_scripts/finetune.py; HIP engine behind the autograd bridge) against the reference's own function
    executed in the build container (vla-scripts/finetune.py:280-451, fixture G21): L1-regression loss + current / next action L1 metrics, and the discrete
    objective's cross entropy + token accuracies + decoded L1."""
    from tests.duck_tokenizer import DuckTokenizer

    g = fixture("g21_ref_run_forward_pass.npz")
    modeling, ft, AT = load("openvla-oft_amd.modeling"), load("openvla-oft_amd.vla_scripts.finetune"), load("openvla-oft_amd.prismatic.vla.action_tokenizer")
    vla, cfg, sd, sub, pp = build(dev, g, mode, False)
    head = modeling.L1RegressionActionHead(cfg.llm_dim, cfg.llm_dim, 7, device=dev, state_dict=sub("action_head."))
    batch = {k: torch.from_numpy(g[k]) for k in ("input_ids", "attention_mask", "labels", "pixel_values", "proprio", "actions")}
    tok = AT.ActionTokenizer(DuckTokenizer())
    P = 2 * cfg.dino.n_patches + 1
    with torch.no_grad():
        loss, m = ft.run_forward_pass(vla, head, None, pp, batch, tok, dev, True, False, True, False, P)
    ref = float(g[f"{mode}.l1.loss"])
    print(f"G21 {mode}: L1 loss hip {float(loss):.4f} reference {ref:.4f}; curr {m['curr_action_l1_loss']:.4f} / {float(g[f'{mode}.l1.curr_action_l1_loss']):.4f}")
    assert abs(float(loss) - ref) <= 3e-2 * ref
    assert abs(m["curr_action_l1_loss"] - float(g[f"{mode}.l1.curr_action_l1_loss"])) <= 6e-2 and abs(m["next_actions_l1_loss"] - float(g[f"{mode}.l1.next_actions_l1_loss"])) <= 3e-2
    # discrete objective: lm_head with the fixture's boosted action rows
    sd_d = dict(ref_sd(g))
    lm = sd_d["language_model.lm_head.weight"].clone()
    lm[31744:32000] *= float(g["lm_action_gain"])
    sd_d["language_model.lm_head.weight"] = lm
    sd_d = {k: v for k, v in sd_d.items() if ".scale." not in k and ".shift." not in k}
    config_mod = load("openvla-oft_amd.config")
    cfg_d = config_mod.VLAConfig.from_any(vo.tiny_config())
    cfg_d.mask_mode = mode
    vla_d = modeling.OpenVLAForActionPrediction(cfg_d, sd_d, device=dev, lora=False, use_film=False)
    with torch.no_grad():
        loss_d, md = ft.run_forward_pass(vla_d, None, None, pp, batch, tok, dev, False, False, True, False, P)
    ref_d = float(g[f"{mode}.discrete.loss"])
    print(f"G21 {mode}: CE hip {float(loss_d):.4f} reference {ref_d:.4f}; decoded L1 curr {md['curr_action_l1_loss']:.4f} / {float(g[f'{mode}.discrete.curr_action_l1_loss']):.4f}")
    assert abs(float(loss_d) - ref_d) <= 2e-2 * ref_d
    for k in ("curr_action_accuracy", "next_actions_accuracy"):
        assert abs(md[k] - float(g[f"{mode}.discrete.{k}"])) <= 0.03
    for k in ("curr_action_l1_loss", "next_actions_l1_loss"):
        assert abs(md[k] - float(g[f"{mode}.discrete.{k}"])) <= 0.08, (k, md[k], float(g[f"{mode}.discrete.{k}"]))


def test_g21_run_forward_pass_diffusion_matches_reference(dev):
    """The diffusion branch of the mirror's run_forward_pass on the reference's recorded draws (noise, x_t, timesteps): noise-prediction MSE."""
    from tests.duck_tokenizer import DuckTokenizer

    g = fixture("g21_ref_run_forward_pass.npz")
    modeling, ft, AT = load("openvla-oft_amd.modeling"), load("openvla-oft_amd.vla_scripts.finetune"), load("openvla-oft_amd.prismatic.vla.action_tokenizer")
    vla, cfg, sd, sub, pp = build(dev, g, "bidirectional", True, diffusion=True)
    T = int(g["diffusion.T"])
    head = modeling.DiffusionActionHead(cfg.llm_dim, cfg.llm_dim, 7, num_diffusion_steps=T, device=dev, state_dict=sub("action_head."))
    nap = modeling.NoisyActionProjector(cfg.llm_dim, device=dev, state_dict=sub("noisy_action_projector."))
    noise, noisy, ts = torch.from_numpy(g["diffusion.noise"]), torch.from_numpy(g["diffusion.noisy_actions"]), torch.from_numpy(g["diffusion.timesteps"])
    temb = head.time_encoder(ts.float()).to(BF).unsqueeze(1)
    head.sample_noisy_actions = lambda gt, generator=None: dict(noise=noise.to(BF), noisy_actions=noisy.to(BF), diffusion_timestep_embeddings=temb, timesteps=ts)
    batch = {k: torch.from_numpy(g[k]) for k in ("input_ids", "attention_mask", "labels", "pixel_values", "proprio", "actions")}
    P = 2 * cfg.dino.n_patches + 2
    with torch.no_grad():
        loss, m = ft.run_forward_pass(vla, head, nap, pp, batch, AT.ActionTokenizer(DuckTokenizer()), dev, False, True, True, True, P, compute_diffusion_l1=False, num_diffusion_steps=T)
    ref = float(g["diffusion.loss"])
    print(f"G21 diffusion: MSE hip {float(loss):.4f} reference {ref:.4f}")
    assert abs(float(loss) - ref) <= 4e-2 * ref
