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


# ---- fixtures whose source of truth is plain torch or the oracle at the time of writing (tests/golden/make_golden_own.py) ----------
def _tiny():
    ocfg = vo.tiny_config()
    sd = vo.random_state_dict(ocfg, seed=0)
    return ocfg, sd, sum(float(v.double().abs().sum()) for v in sd.values())


def test_g6_full_forward_regression():
    """assemble -> LLM -> shift-by-one gather -> L1 head + loss on a ragged collator-shaped batch: the oracle still produces the
    numbers it produced when the fixture was written (its pieces are pinned to the reference by G1-G5, G10)."""
    g = load("g6_full_forward.npz")
    ocfg, sd, chk = _tiny()
    assert chk == pytest.approx(float(g["sd_checksum"]), rel=1e-12), "seeded weights changed: regenerate the fixture deliberately"
    b = {k: torch.from_numpy(g[k]) for k in ("input_ids", "attention_mask", "labels", "pixel_values", "proprio", "actions")}
    o = vo.Oracle(ocfg, sd, mode="fp32")
    with torch.no_grad():
        hidden, P = o.multimodal_hidden(b["input_ids"], b["attention_mask"], b["pixel_values"], b["labels"], b["proprio"])
        loss, pred, ah = o.train_forward(b)
    assert P == int(g["P"]) == 2 * 16 + 1
    assert np.allclose(hidden[0, : g["hidden_valid_row0"].shape[0]].numpy(), g["hidden_valid_row0"], atol=1e-5, rtol=1e-5)
    assert np.allclose(ah.numpy(), g["action_hidden"], atol=1e-5, rtol=1e-5) and np.allclose(pred.numpy(), g["pred"], atol=1e-5)
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-6)
    # the shift-by-one gather: hidden at (token position - 1) of the 56 action slots of row 0 (prompt length 9)
    assert np.allclose(ah[0].numpy(), hidden[0, 1 + P + 9 - 2: 1 + P + 9 - 2 + 56].numpy())


def test_g7_vit_blocks_regression():
    g = load("g7_vit_blocks.npz")
    ocfg, sd, chk = _tiny()
    assert chk == pytest.approx(float(g["sd_checksum"]), rel=1e-12)
    o = vo.Oracle(ocfg, sd, mode="fp32")
    img = torch.from_numpy(g["img"])
    with torch.no_grad():
        assert np.allclose(o.vit(img, "vision_backbone.featurizer.", ocfg.dino).numpy(), g["dino"], atol=1e-5, rtol=1e-5)
        assert np.allclose(o.vit(img, "vision_backbone.fused_featurizer.", ocfg.siglip).numpy(), g["siglip"], atol=1e-5, rtol=1e-5)
    assert g["dino"].shape == (3, 16, 128) and g["siglip"].shape == (3, 16, 144), "prefix tokens dropped, second-to-last block"


def test_g8_lora_linear_matches_plain_torch():
    g = load("g8_lora_linear.npz")
    cfg = vo.tiny_config(lora_rank=8, lora_alpha=4)
    assert cfg.lora_scale == float(g["scale"])
    sd = {"l.weight": torch.from_numpy(g["W"]), "l.bias": torch.from_numpy(g["bias"]),
          "l.lora_A.weight": torch.from_numpy(g["A"]).requires_grad_(True), "l.lora_B.weight": torch.from_numpy(g["B"]).requires_grad_(True)}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = vo.Oracle(cfg, sd).linear(x, "l")
    y.backward(torch.from_numpy(g["dy"]))
    assert np.allclose(y.detach().numpy(), g["y"], atol=1e-6)
    assert np.allclose(x.grad.numpy(), g["dx"], atol=1e-6) and np.allclose(sd["l.lora_A.weight"].grad.numpy(), g["dA"], atol=1e-6)
    assert np.allclose(sd["l.lora_B.weight"].grad.numpy(), g["dB"], atol=1e-6)


def test_g11_config1_plumbing_regression():
    """BASELINE.json configs[0] plumbing on a synthetic observation of the reference pickle's shapes: center crop -> 6*I-channel
    tensor -> tiny fp32 model -> 8 unnormalised actions."""
    g = load("g11_config1_plumbing.npz")
    ocfg, sd, chk = _tiny()
    assert chk == pytest.approx(float(g["sd_checksum"]), rel=1e-12)
    crops = [vo.crop_and_resize_center(g[k]) for k in ("full_image", "wrist_image")]
    assert np.array_equal(crops[0], g["crop_full"]) and np.array_equal(crops[1], g["crop_wrist"])
    pv = torch.cat([vo.image_transform(c, (vo.IMAGENET_MEAN, vo.SIGLIP_MEAN), (vo.IMAGENET_STD, vo.SIGLIP_STD)) for c in crops], 0)[None]
    assert pv.shape == (1, 12, 224, 224) and pv.double().sum().item() == pytest.approx(float(g["pixel_values_sum"]), rel=1e-9)
    assert np.allclose(pv[0, :, 100, 50:54].numpy(), g["pixel_values_probe"])
    prop = vo.normalize_proprio(g["state"], {"q01": [-2.0] * 8, "q99": [2.0] * 8}, ocfg.norm_type)
    assert np.allclose(prop, g["proprio_normalized"])
    ids = torch.from_numpy(g["input_ids"])
    actions, _ = vo.Oracle(ocfg, sd).predict_action(ids, torch.ones_like(ids, dtype=torch.bool), pv[:, :, ::4, ::4].contiguous(), proprio=prop,
                                                    unnorm_stats={"q01": [-1.0] * 7, "q99": [1.0] * 7, "mask": [True] * 6 + [False]})
    assert np.asarray(actions).shape == (8, 7) and np.allclose(actions, g["actions"], atol=1e-5)


def test_jpeg_roundtrip_oracle_reproduces_libjpeg_turbo():
    """G12: the JPEG encode -> decode round trip of resize_image_for_policy (experiments/robot/openvla_utils.py:532-533).  The fixtures are
    libjpeg-turbo's own outputs (tests/golden/make_golden_jpeg.py, through Pillow); the integer restatement in oracle/jpeg_oracle.py must
    reproduce every byte -- odd sizes (edge replication), saturated blocks (range limiting) and three quality settings included."""
    import numpy as np

    from oracle import jpeg_oracle as jo

    g = np.load(G / "g12_jpeg_roundtrip.npz")
    cases = sorted(k[:-4] for k in g.files if k.endswith("__in"))
    assert len(cases) == 7
    for name in cases:
        img = g[name + "__in"]
        for q in (95, 50, 100):
            key = f"{name}__q{q}"
            if key in g.files:
                got = jo.jpeg_roundtrip(img, quality=q)
                assert got.dtype == np.uint8 and np.array_equal(got, g[key]), f"{key}: {int((got != g[key]).sum())} bytes differ from libjpeg-turbo"
    ql, qc = jo.quant_tables(95)
    assert ql[0] == 2 and qc[0] == 2 and ql[63] == 10 and int(ql.min()) == 1


def test_g13_config0_on_the_reference_observation():
    """BASELINE.json configs[0]: `get_vla_action` on the reference's own fixture, experiments/robot/libero/sample_libero_spatial_observation.pkl,
    CPU float32, one chunk.  The observation was extracted from the pickle WITHOUT unpickling (tests/golden/extract_libero_observation.py walks
    the opcodes); here the oracle runs the reference plumbing on it: prompt from the task description (openvla_utils.py:753), center crop of both
    224 x 224 frames (:542-622), dual normalisation into 12 channels (processing_prismatic.py:128-145), proprio normalisation of the 8-d state
    (:645-675), one chunk through the tiny fp32 model, un-normalisation (modeling_prismatic.py:772-791)."""
    import hashlib

    g = load("g13_libero_observation.npz")
    assert hashlib.sha256(g["full_image"].tobytes()).hexdigest()[:16] == "4c2183d66d17204d" and hashlib.sha256(g["wrist_image"].tobytes()).hexdigest()[:16] == "9a8866c682d1f860"
    assert g["full_image"].shape == g["wrist_image"].shape == (224, 224, 3) and g["state"].shape == (8,) and g["state"].dtype == np.float64
    task = str(g["task_description"])
    assert vo.build_prompt(task) == "In: What action should the robot take to pick up the black bowl between the plate and the ramekin and place it on the plate?\nOut:"
    ocfg, sd, _ = _tiny()
    crops = [vo.crop_and_resize_center(g[k]) for k in ("full_image", "wrist_image")]
    pv = torch.cat([vo.image_transform(c, (vo.IMAGENET_MEAN, vo.SIGLIP_MEAN), (vo.IMAGENET_STD, vo.SIGLIP_STD)) for c in crops], 0)[None]
    assert pv.shape == (1, 12, 224, 224) and torch.isfinite(pv).all()
    # LIBERO proprio = eef position | axis-angle | gripper qpos: real magnitudes (z 1.17 m, |rotation| 3.14) need real statistics
    pstats = {"q01": [-0.5, -0.4, 0.8, 2.5, -0.5, -0.5, 0.0, -0.05], "q99": [0.3, 0.4, 1.4, 3.6, 0.5, 0.3, 0.05, 0.0]}
    prop = vo.normalize_proprio(g["state"], pstats, ocfg.norm_type)
    assert prop.shape == (8,) and np.all(np.abs(prop) <= 1.0) and np.abs(prop).max() > 0.05
    rng = np.random.default_rng(13)
    ids = torch.tensor([[1] + rng.integers(3, 31743, 36).tolist() + [29871]])      # no tokenizer files offline: seeded ids of the prompt's length class
    stats = {"q01": [-1.0] * 7, "q99": [1.0] * 7, "mask": [True] * 6 + [False]}
    a1, _ = vo.Oracle(ocfg, sd).predict_action(ids, torch.ones_like(ids, dtype=torch.bool), pv[:, :, ::4, ::4].contiguous(), proprio=prop, unnorm_stats=stats)
    a2, _ = vo.Oracle(ocfg, sd).predict_action(ids, torch.ones_like(ids, dtype=torch.bool), pv[:, :, ::4, ::4].contiguous(), proprio=prop, unnorm_stats=stats)
    assert np.asarray(a1).shape == (8, 7) and np.all(np.isfinite(a1)) and np.array_equal(a1, a2)


@pytest.mark.parametrize("tag,dim,heads,hidden,layerscale,act", [("dino", 128, 2, 256, True, "gelu"), ("siglip", 144, 2, 536, False, "gelu"),
                                                                 ("siglip_tanh", 144, 2, 536, False, "gelu_tanh")])
def test_g14_vit_block_matches_transformers(tag, dim, heads, hidden, layerscale, act):
    """G14: Oracle.vit_block against transformers' Dinov2Layer (LayerScale, head_dim 64) and SiglipEncoderLayer (head_dim 72, MLP width 536, exact and
    tanh GELU) -- independent implementations of the two tower architectures (timm, which the reference calls, is absent).  fp32, <= 2e-5."""
    g = load("g14_hf_vit_blocks.npz")
    sd = {"b." + k[len(tag) + 2:]: torch.from_numpy(g[k]) for k in list(g) if k.startswith(tag + "__") and not k.endswith(("__x", "__y"))}
    vc = vo.VitConfig(dim, 3, heads, hidden, layerscale=layerscale, act=act)
    got = vo.Oracle(vo.tiny_config(), sd).vit_block(torch.from_numpy(g[tag + "__x"]), "b.", vc)
    err = (got - torch.from_numpy(g[tag + "__y"])).abs().max().item()
    assert err < 2e-5, err


def _textured_image(h, w, seed=0):
    rng = np.random.default_rng(seed)
    _, xx = np.mgrid[0:h, 0:w]
    return np.stack([(127 + 100 * np.sin(xx / 17.0 + c) + 20 * rng.standard_normal((h, w))) for c in range(3)], -1).clip(0, 255).astype(np.uint8)


def test_center_crop_bilinear_against_scipy():
    """TensorFlow is absent, so `crop_and_resize_center` (openvla_utils.py:542-622) stays "parity unpinned" against TF itself.  This pins its
    INTERPOLATION against an independent implementation: scipy's order-1 map_coordinates in float64 at the sample positions the TF kernel
    documents (in = y1 (H - 1) + i (y2 - y1)(H - 1) / (out - 1)).  The two may differ by one LSB where float32 and float64 round apart."""
    ndimage = pytest.importorskip("scipy.ndimage")
    img = _textured_image(256, 320)
    out = vo.crop_and_resize_center(img, 0.9, 224)
    side = np.sqrt(0.9)
    o1 = (1 - side) / 2

    def coords(n):
        return o1 * (n - 1) + np.arange(224) * (side * (n - 1) / 223)

    cy, cx = np.meshgrid(coords(256), coords(320), indexing="ij")
    ref = np.stack([ndimage.map_coordinates(img[..., c].astype(np.float64) / 255.0, [cy, cx], order=1, mode="nearest") for c in range(3)], -1)
    ref_u8 = (np.clip(ref, 0, 1) * 255.5).astype(np.uint8)
    d = np.abs(out.astype(int) - ref_u8.astype(int))
    assert d.max() <= 1 and (d == 0).mean() >= 0.999, ((d == 0).mean(), d.max())


@pytest.mark.parametrize("out_hw", [(224, 224), (128, 160)])
def test_lanczos3_resize_close_to_pillow(out_hw):
    """`resize_lanczos3` restates tf.image.resize(method="lanczos3", antialias=True) (scale_and_translate_op.cc; TF absent -> unpinned against TF).
    Pillow's LANCZOS filter is the same kernel family (radius 3, support scaled by the shrink factor, half-pixel centres) in 8-bit fixed point with a
    rounding step between its two passes: the two must agree to one LSB almost everywhere -- a check on the kernel definition and the sampling
    grid, not a bit-exact pin."""
    Image = pytest.importorskip("PIL.Image")
    from oracle import data_oracle as do

    img = _textured_image(256, 320)
    h2, w2 = out_hw
    mine = do.resize_lanczos3(img, h2, w2)
    pil = np.asarray(Image.fromarray(img).resize((w2, h2), Image.LANCZOS))
    d = np.abs(mine.astype(int) - pil.astype(int))
    assert (d <= 1).mean() >= 0.995 and d.mean() <= 0.3 and d.max() <= 8, ((d <= 1).mean(), d.mean(), d.max())


@pytest.mark.parametrize("tag,vc", [("dino", vo.VitConfig(64, 3, 1, 128, n_prefix=5, layerscale=True, image_size=42)),
                                    ("siglip", vo.VitConfig(72, 3, 1, 136, image_size=42))])
def test_g15_tower_wiring_matches_transformers(tag, vc):
    """G15: Oracle.vit -- patch + position embedding, [cls, 4 registers, patches] prefix, the output of block index depth - 2 with no final norm and the
    prefix dropped -- against transformers' Dinov2WithRegistersModel / SiglipVisionModel `hidden_states[-2]` (make_golden_hf_towers.py; timm is absent).
    fp32, <= 5e-5."""
    g = load("g15_hf_towers.npz")
    sd = {"t." + k[len(tag) + 2:]: torch.from_numpy(g[k]) for k in list(g) if k.startswith(tag + "__") and not k.endswith(("__x", "__y"))}
    got = vo.Oracle(vo.tiny_config(), sd).vit(torch.from_numpy(g[tag + "__x"]), "t.", vc)
    want = torch.from_numpy(g[tag + "__y"])
    assert got.shape == want.shape
    err = (got - want).abs().max().item()
    assert err < 5e-5, err


# ======================================================================================================================
# G16-G19: vectors produced by the reference's OWN modeling_prismatic.py / film_vit_wrapper.py (tests/golden/make_golden_ref_model.py)
# ======================================================================================================================
def _ref_sd(g, diffusion=False):
    """The fixtures carry the seed of the oracle's `random_state_dict` their weights came from, and a checksum of the tensors."""
    sd = vo.random_state_dict(vo.tiny_config(), seed=int(g["sd_seed"]), lora=False, film=True, diffusion=diffusion)
    chk = sum(float(v.double().abs().sum()) for v in sd.values())
    assert np.isclose(chk, float(g["sd_diffusion_checksum" if diffusion else "sd_checksum"]), rtol=1e-12), "seeded weights drifted from the fixture's"
    return sd


def test_g16_projector_forward_and_gradients_match_reference():
    """modeling_prismatic.py:231-262 (fused-backbone branch) executed by the reference itself."""
    g = load("g16_ref_projector.npz")
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in g.items() if k.startswith("projector.")}
    o = vo.Oracle(vo.tiny_config(), sd)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = o.projector(x)
    assert np.allclose(y.detach().numpy(), g["y"], atol=2e-6, rtol=1e-5)
    y.backward(torch.from_numpy(g["dy"]))
    assert np.allclose(x.grad.numpy(), g["dx"], atol=2e-6, rtol=1e-4)
    for k, v in sd.items():
        assert np.allclose(v.grad.numpy(), g["grad." + k[len("projector."):]], atol=1e-5, rtol=1e-4), k


def test_g17_multimodal_helpers_match_reference():
    """modeling_prismatic.py:395-496 (masks, embedding replacement, multimodal concat of embeddings / mask / labels, proprio token),
    :734-770 (placeholder ids / labels of predict_action), :772-791 (un-normalisation, both normalisation types)."""
    g = load("g17_ref_multimodal_helpers.npz")
    labels, amask = torch.from_numpy(g["labels"]), torch.from_numpy(g["attention_mask"])
    m = vo.all_actions_mask(labels, 7)
    assert np.array_equal(m.numpy(), g["all_actions_mask"]) and (m.sum(1) == 56).all()
    emb, patches, feats = (torch.from_numpy(g[k]) for k in ("emb", "patches", "noisy_features"))
    # the oracle's in-line restatement of _replace_input_embeddings / the zeroing branch (Oracle.multimodal_hidden)
    rep = emb.clone()
    for b in range(emb.shape[0]):
        rep[b, m[b]] = feats[b]
    assert np.array_equal(rep.numpy(), g["replaced"])
    mm = torch.cat([emb[:, :1], patches, emb[:, 1:]], dim=1)
    ones = torch.ones(patches.shape[:2], dtype=torch.bool)
    assert np.array_equal(mm.numpy(), g["mm_emb"])
    assert np.array_equal(torch.cat([amask[:, :1], ones, amask[:, 1:]], dim=1).numpy(), g["mm_mask"])
    mm_labels = torch.cat([labels[:, :1], torch.full(patches.shape[:2], -100, dtype=labels.dtype), labels[:, 1:]], dim=1)
    assert np.array_equal(mm_labels.numpy(), g["mm_labels"])
    o = vo.Oracle(vo.tiny_config(), {k[3:].replace("fc", "p.fc"): torch.from_numpy(v) for k, v in g.items() if k.startswith("pp.")})
    pf = o.mlp_projector(torch.from_numpy(g["proprio"]), "p.")
    assert np.allclose(torch.cat((patches, pf[:, None, :]), dim=1).numpy(), g["with_proprio"], atol=2e-6)
    # predict_action's input preparation: the oracle builds ids / mask / labels the same way (Oracle.predict_action)
    pid = torch.from_numpy(g["prompt_ids"])
    ids = torch.cat([pid, torch.ones((1, 56), dtype=pid.dtype), torch.full((1, 1), vo.STOP_INDEX, dtype=pid.dtype)], dim=-1)
    lab = torch.full_like(ids, vo.IGNORE_INDEX)
    lab[:, pid.shape[-1]:] = vo.ACTION_TOKEN_BEGIN_IDX + 1
    lab[:, -1] = vo.STOP_INDEX
    assert np.array_equal(ids.numpy(), g["prepared_ids"]) and np.array_equal(lab.numpy(), g["prepared_labels"])
    assert g["prepared_mask"].all() and g["prepared_mask"].shape == ids.shape
    stats = {k[len("stats."):]: v for k, v in g.items() if k.startswith("stats.")}
    assert np.allclose(vo.unnormalize_actions(g["normalized"], stats, "bounds_q99"), g["unnorm_q99"], rtol=0, atol=0)
    assert np.allclose(vo.unnormalize_actions(g["normalized"], stats, "bounds"), g["unnorm_bounds"], rtol=0, atol=0)
    assert np.allclose(vo.unnormalize_actions(g["normalized"], {k: v for k, v in stats.items() if k != "mask"}, "bounds"), g["unnorm_bounds_nomask"], rtol=0, atol=0)


@pytest.mark.parametrize("tower", ["dino", "siglip"])
def test_g18_film_block_forward_and_gradients_match_reference(tower):
    """film_vit_wrapper.py:56-77 executed by the reference around a duck-typed block (LayerScale / none)."""
    g = load("g18_ref_film_backbone.npz")
    cfg = vo.tiny_config()
    prefix, vc = (("vision_backbone.featurizer.", cfg.dino) if tower == "dino" else ("vision_backbone.fused_featurizer.", cfg.siglip))
    sd = _ref_sd(g)
    p = prefix + "blocks.1."
    for nm in ("scale.weight", "scale.bias", "shift.weight", "shift.bias"):
        sd[p + nm] = sd[p + nm].clone().requires_grad_(True)
    o = vo.Oracle(cfg, sd)
    x, avg = torch.from_numpy(g[tower + ".x"]).requires_grad_(True), torch.from_numpy(g[tower + ".avg"]).requires_grad_(True)
    y = o.vit_block(x, p, vc, avg)
    assert np.allclose(y.detach().numpy(), g[tower + ".y"], atol=3e-5, rtol=1e-5)
    y.backward(torch.from_numpy(g[tower + ".dy"]))
    assert np.allclose(x.grad.numpy(), g[tower + ".dx"], atol=3e-5, rtol=1e-4)
    assert np.allclose(avg.grad.numpy(), g[tower + ".davg"], atol=2e-4, rtol=1e-4)
    for nm in ("scale.weight", "scale.bias", "shift.weight", "shift.bias"):
        assert np.allclose(sd[p + nm].grad.numpy(), g[f"{tower}.grad.{nm}"], atol=2e-4, rtol=1e-4), nm


@pytest.mark.parametrize("n_img", [1, 2, 3])
def test_g18_vision_backbones_match_reference(n_img):
    """PrismaticVisionBackbone.forward (modeling_prismatic.py:186-227) and FiLMedPrismaticVisionBackbone.forward (film_vit_wrapper.py:231-276, with
    the reference's own get_intermediate_layers :114-168): block index depth-2, prefix tokens dropped, no final norm, feature / image concat order,
    the language average."""
    g = load("g18_ref_film_backbone.npz")
    cfg = vo.tiny_config(num_images=n_img)
    o = vo.Oracle(cfg, _ref_sd(g))
    pv, lang = torch.from_numpy(g[f"backbone.i{n_img}.pixel_values"]), torch.from_numpy(g[f"backbone.i{n_img}.language"])
    with torch.no_grad():
        assert np.allclose(o.vision_backbone(pv).numpy(), g[f"backbone.i{n_img}.plain"], atol=5e-5)
        assert np.allclose(o.vision_backbone(pv, lang.mean(dim=1)).numpy(), g[f"backbone.i{n_img}.film"], atol=5e-5)


def _g19_batch(g):
    return {k: torch.from_numpy(g[k]) for k in ("input_ids", "attention_mask", "labels", "pixel_values", "proprio", "actions")}


@pytest.mark.parametrize("film", [False, True])
@pytest.mark.parametrize("mode", ["causal", "bidirectional"])
def test_g19_forward_matches_reference(mode, film):
    """PrismaticForConditionalGeneration.forward, multimodal branch (modeling_prismatic.py:571-643) executed by the reference with stock HF Llama
    underneath: hidden_states[-1], logits (band + argmax + logsumexp), the shifted cross entropy, projector features; L1 (zeroed action
    embeddings) and diffusion (noisy-action embeddings + timestep token) inputs; with and without FiLM."""
    g = load("g19_ref_forward_predict.npz")
    cfg = vo.tiny_config()
    tag = f"{mode}.{'film' if film else 'plain'}"
    valid = g["attention_mask"]
    b = _g19_batch(g)
    with torch.no_grad():
        o = vo.Oracle(cfg, _ref_sd(g), mask_mode=mode)
        hidden, P = o.multimodal_hidden(b["input_ids"], b["attention_mask"], b["pixel_values"], b["labels"], b["proprio"], use_film=film)
        mm_valid = np.concatenate([valid[:, :1], np.ones((valid.shape[0], P), bool), valid[:, 1:]], axis=1)
        assert np.abs(hidden.numpy() - g[tag + ".l1.hidden"])[mm_valid].max() < 1e-4
        logits = o.lm_logits(hidden)
        assert np.array_equal(logits.argmax(-1).numpy()[mm_valid], g[tag + ".l1.logits_argmax"][mm_valid])
        assert np.abs(torch.logsumexp(logits, -1).numpy() - g[tag + ".l1.logits_lse"])[mm_valid].max() < 2e-4
        if not film:
            assert np.abs(logits[..., 31700:32064].numpy() - g[tag + ".l1.logits_band"])[mm_valid].max() < 2e-4
        loss, _ = o.train_forward_discrete(b) if not film else (None, None)
        if loss is not None:
            assert abs(loss.item() - float(g[tag + ".l1.loss"])) < 1e-4
        if mode == "causal":
            feats = o.projector(o.vision_backbone(b["pixel_values"], None if not film else o.W("language_model.model.embed_tokens.weight")[b["input_ids"]][
                ~vo.all_actions_mask(b["labels"], 7)].reshape(3, -1, cfg.llm_dim).mean(1)))
            pf = o.mlp_projector(b["proprio"], "proprio_projector.")
            assert np.allclose(torch.cat((feats, pf[:, None]), 1).numpy(), g[tag + ".l1.projector_features"], atol=1e-4)
        # diffusion-style inputs
        od = vo.Oracle(cfg, _ref_sd(g, diffusion=True), mask_mode=mode)
        temb = vo.sinusoidal_encoding(torch.from_numpy(g["timesteps"]), cfg.llm_dim)[:, None, :]
        hidden, P = od.multimodal_hidden(b["input_ids"], b["attention_mask"], b["pixel_values"], b["labels"], b["proprio"],
                                         torch.from_numpy(g["noisy_actions"]), temb, use_film=film)
        mm_valid = np.concatenate([valid[:, :1], np.ones((valid.shape[0], P), bool), valid[:, 1:]], axis=1)
        assert np.abs(hidden.numpy() - g[tag + ".diffusion.hidden"])[mm_valid].max() < 1e-4


@pytest.mark.parametrize("film", [False, True])
@pytest.mark.parametrize("mode", ["causal", "bidirectional"])
def test_g19_predict_action_matches_reference(mode, film):
    """OpenVLAForActionPrediction.predict_action (modeling_prismatic.py:946-1060) executed by the reference: L1 head and discrete decode, a prompt
    that ends with the empty token and one that does not, un-normalised with BOUNDS_Q99 statistics."""
    g = load("g19_ref_forward_predict.npz")
    cfg = vo.tiny_config()
    tag = f"{mode}.{'film' if film else 'plain'}"
    stats = {k[len("stats."):]: v for k, v in g.items() if k.startswith("stats.")}
    o = vo.Oracle(cfg, _ref_sd(g), mask_mode=mode)
    pv, prop = torch.from_numpy(g["pixel_values"][:1]), g["proprio"][0]
    with torch.no_grad():
        for ptag, key in (("p", "prompt_ids"), ("pno", "prompt_ids_no_empty")):
            pid = torch.from_numpy(g[key])
            am = torch.ones_like(pid, dtype=torch.bool)
            act, ah = o.predict_action(pid, am, pv, proprio=prop, unnorm_stats=stats, use_film=film, head="l1")
            assert np.abs(ah.numpy() - g[f"{tag}.predict.{ptag}.l1.hidden"]).max() < 1e-4
            assert np.abs(act - g[f"{tag}.predict.{ptag}.l1.actions"]).max() < 1e-4
            act, _ = o.predict_action(pid, am, pv, proprio=prop, unnorm_stats=stats, use_film=film, head="discrete")
            assert np.array_equal(act, g[f"{tag}.predict.{ptag}.discrete.actions"]), "discrete decode: identical bin centres"


@pytest.mark.parametrize("mode", ["causal", "bidirectional"])
def test_g19_diffusion_predict_action_loop_matches_reference(mode):
    """The reference's denoising loop (modeling_prismatic.py:793-877: timestep token appended to the patches, noisy-action embeddings scattered into the
    action slots, slicing of the action rows, vision patches reused) around the oracle's own DDIM -- pins the wiring, not the scheduler."""
    g = load("g19_ref_forward_predict.npz")
    cfg = vo.tiny_config()
    stats = {k[len("stats."):]: v for k, v in g.items() if k.startswith("stats.")}
    o = vo.Oracle(cfg, _ref_sd(g, diffusion=True), mask_mode=mode)
    pid = torch.from_numpy(g["prompt_ids"])
    with torch.no_grad():
        act, ah = o.predict_action(pid, torch.ones_like(pid, dtype=torch.bool), torch.from_numpy(g["pixel_values"][:1]), proprio=g["proprio"][0],
                                   unnorm_stats=stats, use_film=True, head="diffusion", noise=torch.from_numpy(g["diffusion.start_noise"]),
                                   num_diffusion_steps=int(g["diffusion.T"]))
    assert np.abs(ah.numpy() - g[f"{mode}.film.predict.p.diffusion.hidden"]).max() < 2e-4
    assert np.abs(act - g[f"{mode}.film.predict.p.diffusion.actions"]).max() < 2e-4


# ======================================================================================================================
# G21: the reference's OWN run_forward_pass (vla-scripts/finetune.py:280-451), executed (tests/golden/make_golden_run_forward_pass.py)
# ======================================================================================================================
def _g21_parts(g, mode, sd, diffusion=None):
    """The oracle's pieces composed as the fixture's run composed the reference's: VLM in fp32 (a CPU model), action rows rounded to bf16
    (finetune.py:389-394 `.to(torch.bfloat16)`), head in bf16 (finetune.py:910)."""
    cfg = vo.tiny_config()
    b = {k: torch.from_numpy(g[k]) for k in ("input_ids", "attention_mask", "labels", "pixel_values", "proprio", "actions")}
    o32, o16 = vo.Oracle(cfg, sd, mode="fp32", mask_mode=mode), vo.Oracle(cfg, sd, mode="bf16", mask_mode=mode)
    kw = {}
    if diffusion is not None:
        kw = dict(noisy_actions=diffusion["noisy"], timestep_emb=vo.sinusoidal_encoding(diffusion["timesteps"].float(), cfg.llm_dim)[:, None, :], use_film=True)
    hidden, P = o32.multimodal_hidden(b["input_ids"], b["attention_mask"], b["pixel_values"], b["labels"], b["proprio"], **kw)
    ids = b["labels"][:, 1:]
    m = vo.current_action_mask(ids, cfg.action_dim) | vo.next_actions_mask(ids, cfg.action_dim)
    ah = hidden[:, P:-1][m].reshape(b["input_ids"].shape[0], cfg.chunk * cfg.action_dim, -1)
    return cfg, b, o32, o16, o16.R(ah), P


@pytest.mark.parametrize("mode", ["bidirectional", "causal"])
def test_g21_run_forward_pass_l1_and_discrete_match_reference(mode):
    """L1-regression objective: loss and the current / next action L1 metrics; discrete objective: the next-token cross entropy, the predicted ids
    `logits[:, num_patches:-1].argmax`, both token accuracies and both decoded-L1 metrics -- against the reference function's own numbers."""
    g = load("g21_ref_run_forward_pass.npz")
    sd = _ref_sd(g)
    cfg, b, o32, o16, ah16, P = _g21_parts(g, mode, sd)
    with torch.no_grad():
        pred = o16.l1_head(ah16)
        gt = o16.R(b["actions"])
        l1 = lambda a, c: float(o16.R(o16.R((a - c).abs()).mean()))  # noqa: E731  (torch.nn.L1Loss on bf16 tensors)
        ulp = 2.0 ** -8
        assert abs(l1(gt, pred) - float(g[f"{mode}.l1.loss"])) <= 2 * ulp
        assert abs(l1(gt[:, 0], pred[:, 0]) - float(g[f"{mode}.l1.curr_action_l1_loss"])) <= 3 * ulp
        assert abs(l1(gt[:, 1:], pred[:, 1:]) - float(g[f"{mode}.l1.next_actions_l1_loss"])) <= 2 * ulp
        assert float(g[f"{mode}.l1.loss_value"]) == float(g[f"{mode}.l1.loss"])
        # discrete objective on the lm_head with boosted action rows
        sd_d = dict(sd)
        lm = sd["language_model.lm_head.weight"].clone()
        lm[31744:32000] *= float(g["lm_action_gain"])
        sd_d["language_model.lm_head.weight"] = lm
        loss, pred_ids = vo.Oracle(cfg, sd_d, mode="fp32", mask_mode=mode).train_forward_discrete(b)
    assert abs(loss.item() - float(g[f"{mode}.discrete.loss"])) < 2e-4
    ids = b["labels"][:, 1:]
    valid = ids != -100
    assert np.array_equal(pred_ids.numpy()[valid.numpy()], g[f"{mode}.discrete.predicted_ids"][valid.numpy()])
    for name, mk in (("curr_action", vo.current_action_mask(ids, 7)), ("next_actions", vo.next_actions_mask(ids, 7))):
        acc = ((pred_ids == ids) & mk).sum().item() / mk.sum().item()
        assert abs(acc - float(g[f"{mode}.discrete.{name}_accuracy"])) < 1e-7
        dec = lambda t: vo.decode_token_ids_to_actions(t[mk].numpy())  # noqa: E731
        assert abs(np.abs(dec(pred_ids) - dec(ids)).mean() - float(g[f"{mode}.discrete.{name}_l1_loss"])) < 1e-9


def test_g21_run_forward_pass_diffusion_objective_matches_reference():
    """Diffusion objective (finetune.py:327-333, 402-407): the reference drew noise / timesteps, noised the bf16 actions through `add_noise`, ran the VLM
    with the timestep token and the projected noisy actions in the action slots (FiLM on) and took the MSE of the bf16 noise prediction."""
    g = load("g21_ref_run_forward_pass.npz")
    sd = _ref_sd(g, diffusion=True)
    noise, ts = torch.from_numpy(g["diffusion.noise"]), torch.from_numpy(g["diffusion.timesteps"])
    ddim = vo.DDIM(int(g["diffusion.T"]))
    gt16 = torch.from_numpy(g["actions"]).to(torch.bfloat16).float()
    noisy = ddim.add_noise(gt16, noise, ts).to(torch.bfloat16).float()
    assert np.array_equal(noisy.numpy(), g["diffusion.noisy_actions"]), "x_t = add_noise(actions, noise, t) on the recorded draws"
    cfg, b, o32, o16, ah16, P = _g21_parts(g, "bidirectional", sd, diffusion=dict(noisy=noisy, timesteps=ts))
    with torch.no_grad():
        eps = o16.noise_head(ah16).reshape(noise.shape)
        loss = float(o16.R(((eps - noise) ** 2).mean()))
    ref = float(g["diffusion.loss"])
    assert abs(loss - ref) <= 1.5e-2 * ref, (loss, ref)
