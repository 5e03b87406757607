"""Generates G16-G19: golden vectors produced by the REFERENCE's own hot-path files, executed in the build container:

    prismatic/extern/hf/modeling_prismatic.py   (PrismaticProjector, PrismaticVisionBackbone.forward, the multimodal helpers,
                                                 PrismaticForConditionalGeneration.forward, OpenVLAForActionPrediction.predict_action)
    prismatic/models/film_vit_wrapper.py        (FiLMedVisionTransformerBlock, FiLMedVisionTransformer, FiLMedPrismaticVisionBackbone)

Both files import `timm`, which is absent here.  They are loaded by path (as make_golden.py loads the leaf modules) after registering a
NAMES-ONLY `timm` module: `timm.__version__` and two empty classes `timm.models.vision_transformer.{LayerScale, VisionTransformer}`.  No code
path exercised below calls into timm: the two towers are duck-typed torch modules defined HERE (`DuckViT`: patch embedding, position
embedding + prefix tokens, pre-norm blocks; the same architecture the oracle's `vit` / `vit_block` restate and G14 / G15 pin against
transformers' DINOv2 / SigLIP) and handed to the reference's classes, whose own code then does everything this file is meant to pin:

    * which block's output is taken, prefix tokens dropped, no final norm (film_vit_wrapper.py:114-168: the reference's own copy of timm's
      `get_intermediate_layers`), FiLM placement and formula (:56-77), the language average (:243), the image / feature concat order
      (:231-276 and modeling_prismatic.py:186-227), LayerScale patching (:57-65);
    * projector (:231-262); action masks, zeroed / replaced action embeddings, multimodal concat of embeddings / mask / labels, proprio token,
      diffusion timestep token (:395-496, :571-643); the placeholder / label builders, slicing of the action rows, L1 / discrete decode and
      un-normalisation of predict_action (:734-791, :879-1060).

The language model is stock transformers 5.15 `LlamaForCausalLM` (the reference's fork is un-vendored): "causal" = stock behaviour with the
2-D mask the reference passes; "bidirectional" = the same model behind a shim that turns that 2-D key mask into an all-zero additive 4-D mask
(the construction G5 uses).  `PrismaticVisionBackbone` / `OpenVLAForActionPrediction` instances are assembled with `__new__` + attribute
assignment because their constructors call `timm.create_model` / need a hub config; every METHOD executed on them is the reference's.
The DDIM scheduler inside the diffusion predict_action fixture is the oracle's own (diffusers is absent): that fixture pins the reference's
LOOP WIRING around it (timestep token, embedding replacement, slicing), not the scheduler arithmetic -- DDIM itself stays parity-unpinned.

Weights are the oracle's seeded `random_state_dict` of the tiny config (numbers only); a checksum is stored.  Only data is written.

    python tests/golden/make_golden_ref_model.py
"""
import importlib.util
import sys
import types
from functools import partial
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import transformers  # noqa: F401  (must be imported before the reference files: they take PretrainedConfig & co from it)
from transformers import LlamaConfig, LlamaForCausalLM

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import vla_oracle as vo  # noqa: E402

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


# ----------------------------------------------------------------------------------------------------------------------
# loading the reference files
# ----------------------------------------------------------------------------------------------------------------------
def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    for pk in ["prismatic", "prismatic.vla", "prismatic.models", "prismatic.training", "prismatic.util", "prismatic.extern", "prismatic.extern.hf"]:
        m = types.ModuleType(pk)
        m.__path__ = []
        sys.modules[pk] = m
    timm = types.ModuleType("timm")                        # names only: nothing below calls into it
    timm.__version__ = "0.9.10"
    tm, tv = types.ModuleType("timm.models"), types.ModuleType("timm.models.vision_transformer")
    tv.LayerScale = type("LayerScale", (nn.Module,), {})
    tv.VisionTransformer = type("VisionTransformer", (nn.Module,), {})
    timm.models, tm.vision_transformer = tm, tv
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.vision_transformer": tv})
    dd = types.ModuleType("diffusers.schedulers.scheduling_ddim")
    dd.DDIMScheduler = type("DDIMScheduler", (), {})
    sys.modules.update({"diffusers": types.ModuleType("diffusers"), "diffusers.schedulers": types.ModuleType("diffusers.schedulers"),
                        "diffusers.schedulers.scheduling_ddim": dd})
    r = types.SimpleNamespace(timm_vt=tv)
    r.constants = _load("prismatic.vla.constants", "prismatic/vla/constants.py")
    r.train_utils = _load("prismatic.training.train_utils", "prismatic/training/train_utils.py")
    r.projectors = _load("prismatic.models.projectors", "prismatic/models/projectors.py")
    r.action_heads = _load("prismatic.models.action_heads", "prismatic/models/action_heads.py")
    _load("prismatic.extern.hf.configuration_prismatic", "prismatic/extern/hf/configuration_prismatic.py")
    r.mp = _load("prismatic.extern.hf.modeling_prismatic", "prismatic/extern/hf/modeling_prismatic.py")
    r.film = _load("prismatic.models.film_vit_wrapper", "prismatic/models/film_vit_wrapper.py")
    assert r.constants.ACTION_DIM == 7 and r.constants.NUM_ACTIONS_CHUNK == 8
    return r


# ----------------------------------------------------------------------------------------------------------------------
# duck-typed towers (NOT reference code: the leaf arithmetic of a timm ViT, see the module docstring)
# ----------------------------------------------------------------------------------------------------------------------
def P(t):
    return nn.Parameter(t.clone().float())


def linear_from(sd, name):
    w = sd[name + ".weight"]
    lin = nn.Linear(w.shape[1], w.shape[0], bias=name + ".bias" in sd)
    lin.weight = P(w)
    if lin.bias is not None:
        lin.bias = P(sd[name + ".bias"])
    return lin


def ln_from(sd, name, eps):
    m = nn.LayerNorm(sd[name + ".weight"].shape[0], eps=eps)
    m.weight, m.bias = P(sd[name + ".weight"]), P(sd[name + ".bias"])
    return m


class DuckAttention(nn.Module):
    def __init__(self, sd, p, heads):
        super().__init__()
        self.heads, self.qkv, self.proj = heads, linear_from(sd, p + "qkv"), linear_from(sd, p + "proj")

    def forward(self, x):
        B, T, C = x.shape
        qkv = self.qkv(x).reshape(B, T, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        return self.proj(a.transpose(1, 2).reshape(B, T, C))


class DuckMlp(nn.Module):
    def __init__(self, sd, p):
        super().__init__()
        self.fc1, self.act, self.fc2 = linear_from(sd, p + "fc1"), nn.GELU(), linear_from(sd, p + "fc2")

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


def make_duck_layerscale(ref, gamma):
    """A LayerScale in timm's shape (`gamma`, `inplace`); the reference's `ls_apply_patch` (modeling_prismatic.py:57-65) renames it."""
    ls = ref.timm_vt.LayerScale()
    ls.gamma, ls.inplace = P(gamma), False
    return ls


class DuckBlock(nn.Module):
    def __init__(self, ref, sd, p, vc):
        super().__init__()
        self.norm1, self.attn = ln_from(sd, p + "norm1", vc.eps), DuckAttention(sd, p + "attn.", vc.heads)
        self.ls1 = make_duck_layerscale(ref, sd[p + "ls1.scale_factor"]) if vc.layerscale else nn.Identity()
        self.drop_path1 = nn.Identity()
        self.norm2, self.mlp = ln_from(sd, p + "norm2", vc.eps), DuckMlp(sd, p + "mlp.")
        self.ls2 = make_duck_layerscale(ref, sd[p + "ls2.scale_factor"]) if vc.layerscale else nn.Identity()
        self.drop_path2 = nn.Identity()

    def forward(self, x):
        x = x + self.drop_path1(self.ls1(self.attn(self.norm1(x))))
        return x + self.drop_path2(self.ls2(self.mlp(self.norm2(x))))


class DuckPatchEmbed(nn.Module):
    def __init__(self, sd, p, vc):
        super().__init__()
        self.proj = nn.Conv2d(3, vc.dim, vc.patch, stride=vc.patch)
        self.proj.weight, self.proj.bias = P(sd[p + "proj.weight"]), P(sd[p + "proj.bias"])
        self.num_patches, self.grid_size = vc.n_patches, (vc.image_size // vc.patch,) * 2

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class DuckPosEmbed(nn.Module):
    """`_pos_embed` of a `no_embed_class` ViT: position embedding on the patches, then [cls, registers] in front."""

    def __init__(self, sd, p):
        super().__init__()
        self.pos_embed = P(sd[p + "pos_embed"])
        self.cls_token = P(sd[p + "cls_token"]) if p + "cls_token" in sd else None
        self.reg_token = P(sd[p + "reg_token"]) if p + "reg_token" in sd else None

    def forward(self, x):
        x = x + self.pos_embed
        toks = [t.expand(x.shape[0], -1, -1) for t in (self.cls_token, self.reg_token) if t is not None]
        return torch.cat(toks + [x], dim=1) if toks else x


def make_duck_vit(ref, sd, prefix, vc):
    class DuckViT(ref.timm_vt.VisionTransformer):
        def get_intermediate_layers(self, x, n=1, reshape=False, return_prefix_tokens=False, norm=False):
            """The non-FiLM `get_intermediate_layers`: film_vit_wrapper.py:114-168 (the reference's copy of timm's) minus the language argument."""
            take = set(range(len(self.blocks) - n, len(self.blocks)) if isinstance(n, int) else n)
            outs = []
            x = self.norm_pre(self.patch_drop(self._pos_embed(self.patch_embed(x))))
            for i, blk in enumerate(self.blocks):
                x = blk(x)
                if i in take:
                    outs.append(x)
            return tuple(o[:, self.num_prefix_tokens:] for o in outs)

    v = DuckViT()
    v.patch_embed, v._pos_embed = DuckPatchEmbed(sd, prefix + "patch_embed.", vc), DuckPosEmbed(sd, prefix)
    v.patch_drop, v.norm_pre, v.norm = nn.Identity(), nn.Identity(), nn.Identity()
    v.blocks = nn.Sequential(*[DuckBlock(ref, sd, f"{prefix}blocks.{i}.", vc) for i in range(vc.depth)])
    v.num_prefix_tokens, v.num_features, v.embed_dim = vc.n_prefix, vc.dim, vc.dim
    return v


def make_backbone(ref, sd, cfg, num_images, film):
    """PrismaticVisionBackbone with the two duck towers ([0:3] -> featurizer = DINOv2 shape, [3:6] -> fused_featurizer = SigLIP shape);
    with `film`, wrapped by the reference's FiLMedPrismaticVisionBackbone and its scale / shift Linears loaded from `sd`."""
    mp = ref.mp
    vb = mp.PrismaticVisionBackbone.__new__(mp.PrismaticVisionBackbone)
    nn.Module.__init__(vb)
    vb.use_fused_vision_backbone, vb.num_images_in_input = True, 1
    vb.featurizer = make_duck_vit(ref, sd, "vision_backbone.featurizer.", cfg.dino)
    vb.fused_featurizer = make_duck_vit(ref, sd, "vision_backbone.fused_featurizer.", cfg.siglip)
    vb.embed_dim = cfg.dino.dim + cfg.siglip.dim
    for f in (vb.featurizer, vb.fused_featurizer):      # what _create_featurizer does after timm.create_model (modeling_prismatic.py:133-137)
        f.forward = mp.unpack_tuple(partial(f.get_intermediate_layers, n={len(f.blocks) - 2}))
    vb._patch_layer_scales()                            # the reference's own (:139-156)
    vb.set_num_images_in_input(num_images)
    if not film:
        return vb
    fb = ref.film.FiLMedPrismaticVisionBackbone(vb, llm_dim=cfg.llm_dim)   # wraps every block, swaps the class, patches forward
    for prefix, vit in (("vision_backbone.featurizer.", vb.featurizer), ("vision_backbone.fused_featurizer.", vb.fused_featurizer)):
        for i, blk in enumerate(vit.blocks):
            for nm in ("scale", "shift"):
                lin = getattr(blk, nm)
                lin.weight, lin.bias = P(sd[f"{prefix}blocks.{i}.{nm}.weight"]), P(sd[f"{prefix}blocks.{i}.{nm}.bias"])
    return fb


class BidirectionalLM(nn.Module):
    """Stock LlamaForCausalLM behind the 2-D -> additive 4-D all-zero mask conversion of G5 (stand-in for the un-vendored fork)."""

    def __init__(self, lm):
        super().__init__()
        self.lm = lm

    def get_input_embeddings(self):
        return self.lm.get_input_embeddings()

    def forward(self, attention_mask=None, inputs_embeds=None, **kw):
        B, S = inputs_embeds.shape[:2]
        add = torch.zeros(B, 1, S, S, dtype=inputs_embeds.dtype)
        if attention_mask is not None:
            add = add.masked_fill(~attention_mask.bool()[:, None, None, :], torch.finfo(inputs_embeds.dtype).min)
        return self.lm(attention_mask=add, inputs_embeds=inputs_embeds, **kw)


def make_llama(sd, cfg):
    hc = LlamaConfig(hidden_size=cfg.llm_dim, intermediate_size=cfg.llm_ff, num_hidden_layers=cfg.llm_layers, num_attention_heads=cfg.llm_heads,
                     num_key_value_heads=cfg.llm_heads, vocab_size=cfg.vocab, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                     attention_bias=False, mlp_bias=False, tie_word_embeddings=False)
    lm = LlamaForCausalLM(hc).eval()
    missing = lm.load_state_dict({k[len("language_model."):]: v.float() for k, v in sd.items() if k.startswith("language_model.")}, strict=False)
    assert not [k for k in missing.missing_keys if "rotary" not in k] and not missing.unexpected_keys, missing
    return lm


def make_vla(ref, sd, cfg, num_images, film, mask_mode, norm_stats=None):
    mp = ref.mp
    m = mp.OpenVLAForActionPrediction.__new__(mp.OpenVLAForActionPrediction)
    nn.Module.__init__(m)
    m.config = types.SimpleNamespace(output_attentions=False, output_hidden_states=False, use_return_dict=True)
    m.vision_backbone = make_backbone(ref, sd, cfg, num_images, film)
    m.projector = mp.PrismaticProjector(True, cfg.dino.dim + cfg.siglip.dim, cfg.llm_dim)
    for n in ("fc1", "fc2", "fc3"):
        getattr(m.projector, n).weight, getattr(m.projector, n).bias = P(sd[f"projector.{n}.weight"]), P(sd[f"projector.{n}.bias"])
    lm = make_llama(sd, cfg)
    m.language_model = BidirectionalLM(lm) if mask_mode == "bidirectional" else lm
    m.llm_dim, m.pad_token_id, m.norm_stats = cfg.llm_dim, 32000, norm_stats
    m.bins = np.linspace(-1, 1, cfg.n_action_bins)                                    # OpenVLAForActionPrediction.__init__ :725-732
    m.bin_centers = (m.bins[:-1] + m.bins[1:]) / 2.0
    m.vocab_size = cfg.vocab - cfg.pad_to_multiple_of
    return m.eval()


def load_mlp(mod, sd, prefix):
    for n in ("fc1", "fc2"):
        getattr(mod, n).weight, getattr(mod, n).bias = P(sd[f"{prefix}{n}.weight"]), P(sd[f"{prefix}{n}.bias"])
    return mod


def load_head(head, sd, prefix):
    t = {k[len(prefix):]: v.float() for k, v in sd.items() if k.startswith(prefix)}
    head.load_state_dict(t)
    return head.eval()


def sd_checksum(sd):
    return np.float64(sum(float(v.double().abs().sum()) for v in sd.values()))


def ragged_batch(seed, prompt_lens, num_images, size=56):
    """Collator-shaped batch, right padded with 32000 (the layout G10 pins against the reference's collator)."""
    rng = np.random.default_rng(seed)
    B, L = len(prompt_lens), max(prompt_lens) + 56 + 1
    ids = np.full((B, L), 32000, np.int64); labels = np.full((B, L), -100, np.int64); mask = np.zeros((B, L), bool)
    actions = rng.uniform(-1, 1, (B, 8, 7)).astype(np.float32)
    for b, tp in enumerate(prompt_lens):
        row = np.concatenate([[1], rng.integers(3, 31000, tp - 2), [29871], vo.tokenize_actions(actions[b]).reshape(-1), [2]])
        ids[b, : len(row)] = row; mask[b, : len(row)] = True
        labels[b, tp: len(row)] = row[tp:]
    return dict(input_ids=ids, attention_mask=mask, labels=labels, actions=actions,
                pixel_values=rng.standard_normal((B, 6 * num_images, size, size)).astype(np.float32),
                proprio=rng.uniform(-1, 1, (B, 8)).astype(np.float32))


LOGIT_COLS = slice(31700, 32064)   # the action-token band + the pad band: the columns the fixtures keep of the (B, S, 32064) logits


def lm_outputs(out, tag, band=False, feats=False):
    lg = out.logits.detach().float()
    d = {f"{tag}.hidden": out.hidden_states[-1].detach().numpy(), f"{tag}.logits_argmax": lg.argmax(-1).numpy(),
         f"{tag}.logits_lse": torch.logsumexp(lg, -1).numpy()}
    if band:
        d[f"{tag}.logits_band"] = lg[..., LOGIT_COLS].numpy()
    if feats:
        d[f"{tag}.projector_features"] = out.projector_features.detach().numpy()
    return d


def main():
    torch.manual_seed(0)
    ref = load_reference()
    mp = ref.mp
    cfg = vo.tiny_config()
    SEED = 7
    sd = vo.random_state_dict(cfg, seed=SEED, lora=False, film=True, diffusion=False)
    sd_diff = vo.random_state_dict(cfg, seed=SEED, lora=False, film=True, diffusion=True)
    meta = dict(sd_seed=np.int64(SEED), sd_checksum=sd_checksum(sd), sd_diffusion_checksum=sd_checksum(sd_diff))

    # ==== G16: PrismaticProjector forward + every gradient (modeling_prismatic.py:231-262) =====================================
    g = torch.Generator().manual_seed(16)
    proj = mp.PrismaticProjector(True, 48, 40)          # fused-backbone branch: 48 -> 192 -> 40 -> 40, its own seeded weights (stored)
    with torch.no_grad():
        for p_ in proj.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.15 if p_.dim() > 1 else 0.3))
    x = torch.randn(2, 32, 48, generator=g).requires_grad_(True)
    dy = torch.randn(2, 32, 40, generator=g)
    y = proj(x)
    y.backward(dy)
    g16 = dict(x=x.detach().numpy(), dy=dy.numpy(), y=y.detach().numpy(), dx=x.grad.numpy())
    g16.update({f"projector.{n}": p_.detach().numpy() for n, p_ in proj.named_parameters()})
    g16.update({f"grad.{n}": p_.grad.numpy() for n, p_ in proj.named_parameters()})
    np.savez(OUT / "g16_ref_projector.npz", **g16, **meta)

    # ==== G17: the multimodal helpers on ragged rows (modeling_prismatic.py:395-496, 734-791) ===================================
    vla = make_vla(ref, sd, cfg, 2, False, "causal")
    b = ragged_batch(17, (9, 7, 12, 3), 2)
    ids, labels, amask = (torch.from_numpy(b[k]) for k in ("input_ids", "labels", "attention_mask"))
    D = 24
    emb, patches = torch.randn(4, ids.shape[1], D, generator=g), torch.randn(4, 5, D, generator=g)
    feats = torch.randn(4, 56, D, generator=g)
    all_mask = vla._process_action_masks(labels)
    mm_emb, mm_mask = vla._build_multimodal_attention(emb, patches, amask)
    pp = load_mlp(ref.projectors.ProprioProjector(llm_dim=D, proprio_dim=8), {"p.fc1.weight": torch.randn(D, 8, generator=g), "p.fc1.bias": torch.randn(D, generator=g),
                                                                           "p.fc2.weight": torch.randn(D, D, generator=g) * 0.2, "p.fc2.bias": torch.randn(D, generator=g)}, "p.")
    prop = torch.from_numpy(b["proprio"])
    p_ids, p_mask = torch.from_numpy(b["input_ids"][:1, :9]), torch.ones(1, 9, dtype=torch.bool)
    in2, mask2 = vla._prepare_input_for_action_prediction(p_ids, p_mask)
    lab2 = vla._prepare_labels_for_action_prediction(torch.full_like(p_ids, -100), in2)
    norm = np.random.default_rng(170).uniform(-1, 1, (8, 7))
    stats = {"libero": {"action": {"q01": [-0.9, -0.5, -0.3, -0.1, -0.2, -0.4, 0.0], "q99": [0.9, 0.7, 0.3, 0.2, 0.5, 0.4, 1.0],
                                   "min": [-1.0] * 7, "max": [1.5, 1.0, 1.0, 0.5, 0.5, 0.5, 1.0], "mask": [True] * 6 + [False]}}}
    vla.norm_stats = stats
    un_q99 = vla._unnormalize_actions(norm, "libero")
    mp.ACTION_PROPRIO_NORMALIZATION_TYPE = mp.NormalizationType.BOUNDS          # what constants.py selects for ALOHA (constants.py:26-52)
    un_bounds = vla._unnormalize_actions(norm, "libero")
    vla.norm_stats = {"aloha": {"action": {k: v for k, v in stats["libero"]["action"].items() if k != "mask"}}}
    un_bounds_nomask = vla._unnormalize_actions(norm, None)
    mp.ACTION_PROPRIO_NORMALIZATION_TYPE = mp.NormalizationType.BOUNDS_Q99
    st = stats["libero"]["action"]
    np.savez(OUT / "g17_ref_multimodal_helpers.npz", input_ids=b["input_ids"], labels=b["labels"], attention_mask=b["attention_mask"],
             all_actions_mask=all_mask.numpy(), emb=emb.numpy(), patches=patches.numpy(), noisy_features=feats.numpy(),
             replaced=vla._replace_input_embeddings(emb, all_mask, feats).numpy(), mm_emb=mm_emb.numpy(), mm_mask=mm_mask.numpy(),
             mm_labels=vla._build_multimodal_labels(labels, patches).numpy(),
             proprio=b["proprio"], **{f"pp.{k}": v.detach().numpy() for k, v in pp.state_dict().items()},
             with_proprio=vla._process_proprio_features(patches, prop, pp).detach().numpy(),
             prompt_ids=p_ids.numpy(), prepared_ids=in2.numpy(), prepared_mask=mask2.numpy(), prepared_labels=lab2.numpy(),
             normalized=norm, unnorm_q99=un_q99, unnorm_bounds=un_bounds, unnorm_bounds_nomask=un_bounds_nomask,
             **{f"stats.{k}": np.asarray(v) for k, v in st.items()})

    # ==== G18: FiLM block and FiLM / plain vision backbones (film_vit_wrapper.py:56-77, 108-276; modeling_prismatic.py:186-227) ============
    g18 = {}
    for tag, prefix, vc in (("dino", "vision_backbone.featurizer.", cfg.dino), ("siglip", "vision_backbone.fused_featurizer.", cfg.siglip)):
        blk = DuckBlock(ref, sd, prefix + "blocks.1.", vc)
        for m_ in blk.modules():
            if isinstance(m_, ref.timm_vt.LayerScale):
                mp.ls_apply_patch(m_)
        fblk = ref.film.FiLMedVisionTransformerBlock(blk, vc.dim, cfg.llm_dim)
        for nm in ("scale", "shift"):
            getattr(fblk, nm).weight, getattr(fblk, nm).bias = P(sd[f"{prefix}blocks.1.{nm}.weight"]), P(sd[f"{prefix}blocks.1.{nm}.bias"])
        x = torch.randn(2, vc.n_prefix + 16, vc.dim, generator=g).requires_grad_(True)
        avg = (torch.randn(2, cfg.llm_dim, generator=g) * 0.5).requires_grad_(True)
        dy = torch.randn(2, vc.n_prefix + 16, vc.dim, generator=g)
        y = fblk(x, avg)
        y.backward(dy)
        g18.update({f"{tag}.x": x.detach().numpy(), f"{tag}.avg": avg.detach().numpy(), f"{tag}.dy": dy.numpy(), f"{tag}.y": y.detach().numpy(),
                    f"{tag}.dx": x.grad.numpy(), f"{tag}.davg": avg.grad.numpy()})
        for nm in ("scale", "shift"):
            g18[f"{tag}.grad.{nm}.weight"], g18[f"{tag}.grad.{nm}.bias"] = getattr(fblk, nm).weight.grad.numpy(), getattr(fblk, nm).bias.grad.numpy()
    for n_img in (1, 2, 3):
        nb = 2 if n_img == 2 else 1
        pv = torch.randn(nb, 6 * n_img, 56, 56, generator=g)
        lang = torch.randn(nb, 11, cfg.llm_dim, generator=g) * 0.5
        with torch.no_grad():
            g18[f"backbone.i{n_img}.pixel_values"], g18[f"backbone.i{n_img}.language"] = pv.numpy(), lang.numpy()
            g18[f"backbone.i{n_img}.plain"] = make_backbone(ref, sd, cfg, n_img, False)(pv).numpy()
            g18[f"backbone.i{n_img}.film"] = make_backbone(ref, sd, cfg, n_img, True)(pv, lang).numpy()
    np.savez(OUT / "g18_ref_film_backbone.npz", **g18, **meta)

    # ==== G19: PrismaticForConditionalGeneration.forward (multimodal branch) and predict_action ==========================================
    b = ragged_batch(19, (9, 7, 12), 2)
    tb = {k: torch.from_numpy(v) for k, v in b.items()}
    g19 = dict(b)
    ppj = load_mlp(ref.projectors.ProprioProjector(cfg.llm_dim, cfg.proprio_dim), sd, "proprio_projector.")
    napj = load_mlp(ref.projectors.NoisyActionProjector(cfg.llm_dim), sd_diff, "noisy_action_projector.")
    l1 = load_head(ref.action_heads.L1RegressionActionHead(cfg.llm_dim, cfg.llm_dim, cfg.action_dim), sd, "action_head.")
    noisy = torch.randn(3, 8, 7, generator=g)
    tsteps = torch.tensor([3.0, 41.0, 17.0])
    temb = ref.action_heads.SinusoidalPositionalEncoding(cfg.llm_dim)(tsteps).unsqueeze(1)
    g19.update(noisy_actions=noisy.numpy(), timesteps=tsteps.numpy())
    p_ids = torch.from_numpy(b["input_ids"][:1, :9])                      # ends with 29871
    p_ids_no = torch.from_numpy(b["input_ids"][1:2, :6])                  # does not: predict_action appends it (:974-977)
    pv1, prop1 = tb["pixel_values"][:1], b["proprio"][0]
    stats = {"libero": {"action": {"q01": [-0.9, -0.5, -0.3, -0.1, -0.2, -0.4, 0.0], "q99": [0.9, 0.7, 0.3, 0.2, 0.5, 0.4, 1.0], "mask": [True] * 6 + [False]}}}
    g19.update(prompt_ids=p_ids.numpy(), prompt_ids_no_empty=p_ids_no.numpy(), **{f"stats.{k}": np.asarray(v) for k, v in stats["libero"]["action"].items()})
    for mode in ("causal", "bidirectional"):
        for film in (False, True):
            vla = make_vla(ref, sd, cfg, 2, film, mode, norm_stats=stats)
            tag = f"{mode}.{'film' if film else 'plain'}"
            with torch.no_grad():
                out = vla(input_ids=tb["input_ids"], attention_mask=tb["attention_mask"], pixel_values=tb["pixel_values"], labels=tb["labels"],
                          output_hidden_states=True, proprio=tb["proprio"], proprio_projector=ppj, use_film=film)
                g19.update(lm_outputs(out, tag + ".l1", band=not film, feats=mode == "causal"))
                g19[tag + ".l1.loss"] = np.float64(out.loss.item())
                out = vla(input_ids=tb["input_ids"], attention_mask=tb["attention_mask"], pixel_values=tb["pixel_values"], labels=tb["labels"],
                          output_hidden_states=True, proprio=tb["proprio"], proprio_projector=ppj, noisy_actions=noisy,
                          noisy_action_projector=napj, diffusion_timestep_embeddings=temb, use_film=film)
                g19.update(lm_outputs(out, tag + ".diffusion", feats=mode == "causal" and film))
                # predict_action: L1 head and discrete decode (:879-944), both prompt forms
                for ptag, pid in (("p", p_ids), ("pno", p_ids_no)):
                    am = torch.ones_like(pid, dtype=torch.bool)
                    act, ah = vla.predict_action(input_ids=pid, unnorm_key="libero", proprio=prop1, proprio_projector=ppj, action_head=l1,
                                                 use_film=film, pixel_values=pv1, attention_mask=am)
                    g19[f"{tag}.predict.{ptag}.l1.actions"], g19[f"{tag}.predict.{ptag}.l1.hidden"] = np.asarray(act), ah.numpy()
                    act, ah = vla.predict_action(input_ids=pid, unnorm_key="libero", proprio=prop1, proprio_projector=ppj, action_head=None,
                                                 use_film=film, pixel_values=pv1, attention_mask=am)
                    g19[f"{tag}.predict.{ptag}.discrete.actions"] = np.asarray(act)
    # diffusion predict_action: the reference's denoising loop (:793-877) around the oracle's DDIM (see the module docstring)
    T = 5
    sched = vo.DDIM(T)

    class Sched:
        timesteps = None

        def set_timesteps(self, n):
            sched.set_timesteps(n)
            self.timesteps = sched.timesteps

        def step(self, eps, t, x):
            return types.SimpleNamespace(prev_sample=sched.step(eps, int(t), x))

    ah_mod = ref.action_heads
    dhead = ah_mod.DiffusionActionHead.__new__(ah_mod.DiffusionActionHead)        # __init__ needs diffusers; every method used below is the reference's
    nn.Module.__init__(dhead)
    dhead.action_dim, dhead.num_diffusion_steps, dhead.noise_scheduler = cfg.action_dim, T, Sched()
    dhead.time_encoder = ah_mod.SinusoidalPositionalEncoding(dim=cfg.llm_dim)
    dhead.noise_predictor = ah_mod.NoisePredictionModel(transformer_hidden_dim=cfg.llm_dim * cfg.action_dim, hidden_dim=cfg.llm_dim, action_dim=cfg.action_dim)
    load_head(dhead.noise_predictor, sd_diff, "action_head.noise_predictor.")
    for mode in ("causal", "bidirectional"):
        vla = make_vla(ref, sd_diff, cfg, 2, True, mode, norm_stats=stats)
        torch.manual_seed(191)
        noise = torch.randn(size=(1, 8, 7))
        torch.manual_seed(191)                                                     # predict_action draws the same start noise (:1027-1029)
        with torch.no_grad():
            act, ah = vla.predict_action(input_ids=p_ids, unnorm_key="libero", proprio=prop1, proprio_projector=ppj, action_head=dhead,
                                         noisy_action_projector=napj, use_film=True, pixel_values=pv1, attention_mask=torch.ones_like(p_ids, dtype=torch.bool))
        g19[f"{mode}.film.predict.p.diffusion.actions"], g19[f"{mode}.film.predict.p.diffusion.hidden"] = np.asarray(act), ah.numpy()
        g19["diffusion.start_noise"], g19["diffusion.T"] = noise.numpy(), np.int64(T)
    np.savez_compressed(OUT / "g19_ref_forward_predict.npz", **g19, **meta)
    print("wrote g16-g19:", {p_.name: p_.stat().st_size for p_ in sorted(OUT.glob("g1[6-9]_*.npz"))})


if __name__ == "__main__":
    main()
