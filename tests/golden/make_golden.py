"""Generates tests/golden/*.npz: input/output vectors produced by the REFERENCE's own importable modules (loaded by
path from /root/reference, which exists only in the build container) and by stock transformers' LlamaForCausalLM
(stand-in for the un-vendored transformers fork).  Only data is written; no reference source is copied.

    python tests/golden/make_golden.py

Groups:
  G1  action masks                      <- prismatic/training/train_utils.py
  G2  L1RegressionActionHead fwd/bwd    <- prismatic/models/action_heads.py
  G3  projectors + timestep encoding    <- prismatic/models/projectors.py, action_heads.py
  G4  action tokenizer encode/decode    <- prismatic/vla/action_tokenizer.py
  G5  Llama decoder stack (2 mask modes)<- transformers.LlamaForCausalLM (5.15, stock)
  G10 collator batch layout             <- prismatic/util/data_utils.py
"""
import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    for pk in ["prismatic", "prismatic.vla", "prismatic.models", "prismatic.training", "prismatic.util"]:
        m = types.ModuleType(pk)
        m.__path__ = []
        sys.modules[pk] = m
    # the diffusers import in action_heads.py is only needed by DiffusionActionHead.__init__, which we never construct
    dd = types.ModuleType("diffusers.schedulers.scheduling_ddim")
    dd.DDIMScheduler = type("DDIMScheduler", (), {})
    sys.modules.update({"diffusers": types.ModuleType("diffusers"), "diffusers.schedulers": types.ModuleType("diffusers.schedulers"),
                        "diffusers.schedulers.scheduling_ddim": dd})
    mods = types.SimpleNamespace()
    mods.constants = _load("prismatic.vla.constants", "prismatic/vla/constants.py")
    mods.train_utils = _load("prismatic.training.train_utils", "prismatic/training/train_utils.py")
    mods.projectors = _load("prismatic.models.projectors", "prismatic/models/projectors.py")
    mods.action_heads = _load("prismatic.models.action_heads", "prismatic/models/action_heads.py")
    mods.action_tokenizer = _load("prismatic.vla.action_tokenizer", "prismatic/vla/action_tokenizer.py")
    mods.data_utils = _load("prismatic.util.data_utils", "prismatic/util/data_utils.py")
    return mods


def sd_np(module, prefix=""):
    return {prefix + k: v.detach().numpy() for k, v in module.state_dict().items()}


def label_rows(rng, B, tps, A, pad_to):
    ids = rng.integers(3, 31000, size=(B, pad_to)).astype(np.int64)
    labels = np.full((B, pad_to), -100, dtype=np.int64)
    for b, tp in enumerate(tps):
        ids[b, tp: tp + A] = rng.integers(31744, 32000, size=A)
        ids[b, tp + A] = 2
        ids[b, tp + A + 1:] = 32000
        labels[b, tp: tp + A + 1] = ids[b, tp: tp + A + 1]
    return ids, labels


def main():
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    ref = load_reference()
    assert ref.constants.ACTION_DIM == 7 and ref.constants.NUM_ACTIONS_CHUNK == 8  # LIBERO defaults

    # ---- G1 masks -------------------------------------------------------------------------------------------------------
    ids, labels = label_rows(rng, 5, [38, 34, 36, 40, 3], 56, 100)
    lt = torch.from_numpy(labels)
    np.savez(OUT / "g1_masks.npz", labels=labels, ids=ids,
             current=ref.train_utils.get_current_action_mask(lt).numpy(), next=ref.train_utils.get_next_actions_mask(lt).numpy(),
             current_shift=ref.train_utils.get_current_action_mask(lt[:, 1:]).numpy(),
             next_shift=ref.train_utils.get_next_actions_mask(lt[:, 1:]).numpy())

    # ---- G2 L1 head fwd/bwd (reduced dims: input_dim = hidden = 64) ---------------------------------------------------------
    head = ref.action_heads.L1RegressionActionHead(input_dim=64, hidden_dim=64, action_dim=7)
    for p in head.parameters():
        torch.nn.init.normal_(p, std=0.2) if p.dim() > 1 else torch.nn.init.normal_(p, mean=0.5, std=0.3)
    x = torch.randn(3, 56, 64, requires_grad=True)
    gt = torch.rand(3, 8, 7) * 2 - 1
    pred = head.predict_action(x)
    loss = torch.nn.L1Loss()(gt, pred)
    loss.backward()
    g2 = sd_np(head, "action_head.")
    g2.update({"grad." + k: v.grad.numpy() for k, v in head.named_parameters()})
    np.savez(OUT / "g2_l1_head.npz", x=x.detach().numpy(), gt=gt.numpy(), pred=pred.detach().numpy(), loss=loss.detach().numpy(),
             dx=x.grad.numpy(), **g2)

    # ---- G3 projectors, time encoder -----------------------------------------------------------------------------------
    pp = ref.projectors.ProprioProjector(llm_dim=64, proprio_dim=8)
    nap = ref.projectors.NoisyActionProjector(llm_dim=64)
    te = ref.action_heads.SinusoidalPositionalEncoding(dim=64)
    prop = torch.randn(4, 8)
    noisy = torch.randn(2, 56, 1)
    ts = torch.tensor([0.0, 7.0, 49.0, 23.0])
    g3 = sd_np(pp, "proprio_projector.")
    g3.update(sd_np(nap, "noisy_action_projector."))
    np.savez(OUT / "g3_projectors.npz", proprio=prop.numpy(), proprio_out=pp(prop).detach().numpy(), noisy=noisy.numpy(),
             noisy_out=nap(noisy).detach().numpy(), timesteps=ts.numpy(), time_emb=te(ts).numpy(), **g3)

    # ---- G4 action tokenizer -------------------------------------------------------------------------------------------
    class Tok:  # the reference only touches .vocab_size, .decode and .batch_decode
        vocab_size = 32000

        def decode(self, ids):
            return list(ids)

        def batch_decode(self, ids):
            return ids

    at = ref.action_tokenizer.ActionTokenizer(Tok())
    acts = np.concatenate([rng.uniform(-1.3, 1.3, size=200), np.array([-1.0, 1.0, 0.0, -0.999999, 0.999999]), at.bins[:5], at.bins[-5:]])
    enc = np.array(at(acts), dtype=np.int64)
    all_ids = np.arange(31700, 32064, dtype=np.int64)
    np.savez(OUT / "g4_action_tokenizer.npz", actions=acts, token_ids=enc, all_ids=all_ids,
             decoded=at.decode_token_ids_to_actions(all_ids), begin_idx=np.int64(at.action_token_begin_idx))

    # ---- G5 Llama stack (stock HF) --------------------------------------------------------------------------------------
    from transformers import LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                      vocab_size=320, rms_norm_eps=1e-5, rope_theta=10000.0, attention_bias=False, mlp_bias=False)
    m = LlamaForCausalLM(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "norm" in n:
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            else:
                p.normal_(std=0.05)
    B, S = 2, 45
    emb = torch.randn(B, S, 128)
    mask2d = torch.ones(B, S, dtype=torch.long)
    mask2d[1, 30:] = 0  # right padding
    with torch.no_grad():
        causal = m(inputs_embeds=emb, attention_mask=mask2d, output_hidden_states=True)
        add = torch.zeros(B, 1, S, S)
        add = add.masked_fill(mask2d[:, None, None, :] == 0, torch.finfo(torch.float32).min)
        bidir = m(inputs_embeds=emb, attention_mask=add, output_hidden_states=True)
    g5 = {"language_model." + k: v.detach().numpy() for k, v in m.state_dict().items()}
    np.savez(OUT / "g5_llama.npz", embeds=emb.numpy(), mask=mask2d.numpy().astype(bool), hidden_causal=causal.hidden_states[-1].numpy(),
             hidden_bidirectional=bidir.hidden_states[-1].numpy(), logits_bidirectional=bidir.logits.numpy(), **g5)

    # ---- G10 collator layout ----------------------------------------------------------------------------------------------
    coll = ref.data_utils.PaddedCollatorForActionPrediction(model_max_length=2048, pad_token_id=32000, padding_side="right")
    inst = []
    for i, tp in enumerate([6, 4, 9]):
        L = tp + 56 + 1
        iid, lab = label_rows(rng, 1, [tp], 56, L)
        inst.append(dict(input_ids=torch.from_numpy(iid[0]), labels=torch.from_numpy(lab[0]),
                         pixel_values=torch.randn(6, 8, 8), pixel_values_wrist=torch.randn(6, 8, 8),
                         actions=rng.uniform(-1, 1, size=(8, 7)).astype(np.float32), proprio=rng.uniform(-1, 1, size=(1, 8)).astype(np.float32),
                         dataset_name=b"libero_spatial_no_noops"))
    batch = coll(inst)
    g10 = {}
    for i, it in enumerate(inst):
        for k in ("input_ids", "labels", "pixel_values", "pixel_values_wrist"):
            g10[f"inst{i}.{k}"] = it[k].numpy()
        g10[f"inst{i}.actions"], g10[f"inst{i}.proprio"] = it["actions"], it["proprio"]
    np.savez(OUT / "g10_collator.npz", n=np.int64(len(inst)), **g10,
             **{"batch." + k: v.numpy() for k, v in batch.items() if isinstance(v, torch.Tensor)})
    print("wrote", sorted(p.name for p in OUT.glob("*.npz")))


if __name__ == "__main__":
    main()
