"""G21: the reference's OWN `run_forward_pass` (vla-scripts/finetune.py:280-451), executed in the build container.

finetune.py cannot be imported (draccus, peft, wandb, TensorFlow through the dataset package, `transformers.AutoModelForVision2Seq`), and its
module-level imports are not what is being pinned.  The function itself is taken out of the file with `ast` (the FunctionDef nodes of
`run_forward_pass` and `run_diffusion_sampling`, compiled unchanged from /root/reference at generation time -- nothing of it is written to the repo) and
executed in a namespace that holds exactly the names it uses: torch, the reference's own `train_utils` functions and platform constants.  It runs on
  * the reference's OpenVLAForActionPrediction assembled as for G19 (stock HF tiny Llama, duck-typed towers: make_golden_ref_model.py) behind a
    DDP-style `.module` wrapper that casts floating inputs to the fp32 the CPU model computes in (the reference relies on CUDA autocast for that),
  * the reference's L1RegressionActionHead / NoisePredictionModel in bf16 (finetune.py:910 casts the head), ProprioProjector / NoisyActionProjector,
  * the reference's ActionTokenizer over tests/duck_tokenizer.py.
What this pins: which hidden rows feed the head (`last_hidden[:, num_patches:-1][current | next]`: the shift-by-one gather), the three objectives
(L1 on bf16 actions, next-token cross entropy + token accuracies + decoded L1, diffusion noise MSE with the timestep token and noisy-action embeddings)
and the metrics dictionary.  The DDIM scheduler behind `sample_noisy_actions` is the oracle's (diffusers is absent): its draws are recorded.

    python tests/golden/make_golden_run_forward_pass.py
"""
import ast
import sys
import types
from pathlib import Path
from typing import Dict, Optional, Tuple, Type  # noqa: F401  (names the extracted functions' annotations use)

import numpy as np
import torch
import torch.nn as nn

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_golden_ref_model as mg  # noqa: E402
from oracle import vla_oracle as vo  # noqa: E402
from tests.duck_tokenizer import DuckTokenizer  # noqa: E402

OUT = Path(__file__).resolve().parent
REF_FT = Path("/root/reference/vla-scripts/finetune.py")


def extract_functions(names):
    tree = ast.parse(REF_FT.read_text())
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names], type_ignores=[])
    assert len(mod.body) == len(names)
    return compile(mod, str(REF_FT), "exec")


class DDPish(nn.Module):
    """`.module` like DistributedDataParallel; the CPU model computes in fp32, so floating inputs are widened (CUDA autocast does that in the reference)."""

    def __init__(self, module, widen=False):
        super().__init__()
        self.module, self.widen = module, widen

    def forward(self, *a, **k):
        if self.widen:
            k = {n: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for n, v in k.items()}
        return self.module(*a, **k)


def main():
    torch.manual_seed(0)
    ref = mg.load_reference()
    at_mod = mg._load("prismatic.vla.action_tokenizer", "prismatic/vla/action_tokenizer.py")
    ns = {"torch": torch, "nn": nn, "Tuple": Tuple, "Dict": Dict, "CausalLMOutputWithPast": object,
          "get_current_action_mask": ref.train_utils.get_current_action_mask, "get_next_actions_mask": ref.train_utils.get_next_actions_mask,
          "compute_token_accuracy": ref.train_utils.compute_token_accuracy, "compute_actions_l1_loss": ref.train_utils.compute_actions_l1_loss,
          "NUM_ACTIONS_CHUNK": ref.constants.NUM_ACTIONS_CHUNK, "ACTION_DIM": ref.constants.ACTION_DIM}
    exec(extract_functions({"run_forward_pass", "run_diffusion_sampling"}), ns)
    run_forward_pass = ns["run_forward_pass"]

    cfg = vo.tiny_config()
    SEED = 7
    sd = vo.random_state_dict(cfg, seed=SEED, lora=False, film=True, diffusion=False)
    sd_diff = vo.random_state_dict(cfg, seed=SEED, lora=False, film=True, diffusion=True)
    b = mg.ragged_batch(21, (9, 7, 12), 2)
    b["pixel_values"] = torch.from_numpy(b["pixel_values"]).to(torch.bfloat16).float().numpy()      # bf16-exact, as the step's `.to(bfloat16)` leaves them
    batch = {k: torch.from_numpy(v) for k, v in b.items()}
    g = dict(b)
    g.update(sd_seed=np.int64(SEED), sd_checksum=mg.sd_checksum(sd), sd_diffusion_checksum=mg.sd_checksum(sd_diff))
    tok = at_mod.ActionTokenizer(DuckTokenizer())
    ppj = DDPish(mg.load_mlp(ref.projectors.ProprioProjector(cfg.llm_dim, cfg.proprio_dim), sd, "proprio_projector."))
    l1 = DDPish(mg.load_head(ref.action_heads.L1RegressionActionHead(cfg.llm_dim, cfg.llm_dim, cfg.action_dim), sd, "action_head.").to(torch.bfloat16))
    P = 2 * cfg.dino.n_patches + 1

    # the discrete objective runs on an lm_head whose 256 action rows are scaled up (x LM_ACTION_GAIN), so the argmax lands inside the action range and
    # the decoded-L1 / accuracy metrics depend on the prediction instead of clipping to one bin
    LM_ACTION_GAIN = 8.0
    sd_disc = dict(sd)
    lm = sd["language_model.lm_head.weight"].clone()
    lm[31744:32000] *= LM_ACTION_GAIN
    sd_disc["language_model.lm_head.weight"] = lm
    g["lm_action_gain"] = np.float64(LM_ACTION_GAIN)
    for mode in ("bidirectional", "causal"):
        vla = DDPish(mg.make_vla(ref, sd, cfg, 2, False, mode), widen=True)
        vla_d = DDPish(mg.make_vla(ref, sd_disc, cfg, 2, False, mode), widen=True)
        with torch.no_grad():
            loss, m = run_forward_pass(vla, l1, None, ppj, batch, tok, "cpu", True, False, True, False, P)
            g.update({f"{mode}.l1.loss": np.float64(loss.float().item()), **{f"{mode}.l1.{k}": np.float64(v) for k, v in m.items()}})
            loss, m = run_forward_pass(vla_d, None, None, ppj, batch, tok, "cpu", False, False, True, False, P)
            g.update({f"{mode}.discrete.loss": np.float64(loss.item()), **{f"{mode}.discrete.{k}": np.float64(v) for k, v in m.items()}})
            out = vla_d(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], pixel_values=batch["pixel_values"], labels=batch["labels"],
                        output_hidden_states=True, proprio=batch["proprio"], proprio_projector=ppj, use_film=False)
            g[f"{mode}.discrete.predicted_ids"] = out.logits[:, P:-1].argmax(dim=2).numpy()

    # diffusion objective (MSE on the predicted noise): FiLM on, one more patch token (the timestep embedding)
    T = 50
    ddim = vo.DDIM(T)
    ah = ref.action_heads
    dhead = ah.DiffusionActionHead.__new__(ah.DiffusionActionHead)
    nn.Module.__init__(dhead)
    dhead.action_dim, dhead.num_diffusion_steps = cfg.action_dim, T
    dhead.noise_scheduler = types.SimpleNamespace(config=types.SimpleNamespace(num_train_timesteps=T),
                                                  add_noise=lambda x0, n, t: ddim.add_noise(x0.float(), n.float(), t).to(x0.dtype))
    dhead.time_encoder = ah.SinusoidalPositionalEncoding(dim=cfg.llm_dim)
    dhead.noise_predictor = ah.NoisePredictionModel(transformer_hidden_dim=cfg.llm_dim * cfg.action_dim, hidden_dim=cfg.llm_dim, action_dim=cfg.action_dim)
    mg.load_head(dhead.noise_predictor, sd_diff, "action_head.noise_predictor.")
    dhead = dhead.to(torch.bfloat16)
    napj = DDPish(mg.load_mlp(ref.projectors.NoisyActionProjector(cfg.llm_dim), sd_diff, "noisy_action_projector."))
    drawn = {}
    orig = dhead.sample_noisy_actions

    def recording(gt):
        out = orig(gt)
        drawn.update(out)
        return out

    dhead.sample_noisy_actions = recording
    vla = DDPish(mg.make_vla(ref, sd_diff, cfg, 2, True, "bidirectional"), widen=True)
    torch.manual_seed(210)
    with torch.no_grad():
        loss, m = run_forward_pass(vla, DDPish(dhead), napj, ppj, batch, tok, "cpu", False, True, True, True, P + 1, compute_diffusion_l1=False, num_diffusion_steps=T)
    torch.manual_seed(210)                                           # the same draws, to recover the integer timesteps behind the recorded embeddings
    noise2 = torch.randn(size=(3, 8, 7), dtype=torch.bfloat16)
    ts = torch.randint(low=0, high=T, size=(3,))
    assert torch.equal(noise2, drawn["noise"])
    g.update({"diffusion.loss": np.float64(loss.float().item()), "diffusion.loss_value": np.float64(m["loss_value"]), "diffusion.T": np.int64(T),
              "diffusion.noise": drawn["noise"].float().numpy(), "diffusion.noisy_actions": drawn["noisy_actions"].float().numpy(), "diffusion.timesteps": ts.numpy()})
    np.savez_compressed(OUT / "g21_ref_run_forward_pass.npz", **g)
    print("wrote", (OUT / "g21_ref_run_forward_pass.npz").stat().st_size, "bytes;", {k: float(v) for k, v in g.items() if k.endswith("loss") or k.endswith("accuracy")})


if __name__ == "__main__":
    main()
