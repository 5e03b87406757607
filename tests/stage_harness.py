"""Stage-by-stage comparison of the HIP engine with the oracle (TEST INFRASTRUCTURE: imports `oracle/`).

One seeded state dict and one batch go through
  * the HIP engine (`openvla-oft_amd.engine.VLAEngine`, through the C-ABI),
  * the oracle in fp32                       ("exact" arithmetic on the same bf16-exact weights),
  * the oracle in its bf16-emulation mode    (re-rounding where the reference's autocast path materialises bf16 tensors),
  * the oracle in "native" mode              (stock PyTorch-ROCm eager bf16 ops: the reference's own execution path,
                                              `north_star`'s "reference PyTorch path"),
and every intermediate of the path is compared: tower features -> projected patches -> input of every decoder layer ->
final hidden state -> gathered action rows -> predicted actions -> loss (-> lm_head argmax ids).  The error of each
evaluation is measured against fp32, so the table answers "where does a difference enter and how does it grow" for all
three bf16 evaluations at once.  The oracle runs on the GPU with torch ops here (the CPU would need minutes at full size).

CLI (on a GPU box):  python -m tests.stage_harness [--tiny] [--batch 8] [--out gpurun_out/stage_diff.json]
"""
from __future__ import annotations

import dataclasses
import importlib
import json
import sys
from pathlib import Path
from typing import Dict

import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from oracle import vla_oracle as vo  # noqa: E402  (checker only)

BF = torch.bfloat16


def relmax(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-20)).item()


def rel2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


class RecordingOracle(vo.Oracle):
    """The oracle with taps on the stages of the path (no arithmetic of its own)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.stages: Dict[str, torch.Tensor] = {}
        self._layer = 0

    def vision_backbone(self, pixel_values, film_avg=None):
        out = super().vision_backbone(pixel_values, film_avg)
        self.stages["feats"] = out.detach()
        return out

    def projector(self, x):
        out = super().projector(x)
        self.stages["patches"] = out.detach()
        return out

    def rmsnorm(self, x, name):
        if name.endswith("input_layernorm.weight"):
            self.stages[f"x{int(name.split('.')[3]):02d}"] = x.detach()
        elif name == "language_model.model.norm.weight":
            self.stages["x_final"] = x.detach()
        return super().rmsnorm(x, name)


def oracle_config(cfg) -> vo.OracleConfig:
    vit_fields = ("dim", "depth", "heads", "mlp_hidden", "n_prefix", "layerscale", "patch", "image_size", "eps", "act")
    return vo.OracleConfig(**{f: getattr(cfg, f) for f in ("llm_dim", "llm_layers", "llm_heads", "llm_ff", "vocab", "rms_eps", "rope_theta", "num_images",
                                                           "lora_rank", "lora_alpha", "action_dim", "chunk", "proprio_dim", "norm_type")},
                           dino=vo.VitConfig(**{f: getattr(cfg.dino, f) for f in vit_fields}),
                           siglip=vo.VitConfig(**{f: getattr(cfg.siglip, f) for f in vit_fields}))


def device_batch(batch, dev, float_dtype):
    out = {}
    for k, v in batch.items():
        if not torch.is_tensor(v):
            out[k] = v
        elif v.is_floating_point():
            out[k] = v.to(dev, BF).to(float_dtype)     # every evaluation sees the same bf16-exact inputs
        else:
            out[k] = v.to(dev)
    return out


def oracle_stages(ocfg, sd, batch, dev, mode: str, lm_head: bool = True):
    """Forward of the oracle in `mode` on the GPU (no gradients) -> stage dict (tensors stay on the device)."""
    b = device_batch(batch, dev, BF if mode == "native" else torch.float32)
    o = RecordingOracle(ocfg, sd, mode=mode)
    with torch.no_grad():
        loss, pred, ah = o.train_forward(b)
        st = dict(o.stages)
        st["hidden"] = o.rmsnorm(st["x_final"], "language_model.model.norm.weight")
        st["action_hidden"], st["pred"], st["loss"] = ah, pred, loss.reshape(1)
        if lm_head and "language_model.lm_head.weight" in sd:
            logits = o.lm_logits(ah).float()
            st["token_ids"] = logits.argmax(-1)
            if mode == "fp32":
                top2 = logits.topk(2, dim=-1).values
                st["token_margin"] = top2[..., 0] - top2[..., 1]
                st["logits32"] = logits
    return st


def hip_stages(eng, batch, lm_head: bool = True):
    """Forward of the HIP engine with every stage kept (train=True keeps the per-layer inputs; the full last layer runs)."""
    ops = importlib.import_module("openvla-oft_amd.ops")
    cfg = eng.cfg
    dev = eng.device
    B = batch["input_ids"].shape[0]
    out = eng.forward(batch["input_ids"], batch["attention_mask"], batch["pixel_values"].to(dev, BF), batch["labels"], proprio=batch["proprio"].to(dev, BF),
                      train=True)
    vsaved, _, _, lsaved = out["saved"][:4]
    S, D = out["S"], cfg.llm_dim
    st = {"feats": vsaved[1][0].view(B, -1, cfg.vision_dim), "patches": out["patches"][0][:, : out["patches"][1]]}
    for i, sv in enumerate(lsaved[0]):
        st[f"x{i:02d}"] = sv[0].view(B, S, D)
    st["x_final"] = lsaved[1].view(B, S, D)
    st["hidden"] = out["hidden"]
    ah, _ = eng.gather_action_hidden(out["hidden"], out["action_rows"])
    st["action_hidden"] = ah.view(B, cfg.num_action_tokens, D)
    tgt = batch["actions"].to(dev, BF).reshape(-1, cfg.action_dim).contiguous()
    pred, loss_sum, _ = eng.head.fwd(ah, target=tgt)
    st["pred"] = pred.view(B, cfg.chunk, cfg.action_dim)
    st["loss"] = loss_sum / pred.numel()
    if lm_head and eng.lm_head is not None:
        logits = ops.gemm(ah, eng.lm_head)                    # lm_head on the action rows only (modeling.logits_for)
        st["token_ids"] = logits.float().argmax(-1).view(B, -1)
        st["logits_hip"] = logits.float().view(B, cfg.num_action_tokens, -1)
    return st, out


def valid_rows(batch, P, dev):
    """Rows of the multimodal sequence that are not right padding (pad rows are don't-care on every side)."""
    B = batch["input_ids"].shape[0]
    am = batch["attention_mask"].to(dev).bool()
    return torch.cat([torch.ones(B, 1 + P, dtype=torch.bool, device=dev), am[:, 1:]], 1)


def compare(stages: Dict[str, Dict[str, torch.Tensor]], batch, dev, ref: str = "fp32"):
    """{stage: {evaluation: (relmax, rel-L2) against `ref`}} for every stage the reference has."""
    table = {}
    r = stages[ref]
    S = r["hidden"].shape[1]
    P = S - batch["input_ids"].shape[1]
    vmask = valid_rows(batch, P, dev)
    for name in r:
        if name in ("token_ids", "token_margin", "logits32", "logits_hip"):
            continue
        row = {}
        for ev, st in stages.items():
            if ev == ref or name not in st:
                continue
            a, b = st[name], r[name]
            if a.dim() == 3 and a.shape[1] == S:
                a, b = a[vmask], b[vmask]
            if name == "pred":
                row[ev] = ((a.float() - b.float()).abs().max().item(), rel2(a, b))     # L-inf in action units (north_star's measure)
            elif name == "loss":
                row[ev] = (abs(a.float().item() - b.float().item()), abs(a.float().item() - b.float().item()) / abs(b.float().item()))
            else:
                row[ev] = (relmax(a, b), rel2(a, b))
        table[name] = row
    return table


def token_report(stages, ref: str = "fp32"):
    """Agreement of the argmax action-token ids with the fp32 evaluation, and for every disagreement the fp32 top-2 margin."""
    r = stages[ref]
    rep = {}
    for ev, st in stages.items():
        if ev == ref or "token_ids" not in st:
            continue
        same = st["token_ids"] == r["token_ids"]
        bad_margin = r["token_margin"][~same]
        # fp32 logit gap between the fp32 winner and the id this evaluation picked
        picked = torch.gather(r["logits32"], -1, st["token_ids"].unsqueeze(-1)).squeeze(-1)
        best = r["logits32"].max(-1).values
        rep[ev] = dict(agree=same.float().mean().item(), n=int(same.numel()), n_diff=int((~same).sum()),
                       max_gap_of_a_flip=(best - picked)[~same].max().item() if (~same).any() else 0.0,
                       max_margin_of_a_flip=bad_margin.max().item() if (~same).any() else 0.0,
                       median_margin=r["token_margin"].median().item())
    return rep


def format_table(table, evals):
    lines = [f"{'stage':<14}" + "".join(f"{e + ' max':>14}{e + ' L2':>12}" for e in evals)]
    for name, row in table.items():
        lines.append(f"{name:<14}" + "".join((f"{row[e][0]:>14.3e}{row[e][1]:>12.3e}" if e in row else " " * 26) for e in evals))
    return "\n".join(lines)


def run_all(cfg, sd, batch, dev, eng=None, modes=("fp32", "bf16", "native"), lm_head=True):
    load = importlib.import_module
    engine_mod, weights_mod = load("openvla-oft_amd.engine"), load("openvla-oft_amd.weights")
    ocfg = oracle_config(cfg)
    if eng is None:
        get, has = weights_mod.make_getter(sd, dev)
        eng = engine_mod.VLAEngine(cfg, get, dev, lora=True, use_proprio=True, head="l1", has=has)
    stages = {}
    for m in modes:
        stages[m] = oracle_stages(ocfg, sd, batch, dev, m, lm_head)
    stages["hip"], _ = hip_stages(eng, batch, lm_head)
    torch.cuda.synchronize()
    return eng, stages


def main():
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--tiny", action="store_true", help="the reduced-size model of smoke() / test_engine_gpu.py")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--out", default="gpurun_out/stage_diff.json")
    args = ap.parse_args()
    load = importlib.import_module
    weights_mod, synth, config_mod = load("openvla-oft_amd.weights"), load("openvla-oft_amd.synthetic"), load("openvla-oft_amd.config")
    dev = torch.device("cuda:0")
    if args.tiny:
        ocfg = vo.tiny_config()
        cfg = config_mod.VLAConfig.from_any(ocfg)
        sd = {k: v.to(dev, BF) for k, v in vo.random_state_dict(ocfg, seed=0).items()}
        batch = synth.make_batch(args.batch, seed=3, prompt_lens=([9, 7] * args.batch)[: args.batch], image_size=56)
    else:
        cfg = config_mod.OPENVLA_7B
        sd = weights_mod.random_state_dict(cfg, dev, seed=0, lm_head=True)
        batch = synth.make_batch(args.batch, seed=1000)
    eng, stages = run_all(cfg, sd, batch, dev)
    table = compare(stages, batch, dev)
    evals = ["hip", "bf16", "native"]
    print(format_table(table, evals))
    tok = token_report(stages)
    print(json.dumps(tok, indent=1))
    # HIP against the two other bf16 evaluations directly
    for a, b in (("hip", "bf16"), ("hip", "native"), ("native", "bf16")):
        d = (stages[a]['pred'].float() - stages[b]['pred'].float()).abs()
        ulp = torch.exp2(torch.floor(torch.log2(stages[b]['pred'].float().abs().clamp_min(1e-30))) - 7)    # bf16 ulp of each output value
        print(f"pred L-inf {a} vs {b}: {d.max().item():.3e} = {(d / ulp).max().item():.1f} bf16 ulp of the output; max |pred| {stages[b]['pred'].float().abs().max().item():.2f}")
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps({"table": table, "tokens": tok, "batch": args.batch, "tiny": args.tiny}, indent=1))


if __name__ == "__main__":
    main()


# ======================================================================================================================
# a seeded state dict CONDITIONED LIKE A TRAINED MODEL (tests/test_fullsize_e2e_gpu.py: the absolute-tolerance tests)
# ======================================================================================================================
ACTION_ID0, N_ACTION_IDS = 31744, 256      # prismatic/vla/constants.py:12 (ACTION_TOKEN_BEGIN_IDX + 1) .. vocab 32000 (action_tokenizer.py:30-47)


def conditioned_state_dict(cfg, dev, seed: int = 1, *, branch_gain: float = 0.25, head_gain: float = 0.25, lm_gain: float = 4.0):
    """`weights.random_state_dict` re-conditioned so that absolute parity numbers mean something (no checkpoint exists offline):

      * residual branches: every block's OUTPUT projection (decoder o_proj / down_proj, ViT attn.proj / mlp.fc2, their biases and adapter B
        factors) is scaled by `branch_gain`, so a block adds a fraction of the stream's magnitude as trained transformers do -- with N(0, 0.02..0.03)
        output projections every block rewrites the stream and a bf16 rounding of a branch is a rounding of the whole state;
      * action head: `fc2` scaled by `head_gain` so the predicted actions live in [-1, 1] (the normalised action range, constants.py:26-52)
        instead of +-4, where one output ulp is 4x coarser;
      * lm_head: zero except the 256 action-token rows, which are `lm_gain` x an orthonormal block: the action logits are then 256 independent
        projections of the hidden state with an O(lm_gain) spread, i.e. most rows have a real top-2 margin (a random N(0, 0.05) lm_head gives
        every one of 32 064 ids a chance and no margin anywhere).
    Returns the state dict (bf16, on `dev`)."""
    load = importlib.import_module
    weights_mod = load("openvla-oft_amd.weights")
    sd = weights_mod.random_state_dict(cfg, dev, seed=seed, lm_head=True)
    for k in list(sd):
        out_proj = any(t in k for t in (".self_attn.o_proj.", ".mlp.down_proj.", ".attn.proj.", ".mlp.fc2."))
        if out_proj and (k.endswith(".weight") or k.endswith(".bias")) and ".lora_A." not in k:
            sd[k] = (sd[k].float() * branch_gain).to(BF)
    for k in ("action_head.model.fc2.weight", "action_head.model.fc2.bias"):
        sd[k] = (sd[k].float() * head_gain).to(BF)
    g = torch.Generator(device=dev).manual_seed(seed + 77)
    q, _ = torch.linalg.qr(torch.randn(cfg.llm_dim, N_ACTION_IDS, generator=g, device=dev, dtype=torch.float32))
    lm = torch.zeros_like(sd["language_model.lm_head.weight"])
    lm[ACTION_ID0: ACTION_ID0 + N_ACTION_IDS] = (lm_gain * q.T).to(BF)
    sd["language_model.lm_head.weight"] = lm
    return sd


def fit_lm_head_to_labels(sd, cfg, action_hidden32, batch, *, margin: float = 8.0, ridge: float = 1e-3):
    """`lm_head` of a model that has LEARNED this batch: the 256 action-token rows are the closed-form ridge-regression solution that maps the
    fp32 hidden state of every action row to a logit of `margin` on its label (the batch's own action tokens, datasets.py:75) and 0 on the other
    action ids -- W = Y^T (H H^T + lambda I)^-1 H, lambda = ridge * mean diag -- rounded to bf16; every other vocabulary row is zero.  What a trained
    checkpoint's lm_head does for its training data (a top-2 margin on every row), obtained without a training run.  Returns the fp32 fit quality
    (min margin over the rows it was fitted on)."""
    H = action_hidden32.reshape(-1, cfg.llm_dim).double()
    lab = batch["labels"].to(H.device)
    tgt = lab[(lab > 31743)].reshape(-1)                        # action tokens, row-major (b, slot): the order of the action rows
    assert tgt.numel() == H.shape[0]
    Y = torch.zeros(H.shape[0], N_ACTION_IDS, dtype=torch.float64, device=H.device)
    Y[torch.arange(H.shape[0]), tgt - ACTION_ID0] = margin
    K = H @ H.T
    K += ridge * K.diagonal().mean() * torch.eye(K.shape[0], dtype=K.dtype, device=K.device)
    W = (torch.linalg.solve(K, Y).T @ H).float()               # [256, D]
    lm = torch.zeros_like(sd["language_model.lm_head.weight"])
    lm[ACTION_ID0: ACTION_ID0 + N_ACTION_IDS] = W.to(BF)
    sd["language_model.lm_head.weight"] = lm
    logits = (H.float() @ W.to(BF).float().T)
    top2 = logits.topk(2, dim=-1).values
    ok = logits.argmax(-1) == (tgt - ACTION_ID0)
    return dict(fit_accuracy=ok.float().mean().item(), min_margin=(top2[:, 0] - top2[:, 1]).min().item(), w_rms=W.pow(2).mean().sqrt().item(),
                sv_min_over_max=(torch.linalg.svdvals(H.float())[[-1, 0]]).tolist())
