"""Calibration run for the absolute-tolerance parity test (tests/test_fullsize_e2e_gpu.py::test_conditioned_*): the conditioned full-size
state dict (tests/stage_harness.py: conditioned_state_dict) through HIP / fp32 / bf16-emulation / stock-eager, B = 8 and B = 1.

    python tools/conditioned_parity.py [--branch 0.25 --head 0.25 --lm 4.0] [--out gpurun_out/conditioned.json]
"""
import argparse
import importlib
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests import stage_harness as sh  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--branch", type=float, default=0.25)
    ap.add_argument("--head", type=float, default=0.25)
    ap.add_argument("--lm", type=float, default=4.0)
    ap.add_argument("--fit", action="store_true", help="lm_head fitted to the batch's labels by ridge regression on the fp32 hidden states (stage_harness.fit_lm_head_to_labels)")
    ap.add_argument("--margin", type=float, default=8.0)
    ap.add_argument("--ridge", type=float, default=1e-3)
    ap.add_argument("--out", default="gpurun_out/conditioned.json")
    a = ap.parse_args()
    load = importlib.import_module
    synth, config_mod = load("openvla-oft_amd.synthetic"), load("openvla-oft_amd.config")
    dev = torch.device("cuda:0")
    cfg = config_mod.OPENVLA_7B
    sd = sh.conditioned_state_dict(cfg, dev, 1, branch_gain=a.branch, head_gain=a.head, lm_gain=a.lm)
    eng, rec = None, {}
    batch8 = synth.make_batch(8, seed=2000)
    if a.fit:
        st32 = sh.oracle_stages(sh.oracle_config(cfg), sd, batch8, dev, "fp32", lm_head=False)
        print("lm_head fit:", sh.fit_lm_head_to_labels(sd, cfg, st32["action_hidden"], batch8, margin=a.margin, ridge=a.ridge))
        del st32
    for B in (8, 1):
        batch = batch8 if B == 8 else {k: (v[:1] if torch.is_tensor(v) else v[:1]) for k, v in batch8.items()}
        eng, st = sh.run_all(cfg, sd, batch, dev, eng=eng)
        table = sh.compare(st, batch, dev)
        print(f"\n==== B = {B} ====\n" + sh.format_table(table, ["hip", "bf16", "native"]))
        p32 = st["fp32"]["pred"].float()
        d = {f"{x}-{y}": (st[x]["pred"].float() - st[y]["pred"].float()).abs().max().item() for x, y in (("hip", "fp32"), ("native", "fp32"), ("bf16", "fp32"), ("hip", "native"))}
        print(f"max |a| fp32 {p32.abs().max().item():.3f}, std {p32.std().item():.3f};  actions L-inf: {json.dumps({k: round(v, 5) for k, v in d.items()})}")
        m = st["fp32"]["token_margin"]
        lg = st["fp32"]["logits32"]
        print(f"fp32 action logits: max |.| {lg.abs().max().item():.2f}; top-2 margin median {m.median().item():.3f}, quantiles 5/25% "
              f"{m.flatten().kthvalue(max(1, int(0.05 * m.numel()))).values.item():.3f} / {m.flatten().kthvalue(max(1, int(0.25 * m.numel()))).values.item():.3f}")
        tok = sh.token_report(st)
        for ev, r in tok.items():
            print(f"  ids {ev}: {r['n_diff']} / {r['n']} differ from fp32; largest fp32 margin of a flipped row {r['max_margin_of_a_flip']:.4f}")
        rec[f"B{B}"] = {"table": table, "pred": d, "tokens": tok, "max_abs_action": p32.abs().max().item()}
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
