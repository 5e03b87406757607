"""Runs the same full-size fwd+bwd three times from the same state and compares loss, predictions and every trainable gradient.
The only run-to-run difference allowed is the order of the fp32 atomic adds in the weight-gradient (TN) GEMMs (~1e-7 relative)."""
import collections, importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
load = importlib.import_module
ops, engine_mod, weights_mod, synth, config_mod = (load("openvla-oft_amd.ops"), load("openvla-oft_amd.engine"), load("openvla-oft_amd.weights"),
                                                   load("openvla-oft_amd.synthetic"), load("openvla-oft_amd.config"))
dev = torch.device("cuda:0")
cfg = config_mod.OPENVLA_7B
sd = weights_mod.random_state_dict(cfg, dev, seed=0, lm_head=False)
get, has = weights_mod.make_getter(sd, dev)
eng = engine_mod.VLAEngine(cfg, get, dev, lora=True, use_proprio=True, head="l1", has=has)
del sd, get
batch = synth.make_batch(8, seed=1000, num_images=cfg.num_images, chunk=cfg.chunk, action_dim=cfg.action_dim, proprio_dim=cfg.proprio_dim)
runs = []
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    eng.zero_grad()
    loss_sum, count, pred = eng.train_step_fwd_bwd(batch)
    torch.cuda.synchronize()
    runs.append((loss_sum.item() / count, pred.float().clone(), {k: v.float().clone() for k, v in eng.export_trainable("grad").items()}))
worst = 0.0
for r in range(1, len(runs)):
    dl, dp = abs(runs[r][0] - runs[0][0]), (runs[r][1] - runs[0][1]).abs().max().item()
    per = sorted((((runs[r][2][k] - runs[0][2][k]).norm() / (runs[0][2][k].norm() + 1e-30)).item(), k) for k in runs[0][2])
    worst = max(worst, per[-1][0])
    print(f"run {r} vs 0: loss diff {dl:.3e}, pred max diff {dp:.3e}, gradient rel-L2 per tensor: median {per[len(per) // 2][0]:.3e}, max {per[-1][0]:.3e} ({per[-1][1]})")
fam = collections.defaultdict(list)
for k in runs[0][2]:
    d = ((runs[-1][2][k] - runs[0][2][k]).norm() / (runs[0][2][k].norm() + 1e-30)).item()
    fam[".".join(x for x in k.split(".") if not x.isdigit())[-64:]].append(d)
for k, v in sorted(fam.items(), key=lambda kv: -max(kv[1]))[:6]:
    print(f"   {max(v):.3e} (mean {sum(v) / len(v):.3e}, n={len(v)}) {k}")
print("DETERMINISTIC to atomic-order noise" if worst < 5e-6 else "NOT deterministic")
