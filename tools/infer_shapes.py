"""Where one batch-1 inference chunk (merged LoRA, eager) spends its GPU time: per kernel family and per GEMM shape."""
import importlib, sys, collections
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
load = importlib.import_module
ops, engine_mod, weights_mod, synth, config_mod = (load("openvla-oft_amd.ops"), load("openvla-oft_amd.engine"), load("openvla-oft_amd.weights"),
                                                   load("openvla-oft_amd.synthetic"), load("openvla-oft_amd.config"))
dev = torch.device("cuda:0")
cfg = config_mod.OPENVLA_7B
sd = weights_mod.random_state_dict(cfg, dev, seed=0, lm_head=False, lora=False)
get, has = weights_mod.make_getter(sd, dev)
eng = engine_mod.VLAEngine(cfg, get, dev, lora=False, use_proprio=True, head="l1", has=has)
del sd, get
b1 = synth.make_batch(1, seed=77, num_images=cfg.num_images, chunk=cfg.chunk, action_dim=cfg.action_dim, proprio_dim=cfg.proprio_dim)
b1["pixel_values"] = b1["pixel_values"].to(dev, torch.bfloat16); b1["proprio"] = b1["proprio"].to(dev, torch.bfloat16).reshape(1, -1)
def once():
    out = eng.forward(b1["input_ids"], b1["attention_mask"], b1["pixel_values"], b1["labels"], proprio=b1["proprio"], train=False)
    ah, _ = eng.gather_action_hidden(out["hidden"], out["action_rows"])
    return eng.head.fwd(ah)[0]
for _ in range(3): once()
records = []
orig = ops.gemm
def traced(a, b, **kw):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(a, b, **kw); e1.record()
    records.append(((a.shape[0], b.shape[0], b.shape[1]), e0, e1))
    return out
ops.gemm = traced; engine_mod.ops.gemm = traced
ops.PROFILE = []
once(); torch.cuda.synchronize()
fam = collections.defaultdict(lambda: [0, 0.0])
for family, e0, e1, fl in ops.PROFILE:
    fam[family][0] += 1; fam[family][1] += e0.elapsed_time(e1)
ops.PROFILE = None
print({k: (v[0], round(v[1], 2)) for k, v in fam.items()})
agg = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in records:
    agg[key][0] += 1; agg[key][1] += e0.elapsed_time(e1)
print(f"total gemm ms {sum(v[1] for v in agg.values()):.2f} launches {len(records)}")
for (M, N, K), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{M:6d} {N:6d} {K:6d} n={n:4d} ms={ms:7.2f} us/launch={1e3 * ms / n:8.1f} TF={2.0 * M * N * K * n / ms / 1e9:6.0f} weightGB/s={N * K * 2 * n / ms / 1e6:7.0f}")
