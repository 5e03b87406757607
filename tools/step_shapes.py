"""Per-shape time of every gemm_nt launch in one full 7B LoRA step (HIP events around each launch): which shapes to tune."""
import importlib, sys, collections
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
for k in ("pixel_values", "actions", "proprio"):
    batch[k] = batch[k].to(dev, torch.bfloat16)

def step():
    eng.zero_grad(); eng.train_step_fwd_bwd(batch); eng.adamw_step(lr=5e-4); eng.refresh_derived()

for _ in range(2): step()
records = []
orig = ops.gemm
def traced(a, b, **kw):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(a, b, **kw); e1.record()
    K2 = kw["b2"].shape[1] if kw.get("b2") is not None else 0
    records.append(((a.shape[0], b.shape[0], b.shape[1], K2, kw.get("a_group_n", 0)), e0, e1))
    return out
ops.gemm = traced
engine_mod.ops.gemm = traced
tn_records = []
orig_tn = ops.gemm_tn_grouped
def traced_tn(problems):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); orig_tn(problems); e1.record()
    tn_records.append((tuple((x.shape[0], x.shape[1], y.shape[1]) for x, y, _ in problems), e0, e1))
ops.gemm_tn_grouped = traced_tn
engine_mod.ops.gemm_tn_grouped = traced_tn
step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in records:
    agg[key][0] += 1; agg[key][1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print(f"total gemm_nt ms {tot:.1f}, launches {len(records)}")
print(f"{'M':>6} {'N':>6} {'K':>6} {'K2':>4} {'grp':>4} {'n':>5} {'ms':>8} {'us/launch':>10} {'TF':>6} {'GB/s':>6}")
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, K2, g = key
    G = (N // g) if g else 1
    fl = 2.0 * M * N * (K + K2)
    by = 2.0 * (M * K + N * K + M * N + (M + N) * K2)
    print(f"{M:6d} {N:6d} {K:6d} {K2:4d} {g:4d} {n:5d} {ms:8.2f} {1e3 * ms / n:10.1f} {fl * n / ms / 1e9:6.0f} {by * n / ms / 1e6:6.0f}")
agg = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in tn_records:
    agg[key][0] += 1; agg[key][1] += e0.elapsed_time(e1)
print(f"total gemm_tn_grouped ms {sum(v[1] for v in agg.values()):.1f}, launches {len(tn_records)}")
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    by = sum(2.0 * M * (P + Q) + 4.0 * P * Q for M, P, Q in key)
    print(f"{str(key):90s} {n:4d} {ms:7.2f} ms {1e3 * ms / n:7.1f} us {by * n / ms / 1e6:6.0f} GB/s")
