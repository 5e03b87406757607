"""Per-op time inside the vision towers (single stream, HIP events around every ops.* call) for one forward + backward at B = 8."""
import importlib, os, sys, collections
os.environ["OVLA_VIT_STREAMS"] = "1"
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
pv = batch["pixel_values"].to(dev, torch.bfloat16)
def run():
    eng.zero_grad()
    out, saved = eng.vision_fwd(pv, True)
    eng.vision_bwd(torch.randn_like(out).view(-1, cfg.llm_dim).contiguous(), saved)
for _ in range(2): run()
torch.cuda.synchronize()
rec = []
def wrap(name):
    fn = getattr(ops, name)
    def w(*a, **k):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record()
        key = name
        if name == "gemm":
            key = f"gemm M={a[0].shape[0]} N={a[1].shape[0]} K={a[1].shape[1]}" + (" +lora" if k.get("a2") is not None else "") + (f" act{k['act']}" if k.get("act") else "")
        rec.append((key, e0, e1))
        return r
    setattr(ops, name, w)
for n in ("gemm", "gemm_tn_grouped", "gemm_tn", "attn_fwd", "attn_bwd", "norm_fwd", "norm_bwd", "act_bwd", "colscale", "im2col", "vit_embed", "copy_rows", "transpose", "cvt_f32_to_bf16"):
    if hasattr(ops, n): wrap(n)
run(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for k, e0, e1 in rec:
    agg[k][0] += 1; agg[k][1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print(f"vision fwd+bwd, event-timed ops: {tot:.2f} ms in {len(rec)} calls")
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k:52s} n={n:4d} {ms:7.2f} ms  {1e3 * ms / n:7.1f} us/call")
