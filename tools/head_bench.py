"""L1 action head forward at full size (28672 -> 4096 -> 2 x [4096 -> 4096] -> 7; 302 MB of bf16 weights, prismatic/models/action_heads.py:72-107),
64 rows (the fine-tune step: batch 8 x chunk 8) and 8 rows (one inference chunk): fused tail (ovla_head_tail_fwd) vs the unfused sequence, event-timed
back to back, next to the HBM floor of its weight stream (302 MB / 6.3 TB/s = 48 us).  The weights exceed the 256 MB Infinity Cache, so repeated
calls stream from HBM as the step does."""
import importlib, os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
load = importlib.import_module
dev = torch.device("cuda:0")
BF = torch.bfloat16
def run(fuse):
    os.environ["OVLA_FUSE_HEAD"] = fuse
    for m in [m for m in sys.modules if m.startswith("openvla-oft_amd")]:
        del sys.modules[m]
    engine_mod, weights_mod, config_mod = load("openvla-oft_amd.engine"), load("openvla-oft_amd.weights"), load("openvla-oft_amd.config")
    cfg = config_mod.OPENVLA_7B
    g = torch.Generator(device=dev).manual_seed(0)
    D, A = cfg.llm_dim, cfg.action_dim
    sd = {}
    def lin(n, o, i): sd[n + ".weight"] = (torch.randn(o, i, generator=g, device=dev) * 0.02).to(BF); sd[n + ".bias"] = (torch.randn(o, generator=g, device=dev) * 0.02).to(BF)
    def ln(n, d): sd[n + ".weight"] = torch.ones(d, device=dev, dtype=BF); sd[n + ".bias"] = torch.zeros(d, device=dev, dtype=BF)
    p = "action_head.model."
    ln(p + "layer_norm1", D * A); lin(p + "fc1", D, D * A)
    for b in range(2): ln(f"{p}mlp_resnet_blocks.{b}.ffn.0", D); lin(f"{p}mlp_resnet_blocks.{b}.ffn.1", D, D)
    ln(p + "layer_norm2", D); lin(p + "fc2", A, D)
    get, _ = weights_mod.make_getter(sd, dev)
    head = engine_mod.build_component(engine_mod.ActionHead, dev, get, p, cfg=cfg)
    out = {}
    for rows in (64, 8):
        ah = torch.randn(rows * A, D, device=dev).to(BF)
        tgt = torch.randn(rows, A, device=dev).to(BF)
        for train in (True, False):
            fn = lambda: head.fwd(ah, target=tgt if train else None, train=train)
            for _ in range(5): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): fn()
            e1.record(); torch.cuda.synchronize()
            out[(rows, train)] = e0.elapsed_time(e1) / 30 * 1e3
    head.check_fused_tail()
    return out
res = {f: run(f) for f in ("0", "1")}
print("rows train | unfused us | fused-tail us | floor 302 MB / 6.3 TB/s = 47.9 us")
for k in res["0"]:
    print(f"{k[0]:4d} {str(k[1]):5s} | {res['0'][k]:8.1f}   | {res['1'][k]:8.1f}")
