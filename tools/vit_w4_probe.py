"""Vision-tower / projector GEMM shapes at training size: the planner's choice (tile 0) against the 4-wave 256x256 configuration (118) and the 8-wave one (117),
with the epilogues the model uses (bias; bias + GELU + pre-activation save; bias + residual [+ LayerScale]); LoRA K-extension 32.  us per launch."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
BF = torch.bfloat16


def bench(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e3


shapes = [("dino qkv", 4176, 3072, 1024, "b"), ("dino proj", 4176, 1024, 1024, "brs"), ("dino fc1", 4176, 4096, 1024, "bg"), ("dino fc2", 4176, 1024, 4096, "brs"), ("dino d_fc2", 4176, 4096, 1024, ""),
          ("dino d_fc1", 4176, 1024, 4096, ""), ("sig qkv", 4096, 3456, 1152, "b"), ("sig proj", 4096, 1152, 1152, "br"), ("sig fc1", 4096, 4304, 1152, "bg"), ("sig fc2", 4096, 1152, 4304, "br"),
          ("sig d_fc2", 4096, 4304, 1152, ""), ("sig d_fc1", 4096, 1152, 4304, ""), ("proj fc1", 4096, 8704, 2176, "bg"), ("proj fc2", 4096, 4096, 8704, "bg"), ("proj fc3", 4096, 4096, 4096, "b")]
for name, M, N, K, ep in shapes:
    a = torch.randn(M, K, device=dev).to(BF); b = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    t = torch.randn(M, 32, device=dev).to(BF); lb = torch.randn(N, 32, device=dev).to(BF)
    out = torch.empty(M, N, device=dev, dtype=BF); pre = torch.empty(M, N, device=dev, dtype=BF)
    kw = dict(a2=t, b2=lb)
    if "b" in ep: kw.update(bias=torch.randn(N, device=dev).to(BF))
    if "g" in ep: kw.update(act=1, c_pre=pre)
    if "r" in ep: kw.update(residual=torch.randn(M, N, device=dev).to(BF))
    if "s" in ep: kw.update(colscale=torch.randn(N, device=dev).to(BF))
    fl = 2.0 * M * N * (K + 32)
    res = []
    for tile in (0, 122, 118, 117, 102, 101):
        try:
            us = bench(lambda: ops.gemm(a, b, out=out, tile=tile, **kw))
            res.append("t%d %6.1f (%4.0f)" % (tile, us, fl / us / 1e6))
        except Exception as e:
            res.append("t%d n/a" % tile)
    print("%-10s %5d %5d %5d %-3s plan %s | " % (name, M, N, K, ep, ops.gemm_plan(M, N, K, 32)[:4]) + " | ".join(res), flush=True)
