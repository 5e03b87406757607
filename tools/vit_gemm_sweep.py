"""Vision-tower GEMM shapes (B = 8, two images): every tile config (hybrid schedule) against the auto choice, with and without the fc1 epilogue."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
BF = torch.bfloat16
def bench(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
shapes = [(4096, 1152, 4304, 0), (4176, 1024, 4096, 0), (4096, 4304, 1152, 1), (4096, 4304, 1152, 0), (4176, 4096, 1024, 1), (4176, 4096, 1024, 0),
          (4096, 1152, 3456, 0), (4096, 1152, 1152, 0), (4096, 3456, 1152, 0), (4176, 1024, 1024, 0), (4176, 1024, 3072, 0), (4176, 3072, 1024, 0)]
tiles = [int(t) for t in sys.argv[1:]] or [0, 101, 102, 105, 117, 3]
for M, N, K, act in shapes:
    a = torch.randn(M, K, device=dev).to(BF); b = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    t = torch.randn(M, 32, device=dev).to(BF); lb = torch.randn(N, 32, device=dev).to(BF)
    bias = torch.randn(N, device=dev).to(BF)
    out = torch.empty(M, N, device=dev, dtype=BF); pre = torch.empty(M, N, device=dev, dtype=BF)
    kw = dict(a2=t, b2=lb, bias=bias)
    if act: kw.update(act=1, c_pre=pre)
    fl = 2.0 * M * N * (K + 32)
    res = []
    for tile in tiles:
        try:
            us = bench(lambda: ops.gemm(a, b, out=out, tile=tile, **kw))
            res.append(f"t{tile}: {us:6.1f} ({fl / us / 1e6:4.0f})")
        except Exception as e:
            res.append(f"t{tile}: ERR")
    print(f"{M} {N} {K} act{act} | " + " | ".join(res), flush=True)
