"""Per-tile fixed cost of the 256x256 config: one exact round (4096 x 4096 outputs = 256 tiles on 256 CUs) at growing K;
time = fixed + slope * K.  Also two and four rounds (N = 8192, 16384) to see what a round boundary costs."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
def bench(fn, iters=40):
    for _ in range(15): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for N in (4096, 8192, 16384):
    pts = []
    for K in (1024, 2048, 4096, 8192):
        a = torch.randn(4096, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
        out = torch.empty(4096, N, device=dev, dtype=torch.bfloat16)
        us = bench(lambda: ops.gemm(a, b, out=out, tile=17))
        pts.append((K, us))
    (k1, t1), (k2, t2) = pts[-2], pts[-1]
    slope = (t2 - t1) / (k2 - k1)
    print(f"N={N} ({N // 256 * 16} tiles = {N // 4096} rounds): " + "  ".join(f"K={k}: {t:6.1f} us ({2.0 * 4096 * N * k / t / 1e6:5.0f} TF)" for k, t in pts) +
          f" | slope {slope * 1e3:.2f} ns/K -> {2.0 * 4096 * N / slope / 1e6:5.0f} TF asymptotic, fixed {t2 - slope * k2:5.1f} us = {(t2 - slope * k2) / (N // 4096):4.1f} us/round", flush=True)
