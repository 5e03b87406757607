"""Batch-1 inference ViT / projector GEMM shapes (M ~ 512): which tile / split is fastest (incl. its reduce kernel)."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
def bench(fn, iters=60):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
shapes = [("d_qkv", 522, 3072, 1024), ("d_proj", 522, 1024, 1024), ("d_fc1", 522, 4096, 1024), ("d_fc2", 522, 1024, 4096),
          ("s_qkv", 512, 3456, 1152), ("s_proj", 512, 1152, 1152), ("s_fc1", 512, 4304, 1152), ("s_fc2", 512, 1152, 4304),
          ("p_fc1", 512, 8704, 2176), ("p_fc2", 512, 4096, 8704), ("p_fc3", 512, 4096, 4096)]
for name, M, n, k in shapes:
    a = torch.randn(M, k, device=dev).to(torch.bfloat16); b = torch.randn(n, k, device=dev).to(torch.bfloat16)
    bias = torch.randn(n, device=dev).to(torch.bfloat16)
    out = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    res = []
    for tile in (1, 101, 2, 5):
        for sk in (1, 2, 4):
            if tile == 101 and sk > 1: continue
            if sk * 4 > (k + 63) // 64 * 1 and sk > 1 and k < 1024: continue
            try:
                res.append((bench(lambda: ops.gemm(a, b, out=out, bias=bias, tile=tile, split_k=sk)), f"t{tile}/s{sk}"))
            except Exception:
                pass
    auto = bench(lambda: ops.gemm(a, b, out=out, bias=bias))
    tm = bench(lambda: torch.nn.functional.linear(a, b, bias))
    best = min(res)
    print(f"{name:7s} M={M} N={n} K={k} auto {auto:6.1f} torch {tm:6.1f} best {best[1]} {best[0]:6.1f} | " + " ".join(f"{l}:{t:5.1f}" for t, l in res), flush=True)
