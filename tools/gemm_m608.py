"""Batch-1 inference GEMM shapes (M = 608): tile / split-K sweep next to hipBLASLt."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
def bench(fn, iters=60):
    for _ in range(25): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M = int(sys.argv[1]) if len(sys.argv) > 1 else 608
for name, n, k in [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]:
    a = torch.randn(M, k, device=dev).to(torch.bfloat16); b = torch.randn(n, k, device=dev).to(torch.bfloat16)
    out = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * n * k
    res = [f"auto {bench(lambda: ops.gemm(a, b, out=out)):6.1f}"]
    for tile in (1, 101, 2, 17, 117):
        for sk in (1, 2, 3, 4):
            if tile in (101, 117) and sk > 1: continue
            try:
                us = bench(lambda: ops.gemm(a, b, out=out, tile=tile, split_k=sk))
            except Exception as e:
                continue
            res.append(f"t{tile}/s{sk} {us:6.1f}")
    us_t = bench(lambda: torch.matmul(a, b.t(), out=out))
    print(f"{name:8s} ideal@1.1PF {fl / 1.1e15 * 1e6:5.1f} us | torch {us_t:6.1f} | " + " | ".join(res), flush=True)
