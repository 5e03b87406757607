"""Exact-round probe for the 304 x 256 tile question: M = 4096 (16 x 256: one tile per CU at N = 4096, whole rounds at 12288) on the BK = 64
two-stage kernel (tile 17) next to the BK = 32 LDS rings (tile 15: 3 stages, tile 10: 4 stages) -- what a tile that needs BK = 32 stages to fit
under the 128 KiB LDS-DMA limit would give up in the main loop -- and M = 4864 on the hybrid schedule for reference."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
def bench(fn, iters=30):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tiles = [int(t) for t in sys.argv[1:]] or [17, 15, 10, 18]
for M in (4096, 4864):
    for name, n, k in [("qkv", 12288, 4096), ("o", 4096, 4096), ("down", 4096, 11008), ("d_gate_up", 4096, 22016)]:
        a = torch.randn(M, k, device=dev).to(torch.bfloat16); b = torch.randn(n, k, device=dev).to(torch.bfloat16)
        out = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
        fl = 2.0 * M * n * k
        res = []
        for _ in range(2):   # second pass is the one reported (clocks settled)
            res = []
            for t in [0] + tiles:
                try:
                    us = bench(lambda: ops.gemm(a, b, out=out, tile=t))
                    res.append(f"t{t} {us:6.1f}us {fl / us / 1e6:5.0f}TF")
                except Exception as e:
                    res.append(f"t{t} n/a")
            us_t = bench(lambda: torch.matmul(a, b.t(), out=out))
        print(f"M={M} {name:10s} | torch {us_t:6.1f}us {fl / us_t / 1e6:5.0f}TF | " + " | ".join(res), flush=True)
