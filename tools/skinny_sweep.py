"""Skinny LoRA GEMMs (t = x A^T, dt = dy B): tile / split-K sweep.  HBM floor = bytes(A) / 5 TB/s."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
BF = torch.bfloat16
def bench(fn, iters=40):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
cases = [(4864, 96, 4096, 0), (4864, 32, 4096, 0), (4864, 64, 4096, 0), (4864, 32, 11008, 0), (4864, 96, 4096, 32), (4864, 64, 11008, 32),
         (4176, 32, 1024, 0), (4096, 32, 1152, 0), (4176, 32, 4096, 0), (4096, 32, 4304, 0)]
for M, N, K, grp in cases:
    G = N // grp if grp else 1
    a = torch.randn(M, K * G, device=dev).to(BF); b = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    out = torch.empty(M, N, device=dev, dtype=BF)
    floor = a.numel() * 2 / 5e12 * 1e6
    res = [f"auto {bench(lambda: ops.gemm(a, b, out=out, a_group_n=grp)):6.1f}"]
    for tile in (5, 2):
        if tile == 2 and grp: continue
        for sk in (4, 8, 12, 16, 24, 32):
            if sk * 1 > (K + 63) // 64: continue
            try:
                res.append(f"t{tile}/s{sk} {bench(lambda: ops.gemm(a, b, out=out, a_group_n=grp, tile=tile, split_k=sk)):6.1f}")
            except Exception as e:
                res.append(f"t{tile}/s{sk} ERR")
    print(f"M={M} N={N} K={K} grp={grp} floor {floor:5.1f} us | " + " | ".join(res), flush=True)
