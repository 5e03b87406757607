"""A/B of the in-launch hybrid reduce (ovla_gemm_args.hybrid_counters) against the separate gemm_hybrid_reduce launch on the step's and the chunk's
hybrid-scheduled GEMM shapes, alternated in one process."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
def bench(fn, iters=40):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, m, n, k in [("o", 4864, 4096, 4096), ("down", 4864, 4096, 11008), ("qkv", 4864, 12288, 4096), ("gate_up", 4864, 22016, 4096), ("d_gate_up", 4864, 4096, 22016),
                      ("o@608", 608, 4096, 4096), ("down@608", 608, 4096, 11008), ("gate_up@608", 608, 22016, 4096), ("vit_fc2@522", 522, 1024, 4096)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16); b = torch.randn(n, k, device=dev).to(torch.bfloat16)
    res = torch.randn(m, n, device=dev).to(torch.bfloat16); out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    row = []
    for rep in range(2):
        for flag in (False, True):
            ops._HYB_INLAUNCH = flag
            row.append(bench(lambda: ops.gemm(a, b, out=out, residual=res)))
    print(f"{name:12s} plan {ops.gemm_plan(m, n, k)[:4]} | separate {row[0]:7.1f} {row[2]:7.1f} us | in-launch {row[1]:7.1f} {row[3]:7.1f} us", flush=True)
