"""GPU micro-benchmark: the model's GEMM shapes on gemm_nt (per tile config) next to torch.matmul (hipBLASLt), the
known-good ceiling on the same box and the same random data (cdna_hip_programming.md rules 10, 24, 25)."""
import importlib
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
M = 4864
SHAPES = [("qkv", M, 12288, 4096, 32), ("o", M, 4096, 4096, 32), ("gate_up", M, 22016, 4096, 32), ("down", M, 4096, 11008, 32),
          ("d_qkv", M, 4096, 12288, 96), ("d_gate_up", M, 4096, 22016, 64), ("d_down", M, 11008, 4096, 32),
          ("vit_fc1", 4176, 4096, 1024, 32), ("vit_fc2", 4176, 1024, 4096, 32), ("siglip_qkv", 4096, 3456, 1152, 32),
          ("vit_qkv", 4176, 3072, 1024, 32), ("vit_o", 4176, 1024, 1024, 32), ("siglip_fc1", 4096, 4304, 1152, 32),
          ("siglip_fc2", 4096, 1152, 4304, 32), ("siglip_o", 4096, 1152, 1152, 32), ("proj_fc1", 4096, 8704, 2176, 32), ("lora_t", M, 96, 4096, 0), ("lora_dt", M, 32, 4096, 0), ("head_fc1", 64, 4096, 28672, 0)]


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tiles = [int(t) for t in sys.argv[1:]] or [1, 3]
print(f"{'shape':12s} {'M':>5s} {'N':>6s} {'K':>6s} " + " ".join(f"tile{t:>2d}(TF)" for t in tiles) + "  torch(TF)")
for name, m, n, k, k2 in SHAPES:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    b = torch.randn(n, k, device=dev).to(torch.bfloat16)
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * m * n * k
    res = []
    for t in tiles:
        if t == 0:
            ms = bench(lambda: ops.gemm(a, b, out=out))
        elif m <= 64:
            ms = bench(lambda: ops.gemm(a, b, out=out, tile=2, split_k=8))
        elif n <= 128:
            sk = {1: 1, 3: 1}.get(t, 4)
            ms = bench(lambda: ops.gemm(a, b, out=out, tile=(5 if n <= 32 else 2) if t >= 10 else t, split_k=sk))
        else:
            ms = bench(lambda: ops.gemm(a, b, out=out, tile=t))
        res.append(fl / ms / 1e9)
    ms_t = bench(lambda: torch.matmul(a, b.T, out=out))
    print(f"{name:12s} {m:5d} {n:6d} {k:6d} " + " ".join(f"{r:10.0f}" for r in res) + f"  {fl / ms_t / 1e9:9.0f}")
