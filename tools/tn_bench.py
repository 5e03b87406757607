"""LoRA weight-gradient launches of one Llama layer (gemm_tn_grouped: dB_g += dy_g^T t_g per fused group, dA += dt^T x) vs their HBM floor."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0"); BF = torch.bfloat16
def bench(fn, iters=40):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
r = 32
for name, M, inn, out, G in [("qkv", 4864, 4096, 12288, 3), ("o", 4864, 4096, 4096, 1), ("gate_up", 4864, 4096, 22016, 2), ("down", 4864, 11008, 4096, 1),
                             ("vit qkv", 4176, 1024, 3072, 1), ("vit proj", 4176, 1024, 1024, 1), ("vit fc1", 4176, 1024, 4096, 1), ("vit fc2", 4176, 4096, 1024, 1),
                             ("sig fc1", 4096, 1152, 4304, 1)]:
    gn = out // G
    x = torch.randn(M, inn, device=dev).to(BF); dy = torch.randn(M, out, device=dev).to(BF)
    t = torch.randn(M, G * r, device=dev).to(BF); dt = torch.randn(M, G * r, device=dev).to(BF)
    gB = torch.zeros(out, r, device=dev); gA = torch.zeros(G * r, inn, device=dev)
    probs = [(dy[:, g * gn:(g + 1) * gn], t[:, g * r:(g + 1) * r], gB[g * gn:(g + 1) * gn]) for g in range(G)] + [(dt, x, gA)]
    us = bench(lambda: ops.gemm_tn_grouped(probs))
    byt = (x.numel() + dy.numel() + t.numel() + dt.numel()) * 2
    print(f"{name:8s} {us:7.1f} us  bytes {byt / 1e6:6.0f} MB -> {byt / us / 1e6:5.2f} TB/s  (floor at 5 TB/s: {byt / 5e6:5.1f} us)", flush=True)
