"""LoRA backward at the fine-tune shapes (M = 4864): the one-pass kernel (ovla_lora_bwd: dt + dB from one staging of dy, then dA by the TN
GEMM) next to the three-kernel path it replaces (block-diagonal skinny NT GEMM + split-K reduce for dt, grouped TN GEMM for dB_g and dA).
Rotating operand sets larger than the 256 MB Infinity Cache keep the reads cold, as inside the step."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4864
def bench(fn, iters=24):
    for i in range(6): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
shapes = [("qkv", 4096, 3, 4096), ("o", 4096, 1, 4096), ("gate_up", 11008, 2, 4096), ("down", 4096, 1, 11008)] if M > 4500 else \
         [("vit_qkv", 3072, 1, 1024), ("vit_fc1", 4096, 1, 1024), ("vit_fc2", 1024, 1, 4096), ("sig_fc1", 4304, 1, 1152), ("sig_fc2", 1152, 1, 4304)]
tot = [0.0, 0.0]
for name, gn, G, K in shapes:
    r, s, NSET = 32, 0.5, 4
    sets = [dict(dy=torch.randn(M, G * gn, device=dev).to(BF), t=torch.randn(M, G * r, device=dev).to(BF), x=torch.randn(M, K, device=dev).to(BF)) for _ in range(NSET)]
    Bt = (torch.randn(G * r, gn, device=dev) * 0.1).to(BF)
    dB, dA = torch.zeros(G * gn, r, device=dev), torch.zeros(G * r, K, device=dev)
    def old(i):
        d = sets[i % NSET]
        dt = ops.gemm(d["dy"], Bt, alpha=s, a_group_n=r if G > 1 else 0)
        probs = [(d["dy"][:, g * gn:(g + 1) * gn], d["t"][:, g * r:(g + 1) * r], dB[g * gn:(g + 1) * gn]) for g in range(G)] + [(dt, d["x"], dA)]
        ops.gemm_tn_grouped(probs)
    def new(i):
        d = sets[i % NSET]
        dt = ops.lora_bwd(d["dy"], Bt, d["t"], dB, gn=gn, G=G, scale=s)
        ops.gemm_tn_grouped([(dt, d["x"], dA)])
    def new_only(i):
        d = sets[i % NSET]
        ops.lora_bwd(d["dy"], Bt, d["t"], dB, gn=gn, G=G, scale=s)
    a, b, c = bench(old), bench(new), bench(new_only)
    mb = M * G * gn * 2 / 1e6
    tot[0] += a; tot[1] += b
    print(f"{name:8s} dy {mb:6.1f} MB | three-kernel path {a:6.1f} us | one-pass + dA TN {b:6.1f} us (one-pass alone {c:6.1f} us = {mb / c * 1e-3 * 1e3:5.2f} TB/s of dy)", flush=True)
print(f"sum: {tot[0]:.1f} -> {tot[1]:.1f} us per layer")
