"""Times the attention kernels on the Llama (B=8,H=32,S=608,hd=128) and ViT shapes."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
def bench(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (B, H, S, hd, padded) in [(8, 32, 608, 128, True), (16, 16, 261, 64, False), (16, 16, 256, 72, False)]:
    qkv = torch.randn(B * S, 3 * H * hd, device=dev).to(torch.bfloat16)
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:]
    kv = torch.tensor([S - 4 * (i % 2) for i in range(B)], dtype=torch.int32, device=dev) if padded else None
    out, lse = ops.attn_fwd(q, k, v, B, S, H, hd, kv_len=kv)
    do = torch.randn_like(out)
    f = 4.0 * B * H * S * S * hd
    ms_f = bench(lambda: ops.attn_fwd(q, k, v, B, S, H, hd, kv_len=kv))
    ms_b = bench(lambda: ops.attn_bwd(q, k, v, out, do, lse, B, S, H, hd, kv_len=kv))
    print(f"B{B} H{H} S{S} hd{hd}: fwd {ms_f*1e3:7.1f} us {f/ms_f/1e9:6.0f} TF | bwd {ms_b*1e3:7.1f} us {2.5*f/ms_b/1e9:6.0f} TF", flush=True)
