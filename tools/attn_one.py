"""Runs the Llama-shape attention forward (and optionally backward) a few times (for rocprofv3 --pmc passes)."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
B, H, S, hd = 8, 32, 608, 128
qkv = torch.randn(B * S, 3 * H * hd, device=dev).to(torch.bfloat16)
q, k, v = qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:]
kv = torch.tensor([S - 4 * (i % 2) for i in range(B)], dtype=torch.int32, device=dev)
out, lse = ops.attn_fwd(q, k, v, B, S, H, hd, kv_len=kv)
do = torch.randn_like(out)
for _ in range(5):
    out, lse = ops.attn_fwd(q, k, v, B, S, H, hd, kv_len=kv)
    if len(sys.argv) > 1 and sys.argv[1] == "bwd":
        ops.attn_bwd(q, k, v, out, do, lse, B, S, H, hd, kv_len=kv)
torch.cuda.synchronize()
