"""Which op gives run-to-run different results when another stream keeps the GPU busy?  (Each op is deterministic by construction
except the fp32-atomic weight-gradient GEMMs; a difference here means a kernel reads something it did not write -- stale LDS,
registers, or memory past its operands.)  SigLIP-tower shapes by default."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
BF = torch.bfloat16
torch.manual_seed(0)
def rnd(*shape, scale=1.0): return (torch.randn(*shape, device=dev) * scale).to(BF)
side = torch.cuda.Stream()
big_a, big_b = rnd(4864, 4096), rnd(11008, 4096, scale=0.02)
dis_q = rnd(8 * 608, 3 * 4096)
def disturb():
    with torch.cuda.stream(side):
        for _ in range(3):
            ops.gemm(big_a, big_b)
            ops.attn_fwd(dis_q[:, :4096], dis_q[:, 4096:8192], dis_q[:, 8192:], 8, 608, 32, 128)
def check(name, fn, reps=12):
    ref = None; bad = 0
    for r in range(reps):
        disturb()
        out = fn()
        out = out if isinstance(out, (tuple, list)) else (out,)
        out = [o.clone() for o in out if isinstance(o, torch.Tensor)]
        torch.cuda.synchronize()
        if ref is None: ref = out
        elif any(not torch.equal(a.view(torch.uint8) if a.dtype != torch.float32 else a, b.view(torch.uint8) if b.dtype != torch.float32 else b) for a, b in zip(out, ref)): bad += 1
    print(f"{'DIFFERS' if bad else 'ok     '} {name} ({bad}/{reps - 1})", flush=True)
M = 4096
x1152, x4304 = rnd(M, 1152), rnd(M, 4304)
w_fc1, w_fc2, w_qkv, w_proj = rnd(4304, 1152, scale=0.03), rnd(1152, 4304, scale=0.03), rnd(3456, 1152, scale=0.03), rnd(1152, 1152, scale=0.03)
t32, b4304, b1152, b3456 = rnd(M, 32), rnd(4304, 32, scale=0.1), rnd(1152, 32, scale=0.1), rnd(3456, 32, scale=0.1)
bias4304, bias1152 = rnd(4304), rnd(1152)
pre = torch.empty(M, 4304, device=dev, dtype=BF)
res = rnd(M, 1152)
check("gemm fc1 fwd (gelu, c_pre, lora)", lambda: ops.gemm(x1152, w_fc1, bias=bias4304, act=1, c_pre=pre, a2=t32, b2=b4304))
check("gemm fc2 fwd (residual, lora)", lambda: ops.gemm(x4304, w_fc2, bias=bias1152, residual=res, a2=t32, b2=b1152))
check("gemm qkv fwd", lambda: ops.gemm(x1152, w_qkv, a2=t32, b2=b3456))
check("gemm proj", lambda: ops.gemm(x1152, w_proj, a2=t32, b2=b1152))
wt_fc1, wt_fc2 = ops.transpose(w_fc1), ops.transpose(w_fc2)
check("gemm fc1 dgrad (N=1152,K=4304)", lambda: ops.gemm(x4304, wt_fc1, a2=t32, b2=b1152))
check("gemm fc2 dgrad (N=4304,K=1152)", lambda: ops.gemm(x1152, wt_fc2, a2=t32, b2=b4304))
a32 = rnd(32, 1152, scale=0.05); a32b = rnd(32, 4304, scale=0.05)
check("skinny t (N=32,K=1152)", lambda: ops.gemm(x1152, a32))
check("skinny t (N=32,K=4304)", lambda: ops.gemm(x4304, a32b))
w_ln, b_ln = rnd(1152), rnd(1152)
check("norm_fwd 1152", lambda: ops.norm_fwd(x1152, w_ln, b_ln, eps=1e-6, rms=False, save_stats=True))
y, mean, rstd = ops.norm_fwd(x1152, w_ln, b_ln, eps=1e-6, rms=False, save_stats=True)
dy = rnd(M, 1152, scale=0.1)
check("norm_bwd 1152", lambda: ops.norm_bwd(x1152, dy, w_ln, mean, rstd, rms=False))
acc = rnd(M, 1152)
def nb_acc():
    d = acc.clone()
    return ops.norm_bwd(x1152, dy, w_ln, mean, rstd, rms=False, dx=d, dx_accum=True)
check("norm_bwd 1152 (dx_accum)", nb_acc)
check("act_bwd gelu 4304", lambda: ops.act_bwd(pre, x4304, 1))
B, S, H, hd = 16, 256, 16, 72
qkv = rnd(B * S, 3 * H * hd)
q, k, v = qkv[:, : H * hd], qkv[:, H * hd: 2 * H * hd], qkv[:, 2 * H * hd:]
check("attn_fwd hd72 S256", lambda: ops.attn_fwd(q, k, v, B, S, H, hd))
o, lse = ops.attn_fwd(q, k, v, B, S, H, hd)
do = rnd(B * S, H * hd, scale=0.1)
def ab():
    d = torch.empty_like(qkv)
    ops.attn_bwd(q, k, v, o, do, lse, B, S, H, hd, dq=d[:, : H * hd], dk=d[:, H * hd: 2 * H * hd], dv=d[:, 2 * H * hd:])
    return d
check("attn_bwd hd72 S256", ab)
B2, S2, H2, hd2 = 16, 261, 16, 64
qkv2 = rnd(B2 * S2, 3 * H2 * hd2)
q2, k2, v2 = qkv2[:, : H2 * hd2], qkv2[:, H2 * hd2: 2 * H2 * hd2], qkv2[:, 2 * H2 * hd2:]
o2, lse2 = ops.attn_fwd(q2, k2, v2, B2, S2, H2, hd2)
do2 = rnd(B2 * S2, H2 * hd2, scale=0.1)
def ab2():
    d = torch.empty_like(qkv2)
    ops.attn_bwd(q2, k2, v2, o2, do2, lse2, B2, S2, H2, hd2, dq=d[:, : H2 * hd2], dk=d[:, H2 * hd2: 2 * H2 * hd2], dv=d[:, 2 * H2 * hd2:])
    return d
check("attn_bwd hd64 S261 (DINOv2)", ab2)
def tn_case(name, M, P, Q):
    x, y = rnd(M, P, scale=0.5), rnd(M, Q, scale=0.5)
    outs = []
    for r in range(8):
        disturb()
        out = torch.zeros(P, Q, device=dev)
        ops.gemm_tn_grouped([(x, y, out)])
        torch.cuda.synchronize()
        outs.append(out.clone())
    ref = (x.float().T @ y.float())
    d = max(((o - outs[0]).norm() / outs[0].norm()).item() for o in outs[1:])
    e = ((outs[0] - ref).norm() / ref.norm()).item()
    print(f"gemm_tn {name} M={M} P={P} Q={Q}: run-to-run rel-L2 {d:.3e}, vs fp32 torch {e:.3e}", flush=True)
tn_case("fc1 dB", 4096, 4304, 32); tn_case("fc1 dA", 4096, 32, 1152); tn_case("qkv dB", 4096, 3456, 32); tn_case("fc2 dA", 4096, 32, 4304)
tn_case("dino fc1 dB", 4176, 4096, 32); tn_case("dino fc1 dA", 4176, 32, 1024)
# a cancelling case: columns of y sum to ~0 against a large common component of x
xc = (torch.randn(4096, 1152, device=dev) * 0.05 + 3.0).to(BF); yc = rnd(4096, 32, scale=0.5); yc -= yc.float().mean(0, keepdim=True).to(BF)
outs = []
for r in range(8):
    disturb(); out = torch.zeros(32, 1152, device=dev); ops.gemm_tn_grouped([(yc, xc, out)]); torch.cuda.synchronize(); outs.append(out.clone())
print("cancelling dA: run-to-run rel-L2", max(((o - outs[0]).norm() / outs[0].norm()).item() for o in outs[1:]))
