"""4-wave 256x256 config (tile 18 / 118 = with the hybrid remainder schedule) against the 8-wave one (17 / 117) and hipBLASLt (torch.matmul): correctness
(rel-L2 against an fp32 product of the same bf16 operands) and time, on whole-round shapes and on the decoder's training shapes."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
shapes = [("4096x4096x4096", 4096, 4096, 4096), ("4096x8192x4096", 4096, 8192, 4096), ("4096x4096x11008", 4096, 4096, 11008),
          ("qkv", 4864, 12288, 4096), ("o", 4864, 4096, 4096), ("gate_up", 4864, 22016, 4096), ("down", 4864, 4096, 11008), ("d_gate_up", 4864, 4096, 22016),
          ("edge 300x520x192", 300, 520, 192)]
tiles = [int(x) for x in sys.argv[1:]] or [17, 18, 117, 118]


def bench(fn, reps=20):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3


for name, M, N, K in shapes:
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    b = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    res = (torch.randn(M, N, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    ref = a.float() @ b.float().T
    fl = 2.0 * M * N * K
    us = bench(lambda: torch.matmul(a, b.T))
    line = "%-18s torch %7.1f us %5.0f TF" % (name, us, fl / us / 1e6)
    for t in tiles:
        try:
            y = ops.gemm(a, b, tile=t)
            err = ((y.float() - ref).norm() / ref.norm()).item()
            y2 = ops.gemm(a, b, tile=t, residual=res)
            err2 = ((y2.float() - (ref.to(torch.bfloat16).float() + res.float())).norm() / ref.norm()).item()
            us = bench(lambda: ops.gemm(a, b, tile=t))
            line += " | t%d %7.1f us %5.0f TF err %.1e/%.1e" % (t, us, fl / us / 1e6, err, err2)
        except Exception as e:
            line += " | t%d n/a (%s)" % (t, str(e)[:40])
    print(line, flush=True)
