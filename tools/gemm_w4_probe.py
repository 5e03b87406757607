"""4-wave 256x256 config (tile 118, hand-scheduled K loop) against the 8-wave one (117) and hipBLASLt (torch.matmul) on the decoder's training shapes, with
the epilogues the model uses: LoRA K-extension (K2 = 32), residual, RoPE.  rel-L2 against tile 117's output and time (alternated, best of 3)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
S = 608
shapes = [("sq4096 plain", 4096, 4096, 4096, ""), ("sq4096 lora", 4096, 4096, 4096, "l"), ("qkv lora+rope", 4864, 12288, 4096, "lr"), ("o lora+res", 4864, 4096, 4096, "lR"),
          ("gate_up lora", 4864, 22016, 4096, "l2"), ("down lora+res", 4864, 4096, 11008, "lR"), ("d_gate_up lora", 4864, 4096, 22016, "l"), ("d_qkv lora", 4864, 4096, 12288, "l"),
          ("d_down lora", 4864, 11008, 4096, "l"), ("edge 300x520x192 lora+res", 300, 520, 192, "lR")]
tiles = [int(x) for x in sys.argv[1:]] or [117, 118]


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


cos, sin = ops.rope_table(S, 128, 10000.0, dev)
for name, M, N, K, opt in shapes:
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    b = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    kw = {}
    if "l" in opt:
        G = 3 if "r" in opt else (2 if "2" in opt else 1)
        kw.update(a2=(torch.randn(M, 32 * G, generator=g) * 0.5).to(torch.bfloat16).to(dev), b2=(torch.randn(N, 32, generator=g) * 0.05).to(torch.bfloat16).to(dev))
        if G > 1:
            kw.update(k2_group_n=N // G)
    if "R" in opt:
        kw.update(residual=(torch.randn(M, N, generator=g) * 0.5).to(torch.bfloat16).to(dev))
    if "r" in opt:
        kw.update(rope=(cos, sin, S, 8192))
    fl = 2.0 * M * N * (K + (32 if "l" in opt else 0))
    us = bench(lambda: torch.matmul(a, b.T))
    line = "%-26s torch %7.1f us %5.0f TF" % (name, us, 2.0 * M * N * K / us / 1e6)
    ref = None
    res = {t: [] for t in tiles}
    for rep in range(2):
        for t in tiles:
            y = ops.gemm(a, b, tile=t, **kw).float()
            if ref is None:
                ref = y
            err = ((y - ref).norm() / ref.norm()).item()
            res[t].append((bench(lambda: ops.gemm(a, b, tile=t, **kw)), err))
    for t in tiles:
        us = min(r[0] for r in res[t])
        line += " | t%d %7.1f us %5.0f TF err %.1e" % (t, us, fl / us / 1e6, max(r[1] for r in res[t]))
    print(line, flush=True)
