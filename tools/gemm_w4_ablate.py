"""Timing-only ablations of the 4-wave GEMM's K loop (csrc/build.sh ablate; OVLA_LIB_NAME=libovla_hip_ablate.so): tile 18 = the kernel, 218 = no staging after
the prologue, 318 = no fragment reads, 418 = neither (MFMAs + barriers only); 17 = the 8-wave kernel.  us per K tile from the K = 4096 / K = 11008 slope at M = N = 4096."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")


def bench(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3


M = N = 4096
ab = {K: ((torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16), (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)) for K in (4096, 11008)}
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for rep in range(2):
    for tile in (17, 18, 218, 318, 418, 518, 618, 718):
        t = {K: bench(lambda: ops.gemm(ab[K][0], ab[K][1], out=out, tile=tile)) for K in ab}
        tau = (t[11008] - t[4096]) / (172 - 64)
        print("tile %3d: K=4096 %6.1f us  K=11008 %6.1f us  -> %.3f us per K tile (%.0f TFLOP/s in the loop), %.1f us outside the loop" % (
            tile, t[4096], t[11008], tau, 2 * 256 * 256 * 64 * 256 / tau / 1e6, t[4096] - 64 * tau), flush=True)
