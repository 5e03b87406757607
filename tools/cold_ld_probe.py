"""Cold-weight gate|up / qkv at M = 608 with the weight rows padded (ldb = K + pad): does the row stride (8 KB: a power of two) cost DRAM channel conflicts?"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
M, D, F, L = 608, 4096, 11008, 8
g = torch.Generator(device="cpu").manual_seed(0)
for name, (N, K) in dict(qkv=(3 * D, D), gate_up=(2 * F, D), down=(D, F)).items():
    x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    line = "%-8s" % name
    for pad in (0, 64, 128, 192, 256, 2048):
        Ws = [torch.randn(N, K + pad, device=dev).to(torch.bfloat16)[:, :K] for _ in range(L)]
        def run():
            for W in Ws:
                ops.gemm(x, W, tile=101)
        run(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(gr, stream=s):
                run()
        gr.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(5):
                gr.replay()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 5 / L)
        line += " | pad %4d %6.1f" % (pad, best * 1e6)
        del Ws, gr
    print(line, flush=True)
