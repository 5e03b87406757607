"""Batch-1 (M = 608) projections on COLD weights: each shape runs over 8 distinct weight buffers in rotation (1.4 GB for gate|up: nothing survives in
the 256 MB memory-side cache), per tile config / split.  us per GEMM.   python tools/cold_gemm_probe.py"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vo = importlib.import_module("openvla-oft_amd")
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
M, D, F, L = 608, 4096, 11008, 8
g = torch.Generator(device="cpu").manual_seed(0)
shapes = dict(qkv=(3 * D, D), o=(D, D), gate_up=(2 * F, D), down=(D, F))
cfgs = [(0, 1), (101, 1), (1, 3), (122, 1), (22, 1), (22, 2), (22, 3), (118, 1)] if "quick" in sys.argv else [(0, 1), (1, 1), (101, 1), (14, 1), (14, 2), (14, 3), (17, 1), (117, 1), (20, 1), (21, 1), (10, 1), (15, 1), (1, 2), (1, 3), (2, 1), (102, 1)]
for name, (N, K) in shapes.items():
    Ws = [(torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).to(dev) for _ in range(L)]
    x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    ref = None
    line = "%-8s" % name
    for tile, sk in cfgs:
        try:
            def run():
                for W in Ws:
                    ops.gemm(x, W, tile=tile, split_k=sk)
            y = ops.gemm(x, Ws[0], tile=tile, split_k=sk).float()
            if ref is None:
                ref = x.float() @ Ws[0].float().T
            err = ((y - ref).norm() / ref.norm()).item()
            run(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
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
            line += " | t%d/s%d %6.1f%s" % (tile, sk, best * 1e6, "" if err < 1e-2 else " ERR%.1e" % err)
        except Exception as e:
            line += " | t%d/s%d  n/a" % (tile, sk)
    print(line, flush=True)
    del Ws
