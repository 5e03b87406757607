"""Timing-only ablations of the 256x256 gemm_nt main loop at exact-round shapes (results are wrong by construction): what do the LDS fragment
reads cost?  tile + 128000: no fragment reads after the first K tile; + 256000: no B fragment reads; + 512000: no A fragment reads; + 1000: no
global -> LDS staging after the first two K tiles; + 8000: no epilogue."""
import importlib, os, subprocess, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
subprocess.run(["bash", str(ROOT / "openvla-oft_amd" / "csrc" / "build.sh"), "ablate"], check=True)
os.environ["OVLA_LIB_NAME"] = "libovla_hip_ablate.so"
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
def bench(fn, iters=30):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (m, n, k) in [(4096, 4096, 4096), (4096, 12288, 4096), (4096, 4096, 11008)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16); b = torch.randn(n, k, device=dev).to(torch.bfloat16)
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * m * n * k
    for rep in range(2):
        row = []
        for name, t in (("full", 17), ("no-frag-reads", 128017), ("no-B-reads", 256017), ("no-A-reads", 512017), ("no-staging", 1017), ("no-epilogue", 8017),
                        ("no-reads+no-staging", 129017), ("no-reads+no-staging+no-epi", 137017)):
            ms = bench(lambda: ops.gemm(a, b, out=out, tile=t))
            row.append(f"{name} {fl / ms / 1e9:5.0f}")
    ms_t = bench(lambda: torch.matmul(a, b.t(), out=out))
    print(m, n, k, " | ".join(row), f"| hipBLASLt {fl / ms_t / 1e9:5.0f}", flush=True)
