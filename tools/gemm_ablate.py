"""Timing-only ablations of gemm_nt (results are wrong by construction): where do the main loop's cycles go?
tile + 1000: no global->LDS traffic after the first two K tiles; tile + 2000: every K tile re-reads LDS buffer 0 only;
tile + 4000: every workgroup stages tile (0, 0) (global loads always hit L2)."""
import importlib, os, subprocess, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
# the ablation branches exist only in the -DOVLA_GEMM_ABLATE build (csrc/build.sh ablate); the product library rejects tile >= 1000
subprocess.run(["bash", str(ROOT / "openvla-oft_amd" / "csrc" / "build.sh"), "ablate"], check=True)
os.environ["OVLA_LIB_NAME"] = "libovla_hip_ablate.so"
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
for (m, n, k) in [(4096, 4096, 4096), (4096, 8192, 4096), (4096, 4096, 4096), (4096, 4096, 8192)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16); b = torch.randn(n, k, device=dev).to(torch.bfloat16)
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * m * n * k
    row = []
    for t in (17, 19, 16):
        ms = bench(lambda: ops.gemm(a, b, out=out, tile=t))
        row.append(f"tile{t}: {fl / ms / 1e9:6.0f}")
    print(m, n, k, " | ".join(row), flush=True)
