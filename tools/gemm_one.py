"""Runs ONE gemm_nt shape a few times (for rocprofv3 --pmc passes: HBM traffic of the dominant kernel)."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
m, n, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (4864, 22016, 4096)
a = torch.randn(m, k, device=dev).to(torch.bfloat16); b = torch.randn(n, k, device=dev).to(torch.bfloat16)
out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
for _ in range(6):
    ops.gemm(a, b, out=out)
torch.cuda.synchronize()
print("algorithmic bytes", (m * k + n * k + m * n) * 2)
