"""Like gemm_one.py, but the weight operand rotates over 8 distinct buffers (nothing of a launch's B survives in the 256 MB memory-side cache):
the batch-1 decoder's condition.  For the rocprofv3 --pmc passes of tools/pmc_cold.sh."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
m, n, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (608, 22016, 4096)
tile = int(sys.argv[4]) if len(sys.argv) > 4 else 0
a = torch.randn(m, k, device=dev).to(torch.bfloat16)
bs = [torch.randn(n, k, device=dev).to(torch.bfloat16) for _ in range(8)]
out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
for _ in range(2):
    for b in bs:
        ops.gemm(a, b, out=out, tile=tile)
torch.cuda.synchronize()
print("algorithmic bytes", (m * k + n * k + m * n) * 2)
