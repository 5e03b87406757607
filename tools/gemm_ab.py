"""Same-call A/B of two builds of the library on the step's big GEMM shapes: the product libovla_hip.so vs libovla_hip_exp.so (csrc/build.sh exp
with OVLA_EXP_FLAGS), alternated, each in its own process (a library is loaded once per process)."""
import importlib, os, subprocess, sys, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
SHAPES = [("qkv", 4864, 12288, 4096), ("o", 4864, 4096, 4096), ("gate_up", 4864, 22016, 4096), ("down", 4864, 4096, 11008), ("d_gate_up", 4864, 4096, 22016),
          ("sq4096", 4096, 4096, 4096), ("x12288", 4096, 12288, 4096)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, str(ROOT))
    ops = importlib.import_module("openvla-oft_amd.ops")
    dev = torch.device("cuda:0")
    out = {}
    for name, m, n, k in SHAPES:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16); b = torch.randn(n, k, device=dev).to(torch.bfloat16)
        c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        for _ in range(10): ops.gemm(a, b, out=c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): ops.gemm(a, b, out=c)
        e1.record(); torch.cuda.synchronize()
        out[name] = 2.0 * m * n * k / (e0.elapsed_time(e1) / 40) / 1e9
    print(json.dumps(out))
    sys.exit(0)
res = {"libovla_hip.so": [], "libovla_hip_exp.so": []}
for rep in range(3):
    for lib in res:
        r = subprocess.run([sys.executable, __file__, "child"], env={**os.environ, "OVLA_LIB_NAME": lib}, capture_output=True, text=True)
        res[lib].append(json.loads(r.stdout.strip().splitlines()[-1]))
for name, *_ in SHAPES:
    a = [f"{r[name]:5.0f}" for r in res["libovla_hip.so"]]; b = [f"{r[name]:5.0f}" for r in res["libovla_hip_exp.so"]]
    print(f"{name:10s} product {' '.join(a)} | exp {' '.join(b)} TF")
