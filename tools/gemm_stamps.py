"""In-kernel timeline of the 256x256 GEMM (library built with OVLA_EXP_FLAGS=-DOVLA_GEMM_STAMPS, csrc/build.sh exp; run with OVLA_LIB_NAME=libovla_hip_exp.so):
per workgroup, 100 MHz wall-clock stamps at entry / first LDS-DMA issue / first K tile landed / K loop done / stores acknowledged.  Where do the ~18 us
per round that are not K-loop time go -- dispatch skew, prologue latency, epilogue, or the spread of the finishing times?"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
TILE = int(sys.argv[1]) if len(sys.argv) > 1 else 17   # + 1000 * ablation bits with an OVLA_GEMM_ABLATE build
g = torch.Generator(device="cpu").manual_seed(0)
for name, M, N, K in (("4096^3 (1 round)", 4096, 4096, 4096), ("4096x8192x4096 (2 rounds)", 4096, 8192, 4096), ("4096x4096x11008 (1 round)", 4096, 4096, 11008)):
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    b = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(5):
        ops.gemm(a, b, out=out, tile=TILE)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm(a, b, out=out, tile=TILE); e1.record(); torch.cuda.synchronize()
    nblk = (M // 256) * (N // 256)
    ws = ops._workspace(dev, ops._WS_BYTES)
    st = ws.view(torch.int64)[: nblk * 8].view(nblk, 8).cpu().double() / 100.0   # us
    t0 = st[:, 0].min()
    def q(x):
        x = x.sort().values
        return "min %6.2f  p10 %6.2f  med %6.2f  p90 %6.2f  max %6.2f" % (x[0], x[int(0.1 * len(x))], x[len(x) // 2], x[int(0.9 * len(x))], x[-1])
    print(f"{name}: event time {e0.elapsed_time(e1) * 1e3:.1f} us, {nblk} workgroups, span first entry -> last store ack {float(st[:, 4].max() - t0):.1f} us")
    print("   entry (since first entry)     ", q(st[:, 0] - t0))
    print("   entry -> first DMA issued     ", q(st[:, 1] - st[:, 0]))
    print("   first DMA -> first tile landed", q(st[:, 2] - st[:, 1]))
    print("   K loop                        ", q(st[:, 3] - st[:, 2]))
    print("   epilogue (to store ack)       ", q(st[:, 4] - st[:, 3]))
    print("   finish (since first entry)    ", q(st[:, 4] - t0))
    if nblk > 256:   # second round: entries after the first finish
        late = st[:, 0] - t0 > 20
        print("   second-round workgroups: %d, entry" % int(late.sum()), q((st[:, 0] - t0)[late]))
    loop = st[:, 3] - st[:, 2]
    bid = torch.arange(nblk)
    print("   K loop by XCD (block id & 7):  " + "  ".join("%d: %.1f" % (x, float(loop[(bid & 7) == x].mean())) for x in range(8)))
    if nblk == 256:
        idx = (bid >> 3)
        print("   K loop by position in the XCD's band (block id >> 3), groups of 4: " + "  ".join("%.1f" % float(loop[(idx // 4) == k].mean()) for k in range(8)))
        srt = loop.sort()
        print("   slowest 12 blocks:", [(int(b), int(b) & 7, round(float(v), 1)) for v, b in zip(srt.values[-12:], srt.indices[-12:])])
        print("   fastest 12 blocks:", [(int(b), int(b) & 7, round(float(v), 1)) for v, b in zip(srt.values[:12], srt.indices[:12])])
