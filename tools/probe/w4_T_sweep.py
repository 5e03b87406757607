import importlib, sys, torch
sys.path.insert(0, "/root/repo")
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
W4TILE = int(sys.argv[1]) if len(sys.argv) > 1 else 18
g = torch.Generator(device="cpu").manual_seed(0)
for M, N in ((256, 256), (512, 512), (300, 520)):
    for T in range(1, 10):
        K = 64 * T
        a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        b = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        ref = a.float() @ b.float().T
        y = ops.gemm(a, b, tile=W4TILE).float()
        # per K tile contribution check: which tile's contribution is missing / doubled?
        d = y - ref
        coef = []
        for t in range(T):
            c = a[:, 64 * t:64 * t + 64].float() @ b[:, 64 * t:64 * t + 64].float().T
            coef.append(((d * c).sum() / (c * c).sum()).item())
        print(M, N, "T", T, "err %.2e" % (d.norm() / ref.norm()).item(), "tile coefs", " ".join("%+.2f" % c for c in coef), flush=True)
