import importlib, sys, torch
sys.path.insert(0, "/root/repo")
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
for M, N, K in ((128, 128, 64), (256, 256, 448), (512, 256, 128), (256, 512, 128)):
    for k2 in (32, 64, 96):
        a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        b = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        t = (torch.randn(M, k2, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        lb = (torch.randn(N, k2, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        main = a.float() @ b.float().T
        ext = t.float() @ lb.float().T
        for tile in (18, 118, 117):
            y = ops.gemm(a, b, a2=t, b2=lb, tile=tile).float()
            d = y - main
            coef = ((d * ext).sum() / (ext * ext).sum()).item()
            resid = ((d - coef * ext).norm() / ext.norm()).item()
            # per 128x128 quadrant coefficient
            qs = []
            for mi in range(0, M, 128):
                for ni in range(0, N, 128):
                    e = ext[mi:mi+128, ni:ni+128]; dd = d[mi:mi+128, ni:ni+128]
                    qs.append("%.2f" % ((dd * e).sum() / (e * e).sum()).item())
            print(M, N, K, "K2", k2, "tile", tile, "ext coef %.3f resid %.3f" % (coef, resid), "quadrants", " ".join(qs), flush=True)
