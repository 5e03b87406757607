"""Which B2 row does each output column of the 4-wave GEMM's K-extension read?  lb[n, :] = n / 256, t = 1, a = 0:  y[m, n] = 32 * row / 256."""
import importlib, sys, torch
sys.path.insert(0, "/root/repo")
ops = importlib.import_module("openvla-oft_amd.ops")
dev = torch.device("cuda:0")
M, N, K = 256, 256, 128
for k2 in (32, 64):
    a = torch.zeros(M, K, dtype=torch.bfloat16, device=dev); b = torch.zeros(N, K, dtype=torch.bfloat16, device=dev)
    t = torch.ones(M, k2, dtype=torch.bfloat16, device=dev)
    lb = (torch.arange(N, dtype=torch.float32)[:, None] / 256).expand(N, k2).contiguous().to(torch.bfloat16).to(dev)
    for tile in (117, 18):
        y = ops.gemm(a, b, a2=t, b2=lb, tile=tile).float()
        rows = (y * 256 / k2).round().int()
        print("K2", k2, "tile", tile, "row 0 cols 120..140:", rows[0, 120:140].tolist(), " row 200 cols 250..255:", rows[200, 250:256].tolist(), " mismatches:", int((rows != torch.arange(N, device=dev)[None, :]).sum()))
        if tile == 18 and k2 == 32:
            bad = (rows != torch.arange(N, device=dev)[None, :]).nonzero()
            import collections
            print("  bad rows:", sorted(set(bad[:, 0].tolist()))[:40], " bad cols:", sorted(set(bad[:, 1].tolist()))[:80])
            r0 = int(bad[0, 0]); print("  first bad row", r0, "values at bad cols:", [(int(c), int(rows[r0, c])) for c in sorted(set(bad[bad[:, 0] == r0][:, 1].tolist()))][:20])
