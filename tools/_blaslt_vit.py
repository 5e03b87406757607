import torch
dev = torch.device("cuda:0"); BF = torch.bfloat16
def bench(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, N, K in [(4096, 1152, 4304), (4176, 1024, 4096), (4096, 4304, 1152), (4176, 4096, 1024), (4096, 1152, 3456), (4096, 1152, 1152), (4096, 3456, 1152), (4176, 1024, 1024), (4176, 1024, 3072), (4176, 3072, 1024)]:
    a = torch.randn(M, K, device=dev).to(BF); b = (torch.randn(N, K, device=dev) * 0.05).to(BF); bias = torch.randn(N, device=dev).to(BF)
    us = bench(lambda: torch.nn.functional.linear(a, b, bias))
    us2 = bench(lambda: torch.matmul(a, b.T))
    print(M, N, K, f"linear+bias {us:6.1f} us {2.0*M*N*K/us/1e6:5.0f} TF | matmul {us2:6.1f} us {2.0*M*N*K/us2/1e6:5.0f} TF", flush=True)
