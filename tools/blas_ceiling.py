"""Calibration only (not product code): what the vendor GEMM reaches on the EMIP shapes, to know how much
headroom the hand-written kernel has."""
import torch
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
dev, dt = "cuda:0", torch.bfloat16
for M, N, K in [(15488, 1280, 320), (15488, 320, 1280), (15488, 320, 320), (61952, 512, 128), (61952, 128, 512),
                (247808, 256, 64), (247808, 64, 256), (61952, 1024, 256), (61952, 128, 1024), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    b = torch.randn(N, device=dev).to(dt)
    us = timeit(lambda: torch.nn.functional.linear(a, w, b))
    print("blas %7d %5d %5d : %8.1f us  %7.1f TF/s" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
x = torch.randn(32, 22, 22, 1280, device=dev).to(dt)
us = timeit(lambda: torch.nn.functional.layer_norm(x, (1280,)))
print("torch ln 15488x1280: %.1f us" % us)
q = torch.randn(32, 5, 484, 64, device=dev).to(dt); k = torch.randn(32, 5, 121, 64, device=dev).to(dt)
us = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, k))
print("torch sdpa sra stage3: %.1f us" % us)
