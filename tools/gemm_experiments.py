import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emip_amd import ops, _lib
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
dev, dt = "cuda:0", torch.bfloat16
for M, N, K in [(61952, 1024, 256), (15488, 1280, 320), (15488, 320, 1280), (61952, 512, 128), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    b = torch.randn(N, device=dev); o = torch.empty(M, N, device=dev, dtype=dt)
    row = []
    for tile in (128128, 128064, 64128, 64064):
        _lib.call("emip_debug_set", 1, tile)
        for dbg in (0, 1, 2, 3):
            _lib.call("emip_debug_set", 2, dbg)
            row.append("%6.1f" % timeit(lambda: ops.gemm(a, w, bias=b, out=o)))
    _lib.call("emip_debug_set", 1, 0); _lib.call("emip_debug_set", 2, 0)
    print("%6d %5d %5d | tile x (full, noEpi, noLoad, neither):" % (M, N, K), " | ".join(" ".join(row[i:i + 4]) for i in range(0, 16, 4)))
