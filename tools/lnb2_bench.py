"""LayerNorm backward with the residual-gradient add (emip_layernorm_bwd_res) on the training shapes, bf16."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emip_amd import _lib, ops
_lib.load()
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for M, C in [(30976, 320), (123904, 128), (495616, 64), (7744, 512)]:
    x = torch.randn(M, C, device="cuda").to(torch.bfloat16); dy = torch.randn_like(x); dr = torch.randn_like(x)
    g = torch.ones(C, device="cuda")
    t0 = timeit(lambda: ops.layernorm_bwd_fresh(x, dy, g, 1e-6))
    t1 = timeit(lambda: ops.layernorm_bwd_fresh(x, dy, g, 1e-6, dres=dr))
    print("%7d x %4d: plain %.1f us (%.2f TB/s), + residual %.1f us (%.2f TB/s)" % (M, C, t0, 3 * M * C * 2 / t0 / 1e6, t1, 4 * M * C * 2 / t1 / 1e6))
