"""emip_mdta_attn (Gram + softmax of the MDTA injector) at the benchmark's shapes: B images x 2 heads x 1936 pixels"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emip_amd import ops, _lib

def timed(fn, reps=20, iters=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(iters): g.replay()
        b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * iters)

for B in (8, 16):
    P, C, heads = 1936, 128, 2
    q = torch.randn(B, P, C, device="cuda").to(torch.bfloat16)
    kv = torch.randn(B, P, 2 * C, device="cuda").to(torch.bfloat16)
    temp = torch.ones(heads, device="cuda")
    rec = []
    _lib.profile(rec)
    ops.mdta_attn(q, kv[..., :C], temp, B, heads, P)
    _lib.profile(None)
    print(f"B{B}: emip_mdta_attn {timed(lambda: ops.mdta_attn(q, kv[..., :C], temp, B, heads, P)):.1f} us per call (zero + gram + softmax)")
