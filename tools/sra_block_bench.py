"""emip_sra_block against the three launches it replaces, per stage shape (hipGraph of 20 launches, HIP events)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from emip_amd import ops
import test_sra_block_gpu as T


def timed(fn, reps=20, iters=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(iters):
            g.replay()
        b.record(s)
        torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * iters)


shapes = [(16, 22, 22, 320, 121), (32, 22, 22, 320, 121), (16, 44, 44, 128, 121), (16, 88, 88, 64, 121)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (B, H, W, C, Lk) in shapes:
    heads = C // 64
    x, kv, wq, wp, bq, bp, stats = T._make(B, H, W, C, Lk, 1)
    csq = wq.float().sum(1).contiguous()
    sw = ops.swap23(C, x.device)
    wqf, wpf = wq[sw].contiguous(), wp[sw][:, sw].contiguous()
    st = torch.zeros(B * H * W, 2, device=x.device)
    att = torch.empty_like(x)
    xa, xb = x.clone(), x.clone()

    def three():
        q = ops.gemm(xa, wq, bias=bq, ln_stats=stats, ln_eps=1e-6, colsum=csq)
        ops.sra_attention(q.view(B, H * W, C), kv, att.view(B, H * W, C), B, heads, H * W, Lk, 0.125)
        ops.gemm(att, wp, bias=bp, res=xa, out=xa, out_stats=st.view(-1))

    def one():
        ops.sra_block(xb, stats, 1e-6, wqf, bq, csq, kv, wqf if False else wpf, bp, heads, 0.125, out_stats=st.view(-1))

    def two():
        a = ops.sra_qattn(xa, stats, 1e-6, wqf, bq, csq, kv, heads, 0.125)
        ops.gemm(a, wp, bias=bp, res=xa, out=xa, out_stats=st.view(-1))

    if C == 320:
        print(f"B{B} N{H*W} C{C}: emip_sra_qattn + proj {timed(two):7.1f} us", flush=True)
    print(f"B{B} N{H*W} C{C}: three launches {timed(three):7.1f} us   emip_sra_block {timed(one):7.1f} us  (dbg={os.environ.get('EMIP_SB_DBG', '0')})", flush=True)
