"""error map of emip_sra_block against the f32 restatement (tests/test_sra_block_gpu.py): which rows / channels are off"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from emip_amd import ops
import test_sra_block_gpu as T

for (B, H, W, C, Lk) in [(1, 1, 1, 64, 1), (1, 8, 8, 64, 121), (1, 16, 8, 128, 121), (1, 16, 8, 320, 121), (4, 22, 22, 320, 121)]:
    heads = C // 64
    x, kv, wq, wp, bq, bp, stats = T._make(B, H, W, C, Lk, 7 * C + H + Lk)
    ref = T._ref(x, kv, wq, wp, bq, bp, heads, 0.125).view(-1, C)
    sw = ops.swap23(C, x.device)
    for rep in range(2):
        got = x.clone()
        st = torch.zeros(B * H * W, 2, device=x.device)
        ops.sra_block(got, stats, T.EPS, wq[sw].contiguous(), bq, wq.float().sum(1).contiguous(), kv, wp[sw][:, sw].contiguous(), bp,
                      heads, 0.125, out_stats=st.view(-1))
        torch.cuda.synchronize()
        e = (got.float().view(-1, C) - ref).abs()
        e = torch.nan_to_num(e, nan=1e30, posinf=1e30)
        bad = e > 0.05 * max(1.0, ref.abs().max().item())
        print(f"B{B} N{H*W} C{C} Lk{Lk} rep{rep}: bad {bad.sum().item()} of {bad.numel()}  max {e.max().item():.3g}")
        if bad.any():
            rows = bad.any(1).nonzero().flatten()
            cols = bad.any(0).nonzero().flatten()
            print("   bad rows (first 40):", rows[:40].tolist(), " count", len(rows))
            print("   bad cols (first 64):", cols[:64].tolist(), " count", len(cols))
            r0 = rows[0].item()
            print("   row", r0, "got", got.float().view(-1, C)[r0, :16].tolist())
            print("   row", r0, "ref", ref[r0, :16].tolist())
