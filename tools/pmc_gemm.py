import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emip_amd import ops
dev, dt = "cuda:0", torch.bfloat16
for M, N, K in [(15488, 1280, 320), (15488, 320, 1280), (15488, 320, 320), (247808, 256, 64), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev).to(dt); o = torch.empty(M, N, device=dev, dtype=dt)
    for _ in range(3):
        ops.gemm(a, w, bias=b, res=r, out=o)
    torch.cuda.synchronize()
B, C, Lk = 32, 320, 121
q = torch.randn(B, 484, C, device=dev).to(dt); kv = torch.randn(B, Lk, 2 * C, device=dev).to(dt); o = torch.empty_like(q)
for _ in range(3):
    ops.attention(q, kv, kv[:, :, C:], o, batch=B, heads=5, nwin=1, Lq=484, Lk=Lk, D=64, DV=64, q_bs=484 * C, k_bs=Lk * 2 * C,
                  v_bs=Lk * 2 * C, o_bs=484 * C, ldq=C, ldk=2 * C, ldv=2 * C, ldo=C, q_hs=64, k_hs=64, v_hs=64, o_hs=64, scale=0.125)
torch.cuda.synchronize()
