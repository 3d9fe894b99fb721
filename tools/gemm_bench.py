#!/usr/bin/env python3
"""Micro-benchmark of the GEMM / conv / dwconv / attention kernels on the shapes of the EMIP forward
(batch 16 pairs = 32 images).  Run on the GPU box: python tools/gemm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emip_amd import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    from emip_amd import _lib
    nbuf = int(os.environ.get("NBUF", "1"))
    _lib.call("emip_debug_set", 0, nbuf)
    print("NBUF", nbuf)
    dev = "cuda:0"
    dt = torch.bfloat16
    print("== gemm (M, N, K)   bias+residual epilogue")
    for M, N, K in [(15488, 1280, 320), (15488, 320, 1280), (15488, 320, 320), (15488, 640, 320),
                    (61952, 512, 128), (61952, 128, 512), (61952, 128, 128), (247808, 256, 64), (247808, 64, 256),
                    (247808, 64, 64), (3872, 2048, 512), (3872, 512, 2048), (61952, 384, 128), (61952, 1024, 256),
                    (61952, 128, 1024), (8192, 8192, 8192)]:
        a = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev).to(dt)
        o = torch.empty(M, N, device=dev, dtype=dt)
        us = timeit(lambda: ops.gemm(a, w, bias=b, res=r, out=o))
        fl = 2.0 * M * N * K
        byt = 2.0 * (M * K + N * K + 2 * M * N)
        print("gemm %7d %5d %5d : %8.1f us  %7.1f TF/s  %6.2f TB/s" % (M, N, K, us, fl / us / 1e6, byt / us / 1e6))
    print("== conv")
    for B, H, W, Cin, Cout, k, s, p in [(16, 44, 44, 1936, 968, 3, 1, 1), (16, 44, 44, 968, 128, 3, 1, 1),
                                        (32, 176, 176, 64, 64, 3, 1, 1), (32, 88, 88, 96, 96, 3, 1, 1),
                                        (32, 352, 352, 8, 64, 7, 2, 3), (32, 352, 352, 8, 64, 7, 4, 3),
                                        (32, 88, 88, 64, 64, 8, 8, 0), (32, 44, 44, 128, 128, 3, 1, 1)]:
        x = torch.randn(B, H, W, Cin, device=dev).to(dt)
        w = (torch.randn(Cout, k * k * Cin, device=dev) / (k * k * Cin) ** 0.5).to(dt)
        us = timeit(lambda: ops.conv2d(x, w, k, k, s, p), iters=10)
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        fl = 2.0 * B * Ho * Wo * Cout * k * k * Cin
        print("conv B%d %dx%d Cin%d Cout%d k%d s%d : %8.1f us  %7.1f TF/s" % (B, H, W, Cin, Cout, k, s, us,
                                                                              fl / us / 1e6))
    print("== dwconv3x3 + gelu")
    for B, H, W, C in [(32, 88, 88, 256), (32, 44, 44, 512), (32, 22, 22, 1280), (32, 11, 11, 2048)]:
        x = torch.randn(B, H, W, C, device=dev).to(dt)
        wt = torch.randn(9, C, device=dev)
        bb = torch.randn(C, device=dev)
        o = torch.empty_like(x)
        us = timeit(lambda: ops.dwconv3x3(x, wt, bb, act=ops.ACT_GELU, out=o))
        print("dwconv B%d %dx%d C%d : %8.1f us  %6.2f TB/s" % (B, H, W, C, us, 4.0 * x.numel() / us / 1e6))
    print("== layernorm")
    for M, C in [(247808, 64), (61952, 128), (15488, 320), (3872, 512)]:
        x = torch.randn(M, C, device=dev).to(dt)
        g = torch.ones(C, device=dev)
        o = torch.empty_like(x)
        us = timeit(lambda: ops.layernorm(x, g, g, 1e-6, out=o))
        print("ln %7d %4d : %8.1f us  %6.2f TB/s" % (M, C, us, 4.0 * x.numel() / us / 1e6))
    print("== attention (SRA)")
    for N, heads in [(7744, 1), (1936, 2), (484, 5), (121, 8)]:
        B, C, Lk = 32, heads * 64, 121
        q = torch.randn(B, N, C, device=dev).to(dt)
        kv = torch.randn(B, Lk, 2 * C, device=dev).to(dt)
        o = torch.empty_like(q)
        us = timeit(lambda: ops.attention(q, kv, kv[:, :, C:], o, batch=B, heads=heads, nwin=1, Lq=N, Lk=Lk, D=64,
                                          DV=64, q_bs=N * C, k_bs=Lk * 2 * C, v_bs=Lk * 2 * C, o_bs=N * C, ldq=C,
                                          ldk=2 * C, ldv=2 * C, ldo=C, q_hs=64, k_hs=64, v_hs=64, o_hs=64,
                                          scale=0.125))
        fl = 4.0 * B * heads * N * Lk * 64
        print("sra N%5d heads%d : %8.1f us  %7.1f TF/s" % (N, heads, us, fl / us / 1e6))
    B, L, C = 32, 484, 128
    q = torch.randn(B * 4, L, C, device=dev).to(dt)
    o = torch.empty_like(q)
    us = timeit(lambda: ops.attention(q, q, q, o, batch=B * 4, heads=1, nwin=1, Lq=L, Lk=L, D=128, DV=128,
                                      q_bs=L * C, k_bs=L * C, v_bs=L * C, o_bs=L * C, ldq=C, ldk=C, ldv=C, ldo=C,
                                      scale=C ** -0.5))
    print("swin-like 128 x 484 x 128 : %8.1f us  %7.1f TF/s" % (us, 4.0 * B * 4 * L * L * C / us / 1e6))


if __name__ == "__main__":
    main()
