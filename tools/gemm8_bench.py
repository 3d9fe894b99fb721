#!/usr/bin/env python3
"""Parity + timing sweep of emip_gemm8 / emip_conv8 against the 4-wave emip_gemm / emip_conv2d bodies, on the shapes the
EMIP forward launches at 16 pairs.  Reference for parity: torch matmul / conv in f32 on the device (test tooling only)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emip_amd import _lib, ops  # noqa: E402

NCFG = 11


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def dense(M, N, K, cfgs, hooks=False):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) if hooks else None
    ref = a.float() @ w.float().t() + bias
    act = ops.ACT_GELU if hooks else ops.ACT_NONE
    if hooks:
        ref = torch.nn.functional.gelu(ref) + res.float()
    out_old = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t_old = timeit(lambda: ops.gemm(a, w, bias=bias, res=res, act=act, out=out_old))
    scale = ref.abs().max().item()
    e_old = (out_old.float() - ref).abs().max().item() / scale
    line = "dense %7d x %5d x %5d %s| old %7.1f us (%5.0f TF, err %.1e) |" % (
        M, N, K, "+gelu+res " if hooks else "", t_old, 2.0 * M * N * K / t_old / 1e6, e_old)
    best = (1e9, 0)
    for c in cfgs:
        out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        st = torch.zeros(M, 2, device="cuda") if hooks else None
        ops.gemm8(a, w, bias=bias, res=res, act=act, out=out, out_stats=st, cfg=c)
        err = (out.float() - ref).abs().max().item() / scale
        bad = ""
        if hooks:
            es = (st[:, 0] - out.float().sum(1)).abs().max().item() / max(1.0, out.float().sum(1).abs().max().item())
            bad = " stats %.1e" % es if es > 1e-3 else ""
        t = timeit(lambda: ops.gemm8(a, w, bias=bias, res=res, act=act, out=out, out_stats=None, cfg=c))
        line += " c%d %6.1f%s%s" % (c, t, "" if err < 2e-2 else " ERR %.1e" % err, bad)
        best = min(best, (t, c))
    auto = _lib.load().emip_gemm8_auto_cfg(M, N, K)
    print(line + " | best c%d %.1f us (%5.0f TF) auto c%d" % (best[1], best[0], 2.0 * M * N * K / best[0] / 1e6, auto), flush=True)


def conv(B, H, W, Cin, Cout, k, stride, pad, cfgs):
    g = torch.Generator(device="cuda").manual_seed(B + H + Cin + Cout)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g).to(torch.bfloat16)
    wt = (torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5)
    bias = torch.randn(Cout, device="cuda", generator=g)
    wp = wt.permute(0, 2, 3, 1).reshape(Cout, -1).to(torch.bfloat16).contiguous()
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), wp.float().view(Cout, k, k, Cin).permute(0, 3, 1, 2),
                                     bias, stride=stride, padding=pad).permute(0, 2, 3, 1)
    scale = ref.abs().max().item()
    out_old = ops.conv2d(x, wp, k, k, stride, pad, bias=bias)
    t_old = timeit(lambda: ops.conv2d(x, wp, k, k, stride, pad, bias=bias, out=out_old), 10)
    fl = 2.0 * ref.numel() * Cin * k * k
    line = "conv B%d %dx%d %d->%d k%d s%d | old %8.1f us (%5.0f TF, err %.1e) |" % (
        B, H, W, Cin, Cout, k, stride, t_old, fl / t_old / 1e6, (out_old.float() - ref).abs().max().item() / scale)
    best = (1e9, 0)
    for c in cfgs:
        out = torch.zeros_like(out_old)
        ops.conv8(x, wp, k, k, stride, pad, bias=bias, out=out, cfg=c)
        err = (out.float() - ref).abs().max().item() / scale
        t = timeit(lambda: ops.conv8(x, wp, k, k, stride, pad, bias=bias, out=out, cfg=c), 10)
        line += " c%d %7.1f%s" % (c, t, "" if err < 2e-2 else " ERR %.1e" % err)
        best = min(best, (t, c))
    print(line + " | best c%d %.1f us (%5.0f TF)" % (best[1], best[0], fl / best[0] / 1e6), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--cfgs", default="")
    args = ap.parse_args()
    _lib.load()
    cfgs = [int(c) for c in args.cfgs.split(",")] if args.cfgs else list(range(1, NCFG + 1))
    # ragged edges first (parity of the masks), then the network's shapes
    dense(1000, 200, 128, cfgs, hooks=True)
    dense(333, 72, 64, cfgs)
    if args.quick:
        dense(15488, 1280, 320, cfgs)
        return
    for M, N, K in [(15488, 1280, 320), (15488, 320, 1280), (15488, 320, 320), (3872, 640, 320), (247808, 256, 64),
                    (247808, 64, 256), (247808, 64, 64), (61952, 512, 128), (61952, 128, 512), (61952, 128, 128),
                    (61952, 1024, 256), (61952, 128, 1024), (61952, 384, 128), (61952, 576, 256), (3872, 2048, 512),
                    (3872, 512, 2048), (8192, 8192, 8192)]:
        dense(M, N, K, cfgs)
    dense(15488, 320, 1280, cfgs, hooks=True)
    conv(2, 20, 20, 72, 40, 3, 1, 1, cfgs)
    conv(16, 44, 44, 1936, 968, 3, 1, 1, [1, 2, 3, 7])
    conv(16, 44, 44, 968, 128, 3, 1, 1, cfgs)
    conv(32, 176, 176, 64, 64, 3, 1, 1, cfgs)
    conv(32, 88, 88, 96, 96, 3, 1, 1, cfgs)
    conv(32, 44, 44, 128, 128, 3, 1, 1, cfgs)
    conv(32, 44, 44, 136, 256, 3, 1, 1, cfgs)


if __name__ == "__main__":
    main()
