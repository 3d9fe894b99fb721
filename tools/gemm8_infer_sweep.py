#!/usr/bin/env python3
"""tile sweep of emip_gemm8 at the 32-image inference shapes of the PVT stages"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emip_amd import _lib
from tools.gemm8_bench import dense, NCFG
_lib.load()
cfgs = list(range(1, NCFG + 1))
for M, N, K in [(15488, 320, 1280), (15488, 320, 320), (3872, 640, 320), (61952, 128, 512), (61952, 128, 128), (247808, 64, 256),
                (3872, 512, 2048), (3872, 2048, 512), (61952, 128, 256), (61952, 128, 1024), (61952, 1024, 256)]:
    dense(M, N, K, cfgs, hooks=(K >= 512 and N <= 512))
