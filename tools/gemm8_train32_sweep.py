#!/usr/bin/env python3
"""tile-configuration sweep of emip_gemm8 at the training step's CURRENT shapes: PVT stages 3-4 run on the 32 images whose deep
features the forward reads (M = 15 488 / 3 872), stages 1-2 and GMFlow on 64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emip_amd import _lib
from tools.gemm8_bench import dense, NCFG

_lib.load()
cfgs = list(range(1, NCFG + 1))
for M, N, K in [(15488, 1280, 320), (15488, 320, 1280), (15488, 320, 320), (3872, 640, 320), (3872, 2048, 512),
                (3872, 512, 2048), (3872, 512, 512), (123904, 128, 128), (123904, 1024, 256), (123904, 128, 1024)]:
    dense(M, N, K, cfgs)
