#!/usr/bin/env python3
"""tile-configuration sweep of emip_gemm8 at the TRAINING step's shapes (64 images): is the dispatch heuristic, tuned on the
32-image inference shapes, still picking the fastest tile?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emip_amd import _lib
from tools.gemm8_bench import dense, NCFG

_lib.load()
cfgs = list(range(1, NCFG + 1))
for M, N, K in [(30976, 1280, 320), (30976, 320, 1280), (30976, 320, 320), (7744, 640, 320), (123904, 512, 128),
                (123904, 128, 512), (123904, 128, 128), (123904, 256, 128), (495616, 256, 64), (495616, 64, 256),
                (495616, 64, 64), (7744, 2048, 512), (7744, 512, 2048), (7744, 512, 512), (123904, 1024, 256),
                (123904, 128, 1024), (123904, 128, 256)]:
    dense(M, N, K, cfgs)
