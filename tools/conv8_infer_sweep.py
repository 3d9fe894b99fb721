#!/usr/bin/env python3
"""emip_conv8 tile sweep at the 16-pair inference shapes (GMFlow encoder at 32 images, conv_corr.2, upsampler, decoder)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emip_amd import _lib
from tools.gemm8_bench import conv
_lib.load()
cfgs = [1, 2, 3, 6, 7, 8, 9]
for B, H, W, Cin, Cout, k, s, p in [(32, 176, 176, 64, 64, 3, 1, 1), (32, 176, 176, 64, 96, 3, 2, 1), (32, 88, 88, 96, 96, 3, 1, 1),
                                    (32, 88, 88, 96, 128, 3, 2, 1), (32, 44, 44, 128, 128, 3, 1, 1), (32, 44, 44, 136, 256, 3, 1, 1),
                                    (16, 44, 44, 968, 128, 3, 1, 1), (16, 44, 44, 96, 96, 3, 1, 1), (16, 88, 88, 64, 64, 3, 1, 1),
                                    (16, 88, 88, 96, 96, 3, 1, 1), (16, 44, 44, 64, 64, 3, 1, 1)]:
    conv(B, H, W, Cin, Cout, k, s, p, cfgs)
