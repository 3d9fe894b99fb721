"""Calibration helper: run emip_gemm_tn on one shape a few times (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emip_amd import ops  # noqa: E402

M, N, K = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "30976x320x320").split("x"))
a = torch.randn(M, N, device="cuda").to(torch.bfloat16)
b = torch.randn(M, K, device="cuda").to(torch.bfloat16)
for _ in range(5):
    ops.gemm_tn(a, b)
torch.cuda.synchronize()
