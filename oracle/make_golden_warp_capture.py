"""Golden-vector generator for the bit-exact warp-index contract.  TEST INFRASTRUCTURE ONLY; runs ONLY in the build
container (imports /root/reference/loss/warp_utils.py, which needs nothing but torch).

The fixture holds the tensors the REFERENCE ITSELF hands to `scatter_add_` inside `get_corresponding_map`
(loss/warp_utils.py:62-76): `torch.Tensor.scatter_add_` is wrapped for the duration of the call and the `index`
(int64 [B, 4 H W], stored whole) and `src` (the bilinear corner weights, zeroed where a corner was clamped; stored as a
1-in-61 sample plus float64 row sums, the full array is incompressible) arguments are kept as the reference produced them
-- nothing is re-derived here.  Also stored: the occlusion mask of `get_occu_mask_backward`.
Inputs: the flows are regenerated from seeds by the tests (RandomState(11) N(0, 6) px at 352x352; a second, batch-2 flow
with RandomState(12) N(0, 40) px so that many corners leave the frame and hit the clamp / invalid branch).

usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_warp_capture.py [--out tests/golden]
"""
import argparse
import os
import sys

import numpy as np
import torch

REF = "/root/reference"


def flows():
    a = torch.from_numpy(np.random.RandomState(11).normal(0, 6.0, (1, 2, 352, 352)).astype(np.float32))
    b = torch.from_numpy(np.random.RandomState(12).normal(0, 40.0, (2, 2, 352, 352)).astype(np.float32))
    return {"a": a, "b": b}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    args = ap.parse_args()
    sys.path.insert(0, REF)
    from loss import warp_utils                      # the reference's own module
    captured = []
    orig = torch.Tensor.scatter_add_

    def spy(self, dim, index, src):
        captured.append((dim, index.clone(), src.clone()))
        return orig(self, dim, index, src)

    out = {}
    for name, fl in flows().items():
        captured.clear()
        torch.Tensor.scatter_add_ = spy
        try:
            occ = warp_utils.get_occu_mask_backward(fl)      # -> get_corresponding_map(base + flow) -> scatter_add_
        finally:
            torch.Tensor.scatter_add_ = orig
        assert len(captured) == 1 and captured[0][0] == 1
        _, idx, val = captured[0]
        assert idx.dtype == torch.int64 and idx.shape == (fl.shape[0], 4 * 352 * 352)
        keep = slice(None) if name == "a" else slice(1, 2)               # flow b: the SECOND image of the batch only (size)
        out[name + "_indices"] = idx.numpy().astype(np.int32)[keep]      # < 2^17: int32 holds them exactly
        w = val.numpy().astype(np.float32)
        out[name + "_weights_sample"] = w[:, ::61].copy()                # every 61st corner weight + exact float64 sums
        out[name + "_weights_sum"] = w.astype(np.float64).sum(1)
        out[name + "_occ"] = occ.numpy().astype(np.uint8)
    np.savez_compressed(os.path.join(args.out, "warp_indices_captured.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
