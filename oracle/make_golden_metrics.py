#!/usr/bin/env python3
"""Golden vectors for the validation metrics: imports the reference's own eval/metrics.py (needs numpy, scipy, sklearn --
all present in the build container) and records MAE / S-measure for seeded frames, including the degenerate branches
(empty gt, full gt, constant prediction).  Writes tests/golden/metrics_micro.npz (inputs + expected values).
Test infrastructure only; run in the build container: PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_metrics.py"""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    spec = importlib.util.spec_from_file_location("ref_metrics", "/root/reference/eval/metrics.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rs = np.random.RandomState(7)
    H, W = 90, 120
    yy, xx = np.mgrid[0:H, 0:W]
    blob = ((yy - 50) ** 2 / 400.0 + (xx - 40) ** 2 / 900.0) < 1.0
    cases = []
    p = np.clip(blob * 0.8 + rs.rand(H, W) * 0.3, 0, 1).astype(np.float32)
    cases.append((p, (blob * 255).astype(np.float32)))                                  # a reasonable prediction
    cases.append((rs.rand(H, W).astype(np.float32), ((rs.rand(H, W) > 0.8) * 255).astype(np.float32)))   # noise
    cases.append((p, np.zeros((H, W), np.float32)))                                      # empty gt
    cases.append((p, np.full((H, W), 255, np.float32)))                                  # full gt
    cases.append((np.full((H, W), 0.3, np.float32), (blob * 255).astype(np.float32)))    # constant prediction
    corner = np.zeros((H, W), np.float32)
    corner[:3, :5] = 255
    cases.append((p, corner))                                                             # centroid near a corner
    out = {}
    for i, (pred, gt) in enumerate(cases):
        a, b = m.MAE(), m.Smeasure()
        a.step(pred=pred, gt=gt)
        b.step(pred=pred, gt=gt)
        out["pred%d" % i], out["gt%d" % i] = pred, gt.astype(np.uint8)
        out["mae%d" % i] = np.float64(a.get_results()["mae"])
        out["sm%d" % i] = np.float64(b.get_results()["sm"])
        print(i, out["mae%d" % i], out["sm%d" % i])
    out["n"] = len(cases)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "metrics_micro.npz"), **out)


if __name__ == "__main__":
    main()
