#!/usr/bin/env python3
"""Golden vectors for the validation metrics: imports the reference's own eval/metrics.py (needs numpy, scipy, sklearn --
all present in the build container) and records MAE / S-measure / weighted F-measure for seeded frames, including the
degenerate branches (empty gt, full gt, constant prediction) and masks full of equidistant-neighbour ties (rectangles,
sparse points) that pin the distance transform's nearest-index choice.  Writes tests/golden/metrics_micro.npz (inputs + expected values).
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
    rect = np.zeros((H, W), np.float32)
    rect[20:50, 30:80] = 255
    rect[60:75, 10:25] = 255
    cases.append((rs.rand(H, W).astype(np.float32), rect))                                # rectangles: many exact ties
    pts = ((rs.rand(H, W) > 0.995) * 255).astype(np.float32)
    cases.append((rs.rand(H, W).astype(np.float32), pts))                                 # sparse points
    big = np.zeros((270, 481), np.float32)
    yy2, xx2 = np.mgrid[0:270, 0:481]
    big[((yy2 - 130) ** 2 + (xx2 - 200) ** 2 < 60 ** 2) | ((yy2 - 60) ** 2 + (xx2 - 400) ** 2 < 30 ** 2)] = 255
    pb = np.clip(big / 255 * 0.7 + rs.rand(270, 481) * 0.4, 0, 1).astype(np.float32)
    cases.append((pb, big))                                                               # odd-sized frame, two discs
    out = {}
    for i, (pred, gt) in enumerate(cases):
        a, b, c = m.MAE(), m.Smeasure(), m.WeightedFmeasure()
        a.step(pred=pred, gt=gt)
        b.step(pred=pred, gt=gt)
        c.step(pred=pred, gt=gt)
        out["wfm%d" % i] = np.float64(c.get_results()["wfm"])
        out["pred%d" % i], out["gt%d" % i] = pred, gt.astype(np.uint8)
        out["mae%d" % i] = np.float64(a.get_results()["mae"])
        out["sm%d" % i] = np.float64(b.get_results()["sm"])
        print(i, out["mae%d" % i], out["sm%d" % i], out["wfm%d" % i])
    out["n"] = len(cases)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "metrics_micro.npz"), **out)


if __name__ == "__main__":
    main()
