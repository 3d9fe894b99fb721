#!/usr/bin/env python3
"""Golden bytes for the prediction post-processing of the reference's inference driver.

TEST INFRASTRUCTURE (build container only: /root/reference does not exist on the GPU box).  /root/reference/test.py:29-31
and :35-36 -- `F.upsample(output[0], size=shape, mode='bilinear', align_corners=False)`, `.sigmoid()`, the per-image min-max
normalisation in numpy and `Image.fromarray(output*255).convert('L').save(...)` -- are READ as text at run time, dedented and
`exec`ed: the reference's own statements on stand-ins for `output` (a tuple holding a mask-logit tensor), `shape`, the save
directory and the frame name.  The PNG the reference wrote is read back; its bytes are the golden.  Nothing of the
reference's text is stored: tests/golden/postprocess.npz holds the generator parameters of the inputs and the uint8 outputs.

The inputs are exactly representable (integers / 8 on a 11 x 11 block grid plus integers / 64 of pixel noise, numpy's legacy
RandomState: the same numbers on every machine), so the GPU box rebuilds them from the parameters.

One thread (`torch.set_num_threads(1)`): ATen's vectorised sigmoid handles the last numel % 32 elements of every parallel
chunk with the scalar `1 / (1 + std::exp(-x))` instead of its polynomial `exp_u20`, so WHICH pixels take which path depends
on the thread count of the machine the reference runs on; one chunk pins it (the device kernel reproduces exactly that)."""
import os
import tempfile
import textwrap

import numpy as np
import torch
import torch.nn.functional as F  # noqa: F401  (the reference's lines use it)

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "postprocess.npz")

# (seed, source-frame shape, kind)
CASES = [(11, (352, 352), "field"), (12, (360, 640), "field"), (13, (481, 321), "field"), (14, (97, 61), "field"),
         (15, (288, 512), "constant"), (16, (300, 500), "lowcontrast")]


def mask_logits(seed, kind):
    """f32 [1,1,352,352], every value a multiple of 1/64 (exact in f32, identical on every machine)"""
    rs = np.random.RandomState(seed)
    if kind == "constant":
        return np.full((1, 1, 352, 352), -2.5, np.float32)
    low = rs.randint(-40, 41, size=(11, 11)).astype(np.float32) / 8.0
    m = np.kron(low, np.ones((32, 32), np.float32)) + rs.randint(-8, 9, size=(352, 352)).astype(np.float32) / 64.0
    if kind == "lowcontrast":
        m = np.round(m * 4.0) / 64.0 - 4.0            # an all-background prediction: probabilities 0.01 .. 0.03
    return m.reshape(1, 1, 352, 352).astype(np.float32)


def ref_lines(first, last):
    with open(os.path.join(REF, "test.py")) as f:
        lines = f.readlines()
    return textwrap.dedent("".join(lines[first - 1:last]))


def main():
    torch.set_num_threads(1)
    resize_norm = compile(ref_lines(29, 31), "test.py:29-31", "exec")
    save_png = compile(ref_lines(35, 36), "test.py:35-36", "exec")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for i, (seed, shape, kind) in enumerate(CASES):
            env = {"F": F, "output": (torch.from_numpy(mask_logits(seed, kind)),), "shape": shape,
                   "map_save_path_final": tmp + "/", "name": "case%d" % i}
            exec(resize_norm, env)
            exec(save_png, env)
            from PIL import Image
            png = np.asarray(Image.open(os.path.join(tmp, "case%d.png" % i)))
            assert png.dtype == np.uint8 and png.shape == tuple(shape), (png.dtype, png.shape)
            out["u8_%d" % i] = png
            out["norm_%d" % i] = np.asarray(env["output"], dtype=np.float32)      # the normalised f32 map (train.py:125-127 feeds it to the metrics)
            print("case %d seed %d %s %s: bytes min %d max %d, mean %.2f" % (i, seed, shape, kind, png.min(), png.max(), png.mean()))
    out["cases"] = np.array([[s, h, w, ["field", "constant", "lowcontrast"].index(k)] for s, (h, w), k in CASES], dtype=np.int64)
    # the f32 maps are large: keep them for the two small cases only (the bytes pin the rest)
    for i, (_, shape, _) in enumerate(CASES):
        if shape[0] * shape[1] > 100 * 100:
            del out["norm_%d" % i]
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
