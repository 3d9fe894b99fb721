"""Prediction post-processing (test.py:28-36) on the device, BIT FOR BIT against bytes the reference's own statements produced
(tests/golden/postprocess.npz: oracle/make_golden_postprocess.py executes test.py:29-31,35-36 on stand-ins and reads the PNG
back), and against the oracle's statement-for-statement restatement (torch CPU + numpy + PIL) on further shapes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KINDS = ["field", "constant", "lowcontrast"]


def _golden():
    return np.load(os.path.join(HERE, "golden", "postprocess.npz"))


def test_bytes_equal_what_the_reference_statements_wrote():
    from emip_amd.postprocess import mask_to_float, mask_to_uint8
    from oracle.make_golden_postprocess import mask_logits
    g = _golden()
    for i, (seed, h, w, kind) in enumerate(g["cases"].tolist()):
        m = torch.from_numpy(mask_logits(seed, KINDS[kind])).cuda()
        out = mask_to_uint8(m, (h, w)).cpu().numpy()[0]
        ref = g["u8_%d" % i]
        bad = int((out != ref).sum())
        print(f"  case {i} seed {seed} {h}x{w} {KINDS[kind]}: {bad} of {ref.size} bytes differ")
        assert out.dtype == np.uint8 and out.shape == ref.shape and bad == 0, (i, bad)
        if "norm_%d" % i in g.files:          # the normalised f32 map (what train.py:125-127 hands to the metrics): the same bits
            f = mask_to_float(m, (h, w)).cpu().numpy()[0]
            assert np.array_equal(f.view(np.int32), g["norm_%d" % i].view(np.int32)), i


@pytest.mark.parametrize("shape", [(352, 352), (720, 1280), (300, 500), (97, 61)])
def test_batch_against_the_oracle_restatement(shape):
    """a batch of three smooth predictions (one of them low-contrast), every image normalised by itself; the oracle runs the
    reference's statements on one thread (ATen's sigmoid leaves the last numel % 32 elements of every parallel chunk to its
    scalar path: with one chunk those are the ones the device kernel treats the same way)"""
    from emip_amd.postprocess import mask_to_uint8, mask_to_uint8_host
    from oracle import emip_oracle as O
    g = torch.Generator().manual_seed(shape[0])
    low = torch.randn(3, 1, 11, 11, generator=g) * 3
    mask = torch.nn.functional.interpolate(low, size=(352, 352), mode="bilinear", align_corners=True).contiguous()
    mask[2] = mask[2] * 0.05 - 4.0                     # a low-contrast, all-background prediction
    out = mask_to_uint8(mask.cuda(), shape).cpu().numpy()
    assert out.shape == (3,) + tuple(shape) and out.dtype == np.uint8
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        for b in range(3):
            ref = O.postprocess_mask(mask[b:b + 1], shape)
            assert np.array_equal(out[b], ref), (b, int((out[b] != ref).sum()))
            assert out[b].min() == 0 and out[b].max() >= 254
    finally:
        torch.set_num_threads(nt)
    host, ev = mask_to_uint8_host(mask.cuda(), shape)
    ev.synchronize()
    assert np.array_equal(host.numpy(), out)
