"""Prediction post-processing (test.py:28-31) on the device against the oracle's statement-for-statement restatement
(torch CPU + numpy + PIL, as the reference runs it)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(352, 352), (720, 1280), (300, 500), (97, 61)])
def test_mask_to_uint8_matches_reference_postprocessing(shape):
    from emip_amd.postprocess import mask_to_uint8, mask_to_uint8_host
    from oracle import emip_oracle as O
    g = torch.Generator().manual_seed(shape[0])
    low = torch.randn(3, 1, 11, 11, generator=g) * 3
    mask = torch.nn.functional.interpolate(low, size=(352, 352), mode="bilinear", align_corners=True).contiguous()
    mask[2] = mask[2] * 0.05 - 4.0                     # a low-contrast, all-background prediction
    out = mask_to_uint8(mask.cuda(), shape).cpu().numpy()
    assert out.shape == (3,) + tuple(shape) and out.dtype == np.uint8
    for b in range(3):
        ref = O.postprocess_mask(mask[b:b + 1], shape)
        d = np.abs(out[b].astype(np.int16) - ref.astype(np.int16))
        # truncation after x255: a last-bit difference in sigmoid / the normalisation may move a value across an integer
        assert d.max() <= 1, d.max()
        assert (d > 0).mean() < 5e-3, (d > 0).mean()
        assert out[b].min() == 0 and out[b].max() >= 254
    host, ev = mask_to_uint8_host(mask.cuda(), shape)
    ev.synchronize()
    assert np.array_equal(host.numpy(), out)
