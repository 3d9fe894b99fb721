"""Input preparation (dataset.py:257-260): device Resize + ToTensor + Normalize against Pillow / the oracle, bit-exact."""
import numpy as np
import pytest
import torch


@pytest.mark.parametrize("hw", [(720, 1280), (480, 854), (352, 352), (200, 300), (97, 1000), (353, 351)])
def test_pillow_coefficients_restatement_is_bit_exact_on_cpu(hw):
    """the host-side coefficient tables + the two integer passes (numpy) == PIL.Image.resize(BILINEAR)"""
    from PIL import Image
    from emip_amd.preprocess import pillow_bilinear_coeffs
    H, W = hw
    img = np.random.RandomState(H + W).randint(0, 256, (H, W, 3)).astype(np.uint8)
    kh, bh = pillow_bilinear_coeffs(W, 352)
    kv, bv = pillow_bilinear_coeffs(H, 352)
    tmp = np.zeros((H, 352, 3), np.uint8)
    for xx in range(352):
        acc = np.full((H, 3), 1 << 21, np.int64)
        for x in range(bh[xx, 1]):
            acc += img[:, bh[xx, 0] + x].astype(np.int64) * int(kh[xx, x])
        tmp[:, xx] = np.clip(acc >> 22, 0, 255)
    out = np.zeros((352, 352, 3), np.uint8)
    for yy in range(352):
        acc = np.full((352, 3), 1 << 21, np.int64)
        for y in range(bv[yy, 1]):
            acc += tmp[bv[yy, 0] + y].astype(np.int64) * int(kv[yy, y])
        out[yy] = np.clip(acc >> 22, 0, 255)
    assert np.array_equal(out, np.asarray(Image.fromarray(img, "RGB").resize((352, 352), Image.BILINEAR)))


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(720, 1280), (480, 854), (352, 352), (200, 300), (97, 1000)])
def test_device_preprocess_is_bit_exact(hw):
    from emip_amd.preprocess import rgb_to_model_input
    from oracle import emip_oracle as O
    H, W = hw
    rs = np.random.RandomState(H * 7 + W)
    frames = rs.randint(0, 256, (2, H, W, 3)).astype(np.uint8)
    frames[1, :, : W // 2] = 255                     # saturated half: exercises the clip
    out, u8 = rgb_to_model_input(torch.from_numpy(frames).cuda(), 352, return_resized=True)
    for b in range(2):
        ref, ref_u8 = O.preprocess_rgb(frames[b], 352)
        assert np.array_equal(u8[b].cpu().numpy(), ref_u8)
        assert torch.equal(out[b].cpu(), ref)           # f32 bit-exact: IEEE division / subtraction on the device
