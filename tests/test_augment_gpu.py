"""Training-time augmentation (dataset/data_augment.py:12-45, dataset/dataset.py:94-103) on the device against Pillow (the
reference's own arithmetic, through the restatement in oracle/emip_oracle.py): every stage bit-exact, and the whole
`__getitem__` chain under a shared seed."""
import random

import numpy as np
import pytest
import torch


def _frame(rs, H, W, flat=False):
    img = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
    if flat:
        img[:, : W // 2] = rs.randint(0, 256, 3)          # large constant area: exact-integer blends
        img[: H // 3] = 255
    return img


def test_rotate_matrix_is_pillows():
    """host logic: the matrix handed to the kernel is what Image.rotate computes (checked through the C transform)"""
    from PIL import Image
    from emip_amd.data_augment import rotate_matrix
    rs = np.random.RandomState(0)
    img = Image.fromarray(_frame(rs, 37, 53), "RGB")
    for angle in (-15, -7, 3, 14, 33):
        m = rotate_matrix(angle, 53, 37)
        a = np.asarray(img.rotate(angle, Image.BICUBIC))
        b = np.asarray(img.transform((53, 37), Image.AFFINE, m, Image.BICUBIC))
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(720, 1280), (97, 203), (3, 3), (64, 5)])
def test_color_enhance_is_bit_exact(H, W):
    from emip_amd.data_augment import color_enhance
    from oracle import emip_oracle as O
    rs = np.random.RandomState(H + W)
    grid = [(0.5, 0.5, 0.0, 0.0), (1.5, 1.5, 2.0, 3.0), (1.0, 1.0, 1.0, 1.0), (0.7, 1.3, 0.3, 2.9), (1.2, 0.9, 1.7, 0.1)]
    grid += [(rs.randint(5, 16) / 10.0, rs.randint(5, 16) / 10.0, rs.randint(0, 21) / 10.0, rs.randint(0, 31) / 10.0)
             for _ in range(5)]
    for i, f in enumerate(grid):
        img = _frame(rs, H, W, flat=i % 2 == 1)
        got = color_enhance(torch.from_numpy(img).cuda(), *f).cpu().numpy()
        assert np.array_equal(got, O.aug_color_enhance(img, f)), f


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(720, 1280), (97, 203), (50, 50), (4, 9)])
def test_rotate_bicubic_is_bit_exact(H, W):
    from PIL import Image
    from emip_amd.data_augment import rotate
    rs = np.random.RandomState(H * 3 + W)
    rgb = _frame(rs, H, W)
    yy, xx = np.mgrid[0:H, 0:W]
    mask = (((yy - H / 2) ** 2 + (xx - W / 3) ** 2 < (H / 3) ** 2) * 255).astype(np.uint8)
    angles = range(-15, 15) if H < 200 else (-15, -4, 0, 9, 14)
    if H == W:
        angles = list(angles) + [90, 180, 270]
    for angle in angles:
        got = rotate(torch.from_numpy(rgb).cuda(), angle).cpu().numpy()
        assert np.array_equal(got, np.asarray(Image.fromarray(rgb, "RGB").rotate(angle, Image.BICUBIC))), angle
        got = rotate(torch.from_numpy(mask).cuda(), angle).cpu().numpy()
        assert np.array_equal(got, np.asarray(Image.fromarray(mask, "L").rotate(angle, Image.BICUBIC))), angle


@pytest.mark.gpu
def test_random_peper_consumes_the_same_draws():
    from emip_amd.data_augment import randomPeper
    from oracle import emip_oracle as O
    rs = np.random.RandomState(5)
    for H, W in ((480, 854), (40, 30), (20, 20)):                 # 20x20: noiseNum = 0
        gt = ((rs.rand(H, W) > 0.6) * 255).astype(np.uint8)
        random.seed(11)
        want = O.aug_random_peper(gt)
        after = random.random()
        random.seed(11)
        got = randomPeper(torch.from_numpy(gt).cuda()).cpu().numpy()
        assert np.array_equal(got, want) and random.random() == after


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(720, 1280), (352, 352), (97, 1000)])
def test_gray_transform_is_bit_exact(hw):
    from emip_amd.preprocess import gray_to_model_input
    from oracle import emip_oracle as O
    H, W = hw
    rs = np.random.RandomState(H + 3 * W)
    gts = np.stack([((rs.rand(H, W) > 0.5) * 255).astype(np.uint8), rs.randint(0, 256, (H, W)).astype(np.uint8)])
    out, u8 = gray_to_model_input(torch.from_numpy(gts).cuda(), 352, return_resized=True)
    for b in range(2):
        ref, ref_u8 = O.preprocess_gray(gts[b], 352)
        assert np.array_equal(u8[b].cpu().numpy(), ref_u8)
        assert torch.equal(out[b].cpu(), ref)


@pytest.mark.gpu
def test_train_sample_matches_reference_getitem_under_a_shared_seed():
    """dataset.py:94-103 end to end: same seeds -> same rotation decision / angle, colour factors, pepper positions, and
    bit-identical tensors; the seeds are chosen so that both the rotated and the unrotated branch are taken"""
    from emip_amd.data_augment import train_sample
    from oracle import emip_oracle as O
    rs = np.random.RandomState(9)
    H, W = 360, 640
    im1, im2 = _frame(rs, H, W), _frame(rs, H, W, flat=True)
    yy, xx = np.mgrid[0:H, 0:W]
    gt = (((yy - 200) ** 2 + (xx - 300) ** 2 < 90 ** 2) * 255).astype(np.uint8)
    rotated = set()
    for seed in range(8):
        random.seed(seed)
        np.random.seed(seed)
        rotated.add(random.random() > 0.8)
        random.seed(seed)
        want = O.train_sample(im1, im2, gt)
        state = (random.random(), np.random.randint(1 << 30))
        random.seed(seed)
        np.random.seed(seed)
        got = train_sample(torch.from_numpy(im1).cuda(), torch.from_numpy(im2).cuda(), torch.from_numpy(gt).cuda())
        assert state == (random.random(), np.random.randint(1 << 30))        # the generators advanced identically
        for g, w in zip(got, want):
            assert torch.equal(g.cpu(), w)
    assert rotated == {True, False}
