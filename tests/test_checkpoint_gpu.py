"""Checkpoint files through the device: a file written from a GPU-resident model (train.py:90 format, with and without DDP's
`module.` prefix) loaded by test.py:81-89's rule into a fresh model gives the same masks through the HIP path."""
import pytest
import torch

from emip_amd.filler import synthetic_pair

pytestmark = pytest.mark.gpu


def test_saved_file_reloads_to_identical_masks(tmp_path, model_args, short_sd):
    from emip_amd import checkpoint as C, nn_base
    from emip_amd.model.EMIP_short.model import CoUpdater
    nn_base.set_default_dtype(torch.float32)
    a = CoUpdater(model_args)
    a.load_state_dict(short_sd)
    a = a.to("cuda:0").eval()
    im1, im2 = synthetic_pair(1, seed=11)
    im1, im2 = im1.cuda(), im2.cuda()
    with torch.no_grad():
        ref = a(im1, im2)[0]
    for ddp in (False, True):
        p = str(tmp_path / ("ckpt_%d.pth" % ddp))
        C.save(a, p, ddp_prefix=ddp)
        b = CoUpdater(model_args)
        taken = C.load_for_inference(b, p, multi_gpu=ddp)
        assert len(taken) == len(short_sd)
        b = b.to("cuda:0").eval()
        with torch.no_grad():
            out = b(im1, im2)[0]
        assert (out - ref).abs().max().item() < 1e-3        # f32-atomic jitter of the statistics only
