"""Flow warp / occlusion helpers on MI355X: /root/reference/loss/warp_utils.py."""
from .. import ops


def flow_warp(x, flow12, pad='border', mode='bilinear'):
    assert pad == 'border' and mode == 'bilinear'
    return ops.flow_warp(x.contiguous(), flow12.contiguous())


def get_occu_mask_backward(flow21, th=0.2):
    return ops.occ_mask_backward(flow21.contiguous(), th)


def get_corresponding_indices(flow):
    """int64 corner indices [B, 4*H*W] and weights, reference corner order (warp_utils.py:43-70)."""
    return ops.occ_corners(flow.contiguous())
