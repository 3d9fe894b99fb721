"""Validation metrics with the pixel reductions on MI355X.

Same classes / `step` / `get_results` protocol as /root/reference/eval/metrics.py (`MAE` :90-106, `Smeasure` :109-217),
as `train.py:129-146` uses them, but `pred` and `gt` are device tensors: a frame costs one C-ABI call (three small
reduction launches) and a 320-byte copy instead of a full-resolution f32 map to the host plus numpy.  The scalar
finalisation below follows the reference's formulas in float64.  `WeightedFmeasure` (:333-397) needs an exact Euclidean
distance transform (scipy `distance_transform_edt`) and is not moved."""
import numpy as np
import torch

from . import _lib

_EPS = np.spacing(1)
_TYPE = np.float64


def frame_sums(pred, gt):
    """pred f32 [H,W] (the map the reference passes as `pred=res`), gt [H,W] in 0..255 (any real dtype), both on the
    device -> numpy float64 [40] of pixel sums (layout: csrc/eval_metrics.hip)"""
    assert pred.is_cuda and gt.is_cuda and pred.dim() == 2 and pred.shape == gt.shape
    p = pred.contiguous().float()
    g = gt.contiguous().float()
    acc = torch.empty(40, dtype=torch.float64, device=p.device)
    ws = torch.empty(2, dtype=torch.int32, device=p.device)
    H, W = p.shape
    _lib.call("emip_eval_frame", p.data_ptr(), g.data_ptr(), acc.data_ptr(), ws.data_ptr(), H, W,
              torch.cuda.current_stream().cuda_stream)
    return acc.cpu().numpy(), (H, W)


def logits_to_pred(mask_logits, shape):
    """train.py:125-127 on the device: upsample to the gt size, sigmoid, min-max -> f32 [Ho,Wo] (first image)"""
    x = mask_logits.contiguous()
    B, _, H, W = x.shape
    out = torch.empty((B, int(shape[0]), int(shape[1])), dtype=torch.float32, device=x.device)
    ws = torch.empty(2 * B, dtype=torch.int32, device=x.device)
    _lib.call("emip_postprocess_mask_f32", x.data_ptr(), out.data_ptr(), ws.data_ptr(), B, H, W, int(shape[0]),
              int(shape[1]), torch.cuda.current_stream().cuda_stream)
    return out


def _mae(s, hw):
    return s[2] / (hw[0] * hw[1])                                     # metrics.py:100-102


def _s_object(sum_x, sum_x2, n):
    """metrics.py:138-142 from first / second moments over the selected pixels (std with ddof = 1)"""
    x = sum_x / n
    var = (sum_x2 - n * x * x) / (n - 1) if n > 1 else float("nan")
    sigma = np.sqrt(max(var, 0.0)) if var == var else float("nan")
    return 2 * x / (x * x + 1 + sigma + _EPS)


def _ssim(N, sp, sg, sp2, sg2, spg):
    """metrics.py:193-213 from the quadrant's moments"""
    if N == 0:
        return float("nan")
    x, y = sp / N, sg / N
    if N > 1:
        sigma_x = (sp2 - N * x * x) / (N - 1)
        sigma_y = (sg2 - N * y * y) / (N - 1)
        sigma_xy = (spg - N * x * y) / (N - 1)
    else:
        sigma_x = sigma_y = sigma_xy = float("nan")
    alpha = 4 * x * y * sigma_xy
    beta = (x * x + y * y) * (sigma_x + sigma_y)
    if alpha != 0:
        return alpha / (beta + _EPS)
    if alpha == 0 and beta == 0:
        return 1.0
    return 0.0


def _sm(s, hw, alpha=0.5):
    """metrics.py:120-155"""
    H, W = hw
    n = H * W
    n_gt = s[0]
    y = n_gt / n
    mean_pred = s[1] / n
    if y == 0:
        return 1 - mean_pred
    if y == 1:
        return mean_pred
    obj = y * _s_object(s[3], s[4], n_gt) + (1 - y) * _s_object(s[5], s[6], n - n_gt)
    cx = int(np.round(s[7] / n_gt)) + 1
    cy = int(np.round(s[8] / n_gt)) + 1
    w1 = cx * cy / n
    w2 = cy * (W - cx) / n
    w3 = (H - cy) * cx / n
    w4 = 1 - w1 - w2 - w3
    q = [_ssim(*s[10 + 6 * k:16 + 6 * k]) for k in range(4)]
    region = w1 * q[0] + w2 * q[1] + w3 * q[2] + w4 * q[3]
    return max(0, alpha * obj + (1 - alpha) * region)


class MAE(object):
    def __init__(self):
        self.maes = []

    def step(self, pred, gt):
        s, hw = frame_sums(pred, gt)
        self.maes.append(_mae(s, hw))

    def get_results(self):
        return dict(mae=np.mean(np.array(self.maes, _TYPE)))


class Smeasure(object):
    def __init__(self, alpha=0.5):
        self.sms = []
        self.alpha = alpha

    def step(self, pred, gt):
        s, hw = frame_sums(pred, gt)
        self.sms.append(_sm(s, hw, self.alpha))

    def get_results(self):
        return dict(sm=np.mean(np.array(self.sms, dtype=_TYPE)))


class FrameMetrics(object):
    """MAE and S-measure of the same frames from ONE device pass per frame (what train.py:129-131 steps separately)"""

    def __init__(self, alpha=0.5):
        self.mae, self.sm = MAE(), Smeasure(alpha)

    def step(self, pred, gt):
        s, hw = frame_sums(pred, gt)
        self.mae.maes.append(_mae(s, hw))
        self.sm.sms.append(_sm(s, hw, self.sm.alpha))

    def get_results(self):
        return dict(mae=self.mae.get_results()["mae"], sm=self.sm.get_results()["sm"])
