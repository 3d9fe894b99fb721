"""Validation metrics with the pixel reductions on MI355X.

Same classes / `step` / `get_results` protocol as /root/reference/eval/metrics.py (`MAE` :90-106, `Smeasure` :109-217),
as `train.py:129-146` uses them, but `pred` and `gt` are device tensors: a frame costs one C-ABI call (three small
reduction launches) and a 320-byte copy instead of a full-resolution f32 map to the host plus numpy.  The scalar
finalisation below follows the reference's formulas in float64.  `WeightedFmeasure` (:333-397) runs the exact Euclidean
feature transform (scipy's `distance_transform_edt(..., return_indices=True)`, including its choice among equidistant
pixels), the 7x7 Gaussian and the weighted sums on the device (`emip_eval_wfm`)."""
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


_wfm_consts = {}


def _wfm_constants(device):
    """49 taps of matlab_style_gauss2D((7, 7), sigma=5) (metrics.py:385-393) + log(0.5)/5 (:369), float64 on the device"""
    key = str(device)
    if key not in _wfm_consts:
        y, x = np.ogrid[-3:4, -3:4]
        h = np.exp(-(x * x + y * y) / (2.0 * 5 * 5))
        h[h < np.finfo(h.dtype).eps * h.max()] = 0
        sumh = h.sum()
        if sumh != 0:
            h /= sumh
        kc = np.concatenate([h.reshape(-1), [np.log(0.5) / 5]]).astype(np.float64)
        _wfm_consts[key] = torch.from_numpy(kc).to(device)
    return _wfm_consts[key]


def wfm_sums(pred, gt):
    """-> numpy float64 [4]: n_gt, sum Ew[gt], sum Ew[~gt] (metrics.py:347-372 on the device)"""
    assert pred.is_cuda and gt.is_cuda and pred.dim() == 2 and pred.shape == gt.shape
    p = pred.contiguous().float()
    g = gt.contiguous().float()
    H, W = p.shape
    out = torch.empty(4, dtype=torch.float64, device=p.device)
    ws = torch.empty(256 + 16 * H * W, dtype=torch.uint8, device=p.device)
    _lib.call("emip_eval_wfm", p.data_ptr(), g.data_ptr(), _wfm_constants(p.device).data_ptr(), out.data_ptr(),
              ws.data_ptr(), H, W, torch.cuda.current_stream().cuda_stream)
    return out.cpu().numpy()


def edt_indices(gt):
    """scipy.ndimage.distance_transform_edt(gt == 0, return_indices=True)[1] for gt [H,W] in 0..255 -> int32 [2,H,W]"""
    g = gt.contiguous().float()
    H, W = g.shape
    idx = torch.empty((2, H, W), dtype=torch.int32, device=g.device)
    ws = torch.empty(256 + 16 * H * W, dtype=torch.uint8, device=g.device)
    _lib.call("emip_eval_edt_indices", g.data_ptr(), idx.data_ptr(), ws.data_ptr(), H, W,
              torch.cuda.current_stream().cuda_stream)
    return idx


def _wfm(s, beta=1.0):
    """metrics.py:341-383 from the three sums"""
    n_gt, ew_fg, ew_bg = s[0], s[1], s[2]
    if n_gt == 0:
        return 0
    tpw = n_gt - ew_fg
    fpw = ew_bg
    r = 1 - ew_fg / n_gt
    p = tpw / (tpw + fpw + _EPS)
    return (1 + beta) * r * p / (r + beta * p + _EPS)


class WeightedFmeasure(object):
    def __init__(self, beta=1):
        self.beta = beta
        self.weighted_fms = []

    def step(self, pred, gt):
        self.weighted_fms.append(_wfm(wfm_sums(pred, gt), self.beta))

    def get_results(self):
        return dict(wfm=np.mean(np.array(self.weighted_fms, dtype=_TYPE)))


class FrameMetrics(object):
    """MAE, S-measure and weighted F-measure of the same frames (what train.py:129-131 steps separately): one reduction
    pass for the first two, the distance-transform pipeline for the third, two small D2H copies per frame"""

    def __init__(self, alpha=0.5, beta=1):
        self.mae, self.sm, self.wfm = MAE(), Smeasure(alpha), WeightedFmeasure(beta)

    def step(self, pred, gt):
        s, hw = frame_sums(pred, gt)
        self.mae.maes.append(_mae(s, hw))
        self.sm.sms.append(_sm(s, hw, self.sm.alpha))
        self.wfm.step(pred, gt)

    def get_results(self):
        return dict(mae=self.mae.get_results()["mae"], sm=self.sm.get_results()["sm"],
                    wFm=self.wfm.get_results()["wfm"])
