"""Constant index / position tables of the GMFlow stream (built once per device).

The reference materialises torch.roll + split_feature + merge_splits copies for every
attention call (/root/reference/model/EMIP_short/motion/gmflow/transformer.py:76-101,
utils.py:5-51).  Here the same token permutation is a pair of int32 tables consumed
by emip_attention's row gather, and the shifted-window mask (transformer.py:19-43)
is a table of region ids compared inside the kernel."""
import math

import torch

_cache = {}


def window_tables(h, w, splits, shift, device):
    """rows[win][t] = row (y*w+x) in the un-rolled feature map of token t of window win;
    gid[win][t] = region id of that token in the rolled frame (mask = -100 where ids differ)."""
    key = ("win", h, w, splits, bool(shift), str(device))
    if key in _cache:
        return _cache[key]
    wh, ww = h // splits, w // splits
    sh, sw = (wh // 2, ww // 2) if shift else (0, 0)
    wy, wx, ty, tx = torch.meshgrid(torch.arange(splits), torch.arange(splits), torch.arange(wh), torch.arange(ww),
                                    indexing="ij")
    ry, rx = wy * wh + ty, wx * ww + tx                      # coordinates in the rolled frame
    oy, ox = (ry + sh) % h, (rx + sw) % w                    # torch.roll(x, -s): out[i] = x[(i+s) % n]
    rows = (oy * w + ox).reshape(splits * splits, wh * ww).to(torch.int32)

    def band(c, n, win, s):  # slices (0,-win), (-win,-s), (-s,None) of transformer.py:25-30
        return (c >= n - win).long() + (c >= n - s).long()
    gid = (band(ry, h, wh, wh // 2) * 3 + band(rx, w, ww, ww // 2)).reshape(splits * splits, wh * ww).to(torch.int32)
    out = (rows.contiguous().to(device), gid.contiguous().to(device))
    _cache[key] = out
    return out


def position_table(h, w, c, splits, dtype, device):
    """Sine position encoding of one (h/splits x w/splits) window tiled splits x splits,
    as rows [h*w, c].  position.py:26-46 + utils.py:66-86 (normalize=True, scale=2*pi,
    temperature=10000, channel order cat(pos_y, pos_x), sin on even / cos on odd)."""
    key = ("pos", h, w, c, splits, dtype, str(device))
    if key in _cache:
        return _cache[key]
    wh, ww, npf = h // splits, w // splits, c // 2
    ys = torch.arange(1, wh + 1, dtype=torch.float32) / (float(wh) + 1e-6) * (2 * math.pi)
    xs = torch.arange(1, ww + 1, dtype=torch.float32) / (float(ww) + 1e-6) * (2 * math.pi)
    i = torch.arange(npf, dtype=torch.float32)
    dim_t = 10000.0 ** (2 * torch.div(i, 2, rounding_mode="floor") / npf)
    even = (torch.arange(npf) % 2 == 0)

    def enc(v):  # [n] -> [n, npf]
        a = v[:, None] / dim_t
        return torch.where(even, a.sin(), a.cos())
    py = enc(ys)[:, None, :].expand(wh, ww, npf)
    px = enc(xs)[None, :, :].expand(wh, ww, npf)
    win = torch.cat((py, px), dim=2)                         # [wh, ww, c]
    full = win.repeat(splits, splits, 1).reshape(h * w, c)
    out = full.to(dtype).contiguous().to(device)
    _cache[key] = out
    return out


def grid_values(h, w, dtype, device):
    """V operand of the global-matching softmax: [h*w, 32] with columns (x, y, 0...).
    gmflow/geometry.py:5-21, matching.py:23-24."""
    key = ("grid", h, w, dtype, str(device))
    if key in _cache:
        return _cache[key]
    v = torch.zeros(h * w, 32)
    p = torch.arange(h * w)
    v[:, 0] = (p % w).float()
    v[:, 1] = torch.div(p, w, rounding_mode="floor").float()
    out = v.to(dtype).contiguous().to(device)
    _cache[key] = out
    return out
