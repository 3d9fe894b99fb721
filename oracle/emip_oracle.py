"""CPU oracle for the EMIP two-stream path.  TEST INFRASTRUCTURE ONLY.

A plain PyTorch-fp32 restatement of the reference's algorithm, written as
stateless functions over a flat ``state_dict`` (name -> tensor).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this file; the product (``emip_amd``) never does and fails loudly
when its HIP library is missing.

Parity pin: ``oracle/make_golden.py`` imports the reference itself (in the
build container, where /root/reference exists) and writes fixtures under
``tests/golden``; ``tests/test_oracle_golden.py`` checks every function here
against those fixtures.  The arithmetic below runs in ATen (torch 2.10 CPU);
the reference pins no torch version (README.md:21-25).

Every function cites the reference lines it restates (paths under
/root/reference).
"""
import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# small helpers


def _ln(x, sd, p, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _conv(x, sd, p, stride=1, padding=0, groups=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding, groups=groups)


def _bn(x, sd, p, training=False):
    # nn.BatchNorm2d(eps=1e-5, momentum=0.1); train mode uses batch statistics
    # (per replica, no SyncBN: train.py:279).  Running buffers are not updated here.
    if training:
        return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.0, 1e-5)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, 1e-5)


def _conv_br(x, sd, p, training=False):
    """ConvBR = conv3x3(no bias) + BN + ReLU.  create_backbone.py:22-36."""
    return F.relu(_bn(_conv(x, sd, p + ".conv", padding=1), sd, p + ".bn", training))


# --------------------------------------------------------------------------
# PVTv2-b5   (lib/pvt_v2.py)

PVT_DIMS = (64, 128, 320, 512)
PVT_HEADS = (1, 2, 5, 8)
PVT_DEPTHS = (3, 6, 40, 3)
PVT_SR = (8, 4, 2, 1)


def pvt_sra(x, H, W, sd, p, heads, sr):
    """Spatial-reduction attention.  lib/pvt_v2.py:101-129."""
    B, N, C = x.shape
    d = C // heads
    q = _lin(x, sd, p + ".q").reshape(B, N, heads, d).permute(0, 2, 1, 3)
    if sr > 1:
        x_ = x.permute(0, 2, 1).reshape(B, C, H, W)
        x_ = _conv(x_, sd, p + ".sr", stride=sr).reshape(B, C, -1).permute(0, 2, 1)
        x_ = _ln(x_, sd, p + ".norm", 1e-5)  # plain nn.LayerNorm -> eps 1e-5 (pvt_v2.py:78)
    else:
        x_ = x
    kv = _lin(x_, sd, p + ".kv").reshape(B, -1, 2, heads, d).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    attn = (q @ k.transpose(-2, -1)) * (d ** -0.5)
    attn = attn.softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return _lin(out, sd, p + ".proj")


def pvt_mlp(x, H, W, sd, p):
    """fc1 -> depthwise 3x3 -> GELU(erf) -> fc2.  lib/pvt_v2.py:45-54,321-327."""
    B, N, _ = x.shape
    h = _lin(x, sd, p + ".fc1")
    C = h.shape[-1]
    h = h.transpose(1, 2).reshape(B, C, H, W)
    h = _conv(h, sd, p + ".dwconv.dwconv", padding=1, groups=C)
    h = h.flatten(2).transpose(1, 2)
    return _lin(F.gelu(h), sd, p + ".fc2")


def pvt_forward(img, sd, p, drop_masks=None):
    """pvt_v2_b5 forward_features; returns the 4 stage outputs NCHW.
    lib/pvt_v2.py:291-306,395-401.  ``drop_masks``: optional dict
    {(stage, block, 'attn'|'mlp'): per-sample scale [B]} to restate DropPath
    (train mode, pvt_v2.py:165-167) deterministically."""
    x = img
    outs = []
    for i in range(4):
        pe = f"{p}.patch_embed{i + 1}"
        k, s = (7, 4) if i == 0 else (3, 2)
        x = _conv(x, sd, pe + ".proj", stride=s, padding=k // 2)
        B, C, H, W = x.shape
        x = x.flatten(2).transpose(1, 2)
        x = _ln(x, sd, pe + ".norm", 1e-5)  # OverlapPatchEmbed.norm: eps 1e-5 (pvt_v2.py:189)
        for j in range(PVT_DEPTHS[i]):
            bp = f"{p}.block{i + 1}.{j}"
            a = pvt_sra(_ln(x, sd, bp + ".norm1", 1e-6), H, W, sd, bp + ".attn", PVT_HEADS[i], PVT_SR[i])
            if drop_masks is not None and (i, j, "attn") in drop_masks:
                a = a * drop_masks[(i, j, "attn")].view(-1, 1, 1)
            x = x + a
            m = pvt_mlp(_ln(x, sd, bp + ".norm2", 1e-6), H, W, sd, bp + ".mlp")
            if drop_masks is not None and (i, j, "mlp") in drop_masks:
                m = m * drop_masks[(i, j, "mlp")].view(-1, 1, 1)
            x = x + m
        x = _ln(x, sd, f"{p}.norm{i + 1}", 1e-6)
        x = x.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
        outs.append(x)
    return outs


# --------------------------------------------------------------------------
# GMFlow (model/EMIP_short/motion/gmflow)


def _inorm(x):
    return F.instance_norm(x, eps=1e-5)  # affine=False, no running stats (backbone.py:40)


def gm_resblock(x, sd, p, stride):
    """ResidualBlock.  gmflow/backbone.py:39-69."""
    y = F.relu(_inorm(_conv(x, sd, p + ".conv1", stride=stride, padding=1)))
    y = F.relu(_inorm(_conv(y, sd, p + ".conv2", padding=1)))
    if (p + ".downsample.0.weight") in sd:
        x = _inorm(_conv(x, sd, p + ".downsample.0", stride=stride))
    return F.relu(x + y)


def gm_cnn(img, sd, p):
    """CNNEncoder, one output scale at 1/8.  gmflow/backbone.py:154-192."""
    x = F.relu(_inorm(_conv(img, sd, p + ".conv1", stride=2, padding=3)))
    for name, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
        x = gm_resblock(x, sd, f"{p}.{name}.0", stride)
        x = gm_resblock(x, sd, f"{p}.{name}.1", 1)
    return _conv(x, sd, p + ".conv2")


def gm_position(h, w, c=128, temperature=10000.0):
    """Sine position table for one h x w window, [c, h, w].
    gmflow/position.py:26-46 (normalize=True, scale=2*pi, num_pos_feats=c/2)."""
    npf = c // 2
    y_embed = torch.arange(1, h + 1, dtype=torch.float32).view(h, 1).expand(h, w)
    x_embed = torch.arange(1, w + 1, dtype=torch.float32).view(1, w).expand(h, w)
    y_embed = y_embed / (float(h) + 1e-6) * (2 * math.pi)
    x_embed = x_embed / (float(w) + 1e-6) * (2 * math.pi)
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / npf)
    pos_x = x_embed[:, :, None] / dim_t
    pos_y = y_embed[:, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, 0::2].sin(), pos_x[:, :, 1::2].cos()), dim=3).flatten(2)
    pos_y = torch.stack((pos_y[:, :, 0::2].sin(), pos_y[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((pos_y, pos_x), dim=2).permute(2, 0, 1).contiguous()


def gm_add_position(f, splits=2):
    """feature_add_position with attn_splits=2: the window table tiled 2x2.
    gmflow/utils.py:66-86."""
    B, C, H, W = f.shape
    pos = gm_position(H // splits, W // splits, C)
    return f + pos.repeat(1, splits, splits).unsqueeze(0)


def gm_shift_mask(h, w, splits=2):
    """Additive -100 mask of the shifted windows, [splits^2, L, L].
    gmflow/transformer.py:19-43."""
    wh, ww = h // splits, w // splits
    sh, sw = wh // 2, ww // 2
    ids = torch.zeros(h, w)
    cnt = 0
    for hs in (slice(0, -wh), slice(-wh, -sh), slice(-sh, None)):
        for ws in (slice(0, -ww), slice(-ww, -sw), slice(-sw, None)):
            ids[hs, ws] = cnt
            cnt += 1
    win = ids.view(splits, wh, splits, ww).permute(0, 2, 1, 3).reshape(splits * splits, wh * ww)
    diff = win.unsqueeze(1) - win.unsqueeze(2)
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def _split(x, s):  # [B,H,W,C] -> [B*s*s, H/s*W/s, C]      gmflow/utils.py:5-28
    B, H, W, C = x.shape
    return x.view(B, s, H // s, s, W // s, C).permute(0, 1, 3, 2, 4, 5).reshape(B * s * s, (H // s) * (W // s), C)


def _merge(x, s, H, W):  # inverse of _split                 gmflow/utils.py:31-51
    C = x.shape[-1]
    B = x.shape[0] // (s * s)
    return x.view(B, s, s, H // s, W // s, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)


def gm_window_attention(q, k, v, h, w, shift, mask, splits=2):
    """single_head_split_window_attention.  gmflow/transformer.py:46-105."""
    B, L, C = q.shape
    q, k, v = (t.view(B, h, w, C) for t in (q, k, v))
    sh, sw = h // splits // 2, w // splits // 2
    if shift:
        q, k, v = (torch.roll(t, (-sh, -sw), (1, 2)) for t in (q, k, v))
    q, k, v = (_split(t, splits) for t in (q, k, v))
    scores = q @ k.transpose(1, 2) / (C ** 0.5)
    if shift:
        scores = scores + mask.repeat(B, 1, 1)
    out = _merge(scores.softmax(-1) @ v, splits, h, w)
    if shift:
        out = torch.roll(out, (sh, sw), (1, 2))
    return out.reshape(B, L, C)


def gm_layer(src, tgt, sd, p, h, w, shift, mask, ffn):
    """TransformerLayer.  gmflow/transformer.py:156-196."""
    q = _lin(src, sd, p + ".q_proj")
    k = _lin(tgt, sd, p + ".k_proj")
    v = _lin(tgt, sd, p + ".v_proj")
    msg = gm_window_attention(q, k, v, h, w, shift, mask)
    msg = _ln(_lin(msg, sd, p + ".merge"), sd, p + ".norm1", 1e-5)
    if ffn:
        msg = torch.cat([src, msg], dim=-1)
        msg = _lin(F.gelu(_lin(msg, sd, p + ".mlp.0")), sd, p + ".mlp.2")
        msg = _ln(msg, sd, p + ".norm2", 1e-5)
    return src + msg


def gm_transformer(f0, f1, sd, p, layers=6):
    """FeatureTransformer.  gmflow/transformer.py:433-482."""
    B, C, H, W = f0.shape
    t0 = f0.flatten(2).permute(0, 2, 1)
    t1 = f1.flatten(2).permute(0, 2, 1)
    mask = gm_shift_mask(H, W)
    c0 = torch.cat((t0, t1), 0)
    c1 = torch.cat((t1, t0), 0)
    for i in range(layers):
        lp = f"{p}.layers.{i}"
        shift = i % 2 == 1
        c0 = gm_layer(c0, c0, sd, lp + ".self_attn", H, W, shift, mask, False)
        c0 = gm_layer(c0, c1, sd, lp + ".cross_attn_ffn", H, W, shift, mask, True)
        c1 = torch.cat(c0.chunk(2, 0)[::-1], 0)
    a, b = c0.chunk(2, 0)
    back = lambda t: t.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
    return back(a), back(b)


def coords_grid(h, w):
    """[2,h,w] pixel grid, channel 0 = x.  gmflow/geometry.py:5-21."""
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([x, y], 0).float()


def gm_global_match(f0, f1):
    """global_correlation_softmax with pred_bidir_flow=True.
    gmflow/matching.py:8-41.  Returns (flow [2B,2,h,w], corr [B, h*w(tgt), h, w(src)])."""
    B, C, H, W = f0.shape
    a = f0.view(B, C, -1).permute(0, 2, 1)
    b = f1.view(B, C, -1)
    corr = torch.matmul(a, b) / (C ** 0.5)  # [B, src, tgt]
    corr_out = corr.view(B, H, W, H * W).permute(0, 3, 1, 2)
    grid = coords_grid(H, W)
    g = grid.view(2, -1).t()  # [HW, 2]
    both = torch.cat((corr, corr.permute(0, 2, 1)), 0)
    prob = both.softmax(-1)
    corresp = torch.matmul(prob, g).view(2 * B, H, W, 2).permute(0, 3, 1, 2)
    return corresp - grid.unsqueeze(0), corr_out


def gm_flow_attention(f, flow, sd, p):
    """FeatureFlowAttention (k is projected from the projected q).
    gmflow/transformer.py:503-533."""
    B, C, H, W = f.shape
    q = _lin(f.view(B, C, H * W).permute(0, 2, 1), sd, p + ".q_proj")
    k = _lin(q, sd, p + ".k_proj")
    v = flow.view(B, 2, H * W).permute(0, 2, 1)
    prob = (q @ k.transpose(1, 2) / (C ** 0.5)).softmax(-1)
    return (prob @ v).view(B, H, W, 2).permute(0, 3, 1, 2)


def gm_convex_upsample(flow, feat, sd, p, factor=8):
    """GMFlow.upsample_flow, convex branch.  gmflow/gmflow.py:56-79."""
    x = torch.cat((flow, feat), 1)
    m = _conv(F.relu(_conv(x, sd, p + ".0", padding=1)), sd, p + ".2")
    B, _, H, W = flow.shape
    m = m.view(B, 1, 9, factor, factor, H, W).softmax(2)
    up = F.unfold(factor * flow, [3, 3], padding=1).view(B, 2, 9, 1, 1, H, W)
    up = (m * up).sum(2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(B, 2, factor * H, factor * W)


def gmflow_forward(a, b, sd, p, training=False):
    """GMFlow.forward for num_scales=1, bidirectional.  gmflow/gmflow.py:81-162.
    Returns (flow_fw list, flow_bw list, corr, extras)."""
    f0, f1 = gm_add_position(a), gm_add_position(b)
    f0, f1 = gm_transformer(f0, f1, sd, p + ".transformer")
    flow, corr = gm_global_match(f0, f1)
    preds = []
    if training:
        preds.append(F.interpolate(flow, scale_factor=8, mode="bilinear", align_corners=True) * 8)
    feat = torch.cat((f0, f1), 0)
    flow_prop = gm_flow_attention(feat, flow.detach(), sd, p + ".feature_flow_attn")
    preds.append(gm_convex_upsample(flow_prop, feat, sd, p + ".upsampler"))
    fw = [t[: t.shape[0] // 2] for t in preds]
    bw = [t[t.shape[0] // 2:] for t in preds]
    return fw, bw, corr, {"f0": f0, "f1": f1, "flow_lr": flow, "flow_prop": flow_prop}


# --------------------------------------------------------------------------
# Injector = one MDTA block (model/EMIP_short/motion/PromptInteract.py)


def _ln2d(x, sd, p):
    """WithBias LayerNorm over C of an NCHW tensor (eps 1e-5, biased var).
    PromptInteract.py:333-362."""
    t = x.permute(0, 2, 3, 1)
    t = F.layer_norm(t, (t.shape[-1],), sd[p + ".body.weight"], sd[p + ".body.bias"], 1e-5)
    return t.permute(0, 3, 1, 2)


def injector_forward(x, y, sd, p, heads=2):
    """Injector.forward -> TransformerBlock_MDTA.  PromptInteract.py:367-464."""
    t = p + ".transformer"
    B, C, H, W = x.shape
    xn, yn = _ln2d(x, sd, t + ".norm1"), _ln2d(y, sd, t + ".norm2")
    q = _conv(_conv(xn, sd, t + ".attn.q"), sd, t + ".attn.q_dwconv", padding=1, groups=C)
    kv = _conv(_conv(yn, sd, t + ".attn.kv"), sd, t + ".attn.kv_dwconv", padding=1, groups=2 * C)
    k, v = kv.chunk(2, 1)
    q, k, v = (u.reshape(B, heads, C // heads, H * W) for u in (q, k, v))
    q, k = F.normalize(q, dim=-1), F.normalize(k, dim=-1)
    attn = ((q @ k.transpose(-2, -1)) * sd[t + ".attn.temperature"]).softmax(-1)
    out = (attn @ v).reshape(B, C, H, W)
    x = x + _conv(out, sd, t + ".attn.project_out")
    h = _conv(_ln2d(x, sd, t + ".norm3"), sd, t + ".ffn.project_in")
    h = _conv(h, sd, t + ".ffn.dwconv", padding=1, groups=h.shape[1])
    h1, h2 = h.chunk(2, 1)
    return x + _conv(F.gelu(h1) * h2, sd, t + ".ffn.project_out")


# --------------------------------------------------------------------------
# decoder side (model/EMIP_short/create_backbone.py, model.py)


def dim_reduction(x, sd, p, training=False):
    """DimensionalReduction = 2 x ConvBR.  create_backbone.py:199-208."""
    return _conv_br(_conv_br(x, sd, p + ".reduce.0", training), sd, p + ".reduce.1", training)


def ncd_forward(zt5, zt4, zt3, sd, p, training=False, return_pc=False):
    """NeighborConnectionDecoder.forward.  create_backbone.py:61-76."""
    up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True)
    cb = lambda t, n: _conv_br(t, sd, f"{p}.{n}", training)
    zt4_1 = cb(up(zt5), "conv_upsample1") * zt4
    zt3_1 = cb(up(zt4_1), "conv_upsample2") * cb(up(zt4), "conv_upsample3") * zt3
    zt4_2 = cb(torch.cat((zt4_1, cb(up(zt5), "conv_upsample4")), 1), "conv_concat2")
    zt3_2 = cb(torch.cat((zt3_1, cb(up(zt4_2), "conv_upsample5")), 1), "conv_concat3")
    pc = _conv(cb(zt3_2, "conv4"), sd, p + ".conv5")
    res = F.interpolate(pc, scale_factor=8, mode="bilinear")  # align_corners=False
    return (res, pc) if return_pc else res


def conv_corr_forward(corr, sd, p, training=False):
    """conv3x3(1936->968)+BN+ReLU+conv3x3(968->128).  model.py:59-62."""
    x = F.relu(_bn(_conv(corr, sd, p + ".0", padding=1), sd, p + ".1", training))
    return _conv(x, sd, p + ".3", padding=1)


def short_forward(image1, image2, sd, p="", training=False, drop_masks=None, capture=None):
    """CoUpdater.forward.  model/EMIP_short/model.py:86-102.
    ``capture``: optional dict filled with intermediate tensors."""
    pv = p + "backbone.feat_net.pvtv2_en"
    dm1 = dm2 = None
    if drop_masks is not None:
        dm1, dm2 = drop_masks
    fea_1 = pvt_forward(image1, sd, pv, dm1)[1:]
    fea_2 = pvt_forward(image2, sd, pv, dm2)[1:]
    g1 = gm_cnn(image1, sd, p + "GMFlow.backbone")
    g2 = gm_cnn(image2, sd, p + "GMFlow.backbone")
    a = injector_forward(g1, fea_1[0], sd, p + "injector")
    b = injector_forward(g2, fea_2[0], sd, p + "injector")
    fw, bw, corr, ex = gmflow_forward(a, b, sd, p + "GMFlow", training)
    cc = conv_corr_forward(corr, sd, p + "conv_corr", training)
    fea_new_ori = injector_forward(fea_1[0], cc, sd, p + "injector1")
    fea_new = dim_reduction(fea_new_ori, sd, p + "dr1", training)
    f_2 = dim_reduction(fea_1[1], sd, p + "dr2", training)
    f_3 = dim_reduction(fea_1[2], sd, p + "dr3", training)
    mask, pc = ncd_forward(f_3, f_2, fea_new, sd, p + "decoder", training, return_pc=True)
    if capture is not None:
        capture.update(pvt1=fea_1, pvt2=fea_2, gm1=g1, gm2=g2, inj_a=a, inj_b=b, corr=corr, conv_corr=cc,
                       inj1=fea_new_ori, dr1=fea_new, dr2=f_2, dr3=f_3, pc=pc, **ex)
    return mask, fw, bw


# --------------------------------------------------------------------------
# EMIP-long (model/EMIP_long)


def ltm_memorize(fea0, corr, sd, p, training=False):
    """LTM.memorize: fusion conv on fea+corr, then Key/Value convs.
    LTM.py:38-41,78-79,103-111.  Returns k,v [1,1,128,1,44,44]."""
    f = p + ".fusion.conv1_fusion"
    x = fea0 + corr
    x = _conv(F.relu(_bn(_conv(x, sd, f + ".0", padding=1), sd, f + ".1", training)), sd, f + ".3", padding=1)
    k = _conv(x, sd, p + ".KV_M_r4.Key", padding=1)
    v = _conv(x, sd, p + ".KV_M_r4.Value", padding=1)
    return k[None, :, :, None], v[None, :, :, None]


def ltm_segment(fea0, keys, values, sd, p):
    """LTM.segment + Memory.forward.  LTM.py:49-68,122-132.  keys/values
    [1,1,128,T,44,44]; returns [1,256,44,44]."""
    kq = _conv(fea0, sd, p + ".KV_Q_r4.Key", padding=1)
    vq = _conv(fea0, sd, p + ".KV_Q_r4.Value", padding=1)
    m_in, m_out = keys[0], values[0]
    B, D, T, H, W = m_in.shape
    mi = m_in.reshape(B, D, T * H * W).transpose(1, 2)
    qi = kq.reshape(B, D, H * W)
    pmat = (torch.bmm(mi, qi) / math.sqrt(D)).softmax(dim=1)
    mem = torch.bmm(m_out.reshape(B, D, T * H * W), pmat).view(B, D, H, W)
    return torch.cat([mem, vq], 1)


def long_forward(frame0, frame1, index, memory_k, memory_v, sd, training=False, drop_masks=None):
    """Model_long.forward.  model/EMIP_long/model_long.py:68-117.  training=True: train_long.py:37 puts the whole
    model in train mode (batch-statistics BatchNorm, DropPath via drop_masks) while the short-term part runs under
    torch.no_grad() (model_long.py:70) -- its results are constants of the step."""
    cap = {}
    st = "short_term."
    with torch.no_grad():
        mask, _, _ = short_forward(frame0[None], frame1[None], sd, st, training=training, drop_masks=drop_masks,
                                   capture=cap)
        if index == 0:
            return mask, None, None
        f2_2 = dim_reduction(cap["pvt2"][1], sd, st + "dr2", training)
        f2_3 = dim_reduction(cap["pvt2"][2], sd, st + "dr3", training)
    if training:
        cap = {k: (tuple(t.detach() for t in v) if isinstance(v, (tuple, list)) else v.detach()) for k, v in cap.items()
               if torch.is_tensor(v) or isinstance(v, (tuple, list))}
        pk, pvv = ltm_memorize(cap["pvt1"][0], cap["conv_corr"], sd, "LTM", True)
        keys = pk if index == 1 else torch.cat([memory_k, pk], 3)[:, :, :, -5:]
        values = pvv if index == 1 else torch.cat([memory_v, pvv], 3)[:, :, :, -5:]
        mem = ltm_segment(cap["pvt2"][0], keys, values, sd, "LTM")
        mem = dim_reduction(mem, sd, "long_dr", True)
        fl = injector_forward(cap["pvt2"][0], mem, sd, "injector1")
        fl = dim_reduction(fl, sd, "dr1", True)
        return ncd_forward(f2_3, f2_2, fl, sd, "decoder", True), keys, values
    pk, pvv = ltm_memorize(cap["pvt1"][0], cap["conv_corr"], sd, "LTM")
    if index == 1:
        keys, values = pk, pvv
    else:
        keys = torch.cat([memory_k, pk], 3)[:, :, :, -5:]
        values = torch.cat([memory_v, pvv], 3)[:, :, :, -5:]
    mem = ltm_segment(cap["pvt2"][0], keys, values, sd, "LTM")
    mem = dim_reduction(mem, sd, "long_dr")
    fl = injector_forward(cap["pvt2"][0], mem, sd, "injector1")
    fl = dim_reduction(fl, sd, "dr1")
    return ncd_forward(f2_3, f2_2, fl, sd, "decoder"), keys, values


# --------------------------------------------------------------------------
# losses (loss/)


def mesh_grid(B, H, W):
    """loss/warp_utils.py:7-13.  int64 [B,2,H,W], channel 0 = x."""
    x = torch.arange(0, W).view(1, 1, W).expand(B, H, W)
    y = torch.arange(0, H).view(1, H, 1).expand(B, H, W)
    return torch.stack([x, y], 1)


def flow_warp(x, flow):
    """Bilinear warp, border padding, align_corners=True.  loss/warp_utils.py:16-23,83-93."""
    B, _, H, W = x.shape
    g = mesh_grid(B, H, W).type_as(x) + flow
    gx = 2.0 * g[:, 0] / (W - 1) - 1.0
    gy = 2.0 * g[:, 1] / (H - 1) - 1.0
    return F.grid_sample(x, torch.stack([gx, gy], -1), mode="bilinear", padding_mode="border", align_corners=True)


def corresponding_indices(data):
    """Corner indices / weights of get_corresponding_map.  loss/warp_utils.py:26-70.
    Returns (indices int64 [B,4N], values [B,4N]) in the reference's corner order."""
    B, _, H, W = data.shape
    x = data[:, 0].reshape(B, -1)
    y = data[:, 1].reshape(B, -1)
    x1, y1 = torch.floor(x), torch.floor(y)
    xf, yf = x1.clamp(0, W - 1), y1.clamp(0, H - 1)
    x0, y0 = x1 + 1, y1 + 1
    xc, yc = x0.clamp(0, W - 1), y0.clamp(0, H - 1)
    xco, yco, xfo, yfo = x0 != xc, y0 != yc, x1 != xf, y1 != yf
    invalid = torch.cat([xco | yco, xco | yfo, xfo | yco, xfo | yfo], 1)
    idx = torch.cat([xc + yc * W, xc + yf * W, xf + yc * W, xf + yf * W], 1).long()
    wx_c, wx_f = 1 - (x - xc).abs(), 1 - (x - xf).abs()
    wy_c, wy_f = 1 - (y - yc).abs(), 1 - (y - yf).abs()
    vals = torch.cat([wx_c * wy_c, wx_c * wy_f, wx_f * wy_c, wx_f * wy_f], 1)
    vals = torch.where(invalid, torch.zeros_like(vals), vals)
    return idx, vals


def occu_mask_backward(flow21, th=0.2):
    """get_occu_mask_backward.  loss/warp_utils.py:72-80,106-112."""
    B, _, H, W = flow21.shape
    idx, vals = corresponding_indices(mesh_grid(B, H, W).type_as(flow21) + flow21)
    cmap = torch.zeros(B, H * W).type_as(flow21).scatter_add_(1, idx, vals).view(B, 1, H, W)
    return (cmap.clamp(0.0, 1.0) < th).float()


def ssim_dist(x, y):
    """SSIM distance with 3x3 average pooling.  loss/loss_blocks.py:46-65."""
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ap = lambda t: F.avg_pool2d(t, 3, 1, 0)
    mx, my = ap(x), ap(y)
    sx, sy, sxy = ap(x * x) - mx * mx, ap(y * y) - my * my, ap(x * y) - mx * my
    s = ((2 * mx * my + C1) * (2 * sxy + C2)) / ((mx * mx + my * my + C1) * (sx + sy + C2))
    return ((1 - s) / 2).clamp(0, 1)


def photometric(im, rec, m):
    """loss_photomatric: 0.15*L1 + 0.85*SSIM, each .mean()-ed, / mask.mean().
    loss/loss_flow.py:35-49."""
    l1 = (0.15 * (im - rec).abs() * m).mean()
    ss = (0.85 * ssim_dist(rec * m, im * m)).mean()
    return (l1 + ss) / m.mean()


def unflow_loss(flows, images):
    """unFlowLoss.compute_loss: returns total (= warp) loss.  loss/loss_flow.py:60-138.
    flows: list of [B,4,H,W] (fw;bw), images [B,6,H,W]."""
    im1, im2 = images[:, :3], images[:, 3:]
    total = 0.0
    occ1 = occ2 = None
    for i, flow in enumerate(flows):
        h, w = flow.shape[-2:]
        i1 = F.interpolate(im1, (h, w), mode="area")
        i2 = F.interpolate(im2, (h, w), mode="area")
        r1 = flow_warp(i2, flow[:, :2])
        r2 = flow_warp(i1, flow[:, 2:])
        if i == 0:
            occ1 = 1 - occu_mask_backward(flow[:, 2:])
            occ2 = 1 - occu_mask_backward(flow[:, :2])
            m1, m2 = occ1, occ2
        else:
            m1 = F.interpolate(occ1, (h, w), mode="nearest")
            m2 = F.interpolate(occ2, (h, w), mode="nearest")
        total = total + (photometric(i1, r1, m1) + photometric(i2, r2, m2)) / 2.0
    return total


def hybrid_e_loss(pred, mask):
    """BCE-with-logits + E-loss + soft IoU.  loss/loss_pred.py:4-22."""
    wbce = F.binary_cross_entropy_with_logits(pred, mask, reduction="mean")
    p = torch.sigmoid(pred)
    phi_f = p - p.mean(dim=(2, 3), keepdim=True)
    phi_g = mask - mask.mean(dim=(2, 3), keepdim=True)
    efm = (2.0 * phi_f * phi_g + 1e-8) / (phi_f * phi_f + phi_g * phi_g + 1e-8)
    eloss = 1.0 - ((1 + efm) * (1 + efm) / 4.0).mean(dim=(2, 3))
    inter = (p * mask).sum(dim=(2, 3))
    union = (p + mask).sum(dim=(2, 3))
    wiou = 1.0 - (inter + 1 + 1e-8) / (union - inter + 1 + 1e-8)
    return (wbce + eloss + wiou).mean()


# --------------------------------------------------------------------------
# After the path


def postprocess_mask(mask, shape):
    """The reference's prediction post-processing, statement for statement: test.py:28-31 (bilinear resize to the source
    frame, sigmoid, per-image min-max, x255, PIL 'F' -> 'L').  mask [1,1,H,W] f32 -> uint8 [Ho,Wo] numpy."""
    import numpy as np
    from PIL import Image
    out = F.interpolate(mask, size=tuple(shape), mode="bilinear", align_corners=False)
    out = out.sigmoid().data.cpu().numpy().squeeze()
    out = (out - out.min()) / (out.max() - out.min() + 1e-8)
    return np.asarray(Image.fromarray(out * 255).convert("L"))


def preprocess_rgb(img_u8, size=352):
    """The reference's per-frame transform (dataset.py:257-260 / :76-79): torchvision Resize((size, size)) on the PIL
    image (= PIL.Image.resize with BILINEAR, torchvision/transforms/_functional_pil.py), ToTensor (u8 -> f32 / 255, CHW)
    and Normalize(ImageNet mean, std).  torchvision itself is not installed here; Pillow, which does the arithmetic, is.
    img_u8: numpy uint8 [H,W,3] -> (f32 tensor [3,size,size], resized uint8 [size,size,3])"""
    import numpy as np
    from PIL import Image
    pil = Image.fromarray(img_u8, "RGB").resize((size, size), Image.BILINEAR)
    arr = np.asarray(pil)
    t = torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous().to(dtype=torch.float32).div(255)
    mean = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float32).view(3, 1, 1)
    return t.sub_(mean).div_(std), arr


# --------------------------------------------------------------------------
# Validation metrics (eval/metrics.py), array-level restatement in numpy


def metric_prepare(pred, gt):
    """_prepare_data, eval/metrics.py:20-25"""
    gt = gt > 128
    pred = pred / 255
    if pred.max() != pred.min():
        pred = (pred - pred.min()) / (pred.max() - pred.min())
    return pred, gt


def metric_mae(pred, gt):
    """MAE.cal_mae, eval/metrics.py:100-102"""
    import numpy as np
    pred, gt = metric_prepare(pred, gt)
    return float(np.mean(np.abs(pred - gt)))


def metric_smeasure(pred, gt, alpha=0.5):
    """Smeasure.cal_sm and helpers, eval/metrics.py:120-213"""
    import numpy as np
    eps = np.spacing(1)
    pred, gt = metric_prepare(pred, gt)
    y = np.mean(gt)
    if y == 0:
        return float(1 - np.mean(pred))
    if y == 1:
        return float(np.mean(pred))

    def s_object(p, g):
        x = np.mean(p[g == 1])
        sigma = np.std(p[g == 1], ddof=1)
        return 2 * x / (np.power(x, 2) + 1 + sigma + eps)

    obj = y * s_object(pred * gt, gt) + (1 - y) * s_object((1 - pred) * (1 - gt), 1 - gt)
    h, w = gt.shape
    area = np.sum(gt)
    cx = int(np.round(np.sum(np.sum(gt, axis=0) * np.arange(w)) / area)) + 1
    cy = int(np.round(np.sum(np.sum(gt, axis=1) * np.arange(h)) / area)) + 1

    def ssim(p, g):
        n = p.shape[0] * p.shape[1]
        x, yv = np.mean(p), np.mean(g)
        sx = np.sum((p - x) ** 2) / (n - 1)
        sy = np.sum((g - yv) ** 2) / (n - 1)
        sxy = np.sum((p - x) * (g - yv)) / (n - 1)
        a = 4 * x * yv * sxy
        b = (x ** 2 + yv ** 2) * (sx + sy)
        if a != 0:
            return a / (b + eps)
        return 1 if (a == 0 and b == 0) else 0

    n = h * w
    w1, w2, w3 = cx * cy / n, cy * (w - cx) / n, (h - cy) * cx / n
    w4 = 1 - w1 - w2 - w3
    region = (w1 * ssim(pred[0:cy, 0:cx], gt[0:cy, 0:cx]) + w2 * ssim(pred[0:cy, cx:w], gt[0:cy, cx:w]) +
              w3 * ssim(pred[cy:h, 0:cx], gt[cy:h, 0:cx]) + w4 * ssim(pred[cy:h, cx:w], gt[cy:h, cx:w]))
    return float(max(0, alpha * obj + (1 - alpha) * region))


def metric_wfm(pred, gt, beta=1.0, return_parts=False):
    """WeightedFmeasure.step / cal_wfm, eval/metrics.py:338-383 (scipy's distance_transform_edt and convolve are the
    reference's own dependencies for this metric)"""
    import numpy as np
    from scipy.ndimage import convolve, distance_transform_edt
    eps = np.spacing(1)
    pred, gt = metric_prepare(pred, gt)
    if np.all(~gt):
        return 0.0
    dst, idx = distance_transform_edt(gt == 0, return_indices=True)
    e = np.abs(pred - gt)
    et = np.copy(e)
    et[gt == 0] = et[idx[0][gt == 0], idx[1][gt == 0]]
    y, x = np.ogrid[-3:4, -3:4]                                     # matlab_style_gauss2D((7, 7), sigma=5), :385-393
    k = np.exp(-(x * x + y * y) / (2.0 * 5 * 5))
    k[k < np.finfo(k.dtype).eps * k.max()] = 0
    k /= k.sum()
    ea = convolve(et, weights=k, mode="constant", cval=0)
    min_e_ea = np.where(gt & (ea < e), ea, e)
    b = np.where(gt == 0, 2 - np.exp(np.log(0.5) / 5 * dst), np.ones_like(gt))
    ew = min_e_ea * b
    tpw = np.sum(gt) - np.sum(ew[gt == 1])
    fpw = np.sum(ew[gt == 0])
    r = 1 - np.mean(ew[gt == 1])
    p = tpw / (tpw + fpw + eps)
    q = (1 + beta) * r * p / (r + beta * p + eps)
    if return_parts:
        return float(q), idx, et
    return float(q)


# ---- training-time augmentation (dataset/data_augment.py:12-45, dataset/dataset.py:76-103) ------------------------------
# The reference's functions are thin calls into Pillow (its own dependency, present in the image); the restatement keeps
# them as such, on numpy uint8 arrays, consuming `random` / `np.random` in the reference's order.  The reference module
# itself is not importable here (it imports cv2, which the image lacks), so these are pinned by Pillow directly.

def aug_random_rotation(img1, img2, label):
    """randomRotation, data_augment.py:12-19"""
    import random
    import numpy as np
    from PIL import Image
    if random.random() > 0.8:
        random_angle = np.random.randint(-15, 15)
        img1, img2, label = (np.asarray(Image.fromarray(a).rotate(random_angle, Image.BICUBIC)) for a in (img1, img2, label))
    return img1, img2, label


def aug_color_enhance(image, factors=None):
    """colorEnhance, data_augment.py:22-31 (factors=None draws them like the reference)"""
    import random
    import numpy as np
    from PIL import Image, ImageEnhance
    im = Image.fromarray(image, "RGB")
    f = factors
    b = random.randint(5, 15) / 10.0 if f is None else f[0]
    im = ImageEnhance.Brightness(im).enhance(b)
    c = random.randint(5, 15) / 10.0 if f is None else f[1]
    im = ImageEnhance.Contrast(im).enhance(c)
    k = random.randint(0, 20) / 10.0 if f is None else f[2]
    im = ImageEnhance.Color(im).enhance(k)
    s = random.randint(0, 30) / 10.0 if f is None else f[3]
    im = ImageEnhance.Sharpness(im).enhance(s)
    return np.asarray(im)


def aug_random_peper(img):
    """randomPeper, data_augment.py:34-45"""
    import random
    import numpy as np
    img = np.array(img)
    noise_num = int(0.0015 * img.shape[0] * img.shape[1])
    for _ in range(noise_num):
        rx = random.randint(0, img.shape[0] - 1)
        ry = random.randint(0, img.shape[1] - 1)
        img[rx, ry] = 0 if random.randint(0, 1) == 0 else 255
    return img


def preprocess_gray(gt_u8, size=352):
    """gt_transform, dataset/dataset.py:80-82: Resize((size, size)) on the 'L' image + ToTensor -> (f32 [1,S,S], u8 [S,S])"""
    import numpy as np
    import torch
    from PIL import Image
    r = np.asarray(Image.fromarray(gt_u8, "L").resize((size, size), Image.BILINEAR))
    return (torch.from_numpy(r.astype(np.float32)) / 255).unsqueeze(0), r


def train_sample(image1, image2, gt, size=352):
    """ObjDataset.__getitem__ after decoding, dataset/dataset.py:94-103"""
    image1, image2, gt = aug_random_rotation(image1, image2, gt)
    image1 = aug_color_enhance(image1)
    image2 = aug_color_enhance(image2)
    gt = aug_random_peper(gt)
    return preprocess_rgb(image1, size)[0], preprocess_rgb(image2, size)[0], preprocess_gray(gt, size)[0]
