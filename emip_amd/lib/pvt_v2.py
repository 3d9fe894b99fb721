"""PVTv2-b5 segmentation backbone on MI355X kernels.

Same module tree / state_dict keys as /root/reference/lib/pvt_v2.py (patch_embed{i}, block{i}.{j}.
{norm1,attn.{q,kv,sr,norm,proj},norm2,mlp.{fc1,dwconv.dwconv,fc2}}, norm{i}), so reference checkpoints
load unchanged.  The forward is re-designed for CDNA4: tokens stay channels-last [B,H,W,C] from the patch
embed to the stage output (no permutes), every Linear is the MFMA GEMM with bias/GELU/residual fused in
its epilogue, the spatial-reduction conv is an implicit GEMM over non-overlapping patches, and
softmax(q k^T) v runs in one fused attention kernel whose 121-key K/V tile lives in LDS.
"""
import torch
import torch.nn as nn

from .. import ops
from ..autograd import (DropPathAddFn, ConvFn, DwConvFn, LayerNormFn, LayerNormSkipFn, LinearFn, QSrFn,
                        SraAttentionFn)
from ..nn_base import EmipModule, conv_dgrad_pack, f32, lin_packs, pack_conv, pack_dw, pack_linear, to_cl, to_planar


# LayerNorm elimination on the inference path: the three LayerNorms of a block (norm1, norm2 and the one
# behind the spatial-reduction conv) are not launched at all.  The producer of the residual stream (proj / fc2 GEMM, the sr
# conv, the patch-embed LayerNorm) accumulates per-row (sum, sum of squares) in its epilogue, gamma is folded into the
# consumer's weights and beta into its bias when the weights are packed, and the consumer applies the normalisation on its
# OUTPUT side: LN(x) W^T = rstd (x W^T) - rstd mean colsum(W) (emip_gemm_lne; per tap for the spatial-reduction conv), so the
# main loops multiply raw rows.  The f32 parity mode keeps the operand-side form (emip_gemm_ln / emip_conv2d_ln).
# What was measured and dropped along the way (pair launch of q + sr conv, two split-K forms of the sr conv, the encoder on a
# forked stream, the generic attention kernel for the bf16 stages) lives in tools/experiments/README.md, not here.
FUSED_LN = True
# widest stage whose attention half ALWAYS runs as one launch (emip_sra_block: q + attention + proj + residual).  The
# 320-channel stage has too few 128-query workgroups for it at 16 images (35.8 us against 30.3 for emip_sra_qattn + the proj
# GEMM) and takes it from SRA_BLOCK_WIDE_ROWS token rows on: at the 32 images of a whole 16-pair step it is
# ahead (1587 against 1580 pairs/s in-process, tools/flag_ab.py) and 40 launches fewer
SRA_BLOCK_MAXC = 128
SRA_BLOCK_WIDE_ROWS = 12000    # round 4: with the proj GEMM's row statistics combined inside its launch (STATS_IN_LAUNCH) the two-launch form is reproducible too: at 16 images 11.88 against 12.54 ms one step at a time, 2243 pairs/s either way in flight
# q projection inside the attention launch (emip_sra_block / emip_sra_qattn).  False = q GEMM + emip_sra_attention, the form
# stage 4 (sr_ratio 1) always takes; tests/test_sra_block_gpu.py runs a block both ways and compares
SRA_FUSED = True
# the Mlp half of a block as ONE launch where the shape fits (emip_mlp_block: the 22 x 22 stage; bit-identical to
# emip_mlp_fc1dw + the fc2 GEMM, tests/test_mlp_block_gpu.py runs a block both ways).  OFF: measured on MI355X at 32 images
# the launch takes 119 us against 49 + 30 us for the two it replaces -- its fc1 / depthwise / fc2 phases run one after the
# other on 10 waves per CU (phase ablation: depthwise + GELU 43 us, fc2 26, fc1 18, H stores 8, 60 barriers + prologue 24) --
# and with three steps in flight the throughput follows the SUM of the kernels' isolated times, not the CUs a launch leaves
# free: 1401 against 1536 pairs/s (tools/flag_ab.py).  What it would take: DESIGN.md section 7d.
# row statistics of the proj / fc2 GEMMs whose rows span more than two column tiles (N = 320 on 128-wide tiles) combined inside
# the launch by the last column tile to finish (emip_gemm_ln_ws) instead of a row_stats launch behind it
STATS_IN_LAUNCH = True
SR_WIDE_TILE = False        # the spatial-reduction conv of the 22 x 22 stage: False = 64 x 128 tiles (gemm8 configuration 9: 93 workgroups, ring 5
                            # deep; its row statistics span three column tiles: combined inside the launch, STATS_IN_LAUNCH), 192 = two tiles of
                            # 64 x 192 (configuration 11, ring 4 deep), True = ONE 64 x 320 tile per 64 rows (configuration 10, ring 3 deep).
                            # Round 4, 16 pairs (pairs/s with 4 steps in flight | ms one step at a time): False 2369 | 10.79, 192: 2376 |
                            # 11.07, True 2373 | 11.34 -- round 3 preferred True when False still cost a statistics pass behind the launch
FC1DW_BAND_MIN_ROWS = 3     # banded fc1 + depthwise launch for maps of > 512 tokens when a band has at least this many output rows (0: off;
                            # in-call at 16 pairs: off 1592, the 44 x 44 stage only 1618, the 88 x 88 stage too 1628 pairs/s)
MLP_BLOCK = False
# the Mlp half of a 22 x 22-stage block as ONE launch of quarter- or eighth-image workgroups (emip_mlp_band, round 4): fc1,
# depthwise + GELU and fc2 as a three-stage software pipeline over 32-channel chunks, one barrier per chunk, the hidden tensor on
# the CU only; bit-identical to emip_mlp_fc1dw + the fc2 GEMM (tests/test_mlp_band_gpu.py).  OFF by measurement (MI355X, 16 pairs,
# tools/flag_ab.py, pairs/s with 4 steps in flight | ms one step at a time): two launches 2371 | 12.19; 4 bands 2348-2377 | 14.07;
# 8 bands 2296 | 13.00.  Inside a step a launch takes 99 us (4 bands) / 76 us (8 bands; 56 us with its weights already in L2)
# against 41 + 21 us: 80 launches per step fewer, no gain in either figure
MLP_BAND = False
# the spatial-reduction convs with few output tiles and a long K walk (stages 1-2: 61 tiles x 64 / 32 K tiles at 32 images)
# with K split inside the launch (emip_conv2d_ksplit, the normalising loader) instead of the per-tap ring body.  OFF: shorter
# alone, but with three steps in flight 1554 against 1587 pairs/s (tools/flag_ab.py) -- the third time a split-K form loses
# here: its f32 atomics and 3-5 x the workgroups cost the other steps' kernels more than the walk saves
SR_KSPLIT = False

_lin_packs, _conv_dgrad_pack = lin_packs, conv_dgrad_pack


class DWConv(EmipModule):
    """Parameter holder for the Mlp's depthwise 3x3 (pvt_v2.py:316-327)."""

    def __init__(self, dim=768):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)


class Mlp(EmipModule):
    """fc1 -> depthwise 3x3 -> GELU -> fc2 (pvt_v2.py:45-54); GELU is fused into the depthwise kernel."""

    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.dwconv = DWConv(hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)

    def run_train(self, h, residual, drop=None):
        dt = self.cdtype
        dw = self.dwconv.dwconv
        (w1, w1t), (w2, w2t), wd, wdf = self.packed(
            "mlp_t", (self.fc1.weight, self.fc2.weight, dw.weight),
            lambda a, b, c: (_lin_packs(a, dt), _lin_packs(b, dt), pack_dw(c), pack_dw(c, flip=True)))
        t = LinearFn.apply(h, self.fc1.weight, self.fc1.bias, None, w1, w1t)
        t = DwConvFn.apply(t, dw.weight, dw.bias, wd, wdf, True)
        return LinearFn.apply(t, self.fc2.weight, self.fc2.bias, residual, w2, w2t, None, drop)

    def run(self, h, residual, drop=None):
        if torch.is_grad_enabled():
            return self.run_train(h, residual, drop)
        dt = self.cdtype
        w1, b1, wd, bd, w2, b2 = self.packed(
            "mlp", (self.fc1.weight, self.fc1.bias, self.dwconv.dwconv.weight, self.dwconv.dwconv.bias,
                    self.fc2.weight, self.fc2.bias),
            lambda a, b, c, d, e, f: (pack_linear(a, dt), f32(b), pack_dw(c), f32(d), pack_linear(e, dt), f32(f)))
        t = ops.gemm(h, w1, bias=b1)
        t = ops.dwconv3x3(t, wd, bd, act=ops.ACT_GELU)
        return ops.gemm(t, w2, bias=b2, res=residual, out=residual)


class Attention(EmipModule):
    """Spatial-reduction attention (pvt_v2.py:57-129), head_dim 64, 121 keys in every stage."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, sr_ratio=1):
        super().__init__()
        assert dim % num_heads == 0, f"dim {dim} should be divided by num_heads {num_heads}."
        self.dim, self.num_heads, self.sr_ratio = dim, num_heads, sr_ratio
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim)

    def run_train(self, h, residual, drop=None):
        dt, C, heads, sr = self.cdtype, self.dim, self.num_heads, self.sr_ratio
        B, H, W, _ = h.shape
        (wq, wqt), (wkv, wkvt), (wp, wpt) = self.packed(
            "lin_t", (self.q.weight, self.kv.weight, self.proj.weight),
            lambda a, b, c: (_lin_packs(a, dt), _lin_packs(b, dt), _lin_packs(c, dt)))
        if sr > 1:
            wsr, wsrd = self.packed("sr_t", (self.sr.weight,),
                                    lambda a: (pack_conv(a, dt), _conv_dgrad_pack(a, dt, sr, sr, 0)))
            q, s_ = QSrFn.apply(h, self.q.weight, self.q.bias, self.sr.weight, self.sr.bias, wq, wqt, wsr, wsrd, sr)
            s_ = LayerNormFn.apply(s_, self.norm.weight, self.norm.bias, self.norm.eps)
        else:
            q = LinearFn.apply(h, self.q.weight, self.q.bias, None, wq, wqt)
            s_ = h
        Lk = s_.shape[1] * s_.shape[2]
        kv = LinearFn.apply(s_, self.kv.weight, self.kv.bias, None, wkv, wkvt)
        a = SraAttentionFn.apply(q.view(B, H * W, C), kv.view(B, Lk, 2 * C), heads, self.scale).view(B, H, W, C)
        return LinearFn.apply(a, self.proj.weight, self.proj.bias, residual, wp, wpt, None, drop)

    def run(self, h, residual, drop=None):
        """h: normed tokens [B,H,W,C]; returns residual + proj(attn) written in place."""
        if torch.is_grad_enabled():
            return self.run_train(h, residual, drop)
        dt, C, heads, sr = self.cdtype, self.dim, self.num_heads, self.sr_ratio
        B, H, W, _ = h.shape
        assert C // heads == 64, "the fused attention kernel is built for head_dim 64"
        wq, bq, wkv, bkv, wp, bp = self.packed(
            "lin", (self.q.weight, self.q.bias, self.kv.weight, self.kv.bias, self.proj.weight, self.proj.bias),
            lambda a, b, c, d, e, f: (pack_linear(a, dt), f32(b), pack_linear(c, dt), f32(d), pack_linear(e, dt),
                                      f32(f)))
        q = ops.gemm(h, wq, bias=bq)
        if sr > 1:
            wsr, bsr, g, be = self.packed("sr", (self.sr.weight, self.sr.bias, self.norm.weight, self.norm.bias),
                                          lambda a, b, c, d: (pack_conv(a, dt), f32(b), f32(c), f32(d)))
            s = ops.conv2d(h, wsr, sr, sr, sr, 0, bias=bsr)
            s = ops.layernorm(s, g, be, self.norm.eps)
        else:
            s = h
        Lk = s.shape[1] * s.shape[2]
        kv = ops.gemm(s, wkv, bias=bkv)                      # [B,h,w,2C]: k = [:C], v = [C:], head hd at hd*64
        N = H * W
        a = torch.empty((B, H, W, C), dtype=dt, device=h.device)
        if dt == torch.bfloat16 and Lk <= 128:
            ops.sra_attention(q, kv, a, B, heads, N, Lk, self.scale)
        else:
            ops.attention(q, kv, kv[..., C:], a, batch=B, heads=heads, nwin=1, Lq=N, Lk=Lk, D=64, DV=64, q_bs=N * C,
                          k_bs=Lk * 2 * C, v_bs=Lk * 2 * C, o_bs=N * C, ldq=C, ldk=2 * C, ldv=2 * C, ldo=C, q_hs=64,
                          k_hs=64, v_hs=64, o_hs=64, scale=self.scale)
        return ops.gemm(a, wp, bias=bp, res=residual, out=residual)


class Block(EmipModule):
    """Pre-LN residual block (pvt_v2.py:132-169).  DropPath acts in train mode only."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop_path=0., norm_layer=nn.LayerNorm,
                 sr_ratio=1):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, sr_ratio=sr_ratio)
        self.drop_path_rate = float(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio))

    def _drop_scale(self, x, tag):
        """per-sample DropPath factor Bernoulli(keep) / keep as (f32 [B], the same repeated over channels [B, C], that minus
        one -- the backward's coefficient).  `forced_drop` (dict tag -> [B] factors) overrides the draw (parity tests force
        the same factors on both sides); otherwise the stage drew the factors of all its blocks in one go (`_drop_pre`, see
        draw_drop_tables)."""
        B, C = x.shape[0], x.shape[-1]
        forced = getattr(self, "forced_drop", None)
        pre = getattr(self, "_drop_pre", None)
        if forced is None and pre is not None and pre[tag][1].shape == (B, C) and pre[tag][1].device == x.device:
            return pre[tag]
        if forced is not None:
            sb = forced[tag].to(device=x.device, dtype=torch.float32)
            if sb.numel() != B:          # factors of the whole batch, this block sees images _batch_lo .. _batch_lo + B - 1 (run(deep=))
                lo = getattr(self, "_batch_lo", 0)
                sb = sb[lo:lo + B]
            sb = sb.contiguous()
        else:
            keep = 1.0 - self.drop_path_rate
            sb = torch.floor(keep + torch.rand(B, device=x.device)) / keep
        s = sb.view(B, 1).expand(B, C).contiguous()
        return sb, s, s - 1.0

    def _folded(self):
        """packed weights with the block's LayerNorm affines folded in: y = LN(x) W^T + b = xhat (W*gamma)^T + (b + W beta)"""
        dt, a, m = self.cdtype, self.attn, self.mlp
        sr = a.sr_ratio

        def build(g1, b1, g2, b2, wq, bq, wkv, bkv, wp, bp, w1, bb1, wd, bd, w2, bb2, *srp):
            f = lambda t: t.detach().float()
            mv = lambda w_, v_: (w_ * v_).sum(1)      # W @ v as an elementwise product + row sum (no vendor BLAS at pack time)
            g1, b1, g2, b2 = f(g1), f(b1), f(g2), f(b2)
            out = dict(q=(f(wq) * g1).to(dt).contiguous(), bq=(f(bq) + mv(f(wq), b1)).contiguous(),
                       p=pack_linear(wp, dt), bp=f32(bp),
                       w1=(f(w1) * g2).to(dt).contiguous(), b1=(f(bb1) + mv(f(w1), b2)).contiguous(),
                       wd=pack_dw(wd), bd=f32(bd), w2=pack_linear(w2, dt), b2=f32(bb2))
            if sr > 1:
                wsr, bsr, gs, bs = (f(t) for t in srp)
                out.update(sr=pack_conv(wsr * g1.view(1, -1, 1, 1), dt),
                           bsr=(bsr + (wsr * b1.view(1, -1, 1, 1)).sum((1, 2, 3))).contiguous(),
                           kv=(f(wkv) * gs).to(dt).contiguous(), bkv=(f(bkv) + mv(f(wkv), bs)).contiguous())
            else:
                out.update(kv=(f(wkv) * g1).to(dt).contiguous(), bkv=(f(bkv) + mv(f(wkv), b1)).contiguous())
            # column sums of the packed (rounded) weights: the output-side form of the folded LayerNorm (emip_gemm_lne)
            for k in ("q", "kv", "w1"):
                out["s" + k] = out[k].float().sum(1).contiguous()
            if sr > 1 and dt == torch.bfloat16 and ops.sra_block_eligible(a.dim, 121):
                # emip_sra_block reads the q / proj weights in the order its MFMA accumulators have (bits 2, 3 of the row
                # index swapped inside every 16; for proj also of the column index); the vectors stay in channel order
                sw = ops.swap23(a.dim, wq.device)
                out["qf"] = out["q"][sw].contiguous()
                out["pf"] = f(wp).to(dt)[sw][:, sw].contiguous()
            if sr > 1:      # ... and per tap for the spatial-reduction conv (emip_conv8 with ln_stats): [sr*sr, C]
                out["tsr"] = out["sr"].float().view(out["sr"].shape[0], sr * sr, -1).sum(2).t().contiguous()
            if dt == torch.bfloat16 and out["w1"].shape == (1280, 320):
                # emip_mlp_block: the per-chunk constant blocks (taps, depthwise bias, fc1 bias, column sums of the packed W1)
                out["mcst"] = ops.mlp_block_consts(out["wd"], out["bd"], out["b1"], out["sw1"])
                # emip_mlp_band: the pipeline stages (fragment-order W1 | W2 | fc1 constants) and the chunk-major tap table
                out["mband"] = ops.mlp_band_packs(out["w1"], out["b1"], out["sw1"], out["w2"], out["wd"], out["bd"])
            return out
        params = (self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias, a.q.weight, a.q.bias,
                  a.kv.weight, a.kv.bias, a.proj.weight, a.proj.bias, m.fc1.weight, m.fc1.bias,
                  m.dwconv.dwconv.weight, m.dwconv.dwconv.bias, m.fc2.weight, m.fc2.bias)
        if sr > 1:
            params += (a.sr.weight, a.sr.bias, a.norm.weight, a.norm.bias)
        return self.packed("fln", params, build)

    @staticmethod
    def scratch_floats(B, H, W, C, sr):
        """f32 words of statistics scratch one block needs: [sr-conv rows | x1 rows | x2 rows] x 2"""
        M = B * H * W
        Ms = B * (H // sr) * (W // sr) if sr > 1 else 0
        return 2 * (Ms + 2 * M)

    def run_fused(self, x, stats, buf, alt=None, ws=None):
        """Inference block without LayerNorm launches.  x [B,H,W,C] (updated in place), stats f32 [B*H*W, 2] = (sum, sum of
        squares) of its rows, buf: this block's slice of the stage's ZEROED scratch, alt: a second token buffer of x's shape
        (the one-launch Mlp half writes out of place), ws: the stage's two workspaces (token GEMMs, sr conv) for row statistics
        combined inside a launch (ops.gemm / ops.conv8 stats_ws; their ticket blocks lie in the zeroed scratch) -> (new x, stats of its rows, the token buffer
        that is free now)."""
        a = self.attn
        dt, C, heads, sr = self.cdtype, a.dim, a.num_heads, a.sr_ratio
        B, H, W, _ = x.shape
        N = H * W
        M = B * N
        w = self._folded()
        bf = dt == torch.bfloat16
        ws, ws_sr = ws if ws is not None else (None, None)
        Ms = B * (H // sr) * (W // sr) if sr > 1 else 0
        st_sr, st1, st2 = buf[:2 * Ms], buf[2 * Ms:2 * Ms + 2 * M], buf[2 * Ms + 2 * M:]
        cs = (lambda k: w["s" + k]) if bf else (lambda k: None)       # bf16: output-side LayerNorm (column sums); f32: loader
        # ---- attention half: x += proj(softmax(q k^T scale) v)
        if sr > 1:
            ks = ops.ksplit_for(Ms, C, sr * sr * C, dt) if (SR_KSPLIT and bf) else 0
            if ks:   # 8 x 8 / 4 x 4 reductions: few output tiles walking 32-64 K tiles -> K split inside the launch
                s = ops.conv2d_ksplit(x, w["sr"], sr, sr, sr, 0, ks, bias=w["bsr"], ln_stats=stats, ln_eps=self.norm1.eps,
                                      out_stats=st_sr)
            elif bf:   # raw patches on the LDS-DMA ring, LayerNorm per tap on the output side (the statistics ride the ring)
                # that launch form keeps a tile's tap sums in LDS: one tile per workgroup, at most 256 tiles -- beyond (more than
                # ~44 images at C = 320) the batch goes in image chunks
                rows_img = (H // sr) * (W // sr)
                wide = SR_WIDE_TILE and C == 320           # N = 320 in one 64 x 320 tile (or two of 192): the token panel is read once (twice)
                scfg = (11 if SR_WIDE_TILE == 192 else 10) if wide else 0
                per_tile, ntile_n = (128, 1) if C <= 64 else (64, (2 if SR_WIDE_TILE == 192 else 1) if wide else (C + 127) // 128)
                per = max(1, ((256 // ntile_n) * per_tile) // rows_img)
                if B <= per:
                    s = ops.conv8(x, w["sr"], sr, sr, sr, 0, bias=w["bsr"], ln_stats=stats, tapsum=w["tsr"],
                                  ln_eps=self.norm1.eps, out_stats=st_sr, cfg=scfg,
                                  stats_ws=ws_sr if (STATS_IN_LAUNCH and not wide) else None)
                else:
                    s = torch.empty((B, H // sr, W // sr, C), dtype=dt, device=x.device)
                    for b0 in range(0, B, per):
                        b1 = min(B, b0 + per)
                        ops.conv8(x[b0:b1], w["sr"], sr, sr, sr, 0, bias=w["bsr"], ln_stats=stats[2 * b0 * N:2 * b1 * N],
                                  tapsum=w["tsr"], ln_eps=self.norm1.eps, out=s[b0:b1],
                                  out_stats=st_sr[2 * b0 * rows_img:2 * b1 * rows_img], cfg=scfg)
            else:
                s = ops.conv2d(x, w["sr"], sr, sr, sr, 0, bias=w["bsr"], ln_stats=stats, ln_eps=self.norm1.eps, out_stats=st_sr)
            kv = ops.gemm(s, w["kv"], bias=w["bkv"], ln_stats=st_sr, ln_eps=a.norm.eps, colsum=cs("kv"))
        else:
            s = x
            kv = ops.gemm(x, w["kv"], bias=w["bkv"], ln_stats=stats, ln_eps=self.norm1.eps, colsum=cs("kv"))
        Lk = s.shape[1] * s.shape[2]
        fused = SRA_FUSED and "qf" in w and ops.sra_block_eligible(C, Lk)
        if fused and (C <= SRA_BLOCK_MAXC or M >= SRA_BLOCK_WIDE_ROWS):
            # q projection + attention + proj + residual in ONE launch
            ops.sra_block(x, stats, self.norm1.eps, w["qf"], w["bq"], w["sq"], kv.view(B, -1, 2 * C), w["pf"], w["bp"], heads,
                          a.scale, out_stats=st1)
        else:
            if fused:
                # q + attention per (image, queries, head), the proj GEMM as a launch of its own
                att = ops.sra_qattn(x, stats, self.norm1.eps, w["qf"], w["bq"], w["sq"], kv.view(B, -1, 2 * C), heads, a.scale)
            else:
                q = ops.gemm(x, w["q"], bias=w["bq"], ln_stats=stats, ln_eps=self.norm1.eps, colsum=cs("q"))
                att = torch.empty((B, H, W, C), dtype=dt, device=x.device)
                if bf and Lk <= 128:
                    ops.sra_attention(q, kv, att, B, heads, N, Lk, a.scale)      # keys resident in registers, queries streamed
                else:
                    ops.attention(q, kv, kv[..., C:], att, batch=B, heads=heads, nwin=1, Lq=N, Lk=Lk, D=64, DV=64, q_bs=N * C,
                                  k_bs=Lk * 2 * C, v_bs=Lk * 2 * C, o_bs=N * C, ldq=C, ldk=2 * C, ldv=2 * C, ldo=C, q_hs=64,
                                  k_hs=64, v_hs=64, o_hs=64, scale=a.scale)
            ops.gemm(att, w["p"], bias=w["bp"], res=x, out=x, out_stats=st1, stats_ws=ws if STATS_IN_LAUNCH else None)
        # ---- Mlp half: x += fc2(GELU(dwconv(fc1(LN(x)))))
        hid = w["w1"].shape[0]
        if MLP_BAND and alt is not None and "mband" in w and ops.mlp_band_eligible(B, H, W, C, hid):
            # the whole Mlp half in one launch of B x 4 workgroups (out of place: a band's halo tokens are its neighbours' rows)
            ops.mlp_band(x, w["mband"][0], w["mband"][1], w["b2"], st1, self.norm2.eps, alt, out_stats=st2)
            return alt, st2, x
        if MLP_BLOCK and alt is not None and "mcst" in w and ops.mlp_block_eligible(B, H, W, C, hid):
            # the whole Mlp half in one launch: the hidden tensor never leaves the CU (out of place: bands read halo rows)
            ops.mlp_block(x, w["w1"], w["w2"], w["mcst"], w["b2"], st1, self.norm2.eps, alt, out_stats=st2)
            return alt, st2, x
        if bf and (ops.mlp_fc1dw_eligible(B, H, W, C, hid) or (0 < FC1DW_BAND_MIN_ROWS <= ops.mlp_fc1dw_band_rows(B, H, W, C, hid))):
            # fc1 + depthwise 3x3 + GELU in one launch, one whole image per workgroup: the fc1 output never leaves the CU
            t = ops.mlp_fc1dw(x, w["w1"], w["b1"], w["sw1"], st1, self.norm2.eps, w["wd"], w["bd"])
        else:
            t = ops.gemm(x, w["w1"], bias=w["b1"], ln_stats=st1, ln_eps=self.norm2.eps, colsum=cs("w1"))
            t = ops.dwconv3x3(t, w["wd"], w["bd"], act=ops.ACT_GELU)
        ops.gemm(t, w["w2"], bias=w["b2"], res=x, out=x, out_stats=st2, stats_ws=ws if STATS_IN_LAUNCH else None)
        return x, st2, alt

    def run(self, x):
        if self.training and self.drop_path_rate > 0:
            # stochastic depth: the branch is computed without the fused residual, then scaled per sample
            if not torch.is_grad_enabled():
                # train mode under torch.no_grad() (the frozen short-term part of train_long.py): same Functions, whose
                # inputs do not require grad, so nothing is recorded
                with torch.enable_grad():
                    return self.run(x)
            rps = x.shape[1] * x.shape[2]
            # the per-sample factor is applied in the epilogue of the branch's last GEMM (proj / fc2), which also adds the skip
            h, xs = LayerNormSkipFn.apply(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            x = self.attn.run(h, xs, drop=self._drop_scale(x, "attn") + (rps,))
            h, xs = LayerNormSkipFn.apply(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            return self.mlp.run(h, xs, drop=self._drop_scale(x, "mlp") + (rps,))
        if torch.is_grad_enabled():
            h, xs = LayerNormSkipFn.apply(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            x = self.attn.run(h, xs)
            h, xs = LayerNormSkipFn.apply(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            return self.mlp.run(h, xs)
        g1, b1, g2, b2 = self.packed("ln", (self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias),
                                     lambda a, b, c, d: (f32(a), f32(b), f32(c), f32(d)))
        x = self.attn.run(ops.layernorm(x, g1, b1, self.norm1.eps), x)
        x = self.mlp.run(ops.layernorm(x, g2, b2, self.norm2.eps), x)
        return x


_KEEP_CACHE = {}


def draw_drop_tables(blocks, B, C, device):
    """Stochastic-depth factors of one stage in a handful of launches instead of six per residual branch (612 per
    EMIP-short step): one uniform draw [2 n, B] for the n blocks' attention and Mlp branches, floor(keep + u) / keep per
    row (timm's DropPath as used by pvt_v2.py:167-169), expanded over the C channels the scaling kernel reads."""
    keeps = []
    for b in blocks:
        keeps += [1.0 - b.drop_path_rate] * 2
    if all(k >= 1.0 for k in keeps):
        for b in blocks:
            object.__setattr__(b, "_drop_pre", None)
        return
    ck = (str(device), tuple(keeps))
    k = _KEEP_CACHE.get(ck)
    if k is None:                                   # the keep probabilities of a stage never change: upload them once
        k = _KEEP_CACHE[ck] = torch.tensor(keeps, dtype=torch.float32).to(device).view(-1, 1)
    sb = (torch.floor(k + torch.rand(len(keeps), B, device=device)) / k).contiguous()        # [2 n, B]
    s = sb.view(-1, B, 1).expand(-1, B, C).contiguous()
    sm1 = s - 1.0
    for i, b in enumerate(blocks):
        object.__setattr__(b, "_drop_pre", {"attn": (sb[2 * i], s[2 * i], sm1[2 * i]),
                                            "mlp": (sb[2 * i + 1], s[2 * i + 1], sm1[2 * i + 1])})


class OverlapPatchEmbed(EmipModule):
    """Overlapping strided conv + LayerNorm (pvt_v2.py:172-214)."""

    def __init__(self, img_size=224, patch_size=7, stride=4, in_chans=3, embed_dim=768):
        super().__init__()
        assert patch_size > stride, "Set larger patch_size than stride"
        self.patch_size, self.stride = patch_size, stride
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=patch_size // 2)
        self.norm = nn.LayerNorm(embed_dim)

    def run(self, x, out_stats=None, zero=None):
        dt = self.cdtype
        cin = x.shape[-1]  # the image arrives with its 3 channels zero-padded to 8
        if torch.is_grad_enabled():
            k, st = self.patch_size, self.stride
            wp, wdg = self.packed("pe_t", (self.proj.weight,),
                                  lambda a: (pack_conv(a, dt, cin_pad=cin), _conv_dgrad_pack(a, dt, k, st, k // 2)))
            y = ConvFn.apply(x, self.proj.weight, self.proj.bias, wp, wdg, k, st, k // 2, cin)
            return LayerNormFn.apply(y, self.norm.weight, self.norm.bias, self.norm.eps)
        w, b, g, be = self.packed("pe", (self.proj.weight, self.proj.bias, self.norm.weight, self.norm.bias),
                                  lambda a, bb, c, d: (pack_conv(a, dt, cin_pad=cin), f32(bb), f32(c), f32(d)))
        y = ops.conv2d(x, w, self.patch_size, self.patch_size, self.stride, self.patch_size // 2, bias=b, zero=zero)
        return ops.layernorm(y, g, be, self.norm.eps, out=y, out_stats=out_stats)


class PyramidVisionTransformerV2(EmipModule):
    def __init__(self, img_size=224, in_chans=3, embed_dims=(64, 128, 256, 512), num_heads=(1, 2, 4, 8),
                 mlp_ratios=(4, 4, 4, 4), qkv_bias=False, drop_path_rate=0., norm_layer=nn.LayerNorm,
                 depths=(3, 4, 6, 3), sr_ratios=(8, 4, 2, 1), num_stages=4, pretrained=None):
        super().__init__()
        self.depths, self.num_stages = depths, num_stages
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        cur = 0
        for i in range(num_stages):
            setattr(self, f"patch_embed{i + 1}", OverlapPatchEmbed(
                img_size=img_size if i == 0 else img_size // (2 ** (i + 1)), patch_size=7 if i == 0 else 3,
                stride=4 if i == 0 else 2, in_chans=in_chans if i == 0 else embed_dims[i - 1],
                embed_dim=embed_dims[i]))
            setattr(self, f"block{i + 1}", nn.ModuleList([
                Block(dim=embed_dims[i], num_heads=num_heads[i], mlp_ratio=mlp_ratios[i], qkv_bias=qkv_bias,
                      drop_path=dpr[cur + j], norm_layer=norm_layer, sr_ratio=sr_ratios[i])
                for j in range(depths[i])]))
            setattr(self, f"norm{i + 1}", norm_layer(embed_dims[i]))
            cur += depths[i]

    def run(self, x, deep=None, fork=None):
        """x: channels-last image [B,H,W,8] -> list of the 4 stage outputs, channels-last.
        deep = (lo, hi): only the images lo .. hi - 1 of the batch go on past stage 2 (the outputs of stages 3 and 4 then hold
        hi - lo images) -- for callers that read the deep features of part of the batch only (CoUpdater.run).
        fork: a HIP stream.  Stages 3 and 4 are enqueued on it (behind everything the current stream holds so far), so that
        the caller's following launches run beside them; the caller makes its stream wait for `fork` before it reads outs[2:]
        (inside a graph capture this is a fork / join of the graph).  Under autograd the Functions of stages 3 and 4 record `fork`
        as their stream, so their backward kernels run there too, beside the backward of whatever the caller enqueued meanwhile
        (the engine orders the gradient edges between the streams)."""
        outs = []
        cur = torch.cuda.current_stream() if fork is not None else None
        try:
            return self._run_stages(x, deep, fork, cur, outs)
        finally:
            if fork is not None:
                torch.cuda.set_stream(cur)

    def _run_stages(self, x, deep, fork, cur, outs):
        for i in range(self.num_stages):
            if i == 2 and fork is not None:
                fork.wait_stream(cur)
                x.record_stream(fork)
                torch.cuda.set_stream(fork)
            if i == 2 and deep is not None:
                x = x[deep[0]:deep[1]]
            pe = getattr(self, f"patch_embed{i + 1}")
            blocks = getattr(self, f"block{i + 1}")
            if self.training:
                for blk in blocks:       # which images of the caller's batch the block sees (forced DropPath factors are per image)
                    object.__setattr__(blk, "_batch_lo", deep[0] if (deep is not None and i >= 2) else 0)
            if FUSED_LN and not torch.is_grad_enabled() and not (self.training and any(b.drop_path_rate > 0 for b in blocks)):
                # no LayerNorm launches inside the blocks: row statistics travel with the residual stream
                st = pe.stride
                Ho, Wo = (x.shape[1] + 2 * (pe.patch_size // 2) - pe.patch_size) // st + 1, \
                         (x.shape[2] + 2 * (pe.patch_size // 2) - pe.patch_size) // st + 1
                stats = torch.empty(2 * x.shape[0] * Ho * Wo, dtype=torch.float32, device=x.device)
                # statistics scratch of the whole stage: one allocation, cleared by the patch-embed conv's workgroups
                a0 = blocks[0].attn
                per = Block.scratch_floats(x.shape[0], Ho, Wo, a0.dim, a0.sr_ratio)
                # + the stage's workspace for row statistics combined inside a launch (tickets first: they must be zero)
                # (two of them: the sr conv's output has other row counts, hence another ticket block, than the token GEMMs')
                bf16 = x.dtype == torch.bfloat16
                wsb = (ops.gemm_stats_ws_bytes(x.shape[0] * Ho * Wo, a0.dim) + 63) // 64 * 64 if bf16 else 0
                srr = a0.sr_ratio
                wsb2 = ops.gemm_stats_ws_bytes(x.shape[0] * (Ho // srr) * (Wo // srr), a0.dim) if (bf16 and srr > 1) else 0
                off = (len(blocks) * per + 15) // 16 * 16
                scratch = torch.empty(off + (wsb + wsb2 + 3) // 4, dtype=torch.float32, device=x.device)
                wsu = scratch[off:].view(torch.uint8)
                ws = (wsu[:wsb], wsu[wsb:wsb + wsb2] if wsb2 else None) if wsb else None
                x = pe.run(x, out_stats=stats, zero=scratch)
                alt = torch.empty_like(x)
                for j, blk in enumerate(blocks):
                    x, stats, alt = blk.run_fused(x, stats, scratch[j * per:(j + 1) * per], alt, ws)
            else:
                x = pe.run(x)
                if self.training:
                    draw_drop_tables(blocks, x.shape[0], x.shape[-1], x.device)
                for blk in blocks:
                    x = blk.run(x)
            norm = getattr(self, f"norm{i + 1}")
            if torch.is_grad_enabled():
                x = LayerNormFn.apply(x, norm.weight, norm.bias, norm.eps)
                if fork is not None and i >= 2:
                    x.record_stream(cur)
                outs.append(x)
                continue
            g, b = self.packed(f"n{i}", (norm.weight, norm.bias), lambda a, c: (f32(a), f32(c)))
            x = ops.layernorm(x, g, b, norm.eps)
            if fork is not None and i >= 2:
                x.record_stream(cur)
            outs.append(x)
        return outs

    def forward_features(self, x):
        return [to_planar(o) for o in self.run(to_cl(x, self.cdtype, 8))]

    def forward(self, x):
        return self.forward_features(x)


class pvt_v2_b5(PyramidVisionTransformerV2):
    """embed 64/128/320/512, heads 1/2/5/8, depths 3/6/40/3, sr 8/4/2/1, LN eps 1e-6 (pvt_v2.py:395-401)."""

    def __init__(self, **kwargs):
        from functools import partial
        super().__init__(embed_dims=[64, 128, 320, 512], num_heads=[1, 2, 5, 8], mlp_ratios=[4, 4, 4, 4],
                         qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), depths=[3, 6, 40, 3],
                         sr_ratios=[8, 4, 2, 1], drop_path_rate=0.1, pretrained=kwargs.get("pretrained"))
