"""EMIP-long surface: Model_long(args).forward(frame0, frame1, index, memory_k, memory_v).

Drop-in for /root/reference/model/EMIP_long/model_long.py:52-117: same constructor dictionary, same forward
signature/returns ((mask [1,1,352,352], keys, values) with keys/values [1,1,128,T<=5,44,44]), same state_dict
(LTM.*, short_term.*, long_dr.*, injector1.*, decoder.*, dr1.*).  `forward_streams` is the batched form of the
same step for S independent video streams (BASELINE.json configs[3]: 8 streams): the reference is hard-wired to
one stream, every op here simply carries the stream index as its batch dimension.

What is NOT reproduced: the reference's second conv_corr pass on `corr_bw` (model_long.py:80-84), whose result
is never used (65 GFLOP of dead work per frame).
"""
import torch

from ...nn_base import EmipModule
from ..EMIP_short.create_backbone import DimensionalReduction, NeighborConnectionDecoder
from ..EMIP_short.model import CoUpdater
from ..EMIP_short.motion.PromptInteract import Injector
from .LTM import LTM


class Model_long(EmipModule):
    WINDOW = 5  # frames kept in memory (model_long.py:105-107)

    def __init__(self, args=None):
        super().__init__()
        self.args = args
        self.channel = args['channel']
        self.LTM = LTM()
        self.short_term = CoUpdater(args)
        self.long_dr = DimensionalReduction(256, 128)
        self.injector1 = Injector()
        self.decoder = NeighborConnectionDecoder(32)
        self.dr1 = DimensionalReduction(128, 32)
        object.__setattr__(self, "_mem_cache", None)

    # ---- memory layout helpers: reference layout [S,1,C,T,h,w] planar f32 <-> channels-last [S,T,h*w,C]
    @staticmethod
    def _mem_to_ref(m, h, w):
        S, T, n, C = m.shape
        return m.float().view(S, T, h, w, C).permute(0, 4, 1, 2, 3).unsqueeze(1).contiguous()

    def _mem_from_ref(self, m):
        S, _, C, T, h, w = m.shape
        return m[:, 0].permute(0, 2, 3, 4, 1).reshape(S, T, h * w, C).to(self.cdtype).contiguous()

    def _lookup(self, ref_k, ref_v):
        if ref_k.requires_grad or ref_v.requires_grad:
            raise RuntimeError("feed the memory back detached (train_long.py:52-53)")
        """Reuse the channels-last memory of the previous step when the caller feeds back what we returned
        (possibly .detach()-ed: same storage), instead of converting layouts every frame."""
        c = self._mem_cache
        def sig(t):
            return (t.data_ptr(), t._version, tuple(t.shape))
        if c is not None and c[0] == sig(ref_k) and c[1] == sig(ref_v) and c[2].dtype == self.cdtype:
            return c[2], c[3]
        return self._mem_from_ref(ref_k), self._mem_from_ref(ref_v)

    def forward_streams(self, frames0, frames1, index, memory_k, memory_v):
        """frames0/frames1: [S,3,H,W]; memory_k/v: [S,1,128,T,44,44] or None.  Returns (masks [S,1,H,W], keys, values)."""
        if torch.is_grad_enabled() and not self.training and any(p.requires_grad for p in self.parameters()):
            # eval mode with autograd on: same values as the reference, computed without a graph (see CoUpdater.forward)
            with torch.no_grad():
                return self.forward_streams(frames0, frames1, index, memory_k, memory_v)
        st = self.short_term
        S = frames0.shape[0]
        if index == 0:
            with torch.no_grad():
                mask, _ = st.run(frames0, frames1)
            return mask, None, None
        mk = mv = None
        if not (index == 1 or memory_k is None):
            mk, mv = self._lookup(memory_k, memory_v)
        mask_long, keys, values = self.step_cl(frames0, frames1, mk, mv)
        h = w = int(round(keys.shape[2] ** 0.5))
        ref_k, ref_v = self._mem_to_ref(keys, h, w), self._mem_to_ref(values, h, w)
        object.__setattr__(self, "_mem_cache", ((ref_k.data_ptr(), ref_k._version, tuple(ref_k.shape)),
                                                (ref_v.data_ptr(), ref_v._version, tuple(ref_v.shape)), keys.detach(),
                                                values.detach()))
        return mask_long, ref_k, ref_v

    def step_a(self, frames0, frames1):
        """The part of a step that does not see the memory: short-term encoders (model_long.py:70-79), the two side reductions and
        the key / value pair this frame contributes to the memory (LTM.memorize, :97-103).  Returns
        (f0 features of the second frame [S,h,w,128], f2_2, f2_3, pk, pv [S,1,h*w,128])."""
        st = self.short_term
        S = frames0.shape[0]
        with torch.no_grad():                  # model_long.py:70: the short-term part never carries gradient
            st.run(frames0, frames1, tail=False)       # its mask is not used from frame 1 on: no short-term decoder
            L = st.last
            fea, cc = L["fea"], L["conv_corr"]
            h, w = fea[0].shape[1:3]
            # the deep features of the SECOND frame (with PVT_DEEP_ONE_FRAME the short-term part computed only those)
            f2_2 = st.dr2.run(fea[1][S:] if fea[1].shape[0] == 2 * S else fea[1])
            f2_3 = st.dr3.run(fea[2][S:] if fea[2].shape[0] == 2 * S else fea[2])
        pk, pv = self.LTM.memorize_cl(fea[0][:S], cc)                  # [S,h,w,128] each
        return fea[0][S:], f2_2, f2_3, pk.view(S, 1, h * w, -1), pv.view(S, 1, h * w, -1)

    def step_b(self, f0s, f2_2, f2_3, keys, values):
        """The part that reads the memory window (model_long.py:105-117): LTM.segment over the window's keys / values (their order
        does not matter: one softmax over all of them), long_dr, injector1, dr1, decoder -> mask [S,1,H,W]"""
        mem = self.LTM.segment_cl(f0s, keys, values)                    # [S,h,w,256]
        mem = self.long_dr.run(mem)
        fl = self.injector1.run(f0s, mem)
        fl = self.dr1.run(fl)
        return self.decoder.run(f2_3, f2_2, fl)

    def step_cl(self, frames0, frames1, mem_k, mem_v):
        """One step on channels-last memory (no layout conversion, no host logic beyond shapes): frames [S,3,H,W],
        mem_k / mem_v [S,T,h*w,128] or None -> (mask [S,1,H,W], keys, values [S,min(T+1,5),h*w,128]).
        With a full window the shapes are static, which is what emip_amd.graph.GraphedLong captures."""
        f0s, f2_2, f2_3, pk, pv = self.step_a(frames0, frames1)
        if mem_k is None:
            keys, values = pk, pv
        else:
            keys = torch.cat([mem_k, pk], 1)[:, -self.WINDOW:].contiguous()
            values = torch.cat([mem_v, pv], 1)[:, -self.WINDOW:].contiguous()
        return self.step_b(f0s, f2_2, f2_3, keys, values), keys, values

    def forward(self, frame0, frame1, index, memory_k, memory_v):
        return self.forward_streams(frame0.unsqueeze(0), frame1.unsqueeze(0), index, memory_k, memory_v)
