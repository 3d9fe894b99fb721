"""Fused clamp + AdamW for the EMIP training step (one HIP launch for all parameters).

Mirrors the reference's optimizer usage: `clip_gradient(optimizer, 0.5)` (utils/utils.py:1-11, an ELEMENT-WISE
clamp, not a norm clip) followed by `torch.optim.AdamW(lr=1e-5, weight_decay=1e-7).step()` (train.py:61-62,380).
State layout follows torch.optim.AdamW (exp_avg, exp_avg_sq, step) so checkpoints stay interchangeable."""
import struct

import torch

from . import _lib


class FusedClampAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-7, clip=0.5):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip=clip)
        super().__init__(params, defaults)
        self._tables = {}

    def _table(self, gi, plist):
        """device tables for one param group: the static part (parameter / moment pointers, sizes, block map) is rebuilt only
        when those tensors move; the gradient pointers live in their own small array, re-uploaded when any of them changed
        (gradients are fresh tensors every step unless the step's gradient arena hands out the same slices again)"""
        key = tuple((p.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr())
                    for p in plist)
        hit = self._tables.get(gi)
        if hit is None or hit["key"] != key:
            chunk = _lib.load().emip_adamw_chunk()
            recs, bmap = bytearray(), []
            for i, p in enumerate(plist):
                st = self.state[p]
                recs += struct.pack("<QQQQq", p.data_ptr(), 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                bmap += [(i, c) for c in range((p.numel() + chunk - 1) // chunk)]
            dev = plist[0].device
            hit = dict(key=key, recs=torch.frombuffer(recs, dtype=torch.uint8).clone().to(dev),
                       bmap=torch.tensor(bmap, dtype=torch.int32).to(dev), nb=len(bmap), gkey=None,
                       gdev=torch.empty(len(plist), dtype=torch.int64, device=dev), turn=0,
                       ghost=[torch.empty(len(plist), dtype=torch.int64).pin_memory() for _ in range(2)], gev=[None, None])
            self._tables[gi] = hit
        gkey = [p.grad.data_ptr() for p in plist]
        if hit["gkey"] != gkey:
            hit["gkey"] = gkey
            k = hit["turn"] = hit["turn"] ^ 1                 # two pinned staging buffers: the copy that read this one
            if hit["gev"][k] is not None:                     # (two uploads ago) has long finished; wait if it has not
                hit["gev"][k].synchronize()
            hit["ghost"][k].copy_(torch.tensor(gkey, dtype=torch.int64))
            hit["gdev"].copy_(hit["ghost"][k], non_blocking=True)    # stream-ordered in front of the launch
            hit["gev"][k] = torch.cuda.Event()
            hit["gev"][k].record()
        return hit

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            fresh = [p for p in plist if not self.state[p]]
            for p in fresh:
                assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
                st = self.state[p]
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p)
                st["exp_avg_sq"] = torch.zeros_like(p)
            st0 = self.state[plist[0]]["step"]
            if fresh or gi not in self._tables:               # (the full check once; afterwards the group steps together)
                assert all(p.grad.is_contiguous() for p in plist)
                steps = {int(self.state[p]["step"]) for p in plist}       # torch.optim.AdamW keeps `step` as a tensor
                assert len(steps) == 1, "parameters of one group must share the step count"
            step = int(st0) + 1
            t = self._table(gi, plist)
            b1, b2 = group["betas"]
            _lib.call("emip_clamp_adamw_g", t["recs"].data_ptr(), t["bmap"].data_ptr(), t["gdev"].data_ptr(), t["nb"],
                      float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                      float(group["clip"] or 0.0), step, torch.cuda.current_stream().cuda_stream)
            tens = torch.is_tensor(st0)
            for p in plist:
                st = self.state[p]
                st["step"] = st["step"].new_tensor(float(step)) if tens else step
            # The kernel wrote the parameters (and moments) through raw pointers: bump the version counters so every
            # cache keyed on them (EmipModule.packed: bf16 copies, conv / dgrad packs, folded norms) is rebuilt.
            torch.autograd.graph.increment_version(plist)
        # ... and rewrite the packs that are plain permutations of a master weight in place, with one launch
        # (nn_base.refresh_packs); everything else is rebuilt on first use because its version moved.
        from . import nn_base
        nn_base.refresh_packs()
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables.clear()                      # the moment tensors were replaced: their pointers are in the tables
