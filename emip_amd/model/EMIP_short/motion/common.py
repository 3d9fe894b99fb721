"""LayerNorm2d parameter holder (weight/bias over channels of an NCHW tensor, eps 1e-6).  In EMIP it only
appears inside modules that forward never calls (/root/reference/model/EMIP_short/model.py:66-84), so it
exists for state_dict compatibility."""
import torch
import torch.nn as nn


class LayerNorm2d(nn.Module):
    def __init__(self, num_channels: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps
