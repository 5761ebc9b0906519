"""Leaf layers shared by the ClsWiseFormer modules: parameter containers with the reference's names/shapes whose
forward is a HIP launch sequence (cwf.functional) instead of an ATen call."""
from __future__ import annotations

import torch
import torch.nn as nn

from cwf import functional as CF
from cwf import packing as pk


class HipConv(nn.Module):
    """Holds weight/bias exactly like nn.Conv3d / nn.ConvTranspose3d (same shapes, same default init) and a
    ConvSpec describing how the MFMA kernels consume them."""

    def __init__(self, cin, cout, k=3, stride=1, transposed=False):
        super().__init__()
        ref = nn.ConvTranspose3d(cin, cout, kernel_size=2, stride=2) if transposed else \
            nn.Conv3d(cin, cout, kernel_size=k, stride=stride, padding=k // 2)
        self.weight, self.bias = ref.weight, ref.bias
        if transposed:
            op = pk.CONVT2
        elif k == 1:
            op = pk.CONV1
        else:
            op = pk.CONV3_S2 if stride == 2 else pk.CONV3_S1
        self.spec = CF.ConvSpec(op, cin, cout)

    def forward(self, x, in_norm=None, slope=1.0, residual=None, out_scale=None, want_stats=False, carry=False, out=None, emit16=False):
        return CF.conv(x, self.weight, self.bias, self.spec, in_norm, slope, residual, out_scale, want_stats, carry, out, emit16)


def collect_convs(module, packer, skip=()):
    """register every conv weight for the once-per-step packing launch (`skip`: layers that run as part of a fused layer)"""
    skip = {id(m) for m in skip}
    for m in module.modules():
        if isinstance(m, HipConv) and id(m) not in skip:
            packer.add(m.spec, m.weight)
