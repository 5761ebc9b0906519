"""Sub-region and edge supervision heads (reference SuperviseLabel.py:6-81, EdgeSuperviseLabel.py:5-76):
conv3 -> conv3 -> trilinear upsample -> softmax over 2 channels, per sub-region k in {1,2,4}.
The 2-channel logits live in a 4-channel zero-padded buffer (16-byte voxel rows); upsample + softmax is one
kernel (cwf_upsample_softmax) writing the returned probability map once."""
import torch.nn as nn

from cwf import functional as CF
from .layers import HipConv


import torch


class _Heads(nn.Module):
    lazy_training_maps = True                     # training mode returns CF.LazyProb (head -> loss fusion); set False for real tensors

    def _head(self, first, second, x):
        h, _ = first(x)
        logit, _ = second(h)                      # [N,d,h,w,4], channels 2..3 zero
        if self.training and self.lazy_training_maps and torch.is_grad_enabled():
            return CF.LazyProb(logit, 2, self.sample_scale)      # the losses read the low-res logits; nothing is written at 128^3
        prob = CF.upsample_softmax(logit, 2, self.sample_scale)
        return prob.permute(0, 4, 1, 2, 3)        # logical [N,2,D,H,W], channels-last memory


    def _heads3(self, firsts, seconds, x_all):
        """The three sub-regions' heads on the channel groups of ONE tensor [N,d,h,w,3*C]: both conv stages as channel-grouped
        launches (CF.grouped_conv); in training mode the three lazy maps share the grouped logit buffer (one fused loss call)."""
        G = len(firsts)
        h_all = CF.grouped_conv(x_all, firsts)
        l_all = CF.grouped_conv(h_all, seconds)           # [N,d,h,w,G*4]: 2 logits + 2 pad channels per region
        ca = l_all.shape[-1] // G
        out = {}
        lazy = self.training and self.lazy_training_maps and torch.is_grad_enabled()
        for q, r in enumerate(("01", "02", "04")):
            logit = l_all[..., q * ca:(q + 1) * ca]
            if lazy:
                out[r] = CF.LazyProb(logit, 2, self.sample_scale, parent=(l_all, q, G, ca))
            else:
                out[r] = CF.upsample_softmax(logit.contiguous(), 2, self.sample_scale).permute(0, 4, 1, 2, 3)
        return out


class SuperviseLabel(_Heads):
    def head(self, k, x):
        """One sub-region's head (k in {1,2,4})."""
        return self._head(getattr(self, "supervise_label_%d" % k), getattr(self, "down_label_%d" % k), x)

    def __init__(self, item_future_num):
        super().__init__()
        for k in (1, 2, 4):
            setattr(self, "supervise_label_%d" % k, HipConv(item_future_num, 32))
            setattr(self, "down_label_%d" % k, HipConv(32, 2))
        self.sample_scale = 8

    def heads3(self, x_all):
        return self._heads3([getattr(self, "supervise_label_%d" % k) for k in (1, 2, 4)], [getattr(self, "down_label_%d" % k) for k in (1, 2, 4)], x_all)

    def forward(self, s01, s02, s04):
        return {"01": self._head(self.supervise_label_1, self.down_label_1, s01),
                "02": self._head(self.supervise_label_2, self.down_label_2, s02),
                "04": self._head(self.supervise_label_4, self.down_label_4, s04)}


class EdgeSuperviseLabel(_Heads):
    def head(self, k, x):
        return self._head(getattr(self, "edge_supervise_label_%d" % k), getattr(self, "edge_down_label_%d" % k), x)

    def __init__(self, item_future_num):
        super().__init__()
        for k in (1, 2, 4):
            setattr(self, "edge_supervise_label_%d" % k, HipConv(item_future_num, 8))
            setattr(self, "edge_down_label_%d" % k, HipConv(8, 2))
        self.sample_scale = 4

    def heads3(self, x_all):
        return self._heads3([getattr(self, "edge_supervise_label_%d" % k) for k in (1, 2, 4)], [getattr(self, "edge_down_label_%d" % k) for k in (1, 2, 4)], x_all)

    def forward(self, e01, e02, e04):
        return {"01": self._head(self.edge_supervise_label_1, self.edge_down_label_1, e01),
                "02": self._head(self.edge_supervise_label_2, self.edge_down_label_2, e02),
                "04": self._head(self.edge_supervise_label_4, self.edge_down_label_4, e04)}
