"""U-Net encoder of ClsWiseFormer on the HIP conv kernels (reference Unet_skipconnection.py:22-144).

Same module tree / parameter names as the reference (InitConv.conv, EnBlock*.conv1/conv2, EnDown*.conv,
EnDown_4.conv).  InstanceNorm + ReLU never run as separate passes: a conv's epilogue emits the statistics of its
output and the consumer applies normalise + ReLU while staging its input tile (cwf_conv_mfma prologue).
All activations are [N, D, H, W, C]."""
import torch
import torch.nn as nn

from cwf import functional as CF
from .layers import HipConv


class InitConv(nn.Module):
    """conv 4->16 followed by the reference's ALWAYS-ON channel dropout (F.dropout3d without a training flag,
    Unet_skipconnection.py:29-33).  ``dropout`` may be set to 0.0 for deterministic runs."""

    def __init__(self, in_channels=4, out_channels=16, dropout=0.2):
        super().__init__()
        self.conv = HipConv(in_channels, out_channels)
        self.dropout = dropout

    def forward(self, x, keep=None):
        if keep is None and self.dropout > 0.0:
            p = self.dropout
            keep = CF.dropout_mask((x.shape[0], self.conv.spec.cout), p, x.device)      # HIP kernel over the device generator state
        y, st = self.conv(x, out_scale=keep, want_stats=True)
        y._cwf_wgrad_first = True        # the conv that reads the stem's output runs the last data gradient of backward (cwf.functional)
        return y, st


class EnBlock(nn.Module):
    """IN -> ReLU -> conv -> IN -> ReLU -> conv -> + x   (Unet_skipconnection.py:36-57)"""

    def __init__(self, in_channels):
        super().__init__()
        self.conv1 = HipConv(in_channels, in_channels)
        self.conv2 = HipConv(in_channels, in_channels)

    def forward(self, x, x_stats, want_stats=True, skip_cb=0):
        # carry: the residual's gradient returns to conv1's backward and is added inside the kernel that writes dx
        h, hs, xc = self.conv1(x, in_norm=x_stats, slope=0.0, want_stats=True, carry=True)
        if skip_cb:
            # this block's output is a skip tensor: it is written straight into the lower channels of the buffer the decoder will
            # concatenate in (CF.skip_buffer), so that concatenation costs no copy
            n, d, hh, w, c = h.shape
            buf = CF.skip_buffer(n, d, hh, w, c, skip_cb, h.device)
            y, st = self.conv2(h, in_norm=hs, slope=0.0, residual=xc, want_stats=want_stats, out=CF.alias_channels(buf, 0, c))
            y._cwf_catbuf = buf
            return y, st
        return self.conv2(h, in_norm=hs, slope=0.0, residual=xc, want_stats=want_stats)


class EnDown(nn.Module):
    def __init__(self, in_channels, out_channels, stride=2):
        super().__init__()
        self.conv = HipConv(in_channels, out_channels, stride=stride)

    def forward(self, x, want_stats=True, carry=False):
        return self.conv(x, want_stats=want_stats, carry=carry)


class Unet(nn.Module):
    def __init__(self, in_channels=4, base_channels=16, num_classes=4):
        super().__init__()
        c = base_channels
        self.InitConv = InitConv(in_channels, c, dropout=0.2)
        self.EnBlock1 = EnBlock(c)
        self.EnBlock1_1 = EnBlock(c)
        self.EnDown1 = EnDown(c, 2 * c)
        self.EnBlock2_1 = EnBlock(2 * c)
        self.EnBlock2_2 = EnBlock(2 * c)
        self.EnDown2 = EnDown(2 * c, 4 * c)
        self.EnBlock3_1 = EnBlock(4 * c)
        self.EnBlock3_2 = EnBlock(4 * c)
        self.EnDown3 = EnDown(4 * c, 8 * c)
        self.EnBlock4_1 = EnBlock(8 * c)
        self.EnBlock4_2 = EnBlock(8 * c)
        self.EnDown_4 = EnDown(8 * c, 16 * c, stride=1)
        # channels the decoder concatenates behind the level-1 / level-2 skip tensors (DeUp_Cat: as many as the skip has; 0 = plain tensors)
        self.skip_cat = (c, 2 * c)

    def forward(self, x, stem_keep=None):
        x, s = self.InitConv(x, stem_keep)
        x, s = self.EnBlock1(x, s)
        x1, _ = self.EnBlock1_1(x, s, want_stats=False, skip_cb=self.skip_cat[0])
        # the skip connections leave through the carry alias of the down-sampling conv (their gradient is folded into its dgrad)
        x, s, x1 = self.EnDown1(x1, carry=True)
        x, s = self.EnBlock2_1(x, s)
        x2, _ = self.EnBlock2_2(x, s, want_stats=False, skip_cb=self.skip_cat[1])
        x, s, x2 = self.EnDown2(x2, carry=True)
        x, s = self.EnBlock3_1(x, s)
        x3, _ = self.EnBlock3_2(x, s, want_stats=False)
        x, s, x3 = self.EnDown3(x3, carry=True)
        x, s = self.EnBlock4_1(x, s)
        x, _ = self.EnBlock4_2(x, s, want_stats=False)
        x4, _ = self.EnDown_4(x, want_stats=False)
        return x1, x2, x3, x4
