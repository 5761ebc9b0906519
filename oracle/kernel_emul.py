"""ORACLE (test infrastructure, never shipped in the product path).

Per-kernel CPU oracle: a plain-PyTorch implementation of the tensor-level backend interface of
``cwf/kernels.py`` (HipBackend).  Each method states in torch ops what the corresponding HIP kernel must compute,
i.e. the ATen sequence the reference dispatches for that piece (file:line in the docstrings).  Two uses, both in
``tests/`` only:
  * `-m gpu` tests run a HIP kernel and this emulation on the same inputs and compare (kernel-level parity);
  * `-m "not gpu"` tests inject it via ``cwf.kernels._set_backend_for_testing`` to run the package's autograd glue and
    module tree on CPU and compare the whole model / losses / gradients against ``oracle/reference_model.py`` and the
    golden fixtures (host-logic parity).
The emulation takes the ORIGINAL weights (``w_ref``) -- not the packed buffers -- so the packing maps themselves are
verified only where it matters: HIP kernel (packed) vs emulation (reference layout) on the GPU.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

CONV3_S1, CONV3_S2, CONV1, CONVT2, CONV3_S2_DGRAD, CONVT2_DGRAD = range(6)


def _ncdhw(t):
    return t.permute(0, 4, 1, 2, 3)


def _ndhwc(t):
    return t.permute(0, 2, 3, 4, 1).contiguous()


def _act(v, slope):
    return torch.where(v > 0, v, v * slope)


def _prologue(x, in_scale, in_shift, slope):
    """act(x*scale+shift) per (n,c): the conv kernels' fused InstanceNorm + (Leaky)ReLU staging prologue."""
    if in_scale is not None:
        x = x * in_scale[:, None, None, None, :] + in_shift[:, None, None, None, :]
    if in_scale is not None or slope != 1.0:
        x = _act(x, slope)
    return x


def _fwd_conv(op, xa, w, b):
    x = _ncdhw(xa)
    if op == CONV3_S1:
        y = F.conv3d(x, w, b, stride=1, padding=1)
    elif op == CONV3_S2:
        y = F.conv3d(x, w, b, stride=2, padding=1)
    elif op == CONV1:
        y = F.conv3d(x, w, b)
    elif op == CONVT2:
        y = F.conv_transpose3d(x, w, b, stride=2)
    else:
        raise ValueError(op)
    return _ndhwc(y)


class EmulBackend:
    name = "emul"

    # ------------------------------------------------------------------ K1
    def conv(self, op, x, wpk, bias, cout, in_scale=None, in_shift=None, slope=1.0, residual=None, out_scale=None,
             stats=None, out=None, w_ref=None, out_channels_alloc=None, fwd_op=None, prec=None):
        """cwf_conv_mfma: F.conv3d / F.conv_transpose3d on act(IN(x)) (+bias, +residual, *out_scale) or, for the
        data-gradient forms (fwd_op given), the adjoint of the forward conv w.r.t. its (activated) input."""
        assert w_ref is not None
        if fwd_op is None:
            xa = _prologue(x[..., :w_ref.shape[1] if op != CONVT2 else w_ref.shape[0]], in_scale, in_shift, slope)
            y = _fwd_conv(op, xa, w_ref, bias)
            if residual is not None:
                y = y + residual
            if out_scale is not None:
                y = y * out_scale[:, None, None, None, :]
            if stats is not None:
                stats[:, :, 0] += y.double().sum((1, 2, 3))
                stats[:, :, 1] += (y.double() ** 2).sum((1, 2, 3))
            if out is None:
                ca = out_channels_alloc or cout
                out = torch.zeros(y.shape[:-1] + (ca,), dtype=torch.float32)
            out[..., :cout] = y
            return out
        # data gradient: x is dy (possibly zero-padded channels), out has the forward input's shape
        dy = x[..., :(w_ref.shape[0] if fwd_op != CONVT2 else w_ref.shape[1])]
        xin = torch.zeros(out.shape, dtype=torch.float32, requires_grad=True)
        with torch.enable_grad():
            y = _fwd_conv(fwd_op, xin, w_ref, None)
            (g,) = torch.autograd.grad(y, xin, dy.contiguous())
        out.copy_(g)
        return out

    def wgrad(self, op, x, in_scale, in_shift, slope, dy, cout, inv_map, has_bias_map, w_numel, w_ref_shape=None, prec=None, allow_async=False):
        """cwf_wgrad_mfma + cwf_wgrad_reduce: weight / bias halves of aten::convolution_backward on act(IN(x))."""
        xa = _prologue(x, in_scale, in_shift, slope).detach()
        w = torch.zeros(w_ref_shape, dtype=torch.float32, requires_grad=True)
        b = torch.zeros(cout, dtype=torch.float32, requires_grad=True)
        with torch.enable_grad():
            y = _fwd_conv(op, xa, w, b)
            gw, gb = torch.autograd.grad(y, (w, b), dy.contiguous())
        return gw.reshape(-1), (gb if has_bias_map else None)

    def begin_step(self, device):
        pass

    def gather_batched(self, table, nlayers, max_n, split_bf16=False):
        pass    # packed buffers are unused by the emulation

    # ------------------------------------------------------------------ K3
    def new_stats(self, n, c, device):
        return torch.zeros((n, c, 2), dtype=torch.float64)

    def in_finalize(self, stats, nvox, eps=1e-5):
        """mean / biased variance -> scale = rstd, shift = -mean*rstd (nn.InstanceNorm3d, eps 1e-5)."""
        mean = stats[:, :, 0] / nvox
        var = (stats[:, :, 1] / nvox - mean * mean).clamp_min(0)
        rstd = 1.0 / torch.sqrt(var + eps)
        return rstd.float(), (-mean * rstd).float()

    def in_stats(self, x):
        s = torch.zeros((x.shape[0], x.shape[4], 2), dtype=torch.float64)
        s[:, :, 0] = x.double().sum((1, 2, 3))
        s[:, :, 1] = (x.double() ** 2).sum((1, 2, 3))
        return s

    def norm_act_add(self, x, scale, shift, slope, residual=None):
        y = _act(x * scale[:, None, None, None, :] + shift[:, None, None, None, :], slope)
        return y + residual if residual is not None else y

    def in_bwd(self, dy, x, scale, shift, slope, dx_add=None):
        """Backward of y = act(instance_norm(x)) w.r.t. x, statistics included."""
        sc, sh = scale[:, None, None, None, :], shift[:, None, None, None, :]
        xh = x * sc + sh
        g = dy * torch.where(xh > 0, torch.ones_like(xh), torch.full_like(xh, slope))
        m1 = g.double().mean((1, 2, 3), keepdim=True).float()
        m2 = (g.double() * xh.double()).mean((1, 2, 3), keepdim=True).float()
        dx = sc * (g - m1 - xh * m2)
        return dx + dx_add if dx_add is not None else dx

    # ------------------------------------------------------------------ K6/K7
    def gemm(self, a, sa, b, sb, c, sc, m, n, k, zb=1, zh=1, bias=None, residual=None, sr=(0, 0, 0), alpha=1.0, act=0,
             accumulate=False, a_off=0, b_off=0, c_off=0, r_off=0):
        def view(t, off, sizes, strides):
            return torch.as_strided(t, sizes, strides, t.storage_offset() + off)
        A = view(a, a_off, (zb, zh, m, k), (sa[2], sa[3], sa[0], sa[1]))
        B = view(b, b_off, (zb, zh, k, n), (sb[2], sb[3], sb[0], sb[1]))
        C = view(c, c_off, (zb, zh, m, n), (sc[1], sc[2], sc[0], 1))
        v = torch.matmul(A, B) * alpha
        if bias is not None:
            v = v + bias
        if act == 1:
            v = F.gelu(v)
        if residual is not None:
            v = v + view(residual, r_off, (zb, zh, m, n), (sr[1], sr[2], sr[0], 1))
        if accumulate:
            v = v + C
        C.copy_(v)
        return c

    def layernorm_fwd(self, x, gamma, beta, eps=1e-5):
        mean = x.mean(-1)
        var = x.var(-1, unbiased=False)
        rstd = 1.0 / torch.sqrt(var + eps)
        y = (x - mean[..., None]) * rstd[..., None] * gamma + beta
        return y, mean.reshape(-1), rstd.reshape(-1)

    def layernorm_bwd(self, dy, x, gamma, mean, rstd, dgamma, dbeta):
        e = x.shape[-1]
        xh = (x.reshape(-1, e) - mean[:, None]) * rstd[:, None]
        d = dy.reshape(-1, e)
        g = d * gamma
        dx = rstd[:, None] * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))
        dgamma.copy_((d * xh).sum(0))        # written, not accumulated (matches cwf_layernorm_bwd)
        dbeta.copy_(d.sum(0))
        return dx.reshape(x.shape)

    def softmax_rows_(self, s):
        s.copy_(torch.softmax(s, -1))
        return s

    def softmax_rows_bwd_(self, p, dp):
        dp.copy_(p * (dp - (dp * p).sum(-1, keepdim=True)))
        return dp

    def gelu_bwd(self, x, dy):
        cdf = 0.5 * (1 + torch.erf(x / math.sqrt(2)))
        pdf = torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
        return dy * (cdf + x * pdf)

    def colsum(self, x2d):
        return x2d.sum(0)

    # ------------------------------------------------------------------ K4/K5
    def window_to_tokens(self, x, patch):
        """convert_dim (cls_wise_former.py:15-23) on NDHWC input."""
        b, d, h, w, c = x.shape
        p0, p1, p2 = patch
        t = _ncdhw(x).reshape(b, c, d // p0, p0, h // p1, p1, w // p2, p2).permute(0, 2, 4, 6, 1, 3, 5, 7)
        return t.reshape(b, (d // p0) * (h // p1) * (w // p2), c * p0 * p1 * p2).contiguous()

    def tokens_to_window(self, tok, size, channels, patch):
        """split_dim (cls_wise_former.py:26-39) producing NDHWC."""
        b = tok.shape[0]
        d, h, w = size
        p0, p1, p2 = patch
        t = tok.reshape(b, d // p0, h // p1, w // p2, channels, p0, p1, p2).permute(0, 4, 1, 5, 2, 6, 3, 7)
        return _ndhwc(t.reshape(b, channels, d, h, w))

    def token_scores(self, feats, query):
        return torch.einsum("bte,be->bt", feats, query.expand(feats.shape[0], -1, -1)[:, 0])

    def topk(self, score, k):
        # stable descending sort = ties broken by lower index first (the HIP comparator)
        return torch.sort(score, dim=1, descending=True, stable=True).indices[:, :k].to(torch.int32)

    def gather_tokens(self, feats, index, head, keep=None, pe_odd=1.0):
        b, t, e = feats.shape
        rows = torch.gather(feats, 1, index.long()[:, :, None].expand(-1, -1, e)).clone()
        rows[..., 1::2] += pe_odd
        if keep is not None:
            rows = rows * keep
        return torch.cat((head.expand(b, -1, -1), rows), dim=1)

    def gather_tokens_bwd(self, dseq, index, keep, dfeats, dhead):
        if dhead is not None:
            dhead += dseq[:, 0:1].sum(0, keepdim=True) if dhead.shape[0] == 1 else dseq[:, 0:1]
        if dfeats is not None:
            d = dseq[:, 1:] * keep if keep is not None else dseq[:, 1:]
            dfeats.scatter_add_(1, index.long()[:, :, None].expand(-1, -1, d.shape[2]), d)

    def scatter_rows(self, feats, index, rows, gate=None):
        out = feats.scatter(1, index.long()[:, :, None].expand(-1, -1, feats.shape[2]), rows)
        return out * gate if gate is not None else out

    def scatter_rows_bwd(self, dout, index, scat, gate, k, need_feats=True, need_rows=True):
        idx = index.long()[:, :, None].expand(-1, -1, dout.shape[2])
        dgate = (dout * scat).sum(1, keepdim=True) if gate is not None else None
        g = dout * gate if gate is not None else dout
        drows = torch.gather(g, 1, idx)
        dfeats = g.scatter(1, idx, torch.zeros_like(drows))
        return dfeats, drows, dgate

    # ------------------------------------------------------------------ K8/K10
    def upsample_softmax(self, logit, c, scale):
        up = F.interpolate(_ncdhw(logit[..., :c]), scale_factor=scale, mode="trilinear", align_corners=False)
        return _ndhwc(up.softmax(dim=1))

    def upsample_softmax_bwd(self, dprob, prob, lo_shape, c, scale, ldc_out):
        n, d, h, w = lo_shape
        dl_hi = prob * (dprob - (dprob * prob).sum(-1, keepdim=True))
        lo = torch.zeros((n, c, d, h, w), requires_grad=True)
        with torch.enable_grad():
            up = F.interpolate(lo, scale_factor=scale, mode="trilinear", align_corners=False)
            (g,) = torch.autograd.grad(up, lo, _ncdhw(dl_hi).contiguous())
        out = torch.zeros((n, d, h, w, ldc_out))
        out[..., :c] = _ndhwc(g)
        return out

    def channel_softmax(self, logit, c):
        return logit[..., :c].softmax(-1).contiguous()

    def channel_softmax_bwd(self, dprob, prob):
        return prob * (dprob - (dprob * prob).sum(-1, keepdim=True))

    # ------------------------------------------------------------------ K9
    @staticmethod
    def _onehot(label, posmask, c):
        cls = label if c == 4 else ((posmask >> label) & 1)
        return F.one_hot(cls.long(), c).float()                 # [N,D,H,W,C]

    def dice_ce(self, prob, label, posmask):
        """tools.dice_loss + tools.softmax_weighted_loss (tools.py:8-34) -> (loss [1], coef [N,C,4])."""
        n, d, h, w, c = prob.shape
        v = d * h * w
        t = self._onehot(label, posmask, c)
        p64, t64 = prob.double(), t.double()
        I = (p64 * t64).sum((0, 1, 2, 3)); P = p64.sum((0, 1, 2, 3)); T = t64.sum((0, 1, 2, 3))
        den = P + T + 1e-7
        dice = 1.0 - (2.0 * I / den).sum() / c
        Tn = t64.sum((1, 2, 3))                                  # [N,C]
        wgt = 1.0 - Tn / Tn.sum(1, keepdim=True)
        S = (t64 * torch.log(torch.clamp(prob, 0.005, 1.0)).double()).sum((1, 2, 3))
        ce = (-(wgt * S).sum()) / (n * v)
        coef = torch.zeros((n, c, 4))
        coef[:, :, 0] = (-(2.0 / c) / den).float()
        coef[:, :, 1] = ((2.0 / c) * I / (den * den)).float()
        coef[:, :, 2] = (-wgt / (n * v)).float()
        return (dice + ce).float().reshape(1), coef

    def dice_ce_bwd(self, prob, label, posmask, coef, gscale):
        c = prob.shape[-1]
        t = self._onehot(label, posmask, c)
        k = coef[:, None, None, None, :, :]
        inside = ((prob >= 0.005) & (prob <= 1.0)).float()
        g = k[..., 1] + t * (k[..., 0] + inside * k[..., 2] / prob)
        return gscale[0] * g

    # ------------------------------------------------------------------ K11 / misc
    def adam(self, table, ntensors, max_n, lr, beta1, beta2, eps, wd, step, amsgrad, hyper_dev=None):
        """Pointer-table form of torch.optim.Adam(amsgrad, weight_decay) -- optim.hip / train_no_amp.py:136,239.  Host
        memory only (CPU tests): rows = [param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, n] raw addresses."""
        import ctypes
        import numpy as np

        def view(ptr, n):
            return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_float * n).from_address(int(ptr))))

        if hyper_dev is not None:
            step_size, sqrt_bc2 = float(hyper_dev[0]), float(hyper_dev[1])
        else:
            step_size, sqrt_bc2 = lr / (1.0 - beta1 ** step), math.sqrt(1.0 - beta2 ** step)
        for pp, gp, mp, vp, xp, n in table.tolist()[:ntensors]:
            p_, g_, m_, v_ = view(pp, n), view(gp, n), view(mp, n), view(vp, n)
            g = g_ + wd * p_
            m_.mul_(beta1).add_(g, alpha=1.0 - beta1)
            v_.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
            if amsgrad:
                x_ = view(xp, n)
                torch.maximum(x_, v_, out=x_)
                denom = x_.sqrt() / sqrt_bc2 + eps
            else:
                denom = v_.sqrt() / sqrt_bc2 + eps
            p_.addcdiv_(m_, denom, value=-step_size)

    def dropout_mask(self, shape, p, device, p2=0.0):
        m = (torch.rand(shape, device=device) >= p).float() * (1.0 / (1.0 - p))
        if p2 > 0.0:
            m = m * ((torch.rand(shape, device=device) >= p2).float() * (1.0 / (1.0 - p2)))
        return m

    def mul(self, a, b):
        return a * b

    def add(self, a, b):
        return a + b

    def channel_scale(self, x, s):
        return x * s[:, None, None, None, :]

    def copy_into(self, x, out):
        out.copy_(x)
        return out
