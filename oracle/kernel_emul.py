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


_M64 = (1 << 64) - 1


def _u01(seed, step, ctr):
    """cwf_rng_u01 (csrc/common.h): splitmix64 of (seed + step * K, counter) -> float32 in [0, 1).  ctr: np.uint64 array."""
    import numpy as np
    key = np.uint64((seed + step * 0xD6E8FEB86659FD93) & _M64)
    with np.errstate(over="ignore"):
        z = (key ^ np.uint64(0x9E3779B97F4A7C15)) + ctr * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)


class EmulBackend:
    name = "emul"

    def __init__(self):
        self._seed = torch.initial_seed() & 0x7FFFFFFFFFFFFFFF
        self._step = 0
        self._site = 0

    # ------------------------------------------------------------------ K12: counter-based dropout (bit-exact with cwf_keep)
    def set_rng(self, seed, step):
        self._seed, self._step = int(seed), int(step)

    def rng(self, device):
        return torch.tensor([self._seed, self._step], dtype=torch.int64)

    def rng_site(self, n):
        off = self._site
        self._site = off + 2 * int(n)
        return off

    def keep(self, off, n, p, p2=0.0):
        """flat float32 [n]: keep(i) = [u01(off + i) >= p] / (1 - p)  (* the same for p2 at counter off + n + i)"""
        import numpy as np
        i = np.arange(n, dtype=np.uint64)
        one = np.float32(1.0)
        v = np.where(_u01(self._seed, self._step, np.uint64(off) + i) >= np.float32(p), one / (one - np.float32(p)), np.float32(0.0))
        if p2 > 0.0:
            v = v * np.where(_u01(self._seed, self._step, np.uint64(off + n) + i) >= np.float32(p2), one / (one - np.float32(p2)), np.float32(0.0))
        return torch.from_numpy(v.astype(np.float32))

    # ------------------------------------------------------------------ K1
    def conv(self, op, x, wpk, bias, cout, in_scale=None, in_shift=None, slope=1.0, residual=None, out_scale=None,
             stats=None, out=None, w_ref=None, out_channels_alloc=None, fwd_op=None, prec=None, bias_ref=None):
        """cwf_conv_mfma: F.conv3d / F.conv_transpose3d on act(IN(x)) (+bias, +residual, *out_scale) or, for the
        data-gradient forms (fwd_op given), the adjoint of the forward conv w.r.t. its (activated) input."""
        assert w_ref is not None
        if isinstance(w_ref, (tuple, list)):               # a fused layer (FusedConvSpec): the sources' parameters, concatenated
            w_ref = torch.cat(tuple(w_ref), 0)
        if bias_ref is not None:
            bias = torch.cat(tuple(bias_ref), 0)
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
        out.copy_(g + residual if residual is not None else g)
        return out


    def conv_grouped(self, x_all, cin, wpks, biases, cout, y_all, x_goff, y_goff, w_refs=None, fwd_op=None, prec=None):
        """cwf_conv_mfma_bf16_grouped: G channel-grouped 3x3x3 stride-1 convs (or their data gradients) -- one conv per group"""
        assert w_refs is not None
        for q, w in enumerate(w_refs):
            xs = x_all[..., q * x_goff:q * x_goff + cin]
            if fwd_op is None:
                y = self.conv(CONV3_S1, xs, None, None if biases is None else biases[q], cout, w_ref=w)
                y_all[..., q * y_goff:q * y_goff + cout] = y[..., :cout]
            else:
                out = torch.empty(y_all.shape[:-1] + (cout,), dtype=torch.float32)
                self.conv(CONV3_S1, xs, None, None, cout, out=out, w_ref=w, fwd_op=fwd_op)
                y_all[..., q * y_goff:q * y_goff + cout] = out
        return y_all
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
        self._step += 1
        self._site = 0

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

    def window_to_tokens_g(self, x, groups, patch):
        c = x.shape[-1] // groups
        return torch.stack([self.window_to_tokens(x[..., g * c:(g + 1) * c], patch) for g in range(groups)], 0)

    def tokens_to_window_g(self, tok, size, channels, patch):
        return torch.cat([self.tokens_to_window(tok[g], size, channels, patch) for g in range(tok.shape[0])], -1)

    def cat3_channels(self, parts, shape, device):
        return torch.cat([torch.zeros(shape) if t is None else t for t in parts], -1)

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

    # ------------------------------------------------------------------ K6/K7 round-2 fused forms
    def _mask2d(self, drop, m, n):
        if not drop:
            return None
        off, p, p2 = drop
        return self.keep(off, m * n, p, p2).reshape(m, n)

    def linear_fwd(self, x, w, bias, out, x2=None, split_n=0, act=0, pre=None, drop=None, residual=None):
        """cwf_gemm_ex as nn.Linear (+ GELU, + Dropout, + residual): SelfAttention.py:80-102, ResidualNorm.py:35-47.
        w / bias may be lists of G weight sets: the rows are then G stacked problems."""
        if isinstance(w, (list, tuple)):
            G, mg = len(w), x.shape[0] // len(w)
            parts = []
            for g in range(G):
                r = slice(g * mg, (g + 1) * mg)
                vg = x[r] @ w[g].t()
                if x2 is not None:
                    vg[:, split_n:] = x2[r] @ w[g][split_n:].t()
                parts.append(vg + bias[g] if bias is not None else vg)
            v = torch.cat(parts, 0)
        else:
            v = x @ w.t()
            if x2 is not None:
                v[:, split_n:] = x2 @ w[split_n:].t()
            if bias is not None:
                v = v + bias
        if pre is not None:
            pre.copy_(v)
        if act == 1:
            v = F.gelu(v)
        mk = self._mask2d(drop, *v.shape)
        if mk is not None:
            v = v * mk
        if residual is not None:
            v = v + residual
        out.copy_(v)
        return out

    def linear_dgrad(self, dy, w, drop=None, out=None):
        mk = self._mask2d(drop, *dy.shape)
        d = dy * mk if mk is not None else dy
        if isinstance(w, (list, tuple)):
            mg = dy.shape[0] // len(w)
            dx = torch.cat([d[g * mg:(g + 1) * mg] @ w[g] for g in range(len(w))], 0)
        else:
            dx = d @ w
        if out is not None:
            out.copy_(dx)
            return out
        return dx

    def linear_wgrad(self, dy, x, dw, dbias=None, x2=None, split_m=0, accumulate=False, drop=None):
        mk = self._mask2d(drop, *dy.shape)
        d = dy * mk if mk is not None else dy
        if isinstance(dw, (list, tuple)):
            mg = dy.shape[0] // len(dw)
            for gi in range(len(dw)):
                r = slice(gi * mg, (gi + 1) * mg)
                self.linear_wgrad(d[r], x[r], dw[gi], dbias[gi] if dbias is not None else None, x2[r] if x2 is not None else None, split_m, accumulate)
            return
        g = d.t() @ x
        if x2 is not None:
            g[split_m:] = d[:, split_m:].t() @ x2
        dw.copy_(g + dw if accumulate else g)
        if dbias is not None:
            dbias.copy_(d.sum(0) + dbias if accumulate else d.sum(0))

    @staticmethod
    def _perm(rows, perm_T, device=None):
        r = torch.arange(rows)
        return ((r // perm_T) ^ 1) * perm_T + r % perm_T if perm_T > 0 else r

    def ln_pair_fwd(self, x, x2, perm_T, g1, b1, g2, b2, eps=1e-5):
        """nn.LayerNorm(x), nn.LayerNorm(x2[perm]) (PreNormDrop, ResidualNorm.py:23-32) + (mean, rstd) per row."""
        rows = x.shape[0]
        if isinstance(g1, (list, tuple)):
            G, rg = len(g1), rows // len(g1)
            outs = [self.ln_pair_fwd(x[i * rg:(i + 1) * rg], None if x2 is None else x2[i * rg:(i + 1) * rg], perm_T, g1[i], b1[i],
                                     None if g2 is None else g2[i], None if b2 is None else b2[i], eps) for i in range(G)]
            return (torch.cat([o[0] for o in outs], 0), None if x2 is None else torch.cat([o[1] for o in outs], 0),
                    torch.cat([o[2] for o in outs], 1))

        def one(v, g, b):
            mean = v.mean(-1)
            rstd = 1.0 / torch.sqrt(v.var(-1, unbiased=False) + eps)
            return (v - mean[:, None]) * rstd[:, None] * g + b, torch.stack((mean, rstd), -1)
        ya, sa = one(x, g1, b1)
        if x2 is None:
            return ya, None, sa[None]
        yb, sb = one(x2[self._perm(rows, perm_T)], g2, b2)
        return ya, yb, torch.stack((sa, sb), 0)

    def ln_pair_bwd(self, dy, da, db, x, x2, perm_T, g1, g2, stats, dg1, db1, dg2, db2, accumulate, want_dx2):
        rows, e = x.shape
        if isinstance(g1, (list, tuple)):
            G, rg = len(g1), rows // len(g1)
            sl = lambda t, i: None if t is None else t[i * rg:(i + 1) * rg]
            pick = lambda t, i: None if t is None else t[i]
            outs = [self.ln_pair_bwd(sl(dy, i), sl(da, i), sl(db, i), sl(x, i), sl(x2, i), perm_T, g1[i], pick(g2, i), stats[:, i * rg:(i + 1) * rg],
                                     dg1[i], db1[i], pick(dg2, i), pick(db2, i), accumulate, want_dx2) for i in range(G)]
            return torch.cat([o[0] for o in outs], 0), (torch.cat([o[1] for o in outs], 0) if want_dx2 else None)

        def lnb(d, v, g, st):
            xh = (v - st[:, 0:1]) * st[:, 1:2]
            gg = d * g
            return st[:, 1:2] * (gg - gg.mean(-1, keepdim=True) - xh * (gg * xh).mean(-1, keepdim=True)), (d * xh).sum(0), d.sum(0)
        dx1, pg, pb = lnb(da, x, g1, stats[0])
        dg1.copy_(pg + dg1 if accumulate else pg); db1.copy_(pb + db1 if accumulate else pb)
        dx = dx1 + dy if dy is not None else dx1
        dx2 = None
        if db is not None:
            perm = self._perm(rows, perm_T)
            d2, pg2, pb2 = lnb(db, x2[perm], g2, stats[1])       # gradient w.r.t. the permuted rows the second LayerNorm saw
            dg2.copy_(pg2 + dg2 if accumulate else pg2); db2.copy_(pb2 + db2 if accumulate else pb2)
            if want_dx2:
                dx2 = d2
            else:
                dx = dx.index_add(0, perm, d2)                   # x2 is x: route back through the (involutive) permutation
        return dx, dx2

    def _attn(self, qkv, z, t, heads, drop):
        e = qkv.shape[1] // 3
        hd = e // heads
        q, k, v = (qkv[:, i * e:(i + 1) * e].reshape(z, t, heads, hd).permute(0, 2, 1, 3) for i in range(3))
        att = (torch.einsum("zhxd,zhyd->zhxy", q, k) * (hd ** -0.5)).softmax(-1)
        if drop:
            att = att * self.keep(drop[0], z * heads * t * t, drop[1]).reshape(z, heads, t, t)
        return torch.einsum("zhxy,zhyd->zhxd", att, v).permute(0, 2, 1, 3).reshape(z * t, e)

    def attn_fwd(self, qkv, z, t, heads, drop=None):
        """softmax(q k^T / sqrt(d)) -> attn_drop -> @ v  (SelfAttention.py:94-98)"""
        return self._attn(qkv, z, t, heads, drop)

    def attn_bwd(self, qkv, d_o, z, t, heads, drop=None):
        qq = qkv.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            o = self._attn(qq, z, t, heads, drop)
            (g,) = torch.autograd.grad(o, qq, d_o)
        return g

    def gelu_bwd_drop(self, z, dh, drop=None):
        g = self.gelu_bwd(z, dh)
        if drop:
            g = g * self.keep(drop[0], z.numel(), drop[1]).reshape(z.shape)
        return g

    # ------------------------------------------------------------------ K4/K5 round-2 forms
    def token_scores2(self, feats, q1, q2=None):
        if isinstance(q1, (list, tuple)):                   # G shared queries, sample b in group b // (B / G)
            bg = feats.shape[0] // len(q1)
            q1 = torch.cat([q.expand(bg, -1, -1) for q in q1], 0)
            q2 = torch.cat([q.expand(bg, -1, -1) for q in q2], 0) if q2 is not None else None
        return self.token_scores(feats, q1), (self.token_scores(feats, q2) if q2 is not None else None)

    def _inv(self, index, t):
        b, k = index.shape
        inv = torch.full((b, t), -1, dtype=torch.int32)
        inv.scatter_(1, index.long(), torch.arange(k, dtype=torch.int32)[None].expand(b, -1))
        return inv

    def topk_inv(self, s0, s1, k):
        def one(s):
            s = torch.where(torch.isnan(s), torch.full_like(s, float("inf")), s)
            idx = self.topk(s, k)
            return idx, self._inv(idx, s.shape[1])
        i0, v0 = one(s0)
        if s1 is None:
            return i0, v0, None, None
        i1, v1 = one(s1)
        return i0, v0, i1, v1

    def index_inv(self, index, t):
        index = index.to(torch.int32).contiguous()
        return index, self._inv(index, t)

    def gather_multi(self, jobs, k, e, p=0.0, pe_odd=1.0):
        for feats, index, head, out, off in jobs:
            b = feats.shape[0]
            if isinstance(head, (list, tuple)):
                head = torch.cat([h.expand(b // len(head), -1, -1) for h in head], 0)
            keep = self.keep(off, b * k * e, p).reshape(b, k, e) if p > 0.0 else None
            out.copy_(self.gather_tokens(feats, index, head, keep, pe_odd))

    def scatter_inv(self, feats, inv, rows, gate, want_gated=True, want_scat=False):
        sel = (inv >= 0)[:, :, None]
        picked = torch.gather(rows, 1, inv.clamp_min(0).long()[:, :, None].expand(-1, -1, feats.shape[2]))
        scat = torch.where(sel, picked, feats)
        return (scat * gate if want_gated else None), (scat if want_scat else None)

    def scatter_bwd(self, dgated, dscat, feats, inv, index, rows, gate, dgate_extra, drows, dgate):
        b, t, e = feats.shape
        idx = index.long()[:, :, None].expand(-1, -1, e)
        g = torch.zeros((b, t, e))
        dg = torch.zeros((b, 1, e))
        if dgated is not None:
            _, scat = self.scatter_inv(feats, inv, rows, None, want_gated=False, want_scat=True)
            g = g + dgated * gate
            dg = dg + (dgated * scat).sum(1, keepdim=True)
        if dscat is not None:
            g = g + dscat
        if dgate_extra is not None:
            dg = dg + dgate_extra
        drows.copy_(torch.gather(g, 1, idx))
        dgate.copy_(dg)

    def token_grad(self, dgated, dscat, gate, inv_p, inv_q, dseq_p, dseq_q, k, p=0.0, off_p=0, off_q=0):
        b, t = inv_p.shape
        e = dseq_p.shape[2]
        g = torch.zeros((b, t, e))
        if dgated is not None:
            g = g + dgated * gate
        if dscat is not None:
            g = g + dscat
        out = torch.where((inv_p < 0)[:, :, None], g, torch.zeros_like(g))
        for inv, dseq, off in ((inv_p, dseq_p, off_p), (inv_q, dseq_q, off_q)):
            if inv is None:
                continue
            d = dseq[:, 1:]
            if p > 0.0:
                d = d * self.keep(off, b * k * e, p).reshape(b, k, e)
            picked = torch.gather(d, 1, inv.clamp_min(0).long()[:, :, None].expand(-1, -1, e))
            out = out + torch.where((inv >= 0)[:, :, None], picked, torch.zeros_like(picked))
        return out

    def head_grad(self, a1, c1, a2, c2, out1=None, out2=None):
        if isinstance(out1, (list, tuple)):
            bg = a1.shape[0] // len(out1)
            for g in range(len(out1)):
                r = slice(g * bg, (g + 1) * bg)
                out1[g].copy_((a1[r] + c1[r]).sum(0).reshape(out1[g].shape)); out2[g].copy_((a2[r] + c2[r]).sum(0).reshape(out2[g].shape))
            return out1, out2
        o1, o2 = (a1 + c1).sum(0).reshape(1, 1, -1), (a2 + c2).sum(0).reshape(1, 1, -1)
        if out1 is not None:
            out1.copy_(o1); o1 = out1
        if out2 is not None:
            out2.copy_(o2); o2 = out2
        return o1, o2

    def add3(self, a, b, c):
        return (a + b) + c

    def sum_groups3(self, x):
        return (x[0] + x[1]) + x[2]

    def bcast_groups3(self, d):
        return d.unsqueeze(0).expand((3,) + tuple(d.shape)).contiguous()

    def stats_channel_sum(self, stats, out):
        out.copy_(stats[:, :, 0].sum(0).float())
        return out

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

    def head_loss(self, logits, label, posmasks, scale):
        """cwf_head_loss_sums + cwf_dice_ce_finalize_multi: per map, the unfused chain upsample_softmax -> dice_ce."""
        outs = [self.dice_ce(self.upsample_softmax(lg, 2, scale), label, int(pm)) for lg, pm in zip(logits, posmasks)]
        loss = torch.cat([o[0] for o in outs])
        total = torch.zeros(1)
        for v in loss:
            total = total + v
        return total, loss, torch.stack([o[1] for o in outs])

    def head_loss_bwd(self, logits, label, posmasks, scale, coef, gscale, grouped_out=None):
        dls = []
        for m, (lg, pm) in enumerate(zip(logits, posmasks)):
            prob = self.upsample_softmax(lg, 2, scale)
            dprob = self.dice_ce_bwd(prob, label, int(pm), coef[m], gscale)
            n, d, h, w, ldc = lg.shape
            dl = self.upsample_softmax_bwd(dprob, prob, (n, d, h, w), 2, scale, ldc)
            if grouped_out is not None:                     # channel groups of one gradient buffer (cwf_head_loss_bwd_ex)
                buf, ca = grouped_out
                buf[..., m * ca:(m + 1) * ca] = dl[..., :ca]
                dl = buf[..., m * ca:(m + 1) * ca]
            dls.append(dl)
        return dls

    # ------------------------------------------------------------------ K11 / misc
    def wgrad_to_grouped(self, keys, op, xs, dys, cout, inv_maps, dw_dsts, db_dsts, prec=None, allow_async=False):
        for q in range(len(keys)):
            self.wgrad_to(keys[q], op, xs[q], None, None, 1.0, dys[q], cout, inv_maps[q], dw_dsts[q], db_dsts[q])

    def wgrad_to(self, key, op, x, in_scale, in_shift, slope, dy, cout, inv_map, dw_dst, db_dst, prec=None, allow_async=False):
        """gradient-sink form of wgrad: the result lands in (dw_dst, db_dst) directly (the HIP backend defers the reduction)"""
        shape = (cout,) + tuple(dw_dst.shape[1:]) if op != CONVT2 else tuple(dw_dst.shape)
        gw, gb = self.wgrad(op, x, in_scale, in_shift, slope, dy, cout, inv_map, db_dst is not None, 0, w_ref_shape=shape)
        # a fused layer (three convs on one input) reduces into ADJACENT slices of the flat buffer starting at dw_dst / db_dst
        torch.as_strided(dw_dst, (gw.numel(),), (1,), dw_dst.storage_offset()).copy_(gw.reshape(-1))
        if db_dst is not None:
            torch.as_strided(db_dst, (gb.numel(),), (1,), db_dst.storage_offset()).copy_(gb)

    def wgrad_flush(self, device=None):
        pass

    def adam(self, table, ntensors, max_n, lr, beta1, beta2, eps, wd, step, amsgrad, hyper_dev=None, grad_scale=1.0):
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
            g = g_ * grad_scale + wd * p_
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
        n = 1
        for d in shape:
            n *= d
        return self.keep(self.rng_site(n), n, p, p2).reshape(shape)

    def mul(self, a, b):
        return a * b

    def add(self, a, b):
        return a + b

    def channel_scale(self, x, s):
        return x * s[:, None, None, None, :]

    def copy_into(self, x, out):
        out.copy_(x)
        return out
