"""The two couplers of ClsWiseFormer as whole-block autograd Functions.

    RegionCouplerFn  token selection (cls_wise_former.py:345-376) -> Edge-supported Intra-region Coupler
                     (ClsWiseTransformer.py:41-55) -> row scatter + gate (cls_wise_former.py:457-543) of ONE sub-region
    FusionCouplerFn  selection (:552-560) -> Mutual Cross-region Coupler (FusionClsWiseTransformer.py:43-54) -> scatter + gate
                     (:565-579)
    IntraCouplerBlockFn / FusionBlockFn   the transformer modules' own forward() on given sequences (same block launches)

Round 1 ran these as ~60 small autograd nodes per region (each Linear, LayerNorm, softmax, mask multiply, add ... its own
Function): ~800 launches and most of the ATen glue of a step (gradient accumulation of the weight set shared by the four
cross-attentions, zero-filled scatter targets, slice backward fills), all host-bound.  Here one Function sequences the launches
of a whole coupler by hand, forward and backward:

  * the four cross-attentions of a region run as TWO batches of sequence pairs [B][2][129][512]: (edge <- sem', sem <- edge')
    and then (a <- b, b <- a) -- the second operand of the latter is the first with the halves of each pair swapped, which the
    paired LayerNorm kernel reads in place (perm_T);
  * one block = paired LayerNorm -> ONE GEMM for q | k | v over the reference's [1536, 512] qkv weight (operand switch at column
    512; the reference computes and discards 1/3 resp. 2/3 of a full qkv for each operand, SelfAttention.py:80-93) -> ONE attention
    launch (QK^T, softmax, dropout, PV in LDS) -> out_proj GEMM with bias + dropout + residual in its epilogue;
  * weight gradients of the shared weight set are accumulated by the GEMMs themselves (accumulate flag), bias gradients come out
    of the same GEMMs (rowsum), LayerNorm parameter gradients from a deterministic column kernel: nothing is zero-filled and
    nothing is summed by autograd;
  * dropout masks are never stored: every site has a counter offset, the kernels recompute keep(i) forward and backward.

All arithmetic is in csrc/*.hip (cwf/kernels.py wrappers); this file only sequences launches and owns tensors.
"""
from __future__ import annotations

import torch

from .kernels import backend

_f32 = torch.float32


class CouplerConfig:
    """Static description of one coupler call (not a tensor argument of the Function).  `names`: per group, the keys of its
    selections (4 for a region: edge, sem_supp, sem, edge_supp; 1 for the fusion) for teacher forcing / aux."""

    def __init__(self, heads, k, training, p_select, p_attn, p_pre, p_ffn, forced=None, names=None, groups=1):
        self.heads, self.k, self.training, self.groups = heads, k, training, groups
        self.p_select = p_select if training else 0.0
        self.p_attn = p_attn if training else 0.0          # attention-probability dropout AND the attention's output dropout
        self.p_pre = p_pre if training else 0.0            # PreNormDrop.dropout (acts on the same tensor, in sequence)
        self.p_ffn = p_ffn if training else 0.0
        self.forced = forced or {}
        self.names = names or ()


# parameter order of a transformer weight set as the Functions take it
#   0 ln1_w  1 ln1_b  2 ln2_w  3 ln2_b  4 out_w  5 out_b  6 qkv_w  7 ffn_ln_w  8 ffn_ln_b  9 w1  10 b1  11 w2  12 b2
NP = 13


def transformer_params(model):
    ca = model.cross_attention_list[0].fn
    ff = model.cross_ffn_list[0].fn
    return (ca.norm.weight, ca.norm.bias, ca.norm2.weight, ca.norm2.bias, ca.fn.out_proj.weight, ca.fn.out_proj.bias, ca.fn.qkv.weight,
            ff.norm.weight, ff.norm.bias, ff.fn.net[0].weight, ff.fn.net[0].bias, ff.fn.net[3].weight, ff.fn.net[3].bias)


def _by_param(flat, groups):
    """[set_0 (13 tensors), set_1, ...] -> P with P[i] = [tensor i of set 0, of set 1, ...]  (what the grouped kernels take)"""
    return [[flat[g * NP + i] for g in range(groups)] for i in range(NP)]


def _grad_buffers(P):
    """Destination of parameter gradients (P: list of lists of parameters): slices of the Trainer's flat gradient buffer when
    a sink is active (the Functions then return None for them: autograd never sees these gradients), fresh tensors otherwise."""
    from .optim import active_sink
    sink = active_sink()
    if sink is not None:
        views = [[sink.view(p) for p in ps] for ps in P]
        if all(v is not None for vs in views for v in vs):
            for ps in P:
                for p in ps:
                    sink.mark(p)
            return views, True
    return [[torch.empty_like(p) for p in ps] for ps in P], False


def _site(K, cfg_p, n):
    return K.rng_site(n) if cfg_p > 0.0 else 0


# ---------------------------------------------------------------------------------------------------------------------
# one cross-attention block:  y = x + Drop_pre(Drop_attn(out_proj(Attn(LN1(x), LN2(x2')))))     (ResidualNorm.py:4-32)
# (x: the rows of all groups stacked; P[i]: one tensor per group)
# ---------------------------------------------------------------------------------------------------------------------
def _ca_fwd(K, P, cfg, x, x2, perm_T, z, t):
    """x, x2: [rows, E] (x2 None = x itself).  Returns (y, saved)."""
    rows, e = x.shape
    a, b, stats = K.ln_pair_fwd(x, x if x2 is None else x2, perm_T, P[0], P[1], P[2], P[3])
    qkv = torch.empty((rows, 3 * e), dtype=_f32, device=x.device)
    K.linear_fwd(a, P[6], None, qkv, x2=b, split_n=e)
    d_attn = (_site(K, cfg.p_attn, z * cfg.heads * t * t), cfg.p_attn) if cfg.p_attn > 0 else None
    o = K.attn_fwd(qkv, z, t, cfg.heads, d_attn)
    d_out = None                                   # drop_output inside the attention, then PreNormDrop.dropout, on the same tensor
    if cfg.p_attn > 0 or cfg.p_pre > 0:
        p, p2 = (cfg.p_attn, cfg.p_pre) if cfg.p_attn > 0 else (cfg.p_pre, 0.0)
        d_out = (K.rng_site(rows * e), p, p2)
    y = torch.empty_like(x)
    K.linear_fwd(o, P[4], P[5], y, drop=d_out, residual=x)
    return y, (x, x2, perm_T, a, b, stats, qkv, o, d_attn, d_out, z, t)


def _ca_bwd(K, P, G, cfg, saved, dy, first, dual):
    """dy [rows, E] -> (dx, dx2).  G = gradient buffers of the weight sets (written when `first`, accumulated otherwise)."""
    x, x2, perm_T, a, b, stats, qkv, o, d_attn, d_out, z, t = saved
    e = x.shape[1]
    K.linear_wgrad(dy, o, G[4], G[5], accumulate=not first, drop=d_out)                       # d out_proj.weight / .bias
    d_o = K.linear_dgrad(dy, P[4], drop=d_out)
    dqkv = K.attn_bwd(qkv, d_o, z, t, cfg.heads, d_attn)
    da = K.linear_dgrad(dqkv[:, :e], [w[:e] for w in P[6]])
    db = K.linear_dgrad(dqkv[:, e:], [w[e:] for w in P[6]])
    K.linear_wgrad(dqkv, a, G[6], None, x2=b, split_m=e, accumulate=not first)                 # d qkv.weight = [dq^T a ; dkv^T b]
    return K.ln_pair_bwd(dy, da, db, x, x if x2 is None else x2, perm_T, P[0], P[2], stats, G[0], G[1], G[2], G[3],
                         accumulate=not first, want_dx2=dual)


# ---------------------------------------------------------------------------------------------------------------------
# FFN block:  y = x + Drop(W2 Drop(GELU(W1 LN(x) + b1)) + b2)                                (ResidualNorm.py:13-20,35-47)
# ---------------------------------------------------------------------------------------------------------------------
def _ffn_fwd(K, P, cfg, x):
    rows, e = x.shape
    h0, _, stats = K.ln_pair_fwd(x, None, 0, P[7], P[8], None, None)
    hid = P[9][0].shape[0]
    zpre = torch.empty((rows, hid), dtype=_f32, device=x.device)
    h1 = torch.empty((rows, hid), dtype=_f32, device=x.device)
    d1 = (_site(K, cfg.p_ffn, rows * hid), cfg.p_ffn, 0.0) if cfg.p_ffn > 0 else None
    K.linear_fwd(h0, P[9], P[10], h1, act=1, pre=zpre, drop=d1)
    d2 = (_site(K, cfg.p_ffn, rows * e), cfg.p_ffn, 0.0) if cfg.p_ffn > 0 else None
    y = torch.empty_like(x)
    K.linear_fwd(h1, P[11], P[12], y, drop=d2, residual=x)
    return y, (x, h0, stats, zpre, h1, d1, d2)


def _ffn_bwd(K, P, G, saved, dy):
    x, h0, stats, zpre, h1, d1, d2 = saved
    K.linear_wgrad(dy, h1, G[11], G[12], drop=d2)
    dh1 = K.linear_dgrad(dy, P[11], drop=d2)
    dz = K.gelu_bwd_drop(zpre, dh1, (d1[0], d1[1]) if d1 else None)
    K.linear_wgrad(dz, h0, G[9], G[10])
    dh0 = K.linear_dgrad(dz, P[9])
    dx, _ = K.ln_pair_bwd(dy, dh0, None, x, None, 0, P[7], None, stats, G[7], G[8], None, None, accumulate=False, want_dx2=False)
    return dx


def _select(K, cfg, j0, j1, feats, q0, q1, k, bg):
    """Two selections of one stacked token matrix [G*bg, T, E] (scored by the groups' q0 / q1) -> (idx0, inv0, idx1, inv1);
    teacher-forced sets (cfg.forced[cfg.names[g][j]]) override a group's rows."""
    s0, s1 = K.token_scores2(feats, q0, q1)
    idx0, inv0, idx1, inv1 = K.topk_inv(s0, s1, k)
    t = feats.shape[1]
    out = []
    for j, idx, inv in ((j0, idx0, inv0), (j1, idx1, inv1)):
        if j is not None and any(names[j] in cfg.forced for names in cfg.names):
            # (assembled with torch.cat -- a kernel -- rather than slice assignment: a contiguous device-to-device copy is a memcpy
            # node under stream capture, which the launch plan of cwf.trainer does not take)
            pi, pv = [], []
            for g, names in enumerate(cfg.names):
                if names[j] in cfg.forced:
                    fi, fv = K.index_inv(cfg.forced[names[j]], t)
                else:
                    fi, fv = idx[g * bg:(g + 1) * bg], inv[g * bg:(g + 1) * bg]
                pi.append(fi)
                pv.append(fv)
            idx, inv = torch.cat(pi, 0), torch.cat(pv, 0)
        out += [idx, inv]
    return tuple(out)


class RegionCouplerFn(torch.autograd.Function):
    """The Edge-supported Intra-region Couplers of ALL sub-regions in one pass (groups = 3: every launch covers the three regions;
    their weight sets, class tokens and LayerNorm parameters reach the kernels as per-group pointer tables).

    (E [G,B,Te,512], S [G,B,Ts,512], G e_tokens, G s_tokens, G x 13 weights)
        -> (gated_e [G,B,Te,512], gated_s, scat_s [G,B,Ts,512], sem_tok [G,B,1,512], 4 index sets [G*B,k])
    cfg.names[g] = (edge, sem_supp, sem, edge_supp) keys of group g for teacher forcing / aux."""

    @staticmethod
    def forward(ctx, cfg, E, S, *flat):
        K = backend()
        ctx.set_materialize_grads(False)     # unused outputs arrive as None in backward, not as zero-filled tensors
        G = cfg.groups
        e_toks, s_toks = list(flat[:G]), list(flat[G:2 * G])
        P = _by_param(flat[2 * G:], G)
        E, S = E.contiguous(), S.contiguous()
        _, b, te, e = E.shape
        ts = S.shape[2]
        Ef, Sf = E.view(G * b, te, e), S.view(G * b, ts, e)
        k = min(cfg.k, ts, te)
        t = k + 1
        qe, qs = [q.detach() for q in e_toks], [q.detach() for q in s_toks]
        idx_e, inv_e, idx_es, inv_es = _select(K, cfg, 0, 3, Ef, qe, qs, k, b)         # E scored by e_tok (primary) / by s_tok (supplement)
        idx_se, inv_se, idx_s, inv_s = _select(K, cfg, 1, 2, Sf, qe, qs, k, b)         # S scored by e_tok (supplement) / by s_tok (primary)
        gb = G * b
        X1 = torch.empty((gb, 2, t, e), dtype=_f32, device=E.device)                  # [edge_seq ; sem_seq]
        X2 = torch.empty((gb, 2, t, e), dtype=_f32, device=E.device)                  # [sem_supp ; edge_supp]
        offs = [_site(K, cfg.p_select, gb * k * e) for _ in range(4)]
        K.gather_multi([(Ef, idx_e, e_toks, X1[:, 0], offs[0]), (Sf, idx_s, s_toks, X1[:, 1], offs[1]),
                        (Sf, idx_se, s_toks, X2[:, 0], offs[2]), (Ef, idx_es, e_toks, X2[:, 1], offs[3])], k, e, p=cfg.p_select)
        rows, z = gb * 2 * t, gb * 2
        y1, sv1 = _ca_fwd(K, P, cfg, X1.view(rows, e), X2.view(rows, e), 0, z, t)      # a = CA(edge, sem'), b = CA(sem, edge')
        y2, sv2 = _ca_fwd(K, P, cfg, y1, None, t, z, t)                                 # CA(a, b), CA(b, a)
        r, sv3 = _ffn_fwd(K, P, cfg, y2)                                                # FFN(cat(result_edge, result_sem))
        R = r.view(gb, 2, t, e)
        gated_e, _ = K.scatter_inv(Ef, inv_e, R[:, 0, 1:], R[:, 0, 0:1], want_gated=True, want_scat=False)
        gated_s, scat_s = K.scatter_inv(Sf, inv_s, R[:, 1, 1:], R[:, 1, 0:1], want_gated=True, want_scat=True)
        sem_tok = R[:, 1, 0:1]
        ctx.cfg, ctx.sv, ctx.offs, ctx.k = cfg, (sv1, sv2, sv3), offs, k
        ctx.toks = (e_toks, s_toks)
        ctx.save_for_backward(Ef, Sf, R, idx_e, inv_e, inv_es, idx_s, inv_s, inv_se, *flat[2 * G:])
        idx = (idx_e, idx_se, idx_s, idx_es)
        ctx.mark_non_differentiable(*idx)
        return (gated_e.view(G, b, te, e), gated_s.view(G, b, ts, e), scat_s.view(G, b, ts, e), sem_tok.reshape(G, b, 1, e)) + idx

    @staticmethod
    def backward(ctx, dgated_e, dgated_s, dscat_s, dsem_tok, *_):
        K = backend()
        cfg, (sv1, sv2, sv3), offs, k = ctx.cfg, ctx.sv, ctx.offs, ctx.k
        G = cfg.groups
        Ef, Sf, R, idx_e, inv_e, inv_es, idx_s, inv_s, inv_se = ctx.saved_tensors[:9]
        P = _by_param(ctx.saved_tensors[9:], G)
        gb, _, t, e = R.shape
        te, ts = Ef.shape[1], Sf.shape[1]
        c = lambda g, n: None if g is None else g.contiguous().view(gb, n, e)
        dgated_e, dgated_s, dscat_s, dsem_tok = c(dgated_e, te), c(dgated_s, ts), c(dscat_s, ts), c(dsem_tok, 1)
        Gw, sunk = _grad_buffers(P)
        dR = torch.empty_like(R)
        K.scatter_bwd(dgated_e, None, Ef, inv_e, idx_e, R[:, 0, 1:], R[:, 0, 0:1], None, dR[:, 0, 1:], dR[:, 0, 0:1])
        K.scatter_bwd(dgated_s, dscat_s, Sf, inv_s, idx_s, R[:, 1, 1:], R[:, 1, 0:1], dsem_tok, dR[:, 1, 1:], dR[:, 1, 0:1])
        rows = gb * 2 * t
        dy2 = _ffn_bwd(K, P, Gw, sv3, dR.view(rows, e))
        dy1, _ = _ca_bwd(K, P, Gw, cfg, sv2, dy2, first=True, dual=False)
        dx1, dx2 = _ca_bwd(K, P, Gw, cfg, sv1, dy1, first=False, dual=True)
        dX1, dX2 = dx1.view(gb, 2, t, e), dx2.view(gb, 2, t, e)
        dE = K.token_grad(dgated_e, None, R[:, 0, 0:1], inv_e, inv_es, dX1[:, 0], dX2[:, 1], k, cfg.p_select, offs[0], offs[3])
        dS = K.token_grad(dgated_s, dscat_s, R[:, 1, 0:1], inv_s, inv_se, dX1[:, 1], dX2[:, 0], k, cfg.p_select, offs[1], offs[2])
        e_toks, s_toks = ctx.toks
        (tk, tk_sunk) = _grad_buffers([e_toks, s_toks])
        K.head_grad(dX1[:, 0, 0], dX2[:, 1, 0], dX1[:, 1, 0], dX2[:, 0, 0], out1=tk[0], out2=tk[1])
        b = gb // G
        dtoks = ((None,) * (2 * G)) if tk_sunk else (tuple(tk[0]) + tuple(tk[1]))
        dP = ((None,) * (NP * G)) if sunk else tuple(Gw[i][g] for g in range(G) for i in range(NP))
        return (None, dE.view(G, b, te, e), dS.view(G, b, ts, e)) + dtoks + dP


class FusionCouplerFn(torch.autograd.Function):
    """(feats [B,Ts,512], tok [B,1,512] per-sample class token, 13 weights) -> (fused = scatter * gate, index set)."""

    @staticmethod
    def forward(ctx, cfg, feats, tok, *flat):
        K = backend()
        ctx.set_materialize_grads(False)
        P = _by_param(flat, 1)
        feats, tok = feats.contiguous(), tok.contiguous()
        b, ts, e = feats.shape
        k = min(cfg.k, ts)
        t = k + 1
        s0, _ = K.token_scores2(feats, tok.detach(), None)
        idx, inv, _, _ = K.topk_inv(s0, None, k)
        name = cfg.names[0][0]
        if name in cfg.forced:
            idx, inv = K.index_inv(cfg.forced[name], ts)
        X = torch.empty((b, t, e), dtype=_f32, device=feats.device)
        off = _site(K, cfg.p_select, b * k * e)
        K.gather_multi([(feats, idx, tok, X, off)], k, e, p=cfg.p_select)
        rows = b * t
        y1, sv1 = _ca_fwd(K, P, cfg, X.view(rows, e), None, 0, b, t)
        r, sv2 = _ffn_fwd(K, P, cfg, y1)
        R = r.view(b, t, e)
        fused, _ = K.scatter_inv(feats, inv, R[:, 1:], R[:, 0:1], want_gated=True, want_scat=False)
        ctx.cfg, ctx.sv, ctx.off, ctx.k = cfg, (sv1, sv2), off, k
        ctx.save_for_backward(feats, R, idx, inv, *flat)
        ctx.mark_non_differentiable(idx)
        return fused, idx

    @staticmethod
    def backward(ctx, dfused, _):
        K = backend()
        cfg, (sv1, sv2), off, k = ctx.cfg, ctx.sv, ctx.off, ctx.k
        feats, R, idx, inv = ctx.saved_tensors[:4]
        P = _by_param(ctx.saved_tensors[4:], 1)
        b, t, e = R.shape
        dfused = dfused.contiguous()
        Gw, sunk = _grad_buffers(P)
        dR = torch.empty_like(R)
        K.scatter_bwd(dfused, None, feats, inv, idx, R[:, 1:], R[:, 0:1], None, dR[:, 1:], dR[:, 0:1])
        dy1 = _ffn_bwd(K, P, Gw, sv2, dR.view(b * t, e))
        dx, _ = _ca_bwd(K, P, Gw, cfg, sv1, dy1, first=True, dual=False)
        dX = dx.view(b, t, e)
        dfeats = K.token_grad(dfused, None, R[:, 0:1], inv, None, dX, None, k, cfg.p_select, off, 0)
        return (None, dfeats, dX[:, 0:1]) + (((None,) * NP) if sunk else tuple(Gw[i][0] for i in range(NP)))


class IntraCouplerBlockFn(torch.autograd.Function):
    """TwoClsWiseTransformerModel.forward (ClsWiseTransformer.py:41-55) on four given sequences [B,t,512]:
        a = CA(edge, sem_supp)  b = CA(sem, edge_supp)  re = CA(a, b)  rs = CA(b, a)  ->  FFN(cat(re, rs))   [B, 2t, 512]
    -- the same three launch groups (_ca_fwd x 2, _ffn_fwd) RegionCouplerFn runs between its selection and its scatter, for callers
    that use the module on its own."""

    @staticmethod
    def forward(ctx, cfg, edge, sem_supp, sem, edge_supp, *flat):
        K = backend()
        ctx.set_materialize_grads(False)
        P = _by_param(flat, 1)
        b, t, e = edge.shape
        X1 = torch.stack((edge, sem), 1).contiguous()            # [B,2,t,E]: the pair layout of the coupler kernels
        X2 = torch.stack((sem_supp, edge_supp), 1).contiguous()
        rows, z = b * 2 * t, b * 2
        y1, sv1 = _ca_fwd(K, P, cfg, X1.view(rows, e), X2.view(rows, e), 0, z, t)
        y2, sv2 = _ca_fwd(K, P, cfg, y1, None, t, z, t)
        r, sv3 = _ffn_fwd(K, P, cfg, y2)
        ctx.cfg, ctx.sv, ctx.shape = cfg, (sv1, sv2, sv3), (b, t, e)
        ctx.save_for_backward(*flat)
        return r.view(b, 2 * t, e)

    @staticmethod
    def backward(ctx, dr):
        K = backend()
        cfg, (sv1, sv2, sv3) = ctx.cfg, ctx.sv
        b, t, e = ctx.shape
        P = _by_param(ctx.saved_tensors, 1)
        Gw, sunk = _grad_buffers(P)
        dy2 = _ffn_bwd(K, P, Gw, sv3, dr.contiguous().view(b * 2 * t, e))
        dy1, _ = _ca_bwd(K, P, Gw, cfg, sv2, dy2, first=True, dual=False)
        dx1, dx2 = _ca_bwd(K, P, Gw, cfg, sv1, dy1, first=False, dual=True)
        dX1, dX2 = dx1.view(b, 2, t, e), dx2.view(b, 2, t, e)
        dP = ((None,) * NP) if sunk else tuple(Gw[i][0] for i in range(NP))
        return (None, dX1[:, 0], dX2[:, 0], dX1[:, 1], dX2[:, 1]) + dP


class FusionBlockFn(torch.autograd.Function):
    """FusionClsWiseTransformerModel.forward (FusionClsWiseTransformer.py:43-54): FFN(CA(x, x)) on a given sequence [B,t,512]."""

    @staticmethod
    def forward(ctx, cfg, x, *flat):
        K = backend()
        ctx.set_materialize_grads(False)
        P = _by_param(flat, 1)
        b, t, e = x.shape
        y1, sv1 = _ca_fwd(K, P, cfg, x.contiguous().view(b * t, e), None, 0, b, t)
        r, sv2 = _ffn_fwd(K, P, cfg, y1)
        ctx.cfg, ctx.sv, ctx.shape = cfg, (sv1, sv2), (b, t, e)
        ctx.save_for_backward(*flat)
        return r.view(b, t, e)

    @staticmethod
    def backward(ctx, dr):
        K = backend()
        cfg, (sv1, sv2) = ctx.cfg, ctx.sv
        b, t, e = ctx.shape
        P = _by_param(ctx.saved_tensors, 1)
        Gw, sunk = _grad_buffers(P)
        dy1 = _ffn_bwd(K, P, Gw, sv2, dr.contiguous().view(b * t, e))
        dx, _ = _ca_bwd(K, P, Gw, cfg, sv1, dy1, first=True, dual=False)
        return (None, dx.view(b, t, e)) + (((None,) * NP) if sunk else tuple(Gw[i][0] for i in range(NP)))


class _Add3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        return backend().add3(a, b, c)

    @staticmethod
    def backward(ctx, d):
        return d, d, d


def add3(a, b, c):
    return _Add3Fn.apply(a, b, c)


class _SumGroups3Fn(torch.autograd.Function):
    """x [3, ...] -> x[0] + x[1] + x[2] (the three-region sums of the Mutual Cross-region Coupler, cls_wise_former.py:549-552); the
    adjoint is one broadcast launch (indexing the stacked tensor would cost three zero fills, three copies and two adds in autograd)."""

    @staticmethod
    def forward(ctx, x):
        return backend().sum_groups3(x.contiguous())

    @staticmethod
    def backward(ctx, d):
        return backend().bcast_groups3(d)


def sum_groups3(x):
    return _SumGroups3Fn.apply(x)


class _SplitChannels3Fn(torch.autograd.Function):
    """[N,D,H,W,3C] -> three channel-slice VIEWS (no copy: the kernels take (pointer, channel stride)); the adjoint assembles the
    three slice gradients with one launch (autograd's own slice backward would be three zero fills, three copies and two adds)."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        c = x.shape[-1] // 3
        ctx.shape, ctx.dev = tuple(x.shape[:-1]) + (c,), x.device
        return x[..., :c], x[..., c:2 * c], x[..., 2 * c:]

    @staticmethod
    def backward(ctx, d0, d1, d2):
        return backend().cat3_channels([d0, d1, d2], ctx.shape, ctx.dev)


def split_channels3(x):
    return _SplitChannels3Fn.apply(x)


class _WindowToTokensGFn(torch.autograd.Function):
    """grouped convert_dim (cls_wise_former.py:15-23): the three sub-regions' token matrices [3,B,T,512] in one launch"""

    @staticmethod
    def forward(ctx, x, groups, patch):
        ctx.patch, ctx.size, ctx.c = patch, tuple(x.shape[1:4]), x.shape[4] // groups
        return backend().window_to_tokens_g(x, groups, patch)

    @staticmethod
    def backward(ctx, d):
        return backend().tokens_to_window_g(d, ctx.size, ctx.c, ctx.patch), None, None


class _TokensToWindowGFn(torch.autograd.Function):
    """grouped split_dim (cls_wise_former.py:26-39)"""

    @staticmethod
    def forward(ctx, tok, size, channels, patch):
        ctx.patch, ctx.groups = patch, tok.shape[0]
        return backend().tokens_to_window_g(tok, size, channels, patch)

    @staticmethod
    def backward(ctx, d):
        return backend().window_to_tokens_g(d, ctx.groups, ctx.patch), None, None, None


def window_to_tokens_g(x, groups, patch):
    return _WindowToTokensGFn.apply(x, groups, tuple(patch))


def tokens_to_window_g(tok, size, channels, patch):
    return _TokensToWindowGFn.apply(tok, tuple(size), channels, tuple(patch))
