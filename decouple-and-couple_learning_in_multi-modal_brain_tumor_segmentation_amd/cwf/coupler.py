"""The two couplers of ClsWiseFormer as whole-block autograd Functions.

    RegionCouplerFn  token selection (cls_wise_former.py:345-376) -> Edge-supported Intra-region Coupler
                     (ClsWiseTransformer.py:41-55) -> row scatter + gate (cls_wise_former.py:457-543) of ONE sub-region
    FusionCouplerFn  selection (:552-560) -> Mutual Cross-region Coupler (FusionClsWiseTransformer.py:43-54) -> scatter + gate
                     (:565-579)

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
    """Static description of one coupler call (not a tensor argument of the Function)."""

    def __init__(self, heads, k, training, p_select, p_attn, p_pre, p_ffn, forced=None, names=None):
        self.heads, self.k, self.training = heads, k, training
        self.p_select = p_select if training else 0.0
        self.p_attn = p_attn if training else 0.0          # attention-probability dropout AND the attention's output dropout
        self.p_pre = p_pre if training else 0.0            # PreNormDrop.dropout (acts on the same tensor, in sequence)
        self.p_ffn = p_ffn if training else 0.0
        self.forced = forced or {}
        self.names = names or ()


# parameter order of a transformer weight set as the Functions take it
#   0 ln1_w  1 ln1_b  2 ln2_w  3 ln2_b  4 out_w  5 out_b  6 qkv_w  7 ffn_ln_w  8 ffn_ln_b  9 w1  10 b1  11 w2  12 b2
def transformer_params(model):
    ca = model.cross_attention_list[0].fn
    ff = model.cross_ffn_list[0].fn
    return (ca.norm.weight, ca.norm.bias, ca.norm2.weight, ca.norm2.bias, ca.fn.out_proj.weight, ca.fn.out_proj.bias, ca.fn.qkv.weight,
            ff.norm.weight, ff.norm.bias, ff.fn.net[0].weight, ff.fn.net[0].bias, ff.fn.net[3].weight, ff.fn.net[3].bias)


def _grad_buffers(P):
    """Destination of the weight-set gradients: slices of the Trainer's flat gradient buffer when a sink is active (the
    Functions then return None for them: autograd never sees these gradients), fresh tensors otherwise."""
    from .optim import active_sink
    sink = active_sink()
    if sink is not None:
        views = [sink.view(p) for p in P]
        if all(v is not None for v in views):
            for p in P:
                sink.mark(p)
            return views, True
    return [torch.empty_like(p) for p in P], False


def _site(K, cfg_p, n):
    return K.rng_site(n) if cfg_p > 0.0 else 0


# ---------------------------------------------------------------------------------------------------------------------
# one cross-attention block:  y = x + Drop_pre(Drop_attn(out_proj(Attn(LN1(x), LN2(x2')))))     (ResidualNorm.py:4-32)
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
    """dy [rows, E] -> (dx, dx2).  G = gradient buffers of the weight set (written when `first`, accumulated otherwise)."""
    x, x2, perm_T, a, b, stats, qkv, o, d_attn, d_out, z, t = saved
    e = x.shape[1]
    K.linear_wgrad(dy, o, G[4], G[5], accumulate=not first, drop=d_out)                       # d out_proj.weight / .bias
    d_o = K.linear_dgrad(dy, P[4], drop=d_out)
    dqkv = K.attn_bwd(qkv, d_o, z, t, cfg.heads, d_attn)
    da = K.linear_dgrad(dqkv[:, :e], P[6][:e])
    db = K.linear_dgrad(dqkv[:, e:], P[6][e:])
    K.linear_wgrad(dqkv, a, G[6], None, x2=b, split_m=e, accumulate=not first)                 # d qkv.weight = [dq^T a ; dkv^T b]
    return K.ln_pair_bwd(dy, da, db, x, x if x2 is None else x2, perm_T, P[0], P[2], stats, G[0], G[1], G[2], G[3],
                         accumulate=not first, want_dx2=dual)


# ---------------------------------------------------------------------------------------------------------------------
# FFN block:  y = x + Drop(W2 Drop(GELU(W1 LN(x) + b1)) + b2)                                (ResidualNorm.py:13-20,35-47)
# ---------------------------------------------------------------------------------------------------------------------
def _ffn_fwd(K, P, cfg, x):
    rows, e = x.shape
    h0, _, stats = K.ln_pair_fwd(x, None, 0, P[7], P[8], None, None)
    hid = P[9].shape[0]
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


def _select(K, cfg, name0, name1, feats, q0, q1, k):
    """Two selections of one token matrix (scored by q0 / q1) -> (idx0, inv0, idx1, inv1); teacher-forced sets override."""
    s0, s1 = K.token_scores2(feats, q0, q1)
    idx0, inv0, idx1, inv1 = K.topk_inv(s0, s1, k)
    t = feats.shape[1]
    if name0 in cfg.forced:
        idx0, inv0 = K.index_inv(cfg.forced[name0], t)
    if name1 is not None and name1 in cfg.forced:
        idx1, inv1 = K.index_inv(cfg.forced[name1], t)
    return idx0, inv0, idx1, inv1


class RegionCouplerFn(torch.autograd.Function):
    """(E [B,Te,512], S [B,Ts,512], e_tok, s_tok, 13 weights) -> (gated_e, gated_s, scat_s, sem_tok [B,1,512], 4 index sets).
    cfg.names = (edge, sem_supp, sem, edge_supp) keys for teacher forcing / aux."""

    @staticmethod
    def forward(ctx, cfg, E, S, e_tok, s_tok, *P):
        K = backend()
        ctx.set_materialize_grads(False)     # unused outputs arrive as None in backward, not as zero-filled tensors
        E, S = E.contiguous(), S.contiguous()
        b, _, e = E.shape
        k = min(cfg.k, S.shape[1], E.shape[1])
        t = k + 1
        n_e, n_ss, n_s, n_es = cfg.names
        qe, qs = e_tok.detach(), s_tok.detach()
        idx_e, inv_e, idx_es, inv_es = _select(K, cfg, n_e, n_es, E, qe, qs, k)        # E scored by e_tok (primary) / by s_tok (supplement)
        idx_se, inv_se, idx_s, inv_s = _select(K, cfg, n_ss, n_s, S, qe, qs, k)        # S scored by e_tok (supplement) / by s_tok (primary)
        X1 = torch.empty((b, 2, t, e), dtype=_f32, device=E.device)                  # [edge_seq ; sem_seq]
        X2 = torch.empty((b, 2, t, e), dtype=_f32, device=E.device)                  # [sem_supp ; edge_supp]
        offs = [_site(K, cfg.p_select, b * k * e) for _ in range(4)]
        K.gather_multi([(E, idx_e, e_tok, X1[:, 0], offs[0]), (S, idx_s, s_tok, X1[:, 1], offs[1]),
                        (S, idx_se, s_tok, X2[:, 0], offs[2]), (E, idx_es, e_tok, X2[:, 1], offs[3])], k, e, p=cfg.p_select)
        rows, z = b * 2 * t, b * 2
        y1, sv1 = _ca_fwd(K, P, cfg, X1.view(rows, e), X2.view(rows, e), 0, z, t)      # a = CA(edge, sem'), b = CA(sem, edge')
        y2, sv2 = _ca_fwd(K, P, cfg, y1, None, t, z, t)                                 # CA(a, b), CA(b, a)
        r, sv3 = _ffn_fwd(K, P, cfg, y2)                                                # FFN(cat(result_edge, result_sem))
        R = r.view(b, 2, t, e)
        gated_e, _ = K.scatter_inv(E, inv_e, R[:, 0, 1:], R[:, 0, 0:1], want_gated=True, want_scat=False)
        gated_s, scat_s = K.scatter_inv(S, inv_s, R[:, 1, 1:], R[:, 1, 0:1], want_gated=True, want_scat=True)
        sem_tok = R[:, 1, 0:1]
        ctx.cfg, ctx.sv, ctx.offs, ctx.k = cfg, (sv1, sv2, sv3), offs, k
        ctx.toks = (e_tok, s_tok)
        ctx.save_for_backward(E, S, R, idx_e, inv_e, inv_es, idx_s, inv_s, inv_se, *P)
        idx = (idx_e, idx_se, idx_s, idx_es)
        ctx.mark_non_differentiable(*idx)
        return (gated_e, gated_s, scat_s, sem_tok) + idx

    @staticmethod
    def backward(ctx, dgated_e, dgated_s, dscat_s, dsem_tok, *_):
        K = backend()
        cfg, (sv1, sv2, sv3), offs, k = ctx.cfg, ctx.sv, ctx.offs, ctx.k
        E, S, R, idx_e, inv_e, inv_es, idx_s, inv_s, inv_se = ctx.saved_tensors[:9]
        P = ctx.saved_tensors[9:]
        b, _, t, e = R.shape
        c = lambda g: None if g is None else g.contiguous()
        dgated_e, dgated_s, dscat_s, dsem_tok = c(dgated_e), c(dgated_s), c(dscat_s), c(dsem_tok)
        G, sunk = _grad_buffers(P)
        dR = torch.empty_like(R)
        K.scatter_bwd(dgated_e, None, E, inv_e, idx_e, R[:, 0, 1:], R[:, 0, 0:1], None, dR[:, 0, 1:], dR[:, 0, 0:1])
        K.scatter_bwd(dgated_s, dscat_s, S, inv_s, idx_s, R[:, 1, 1:], R[:, 1, 0:1], dsem_tok, dR[:, 1, 1:], dR[:, 1, 0:1])
        rows = b * 2 * t
        dy2 = _ffn_bwd(K, P, G, sv3, dR.view(rows, e))
        dy1, _ = _ca_bwd(K, P, G, cfg, sv2, dy2, first=True, dual=False)
        dx1, dx2 = _ca_bwd(K, P, G, cfg, sv1, dy1, first=False, dual=True)
        dX1, dX2 = dx1.view(b, 2, t, e), dx2.view(b, 2, t, e)
        dE = K.token_grad(dgated_e, None, R[:, 0, 0:1], inv_e, inv_es, dX1[:, 0], dX2[:, 1], k, cfg.p_select, offs[0], offs[3])
        dS = K.token_grad(dgated_s, dscat_s, R[:, 1, 0:1], inv_s, inv_se, dX1[:, 1], dX2[:, 0], k, cfg.p_select, offs[1], offs[2])
        e_tok, s_tok = ctx.toks
        (tk, tk_sunk) = _grad_buffers((e_tok, s_tok))
        d_etok, d_stok = K.head_grad(dX1[:, 0, 0], dX2[:, 1, 0], dX1[:, 1, 0], dX2[:, 0, 0], out1=tk[0], out2=tk[1])
        if tk_sunk:
            d_etok = d_stok = None
        return (None, dE, dS, d_etok, d_stok) + (tuple(G) if not sunk else (None,) * len(G))


class FusionCouplerFn(torch.autograd.Function):
    """(feats [B,Ts,512], tok [B,1,512] per-sample class token, 13 weights) -> (fused = scatter * gate, index set)."""

    @staticmethod
    def forward(ctx, cfg, feats, tok, *P):
        K = backend()
        ctx.set_materialize_grads(False)
        feats, tok = feats.contiguous(), tok.contiguous()
        b, ts, e = feats.shape
        k = min(cfg.k, ts)
        t = k + 1
        idx, inv, _, _ = _select(K, cfg, cfg.names[0], None, feats, tok.detach(), None, k)
        X = torch.empty((b, t, e), dtype=_f32, device=feats.device)
        off = _site(K, cfg.p_select, b * k * e)
        K.gather_multi([(feats, idx, tok, X, off)], k, e, p=cfg.p_select)
        rows = b * t
        y1, sv1 = _ca_fwd(K, P, cfg, X.view(rows, e), None, 0, b, t)
        r, sv2 = _ffn_fwd(K, P, cfg, y1)
        R = r.view(b, t, e)
        fused, _ = K.scatter_inv(feats, inv, R[:, 1:], R[:, 0:1], want_gated=True, want_scat=False)
        ctx.cfg, ctx.sv, ctx.off, ctx.k = cfg, (sv1, sv2), off, k
        ctx.save_for_backward(feats, R, idx, inv, *P)
        ctx.mark_non_differentiable(idx)
        return fused, idx

    @staticmethod
    def backward(ctx, dfused, _):
        K = backend()
        cfg, (sv1, sv2), off, k = ctx.cfg, ctx.sv, ctx.off, ctx.k
        feats, R, idx, inv = ctx.saved_tensors[:4]
        P = ctx.saved_tensors[4:]
        b, t, e = R.shape
        dfused = dfused.contiguous()
        G, sunk = _grad_buffers(P)
        dR = torch.empty_like(R)
        K.scatter_bwd(dfused, None, feats, inv, idx, R[:, 1:], R[:, 0:1], None, dR[:, 1:], dR[:, 0:1])
        dy1 = _ffn_bwd(K, P, G, sv2, dR.view(b * t, e))
        dx, _ = _ca_bwd(K, P, G, cfg, sv1, dy1, first=True, dual=False)
        dX = dx.view(b, t, e)
        dfeats = K.token_grad(dfused, None, R[:, 0:1], inv, None, dX, None, k, cfg.p_select, off, 0)
        return (None, dfeats, dX[:, 0:1]) + (tuple(G) if not sunk else (None,) * len(G))


class _Add3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        return backend().add3(a, b, c)

    @staticmethod
    def backward(ctx, d):
        return d, d, d


def add3(a, b, c):
    return _Add3Fn.apply(a, b, c)
