"""Edge-supported Intra-region Coupler and Mutual Cross-region Coupler on the HIP token kernels.

Module tree and parameter names follow the reference so that checkpoints load unchanged:
    <model>.cross_attention_list.0 = Residual(fn=PreNormDrop(norm, norm2, fn=DualSelfAttention(out_proj, qkv)))
    <model>.cross_ffn_list.0       = Residual(fn=PreNorm(norm, fn=FeedForward(net=[Linear, GELU, Dropout, Linear, Dropout])))
(ClsWiseTransformer.py:7-55, FusionClsWiseTransformer.py:8-54, SelfAttention.py:50-102, ResidualNorm.py:4-47).

Compute differences that do not change results: only W_q.x and W_kv.x2 are evaluated (the reference computes the
full 1536-wide qkv of both inputs and discards 1/3 resp. 2/3, SelfAttention.py:80-93); QK^T, softmax, PV run as
strided batched MFMA GEMMs over the 8 heads without permute/contiguous copies; residual adds are fused into the
GEMM epilogues when dropout is off.
"""
import torch
import torch.nn as nn

from cwf import functional as CF


class DualSelfAttention(nn.Module):
    def __init__(self, hidden_size, num_heads, dropout_rate=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.out_proj = nn.Linear(hidden_size, hidden_size)
        self.qkv = nn.Linear(hidden_size, hidden_size * 3, bias=False)
        self.dropout_rate = dropout_rate
        self.hidden = hidden_size

    def forward(self, x, x2, residual=None, out_keep=None):
        """x, x2 already layer-normed.  Returns out_proj(attn) (+ residual when given and no output dropout)."""
        e = self.hidden
        w = self.qkv.weight
        q = CF.linear(x, w[:e])
        kv = CF.linear(x2, w[e:])
        pmask = None
        if self.training and self.dropout_rate > 0:
            pmask = CF.dropout_mask((x.shape[0], self.num_heads, x.shape[1], x2.shape[1]), self.dropout_rate, x.device)
        o = CF.attention_core(q, kv, self.num_heads, pmask)
        if out_keep is None:
            return CF.linear(o, self.out_proj.weight, self.out_proj.bias, residual=residual)
        y = CF._MulMaskFn.apply(CF.linear(o, self.out_proj.weight, self.out_proj.bias), out_keep)
        return CF.add(y, residual) if residual is not None else y


class PreNormDrop(nn.Module):
    def __init__(self, dim, dropout_rate, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.dropout_rate = dropout_rate
        self.fn = fn


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout_rate):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(p=dropout_rate),
                                 nn.Linear(hidden_dim, dim), nn.Dropout(p=dropout_rate))
        self.dropout_rate = dropout_rate


class Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn


def _cross_attention(block: Residual, x, x2):
    """x + Drop(Drop(out_proj(Attn(LN1(x), LN2(x2)))))   (ResidualNorm.py:4-32 around SelfAttention.py:74-102)"""
    pre = block.fn
    attn = pre.fn
    a = CF.layer_norm(x, pre.norm.weight, pre.norm.bias)
    b = CF.layer_norm(x2, pre.norm2.weight, pre.norm2.bias)
    keep = None
    if block.training and (pre.dropout_rate > 0 or attn.dropout_rate > 0):
        # drop_output (inside the attention) and PreNormDrop.dropout act in sequence on the same tensor
        keep = CF.dropout_mask(x.shape, attn.dropout_rate, x.device, p2=pre.dropout_rate)
    return attn(a, b, residual=x, out_keep=keep)


def _ffn(block: Residual, x):
    """x + Drop(W2 Drop(GELU(W1 LN(x))))   (ResidualNorm.py:13-20,35-47)"""
    pre = block.fn
    ff = pre.fn
    h = CF.layer_norm(x, pre.norm.weight, pre.norm.bias)
    h = CF.linear(h, ff.net[0].weight, ff.net[0].bias, act=1)
    if block.training and ff.dropout_rate > 0:
        h = CF.dropout(h, ff.dropout_rate, True)
        y = CF.dropout(CF.linear(h, ff.net[3].weight, ff.net[3].bias), ff.dropout_rate, True)
        return CF.add(y, x)
    return CF.linear(h, ff.net[3].weight, ff.net[3].bias, residual=x)


def _make_lists(model, depth, heads, mlp_dim, dropout_rate, attn_dropout_rate):
    att, ffn = [], []
    for _ in range(depth):
        att.append(Residual(PreNormDrop(mlp_dim, dropout_rate, DualSelfAttention(mlp_dim, num_heads=heads, dropout_rate=attn_dropout_rate))))
        ffn.append(Residual(PreNorm(mlp_dim, FeedForward(mlp_dim, mlp_dim, dropout_rate))))
    model.cross_attention_list = nn.ModuleList(att)
    model.cross_ffn_list = nn.ModuleList(ffn)


class TwoClsWiseTransformerModel(nn.Module):
    """Edge-supported Intra-region Coupler: four cross-attentions sharing ONE weight set, then an FFN over the
    concatenated 258 tokens (ClsWiseTransformer.py:41-55)."""

    def __init__(self, depth, heads, mlp_dim, dropout_rate=0.1, attn_dropout_rate=0.1):
        super().__init__()
        self.depth = depth
        _make_lists(self, depth, heads, mlp_dim, dropout_rate, attn_dropout_rate)

    def forward(self, edge_fea, se_fea_supple, semantic_fea, supple_fea_edge):
        ca = self.cross_attention_list[0]
        edge_q_sem = _cross_attention(ca, edge_fea, se_fea_supple)
        sem_q_edge = _cross_attention(ca, semantic_fea, supple_fea_edge)
        result_edge = _cross_attention(ca, edge_q_sem, sem_q_edge)
        result_sem = _cross_attention(ca, sem_q_edge, edge_q_sem)
        return _ffn(self.cross_ffn_list[0], torch.cat((result_edge, result_sem), dim=1))


class FusionClsWiseTransformerModel(nn.Module):
    """Mutual Cross-region Coupler: self-attention + FFN on the fused tokens (FusionClsWiseTransformer.py:43-54)."""

    def __init__(self, depth, heads, mlp_dim, dropout_rate=0.1, attn_dropout_rate=0.1):
        super().__init__()
        self.depth = depth
        _make_lists(self, depth, heads, mlp_dim, dropout_rate, attn_dropout_rate)

    def forward(self, fusion_semantic):
        return _ffn(self.cross_ffn_list[0], _cross_attention(self.cross_attention_list[0], fusion_semantic, fusion_semantic))
