"""Edge-supported Intra-region Coupler and Mutual Cross-region Coupler: parameter containers with the reference's module tree
and parameter names, so that checkpoints load unchanged:
    <model>.cross_attention_list.0 = Residual(fn=PreNormDrop(norm, norm2, fn=DualSelfAttention(out_proj, qkv)))
    <model>.cross_ffn_list.0       = Residual(fn=PreNorm(norm, fn=FeedForward(net=[Linear, GELU, Dropout, Linear, Dropout])))
(ClsWiseTransformer.py:7-55, FusionClsWiseTransformer.py:8-54, SelfAttention.py:50-102, ResidualNorm.py:4-47).

ClsWiseFormer.encode does not call these modules: it hands their parameters to the whole-coupler Functions of cwf.coupler
(selection + coupler + scatter of all three sub-regions per launch).  The two model classes keep the reference's forward()
signatures for callers that use a coupler on its own; those run the SAME block launches (cwf.coupler.IntraCouplerBlockFn /
FusionBlockFn: paired LayerNorm, one q|k|v GEMM, one-launch attention, fused out_proj / FFN epilogues).  The inner wrapper modules
(Residual, PreNormDrop, PreNorm, FeedForward, DualSelfAttention) are parameter holders only.
"""
import torch.nn as nn

from cwf import coupler as CP


class DualSelfAttention(nn.Module):
    def __init__(self, hidden_size, num_heads, dropout_rate=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.out_proj = nn.Linear(hidden_size, hidden_size)
        self.qkv = nn.Linear(hidden_size, hidden_size * 3, bias=False)
        self.dropout_rate = dropout_rate
        self.hidden = hidden_size


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


def _block_cfg(model):
    pre = model.cross_attention_list[0].fn
    ff = model.cross_ffn_list[0].fn.fn
    return CP.CouplerConfig(pre.fn.num_heads, 0, model.training, 0.0, pre.fn.dropout_rate, pre.dropout_rate, ff.dropout_rate)


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
        return CP.IntraCouplerBlockFn.apply(_block_cfg(self), edge_fea, se_fea_supple, semantic_fea, supple_fea_edge,
                                            *CP.transformer_params(self))


class FusionClsWiseTransformerModel(nn.Module):
    """Mutual Cross-region Coupler: self-attention + FFN on the fused tokens (FusionClsWiseTransformer.py:43-54)."""

    def __init__(self, depth, heads, mlp_dim, dropout_rate=0.1, attn_dropout_rate=0.1):
        super().__init__()
        self.depth = depth
        _make_lists(self, depth, heads, mlp_dim, dropout_rate, attn_dropout_rate)

    def forward(self, fusion_semantic):
        return CP.FusionBlockFn.apply(_block_cfg(self), fusion_semantic, *CP.transformer_params(self))
