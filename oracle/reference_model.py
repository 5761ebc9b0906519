"""ORACLE (test infrastructure, never shipped in the product path).

A plain-PyTorch, CPU, fp32 restatement of the reference hot path: the ClsWiseFormer
forward (U-Net encoder, Anatomy-induced Region Decoupler, Edge-supported Intra-region
Coupler, Mutual Cross-region Coupler, decoder, sub-region / edge heads) and the
Dice + weighted-CE losses.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file.

It is written functionally over a reference-layout ``state_dict`` (the 222 tensors of
SURVEY.md Appendix B) so that the same weights can be pushed through (a) the real
reference imported from /root/reference (only inside ``oracle/make_golden.py``, in the
build container), (b) this restatement and (c) the HIP product path.

Pinning: ``oracle/make_golden.py`` checks this file against the imported reference on
identical weights/inputs (forward outputs, five losses, parameter gradients) and writes
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` re-checks this file against those
fixtures on every run.

Each function cites the reference file:line it follows (paths relative to the
reference repo root).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

REGIONS = ("01", "02", "04")      # sub-region keys, cls_wise_former.py:88-127
REGION_K = ("1", "2", "4")        # suffixes of conv_semantic_k / conv_mid_fea_k
TOP_NUM = 128                     # cls_wise_former.py:80
EMBED = 512                       # 128 ch * (2*2*1) window = 32 ch * (4*2*2) window
HEADS = 8                         # cls_wise_former.py:770
SEM_PATCH = (2, 2, 1)             # cls_wise_former.py:77
EDGE_PATCH = (4, 2, 2)            # cls_wise_former.py:85
IN_EPS = 1e-5                     # nn.InstanceNorm3d default
LN_EPS = 1e-5                     # nn.LayerNorm default


# --------------------------------------------------------------------------------------
# window <-> token reshapes                                     cls_wise_former.py:15-39
# --------------------------------------------------------------------------------------
def convert_dim(fea: torch.Tensor, patch: Sequence[int]) -> torch.Tensor:
    """[B,C,D,H,W] -> [B, (D/p0)(H/p1)(W/p2), C*p0*p1*p2]   (cls_wise_former.py:15-23)."""
    b, c, d, h, w = fea.shape
    p0, p1, p2 = patch
    t = fea.reshape(b, c, d // p0, p0, h // p1, p1, w // p2, p2)
    t = t.permute(0, 2, 4, 6, 1, 3, 5, 7)
    return t.reshape(b, (d // p0) * (h // p1) * (w // p2), c * p0 * p1 * p2)


def split_dim(tok: torch.Tensor, channels: int, size: Sequence[int], patch: Sequence[int]) -> torch.Tensor:
    """Inverse of convert_dim                                (cls_wise_former.py:26-39)."""
    b = tok.shape[0]
    p0, p1, p2 = patch
    d, h, w = size
    t = tok.reshape(b, d // p0, h // p1, w // p2, channels, p0, p1, p2)
    t = t.permute(0, 4, 1, 5, 2, 6, 3, 7)
    return t.reshape(b, channels, d, h, w)


def pe_row0(dim: int = EMBED) -> torch.Tensor:
    """Row 0 of ExtendFixedPositionalEncoding.pe: sin(0)=0 on even, cos(0)=1 on odd
    channels.  The reference adds ``pe[:x.size(0)]`` with x = [B=1,128,512], i.e. this one
    row to every token (PositionalEncoding.py:17-22)."""
    r = torch.zeros(dim)
    r[1::2] = 1.0
    return r


def fixed_pe_table(dim: int = EMBED, max_len: int = 1024) -> torch.Tensor:
    """The full [max_len,1,dim] buffer kept for checkpoint compatibility
    (PositionalEncoding.py:9-17)."""
    pe = torch.zeros(max_len, dim)
    pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2).float() * (-torch.log(torch.tensor(10000.0)) / dim))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(0).transpose(0, 1).contiguous()


# --------------------------------------------------------------------------------------
# small building blocks
# --------------------------------------------------------------------------------------
def _conv(p, name, x, stride=1, padding=1):
    return F.conv3d(x, p[name + ".weight"], p[name + ".bias"], stride=stride, padding=padding)


def _inorm(x):
    return F.instance_norm(x, eps=IN_EPS)


def en_block(p, pre, x):
    """Pre-activation residual block  (Unet_skipconnection.py:36-57)."""
    h = _conv(p, pre + ".conv1", F.relu(_inorm(x)))
    h = _conv(p, pre + ".conv2", F.relu(_inorm(h)))
    return h + x


def post_block(p, pre, x):
    """conv-IN-LeakyReLU x2 + residual: EnBlock2 / DeBlock (cls_wise_former.py:691-754)."""
    h = F.leaky_relu(_inorm(_conv(p, pre + ".conv1", x)), 0.01)
    h = F.leaky_relu(_inorm(_conv(p, pre + ".conv2", h)), 0.01)
    return h + x


def de_up_cat(p, pre, x, prev):
    """1x1 conv, ConvTranspose k2 s2, concat(prev, y), 1x1 conv (cls_wise_former.py:716-729)."""
    x1 = F.conv3d(x, p[pre + ".conv1.weight"], p[pre + ".conv1.bias"])
    y = F.conv_transpose3d(x1, p[pre + ".conv2.weight"], p[pre + ".conv2.bias"], stride=2)
    y = torch.cat((prev, y), dim=1)
    return F.conv3d(y, p[pre + ".conv3.weight"], p[pre + ".conv3.bias"])


def unet_encoder(p, x, stem_keep: Optional[torch.Tensor] = None):
    """Unet.forward (Unet_skipconnection.py:114-144).  ``stem_keep`` is an optional
    [B,16] channel keep-mask already scaled by 1/(1-p) standing for the always-on
    F.dropout3d of InitConv (:29-33); None = dropout disabled."""
    u = "Unet_list."
    x = _conv(p, u + "InitConv.conv", x)
    if stem_keep is not None:
        x = x * stem_keep[:, :, None, None, None]
    x1 = en_block(p, u + "EnBlock1_1", en_block(p, u + "EnBlock1", x))
    x = _conv(p, u + "EnDown1.conv", x1, stride=2)
    x2 = en_block(p, u + "EnBlock2_2", en_block(p, u + "EnBlock2_1", x))
    x = _conv(p, u + "EnDown2.conv", x2, stride=2)
    x3 = en_block(p, u + "EnBlock3_2", en_block(p, u + "EnBlock3_1", x))
    x = _conv(p, u + "EnDown3.conv", x3, stride=2)
    x = en_block(p, u + "EnBlock4_2", en_block(p, u + "EnBlock4_1", x))
    x = _conv(p, u + "EnDown_4.conv", x, stride=1)
    return x1, x2, x3, x


def layer_norm(p, pre, x):
    return F.layer_norm(x, (x.shape[-1],), p[pre + ".weight"], p[pre + ".bias"], LN_EPS)


def dual_attention(p, pre, x, x2):
    """DualSelfAttention.forward (SelfAttention.py:74-102): q from x, k/v from x2 through
    ONE bias-free qkv weight; 8 heads x 64; softmax(q k^T / 8); out_proj.  The reference
    evaluates the full 1536-wide projection of both inputs and discards 1/3 resp. 2/3;
    only the used rows are evaluated here (identical result)."""
    wqkv = p[pre + ".qkv.weight"]
    e = wqkv.shape[1]
    b, n, _ = x.shape
    n2 = x2.shape[1]
    hd = e // HEADS
    q = F.linear(x, wqkv[:e]).reshape(b, n, HEADS, hd).permute(0, 2, 1, 3)
    k = F.linear(x2, wqkv[e:2 * e]).reshape(b, n2, HEADS, hd).permute(0, 2, 1, 3)
    v = F.linear(x2, wqkv[2 * e:]).reshape(b, n2, HEADS, hd).permute(0, 2, 1, 3)
    att = (torch.einsum("bhxd,bhyd->bhxy", q, k) * (hd ** -0.5)).softmax(dim=-1)
    o = torch.einsum("bhxy,bhyd->bhxd", att, v).permute(0, 2, 1, 3).reshape(b, n, e)
    return F.linear(o, p[pre + ".out_proj.weight"], p[pre + ".out_proj.bias"])


def cross_attention(p, tpre, x, x2):
    """Residual(PreNormDrop(DualSelfAttention)) (ResidualNorm.py:4-32), dropout off."""
    a = tpre + ".cross_attention_list.0.fn"
    return dual_attention(p, a + ".fn", layer_norm(p, a + ".norm", x), layer_norm(p, a + ".norm2", x2)) + x


def ffn(p, tpre, x):
    """Residual(PreNorm(FeedForward)) (ResidualNorm.py:13-20,35-47), dropout off."""
    f = tpre + ".cross_ffn_list.0.fn"
    h = layer_norm(p, f + ".norm", x)
    h = F.gelu(F.linear(h, p[f + ".fn.net.0.weight"], p[f + ".fn.net.0.bias"]))
    h = F.linear(h, p[f + ".fn.net.3.weight"], p[f + ".fn.net.3.bias"])
    return h + x


def intra_region_coupler(p, tpre, edge_seq, sem_supp, sem_seq, edge_supp):
    """TwoClsWiseTransformerModel.forward (ClsWiseTransformer.py:41-55)."""
    a = cross_attention(p, tpre, edge_seq, sem_supp)
    b = cross_attention(p, tpre, sem_seq, edge_supp)
    re = cross_attention(p, tpre, a, b)
    rs = cross_attention(p, tpre, b, a)
    return ffn(p, tpre, torch.cat((re, rs), dim=1))


def cross_region_coupler(p, tpre, seq):
    """FusionClsWiseTransformerModel.forward (FusionClsWiseTransformer.py:43-54)."""
    return ffn(p, tpre, cross_attention(p, tpre, seq, seq))


def select_tokens(token, feats, top_num=TOP_NUM, forced_index=None):
    """score = token @ feats^T -> top-k -> gather -> + pe[0] -> prepend class token
    (cls_wise_former.py:345-350 and 12 siblings).  feats [B,N,E], token [1,1,E].
    Returns (seq [B,k+1,E], index [B,k]).  Per-sample selection."""
    b = feats.shape[0]
    score = torch.einsum("e,bne->bn", token[0, 0], feats)
    if forced_index is None:
        k = min(top_num, feats.shape[1])
        index = score.topk(k, dim=1, largest=True, sorted=True).indices
    else:
        index = forced_index
    rows = torch.gather(feats, 1, index[:, :, None].expand(-1, -1, feats.shape[2]))
    rows = rows + pe_row0(feats.shape[2]).to(feats)
    seq = torch.cat((token.expand(b, -1, -1), rows), dim=1)
    return seq, index


def scatter_rows(feats, index, rows):
    """``X[idx[j], :] = rows[j, :]`` per sample (cls_wise_former.py:463-468; the
    reference's fix_index.txt row table makes scatter_ a row scatter).
    Functional (out of place) so that autograd sees the same dependency structure as the
    reference's in-place scatter on a conv output."""
    idx = index[:, :, None].expand(-1, -1, feats.shape[2])
    return feats.scatter(1, idx, rows)


def sub_region_heads(p, pre, names, feats, scale):
    """SuperviseLabel / EdgeSuperviseLabel.forward (SuperviseLabel.py:58-81,
    EdgeSuperviseLabel.py:56-76): conv3 -> conv3 -> trilinear xscale -> softmax."""
    out = {}
    for key, k, f in zip(REGIONS, REGION_K, feats):
        h = _conv(p, pre + names[0] + k, f)
        h = _conv(p, pre + names[1] + k, h)
        h = F.interpolate(h, scale_factor=scale, mode="trilinear", align_corners=False)
        out[key] = h.softmax(dim=1)
    return out


def decoder(p, x1, x2, x3, x, return_logits=False):
    """Decoder.forward (cls_wise_former.py:644-664)."""
    d = "decoder."
    x8 = F.conv3d(x, p[d + "down_channel.weight"], p[d + "down_channel.bias"])
    x8 = post_block(p, d + "Enblock8_2", post_block(p, d + "Enblock8_1", x8))
    y4 = de_up_cat(p, d + "DeUp4", x8, x3)
    y4 = post_block(p, d + "DeBlock4_1", post_block(p, d + "DeBlock4", y4))
    y3 = de_up_cat(p, d + "DeUp3", y4, x2)
    y3 = post_block(p, d + "DeBlock3_1", post_block(p, d + "DeBlock3", y3))
    y2 = de_up_cat(p, d + "DeUp2", y3, x1)
    y2 = post_block(p, d + "DeBlock2_1", post_block(p, d + "DeBlock2", y2))
    logits = F.conv3d(y2, p[d + "endconv.weight"], p[d + "endconv.bias"])
    prob = logits.softmax(dim=1)
    return (prob, logits) if return_logits else prob


# --------------------------------------------------------------------------------------
# full forward                                       cls_wise_former.py:280-592
# --------------------------------------------------------------------------------------
def forward(p: Dict[str, torch.Tensor], x: torch.Tensor, stem_keep=None, forced_index=None,
            return_aux: bool = False):
    """ClsWiseFormer.forward.  ``x`` [B,4,D,H,W]; every sample is processed with the
    reference's B=1 semantics.  Returns (prob, sup, edge, mid_sup, mid_edge);
    with ``return_aux`` also a dict of intermediates (top-k index sets, logits, bottleneck).
    ``forced_index``: optional dict name->[B,k] int64 of teacher-forced selections."""
    aux = {}
    fi = forced_index or {}
    x1, x2, x3, x4 = unet_encoder(p, x, stem_keep)

    # ---- edge decoupler  (cls_wise_former.py:284-296)
    x2d = _conv(p, "conv_64_to_32", x2, stride=2)
    x23 = torch.cat((x2d, x3), dim=1)
    edge_f = [F.leaky_relu(_inorm(_conv(p, "conv_mid_fea_" + k, x23)), 0.01) for k in REGION_K]
    # ---- Anatomy-induced Region Decoupler (:314-324)
    sem_f = [F.leaky_relu(_inorm(_conv(p, "conv_semantic_" + k, x4)), 0.01) for k in REGION_K]

    # ---- mid supervision (:332-333)
    mid_sup = sub_region_heads(p, "mid_supervise_label.", ("supervise_label_", "down_label_"), sem_f, 8)
    mid_edge = sub_region_heads(p, "mid_edge_supervise_label.",
                                ("edge_supervise_label_", "edge_down_label_"), edge_f, 4)

    sem_size = tuple(sem_f[0].shape[2:])
    edge_size = tuple(edge_f[0].shape[2:])
    sup_sem, sup_edge, sem_tokens, sem_after = [], [], [], []
    for r, ef, sf in zip(REGIONS, edge_f, sem_f):
        E = convert_dim(ef, EDGE_PATCH)                       # [B,Ne,512]   :341
        S = convert_dim(sf, SEM_PATCH)                        # [B,Ns,512]   :342
        e_tok, s_tok = p["e_token_" + r], p["s_token_" + r]
        edge_seq, idx_e = select_tokens(e_tok, E, forced_index=fi.get(r + "_edge"))         # :345-350
        sem_supp, idx_es = select_tokens(e_tok, S, forced_index=fi.get(r + "_sem_supp"))    # :352-357 (scores by e_tok, prepends s_tok)
        sem_supp = torch.cat((s_tok.expand(sem_supp.shape[0], -1, -1), sem_supp[:, 1:]), dim=1)
        sem_seq, idx_s = select_tokens(s_tok, S, forced_index=fi.get(r + "_sem"))           # :360-367
        edge_supp, idx_se = select_tokens(s_tok, E, forced_index=fi.get(r + "_edge_supp"))  # :370-376 (scores by s_tok, prepends e_tok)
        edge_supp = torch.cat((e_tok.expand(edge_supp.shape[0], -1, -1), edge_supp[:, 1:]), dim=1)
        aux.update({r + "_edge": idx_e, r + "_sem_supp": idx_es, r + "_sem": idx_s, r + "_edge_supp": idx_se})

        res = intra_region_coupler(p, "transformer_" + r, edge_seq, sem_supp, sem_seq, edge_supp)  # :379
        n1 = edge_seq.shape[1]
        res_e, res_s = res[:, :n1], res[:, n1:2 * n1]
        E2 = scatter_rows(E, idx_e, res_e[:, 1:])             # :467
        S2 = scatter_rows(S, idx_s, res_s[:, 1:])             # :477
        sup_edge.append(split_dim(res_e[:, 0:1] * E2, ef.shape[1], edge_size, EDGE_PATCH))  # :481-482
        sup_sem.append(split_dim(res_s[:, 0:1] * S2, sf.shape[1], sem_size, SEM_PATCH))     # :484-485
        sem_tokens.append(res_s[:, 0:1])
        sem_after.append(S2)

    sup = sub_region_heads(p, "supervise_label.", ("supervise_label_", "down_label_"), sup_sem, 8)        # :545
    edge = sub_region_heads(p, "edge_supervise_label.",
                            ("edge_supervise_label_", "edge_down_label_"), sup_edge, 4)                  # :546

    # ---- Mutual Cross-region Coupler (:549-579)
    f_tok = sem_tokens[0] + sem_tokens[1] + sem_tokens[2]                       # [B,1,512]
    f_feat = sem_after[0] + sem_after[1] + sem_after[2]                         # post-scatter, un-gated
    score = torch.einsum("be,bne->bn", f_tok[:, 0], f_feat)
    if "fusion" in fi:
        f_idx = fi["fusion"]
    else:
        f_idx = score.topk(min(TOP_NUM, f_feat.shape[1]), dim=1, largest=True, sorted=True).indices
    aux["fusion"] = f_idx
    rows = torch.gather(f_feat, 1, f_idx[:, :, None].expand(-1, -1, f_feat.shape[2])) + pe_row0(f_feat.shape[2]).to(f_feat)
    f_seq = torch.cat((f_tok, rows), dim=1)
    f_res = cross_region_coupler(p, "fusion_transformer_1_2_4", f_seq)
    fused = f_res[:, 0:1] * scatter_rows(f_feat, f_idx, f_res[:, 1:])
    xb = split_dim(fused, sem_f[0].shape[1], sem_size, SEM_PATCH)
    xb = _conv(p, "sum_fusion", xb)                                            # :582
    prob, logits = decoder(p, x1, x2, x3, xb, return_logits=True)
    if return_aux:
        aux["logits"] = logits
        aux["bottleneck"] = xb
        return (prob, sup, edge, mid_sup, mid_edge), aux
    return prob, sup, edge, mid_sup, mid_edge


# --------------------------------------------------------------------------------------
# losses                                     utils/tools.py:8-34,112-231; criterions.py:49-62
# --------------------------------------------------------------------------------------
def dice_loss(prob, onehot, num_cls, eps=1e-7):
    """tools.dice_loss (tools.py:8-18): batch-global sums per class."""
    t = onehot.float()
    dice = 0.0
    for i in range(num_cls):
        num = torch.sum(prob[:, i] * t[:, i])
        dice = dice + 2.0 * num / (torch.sum(prob[:, i]) + torch.sum(t[:, i]) + eps)
    return 1.0 - dice / num_cls


def softmax_weighted_loss(prob, onehot, num_cls):
    """tools.softmax_weighted_loss (tools.py:21-34): per-sample class-frequency weight
    1 - sum(t_c)/sum(t); -w t log(clamp(p, 0.005, 1)); mean over B*D*H*W."""
    t = onehot.float()
    tot = torch.sum(t, (1, 2, 3, 4))
    loss = 0.0
    for i in range(num_cls):
        w = 1.0 - torch.sum(t[:, i], (1, 2, 3)) / tot
        loss = loss - w.reshape(-1, 1, 1, 1) * t[:, i] * torch.log(torch.clamp(prob[:, i], min=0.005, max=1))
    return torch.mean(loss)


def _dice_ce(prob, labels, num_cls):
    oh = F.one_hot(labels, num_cls).permute(0, 4, 1, 2, 3).contiguous()
    return dice_loss(prob, oh, num_cls) + softmax_weighted_loss(prob, oh, num_cls)


def softmax_dice(prob, target):
    """criterions.softmax_dice (criterions.py:49-62)."""
    return _dice_ce(prob, target, 4)


def get_separate_loss(out, target):
    """tools.get_separate_loss (tools.py:112-162): binary problems {target==k}, k=1,2,3."""
    total = 0.0
    for key, k in zip(REGIONS, (1, 2, 3)):
        total = total + _dice_ce(out[key], (target == k).long(), 2)
    return total


EDGE_SETS = {"01": (1, 5, 6, 7), "02": (2, 5, 6, 8), "04": (4, 5, 7, 8)}


def get_edge_separate_loss(out, edge):
    """tools.get_edge_separate_loss (tools.py:165-231): edge code sets E1/E2/E4."""
    total = 0.0
    for key in REGIONS:
        m = torch.zeros_like(edge, dtype=torch.bool)
        for c in EDGE_SETS[key]:
            m |= edge == c
        total = total + _dice_ce(out[key], m.long(), 2)
    return total


def total_loss(outputs, target, edge):
    """train_no_amp.py:205-211 (all weights 1.0).  Returns (total, [five parts])."""
    parts = [softmax_dice(outputs[0], target), get_separate_loss(outputs[1], target),
             get_edge_separate_loss(outputs[2], edge), get_separate_loss(outputs[3], target),
             get_edge_separate_loss(outputs[4], edge)]
    return sum(parts), parts


# --------------------------------------------------------------------------------------
# eval helpers                                tools.py:44-47,89-109; predict_overlap.py:31-58
# --------------------------------------------------------------------------------------
def softmax_output_dice(output, target, eps=1e-8):
    """tools.softmax_output_dice (tools.py:89-109) on integer label maps -> [WT,TC,ET]."""
    def ds(o, t):
        o, t = o.double(), t.double()
        return float((2 * (o * t).sum() + eps) / (o.sum() + t.sum() + eps))
    return [ds(output > 0, target > 0),
            ds((output == 1) | (output == 3), (target == 1) | (target == 3)),
            ds(output == 3, target == 3)]


def softmax_miou_score(output, target, eps=1e-8):
    """tools.softmax_mIOU_score (tools.py:50-61) on integer label maps -> IoU of classes 1, 2, 3."""
    def iou(o, t):
        return float(((o & t).double().sum() + eps) / ((o | t).double().sum() + eps))
    return [iou(output == c, target == c) for c in (1, 2, 3)]


def tailor_and_concat(x, fwd):
    """predict_overlap.tailor_and_concat (predict_overlap.py:31-58) incl. its D-axis
    stitch offset; ``fwd(window)->prob``; x [B,4,240,240,>=155]."""
    wins = [(0, 0, 0), (0, 112, 0), (112, 0, 0), (112, 112, 0), (0, 0, 27), (0, 112, 27), (112, 0, 27), (112, 112, 27)]
    t = [fwd(x[..., a:a + 128, b:b + 128, c:c + 128]) for a, b, c in wins]
    y = x.clone()
    y[..., :128, :128, :128] = t[0]
    y[..., :128, 128:240, :128] = t[1][..., :, 16:128, :]
    y[..., 128:240, :128, :128] = t[2][..., 16:128, :, :]
    y[..., 128:240, 128:240, :128] = t[3][..., 16:128, 16:128, :]
    y[..., :128, :128, 128:155] = t[4][..., 96:123]
    y[..., :128, 128:240, 128:155] = t[5][..., :, 16:128, 96:123]
    y[..., 128:240, :128, 128:155] = t[6][..., 16:128, :, 96:123]
    y[..., 128:240, 128:240, 128:155] = t[7][..., 16:128, 16:128, 96:123]
    return y[..., :155]


def flip_tta(x, fwd):
    """8-flip test-time augmentation exactly as predict_simple.py:333-349 writes it: eight sequential forwards, softmax
    re-applied to each (already normalised) output, flipped back, summed in this order, divided by 8.  ``fwd(x)->prob``."""
    logit = F.softmax(fwd(x), 1)
    for dims in ((2,), (3,), (4,), (2, 3), (2, 4), (3, 4), (2, 3, 4)):
        logit = logit + F.softmax(fwd(x.flip(dims=dims)).flip(dims=dims), 1)
    return logit / 8


# --------------------------------------------------------------------------------------
# parameter inventory and a reference-style train step
# --------------------------------------------------------------------------------------
def param_shapes() -> "List[Tuple[str, Tuple[int, ...], bool]]":
    """(name, shape, is_buffer) of the 222 state_dict tensors in registration order."""
    out = []
    for r in REGIONS:
        out += [("e_token_" + r, (1, 1, EMBED), False), ("s_token_" + r, (1, 1, EMBED), False)]
    for r in REGIONS:
        out.append(("label_%s_position_encoding.pe" % r, (1024, 1, EMBED), True))

    def tr(pre):
        a = pre + ".cross_attention_list.0.fn"
        f = pre + ".cross_ffn_list.0.fn"
        return [(a + ".norm.weight", (EMBED,), False), (a + ".norm.bias", (EMBED,), False),
                (a + ".norm2.weight", (EMBED,), False), (a + ".norm2.bias", (EMBED,), False),
                (a + ".fn.out_proj.weight", (EMBED, EMBED), False), (a + ".fn.out_proj.bias", (EMBED,), False),
                (a + ".fn.qkv.weight", (3 * EMBED, EMBED), False),
                (f + ".norm.weight", (EMBED,), False), (f + ".norm.bias", (EMBED,), False),
                (f + ".fn.net.0.weight", (EMBED, EMBED), False), (f + ".fn.net.0.bias", (EMBED,), False),
                (f + ".fn.net.3.weight", (EMBED, EMBED), False), (f + ".fn.net.3.bias", (EMBED,), False)]
    for r in REGIONS:
        out += tr("transformer_" + r)
    out.append(("fusion_label_pos.pe", (1024, 1, EMBED), True))
    out += tr("fusion_transformer_1_2_4")

    def conv(name, co, ci, k=3):
        return [(name + ".weight", (co, ci, k, k, k), False), (name + ".bias", (co,), False)]
    for k in REGION_K:
        out += conv("conv_semantic_" + k, 128, 256)
    for k in REGION_K:
        out += conv("conv_mid_fea_" + k, 32, 96)
    u = "Unet_list."
    out += conv(u + "InitConv.conv", 16, 4)
    for b, c in (("EnBlock1", 16), ("EnBlock1_1", 16)):
        out += conv(u + b + ".conv1", c, c) + conv(u + b + ".conv2", c, c)
    out += conv(u + "EnDown1.conv", 32, 16)
    for b in ("EnBlock2_1", "EnBlock2_2"):
        out += conv(u + b + ".conv1", 32, 32) + conv(u + b + ".conv2", 32, 32)
    out += conv(u + "EnDown2.conv", 64, 32)
    for b in ("EnBlock3_1", "EnBlock3_2"):
        out += conv(u + b + ".conv1", 64, 64) + conv(u + b + ".conv2", 64, 64)
    out += conv(u + "EnDown3.conv", 128, 64)
    for b in ("EnBlock4_1", "EnBlock4_2"):
        out += conv(u + b + ".conv1", 128, 128) + conv(u + b + ".conv2", 128, 128)
    out += conv(u + "EnDown_4.conv", 256, 128)
    d = "decoder."
    out += conv(d + "down_channel", 128, 256, 1)
    for b in ("Enblock8_1", "Enblock8_2"):
        out += conv(d + b + ".conv1", 128, 128) + conv(d + b + ".conv2", 128, 128)
    for up, blocks, c in (("DeUp4", ("DeBlock4", "DeBlock4_1"), 64), ("DeUp3", ("DeBlock3", "DeBlock3_1"), 32),
                          ("DeUp2", ("DeBlock2", "DeBlock2_1"), 16)):
        out += conv(d + up + ".conv1", c, 2 * c, 1) + conv(d + up + ".conv2", c, c, 2) + conv(d + up + ".conv3", c, 2 * c, 1)
        for b in blocks:
            out += conv(d + b + ".conv1", c, c) + conv(d + b + ".conv2", c, c)
    out += conv(d + "endconv", 4, 16, 1)
    for pre in ("supervise_label.",):
        for k in REGION_K:
            out += conv(pre + "supervise_label_" + k, 32, 128) + conv(pre + "down_label_" + k, 2, 32)
    for pre in ("edge_supervise_label.",):
        for k in REGION_K:
            out += conv(pre + "edge_supervise_label_" + k, 8, 32) + conv(pre + "edge_down_label_" + k, 2, 8)
    for k in REGION_K:
        out += conv("mid_supervise_label.supervise_label_" + k, 32, 128) + conv("mid_supervise_label.down_label_" + k, 2, 32)
    for k in REGION_K:
        out += conv("mid_edge_supervise_label.edge_supervise_label_" + k, 8, 32) + \
            conv("mid_edge_supervise_label.edge_down_label_" + k, 2, 8)
    out += conv("sum_fusion", 256, 128) + conv("conv_64_to_32", 32, 32)
    return out


def poly_lr(init_lr, epoch, max_epoch, power=0.9):
    """train_no_amp.adjust_learning_rate (train_no_amp.py:270-273)."""
    return round(init_lr * (1 - epoch / max_epoch) ** power, 8)


class CpuTrainer:
    """The reference training step (train_no_amp.py:183-239) driven on CPU tensors:
    forward, five losses, backward, Adam(lr 2e-4, wd 1e-5, amsgrad).  Used as the CPU
    baseline ("port") and by the parity tests."""

    def __init__(self, state: Dict[str, torch.Tensor], lr=2e-4, weight_decay=1e-5, amsgrad=True):
        self.p = {k: v.clone().requires_grad_(not k.endswith(".pe")) for k, v in state.items()}
        self.params = [v for k, v in self.p.items() if v.requires_grad]
        self.opt = torch.optim.Adam(self.params, lr=lr, weight_decay=weight_decay, amsgrad=amsgrad)

    def step(self, x, target, edge, stem_keep=None):
        outs = forward(self.p, x, stem_keep=stem_keep)
        loss, parts = total_loss(outs, target, edge)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return float(loss), [float(v) for v in parts]
