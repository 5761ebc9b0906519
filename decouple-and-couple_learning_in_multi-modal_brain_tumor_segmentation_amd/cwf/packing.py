"""Index maps between the reference parameter layouts (nn.Conv3d.weight [Cout,Cin,k,k,k],
nn.ConvTranspose3d.weight [Cin,Cout,2,2,2]; cls_wise_former.py:157-273,623-642) and the layouts the MFMA
kernels consume / produce (csrc/conv_mfma.hip, csrc/wgrad_mfma.hip).

Packed weights (input of cwf_conv_mfma):   [class][ci_chunk16][tap][co_tile16][lane64][4]
    element (lane, j): ci = chunk*16 + (lane>>4)*4 + j,  co = tile*16 + (lane & 15)
Gradient slab (output of cwf_wgrad_mfma):  [class][ci_chunk16][co_group][tap slot 0..ntaps][tile in group][lane64][4]
    element (lane, i): ci = chunk*16 + (lane>>4)*4 + i,  co = (group*CG + tile)*16 + (lane & 15); slot ntaps = bias row.

The maps are int32 arrays: packed[i] = W.flat[map[i]] (or 0 where map < 0), dW.flat[e] = sum_splits slab[map[e]].
Tap / class enumeration must match cwf_build_geom() in csrc/conv_mfma.hip.
"""
from __future__ import annotations

import numpy as np

CONV3_S1, CONV3_S2, CONV1, CONVT2, CONV3_S2_DGRAD, CONVT2_DGRAD = range(6)


def _cdiv(a, b):
    return (a + b - 1) // b


def _pack_map(ncls_taps, cin, cout, src_index):
    """Generic packer.  ncls_taps: list (per class) of tap descriptors; src_index(cls, tap_desc, ci, co) -> flat index
    (vectorised over ci [16*nchunks,1] and co [1,16*ntiles]) or -1."""
    nch, nt = _cdiv(cin, 16), _cdiv(cout, 16)
    ci = np.arange(nch * 16).reshape(-1, 1)
    co = np.arange(nt * 16).reshape(1, -1)
    valid = (ci < cin) & (co < cout)
    blocks = []
    for cls, taps in enumerate(ncls_taps):
        per_cls = np.full((nch, len(taps), nt, 64, 4), -1, dtype=np.int64)
        for t, tap in enumerate(taps):
            idx = np.where(valid, src_index(cls, tap, np.minimum(ci, cin - 1), np.minimum(co, cout - 1)), -1)  # [nch*16, nt*16]
            # -> [chunk][tile][lane = (m//4)*16 + c][j = m%4]
            a = idx.reshape(nch, 4, 4, nt, 16)            # chunk, m//4, m%4, tile, c
            a = a.transpose(0, 3, 1, 4, 2)                 # chunk, tile, m//4, c, m%4
            per_cls[:, t] = a.reshape(nch, nt, 64, 4)
        blocks.append(per_cls.reshape(-1))
    return np.concatenate(blocks).astype(np.int32)


def _taps3():
    return [(kd, kh, kw) for kd in range(3) for kh in range(3) for kw in range(3)]


def _taps3_order16():
    """Tap order of the split-bf16 kernels for 3x3x3 stride-1 layers with <= 16 input and <= 16 output channels (conv16 and
    its small-volume stand-in): the two taps of a K = 32 MFMA step must differ by an offset that does not depend on the row's
    position in conv16's ring of LDS rows, so pairs are (kw0, kw1) of one (kd, kh) row, then (kd0, kd1) of the kw = 2 taps,
    then the two remaining row neighbours, then the single last tap.  Mirrored by kOrder16 in csrc/conv_bf16.hip."""
    order = []
    for kd in range(3):
        for kh in range(3):
            order += [(kd, kh, 0), (kd, kh, 1)]
    for kh in range(3):
        order += [(0, kh, 2), (1, kh, 2)]
    order += [(2, 0, 2), (2, 1, 2), (2, 2, 2)]
    assert sorted(order) == sorted(_taps3())
    return order


def _taps3_for(op, cin, cout):
    return _taps3_order16() if (op == CONV3_S1 and cin <= 16 and cout <= 16) else _taps3()


def _s2_dgrad_classes():
    out = []
    for c in range(8):
        p = ((c >> 2) & 1, (c >> 1) & 1, c & 1)
        taps = []
        for a0 in range(p[0] + 1):
            for a1 in range(p[1] + 1):
                for a2 in range(p[2] + 1):
                    k = tuple(1 if pp == 0 else (2 if aa == 0 else 0) for pp, aa in zip(p, (a0, a1, a2)))
                    taps.append(k)
        out.append(taps)
    return out


def fwd_map(op, cin, cout):
    """Packed-weight map for the forward op.  Weight ref layouts: conv [cout,cin,k,k,k]; convT [cin,cout,2,2,2]."""
    if op in (CONV3_S1, CONV3_S2):
        return _pack_map([_taps3()], cin, cout, lambda c, k, ci, co: ((co * cin + ci) * 3 + k[0]) * 9 + k[1] * 3 + k[2])
    if op == CONV1:
        return _pack_map([[None]], cin, cout, lambda c, k, ci, co: co * cin + ci)
    if op == CONVT2:
        return _pack_map([[c] for c in range(8)], cin, cout, lambda c, k, ci, co: (ci * cout + co) * 8 + k)
    raise ValueError(op)


def dgrad_op(op):
    return {CONV3_S1: CONV3_S1, CONV1: CONV1, CONV3_S2: CONV3_S2_DGRAD, CONVT2: CONVT2_DGRAD}[op]


def dgrad_map(op, cin, cout, cout_alloc=None):
    """Packed-weight map for the data gradient of forward op (kernel input channels = cout, output = cin).
    cout_alloc > cout: the gradient tensor carries zero-padded channels (2-channel heads live in 4-channel buffers)."""
    ca = cout_alloc or cout

    def guard(f):
        return lambda c, k, ci, co: np.where(ci < cout, f(c, k, np.minimum(ci, cout - 1), co), -1)
    if op == CONV3_S1:     # tap t' at offset (a-1) uses k = 2 - a
        return _pack_map([_taps3()], ca, cin,
                         guard(lambda c, k, ci, co: ((ci * cin + co) * 3 + (2 - k[0])) * 9 + (2 - k[1]) * 3 + (2 - k[2])))
    if op == CONV1:
        return _pack_map([[None]], ca, cin, guard(lambda c, k, ci, co: ci * cin + co))
    if op == CONV3_S2:
        return _pack_map(_s2_dgrad_classes(), ca, cin,
                         guard(lambda c, k, ci, co: ((ci * cin + co) * 3 + k[0]) * 9 + k[1] * 3 + k[2]))
    if op == CONVT2:       # convT weight [cin_t, cout_t, 2,2,2]; dgrad: kernel in = cout_t, out = cin_t, taps = parity
        return _pack_map([list(range(8))], ca, cin, guard(lambda c, k, ci, co: (co * cout + ci) * 8 + k))
    raise ValueError(op)


def wgrad_cg(op, cout):
    nt = _cdiv(cout, 16)
    if op in (CONV3_S1, CONV3_S2):
        return 1 if nt == 1 else 2
    return 1 if nt == 1 else (2 if nt == 2 else 4)


def wgrad_maps(op, cin, cout):
    """(w_map, b_map, slab_floats) for cwf_wgrad_reduce.  b_map is None for CONVT2 (its bias gradient spans 8 classes)."""
    nch, nt = _cdiv(cin, 16), _cdiv(cout, 16)
    cg = wgrad_cg(op, cout)
    ng = _cdiv(nt, cg)
    ntaps = 27 if op in (CONV3_S1, CONV3_S2) else 1
    ncls = 8 if op == CONVT2 else 1
    cls_blocks = nch * ng * (ntaps + 1) * cg

    def slab_index(cls, t, ci, co):
        chunk, m = ci // 16, ci % 16
        tile = co // 16
        grp, j = tile // cg, tile % cg
        lane = (m // 4) * 16 + co % 16
        blk = cls * cls_blocks + ((chunk * ng + grp) * (ntaps + 1) + t) * cg + j
        return blk * 256 + lane * 4 + m % 4

    if op in (CONV3_S1, CONV3_S2):
        co, ci, kd, kh, kw = np.meshgrid(np.arange(cout), np.arange(cin), np.arange(3), np.arange(3), np.arange(3), indexing="ij")
        w_map = slab_index(0, (kd * 3 + kh) * 3 + kw, ci, co)
    elif op == CONV1:
        co, ci = np.meshgrid(np.arange(cout), np.arange(cin), indexing="ij")
        w_map = slab_index(0, 0, ci, co)
    elif op == CONVT2:
        ci, co, p = np.meshgrid(np.arange(cin), np.arange(cout), np.arange(8), indexing="ij")
        w_map = slab_index(p, 0, ci, co)
    else:
        raise ValueError(op)
    b_map = None
    if op != CONVT2:
        b_map = slab_index(0, ntaps, np.zeros(cout, dtype=np.int64), np.arange(cout)).astype(np.int32)
    return w_map.reshape(-1).astype(np.int32), b_map, ncls * cls_blocks * 256


def out_dims(op, d, h, w):
    if op == CONV3_S2:
        return ((d - 1) // 2 + 1, (h - 1) // 2 + 1, (w - 1) // 2 + 1)
    if op == CONVT2:
        return (2 * d, 2 * h, 2 * w)
    return (d, h, w)


def wgrad_inverse_map(op, cin, cout):
    """(inv, has_bias, slab_floats): inv[slab index] = index into dW.flat, or -2 - co for the bias row, or -1 (padding)."""
    w_map, b_map, slab = wgrad_maps(op, cin, cout)
    inv = np.full(slab, -1, dtype=np.int32)
    inv[w_map] = np.arange(w_map.size, dtype=np.int32)
    if b_map is not None:
        inv[b_map] = -2 - np.arange(b_map.size, dtype=np.int32)
    return inv, b_map is not None, slab


# ----------------------------------------------------------------------------------------------------------------
# split-bf16 packing (csrc/conv_bf16.hip): [class][ci_chunk16][tap pair][co_tile16][lane64][8]  (one int32 per bf16 of the
# hi image; the kernel writes hi and lo side by side).  lane = kq*16 + r: tap = 2*pair + (kq>>1), ci = chunk*16 + (kq&1)*8 + j,
# co = tile*16 + r.
# ----------------------------------------------------------------------------------------------------------------
def _pack_map16(ncls_taps, cin, cout, src_index):
    nch, nt = _cdiv(cin, 16), _cdiv(cout, 16)
    ci = np.arange(nch * 16).reshape(-1, 1)
    co = np.arange(nt * 16).reshape(1, -1)
    valid = (ci < cin) & (co < cout)
    blocks = []
    for cls, taps in enumerate(ncls_taps):
        npairs = (len(taps) + 1) // 2
        per_cls = np.full((nch, npairs, nt, 4, 16, 8), -1, dtype=np.int64)       # chunk, pair, tile, kq, r, j
        for t, tap in enumerate(taps):
            idx = np.where(valid, src_index(cls, tap, np.minimum(ci, cin - 1), np.minimum(co, cout - 1)), -1)   # [nch*16, nt*16]
            a = idx.reshape(nch, 2, 8, nt, 16)             # chunk, half (kq&1), j, tile, r
            a = a.transpose(0, 3, 1, 4, 2)                  # chunk, tile, half, r, j
            for half in range(2):
                per_cls[:, t // 2, :, (t % 2) * 2 + half] = a[:, :, half]
        blocks.append(per_cls.reshape(-1))
    return np.concatenate(blocks).astype(np.int32)


def fwd_map16(op, cin, cout):
    if op in (CONV3_S1, CONV3_S2):
        return _pack_map16([_taps3_for(op, cin, cout)], cin, cout, lambda c, k, ci, co: ((co * cin + ci) * 3 + k[0]) * 9 + k[1] * 3 + k[2])
    if op == CONV1:
        return _pack_map16([[None]], cin, cout, lambda c, k, ci, co: co * cin + ci)
    if op == CONVT2:
        return _pack_map16([[c] for c in range(8)], cin, cout, lambda c, k, ci, co: (ci * cout + co) * 8 + k)
    raise ValueError(op)


def dgrad_map16(op, cin, cout, cout_alloc=None):
    ca = cout_alloc or cout

    def guard(f):
        return lambda c, k, ci, co: np.where(ci < cout, f(c, k, np.minimum(ci, cout - 1), co), -1)
    if op == CONV3_S1:
        return _pack_map16([_taps3_for(op, cin, cout)], ca, cin,
                           guard(lambda c, k, ci, co: ((ci * cin + co) * 3 + (2 - k[0])) * 9 + (2 - k[1]) * 3 + (2 - k[2])))
    if op == CONV1:
        return _pack_map16([[None]], ca, cin, guard(lambda c, k, ci, co: ci * cin + co))
    if op == CONV3_S2:
        return _pack_map16(_s2_dgrad_classes(), ca, cin,
                           guard(lambda c, k, ci, co: ((ci * cin + co) * 3 + k[0]) * 9 + k[1] * 3 + k[2]))
    if op == CONVT2:
        return _pack_map16([list(range(8))], ca, cin, guard(lambda c, k, ci, co: (co * cout + ci) * 8 + k))
    raise ValueError(op)
