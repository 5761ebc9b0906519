"""GPU: every HIP kernel family (through the C ABI wrappers of cwf/kernels.py) against the CPU kernel oracle
(oracle/kernel_emul.py) on the same seeded inputs.  Tolerances are for exact-f32 MFMA chains vs oneDNN/ATen fp32:
differences come from summation order only."""
import math

import numpy as np
import pytest
import torch

from cwf import packing as pk
from oracle.kernel_emul import EmulBackend

pytestmark = pytest.mark.gpu
E = EmulBackend()
DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + int(np.prod(shape)) % 9973)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def close(got, ref, rtol=2e-5, atol=None, what=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    atol = atol if atol is not None else rtol * float(ref.abs().max() + 1e-30)
    err = float((got - ref).abs().max())
    assert err <= atol + rtol * float(ref.abs().max()), (what, err, float(ref.abs().max()))


def _packed(spec, w, prec="fp32"):
    """Pack through the product path (WeightPacker -> cwf_gather_batched / cwf_gather_split_bf16)."""
    from cwf import functional as CF, kernels
    packer = CF.WeightPacker()
    packer.add(spec, torch.nn.Parameter(w.to(DEV).contiguous()))
    kernels.set_precision(prec)
    try:
        packer.refresh()
    finally:
        kernels.set_precision("fp32")
    spec._keepalive = packer
    return spec


PREC_TOL = {"fp32": 2e-5, "bf16x3": 2e-4, "bf16": 3e-2}


CONV_CASES = [
    # op, cin, cout, size (D,H,W), batch
    (pk.CONV3_S1, 4, 16, (8, 12, 20), 2),
    (pk.CONV3_S1, 16, 16, (16, 16, 32), 1),
    (pk.CONV3_S1, 16, 16, (36, 30, 40), 2),      # >= 32768 voxels: the persistent conv16 kernel, ragged tiles
    (pk.CONV3_S1, 4, 16, (32, 32, 32), 2),
    (pk.CONV3_S1, 32, 32, (8, 8, 16), 1),
    (pk.CONV3_S1, 96, 32, (8, 8, 8), 1),
    (pk.CONV3_S1, 256, 128, (4, 6, 6), 1),
    (pk.CONV3_S1, 128, 32, (6, 4, 4), 2),
    (pk.CONV3_S1, 32, 2, (8, 8, 8), 1),
    (pk.CONV3_S1, 8, 2, (8, 8, 8), 1),
    (pk.CONV3_S2, 16, 32, (16, 16, 32), 1),
    (pk.CONV3_S2, 64, 128, (8, 8, 8), 2),
    (pk.CONV3_S2, 32, 32, (10, 12, 14), 1),
    (pk.CONV1, 256, 128, (4, 4, 8), 1),
    (pk.CONV1, 32, 16, (8, 8, 32), 1),
    (pk.CONV1, 16, 4, (8, 8, 16), 2),
    (pk.CONVT2, 64, 64, (4, 4, 8), 1),
    (pk.CONVT2, 16, 16, (6, 8, 10), 2),
]


@pytest.mark.parametrize("prec", ["fp32", "bf16x3", "bf16"])
@pytest.mark.parametrize("op,cin,cout,size,n", CONV_CASES)
def test_conv_family_fwd_dgrad_wgrad(hip, op, cin, cout, size, n, prec):
    tol = PREC_TOL[prec]
    from cwf import functional as CF
    torch.manual_seed(0)
    d, h, w_ = size
    x = rnd(n, d, h, w_, cin, seed=1)
    wshape = (cin, cout, 2, 2, 2) if op == pk.CONVT2 else ((cout, cin, 1, 1, 1) if op == pk.CONV1 else (cout, cin, 3, 3, 3))
    w = rnd(*wshape, seed=2, scale=1.0 / math.sqrt(cin * (27 if len(wshape) == 5 and wshape[2] == 3 else 1)))
    b = rnd(cout, seed=3, scale=0.1)
    in_scale = rnd(n, cin, seed=4).abs() + 0.5
    in_shift = rnd(n, cin, seed=5)
    out_scale = (rnd(n, cout, seed=6) > -0.5).float() * 1.25
    spec = _packed(CF.ConvSpec(op, cin, cout), w, prec)
    wf, wd = (spec.wpk_f, spec.wpk_d) if prec == "fp32" else (spec.wpk16_f, spec.wpk16_d)
    do, ho, wo = pk.out_dims(op, d, h, w_)
    res = rnd(n, do, ho, wo, cout, seed=7) if cout % 4 == 0 else None

    # ---- forward with every epilogue / prologue feature
    st_ref = E.new_stats(n, cout, None)
    y_ref = E.conv(op, x, None, b, cout, in_scale, in_shift, 0.01, res, out_scale, st_ref, w_ref=w, out_channels_alloc=spec.cout_alloc)
    st = hip.new_stats(n, cout, DEV)
    y = hip.conv(op, x.to(DEV), wf, b.to(DEV), cout, in_scale.to(DEV), in_shift.to(DEV), 0.01, None if res is None else res.to(DEV), out_scale.to(DEV), st,
                 out_channels_alloc=spec.cout_alloc, prec=prec)
    assert y.shape == y_ref.shape                       # (padding channels of a 2-channel head are allocated but not initialised)
    close(y[..., :cout], y_ref[..., :cout], rtol=tol, what="fwd")
    close(st, st_ref, rtol=max(1e-5, tol), what="stats")
    # ---- plain forward (no prologue / epilogue extras)
    y2_ref = E.conv(op, x, None, None, cout, w_ref=w, out_channels_alloc=spec.cout_alloc)
    y2 = hip.conv(op, x.to(DEV), wf, None, cout, out_channels_alloc=spec.cout_alloc, prec=prec)
    close(y2[..., :cout], y2_ref[..., :cout], rtol=tol, what="fwd plain")

    # ---- data gradient
    dy = torch.zeros(n, do, ho, wo, spec.cout_alloc)
    dy[..., :cout] = rnd(n, do, ho, wo, cout, seed=8)
    dx_ref = E.conv(pk.dgrad_op(op), dy, None, None, cin, out=torch.empty(n, d, h, w_, cin), w_ref=w, fwd_op=op)
    dx = hip.conv(pk.dgrad_op(op), dy.to(DEV), wd, None, cin, out=torch.empty((n, d, h, w_, cin), device=DEV), prec=prec)
    close(dx, dx_ref, rtol=tol, what="dgrad")

    # ---- weight / bias gradient with the recomputed prologue
    dyv = dy[..., :cout]
    gw_ref, gb_ref = E.wgrad(op, x, in_scale, in_shift, 0.0, dyv, cout, None, spec.has_bias_map, w.numel(), w_ref_shape=w.shape)
    gw, gb = hip.wgrad(op, x.to(DEV), in_scale.to(DEV), in_shift.to(DEV), 0.0, dy.to(DEV)[..., :cout], cout, spec.inv_map, spec.has_bias_map, w.numel(), prec=prec)
    close(gw, gw_ref, rtol=max(5e-5, tol), what="wgrad")
    if gb is not None:
        close(gb, gb_ref, rtol=max(5e-5, tol), what="bgrad")


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("op,cin,cout,size,n", [
    (pk.CONV3_S1, 16, 16, (64, 64, 64), 2),      # conv16s (sliding window), interior + edge tiles
    (pk.CONV3_S1, 8, 16, (34, 38, 50), 1),       # conv16s with partial tiles in H and W, Cin < 16
    (pk.CONV3_S1, 32, 32, (12, 12, 16), 2),      # generic kernel, fast epilogue
    (pk.CONV3_S1, 48, 16, (9, 10, 20), 1),       # generic kernel, partial tiles (slow epilogue)
    (pk.CONV3_S2, 16, 32, (16, 16, 32), 1),      # stride-2 data gradient (8 output-parity classes)
    (pk.CONV1, 32, 16, (8, 8, 32), 2),
    (pk.CONVT2, 16, 16, (6, 8, 16), 1),
])
def test_fused_norm_backward_sums(hip, op, cin, cout, size, n, prec):
    """conv(..., nb=(x, scale, shift, slope)): the data-gradient epilogue accumulates the InstanceNorm-backward sums S1, S2 that
    cwf_in_bwd_stats computes in a separate pass; dx from in_bwd_apply must match the two-pass in_bwd (same split-bf16 dgrad)."""
    from cwf import functional as CF
    d, h, w_ = size
    x = rnd(n, d, h, w_, cin, seed=11)
    wshape = (cin, cout, 2, 2, 2) if op == pk.CONVT2 else ((cout, cin, 1, 1, 1) if op == pk.CONV1 else (cout, cin, 3, 3, 3))
    w = rnd(*wshape, seed=12, scale=1.0 / math.sqrt(cin * (27 if wshape[2] == 3 else 1)))
    in_scale = rnd(n, cin, seed=14).abs() + 0.5
    in_shift = rnd(n, cin, seed=15)
    spec = _packed(CF.ConvSpec(op, cin, cout), w, prec)
    do, ho, wo = pk.out_dims(op, d, h, w_)
    dy = torch.zeros(n, do, ho, wo, spec.cout_alloc)
    dy[..., :cout] = rnd(n, do, ho, wo, cout, seed=18)
    xd, dyd, sc, sh = x.to(DEV), dy.to(DEV), in_scale.to(DEV), in_shift.to(DEV)
    g1 = hip.conv(pk.dgrad_op(op), dyd, spec.wpk16_d, None, cin, out=torch.empty((n, d, h, w_, cin), device=DEV), prec=prec)
    dx_two_pass = hip.in_bwd(g1, xd, sc, sh, 0.01)
    sums = hip.new_stats(n, cin, DEV)
    g2 = hip.conv(pk.dgrad_op(op), dyd, spec.wpk16_d, None, cin, out=torch.empty((n, d, h, w_, cin), device=DEV), prec=prec,
                  stats=sums, nb=(xd, sc, sh, 0.01))
    assert torch.equal(g1, g2)
    dx_fused = hip.in_bwd_apply(g2, xd, sc, sh, 0.01, sums)
    close(dx_fused, dx_two_pass.cpu(), rtol=2e-5, what="fused norm backward")


@pytest.mark.parametrize("size,n", [((64, 64, 64), 2), ((34, 38, 50), 1)])
def test_conv16s_reads_a_bf16_input_image(hip, size, n):
    """Round 3: the full-resolution 16-channel data gradient with its input as a bf16 image (conv(..., x16=): LDS-DMA loaders, a
    16-row ring): bit-equal to the fp32-input launch in single-bf16 mode (same operand values, same MFMA order), with the residual
    and the fused InstanceNorm-backward sums; ragged tiles / borders included."""
    from cwf import functional as CF
    d, h, w_ = size
    c = 16
    dy = rnd(n, d, h, w_, c, seed=31).to(DEV)
    x = rnd(n, d, h, w_, c, seed=32).to(DEV)
    res = rnd(n, d, h, w_, c, seed=33).to(DEV)
    sc = (rnd(n, c, seed=34).abs() + 0.5).to(DEV)
    sh = rnd(n, c, seed=35).to(DEV)
    w = rnd(c, c, 3, 3, 3, seed=36, scale=1.0 / math.sqrt(c * 27))
    spec = _packed(CF.ConvSpec(pk.CONV3_S1, c, c), w, "bf16")
    dy16 = hip.to_bf16(dy)
    for kw in (dict(), dict(residual=res), dict(nb=(x, sc, sh, 0.01)), dict(residual=res, nb=(x, sc, sh, 0.0))):
        outs = []
        for img in (None, dy16):
            st = hip.new_stats(n, c, DEV) if "nb" in kw else None
            y = hip.conv(pk.CONV3_S1, dy, spec.wpk16_d, None, c, out=torch.empty((n, d, h, w_, c), device=DEV), prec="bf16",
                         fwd_op=pk.CONV3_S1, stats=st, x16=img, **kw)
            outs.append((y, st))
        assert torch.equal(outs[0][0], outs[1][0]), sorted(kw)
        if outs[0][1] is not None:
            close(outs[1][1], outs[0][1].cpu(), rtol=1e-6, what="nb sums")          # (fp32 partial sums folded by f64 atomics)


@pytest.mark.parametrize("cin,cout,size,n", [(32, 32, (16, 16, 32), 2), (64, 64, (8, 12, 16), 1), (128, 256, (6, 6, 10), 2), (96, 96, (8, 8, 16), 1)])
def test_dma_weight_gradient_32_channel_groups(hip, cin, cout, size, n):
    """cwf_wgrad_s1_bf16 (wgrad_s1d_kernel: bf16 images by LDS-DMA, 16-channel chunks x 32-channel groups) against the fp32-tensor
    kernel (same single-bf16 operands: summation order only) and the oracle; ragged tiles included."""
    from cwf import functional as CF, kernels
    import os
    d, h, w_ = size
    x = rnd(n, d, h, w_, cin, seed=41); g = rnd(n, d, h, w_, cout, seed=42)
    sc = rnd(n, cin, seed=43).abs() + 0.5; sh = rnd(n, cin, seed=44)
    xd, gd, scd, shd = x.to(DEV), g.to(DEV), sc.to(DEV), sh.to(DEV)
    spec = CF.ConvSpec(pk.CONV3_S1, cin, cout).to(torch.device(DEV))
    wn = cout * cin * 27
    kernels.set_precision("bf16x3", wgrad="bf16", dgrad="bf16")
    try:
        assert hip.bf16_operands_ok(pk.CONV3_S1, cin, cout, d * h * w_) == 32
        dw_a = torch.zeros(wn, device=DEV); db_a = torch.zeros(cout, device=DEV)
        dw_b = torch.zeros(wn, device=DEV); db_b = torch.zeros(cout, device=DEV)
        os.environ["CWF_NO_BF16_OPERANDS"] = "1"
        try:
            hip.wgrad_to(("s1a", cin, cout), pk.CONV3_S1, xd, scd, shd, 0.01, gd, cout, spec.inv_map, dw_a, db_a)
            hip.wgrad_flush(torch.device(DEV))
        finally:
            del os.environ["CWF_NO_BF16_OPERANDS"]
        hip.wgrad_to(("s1b", cin, cout), pk.CONV3_S1, xd, scd, shd, 0.01, gd, cout, spec.inv_map, dw_b, db_b)
        hip.wgrad_flush(torch.device(DEV))
    finally:
        kernels.set_precision("fp32")
    torch.cuda.synchronize()
    close(dw_b, dw_a.cpu(), rtol=2e-5, what="dma wgrad vs fp32-tensor wgrad")
    close(db_b, db_a.cpu(), rtol=2e-5, what="dma bgrad")
    gw_ref, gb_ref = E.wgrad(pk.CONV3_S1, x, sc, sh, 0.01, g, cout, None, True, wn, w_ref_shape=(cout, cin, 3, 3, 3))
    close(dw_b, gw_ref, rtol=PREC_TOL["bf16"], what="dma wgrad vs oracle")
    close(db_b, gb_ref, rtol=PREC_TOL["bf16"], what="dma bgrad vs oracle")


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("size,n", [((32, 32, 32), 2), ((34, 38, 50), 1), ((64, 64, 64), 2)])
def test_stem_kernel_packs_eight_taps_times_four_channels(hip, size, n, prec):
    """cwf_conv_stem_bf16 (K = 8 taps x 4 channels, raw weights) against the oracle and against the 16-channel-slot kernel it replaces
    (same operand values, another summation order), with bias, dropout3d scale and output statistics; ragged tiles included."""
    from cwf import functional as CF
    d, h, w_ = size
    x = rnd(n, d, h, w_, 4, seed=51)
    w = rnd(16, 4, 3, 3, 3, seed=52, scale=1.0 / math.sqrt(4 * 27))
    b = rnd(16, seed=53, scale=0.1)
    osc = (rnd(n, 16, seed=54) > -0.5).float() * 1.25
    spec = _packed(CF.ConvSpec(pk.CONV3_S1, 4, 16), w, prec)
    xd, wd, bd, od = x.to(DEV), w.to(DEV).contiguous(), b.to(DEV), osc.to(DEV)
    st_a, st_b = hip.new_stats(n, 16, DEV), hip.new_stats(n, 16, DEV)
    y_a = hip.conv(pk.CONV3_S1, xd, spec.wpk16_f, bd, 16, out_scale=od, stats=st_a, prec=prec, w_ref=wd)           # stem kernel
    y_b = hip.conv(pk.CONV3_S1, xd, spec.wpk16_f, bd, 16, out_scale=od, stats=st_b, prec=prec)                     # conv16s
    st_ref = E.new_stats(n, 16, None)
    y_ref = E.conv(pk.CONV3_S1, x, None, b, 16, None, None, 1.0, None, osc, st_ref, w_ref=w)
    tol = PREC_TOL[prec]
    close(y_a, y_ref, rtol=tol, what="stem vs oracle")
    close(st_a, st_ref, rtol=max(1e-5, tol), what="stem stats vs oracle")
    close(y_a, y_b.cpu(), rtol=2e-5 if prec == "bf16x3" else 2e-5, what="stem vs conv16s")
    close(st_a, st_b.cpu(), rtol=2e-5, what="stem stats vs conv16s")


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("size,n", [((64, 64, 64), 2), ((34, 38, 70), 1), ((128, 128, 128), 1)])
def test_first_downsampling_kernel(hip, size, n, prec):
    """cwf_conv_s2c16_bf16 (3x3x3 stride 2, 16 -> 32, parity-split halo rows, raw weights) against the oracle and against the tap-table
    kernel it replaces, with bias and output statistics; odd extents / ragged tiles included."""
    from cwf import functional as CF
    d, h, w_ = size
    x = rnd(n, d, h, w_, 16, seed=61)
    w = rnd(32, 16, 3, 3, 3, seed=62, scale=1.0 / math.sqrt(16 * 27))
    b = rnd(32, seed=63, scale=0.1)
    spec = _packed(CF.ConvSpec(pk.CONV3_S2, 16, 32), w, prec)
    xd, wd, bd = x.to(DEV), w.to(DEV).contiguous(), b.to(DEV)
    st_a, st_b = hip.new_stats(n, 32, DEV), hip.new_stats(n, 32, DEV)
    y_a = hip.conv(pk.CONV3_S2, xd, spec.wpk16_f, bd, 32, stats=st_a, prec=prec, w_ref=wd)           # new kernel
    y_b = hip.conv(pk.CONV3_S2, xd, spec.wpk16_f, bd, 32, stats=st_b, prec=prec)                     # tap-table kernel
    tol = PREC_TOL[prec]
    close(y_a, y_b.cpu(), rtol=2e-5, what="s2 kernel vs tap-table kernel")
    close(st_a, st_b.cpu(), rtol=2e-5, what="s2 stats vs tap-table kernel")
    if d * h * w_ <= 64 ** 3:
        st_ref = E.new_stats(n, 32, None)
        y_ref = E.conv(pk.CONV3_S2, x, None, b, 32, None, None, 1.0, None, None, st_ref, w_ref=w)
        close(y_a, y_ref, rtol=tol, what="s2 vs oracle")
        close(st_a, st_ref, rtol=max(1e-5, tol), what="s2 stats vs oracle")


@pytest.mark.parametrize("cin,cout,size,n", [(32, 16, (32, 32, 32), 2), (64, 32, (16, 16, 16), 1), (128, 64, (8, 8, 8), 2), (16, 4, (8, 8, 16), 1)])
def test_pointwise_conv_writes_its_bf16_image(hip, cin, cout, size, n):
    """conv(..., y16=): the 1x1x1 stream kernel leaves bf16(y) beside y in the same launch (other layers: conversion afterwards)."""
    from cwf import functional as CF
    d, h, w_ = size
    x = rnd(n, d, h, w_, cin, seed=71).to(DEV)
    w = rnd(cout, cin, 1, 1, 1, seed=72, scale=1.0 / math.sqrt(cin))
    b = rnd(cout, seed=73, scale=0.1).to(DEV)
    spec = _packed(CF.ConvSpec(pk.CONV1, cin, cout), w, "bf16x3")
    y_ref = hip.conv(pk.CONV1, x, spec.wpk16_f, b, cout, prec="bf16x3")
    y16 = torch.empty((n, d, h, w_, cout), dtype=torch.bfloat16, device=DEV)
    y = hip.conv(pk.CONV1, x, spec.wpk16_f, b, cout, prec="bf16x3", y16=y16)
    assert torch.equal(y, y_ref) and torch.equal(y16, y_ref.to(torch.bfloat16))


def _bf16_rne(t):
    """round-to-nearest-even bf16 of an fp32 tensor (what v_cvt_pk_bf16_f32 does)"""
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("size,n", [((64, 64, 64), 2), ((34, 38, 50), 1), ((32, 32, 32), 1)])
def test_bf16_operand_images_and_dma_weight_gradient(hip, size, n):
    """Round 3: the full-resolution 16-channel weight gradient on bf16 operand images (cwf_wgrad16_bf16, LDS-DMA staging).
    (a) cwf_in_bwd_apply_ex: dx bit-equal to cwf_in_bwd_apply, dx16 = bf16(dx), xa16 = bf16(act(IN(x))) -- with and without the
        fp32 output; cwf_norm_act_add_ex / cwf_to_bf16 likewise.
    (b) the DMA kernel's dW / db equal the fp32-tensor kernel's (same single-bf16 operand values: tolerance = summation order only),
        both against the oracle (kernel_emul wgrad) within the single-bf16 tolerance; ragged tiles and borders included."""
    from cwf import functional as CF, kernels
    d, h, w_ = size
    c = 16
    x = rnd(n, d, h, w_, c, seed=21)
    g = rnd(n, d, h, w_, c, seed=22)
    add = rnd(n, d, h, w_, c, seed=23)
    sc = rnd(n, c, seed=24).abs() + 0.5
    sh = rnd(n, c, seed=25)
    xd, gd, ad, scd, shd = x.to(DEV), g.to(DEV), add.to(DEV), sc.to(DEV), sh.to(DEV)
    sums = hip.in_stats(gd)                       # any (N, C, 2) doubles will do for the apply formula
    for slope in (0.0, 0.01):
        dx_ref = hip.in_bwd_apply(gd, xd, scd, shd, slope, sums, dx_add=ad)
        dx, dx16, xa16 = hip.in_bwd_apply16(gd, xd, scd, shd, slope, sums, dx_add=ad, want_dx16=True, want_xa16=True)
        assert torch.equal(dx, dx_ref)
        assert torch.equal(dx16, _bf16_rne(dx_ref))
        hh = torch.addcmul(shd[:, None, None, None, :], xd, scd[:, None, None, None, :])     # (fma: the kernel's fmaf)
        xa_ref = torch.maximum(hh, hh * slope)
        # the host formula may differ from the kernel's fma in the last bit before rounding: compare in bf16 ulps
        diff = (xa16.float() - _bf16_rne(xa_ref).float()).abs()
        assert float((diff > 0).float().mean()) < 1e-3 and float(diff.max()) <= float(xa_ref.abs().max()) * 2 ** -7
        assert torch.equal(xa16, hip.to_bf16(xd, scd, shd, slope))
        _, dx16b, _ = hip.in_bwd_apply16(gd, xd, scd, shd, slope, sums, dx_add=ad, want_dx16=True, need_f32=False)
        assert torch.equal(dx16b, dx16)
    y_ref = hip.norm_act_add(gd, scd, shd, 0.01, ad)
    y, y16 = hip.norm_act_add(gd, scd, shd, 0.01, ad, want16=True)
    assert torch.equal(y, y_ref) and torch.equal(y16, _bf16_rne(y_ref))
    assert torch.equal(hip.to_bf16(gd), _bf16_rne(gd))

    # ---- weight gradient
    spec = CF.ConvSpec(pk.CONV3_S1, c, c).to(torch.device(DEV))
    wn = c * c * 27
    kernels.set_precision("bf16x3", wgrad="bf16", dgrad="bf16")
    try:
        assert bool(hip.bf16_operands_ok(pk.CONV3_S1, c, c, d * h * w_)) == (d * h * w_ >= 32768)
        dw_a = torch.zeros(wn, device=DEV); db_a = torch.zeros(c, device=DEV)
        dw_b = torch.zeros(wn, device=DEV); db_b = torch.zeros(c, device=DEV)
        dw_c = torch.zeros(wn, device=DEV); db_c = torch.zeros(c, device=DEV)
        import os
        os.environ["CWF_NO_BF16_OPERANDS"] = "1"                                                                   # the fp32-tensor kernel
        try:
            hip.wgrad_to("t16a", pk.CONV3_S1, xd, scd, shd, 0.01, gd, c, spec.inv_map, dw_a, db_a)
            hip.wgrad_flush(torch.device(DEV))
        finally:
            del os.environ["CWF_NO_BF16_OPERANDS"]
        xa16 = hip.to_bf16(xd, scd, shd, 0.01)
        hip.wgrad_to("t16b", pk.CONV3_S1, xd, scd, shd, 0.01, gd, c, spec.inv_map, dw_b, db_b, x16=xa16, dy16=hip.to_bf16(gd))
        hip.wgrad_flush(torch.device(DEV))
        hip.wgrad_to("t16c", pk.CONV3_S1, xd, scd, shd, 0.01, gd, c, spec.inv_map, dw_c, db_c)                          # both images made inside
        hip.wgrad_flush(torch.device(DEV))
    finally:
        kernels.set_precision("fp32")
    torch.cuda.synchronize()
    if d * h * w_ >= 32768:
        close(dw_b, dw_a.cpu(), rtol=2e-5, what="dma wgrad vs fp32-tensor wgrad")
        close(db_b, db_a.cpu(), rtol=2e-5, what="dma bgrad vs fp32-tensor bgrad")
    else:
        assert torch.equal(dw_b, dw_a)            # not eligible: the same (fp32-tensor) kernel ran
    assert torch.equal(dw_c, dw_b) and torch.equal(db_c, db_b)
    gw_ref, gb_ref = E.wgrad(pk.CONV3_S1, x, sc, sh, 0.01, g, c, None, True, wn, w_ref_shape=(c, c, 3, 3, 3))
    close(dw_b, gw_ref, rtol=PREC_TOL["bf16"], what="dma wgrad vs oracle")
    close(db_b, gb_ref, rtol=PREC_TOL["bf16"], what="dma bgrad vs oracle")


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("cin,cout,size,n", [
    (32, 32, (16, 16, 32), 2),       # 64 tiles, two samples: 32 workgroups x one round
    (32, 32, (64, 32, 32), 2),       # 512 tiles x 1 output group: the product's geometry (256 workgroups, two tiles each), XCD tile map
    (32, 32, (20, 12, 48), 3),       # 135 tiles over 3 samples: ranges of 1-2 tiles, odd rounds (group 1 idle), sample boundaries inside ranges
    (32, 32, (8, 4, 16), 3),         # 6 tiles over 3 samples: fewer tiles than workgroup slots
    (64, 64, (16, 16, 32), 2),       # four chunks, two output groups (128 workgroups)
    (64, 64, (8, 8, 16), 2),
    (128, 128, (4, 8, 16), 1),       # eight chunks, four output groups
    (96, 96, (4, 4, 16), 2),         # the fused edge decoupler's data gradient shape (three output groups)
    (32, 64, (4, 4, 32), 1),
    (64, 32, (8, 4, 16), 1),
])
def test_weight_stationary_conv(hip, cin, cout, size, n, prec):
    """convws_kernel (conv_ws.hip): the 3x3x3 stride-1 layers with >= 32 input channels whose extents are multiples of the 4x4x16
    tile -- weights resident in LDS, persistent workgroups.  Every prologue / epilogue combination that reaches it in the model:
    forward with InstanceNorm + LeakyReLU prologue, bias, residual, statistics (EnBlock / DeBlock conv2); plain forward into a
    channel slice; data gradient with a residual operand and the norm-backward sums (NaaLink / in_bwd fusion)."""
    tol = PREC_TOL[prec]
    from cwf import functional as CF
    old = hip.lib.cwf_debug_ws_min_units(1)          # (the product sends only layers with >= 256 (tile, output group) units here ...
    old3 = hip.lib.cwf_debug_ws_x3(1)                # ... and only single-bf16 launches: the split-bf16 form is tested all the same)
    try:
        _weight_stationary_case(hip, cin, cout, size, n, prec, tol, CF)
    finally:
        hip.lib.cwf_debug_ws_min_units(old)
        hip.lib.cwf_debug_ws_x3(old3)


def _weight_stationary_case(hip, cin, cout, size, n, prec, tol, CF):
    d, h, w_ = size
    op = pk.CONV3_S1
    x = rnd(n, d, h, w_, cin, seed=31)
    w = rnd(cout, cin, 3, 3, 3, seed=32, scale=1.0 / math.sqrt(cin * 27))
    b = rnd(cout, seed=33, scale=0.1)
    in_scale = rnd(n, cin, seed=34).abs() + 0.5
    in_shift = rnd(n, cin, seed=35)
    res = rnd(n, d, h, w_, cout, seed=37)
    spec = _packed(CF.ConvSpec(op, cin, cout), w, prec)
    st_ref = E.new_stats(n, cout, None)
    y_ref = E.conv(op, x, None, b, cout, in_scale, in_shift, 0.01, res, None, st_ref, w_ref=w)
    st = hip.new_stats(n, cout, DEV)
    y = hip.conv(op, x.to(DEV), spec.wpk16_f, b.to(DEV), cout, in_scale.to(DEV), in_shift.to(DEV), 0.01, res.to(DEV), None, st, prec=prec)
    close(y, y_ref, rtol=tol, what="fwd")
    close(st, st_ref, rtol=max(1e-5, tol), what="stats")
    # ReLU prologue, no residual, into a channel slice of a wider buffer, strided input
    wide = torch.full((n, d, h, w_, cout + 8), 7.0, device=DEV)
    xw = torch.zeros((n, d, h, w_, cin + 4), device=DEV); xw[..., :cin] = x.to(DEV)
    hip.conv(op, xw[..., :cin], spec.wpk16_f, b.to(DEV), cout, in_scale.to(DEV), in_shift.to(DEV), 0.0, out=wide[..., 8:], prec=prec)
    close(wide[..., 8:], E.conv(op, x, None, b, cout, in_scale, in_shift, 0.0, w_ref=w), rtol=tol, what="fwd into slice")
    assert bool((wide[..., :8] == 7.0).all())
    # data gradient with a residual (the carried skip gradient) and the norm-backward sums, against the two-pass form
    dy = rnd(n, d, h, w_, cout, seed=38)
    carry = rnd(n, d, h, w_, cin, seed=39)
    dx_ref = E.conv(pk.dgrad_op(op), dy, None, None, cin, out=torch.empty(n, d, h, w_, cin), w_ref=w, fwd_op=op, residual=carry)
    sums = hip.new_stats(n, cin, DEV)
    xd, sc, sh = x.to(DEV), in_scale.to(DEV), in_shift.to(DEV)
    g = hip.conv(pk.dgrad_op(op), dy.to(DEV), spec.wpk16_d, None, cin, out=torch.empty((n, d, h, w_, cin), device=DEV), residual=carry.to(DEV),
                 prec=prec, stats=sums, nb=(xd, sc, sh, 0.01))
    close(g, dx_ref, rtol=tol, what="dgrad + residual")
    close(hip.in_bwd_apply(g, xd, sc, sh, 0.01, sums), hip.in_bwd(g, xd, sc, sh, 0.01).cpu(), rtol=2e-5, what="norm-backward sums")


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("op,cin,cout,size,n", [
    (pk.CONV1, 32, 16, (8, 8, 32), 2),           # DeUp conv3 (concat -> C)
    (pk.CONV1, 16, 4, (4, 4, 16), 3),            # endconv; waves span samples (stats flush / per-sample reload inside a wave)
    (pk.CONV1, 16, 32, (8, 8, 16), 2),           # two output tiles
    (pk.CONV1, 4, 16, (4, 8, 8), 3),             # 4 input channels (endconv's data gradient): second float4 masked
    (pk.CONV1, 64, 32, (8, 8, 16), 1),           # two K-steps x two tiles
    (pk.CONV1, 48, 16, (4, 4, 16), 2),           # partial second K-step
    (pk.CONV1, 20, 12, (4, 4, 16), 2),           # channel counts that are multiples of 4 only
    (pk.CONVT2, 16, 16, (4, 6, 16), 2),          # eight parity classes from LDS-resident weights
    (pk.CONVT2, 32, 32, (4, 4, 32), 1),
])
def test_pointwise_stream_kernel(hip, op, cin, cout, size, n, prec):
    """pw_conv_kernel (conv_bf16.hip): 1x1x1 convs and the ConvTranspose forward as register-level streams.  The epilogue
    combinations that reach it (bias / residual / statistics; norm-backward sums on data gradients) against the oracle
    emulation; launches with a prologue or out_scale stay on the generic kernel (test_conv_family_fwd_dgrad_wgrad)."""
    tol = PREC_TOL[prec]
    from cwf import functional as CF
    d, h, w_ = size
    x = rnd(n, d, h, w_, cin, seed=21)
    wshape = (cin, cout, 2, 2, 2) if op == pk.CONVT2 else (cout, cin, 1, 1, 1)
    w = rnd(*wshape, seed=22, scale=1.0 / math.sqrt(cin))
    b = rnd(cout, seed=23, scale=0.1)
    spec = _packed(CF.ConvSpec(op, cin, cout), w, prec)
    do, ho, wo = pk.out_dims(op, d, h, w_)
    res = rnd(n, do, ho, wo, cout, seed=27) if op == pk.CONV1 else None
    st_ref = E.new_stats(n, cout, None)
    y_ref = E.conv(op, x, None, b, cout, None, None, 1.0, res, None, st_ref, w_ref=w)
    st = hip.new_stats(n, cout, DEV)
    y = hip.conv(op, x.to(DEV), spec.wpk16_f, b.to(DEV), cout, None, None, 1.0, None if res is None else res.to(DEV), None, st, prec=prec)
    close(y, y_ref, rtol=tol, what="fwd")
    close(st, st_ref, rtol=max(1e-5, tol), what="stats")
    # into a channel slice of a wider buffer (the concatenation buffer of DeUp_Cat), strided input
    wide = torch.full((n, do, ho, wo, cout + 8), 7.0, device=DEV)
    xw = torch.zeros((n, d, h, w_, cin + 4), device=DEV); xw[..., :cin] = x.to(DEV)
    hip.conv(op, xw[..., :cin], spec.wpk16_f, b.to(DEV), cout, out=wide[..., 8:], prec=prec)
    y_plain = E.conv(op, x, None, b, cout, w_ref=w)
    close(wide[..., 8:], y_plain, rtol=tol, what="fwd into slice")
    assert bool((wide[..., :8] == 7.0).all())
    if op != pk.CONV1:
        return
    # data gradient (a 1x1x1 conv with the transposed weights) with a residual operand and the norm-backward sums
    dy = rnd(n, d, h, w_, cout, seed=28)
    carry = rnd(n, d, h, w_, cin, seed=29)
    in_scale = rnd(n, cin, seed=24).abs() + 0.5
    in_shift = rnd(n, cin, seed=25)
    dx_ref = E.conv(pk.CONV1, dy, None, None, cin, out=torch.empty(n, d, h, w_, cin), w_ref=w, fwd_op=op) + carry
    sums = hip.new_stats(n, cin, DEV)
    xd, sc, sh = x.to(DEV), in_scale.to(DEV), in_shift.to(DEV)
    dx = hip.conv(pk.CONV1, dy.to(DEV), spec.wpk16_d, None, cin, residual=carry.to(DEV), out=torch.empty((n, d, h, w_, cin), device=DEV),
                  prec=prec, stats=sums, nb=(xd, sc, sh, 0.01))
    close(dx, dx_ref, rtol=tol, what="dgrad + residual")
    close(hip.in_bwd_apply(dx, xd, sc, sh, 0.01, sums), hip.in_bwd(dx, xd, sc, sh, 0.01).cpu(), rtol=2e-5, what="norm-backward sums")


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
def test_parity_class_outputs_take_the_fast_epilogue(hip, prec):
    """Eight-class launches of the generic kernel (stride-2 data gradient, ConvTranspose forward beyond the pointwise kernel's
    shapes) on interior tiles: row-base + lane-offset stores with output voxel stride 2, residual operand, statistics and
    norm-backward sums."""
    tol = PREC_TOL[prec]
    from cwf import functional as CF
    # EnDown data gradient: 32 -> 16 channels onto a 2x finer grid, carried gradient added, sums of the block tail it feeds
    n, cin, cout, (d, h, w_) = 2, 16, 32, (16, 16, 32)
    w = rnd(cout, cin, 3, 3, 3, seed=32, scale=1.0 / math.sqrt(cin * 27))
    spec = _packed(CF.ConvSpec(pk.CONV3_S2, cin, cout), w, prec)
    do, ho, wo = pk.out_dims(pk.CONV3_S2, d, h, w_)
    x = rnd(n, d, h, w_, cin, seed=31)
    dy = rnd(n, do, ho, wo, cout, seed=38)
    carry = rnd(n, d, h, w_, cin, seed=39)
    sc, sh = (rnd(n, cin, seed=34).abs() + 0.5).to(DEV), rnd(n, cin, seed=35).to(DEV)
    dx_ref = E.conv(pk.CONV3_S2_DGRAD, dy, None, None, cin, out=torch.empty(n, d, h, w_, cin), w_ref=w, fwd_op=pk.CONV3_S2) + carry
    sums = hip.new_stats(n, cin, DEV)
    dx = hip.conv(pk.CONV3_S2_DGRAD, dy.to(DEV), spec.wpk16_d, None, cin, residual=carry.to(DEV), out=torch.empty((n, d, h, w_, cin), device=DEV),
                  prec=prec, stats=sums, nb=(x.to(DEV), sc, sh, 0.01))
    close(dx, dx_ref, rtol=tol, what="s2 dgrad + residual")
    close(hip.in_bwd_apply(dx, x.to(DEV), sc, sh, 0.01, sums), hip.in_bwd(dx, x.to(DEV), sc, sh, 0.01).cpu(), rtol=2e-5, what="norm-backward sums")
    # ConvTranspose 64 -> 64 (weights too large for the pointwise kernel's LDS copy): bias + statistics
    n, c, (d, h, w_) = 1, 64, (4, 4, 16)
    w = rnd(c, c, 2, 2, 2, seed=42, scale=1.0 / math.sqrt(c))
    b = rnd(c, seed=43, scale=0.1)
    spec = _packed(CF.ConvSpec(pk.CONVT2, c, c), w, prec)
    x = rnd(n, d, h, w_, c, seed=41)
    st_ref = E.new_stats(n, c, None)
    y_ref = E.conv(pk.CONVT2, x, None, b, c, None, None, 1.0, None, None, st_ref, w_ref=w)
    st = hip.new_stats(n, c, DEV)
    y = hip.conv(pk.CONVT2, x.to(DEV), spec.wpk16_f, b.to(DEV), c, None, None, 1.0, None, None, st, prec=prec)
    close(y, y_ref, rtol=tol, what="convT fwd")
    close(st, st_ref, rtol=max(1e-5, tol), what="convT stats")


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("op,cin,cout,size,n", [
    (pk.CONV1, 32, 16, (8, 8, 32), 2),
    (pk.CONV1, 16, 4, (4, 6, 10), 3),            # 240 voxels per sample: ragged runs of 4-voxel groups
    (pk.CONV1, 64, 32, (8, 8, 16), 1),
    (pk.CONV1, 128, 64, (4, 4, 8), 2),
    (pk.CONV1, 20, 12, (4, 4, 16), 2),           # partial channel blocks
    (pk.CONVT2, 16, 16, (4, 6, 8), 2),
    (pk.CONVT2, 32, 32, (4, 4, 12), 1),
])
def test_pointwise_weight_gradient_stream(hip, op, cin, cout, size, n, prec):
    """pw_wgrad_kernel (wgrad_bf16.hip): weight / bias gradients of the 1x1x1 and transposed layers from fp32 MFMA fragments loaded
    straight from global memory, with the recomputed InstanceNorm + activation prologue; against the oracle emulation at the fp32
    tolerance in BOTH precision modes (the products are exact fp32)."""
    from cwf import functional as CF
    d, h, w_ = size
    x = rnd(n, d, h, w_, cin, seed=51)
    wshape = (cin, cout, 2, 2, 2) if op == pk.CONVT2 else (cout, cin, 1, 1, 1)
    w = rnd(*wshape, seed=52, scale=1.0 / math.sqrt(cin))
    spec = _packed(CF.ConvSpec(op, cin, cout), w, prec)
    do, ho, wo = pk.out_dims(op, d, h, w_)
    dy = rnd(n, do, ho, wo, cout, seed=58)
    in_scale, in_shift = rnd(n, cin, seed=54).abs() + 0.5, rnd(n, cin, seed=55)
    for sc, sh, slope in ((in_scale, in_shift, 0.01), (None, None, 1.0)):
        gw_ref, gb_ref = E.wgrad(op, x, sc, sh, slope, dy, cout, None, spec.has_bias_map, w.numel(), w_ref_shape=w.shape)
        gw, gb = hip.wgrad(op, x.to(DEV), None if sc is None else sc.to(DEV), None if sh is None else sh.to(DEV), slope, dy.to(DEV), cout,
                           spec.inv_map, spec.has_bias_map, w.numel(), prec=prec)
        close(gw, gw_ref, rtol=5e-5, what="wgrad")
        if gb is not None:
            close(gb, gb_ref, rtol=5e-5, what="bgrad")
    # strided operands: x and dy as channel slices of wider buffers (concatenation buffer / padded gradient)
    xw = torch.zeros((n, d, h, w_, cin + 4), device=DEV); xw[..., :cin] = x.to(DEV)
    dw = torch.zeros((n, do, ho, wo, cout + 8), device=DEV); dw[..., 4:4 + cout] = dy.to(DEV)
    gw2, _ = hip.wgrad(op, xw[..., :cin], None, None, 1.0, dw[..., 4:4 + cout], cout, spec.inv_map, spec.has_bias_map, w.numel(), prec=prec)
    close(gw2, gw_ref, rtol=5e-5, what="wgrad strided")


@pytest.mark.parametrize("prec", ["fp32", "bf16x3", "bf16"])
@pytest.mark.parametrize("cin,cout,size,n,G", [
    (128, 32, (8, 8, 16), 2, 3),                 # label heads, first stage
    (32, 2, (8, 8, 16), 2, 3),                   # second stage: 2 logits in 4-channel groups
    (32, 8, (8, 12, 16), 1, 3),                  # edge heads
    (8, 2, (8, 12, 16), 2, 3),                   # <= 16 channels: conv16 tap order of the packed weights
    (16, 16, (6, 8, 20), 1, 2),                  # two groups, ragged tiles
])
def test_channel_grouped_conv(hip, cin, cout, size, n, G, prec):
    """cwf_conv_mfma_bf16_grouped (forward and data gradient) against one oracle conv per group; other channels of the output
    buffer stay untouched."""
    tol = PREC_TOL[prec]
    from cwf import functional as CF
    d, h, w_ = size
    ca = (cout + 3) // 4 * 4
    x_all = rnd(n, d, h, w_, G * cin, seed=61)
    ws = [rnd(cout, cin, 3, 3, 3, seed=62 + q, scale=1.0 / math.sqrt(27 * cin)) for q in range(G)]
    bs = [rnd(cout, seed=72 + q, scale=0.1) for q in range(G)]
    specs = [_packed(CF.ConvSpec(pk.CONV3_S1, cin, cout), w, prec) for w in ws]
    wf = [(s_.wpk_f if prec == "fp32" else s_.wpk16_f) for s_ in specs]
    wd = [(s_.wpk_d if prec == "fp32" else s_.wpk16_d) for s_ in specs]
    y_all = torch.full((n, d, h, w_, G * ca), 5.0, device=DEV)
    hip.conv_grouped(x_all.to(DEV), cin, wf, [b.to(DEV) for b in bs], cout, y_all, x_goff=cin, y_goff=ca, prec=prec)
    for q in range(G):
        ref = E.conv(pk.CONV3_S1, x_all[..., q * cin:(q + 1) * cin], None, bs[q], cout, w_ref=ws[q])
        close(y_all[..., q * ca:q * ca + cout], ref[..., :cout], rtol=tol, what="fwd group %d" % q)
        if ca != cout:
            assert bool((y_all[..., q * ca + cout:(q + 1) * ca] == 5.0).all())
    dy_all = torch.zeros(n, d, h, w_, G * ca)
    for q in range(G):
        dy_all[..., q * ca:q * ca + cout] = rnd(n, d, h, w_, cout, seed=82 + q)
    dx_all = torch.empty((n, d, h, w_, G * cin), device=DEV)
    hip.conv_grouped(dy_all.to(DEV), ca, wd, None, cin, dx_all, x_goff=ca, y_goff=cin, fwd_op=pk.CONV3_S1, prec=prec)
    for q in range(G):
        ref = E.conv(pk.CONV3_S1, dy_all[..., q * ca:(q + 1) * ca], None, None, cin, out=torch.empty(n, d, h, w_, cin), w_ref=ws[q], fwd_op=pk.CONV3_S1)
        close(dx_all[..., q * cin:(q + 1) * cin], ref, rtol=tol, what="dgrad group %d" % q)


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("cin,cout,size,n", [(128, 32, (8, 8, 16), 2), (32, 2, (8, 8, 16), 2), (32, 8, (8, 12, 16), 1), (8, 2, (8, 12, 16), 2)])
def test_channel_grouped_weight_gradient(hip, cin, cout, size, n, prec):
    """cwf_wgrad_mfma_bf16_grouped: three layers' slabs from one launch (producer/consumer kernel for 128 -> 32 and 32 -> 8, the
    generic tiled kernel for the 2-channel layers), reduced by the batched reduce, equal the per-layer weight / bias gradients."""
    from cwf import functional as CF
    G = 3
    d, h, w_ = size
    ca = (cout + 3) // 4 * 4
    x_all = rnd(n, d, h, w_, G * cin, seed=101).to(DEV)
    dy_all = torch.zeros(n, d, h, w_, G * ca)
    for q in range(G):
        dy_all[..., q * ca:q * ca + cout] = rnd(n, d, h, w_, cout, seed=111 + q)
    dy_all = dy_all.to(DEV)
    ws = [rnd(cout, cin, 3, 3, 3, seed=121 + q) for q in range(G)]
    specs = [_packed(CF.ConvSpec(pk.CONV3_S1, cin, cout), w, prec) for w in ws]
    xs = [x_all[..., q * cin:(q + 1) * cin] for q in range(G)]
    dys = [dy_all[..., q * ca:q * ca + cout] for q in range(G)]
    dws = [torch.full((cout, cin, 3, 3, 3), 3.0, device=DEV) for _ in range(G)]
    dbs = [torch.full((cout,), 3.0, device=DEV) for _ in range(G)]
    hip.wgrad_to_grouped(specs, pk.CONV3_S1, xs, dys, cout, [s_.inv_map for s_ in specs], dws, dbs, prec=prec)
    hip.wgrad_flush(torch.device(DEV))
    for q in range(G):
        gw, gb = hip.wgrad(pk.CONV3_S1, xs[q], None, None, 1.0, dys[q], cout, specs[q].inv_map, True, ws[q].numel(), prec=prec)
        close(dws[q].reshape(-1), gw.reshape(-1).cpu(), rtol=2e-6, what="dW group %d" % q)
        close(dbs[q], gb.cpu(), rtol=2e-6, what="db group %d" % q)


def test_head_loss_backward_into_channel_groups(hip):
    """cwf_head_loss_bwd_ex: the three maps' logit gradients written as 4-channel groups of one [N,d,h,w,12] buffer (2 values + zeroed
    padding each) equal the per-map tensors of cwf_head_loss_bwd; forward sums from the strided slices equal the contiguous ones."""
    n, (d, h, w_), scale = 2, (4, 4, 8), 4
    l_all = rnd(n, d, h, w_, 12, seed=91).to(DEV)
    label = torch.randint(0, 4, (n, d * scale, h * scale, w_ * scale), generator=torch.Generator().manual_seed(5)).to(DEV)
    masks = (0b0010, 0b0100, 0b1000)
    slices = [l_all[..., 4 * q:4 * q + 4] for q in range(3)]
    conts = [s_.contiguous() for s_ in slices]
    t0, l0, c0 = hip.head_loss(conts, label, masks, scale)
    t1, l1, c1 = hip.head_loss(slices, label, masks, scale)
    assert torch.equal(l0, l1) and torch.equal(c0, c1)
    gs = torch.full((1,), 0.7, device=DEV)
    ref = hip.head_loss_bwd(conts, label, masks, scale, c0, gs)
    d_all = torch.full_like(l_all, 9.0)
    outs = hip.head_loss_bwd(slices, label, masks, scale, c0, gs, grouped_out=(d_all, 4))
    for q in range(3):
        assert torch.equal(d_all[..., 4 * q:4 * q + 4], ref[q]) and outs[q].data_ptr() == d_all[..., 4 * q:4 * q + 4].data_ptr()
        assert bool((d_all[..., 4 * q + 2:4 * q + 4] == 0).all())


def test_gather_batched_matches_index_maps(hip):
    from cwf import functional as CF
    packer = CF.WeightPacker()
    ws = []
    for op, cin, cout in ((pk.CONV3_S1, 16, 32), (pk.CONV1, 32, 16), (pk.CONVT2, 16, 16), (pk.CONV3_S2, 32, 64)):
        shape = (cin, cout, 2, 2, 2) if op == pk.CONVT2 else ((cout, cin, 1, 1, 1) if op == pk.CONV1 else (cout, cin, 3, 3, 3))
        w = torch.nn.Parameter(rnd(*shape, seed=cin + cout).to(DEV))
        spec = CF.ConvSpec(op, cin, cout)
        packer.add(spec, w)
        ws.append((spec, w))
    packer.refresh()
    for spec, w in ws:
        flat = w.detach().reshape(-1)
        for buf, mp in ((spec.wpk_f, spec.fwd_map), (spec.wpk_d, spec.dgrad_map)):
            ref = torch.where(mp >= 0, flat[mp.clamp_min(0).long()], torch.zeros((), device=DEV))
            assert torch.equal(buf, ref)


@pytest.mark.parametrize("c,shape", [(16, (2, 8, 8, 16)), (96, (1, 4, 6, 8)), (384, (1, 4, 4, 4)), (128, (2, 5, 3, 7))])
def test_instance_norm_pieces(hip, c, shape):
    n, d, h, w = shape
    x = rnd(n, d, h, w, c, seed=1) + 0.3
    st_ref = E.in_stats(x)
    st = hip.in_stats(x.to(DEV))
    close(st, st_ref, rtol=1e-6, what="stats")
    sc_ref, sh_ref = E.in_finalize(st_ref, d * h * w)
    sc, sh = hip.in_finalize(st, d * h * w)
    close(sc, sc_ref, rtol=1e-5); close(sh, sh_ref, rtol=1e-5)
    res = rnd(n, d, h, w, c, seed=2)
    close(hip.norm_act_add(x.to(DEV), sc, sh, 0.01, res.to(DEV)), E.norm_act_add(x, sc_ref, sh_ref, 0.01, res), what="norm_act_add")
    dy = rnd(n, d, h, w, c, seed=3)
    for slope in (0.0, 0.01):
        close(hip.in_bwd(dy.to(DEV), x.to(DEV), sc, sh, slope), E.in_bwd(dy, x, sc_ref, sh_ref, slope), rtol=5e-5, what="in_bwd")
    # against autograd of the reference op chain
    xr = x.clone().requires_grad_(True)
    y = torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(xr.permute(0, 4, 1, 2, 3)), 0.01)
    y.backward(dy.permute(0, 4, 1, 2, 3))
    close(hip.in_bwd(dy.to(DEV), x.to(DEV), sc, sh, 0.01), xr.grad, rtol=1e-4, what="in_bwd vs autograd")


def test_single_op_token_kernels(hip):
    """The single-operation token kernels of the C ABI (cwf_layernorm_fwd / _bwd, cwf_gemm with bias / GELU / residual / strided batched
    operands, cwf_softmax_rows / _bwd, cwf_gelu_bwd, cwf_colsum, cwf_mul, cwf_add) against the kernel oracle.  The model runs the
    fused forms of these (cwf_ln_pair_*, cwf_gemm_ex, cwf_attn_*: tests/test_coupler_gpu.py); the single-op entry points stay part
    of the library's surface."""
    b, t, e, heads = 2, 129, 512, 8
    hd = e // heads
    x = rnd(b, t, e, seed=1)
    gam, bet = rnd(e, seed=3) * 0.1 + 1, rnd(e, seed=4) * 0.1
    y, mean, rstd = hip.layernorm_fwd(x.to(DEV), gam.to(DEV), bet.to(DEV))
    y_r, mean_r, rstd_r = E.layernorm_fwd(x, gam, bet)
    close(y, y_r, rtol=2e-5, what="layernorm"); close(mean, mean_r, rtol=1e-5); close(rstd, rstd_r, rtol=1e-5)
    dy = rnd(b, t, e, seed=5)
    dg, db = torch.empty(e, device=DEV), torch.empty(e, device=DEV)
    dx = hip.layernorm_bwd(dy.to(DEV), x.to(DEV), gam.to(DEV), mean, rstd, dg, db)
    dg_r, db_r = torch.empty(e), torch.empty(e)
    dx_r = E.layernorm_bwd(dy, x, gam, mean_r, rstd_r, dg_r, db_r)
    close(dx, dx_r, rtol=1e-4, what="layernorm dx"); close(dg, dg_r, rtol=1e-4); close(db, db_r, rtol=1e-4)
    # y = gelu(x W^T + bias) + residual
    w, bias, res = rnd(e, e, seed=6) / math.sqrt(e), rnd(e, seed=7) * 0.1, rnd(b, t, e, seed=8)
    args = lambda dev: dict(bias=bias.to(dev), residual=res.to(dev), sr=(e, 0, 0), act=1)
    out = hip.gemm(x.to(DEV), (e, 1, 0, 0), w.to(DEV), (1, e, 0, 0), torch.empty(b, t, e, device=DEV), (e, 0, 0), b * t, e, e, **args(DEV))
    out_r = E.gemm(x, (e, 1, 0, 0), w, (1, e, 0, 0), torch.empty(b, t, e), (e, 0, 0), b * t, e, e, **args("cpu"))
    close(out, out_r, rtol=2e-5, what="gemm + bias + gelu + residual")
    # per-head Q K^T over strided operands (SelfAttention.py:94-98), softmax rows and its adjoint
    q, kv = rnd(b, t, e, seed=9), rnd(b, t, 2 * e, seed=10)
    sa, sb_, sc = (e, 1, t * e, hd), (1, 2 * e, t * 2 * e, hd), (t, heads * t * t, t * t)
    p_h = hip.gemm(q.to(DEV), sa, kv.to(DEV), sb_, torch.empty(b, heads, t, t, device=DEV), sc, t, t, hd, zb=b, zh=heads, alpha=hd ** -0.5)
    p_r = E.gemm(q, sa, kv, sb_, torch.empty(b, heads, t, t), sc, t, t, hd, zb=b, zh=heads, alpha=hd ** -0.5)
    close(p_h, p_r, rtol=2e-5, what="strided batched gemm")
    hip.softmax_rows_(p_h); E.softmax_rows_(p_r)
    close(p_h, p_r, rtol=2e-5, what="softmax rows")
    dp = rnd(b, heads, t, t, seed=11)
    close(hip.softmax_rows_bwd_(p_h, dp.to(DEV)), E.softmax_rows_bwd_(p_r, dp.clone()), rtol=1e-4, what="softmax rows bwd")
    close(hip.gelu_bwd(x.to(DEV), dy.to(DEV)), E.gelu_bwd(x, dy), rtol=2e-5, what="gelu bwd")
    close(hip.colsum(x.to(DEV).view(b * t, e)), E.colsum(x.view(b * t, e)), rtol=2e-5, what="colsum")
    close(hip.mul(x.to(DEV), dy.to(DEV)), x * dy, rtol=1e-6); close(hip.add(x.to(DEV), dy.to(DEV)), x + dy, rtol=1e-6)


@pytest.mark.parametrize("T", [1024, 2048, 129, 4800])
def test_scores_topk_gather_scatter(hip, T):
    b, e, k = 2, 512, 128 if T >= 128 else 64
    feats = rnd(b, T, e, seed=T)
    tok = rnd(1, 1, e, seed=1)
    sc_ref = E.token_scores(feats, tok)
    sc = hip.token_scores(feats.to(DEV), tok.to(DEV))
    close(sc, sc_ref, rtol=1e-5, what="scores")
    idx = hip.topk(sc, k)
    idx_same_scores = E.topk(sc.cpu(), k)
    assert torch.equal(idx.cpu(), idx_same_scores), "top-k (sorted, ties by index) differs on identical scores"
    ref_top = torch.topk(sc_ref, k, dim=1).indices
    assert all(len(set(idx[i].tolist()) ^ set(ref_top[i].tolist())) <= 2 for i in range(b))
    # duplicates / ties
    tie = torch.zeros(1, 300); tie[0, 5] = 1.0
    assert hip.topk(tie.to(DEV), 4).cpu().tolist() == [[5, 0, 1, 2]]
    keep = (rnd(b, k, e, seed=3) > -0.8).float() * 1.1
    head = rnd(b, 1, e, seed=4)
    seq = hip.gather_tokens(feats.to(DEV), idx, head.to(DEV), keep.to(DEV), 1.0)
    close(seq, E.gather_tokens(feats, idx.cpu(), head, keep, 1.0), rtol=1e-6, what="gather")
    dseq = rnd(b, k + 1, e, seed=5)
    df, dh = torch.zeros(b, T, e, device=DEV), torch.zeros(1, 1, e, device=DEV)
    hip.gather_tokens_bwd(dseq.to(DEV), idx, keep.to(DEV), df, dh)
    df_r, dh_r = torch.zeros(b, T, e), torch.zeros(1, 1, e)
    E.gather_tokens_bwd(dseq, idx.cpu(), keep, df_r, dh_r)
    close(df, df_r, rtol=1e-6); close(dh, dh_r, rtol=1e-5)
    res = rnd(b, 2 * k + 2, e, seed=6)
    rows, gate = res[:, 1:k + 1], res[:, 0:1]
    resd = res.to(DEV)
    for g_ in (None, gate):
        out = hip.scatter_rows(feats.to(DEV), idx, resd[:, 1:k + 1], None if g_ is None else resd[:, 0:1])
        close(out, E.scatter_rows(feats, idx.cpu(), rows, g_), rtol=1e-6, what="scatter")
    scat = E.scatter_rows(feats, idx.cpu(), rows, None)
    dout = rnd(b, T, e, seed=7)
    a1, a2, a3 = hip.scatter_rows_bwd(dout.to(DEV), idx, scat.to(DEV), resd[:, 0:1], k)
    r1, r2, r3 = E.scatter_rows_bwd(dout, idx.cpu(), scat, gate, k)
    close(a1, r1, rtol=1e-6); close(a2, r2, rtol=1e-6); close(a3, r3, rtol=1e-4, what="dgate")


@pytest.mark.parametrize("c,size,patch", [(128, (8, 8, 8), (2, 2, 1)), (32, (16, 16, 16), (4, 2, 2)), (128, (20, 24, 20), (2, 2, 1))])
def test_window_token_reshapes(hip, c, size, patch):
    d, h, w = size
    x = rnd(2, d, h, w, c, seed=1)
    tok = hip.window_to_tokens(x.to(DEV), patch)
    assert torch.equal(tok.cpu(), E.window_to_tokens(x, patch))
    assert torch.equal(hip.tokens_to_window(tok, size, c, patch).cpu(), x)
    # against the reference's convert_dim on NCDHW
    from oracle import reference_model as rm
    assert torch.equal(tok.cpu(), rm.convert_dim(x.permute(0, 4, 1, 2, 3), patch))


@pytest.mark.parametrize("scale,lo", [(8, (4, 4, 4)), (4, (8, 8, 8)), (8, (5, 6, 5))])
def test_upsample_softmax_heads(hip, scale, lo):
    n = 2
    logit = torch.zeros(n, *lo, 4)
    logit[..., :2] = rnd(n, *lo, 2, seed=1) * 3
    p = hip.upsample_softmax(logit.to(DEV), 2, scale)
    p_ref = E.upsample_softmax(logit, 2, scale)
    close(p, p_ref, rtol=2e-6, atol=2e-6, what="upsample fwd")
    dprob = rnd(*p_ref.shape, seed=2)
    dl = hip.upsample_softmax_bwd(dprob.to(DEV), p, (n,) + lo, 2, scale, 4)
    close(dl, E.upsample_softmax_bwd(dprob, p_ref, (n,) + lo, 2, scale, 4), rtol=2e-5, what="upsample bwd")
    assert float(dl[..., 2:].abs().max()) == 0.0
    l4 = rnd(n, 6, 6, 6, 4, seed=3) * 2
    p4 = hip.channel_softmax(l4.to(DEV), 4)
    close(p4, E.channel_softmax(l4, 4), rtol=1e-6, atol=1e-7)
    d4 = rnd(n, 6, 6, 6, 4, seed=4)
    close(hip.channel_softmax_bwd(d4.to(DEV), p4), E.channel_softmax_bwd(d4, p4.cpu()), rtol=1e-5)


@pytest.mark.parametrize("scale,lo,codes", [(8, (4, 6, 4), (0, 1, 2, 3)), (4, (8, 8, 12), (0, 1, 2, 4, 5, 6, 7, 8)), (8, (16, 16, 16), (0, 1, 2, 3))])
def test_head_loss_fused_vs_unfused_chain(hip, scale, lo, codes):
    """Head -> loss fusion: Dice/CE sums, losses and the logit gradients of three sub-region maps computed straight from the
    low-resolution logits (cwf_head_loss_*) against (a) the kernel oracle and (b) the UNFUSED HIP chain upsample_softmax ->
    dice_ce -> backward, whose probabilities the fused kernels reproduce with identical arithmetic."""
    from cwf import functional as CF
    from utils import tools
    n = 2
    hi = tuple(v * scale for v in lo)
    logits = []
    for m in range(3):
        lg = torch.zeros(n, *lo, 4)
        lg[..., :2] = rnd(n, *lo, 2, seed=m + 1) * 4          # up to |8| logit difference: probabilities on both sides of the 0.005 clamp
        logits.append(lg)
    g = torch.Generator().manual_seed(5)
    label = torch.tensor(codes)[torch.randint(0, len(codes), (n,) + hi, generator=g)]
    masks = [tools.REGION_MASKS, tools.EDGE_MASKS][len(codes) > 4]
    pm = [masks[r] for r in ("01", "02", "04")]
    ld = [t.to(DEV) for t in logits]
    total, loss, coef = hip.head_loss(ld, label.to(DEV), pm, scale)
    t_e, l_e, c_e = E.head_loss(logits, label, pm, scale)
    close(loss, l_e, rtol=2e-6, what="fused losses"); close(total, t_e, rtol=2e-6); close(coef, c_e, rtol=1e-5, what="coef")
    gs = torch.tensor([0.7])
    dls = hip.head_loss_bwd(ld, label.to(DEV), pm, scale, coef, gs.to(DEV))
    dls_e = E.head_loss_bwd(logits, label, pm, scale, c_e, gs)
    for a, b in zip(dls, dls_e):
        close(a, b, rtol=5e-5, what="fused dlogit")
        assert float(a[..., 2:].abs().max()) == 0.0                      # pad channels written as zeros
    # (b) unfused HIP chain, through autograd, on the same logits
    leaves = [t.clone().requires_grad_(True) for t in ld]
    unf = sum(tools.dice_ce(CF.upsample_softmax(t, 2, scale).permute(0, 4, 1, 2, 3), label.to(DEV), 2, m) for t, m in zip(leaves, pm))
    (unf * 0.7).backward()
    assert abs(float(unf) - float(total)) <= 2e-6 * abs(float(unf))
    for a, t in zip(dls, leaves):
        close(a, t.grad, rtol=2e-5, what="fused vs unfused dlogit")
    # the lazy map: the losses take the fused route, anything else sees the real tensor
    lz = {r: CF.LazyProb(t.clone().requires_grad_(True), 2, scale) for r, t in zip(("01", "02", "04"), ld)}
    fused = (tools.get_separate_loss if len(codes) == 4 else tools.get_edge_separate_loss)(lz, label.to(DEV))
    assert abs(float(fused) - float(total)) <= 1e-6 * abs(float(total)) and all(z._t is None for z in lz.values())
    (fused * 0.7).backward()
    close(lz["02"].logit.grad, dls[1], rtol=1e-6)
    assert lz["01"].shape == (n, 2) + hi
    real = CF.upsample_softmax(ld[0], 2, scale).permute(0, 4, 1, 2, 3)
    assert torch.equal(lz["01"].detach(), real) and torch.equal(torch.argmax(lz["01"], dim=1), real.argmax(1)) and torch.equal(lz["01"][:, 1], real[:, 1])


def test_losses_against_reference_fixture(hip):
    """The HIP Dice/CE path on the reference's own golden losses (tests/golden/losses.npz, produced by the imported
    reference utils.tools / models.criterions)."""
    import os
    from conftest import GOLDEN
    from models import criterions
    from utils import tools
    g = np.load(os.path.join(GOLDEN, "losses.npz"))
    target, edge = torch.from_numpy(g["target"]).to(DEV), torch.from_numpy(g["edge"]).to(DEV)
    p4 = torch.from_numpy(g["p4"]).to(DEV).requires_grad_(True)
    l = criterions.softmax_dice(p4, target)
    l.backward()
    assert abs(float(l) - float(g["softmax_dice"])) < 2e-6
    close(p4.grad, torch.from_numpy(g["softmax_dice_grad"]), rtol=1e-5, what="softmax_dice grad")
    outs = {r: torch.from_numpy(g["p2_" + r]).to(DEV).requires_grad_(True) for r in ("01", "02", "04")}
    ls = tools.get_separate_loss(outs, target)
    ls.backward()
    assert abs(float(ls) - float(g["separate_loss"])) < 5e-6
    for r in outs:
        close(outs[r].grad, torch.from_numpy(g["sep_grad_" + r]), rtol=1e-5, what="sep grad")
        outs[r].grad = None
    le = tools.get_edge_separate_loss(outs, edge)
    le.backward()
    assert abs(float(le) - float(g["edge_separate_loss"])) < 5e-6
    for r in outs:
        close(outs[r].grad, torch.from_numpy(g["edge_grad_" + r]), rtol=1e-5, what="edge grad")


def test_loss_edge_cases(hip):
    """Empty classes (a class absent from a sample), probabilities on both sides of the 0.005 clamp."""
    from models import criterions
    n, s = 2, 8
    target = torch.zeros(n, s, s, s, dtype=torch.int64)
    target[1, :2] = 3
    prob = torch.full((n, 4, s, s, s), 0.25)
    prob[0, 0] = 0.994; prob[0, 1:] = 0.002
    from oracle import reference_model as rm
    pr = prob.clone().requires_grad_(True)
    lr = rm.softmax_dice(pr, target); lr.backward()
    pg = prob.to(DEV).requires_grad_(True)
    lg = criterions.softmax_dice(pg, target.to(DEV)); lg.backward()
    assert abs(float(lr) - float(lg)) < 2e-6
    close(pg.grad, pr.grad, rtol=1e-5, what="clamped grad")


def test_fused_adam_against_torch_fixture(hip):
    import os
    from conftest import GOLDEN
    from cwf.optim import FusedAdam
    g = np.load(os.path.join(GOLDEN, "adam.npz"))
    w = torch.nn.Parameter(torch.from_numpy(g["p0"]).to(DEV))
    w2 = torch.nn.Parameter(torch.from_numpy(g["p0"]).to(DEV).clone())
    opt = FusedAdam([w, w2], lr=2e-4, weight_decay=1e-5, amsgrad=True)
    for i in range(3):
        w.grad = torch.from_numpy(g["grads"][i]).to(DEV)
        w2.grad = torch.from_numpy(g["grads"][i]).to(DEV)
        opt.step()
        close(w, torch.from_numpy(g["traj"][i]), rtol=1e-6, atol=1e-7, what="adam step %d" % i)
    assert torch.equal(w, w2)
    st = opt.state[w]
    close(st["exp_avg"], torch.from_numpy(g["exp_avg"]), rtol=1e-6)
    close(st["max_exp_avg_sq"], torch.from_numpy(g["max_exp_avg_sq"]), rtol=1e-6)
    sd = opt.state_dict()     # torch.optim.Adam-compatible layout (train_no_amp.py:252 'optim_dict')
    assert set(sd["state"][0].keys()) >= {"step", "exp_avg", "exp_avg_sq", "max_exp_avg_sq"}


def test_fused_adam_unsynced_steps_and_state_reload(hip):
    """(a) 40 optimizer steps enqueued back to back with NO host sync, the learning rate changing every step (poly schedule):
    the result must equal torch.optim.Adam's -- lr / bias corrections travel as kernel arguments, so a host that runs ahead
    of the GPU cannot corrupt a step still in flight.  (b) state_dict -> fresh FusedAdam.load_state_dict -> more steps equals
    the uninterrupted torch run (the descriptor table is rebuilt over the loaded moments, the step counter continues)."""
    from cwf.optim import FusedAdam, poly_lr
    gen = torch.Generator().manual_seed(11)
    shapes = [(1 << 20,), (513, 7), (5,)]
    p0 = [torch.randn(s, generator=gen) for s in shapes]
    grads = [[torch.randn(s, generator=gen) * (0.1 + 0.05 * i) for s in shapes] for i in range(60)]
    ref = [torch.nn.Parameter(t.clone()) for t in p0]
    ropt = torch.optim.Adam(ref, lr=2e-4, weight_decay=1e-5, amsgrad=True)
    for i in range(60):
        ropt.param_groups[0]["lr"] = float(poly_lr(2e-4, i, 100))
        for p, g in zip(ref, grads[i]):
            p.grad = g.clone()
        ropt.step()
        if i == 39:
            ref40 = [p.detach().clone() for p in ref]
    ours = [torch.nn.Parameter(t.clone().to(DEV)) for t in p0]
    dg = [[g.to(DEV) for g in gs] for gs in grads]
    opt = FusedAdam(ours, lr=2e-4, weight_decay=1e-5, amsgrad=True)
    torch.cuda.synchronize()
    for i in range(40):                                           # no sync inside
        opt.param_groups[0]["lr"] = float(poly_lr(2e-4, i, 100))
        for p, g in zip(ours, dg[i]):
            p.grad = g
        opt.step()
    torch.cuda.synchronize()
    for p, r in zip(ours, ref40):
        close(p, r, rtol=2e-6, atol=2e-7, what="40 unsynced steps")
    import copy
    sd = copy.deepcopy(opt.state_dict())           # (a live state_dict shares its 'step' tensors with the optimizer)
    opt2 = FusedAdam(ours, lr=2e-4, weight_decay=1e-5, amsgrad=True)
    opt2.load_state_dict(copy.deepcopy(sd))
    assert opt2._steps == 40
    for i in range(40, 60):
        opt2.param_groups[0]["lr"] = float(poly_lr(2e-4, i, 100))
        for p, g in zip(ours, dg[i]):
            p.grad = g
        opt2.step()
    for p, r in zip(ours, ref):
        close(p, r, rtol=3e-6, atol=3e-7, what="resume after load_state_dict")
    # loading AFTER the first step of an optimizer must also retarget the kernel at the new moment tensors
    opt2.load_state_dict(copy.deepcopy(sd))
    assert opt2._steps == 40 and opt2._table is None


def test_dropout_mask_kernel(hip):
    """K12 fused keep-mask: values in {0, 1/(1-p)}, drop rate ~ p, independent second mask, reproducible per (seed, step, site offset) of the device generator state."""
    torch.manual_seed(123)
    hip._site = 0
    m = hip.dropout_mask((4, 8, 129, 129), 0.1, DEV).cpu()
    vals = torch.unique(m)
    assert len(vals) == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1 / 0.9) < 1e-6
    assert abs(float((m == 0).float().mean()) - 0.1) < 5e-3
    m2 = hip.dropout_mask((4, 8, 129, 129), 0.1, DEV, p2=0.2).cpu()
    assert abs(float((m2 == 0).float().mean()) - (1 - 0.9 * 0.8)) < 5e-3 and abs(float(m2.mean()) - 1.0) < 1e-2
    assert not torch.equal(m != 0, m2 != 0)
    hip._site = 0                      # same (seed, step, site offset) -> same mask
    assert torch.equal(hip.dropout_mask((4, 8, 129, 129), 0.1, DEV).cpu(), m)
    # neighbouring elements are uncorrelated (lag-1 autocorrelation of the keep bits)
    k = (m.reshape(-1) != 0).float(); k = k - k.mean()
    assert abs(float((k[1:] * k[:-1]).mean() / (k * k).mean())) < 1e-2
