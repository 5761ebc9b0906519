"""GPU: the round-2 fused coupler kernels (one-launch attention forward / backward, paired LayerNorm, the extended GEMM, the
write-not-accumulate token kernels, counter-based dropout) through the C ABI against the CPU kernel oracle
(oracle/kernel_emul.py), and the whole RegionCouplerFn / FusionCouplerFn on HIP against the same Functions run over the oracle.
Dropout is compared EXACTLY: the oracle evaluates the same counter-based generator (seed, step, site offset + element index)."""
import math

import numpy as np
import pytest
import torch

from oracle.kernel_emul import EmulBackend

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + int(np.prod(shape)) % 9973)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def close(got, ref, rtol=2e-5, what=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = float((got - ref).abs().max())
    assert err <= 2 * rtol * float(ref.abs().max() + 1e-30), (what, err, float(ref.abs().max()))


def _emul_synced(hip):
    """An oracle backend whose generator state equals the device's {seed, step}."""
    E = EmulBackend()
    seed, step = hip.rng(DEV).cpu().tolist()
    E.set_rng(seed, step)
    return E


def test_gemm_ex_linear_forms(hip):
    E = _emul_synced(hip)
    m, k, n = 516, 512, 1536
    x, x2 = rnd(m, k, seed=1), rnd(m, k, seed=2)
    w = rnd(n, k, seed=3) / math.sqrt(k)
    bias = rnd(n, seed=4) * 0.1
    # q | kv in one launch: columns >= 512 read the second operand
    out_h = torch.empty(m, n, device=DEV); out_e = torch.empty(m, n)
    hip.linear_fwd(x.to(DEV), w.to(DEV), None, out_h, x2=x2.to(DEV), split_n=512)
    E.linear_fwd(x, w, None, out_e, x2=x2, split_n=512)
    close(out_h, out_e, what="split_n")
    assert float((out_e[:, 512:] - x2 @ w[512:].t()).abs().max()) < 1e-5
    # bias + GELU + pre-activation + dropout + residual, 512 x 512 (32x32 tiles) and the two-stage mask
    w2 = w[:512].contiguous()
    res = rnd(m, 512, seed=5)
    for act, drop in ((1, (1000, 0.1, 0.0)), (0, (77, 0.1, 0.2)), (0, None)):
        oh, ph = torch.empty(m, 512, device=DEV), torch.empty(m, 512, device=DEV)
        oe, pe = torch.empty(m, 512), torch.empty(m, 512)
        hip.linear_fwd(x.to(DEV), w2.to(DEV), bias[:512].to(DEV), oh, act=act, pre=ph, drop=drop, residual=res.to(DEV))
        E.linear_fwd(x, w2, bias[:512], oe, act=act, pre=pe, drop=drop, residual=res)
        close(ph, pe, what="pre-activation")
        if drop:                                        # identical keep pattern, not just statistics
            zero_h = (oh.cpu() - res).abs() < 1e-12
            zero_e = (oe - res).abs() < 1e-12
            assert torch.equal(zero_h, zero_e) and 0.05 < float(zero_e.float().mean()) < 0.4
        close(oh, oe, what="epilogue act=%d drop=%s" % (act, drop))
    # data gradient with the dropout of dy recomputed; column-slice views of a [m, 1536] matrix
    dy = rnd(m, n, seed=6)
    dyd = dy.to(DEV)
    close(hip.linear_dgrad(dyd[:, 512:], w.to(DEV)[512:]), E.linear_dgrad(dy[:, 512:], w[512:]), what="dgrad slice")
    d5 = dy[:, :512].contiguous()
    close(hip.linear_dgrad(d5.to(DEV), w2.to(DEV), drop=(5, 0.1, 0.2)), E.linear_dgrad(d5, w2, drop=(5, 0.1, 0.2)), what="dgrad drop")
    # weight gradient: operand switch on rows, bias gradient by rowsum, accumulate
    dw_h, dw_e = torch.empty(n, k, device=DEV), torch.empty(n, k)
    hip.linear_wgrad(dyd, x.to(DEV), dw_h, None, x2=x2.to(DEV), split_m=512)
    E.linear_wgrad(dy, x, dw_e, None, x2=x2, split_m=512)
    close(dw_h, dw_e, rtol=5e-5, what="wgrad split_m")
    db_h, db_e = torch.empty(512, device=DEV), torch.empty(512)
    dw_h, dw_e = torch.empty(512, k, device=DEV), torch.empty(512, k)
    for acc in (False, True):
        hip.linear_wgrad(d5.to(DEV), x.to(DEV), dw_h, db_h, accumulate=acc, drop=(9, 0.1, 0.0))
        E.linear_wgrad(d5, x, dw_e, db_e, accumulate=acc, drop=(9, 0.1, 0.0))
        close(dw_h, dw_e, rtol=5e-5, what="wgrad acc=%s" % acc)
        close(db_h, db_e, rtol=5e-5, what="bias grad acc=%s" % acc)
    # small M (258 rows): the fusion coupler's shape
    xs = x[:258].contiguous()
    oh, oe = torch.empty(258, 512, device=DEV), torch.empty(258, 512)
    hip.linear_fwd(xs.to(DEV), w2.to(DEV), bias[:512].to(DEV), oh)
    E.linear_fwd(xs, w2, bias[:512], oe)
    close(oh, oe, what="M=258")


@pytest.mark.parametrize("rows,perm_T", [(516, 129), (258, 0), (516, 0)])
def test_paired_layernorm(hip, rows, perm_T):
    E = EmulBackend()
    e = 512
    x, x2 = rnd(rows, e, seed=1) * 2 + 0.3, rnd(rows, e, seed=2)
    g1, b1, g2, b2 = rnd(e, seed=3) * 0.1 + 1, rnd(e, seed=4) * 0.1, rnd(e, seed=5) * 0.1 + 1, rnd(e, seed=6) * 0.1
    D = lambda *ts: [None if t is None else t.to(DEV) for t in ts]
    second = x if perm_T else x2                        # perm_T > 0: "a attends b, b attends a" -- the second operand is x swapped
    ya, yb, st = hip.ln_pair_fwd(*D(x, second), perm_T, *D(g1, b1, g2, b2))
    ra, rb, rs = E.ln_pair_fwd(x, second, perm_T, g1, b1, g2, b2)
    close(ya, ra, what="ya"); close(yb, rb, what="yb"); close(st, rs, rtol=1e-5, what="stats")
    ref = torch.nn.functional.layer_norm(second[E._perm(rows, perm_T)], (e,), g2, b2)
    close(yb, ref, what="yb vs F.layer_norm")
    dy, da, db = rnd(rows, e, seed=7), rnd(rows, e, seed=8), rnd(rows, e, seed=9)
    for dual in ((False,) if perm_T else (True, False)):
        for acc in (False, True):
            gh = [torch.full((e,), 0.5, device=DEV) for _ in range(4)]
            ge = [torch.full((e,), 0.5) for _ in range(4)]
            x2_ = x2 if dual else x
            _, _, st_h = hip.ln_pair_fwd(*D(x, x2_), perm_T, *D(g1, b1, g2, b2))
            _, _, st_e = E.ln_pair_fwd(x, x2_, perm_T, g1, b1, g2, b2)
            dxh, dx2h = hip.ln_pair_bwd(*D(dy, da, db, x, x2_), perm_T, *D(g1, g2), st_h, *gh, accumulate=acc, want_dx2=dual)
            dxe, dx2e = E.ln_pair_bwd(dy, da, db, x, x2_, perm_T, g1, g2, st_e, *ge, accumulate=acc, want_dx2=dual)
            close(dxh, dxe, rtol=5e-5, what="dx dual=%s" % dual)
            if dual:
                close(dx2h, dx2e, rtol=5e-5, what="dx2")
            for a, b_ in zip(gh, ge):
                close(a, b_, rtol=5e-5, what="LN param grads acc=%s" % acc)
    # single LayerNorm (the FFN's PreNorm) against autograd
    xr = x.clone().requires_grad_(True); gr = g1.clone().requires_grad_(True); br = b1.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (e,), gr, br).backward(da)
    gh = [torch.empty(e, device=DEV) for _ in range(2)]
    _, _, st1 = hip.ln_pair_fwd(x.to(DEV), None, 0, g1.to(DEV), b1.to(DEV), None, None)
    dx, _ = hip.ln_pair_bwd(dy.to(DEV), da.to(DEV), None, x.to(DEV), None, 0, g1.to(DEV), None, st1, gh[0], gh[1], None, None,
                            accumulate=False, want_dx2=False)
    close(dx, xr.grad + dy, rtol=5e-5, what="single dx vs autograd")
    close(gh[0], gr.grad, rtol=5e-5); close(gh[1], br.grad, rtol=5e-5)


@pytest.mark.parametrize("z,t,p", [(4, 129, 0.0), (4, 129, 0.1), (2, 129, 0.1), (3, 65, 0.0), (1, 144, 0.1)])
def test_attention_one_launch(hip, z, t, p):
    E = _emul_synced(hip)
    e, heads = 512, 8
    qkv = rnd(z * t, 3 * e, seed=1)
    qkv[:, :e] *= 3.0                                    # peaked softmax rows as well as flat ones
    drop = (12345, p) if p > 0 else None
    o_h = hip.attn_fwd(qkv.to(DEV), z, t, heads, drop)
    o_e = E.attn_fwd(qkv, z, t, heads, drop)
    close(o_h, o_e, what="attention forward")
    d_o = rnd(z * t, e, seed=2)
    g_h = hip.attn_bwd(qkv.to(DEV), d_o.to(DEV), z, t, heads, drop)
    g_e = E.attn_bwd(qkv, d_o, z, t, heads, drop)
    close(g_h[:, :e], g_e[:, :e], rtol=1e-4, what="dq")
    close(g_h[:, e:2 * e], g_e[:, e:2 * e], rtol=1e-4, what="dk")
    close(g_h[:, 2 * e:], g_e[:, 2 * e:], rtol=1e-4, what="dv")
    if p == 0.0:                                          # the reference restatement of DualSelfAttention's core
        hd = e // heads
        q, k, v = (qkv[:, i * e:(i + 1) * e].reshape(z, t, heads, hd).permute(0, 2, 1, 3) for i in range(3))
        att = (torch.einsum("bhxd,bhyd->bhxy", q, k) * (hd ** -0.5)).softmax(dim=-1)        # SelfAttention.py:94-96
        ref = torch.einsum("bhxy,bhyd->bhxd", att, v).permute(0, 2, 1, 3).reshape(z * t, e)  # :98
        close(o_h, ref, what="vs reference formula")


@pytest.mark.parametrize("T", [1024, 2048, 4800, 128])
def test_round2_token_kernels(hip, T):
    E = _emul_synced(hip)
    b, e, k = 2, 512, 128
    feats = rnd(b, T, e, seed=T)
    q1, q2 = rnd(1, 1, e, seed=1), rnd(1, 1, e, seed=2)
    fd = feats.to(DEV)
    s1, s2 = hip.token_scores2(fd, q1.to(DEV), q2.to(DEV))
    r1, r2 = E.token_scores2(feats, q1, q2)
    close(s1, r1, rtol=1e-5); close(s2, r2, rtol=1e-5)
    qb = rnd(b, 1, e, seed=3)                               # per-sample query (the fusion coupler's class token)
    close(hip.token_scores2(fd, qb.to(DEV), None)[0], E.token_scores2(feats, qb, None)[0], rtol=1e-5)
    i0, v0, i1, v1 = hip.topk_inv(s1, s2, k)
    e0, w0, e1, w1 = E.topk_inv(s1.cpu(), s2.cpu(), k)     # same scores -> identical sorted selections and inverse maps
    assert torch.equal(i0.cpu(), e0) and torch.equal(i1.cpu(), e1) and torch.equal(v0.cpu(), w0) and torch.equal(v1.cpu(), w1)
    fi, fv = hip.index_inv(e1.to(DEV).long(), T)
    assert torch.equal(fv.cpu(), w1) and torch.equal(fi.cpu(), e1)
    # NaN scores (a diverged step) order as largest and never produce an out-of-range index
    sn = s1.clone(); sn[0, 7] = float("nan"); sn[1, T - 1] = float("nan")
    n0, nv, _, _ = hip.topk_inv(sn, None, k)
    assert int(n0.min()) >= 0 and int(n0.max()) < T and n0[0, 0].item() == 7 and n0[1, 0].item() == T - 1
    assert int((nv >= 0).sum()) == b * k
    # four gathers in one launch, dropout recomputed
    X1h, X2h = torch.empty(b, 2, k + 1, e, device=DEV), torch.empty(b, 2, k + 1, e, device=DEV)
    X1e, X2e = torch.empty(b, 2, k + 1, e), torch.empty(b, 2, k + 1, e)
    for p in (0.0, 0.1):
        hip.gather_multi([(fd, i0, q1.to(DEV), X1h[:, 0], 0), (fd, i1, q2.to(DEV), X1h[:, 1], 10 ** 6), (fd, i1, qb.to(DEV), X2h[:, 0], 2 * 10 ** 6),
                          (fd, i0, q1.to(DEV), X2h[:, 1], 3 * 10 ** 6)], k, e, p=p)
        E.gather_multi([(feats, e0, q1, X1e[:, 0], 0), (feats, e1, q2, X1e[:, 1], 10 ** 6), (feats, e1, qb, X2e[:, 0], 2 * 10 ** 6),
                        (feats, e0, q1, X2e[:, 1], 3 * 10 ** 6)], k, e, p=p)
        close(X1h, X1e, rtol=1e-6, what="gather p=%s" % p); close(X2h, X2e, rtol=1e-6)
    # scatter (+gate, + un-gated copy) through the inverse map
    R = rnd(b, 2, k + 1, e, seed=9)
    Rd = R.to(DEV)
    gh, sh = hip.scatter_inv(fd, v0, Rd[:, 1, 1:], Rd[:, 1, 0:1], want_gated=True, want_scat=True)
    ge, se = E.scatter_inv(feats, w0, R[:, 1, 1:], R[:, 1, 0:1], want_gated=True, want_scat=True)
    close(gh, ge, rtol=1e-6); close(sh, se, rtol=1e-6)
    assert torch.equal(se, E.scatter_rows(feats, e0, R[:, 1, 1:]))            # == the round-1 scatter (reference :463-468)
    # its backward into the coupler's output gradient
    dg, dsc, dtok = rnd(b, T, e, seed=10), rnd(b, T, e, seed=11), rnd(b, 1, e, seed=12)
    for use_g, use_s, use_x in ((True, True, True), (True, False, False), (False, True, False), (False, False, True)):
        dRh = torch.full((b, 2, k + 1, e), 7.0, device=DEV); dRe = torch.full((b, 2, k + 1, e), 7.0)
        hip.scatter_bwd(dg.to(DEV) if use_g else None, dsc.to(DEV) if use_s else None, fd, v0, i0, Rd[:, 1, 1:], Rd[:, 1, 0:1],
                        dtok.to(DEV) if use_x else None, dRh[:, 1, 1:], dRh[:, 1, 0:1])
        E.scatter_bwd(dg if use_g else None, dsc if use_s else None, feats, w0, e0, R[:, 1, 1:], R[:, 1, 0:1], dtok if use_x else None,
                      dRe[:, 1, 1:], dRe[:, 1, 0:1])
        close(dRh, dRe, rtol=5e-5, what="scatter_bwd %s" % ((use_g, use_s, use_x),))
    # token-matrix gradient from its three uses; against autograd of the round-1 op chain
    dX1, dX2 = rnd(b, 2, k + 1, e, seed=13), rnd(b, 2, k + 1, e, seed=14)
    for p in (0.0, 0.1):
        th = hip.token_grad(dg.to(DEV), dsc.to(DEV), Rd[:, 1, 0:1], v0, v1, dX1.to(DEV)[:, 0], dX2.to(DEV)[:, 1], k, p, 111, 999999)
        te = E.token_grad(dg, dsc, R[:, 1, 0:1], w0, w1, dX1[:, 0], dX2[:, 1], k, p, 111, 999999)
        close(th, te, rtol=1e-6, what="token_grad p=%s" % p)
    fr = feats.clone().requires_grad_(True)
    seq_a = E.gather_tokens(fr, e0, q1)
    seq_b = E.gather_tokens(fr, e1, q2)
    scat = E.scatter_rows(fr, e0, R[:, 1, 1:])
    ((seq_a * dX1[:, 0]).sum() + (seq_b * dX2[:, 1]).sum() + (scat * R[:, 1, 0:1] * dg).sum() + (scat * dsc).sum()).backward()
    th = hip.token_grad(dg.to(DEV), dsc.to(DEV), Rd[:, 1, 0:1], v0, v1, dX1.to(DEV)[:, 0], dX2.to(DEV)[:, 1], k)
    close(th, fr.grad, rtol=1e-6, what="token_grad vs autograd")
    th1 = hip.token_grad(None, None, None, v0, None, dX1.to(DEV)[:, 0], None, k)
    close(th1, E.token_grad(None, None, None, w0, None, dX1[:, 0], None, k), rtol=1e-6)
    o1, o2 = hip.head_grad(dX1.to(DEV)[:, 0, 0], dX2.to(DEV)[:, 1, 0], dX1.to(DEV)[:, 1, 0], dX2.to(DEV)[:, 0, 0])
    close(o1, (dX1[:, 0, 0] + dX2[:, 1, 0]).sum(0).reshape(1, 1, e), rtol=1e-6)
    close(o2, (dX1[:, 1, 0] + dX2[:, 0, 0]).sum(0).reshape(1, 1, e), rtol=1e-6)
    close(hip.add3(fd, fd * 2, fd * 0.5), feats * 3.5, rtol=1e-6)


def _weights(seed):
    e = 512
    shapes = [(e,), (e,), (e,), (e,), (e, e), (e,), (3 * e, e), (e,), (e,), (e, e), (e,), (e, e), (e,)]
    out = []
    for i, s in enumerate(shapes):
        w = rnd(*s, seed=seed + i) * (1.0 / math.sqrt(e) if len(s) == 2 else 0.1)
        if len(s) == 1 and i in (0, 2, 7):
            w = w + 1.0
        out.append(w)
    return out


@pytest.mark.parametrize("training", [False, True])
def test_region_and_fusion_coupler_functions(hip, training):
    """The whole Functions, HIP vs the same host code over the oracle backend: outputs and every gradient (token matrices, class
    tokens, the 13 weights of each of the THREE sub-regions' weight sets -- one grouped launch per stage covers all three).
    Selections are teacher-forced from the oracle run (a near-tie flip would change everything); training=True switches every
    dropout site on (p = 0.1) with the generator states synchronised."""
    from cwf import coupler as CP, kernels
    G, b, te, ts, e = 3, 2, 2048, 1024, 512
    Em, Sm = rnd(G, b, te, e, seed=1), rnd(G, b, ts, e, seed=2)
    etoks = [rnd(1, 1, e, seed=3 + g) * 0.05 for g in range(G)]
    stoks = [rnd(1, 1, e, seed=13 + g) * 0.05 for g in range(G)]
    W = [w for g in range(G) for w in _weights(100 + 50 * g)]
    names = [tuple("r%d_%s" % (g, n) for n in ("edge", "sem_supp", "sem", "edge_supp")) for g in range(G)]
    douts = [rnd(G, b, te, e, seed=5), rnd(G, b, ts, e, seed=6), rnd(G, b, ts, e, seed=7), rnd(G, b, 1, e, seed=8)]

    def run(on_hip, forced):
        if on_hip:
            kernels._set_backend_for_testing(hip)          # the fixture's instance: its generator state is what the oracle copies
            hip.begin_step(DEV)
            dev = DEV
        else:
            seed, step = hip.rng(DEV).cpu().tolist()
            K = EmulBackend(); K.set_rng(seed, step + (0 if forced is not None else 1)); K._site = 0
            kernels._set_backend_for_testing(K)
            dev = "cpu"
        leaves = [t.clone().to(dev).requires_grad_(True) for t in [Em, Sm] + etoks + stoks + W]
        cfg = CP.CouplerConfig(8, 128, training, 0.1, 0.1, 0.1, 0.1, forced={k: v.to(dev) for k, v in (forced or {}).items()}, names=names, groups=G)
        out = CP.RegionCouplerFn.apply(cfg, *leaves)
        sum((o * d.to(dev)).sum() for o, d in zip(out[:4], douts)).backward()
        return out, [t.grad for t in leaves]

    try:
        oe, ge = run(False, None)                       # oracle first: its selections are then forced on both sides
        forced = {names[g][j]: oe[4 + j][g * b:(g + 1) * b].long() for g in range(G) for j in range(4)}
        oh, gh = run(True, forced)
        oe, ge = run(False, forced)
    finally:
        kernels._set_backend_for_testing(hip)
    for a, r, nm in zip(oh[:4], oe[:4], ("gated_e", "gated_s", "scat_s", "sem_tok")):
        close(a, r, rtol=1e-4, what=nm)
    wn = ["ln1.w", "ln1.b", "ln2.w", "ln2.b", "out.w", "out.b", "qkv.w", "ffn.ln.w", "ffn.ln.b", "w1", "b1", "w2", "b2"]
    gn = ["dE", "dS"] + ["d e_tok%d" % g for g in range(G)] + ["d s_tok%d" % g for g in range(G)] + ["set%d %s" % (g, n) for g in range(G) for n in wn]
    assert len(gn) == len(gh)
    for a, r, nm in zip(gh, ge, gn):
        close(a, r, rtol=5e-4, what=nm)

    # fusion coupler (per-sample class token, self-attention)
    feats, tok = rnd(b, ts, e, seed=21), rnd(b, 1, e, seed=22) * 0.05
    dfused = rnd(b, ts, e, seed=23)

    def runf(on_hip, forced):
        if on_hip:
            kernels._set_backend_for_testing(hip)
            hip.begin_step(DEV)
            dev = DEV
        else:
            seed, step = hip.rng(DEV).cpu().tolist()
            K = EmulBackend(); K.set_rng(seed, step + (0 if forced is not None else 1)); K._site = 0
            kernels._set_backend_for_testing(K)
            dev = "cpu"
        leaves = [t.clone().to(dev).requires_grad_(True) for t in [feats, tok] + W[:13]]
        cfg = CP.CouplerConfig(8, 128, training, 0.1, 0.1, 0.1, 0.1, forced={k: v.to(dev) for k, v in (forced or {}).items()}, names=[("fusion",)])
        fused, idx = CP.FusionCouplerFn.apply(cfg, *leaves)
        (fused * dfused.to(dev)).sum().backward()
        return fused, idx, [t.grad for t in leaves]

    try:
        _, idx, _ = runf(False, None)
        fh, _, gh = runf(True, {"fusion": idx.long()})
        fe, _, ge = runf(False, {"fusion": idx.long()})
    finally:
        kernels._set_backend_for_testing(hip)
    close(fh, fe, rtol=1e-4, what="fused")
    for a, r, nm in zip(gh, ge, ["dfeats", "dtok"] + wn):
        close(a, r, rtol=5e-4, what="fusion " + nm)


def test_grouped_window_kernels_and_fused_decoupler_conv(hip):
    """(a) window <-> token reshapes of three channel groups in one launch == the per-group kernels; cat3 = adjoint of the channel
    split.  (b) FusedConvSpec: conv_semantic_{1,2,4} / conv_mid_fea_{1,2,4} (three convs on one input) as ONE conv with three
    separate parameter tensors -- forward, data gradient and weight / bias gradients against the three separate convs."""
    from cwf import functional as CF, kernels, packing as pk
    E = EmulBackend()
    x = rnd(2, 8, 8, 8, 3 * 32, seed=1)
    tok = hip.window_to_tokens_g(x.to(DEV), 3, (4, 2, 2))
    assert torch.equal(tok.cpu(), E.window_to_tokens_g(x, 3, (4, 2, 2)))
    assert torch.equal(hip.tokens_to_window_g(tok, (8, 8, 8), 32, (4, 2, 2)).cpu(), x)
    parts = [rnd(2, 4, 4, 4, 16, seed=s_) for s_ in (2, 3)]
    y = hip.cat3_channels([parts[0].to(DEV), None, parts[1].to(DEV)], (2, 4, 4, 4, 16), DEV).cpu()
    assert torch.equal(y[..., :16], parts[0]) and float(y[..., 16:32].abs().max()) == 0 and torch.equal(y[..., 32:], parts[1])
    for prec, tol in (("fp32", 2e-5), ("bf16x3", 2e-4)):
        kernels.set_precision(prec)
        try:
            for cin, cout, size in ((96, 32, (8, 8, 8)), (256, 128, (4, 6, 4))):
                convs = [torch.nn.Conv3d(cin, cout, 3, padding=1).to(DEV) for _ in range(3)]
                spec = CF.FusedConvSpec(pk.CONV3_S1, cin, cout)
                packer = CF.WeightPacker()
                single = []
                from models.clswiseformer.layers import HipConv
                for c in convs:
                    h = HipConv(cin, cout).to(DEV)
                    h.weight.data.copy_(c.weight.data); h.bias.data.copy_(c.bias.data)
                    single.append(h)
                    packer.add(h.spec, h.weight)
                packer.add_fused(spec, [h.weight for h in single], [h.bias for h in single])
                packer.refresh()
                xin = rnd(2, *size, cin, seed=9).to(DEV).requires_grad_(True)
                y, (sc, sh) = CF.fused_conv3(xin, single, spec)
                dy = rnd(*y.shape, seed=10).to(DEV)
                y.backward(dy)
                gx = xin.grad.clone(); gw = [h.weight.grad.clone() for h in single]; gb = [h.bias.grad.clone() for h in single]
                xin.grad = None
                for h in single:
                    h.weight.grad = None; h.bias.grad = None
                ys = [h(xin, want_stats=True) for h in single]
                yref = torch.cat([t[0] for t in ys], -1)
                close(y, yref, rtol=tol, what="fused forward")
                close(sc, torch.cat([t[1][0] for t in ys], -1), rtol=10 * tol, what="fused stats")
                yref.backward(dy)
                close(gx, xin.grad, rtol=tol, what="fused dgrad")
                for g in range(3):
                    close(gw[g], single[g].weight.grad, rtol=5 * tol, what="fused wgrad %d" % g)
                    close(gb[g], single[g].bias.grad, rtol=5 * tol, what="fused bias grad %d" % g)
        finally:
            kernels.set_precision("fp32")


@pytest.mark.parametrize("training", [False, True])
def test_transformer_module_forwards_vs_reference(hip, training):
    """The coupler modules called on their own, with the reference's signatures (ClsWiseTransformer.py:41-55,
    FusionClsWiseTransformer.py:43-54): TwoClsWiseTransformerModel(edge, sem_supp, sem, edge_supp) -> [B,258,512] and
    FusionClsWiseTransformerModel(x) -> [B,129,512] run the fused block launches (cwf.coupler.IntraCouplerBlockFn / FusionBlockFn);
    outputs and every gradient (inputs + the 13 parameters) against torch autograd through the oracle's restatement.  Training mode
    with the dropout rates zeroed (masks are device-generated; the dropout sites themselves are covered by
    test_region_and_fusion_coupler_functions)."""
    from models.clswiseformer.transformer import TwoClsWiseTransformerModel, FusionClsWiseTransformerModel
    from oracle import reference_model as rm
    torch.manual_seed(5)
    b, t, e = 2, 129, 512
    for cls, nin in ((TwoClsWiseTransformerModel, 4), (FusionClsWiseTransformerModel, 1)):
        m = cls(1, 8, e, 0.0, 0.0).to(DEV)
        m.train(training)
        xs = [(torch.randn(b, t, e) * 0.5) for _ in range(nin)]
        xd = [x.to(DEV).requires_grad_(True) for x in xs]
        out = m(*xd)
        wgt = torch.linspace(-1, 1, out.numel(), device=DEV).reshape(out.shape)
        (out * wgt).sum().backward()
        # oracle: same parameters under the reference's key names, plain torch autograd
        p = {"T." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
        xr = [x.clone().requires_grad_(True) for x in xs]
        ref = rm.intra_region_coupler(p, "T", *xr) if nin == 4 else rm.cross_region_coupler(p, "T", xr[0])
        (ref * wgt.cpu()).sum().backward()
        assert out.shape == ref.shape
        assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 2e-4 * float(ref.detach().abs().max())
        for a_, r_ in zip(xd, xr):
            assert float((a_.grad.cpu() - r_.grad).norm() / r_.grad.norm()) < 2e-4
        for k, v in m.named_parameters():
            g_ref = p["T." + k].grad
            assert v.grad is not None and float((v.grad.cpu() - g_ref).norm() / (g_ref.norm() + 1e-30)) < 5e-4, k
