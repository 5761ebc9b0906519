"""CPU: the package's module tree + autograd glue, driven through the kernel EMULATION (oracle/kernel_emul.py),
against the oracle model and the golden fixtures.  This checks everything except the HIP kernels themselves
(those are compared with the same emulation in the -m gpu tests)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import reference_model as rm
from utils import synthetic as syn


def _model():
    from models.clswiseformer.cls_wise_former import get_cls_wise_former
    m = get_cls_wise_former(dataset="brats", _conv_repr=True, _pe_type="fixed")
    m.load_state_dict(syn.det_state_dict(rm.param_shapes()), strict=False)
    m.Unet_list.InitConv.dropout = 0.0
    return m


def test_state_dict_layout_matches_reference():
    from models.clswiseformer.cls_wise_former import get_cls_wise_former
    m = get_cls_wise_former(dataset="brats", _conv_repr=True, _pe_type="fixed")
    sd = m.state_dict()
    assert list(sd.keys()) == [n for n, _, _ in rm.param_shapes()]
    for n, shp, _ in rm.param_shapes():
        assert tuple(sd[n].shape) == tuple(shp), n
    assert torch.equal(sd["fusion_label_pos.pe"], rm.fixed_pe_table())
    assert sum(p.numel() for p in m.parameters()) == 16824556
    # checkpoint layout of train_no_amp.py:248-253 ('module.' prefix from the DDP wrapper) loads through DataParallel-style prefixing
    wrapped = {"module." + k: v for k, v in sd.items()}
    m.load_state_dict({k[len("module."):]: v for k, v in wrapped.items()})


def test_factory_default_pe_type_constructs_like_the_reference_and_cannot_run():
    """get_cls_wise_former() with its OWN default _pe_type='learned' (cls_wise_former.py:757-780): the reference builds
    LearnedPositionalEncoding(129, 512) -- a [1,512,129] parameter its forward cannot add to [B,128,512] rows.  Same here:
    construction works with the reference's parameter names / shapes, forward raises."""
    from models.clswiseformer.cls_wise_former import get_cls_wise_former
    m = get_cls_wise_former()
    sd = m.state_dict()
    assert tuple(sd["label_01_position_encoding.position_embeddings"].shape) == (1, 512, 129)
    assert "label_01_position_encoding.pe" not in sd and "fusion_label_pos.pe" in sd
    with pytest.raises(Exception):
        m(torch.zeros(1, 4, 64, 64, 64), None)


def test_forward_needs_hip_library_or_gpu():
    """The product path must fail loudly without the HIP backend (no CPU fallback)."""
    from cwf import kernels, _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    old = kernels._backend
    kernels._set_backend_for_testing(None)
    try:
        with pytest.raises(_lib.CwfError):
            _model()(torch.zeros(1, 4, 64, 64, 64), None)
    finally:
        kernels._set_backend_for_testing(old)


def test_model_and_losses_64_vs_oracle_and_golden(emul_backend):
    from models import criterions
    from utils import tools
    g = np.load(os.path.join(GOLDEN, "model_64.npz"))
    m = _model().eval()
    m.collect_aux = True
    x, target, edge = syn.synthetic_batch([0], (64, 64, 64))
    outs = m(x, None)
    assert outs[0].shape == (1, 4, 64, 64, 64) and outs[1]["01"].shape == (1, 2, 64, 64, 64)
    prob = outs[0].reshape(-1)
    assert np.allclose(prob[g["prob_sample_idx"]].detach().numpy(), g["prob_sample"], atol=5e-6)
    for k in ("01_edge", "02_sem", "04_edge_supp", "fusion"):
        assert set(m.aux[k][0].tolist()) == set(g["topk_" + k][0].tolist()), k
    parts = [criterions.softmax_dice(outs[0], target), tools.get_separate_loss(outs[1], target),
             tools.get_edge_separate_loss(outs[2], edge), tools.get_separate_loss(outs[3], target),
             tools.get_edge_separate_loss(outs[4], edge)]
    assert np.allclose([float(v) for v in parts], g["loss_parts"], rtol=2e-6)
    sum(parts).backward()
    names = list(g["grad_names"])
    l2, noise = g["grad_l2_f64"], g["grad_noise_ref32"]
    for n, p in m.named_parameters():
        i = names.index(n)
        assert p.grad is not None, n
        if l2[i] > 1e-7:      # conv biases in front of an InstanceNorm have zero true gradient
            got = float(p.grad.double().norm())
            assert abs(got - l2[i]) <= max(10 * noise[i], 2e-3) * l2[i], (n, got, l2[i])
    for key in g.files:
        if key.startswith("grad::"):
            n = key[6:]
            ref = torch.from_numpy(g[key]).double()
            got = dict(m.named_parameters())[n].grad.double()
            if float(ref.norm()) > 1e-7:
                assert float((got - ref).norm() / ref.norm()) < max(10 * noise[names.index(n)], 2e-3), n


def test_batch_two_equals_two_single_samples(emul_backend):
    m = _model().eval()
    x, _, _ = syn.synthetic_batch([0, 1], (64, 64, 64))
    with torch.no_grad():
        both = m(x, None)
        one = m(x[1:2], None)
    assert torch.allclose(both[0][1:2], one[0], atol=1e-6)
    assert torch.allclose(both[2]["04"][1:2], one[2]["04"], atol=1e-6)


def test_training_mode_dropout_runs(emul_backend):
    from models import criterions
    m = _model().train()
    m.Unet_list.InitConv.dropout = 0.2
    from utils import tools
    x, target, edge = syn.synthetic_batch([0], (64, 64, 64))
    torch.manual_seed(1)
    out = m(x, None)
    loss = criterions.softmax_dice(out[0], target) + tools.get_separate_loss(out[1], target) + \
        tools.get_edge_separate_loss(out[2], edge) + tools.get_separate_loss(out[3], target) + \
        tools.get_edge_separate_loss(out[4], edge)
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None for p in m.parameters())


def test_non_cubic_patch_matches_reference_fixture(emul_backend):
    """BASELINE configs[4] trains on non-cubic 160x192x160 patches.  The package's module tree (sizes derived from the input,
    SURVEY F3) on a 64x96x80 patch against the REFERENCE's outputs for that shape (tests/golden/model_noncubic.npz: the reference
    run with image_size / edge_image_size patched and a 8192-key fix_index.txt, oracle/make_golden.py) -- parity pinned."""
    g = np.load(os.path.join(GOLDEN, "model_noncubic.npz"))
    m = _model().eval()
    m.collect_aux = True
    x, _, _ = syn.synthetic_batch([0], (64, 96, 80))
    with torch.no_grad():
        out = m(x, None)
    assert out[0].shape == (1, 4, 64, 96, 80)
    assert np.allclose(out[0].reshape(-1)[g["prob_sample_idx"]].numpy(), g["prob_sample"], atol=5e-6)
    for j, nm in ((1, "sup"), (2, "edge"), (3, "mid_sup"), (4, "mid_edge")):
        for r in rm.REGIONS:
            t = out[j][r].reshape(-1)
            idx = (np.arange(1024, dtype=np.int64) * 2654435761 % t.numel()).astype(np.int64)
            assert np.allclose(t[idx].numpy(), g["%s_%s_sample" % (nm, r)], atol=5e-6), (nm, r)
    for k in ("01_edge", "02_sem", "04_edge_supp", "fusion"):
        assert set(m.aux[k][0].tolist()) == set(g["topk_" + k][0].tolist()), k
    with pytest.raises(ValueError):          # fewer than 128 semantic tokens: rejected, not silently truncated
        m(torch.zeros(1, 4, 32, 64, 48), None)


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    """include/cwf_hip.h is the drop-in boundary: the built library must load (no GPU needed for that) and export every entry point
    the header declares; the ctypes binding (cwf/_lib.py) must cover them.  No compute call is made here."""
    import ctypes
    import re
    from cwf import _lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "cwf_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(cwf_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 88, len(declared)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    bound = set(_lib.SIGNATURES) | {"cwf_arch", "cwf_plan_last_error"}
    unbound = [n for n in declared if n not in bound]
    assert not unbound, unbound
    assert lib.cwf_version() >= 2


def test_carry_link_routes_the_residual_gradient_and_sums_other_consumers(emul_backend):
    """functional.CarryLink: a conv that takes a carried alias as its residual hands dL/d(residual) to the producing conv's backward
    directly (autograd sees None for it); a SECOND consumer of the alias still reaches that backward through autograd and the two are
    added there.  Gradients must equal plain torch autograd on the same graph; the skip-buffer aliasing (alias_channels) must not trip
    autograd's view checks."""
    import torch.nn.functional as F
    from cwf import functional as CF, packing as pk
    torch.manual_seed(3)
    n, d, c = 1, 6, 8
    x = torch.randn(n, d, d, d, c, requires_grad=True)
    w1 = torch.randn(c, c, 3, 3, 3, requires_grad=True) * 0.1
    w2 = torch.randn(c, c, 3, 3, 3, requires_grad=True) * 0.1
    w1.retain_grad(); w2.retain_grad()
    b1 = torch.zeros(c, requires_grad=True); b2 = torch.zeros(c, requires_grad=True)
    s1, s2 = CF.ConvSpec(pk.CONV3_S1, c, c), CF.ConvSpec(pk.CONV3_S1, c, c)
    buf = CF.skip_buffer(n, d, d, d, c, c, x.device)
    h, _, xc = CF.conv(x, w1, b1, s1, carry=True)
    assert getattr(xc, "_cwf_carry", None) is not None
    y, _ = CF.conv(h, w2, b2, s2, residual=xc, out=CF.alias_channels(buf, 0, c))      # the residual's gradient travels through the link
    loss = (y * y).sum() + (xc * 0.5).sum()                                            # ... and a second consumer of the alias through autograd
    loss.backward()
    # reference: the same graph in plain torch ops (NCDHW)
    xr = x.detach().permute(0, 4, 1, 2, 3).clone().requires_grad_(True)
    w1r, w2r = w1.detach().clone().requires_grad_(True), w2.detach().clone().requires_grad_(True)
    hr = F.conv3d(xr, w1r, None, padding=1)
    yr = F.conv3d(hr, w2r, None, padding=1) + xr
    ((yr * yr).sum() + (xr * 0.5).sum()).backward()
    assert torch.allclose(x.grad.permute(0, 4, 1, 2, 3), xr.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(w1.grad, w1r.grad, rtol=1e-4, atol=1e-5) and torch.allclose(w2.grad, w2r.grad, rtol=1e-4, atol=1e-5)
    assert y.data_ptr() == buf.data_ptr() and torch.equal(buf[..., :c], y.detach())


def test_package_import_asks_for_eight_hardware_queues():
    """cwf/__init__.py: GPU_MAX_HW_QUEUES defaults to 8 before the first HIP call (with the runtime's default 4 the data-parallel step's
    streams share hardware queues and the launch plan's cross-stream waits serialise: 27.6 instead of 18.3 ms per step, tools/r3_comm.sh)."""
    import cwf  # noqa: F401
    assert os.environ.get("GPU_MAX_HW_QUEUES") is not None
