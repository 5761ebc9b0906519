"""CPU: the oracle restatement (oracle/reference_model.py) against the golden fixtures produced from the imported
reference (oracle/make_golden.py), plus the reference-interface checks that need no GPU."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import reference_model as rm
from utils import synthetic as syn


def _sample_idx(n, k=4096):
    return (np.arange(k, dtype=np.int64) * 2654435761 % n).astype(np.int64)


def _full_state():
    state = syn.det_state_dict(rm.param_shapes())
    for n, shp, is_buf in rm.param_shapes():
        if is_buf:
            state[n] = rm.fixed_pe_table()
    return state


def test_param_inventory():
    shapes = rm.param_shapes()
    assert len(shapes) == 222
    assert sum(int(np.prod(s)) for _, s, b in shapes if not b) == 16824556
    assert sum(int(np.prod(s)) for _, s, b in shapes if b) == 4 * 1024 * 512


def test_losses_against_reference_fixture():
    g = np.load(os.path.join(GOLDEN, "losses.npz"))
    target, edge = torch.from_numpy(g["target"]), torch.from_numpy(g["edge"])
    p4 = torch.from_numpy(g["p4"]).requires_grad_(True)
    l = rm.softmax_dice(p4, target)
    l.backward()
    assert abs(float(l) - float(g["softmax_dice"])) < 1e-6
    assert np.allclose(p4.grad.numpy(), g["softmax_dice_grad"], rtol=1e-5, atol=1e-9)
    outs = {r: torch.from_numpy(g["p2_" + r]).requires_grad_(True) for r in rm.REGIONS}
    ls = rm.get_separate_loss(outs, target)
    ls.backward()
    assert abs(float(ls) - float(g["separate_loss"])) < 1e-6
    for r in rm.REGIONS:
        assert np.allclose(outs[r].grad.numpy(), g["sep_grad_" + r], rtol=1e-5, atol=1e-9)
        outs[r].grad = None
    le = rm.get_edge_separate_loss(outs, edge)
    le.backward()
    assert abs(float(le) - float(g["edge_separate_loss"])) < 1e-6
    for r in rm.REGIONS:
        assert np.allclose(outs[r].grad.numpy(), g["edge_grad_" + r], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("tag,size", [("64", (64, 64, 64)), ("noncubic", (64, 96, 80))])
def test_forward_against_reference_fixture(tag, size):
    """The oracle against the REFERENCE's own outputs (oracle/make_golden.py imports /root/reference): the cubic 64^3 case and a
    non-cubic 64x96x80 patch (reference with image_size / edge_image_size patched) -- the latter pins the size
    generalisation (F3) every other non-cubic test relies on."""
    g = np.load(os.path.join(GOLDEN, "model_%s.npz" % tag))
    x, target, edge = syn.synthetic_batch([0], size)
    with torch.no_grad():
        outs, aux = rm.forward(_full_state(), x, return_aux=True)
        loss, parts = rm.total_loss(outs, target, edge)
    prob = outs[0].reshape(-1)
    assert np.allclose(prob[g["prob_sample_idx"]].numpy(), g["prob_sample"], atol=2e-6)
    assert np.allclose([float(v) for v in parts], g["loss_parts"], rtol=1e-6)
    for j, nm in ((1, "sup"), (2, "edge"), (3, "mid_sup"), (4, "mid_edge")):
        for r in rm.REGIONS:
            t = outs[j][r].reshape(-1)
            assert np.allclose(t[_sample_idx(t.numel(), 1024)].numpy(), g["%s_%s_sample" % (nm, r)], atol=2e-6)
    for k, v in aux.items():
        if v.dtype == torch.int64:
            assert set(v[0].tolist()) == set(g["topk_" + k][0].tolist()), k


def test_edge_codes_and_poly_lr():
    _, t, e = syn.synthetic_sample(0, (32, 32, 32))
    assert set(np.unique(e.numpy()).tolist()) <= {0, 1, 2, 4, 5, 6, 7, 8}
    assert set(np.unique(t.numpy()).tolist()) == {0, 1, 2, 3}
    assert rm.poly_lr(2e-4, 0, 1000) == 2e-4
    assert rm.poly_lr(2e-4, 500, 1000) == round(2e-4 * 0.5 ** 0.9, 8)


def test_tailor_and_concat_geometry():
    x = torch.zeros(1, 4, 240, 240, 155)
    calls = []

    def fwd(win):
        calls.append(tuple(win.shape))
        return torch.full_like(win, float(len(calls)))
    y = rm.tailor_and_concat(x, fwd)
    assert y.shape == (1, 4, 240, 240, 155) and len(calls) == 8 and all(c == (1, 4, 128, 128, 128) for c in calls)
    assert float(y[0, 0, 0, 0, 0]) == 1 and float(y[0, 0, 239, 239, 154]) == 8 and float(y[0, 0, 0, 200, 0]) == 2
