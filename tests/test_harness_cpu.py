"""CPU: the rows SURVEY section 8(f) marks "next" -- data path (N3), training harness (N2), flip-TTA (N4) -- host logic only
(the model runs through the kernel emulation, oracle/kernel_emul.py)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import reference_model as rm
from utils import synthetic as syn


def test_datasets_shapes_codes_and_crops(tmp_path):
    from utils import data
    ds = data.SyntheticBraTS(3, (32, 32, 32), seed=5, full_size=(48, 40, 36))
    x, t, e, mm = ds[1]
    assert x.shape == (4, 32, 32, 32) and x.dtype == torch.float32 and t.shape == (32, 32, 32) and t.dtype == torch.int64
    assert set(torch.unique(t).tolist()) <= {0, 1, 2, 3} and set(torch.unique(e).tolist()) <= {0, 1, 2, 4, 5, 6, 7, 8}
    assert torch.equal(e, syn.edge_codes(t)) and mm.shape == (4,)
    x2, _, _, _ = ds[1]
    assert torch.equal(x, x2)                                   # same (seed, epoch, index) -> same crop
    ds.set_epoch(1)
    assert not torch.equal(ds[1][0], x)                         # another epoch -> another crop
    # crop_pad zero-pads past the volume end (D = 155 -> 160 in the reference's flags)
    v = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)
    c = data.crop_pad(v, (1, 2, 3), (2, 4, 4))
    assert c.shape == (2, 2, 4, 4) and torch.equal(c[:, :, :2, :2], v[:, 1:3, 2:4, 3:5]) and float(c[:, :, 2:, :].abs().sum()) == 0
    # npz subjects: channel-last image, BraTS label 4 -> class 3
    img = np.random.default_rng(0).standard_normal((40, 36, 34, 4)).astype(np.float32)
    lab = np.zeros((40, 36, 34), np.int16); lab[10:20, 10:20, 10:20] = 4; lab[12:16, 12:16, 12:16] = 1
    np.savez(tmp_path / "subj0.npz", image=img, label=lab)
    nz = data.NpzBraTS(str(tmp_path), crop=(32, 32, 32), seed=1)
    x, t, e, _ = nz[0]
    assert x.shape == (4, 32, 32, 32) and set(torch.unique(t).tolist()) <= {0, 1, 3}


def test_should_save_matches_reference_cadence():
    import train_no_amp as tn
    # train_no_amp.py:243-246 with end_epoch=1000, save_freq=50: every 50th epoch, and epochs 996, 997, 998 (0-based)
    saved = [e for e in range(1000) if tn.should_save(e, 1000, 50)]
    assert saved[:3] == [49, 99, 149] and {996, 997, 998} <= set(saved) and 995 not in saved


def test_flag_surface_matches_reference():
    import train_no_amp as tn
    a = tn.build_parser().parse_args([])
    for name, val in (("lr", 0.0002), ("weight_decay", 1e-5), ("amsgrad", True), ("criterion", "softmax_dice"), ("seed", 1000),
                      ("batch_size", 1), ("end_epoch", 1000), ("save_freq", 50), ("crop_H", 128), ("input_D", 160), ("load", True)):
        assert getattr(a, name) == val, name
    assert tn.build_parser().parse_args(["--amsgrad", "false"]).amsgrad is False


def test_harness_runs_two_iterations_and_writes_reference_layout(emul_backend, tmp_path):
    import train_no_amp as tn
    rc = tn.main(["--synthetic", "2", "--crop_H", "64", "--crop_W", "64", "--crop_D", "64", "--end_epoch", "1", "--max_iters", "2",
                  "--log_every", "1", "--project_root", str(tmp_path), "--experiment", "t", "--date", "d", "--no_cuda", "true"])
    assert rc == 0
    files = sorted(os.path.basename(p) for p in glob.glob(str(tmp_path / "checkpoint" / "td" / "*.pth")))
    assert "model_epoch_last.pth" in files
    ck = torch.load(tmp_path / "checkpoint" / "td" / "model_epoch_last.pth", weights_only=True)
    assert set(ck) == {"epoch", "state_dict", "optim_dict"} and ck["epoch"] == 1
    assert list(ck["state_dict"]) == ["module." + n for n, _, _ in rm.param_shapes()]
    assert set(ck["optim_dict"]) == {"state", "param_groups"}
    log = open(glob.glob(str(tmp_path / "log" / "*.txt"))[0]).read()
    assert "loss:" in log and "training process finished" in log
    # resume path: weights-only load of the file just written
    rc = tn.main(["--synthetic", "1", "--crop_H", "64", "--crop_W", "64", "--crop_D", "64", "--end_epoch", "1", "--max_iters", "1",
                  "--resume", str(tmp_path / "checkpoint" / "td" / "model_epoch_last.pth"), "--project_root", str(tmp_path / "r2"),
                  "--no_cuda", "true"])
    assert rc == 0


def test_flip_tta_equals_reference_formula():
    import predict_overlap as po
    g = torch.Generator().manual_seed(3)
    w = torch.randn(4, 4, 3, 3, 3, generator=g)
    fwd = lambda xb, mm=None: torch.softmax(torch.nn.functional.conv3d(xb, w, padding=1), dim=1)     # not flip-equivariant
    x = torch.randn(2, 4, 8, 10, 12, generator=g)
    want = rm.flip_tta(x, fwd)
    for batch in (1, 4, 8, 16):
        got = po.flip_tta(x, None, fwd, batch=batch)
        assert float((got - want).abs().max()) < 1e-6
    assert float((po.flip_tta(x, None, fwd, resoftmax=False) - want).abs().max()) > 1e-3


def test_fused_adam_state_reload_cpu(emul_backend):
    """FusedAdam (pointer-table launch through the kernel emulation) == torch.optim.Adam(amsgrad, weight_decay) over a
    save / load_state_dict / continue cycle: 'optim_dict' (train_no_amp.py:252) is interchangeable in both directions."""
    import copy
    from cwf.optim import FusedAdam
    gen = torch.Generator().manual_seed(3)
    p0 = [torch.randn(300, generator=gen), torch.randn(7, 5, generator=gen)]
    grads = [[torch.randn(300, generator=gen), torch.randn(7, 5, generator=gen)] for _ in range(6)]
    ref = [torch.nn.Parameter(t.clone()) for t in p0]
    ropt = torch.optim.Adam(ref, lr=1e-2, weight_decay=1e-3, amsgrad=True)
    ours = [torch.nn.Parameter(t.clone()) for t in p0]
    opt = FusedAdam(ours, lr=1e-2, weight_decay=1e-3, amsgrad=True)
    for i in range(6):
        for p, g in zip(ref, grads[i]):
            p.grad = g.clone()
        ropt.step()
        if i == 3:                                   # torch state -> FusedAdam (and our own state_dict round trip)
            opt = FusedAdam(ours, lr=1e-2, weight_decay=1e-3, amsgrad=True)
            opt.load_state_dict(copy.deepcopy(ropt.state_dict()))   # (torch shares the 'step' tensor otherwise)
            for p, r in zip(ours, ref):
                p.data.copy_(r.data)
            assert opt._steps == 4
            continue
        for p, g in zip(ours, grads[i]):
            p.grad = g.clone()
        opt.step()
        if i == 1:
            sd = opt.state_dict()
            opt = FusedAdam(ours, lr=1e-2, weight_decay=1e-3, amsgrad=True)
            opt.load_state_dict(sd)
            assert opt._steps == 2
    for p, r in zip(ours, ref):
        assert torch.allclose(p, r, rtol=1e-5, atol=1e-7)
    # and the other direction: torch.optim.Adam accepts our state
    t2 = torch.optim.Adam(ours, lr=1e-2, weight_decay=1e-3, amsgrad=True)
    t2.load_state_dict(opt.state_dict())
    assert float(t2.state[ours[0]]["step"]) == 6.0


def test_miou_drop_in_equals_reference_formula():
    """utils.tools.mIOU / softmax_mIOU_score (reference utils/tools.py:50-61, used by predict_simple.py) on numpy and torch label maps
    against the oracle restatement, incl. an absent class (eps / eps = 1, as in the reference)."""
    import numpy as np
    import torch
    from oracle import reference_model as rm
    from utils import tools
    g = torch.Generator().manual_seed(3)
    o = torch.randint(0, 4, (2, 9, 10, 11), generator=g)
    t = torch.randint(0, 3, (2, 9, 10, 11), generator=g)          # class 3 absent from the target
    want = rm.softmax_miou_score(o, t)
    got_t = [float(v) for v in tools.softmax_mIOU_score(o, t)]
    got_n = [float(v) for v in tools.softmax_mIOU_score(o.numpy(), t.numpy())]
    assert np.allclose(got_n, want, rtol=0, atol=1e-9)            # numpy label maps (what predict_simple.py passes): float64
    assert np.allclose(got_t, want, rtol=0, atol=1e-6)            # torch label maps: the reference's expression divides in float32
    assert float(tools.mIOU(o == 7, t == 7)) == 1.0
