"""GPU: the whole HIP model + losses + backward against the golden fixtures generated from the imported reference
(tests/golden/model_{64,128}.npz) and against the CPU oracle run on this box.  Target (BASELINE.json): logits /
probabilities within 1e-3 relative of the fp32 CPU reference on identical inputs."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import reference_model as rm
from utils import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model():
    from models.clswiseformer.cls_wise_former import get_cls_wise_former
    m = get_cls_wise_former(dataset="brats", _conv_repr=True, _pe_type="fixed")
    m.load_state_dict(syn.det_state_dict(rm.param_shapes()), strict=False)
    m.Unet_list.InitConv.dropout = 0.0
    return m.to(DEV)


def _sample_idx(n, k=4096):
    return (np.arange(k, dtype=np.int64) * 2654435761 % n).astype(np.int64)


def _losses(outs, target, edge):
    from models import criterions
    from utils import tools
    return [criterions.softmax_dice(outs[0], target), tools.get_separate_loss(outs[1], target),
            tools.get_edge_separate_loss(outs[2], edge), tools.get_separate_loss(outs[3], target),
            tools.get_edge_separate_loss(outs[4], edge)]


@pytest.fixture()
def precision_mode(request):
    """"bf16x3+bf16grad" = the bench default: split-bf16 forward (meets the 1e-3 logit bound), single-bf16 operand products in the
    data- and weight-gradient kernels (fp32 accumulate / storage / master weights): the usual mixed-precision training arithmetic."""
    from cwf import kernels
    if request.param == "bf16x3+bf16grad":
        kernels.set_precision("bf16x3", wgrad="bf16", dgrad="bf16")
    else:
        kernels.set_precision(request.param)
    yield request.param
    kernels.set_precision("fp32")


@pytest.mark.parametrize("precision_mode", ["fp32", "bf16x3", "bf16x3+bf16grad"], indirect=True)
@pytest.mark.parametrize("tag,size", [("64", (64, 64, 64)), ("128", (128, 128, 128)), ("noncubic", (64, 96, 80))])
def test_forward_backward_vs_reference_golden(hip, tag, size, precision_mode):
    """fp32: exact-f32 MFMA.  bf16x3: split-bf16 MFMA operands (hi.hi + hi.lo + lo.hi, fp32 accumulate) -- the throughput
    mode; it must meet the same 1e-3 logits/probability bound of BASELINE.json against the fp32 CPU reference.
    "noncubic" = a 64x96x80 patch against the reference run with patched sizes (configs[4]'s non-cubic geometry, parity pinned)."""
    g = np.load(os.path.join(GOLDEN, "model_%s.npz" % tag))
    m = _model().eval()
    m.collect_aux = True
    x, target, edge = syn.synthetic_batch([0], size)
    x, target, edge = x.to(DEV), target.to(DEV), edge.to(DEV)
    outs = m(x, None)
    # ---- discrete decisions: the 13 top-k sets
    for k in g.files:
        if k.startswith("topk_"):
            got, ref = set(m.aux[k[5:]][0].tolist()), set(g[k][0].tolist())
            assert len(got ^ ref) <= (2 if precision_mode == "fp32" else 4), (k, len(got ^ ref))
    # ---- probabilities and logits, 1e-3 relative (north_star); achieved is ~1e-5
    prob = outs[0].detach().reshape(-1)[torch.from_numpy(g["prob_sample_idx"]).to(DEV)].cpu().numpy()
    assert np.abs(prob - g["prob_sample"]).max() <= 1e-3 * np.abs(g["prob_sample"]).max()
    logit = m.aux["logits"].detach().reshape(-1)[torch.from_numpy(g["prob_sample_idx"]).to(DEV)].cpu().numpy()
    rel = np.abs(logit - g["logits_sample"]).max() / np.abs(g["logits_sample"]).max()
    assert rel <= 1e-3, rel
    assert np.allclose(outs[0].detach().double().sum((0, 2, 3, 4)).cpu().numpy(), g["prob_sum_per_class"], rtol=1e-4)
    am = outs[0].detach().argmax(1).reshape(-1)
    assert np.abs(np.bincount(am.cpu().numpy(), minlength=4) - g["argmax_hist"]).sum() <= 1e-4 * am.numel()
    for j, nm in ((1, "sup"), (2, "edge"), (3, "mid_sup"), (4, "mid_edge")):
        for r in rm.REGIONS:
            t = outs[j][r].detach().reshape(-1)
            got = t[torch.from_numpy(_sample_idx(t.numel(), 1024)).to(DEV)].cpu().numpy()
            assert np.abs(got - g["%s_%s_sample" % (nm, r)]).max() <= 1e-3, (nm, r)
    bt = m.aux["bottleneck"].detach().permute(0, 4, 1, 2, 3).reshape(-1)
    got = bt[torch.from_numpy(_sample_idx(bt.numel(), 8192)).to(DEV)].cpu().numpy()
    assert np.abs(got - g["bottleneck_sample"]).max() <= 1e-3 * np.abs(g["bottleneck_sample"]).max()
    # ---- five losses
    parts = _losses(outs, target, edge)
    assert np.allclose([float(v) for v in parts], g["loss_parts"], rtol=1e-4), ([float(v) for v in parts], g["loss_parts"])
    # ---- gradients against the float64 truth, tolerance = the fp32 reference's own noise floor (x10) or 1e-2 on every per-tensor norm
    #      (all precisions; DESIGN section 3 measures 5.7e-3 worst in the bench precision); the ten full gradients elementwise: relative L2
    #      distance 1e-2 (fp32) / 2e-2 (single-bf16 gradient operands: 2^-9 per product, random sign, on top of the norm agreement)
    sum(parts).backward()
    names, l2, noise = list(g["grad_names"]), g["grad_l2_f64"], g["grad_noise_ref32"]
    bad = []
    for n, p in m.named_parameters():
        i = names.index(n)
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
        if l2[i] > 1e-7:
            got = float(p.grad.double().norm())
            if abs(got - l2[i]) > max(10 * noise[i], 1e-2) * l2[i]:
                bad.append((n, got, float(l2[i])))
    assert not bad, bad[:10]
    for key in g.files:
        if key.startswith("grad::"):
            n = key[6:]
            ref = torch.from_numpy(g[key]).double()
            got = dict(m.named_parameters())[n].grad.double().cpu()
            if float(ref.norm()) > 1e-7:
                assert float((got - ref).norm() / ref.norm()) < max(10 * noise[names.index(n)], 1e-2 if precision_mode == "fp32" else 2e-2), n


def test_teacher_forced_topk_and_cpu_oracle_64(hip):
    """Forward with the oracle's selected index sets injected (teacher forcing) must match the oracle run here."""
    m = _model().eval()
    x, _, _ = syn.synthetic_batch([3], (64, 64, 64))
    state = syn.det_state_dict(rm.param_shapes())
    with torch.no_grad():
        ref, aux = rm.forward(state, x, return_aux=True)
    m.forced_index = {k: v.to(DEV) for k, v in aux.items() if v.dtype == torch.int64}
    with torch.no_grad():
        outs = m(x.to(DEV), None)
    assert float((outs[0].cpu() - ref[0]).abs().max()) < 1e-4
    assert float((outs[4]["02"].cpu() - ref[4]["02"]).abs().max()) < 1e-4


def test_batch_two_and_channels_last_views(hip):
    m = _model().eval()
    x, _, _ = syn.synthetic_batch([0, 1], (64, 64, 64))
    with torch.no_grad():
        both = m(x.to(DEV), None)
        one = m(x[1:2].to(DEV), None)
    assert both[0].shape == (2, 4, 64, 64, 64) and both[0].permute(0, 2, 3, 4, 1).is_contiguous()
    assert float((both[0][1:2] - one[0]).abs().max()) < 1e-5
    assert float((both[1]["01"][1:2] - one[1]["01"]).abs().max()) < 1e-5


def test_training_step_updates_weights(hip):
    from cwf.optim import FusedAdam
    m = _model().train()
    m.Unet_list.InitConv.dropout = 0.2
    opt = FusedAdam(m.parameters(), lr=2e-4, weight_decay=1e-5, amsgrad=True)
    x, target, edge = syn.synthetic_batch([0], (64, 64, 64))
    x, target, edge = x.to(DEV), target.to(DEV), edge.to(DEV)
    losses = []
    for _ in range(3):
        outs = m(x, None)
        loss = sum(_losses(outs, target, edge))
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_sliding_window_predictor_matches_oracle_stitching(hip):
    """predict_overlap.tailor_and_concat (8 windows, incl. the D-axis stitch offset): batched-8 HIP forward vs the oracle's
    stitching of per-window HIP forwards, and window 0 vs the CPU oracle model."""
    import predict_overlap as po
    m = _model().eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 4, 240, 240, 155, generator=g)
    with torch.no_grad():
        y = po.tailor_and_concat(x.to(DEV), None, m).cpu()
        ref = rm.tailor_and_concat(x, lambda w: m(w.to(DEV), None)[0].cpu())
        assert y.shape == (1, 4, 240, 240, 155)
        assert float((y - ref).abs().max()) < 1e-5
        w0 = rm.forward(syn.det_state_dict(rm.param_shapes()), x[..., :128, :128, :128])[0]
    assert float((y[..., :128, :128, :128] - w0).abs().max()) < 1e-3
    tgt = torch.randint(0, 4, (1, 240, 240, 155))
    seg, prob, dice = po.validate_softmax(x.to(DEV), tgt.to(DEV), m)
    assert seg.shape == (1, 240, 240, 155) and len(dice) == 3
    # device-side argmax + WT/TC/ET counts (cwf_argmax_dice) == torch.argmax + tools.softmax_output_dice (utils/tools.py:89-109)
    from utils import tools
    assert torch.equal(seg.cpu(), prob.cpu().argmax(1))
    want = tools.softmax_output_dice(prob.cpu().argmax(1), tgt)
    assert all(abs(float(a) - float(b)) < 1e-6 for a, b in zip(dice, want))      # (the torch expression divides in float32)
    # N4: per-class IoU (tools.softmax_mIOU_score, utils/tools.py:50-61) from the same launch == the reference formula (oracle) == the host drop-in
    seg3, _, dice3, iou3 = po.validate_softmax(x.to(DEV), tgt.to(DEV), m, with_miou=True)
    assert torch.equal(seg3, seg) and all(abs(float(a) - float(b)) < 1e-12 for a, b in zip(dice3, dice))
    want_iou = rm.softmax_miou_score(prob.cpu().argmax(1), tgt)
    assert all(abs(float(a) - b) < 1e-9 for a, b in zip(iou3, want_iou)), (iou3, want_iou)
    host_iou = tools.softmax_mIOU_score(prob.cpu().argmax(1).numpy(), tgt.numpy())
    assert all(abs(float(a) - b) < 1e-9 for a, b in zip(host_iou, want_iou))
    seg2, d2 = hip.argmax_dice(m(x[..., :128, :128, :128].to(DEV), None)[0], None)      # channels-last view, no target
    assert d2 is None and seg2.shape == (1, 128, 128, 128)


def test_config4_patch_160x192x160_vs_reference_and_trains(hip):
    """BASELINE configs[4] patch size (160x192x160, 2400 semantic / 4800 edge tokens, all three sub-region + edge heads): the HIP
    forward against the REFERENCE's outputs at that size (tests/golden/model_config4.npz: reference with patched image sizes and
    a 8192-key fix_index.txt, forward + five losses), then one training step in the bench's precision configuration."""
    from cwf import kernels
    from cwf.optim import FusedAdam
    g = np.load(os.path.join(GOLDEN, "model_config4.npz"))
    kernels.set_precision("bf16x3", wgrad="bf16", dgrad="bf16")
    try:
        m = _model().eval()
        m.collect_aux = True
        x, target, edge = syn.synthetic_batch([0], (160, 192, 160))
        x, target, edge = x.to(DEV), target.to(DEV), edge.to(DEV)
        with torch.no_grad():
            outs = m(x, None)
            assert outs[0].shape == (1, 4, 160, 192, 160) and set(outs[1]) == {"01", "02", "04"} and set(outs[2]) == {"01", "02", "04"}
            si = torch.from_numpy(g["prob_sample_idx"]).to(DEV)
            prob = outs[0].reshape(-1)[si].cpu().numpy()
            assert np.abs(prob - g["prob_sample"]).max() <= 1e-3 * np.abs(g["prob_sample"]).max()
            logit = m.aux["logits"].reshape(-1)[si].cpu().numpy()
            assert np.abs(logit - g["logits_sample"]).max() / np.abs(g["logits_sample"]).max() <= 1e-3
            for j, nm in ((1, "sup"), (2, "edge"), (3, "mid_sup"), (4, "mid_edge")):
                for r in rm.REGIONS:
                    t = outs[j][r].reshape(-1)
                    got = t[torch.from_numpy(_sample_idx(t.numel(), 1024)).to(DEV)].cpu().numpy()
                    assert np.abs(got - g["%s_%s_sample" % (nm, r)]).max() <= 1e-3, (nm, r)
            for k in g.files:
                if k.startswith("topk_"):
                    assert len(set(m.aux[k[5:]][0].tolist()) ^ set(g[k][0].tolist())) <= 4, k
            parts = _losses(outs, target, edge)
            assert np.allclose([float(v) for v in parts], g["loss_parts"], rtol=1e-4)
        m = m.train()
        opt = FusedAdam(m.parameters(), lr=2e-4, weight_decay=1e-5, amsgrad=True)
        loss = sum(_losses(m(x, None), target, edge))
        opt.zero_grad(); loss.backward(); opt.step()
        assert np.isfinite(float(loss))
    finally:
        kernels.set_precision("fp32")


def test_flip_tta_matches_oracle_64(hip):
    """N4: 8-flip TTA through the HIP model (flips batched) vs the reference's formula evaluated on the CPU oracle model."""
    import predict_overlap as po
    m = _model().eval()
    x, _, _ = syn.synthetic_batch([4], (64, 64, 64))
    state = syn.det_state_dict(rm.param_shapes())
    with torch.no_grad():
        want = rm.flip_tta(x, lambda v: rm.forward(state, v)[0])
        got = po.flip_tta(x.to(DEV), None, lambda xb, mm: m(xb, mm)[0]).cpu()
    assert got.shape == (1, 4, 64, 64, 64)
    assert float((got - want).abs().max()) < 1e-3


def test_training_harness_on_gpu(hip, tmp_path):
    """N2: the train_no_amp.py counterpart end to end on the GPU (synthetic subjects, 64^3 crops, 3 iterations)."""
    import train_no_amp as tn
    rc = tn.main(["--synthetic", "2", "--crop_H", "64", "--crop_W", "64", "--crop_D", "64", "--end_epoch", "2", "--max_iters", "3",
                  "--log_every", "1", "--num_workers", "0", "--batch_size", "2", "--project_root", str(tmp_path), "--experiment", "g", "--date", "d"])
    assert rc == 0
    # the same harness with the step captured once and re-issued as a launch list (two eager steps, the capture, three listed steps)
    rc = tn.main(["--synthetic", "2", "--crop_H", "64", "--crop_W", "64", "--crop_D", "64", "--end_epoch", "6", "--max_iters", "6", "--step_mode", "plan",
                  "--log_every", "1", "--num_workers", "0", "--batch_size", "2", "--project_root", str(tmp_path), "--experiment", "p", "--date", "d"])
    assert rc == 0
    ck = torch.load(tmp_path / "checkpoint" / "gd" / "model_epoch_last.pth", weights_only=True)
    assert len(ck["state_dict"]) == 222 and all(k.startswith("module.") for k in ck["state_dict"])
    from cwf import kernels
    kernels.set_precision("fp32")


def test_wgrad_side_stream_gives_same_gradients(hip):
    """Trainer(wgrad_async=True) runs the weight-gradient kernels of single-use conv weights on a side stream.  Token selection
    is teacher-forced (a top-k flip from reduction-order noise would change the gradients wholesale); what remains
    between two runs of the SAME configuration is ~3e-6 relative (float atomics in the token reductions), so "same" means
    within that noise (tools/wg_async_check.py prints all three pairings)."""
    from cwf.trainer import Trainer
    xc, target, edge = syn.synthetic_batch([0], (64, 64, 64))
    with torch.no_grad():
        _, aux = rm.forward(syn.det_state_dict(rm.param_shapes()), xc, return_aux=True)
    forced = {k: v.to(DEV) for k, v in aux.items() if v.dtype == torch.int64}
    x, target, edge = xc.to(DEV), target.to(DEV), edge.to(DEV)
    flats = []
    for flag in (False, True, False):
        m = _model().train()
        m.forced_index = forced
        m.Unet_list.InitConv.dropout = 0.0
        for mod in m.modules():                     # no dropout anywhere
            if hasattr(mod, "dropout_rate"):
                mod.dropout_rate = 0.0
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        tr = Trainer(m, wgrad_async=flag)
        assert tr.wgrad_async == flag
        tr._fwd_bwd(x, target, edge)
        torch.cuda.synchronize()
        flats.append(tr.opt.flat_grad.clone())
    noise = float((flats[0] - flats[2]).norm() / flats[0].norm())          # sync vs sync
    diff = float((flats[0] - flats[1]).norm() / flats[0].norm())           # sync vs side stream
    assert float(flats[0].abs().sum()) > 0 and bool(torch.isfinite(flats[1]).all())
    assert diff < max(5e-5, 10 * noise), (diff, noise)


def _no_dropout_model(forced):
    m = _model().train()
    m.forced_index = forced
    m.Unet_list.InitConv.dropout = 0.0
    for mod in m.modules():
        if hasattr(mod, "dropout_rate"):
            mod.dropout_rate = 0.0
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


@pytest.mark.parametrize("tag,size", [("64", (64, 64, 64)), ("128", (128, 128, 128))])
def test_trainer_training_step_vs_reference_golden(hip, tag, size):
    """The path bench.py times, end to end against the REFERENCE's values (tests/golden/model_*.npz, float64 gradients of the imported
    reference): cwf.trainer.Trainer on the model in TRAINING mode (dropout rates zeroed, free top-k) in the bench precision -- fused
    head -> loss (LazyProb, grouped heads), gradient sink (flat buffer written by the backward kernels, batched split-K reduce),
    side-stream weight gradients, deferred decoder weight gradients.  (a) the five loss parts, (b) every per-parameter gradient norm,
    (c) the ten full gradients elementwise.  Reference: train_no_amp.py:181-239."""
    from cwf import kernels
    from cwf.trainer import Trainer
    g = np.load(os.path.join(GOLDEN, "model_%s.npz" % tag))
    kernels.set_precision("bf16x3", wgrad="bf16", dgrad="bf16")
    try:
        m = _no_dropout_model(None)
        assert m.training
        tr = Trainer(m)
        assert tr.wgrad_async
        x, target, edge = syn.synthetic_batch([0], size)
        x, target, edge = x.to(DEV), target.to(DEV), edge.to(DEV)
        for rep in range(2):                       # the second step runs on cached descriptor tables / slab buffers
            loss, parts = tr._fwd_bwd(x, target, edge)
            torch.cuda.synchronize()
            assert all(p.grad is None for p in m.parameters())          # every gradient went through the sink
            assert np.allclose([float(v) for v in parts], g["loss_parts"], rtol=1e-4), ([float(v) for v in parts], g["loss_parts"])
            assert abs(float(loss) - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
            names, l2, noise = list(g["grad_names"]), g["grad_l2_f64"], g["grad_noise_ref32"]
            pname = {id(p): n for n, p in m.named_parameters()}
            got = {}
            off = 0
            for p in tr.opt.sink.params:
                got[pname[id(p)]] = tr.opt.flat_grad[off:off + p.numel()].view_as(p)
                off += p.numel()
            assert off == tr.opt.flat_grad.numel() and len(got) == 218
            bad = []
            for n, gr in got.items():
                i = names.index(n)
                assert bool(torch.isfinite(gr).all()), n
                if l2[i] > 1e-7:
                    v = float(gr.double().norm())
                    if abs(v - l2[i]) > max(10 * noise[i], 1e-2) * l2[i]:
                        bad.append((n, v, float(l2[i])))
            assert not bad, (rep, bad[:10])
            for key in g.files:
                if key.startswith("grad::"):
                    n = key[6:]
                    ref = torch.from_numpy(g[key]).double()
                    if float(ref.norm()) > 1e-7:
                        d = float((got[n].double().cpu() - ref).norm() / ref.norm())
                        assert d < max(10 * noise[names.index(n)], 2e-2), (n, d)
    finally:
        kernels.set_precision("fp32")


def test_batch_two_at_bench_shape_128(hip):
    """The bench shape (batch 2 x 4 x 128^3, bench precision): sample 0 of the batch against the reference's B = 1 fixture of that
    sample (1e-3 on logits / probabilities) and sample 1 against a B = 1 HIP run of it (batch > 1 = independent samples)."""
    from cwf import kernels
    g = np.load(os.path.join(GOLDEN, "model_128.npz"))
    kernels.set_precision("bf16x3", wgrad="bf16", dgrad="bf16")
    try:
        m = _model().eval()
        m.collect_aux = True
        x, _, _ = syn.synthetic_batch([0, 1], (128, 128, 128))
        with torch.no_grad():
            both = m(x.to(DEV), None)
            logits2 = m.aux["logits"]
            si = torch.from_numpy(g["prob_sample_idx"]).to(DEV)
            prob = both[0][0:1].reshape(-1)[si].cpu().numpy()
            assert np.abs(prob - g["prob_sample"]).max() <= 1e-3 * np.abs(g["prob_sample"]).max()
            logit = logits2[0:1].reshape(-1)[si].cpu().numpy()
            assert np.abs(logit - g["logits_sample"]).max() / np.abs(g["logits_sample"]).max() <= 1e-3
            one = m(x[1:2].to(DEV), None)
            # (independent samples up to the summation order of the InstanceNorm statistics: the persistent kernels split their tile runs
            # over the workgroups by the TOTAL tile count, so a sample's fp32 partial sums group differently at B = 1 and B = 2)
            assert float((both[0][1:2] - one[0]).abs().max()) < 3e-5
            assert float((both[2]["04"][1:2] - one[2]["04"]).abs().max()) < 3e-5
    finally:
        kernels.set_precision("fp32")


def _replay(tr):
    if tr._plan is not None:
        tr._run_plan()
    else:
        tr._graph.replay()


@pytest.mark.parametrize("mode", ["plan", "hipgraph"])
def test_graph_replay_matches_eager_step(hip, mode):
    """Trainer(use_graph="plan" | "hipgraph"): forward + 5 losses + backward + gradient reduces captured ONCE -- the side-stream weight
    gradients as a parallel branch -- and re-issued as the library's launch list (csrc/plan.hip: plain launches on two streams, events
    for the cross-stream edges) or replayed with hipGraphLaunch, must give the eager step's flat gradient
    (teacher-forced top-k, dropout off: what remains is float-atomic reduction-order noise, measured eager vs eager), on the
    capture's own inputs and on NEW inputs copied into the static buffers; with dropout ON, two replays must draw different masks
    (the generator state is advanced by a captured kernel)."""
    from cwf import kernels
    from cwf.trainer import Trainer
    kernels.set_precision("bf16x3")
    try:
        xs, ts, es = zip(*[syn.synthetic_batch([i], (64, 64, 64)) for i in (0, 5)])
        with torch.no_grad():
            _, aux = rm.forward(syn.det_state_dict(rm.param_shapes()), xs[0], return_aux=True)
        forced = {k: v.to(DEV) for k, v in aux.items() if v.dtype == torch.int64}
        dev = lambda i: (xs[i].to(DEV), ts[i].to(DEV), es[i].to(DEV))
        eager = []
        for rep in range(2):
            tr = Trainer(_no_dropout_model(forced))
            per = []
            for i in (0, 1):
                tr._fwd_bwd(*dev(i))
                torch.cuda.synchronize()
                per.append(tr.opt.flat_grad.clone())
            eager.append(per)
        noise = max(float((eager[0][i] - eager[1][i]).norm() / eager[0][i].norm()) for i in (0, 1))
        trg = Trainer(_no_dropout_model(forced), use_graph=mode)
        assert trg.wgrad_async
        trg._fwd_bwd(*dev(0))                               # eager warm-up (allocations, weight-pack tables)
        torch.cuda.synchronize()
        trg._capture(*dev(0))
        if mode == "plan":
            pi = trg.plan_info
            assert trg._plan is not None and "error" not in pi, pi
            # every node is a plain launch; the weight-gradient chain sits on a stream of its own
            assert pi["kernels"] > 300 and pi["streams"] >= 2 and pi["on_stream1"] >= 20 and pi["markers"] == 0, pi
            assert pi["on_stream0"] > 4 * (pi["on_stream1"] + pi["on_streams2plus"]), pi
        else:
            assert trg._plan is None
        for i in (0, 1, 0):
            for dst, src in zip(trg._static, dev(i)):
                dst.copy_(src)
            _replay(trg)
            torch.cuda.synchronize()
            g = trg.opt.flat_grad
            assert bool(torch.isfinite(g).all()) and float(g.abs().sum()) > 0
            diff = float((g - eager[0][i]).norm() / eager[0][i].norm())
            assert diff < max(5e-5, 10 * noise), (i, diff, noise)
        # dropout on: replays differ (fresh masks), and stay finite
        m = _model().train()
        trd = Trainer(m, use_graph=mode)
        trd._fwd_bwd(*dev(0)); torch.cuda.synchronize()
        trd._capture(*dev(0))
        _replay(trd); torch.cuda.synchronize(); g1 = trd.opt.flat_grad.clone()
        _replay(trd); torch.cuda.synchronize(); g2 = trd.opt.flat_grad.clone()
        assert bool(torch.isfinite(g1).all()) and bool(torch.isfinite(g2).all())
        assert float((g1 - g2).norm() / g1.norm()) > 1e-3
    finally:
        kernels.set_precision("fp32")


def test_unplannable_capture_falls_back_to_eager_steps(hip, monkeypatch):
    """A capture the launch list cannot express (here: cwf_plan_create made to refuse) must not leave the Trainer on hipGraph replay
    (30 % slower than eager launches on this runtime): it goes on eagerly, and the steps still train."""
    from cwf import kernels
    from cwf.trainer import Trainer
    kernels.set_precision("bf16x3", wgrad="bf16", dgrad="bf16")
    try:
        x, t, e = syn.synthetic_batch([0], (64, 64, 64))
        x, t, e = x.to(DEV), t.to(DEV), e.to(DEV)
        tr = Trainer(_model().train(), use_graph="plan", graph_warmup=1)
        K = kernels.backend()
        monkeypatch.setattr(K.lib, "cwf_plan_create", lambda graph, out: -2)
        losses = [float(tr.step(x, t, e, 0)[0]) for _ in range(4)]
        assert tr._plan is None and tr._graph is None and not tr.use_graph and tr.plan_info.get("fallback") == "eager", tr.plan_info
        assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    finally:
        kernels.set_precision("fp32")


def test_gradient_sink_equals_plain_autograd(hip):
    """Trainer path (gradients written by the backward kernels into the flat buffer: batched split-K reduce per backward phase,
    coupler Functions writing their weight-set gradients in place, side-stream weight gradients) == plain `loss.backward()` of the
    same model with per-parameter .grad tensors, to reduction-order noise (teacher-forced top-k, dropout off)."""
    from cwf import kernels
    from cwf.trainer import Trainer, total_loss
    kernels.set_precision("bf16x3", wgrad="bf16", dgrad="bf16")
    try:
        xc, target, edge = syn.synthetic_batch([0], (64, 64, 64))
        with torch.no_grad():
            _, aux = rm.forward(syn.det_state_dict(rm.param_shapes()), xc, return_aux=True)
        forced = {k: v.to(DEV) for k, v in aux.items() if v.dtype == torch.int64}
        x, target, edge = xc.to(DEV), target.to(DEV), edge.to(DEV)
        plain = []
        for _ in range(2):
            m = _no_dropout_model(forced)
            loss, _ = total_loss(m(x, None), target, edge)
            loss.backward()
            plain.append({n: p.grad.clone() for n, p in m.named_parameters()})
        noise = max(float((plain[0][n] - plain[1][n]).norm() / (plain[0][n].norm() + 1e-30)) for n in plain[0] if float(plain[0][n].norm()) > 1e-7)
        gold = np.load(os.path.join(GOLDEN, "model_64.npz"))
        true_zero = {n for n, v in zip(gold["grad_names"], gold["grad_l2_f64"]) if v <= 1e-7}   # conv biases in front of an InstanceNorm:
        m = _no_dropout_model(forced)                      # their computed "gradient" is cancellation noise, summation order matters
        tr = Trainer(m)
        for rep in range(2):                               # twice: the second step reuses the cached descriptor tables / slab buffers
            tr._fwd_bwd(x, target, edge)
            torch.cuda.synchronize()
            assert all(p.grad is None for p in m.parameters())
            names = {id(p): n for n, p in m.named_parameters()}
            off = 0
            for p in tr.opt.sink.params:
                g = tr.opt.flat_grad[off:off + p.numel()].view_as(p)
                off += p.numel()
                ref = plain[0][names[id(p)]]
                if float(ref.norm()) > 1e-7 and names[id(p)] not in true_zero:
                    d = float((g - ref).norm() / ref.norm())
                    assert d < max(5e-5, 10 * noise), (names[id(p)], d, noise, rep)
        assert [hi - lo for lo, hi in tr.opt.sink.chunks] == [sum(p.numel() for p in ph) for ph in m.grad_phases()]
    finally:
        kernels.set_precision("fp32")


def test_dataparallel_and_ddp_wrapping_one_device(hip):
    """The reference wraps the model in nn.DataParallel for evaluation (test_overlap.py:78, one visible device) and in
    DistributedDataParallel for training (train_no_amp.py:133, one process per GPU).  Both wrappers around this package's module on
    ONE device: forward / backward run, the wrapped state_dict has the reference's 'module.' keys, and DDP's gradients (its reducer
    consumes the per-parameter .grad tensors of the plain autograd path) equal the unwrapped model's.  Multi-device DataParallel
    replicas inside one process are NOT supported: the packed-weight buffers belong to one device."""
    import torch.distributed as dist
    from cwf.trainer import total_loss
    xc, target, edge = syn.synthetic_batch([0], (64, 64, 64))
    with torch.no_grad():
        _, aux = rm.forward(syn.det_state_dict(rm.param_shapes()), xc, return_aux=True)
    forced = {k: v.to(DEV) for k, v in aux.items() if v.dtype == torch.int64}
    x, target, edge = xc.to(DEV), target.to(DEV), edge.to(DEV)
    base = _no_dropout_model(forced)
    loss0, _ = total_loss(base(x, None), target, edge)
    loss0.backward()
    ref = {n: p.grad.clone() for n, p in base.named_parameters()}
    # ---- nn.DataParallel, device_ids=[0]
    m = _no_dropout_model(forced)
    dp = torch.nn.DataParallel(m, device_ids=[0])
    assert list(dp.state_dict().keys())[0] == "module.e_token_01" and len(dp.state_dict()) == 222
    loss, _ = total_loss(dp(x, None), target, edge)
    loss.backward()
    assert abs(float(loss) - float(loss0)) <= 1e-6 * abs(float(loss0))
    assert all(torch.equal(p.grad, ref[n]) for n, p in m.named_parameters())
    with torch.no_grad():
        assert dp.eval()(x, None)[0].shape == (1, 4, 64, 64, 64)
    # ---- DistributedDataParallel, world size 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        m = _no_dropout_model(forced)
        ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0])
        loss, _ = total_loss(ddp(x, None), target, edge)
        loss.backward()
        assert abs(float(loss) - float(loss0)) <= 1e-6 * abs(float(loss0))
        for n, p in m.named_parameters():
            assert p.grad is not None and torch.allclose(p.grad, ref[n], rtol=1e-6, atol=1e-12), n
    finally:
        dist.destroy_process_group()


def test_rccl_phase_allreduce_path_one_rank(hip, monkeypatch):
    """The data-parallel path on the real backend: a one-rank RCCL group with CWF_FORCE_COMM=1 runs everything N > 1 runs -- the
    parameter broadcast, the three phase-wise all-reduces on the high-priority communication stream (waiting on the main and the
    weight-gradient streams), the joins before Adam -- and must leave the flat gradient and the updated weights of a plain
    single-process step (a one-rank sum is the identity; what remains is the float-atomic noise of two runs).  Also checks that
    the weight-gradient side stream really has a non-default priority: a default-priority stream shares the main stream's
    hardware queue once RCCL's streams exist (measured: zero overlap, -10 % throughput)."""
    import torch.distributed as dist
    from cwf.trainer import Trainer
    from cwf.kernels import _priority_stream
    assert _priority_stream(DEV, "low").priority > 0 and _priority_stream(DEV, "high").priority < 0
    xc, target, edge = syn.synthetic_batch([0], (64, 64, 64))
    with torch.no_grad():
        _, aux = rm.forward(syn.det_state_dict(rm.param_shapes()), xc, return_aux=True)
    forced = {k: v.to(DEV) for k, v in aux.items() if v.dtype == torch.int64}
    x, target, edge = xc.to(DEV), target.to(DEV), edge.to(DEV)

    def run(mode=False, steps=2):
        tr = Trainer(_no_dropout_model(forced), use_graph=mode)
        for _ in range(steps):
            tr.step(x, target, edge, 0)
        torch.cuda.synchronize()
        return tr, tr.opt.flat_grad.clone(), torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()])

    tr0, g0, w0 = run()
    _, g0b, _ = run()
    noise = float((g0 - g0b).norm() / g0.norm())
    assert not tr0.comm and tr0._comm_stream is None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 200))
    monkeypatch.setenv("CWF_FORCE_COMM", "1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        tr1, g1, w1 = run()
        assert tr1.comm and tr1.overlap_comm and tr1._comm_stream is not None and tr1._comm_stream.priority < 0
        assert len(tr1.opt.sink.chunks) == 3 and tr1._works == []
        # plan mode, data-parallel: the three cut points are captured as marker nodes on the communication stream; the launch list stops
        # at each, the phase's all-reduce is enqueued behind it, and the list continues (two eager steps, the capture, one listed step)
        tr2, g2, w2 = run("plan", steps=4)
        _, g0c, w0c = run(False, steps=4)
        assert tr2._plan is not None and tr2.plan_info["markers"] == 3 and tr2.overlap_comm and tr2._works == [], tr2.plan_info
    finally:
        dist.destroy_process_group()
    assert float((g1 - g0).norm() / g0.norm()) < max(5e-5, 10 * noise)
    assert float((w1 - w0).norm() / w0.norm()) < 1e-5
    assert float((g2 - g0c).norm() / g0c.norm()) < 1e-3
    assert float((w2 - w0c).norm() / w0c.norm()) < 1e-5


def test_grouped_head_maps_materialise_like_eval_and_ungrouped(hip, monkeypatch):
    """The three sub-regions' heads run as channel-grouped launches; their training-mode outputs are LazyProb maps that share one
    grouped logit buffer.  (1) Touching such a map materialises the real tensor (strided logits through upsample_softmax) and a loss
    written against it back-propagates; (2) the fused loss over the grouped maps and its gradients equal the ungrouped path
    (CWF_GROUPED_HEADS=0: one conv launch per region, per-map head-loss gradients)."""
    from cwf import functional as CF
    xc, target, edge = syn.synthetic_batch([0], (64, 64, 64))
    with torch.no_grad():
        _, aux = rm.forward(syn.det_state_dict(rm.param_shapes()), xc, return_aux=True)
    forced = {k: v.to(DEV) for k, v in aux.items() if v.dtype == torch.int64}      # teacher-forced top-k, dropout off: comparable runs
    x, target, edge = xc.to(DEV), target.to(DEV), edge.to(DEV)

    def run(grouped):
        monkeypatch.setenv("CWF_GROUPED_HEADS", "1" if grouped else "0")
        m = _no_dropout_model(forced)
        outs = m(x, None)
        assert all(isinstance(v, CF.LazyProb) and (v.parent is not None) == grouped for v in outs[1].values())
        loss = sum(_losses(outs, target, edge))
        loss.backward()
        g = {n: p.grad.detach().clone() for n, p in m.named_parameters() if "supervise_label" in n or "down_label" in n}
        return m, outs, float(loss), g

    m1, outs1, l1, g1 = run(True)
    m0, outs0, l0, g0 = run(False)
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    assert len(g1) == 48
    for n in g1:
        assert torch.allclose(g1[n], g0[n], rtol=2e-4, atol=1e-7 * float(g0[n].abs().max() + 1e-30)), n
    # materialisation of a grouped lazy map: same values as the ungrouped one, differentiable
    pm = outs1[1]["02"].materialize()
    assert tuple(pm.shape) == (1, 2, 64, 64, 64) and torch.allclose(pm, outs0[1]["02"].materialize(), rtol=1e-5, atol=1e-6)
    m2 = _no_dropout_model(forced)
    o2 = m2(x, None)
    (o2[3]["01"][:, 1].mean() + o2[4]["04"].sum() * 1e-6).backward()        # indexing / tensor methods materialise
    gl = m2.mid_supervise_label.down_label_1.weight.grad            # (outputs[3] = the mid-level label heads)
    assert gl is not None and bool(torch.isfinite(gl).all()) and float(gl.abs().sum()) > 0
