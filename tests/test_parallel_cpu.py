"""CPU, world_size 2 over gloo: the data-parallel pieces (cwf/parallel.py) -- bucketed gradient all-reduce from autograd hooks
equals the mean of the per-rank gradients (what N ranks x B=1 compute in the reference's DDP run, train_no_amp.py:127-133),
parameter broadcast, and the DistributedSampler-equivalent sharding."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, REPO


def _worker(rank, world, port, ret):
    for p in (PKG, REPO):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cwf.parallel import GradSync, shard_indices
        torch.manual_seed(123 + rank)                       # different initial weights per rank ...
        model = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.GELU(), torch.nn.Linear(64, 64), torch.nn.GELU(),
                                    torch.nn.Linear(64, 8))
        unused = torch.nn.Parameter(torch.ones(5))          # a parameter that never receives a gradient
        params = list(model.parameters()) + [unused]
        sync = GradSync(params, bucket_mb=0.002)             # tiny buckets -> several all-reduces, launched from hooks
        sync.broadcast_parameters(params)                   # ... made identical by the rank-0 broadcast
        assert len(sync.buckets) > 2
        g = torch.Generator().manual_seed(7)
        xs = torch.randn(world, 4, 32, generator=g)
        ys = torch.randn(world, 4, 8, generator=g)
        for step in range(2):
            for p in params:
                p.grad = None
            loss = ((model(xs[rank]) - ys[rank]) ** 2).mean()
            loss.backward()
            sync.finish()
            # reference: mean over ranks of the single-rank gradients, computed locally
            ref_model = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.GELU(), torch.nn.Linear(64, 64), torch.nn.GELU(),
                                            torch.nn.Linear(64, 8))
            ref_model.load_state_dict(model.state_dict())
            acc = [torch.zeros_like(p) for p in ref_model.parameters()]
            for r in range(world):
                ref_model.zero_grad()
                ((ref_model(xs[r]) - ys[r]) ** 2).mean().backward()
                for a, p in zip(acc, ref_model.parameters()):
                    a += p.grad / world
            for a, p in zip(acc, model.parameters()):
                assert torch.allclose(p.grad, a, atol=1e-6), "averaged gradient mismatch"
            assert unused.grad is not None and float(unused.grad.abs().max()) == 0.0
            with torch.no_grad():
                for p in model.parameters():
                    p -= 0.1 * p.grad
        # replicas stay identical
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert torch.equal(gathered[0], gathered[1])
        # sharding: disjoint, covers everything, changes with the epoch
        mine = shard_indices(11, rank, world, epoch=3)
        allidx = [None] * world
        dist.all_gather_object(allidx, mine)
        assert len(mine) == 6 and set(sum(allidx, [])) == set(range(11))
        assert shard_indices(11, rank, world, epoch=4) != mine
        ret[rank] = "ok"
    except Exception as e:      # surface the failure in the parent
        ret[rank] = "FAIL: %r" % (e,)
        raise
    finally:
        dist.destroy_process_group()


def test_gradsync_world2_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert [ret.get(r) for r in range(world)] == ["ok"] * world


def _trainer_worker(rank, world, port, ret):
    for p in (PKG, REPO):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cwf import kernels
        from cwf.trainer import Trainer, total_loss
        from oracle.kernel_emul import EmulBackend
        from oracle import reference_model as rm
        from utils import synthetic as syn
        from models.clswiseformer.cls_wise_former import get_cls_wise_former
        kernels._set_backend_for_testing(EmulBackend())

        def build(perturb):
            m = get_cls_wise_former(dataset="brats", _conv_repr=True, _pe_type="fixed")
            m.load_state_dict(syn.det_state_dict(rm.param_shapes()), strict=False)
            m.Unet_list.InitConv.dropout = 0.0
            for mod in m.modules():                      # deterministic: every dropout off (the fused training paths stay on)
                if hasattr(mod, "dropout_rate"):
                    mod.dropout_rate = 0.0
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
            if perturb:
                with torch.no_grad():
                    for p in m.parameters():
                        p.add_(0.01)
            return m.train()

        model = build(perturb=(rank == 1))              # rank 1 starts from other weights: the Trainer's broadcast must fix that
        tr = Trainer(model)
        assert tr.world == 2 and tr.overlap_comm and tr.opt.grad_scale == 0.5
        batches = [syn.synthetic_batch([i], (64, 64, 64)) for i in range(world)]
        # ---- step 1 by hand: flat gradient after the phase-wise all-reduces == sum of the two single-rank gradients (plain autograd)
        tr._fwd_bwd(*batches[rank])
        assert len(tr._works) == 3                       # decoder slice, middle slice, encoder slice
        tr._finish_comm()
        ref = build(perturb=False)
        want = torch.zeros_like(tr.opt.flat_grad)
        for xb, tb, eb in batches:
            ref.zero_grad(set_to_none=True)
            loss, _ = total_loss(ref(xb, None), tb, eb)
            loss.backward()
            want += torch.cat([dict(zip(map(id, model.parameters()), ref.parameters()))[id(p)].grad.reshape(-1) for p in tr.opt.sink.params])
        err = float((tr.opt.flat_grad - want).norm() / want.norm())
        assert err < 2e-6, "flat gradient != sum of single-rank gradients: %g" % err
        assert all(p.grad is None for p in model.parameters())          # the gradients live in the flat buffer only
        tr.opt.advance_host(); tr.opt.launch()
        # ---- step 2 through the public entry point; replicas must stay bit-identical
        tr.step(*batches[rank], epoch=0)
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert torch.equal(gathered[0], gathered[1]), "replicas diverged"
        start = torch.cat([p.detach().reshape(-1) for p in build(False).parameters()])
        assert float((flat - start).abs().max()) > 1e-5  # and they did move
        ret[rank] = "ok"
    except Exception as e:
        ret[rank] = "FAIL: %r" % (e,)
        raise
    finally:
        dist.destroy_process_group()


def test_real_trainer_world2_gloo_overlapped_allreduce():
    """The ACTUAL cwf.trainer.Trainer on the ACTUAL ClsWiseFormer (kernels through the oracle emulation, 64^3, one different sample
    per rank) at world size 2: rank-0 broadcast, phase-wise all-reduce launched from the backward cut points, flat gradient ==
    sum over ranks of the plain-autograd single-rank gradients (Adam divides by the world size: the DDP average of
    train_no_amp.py:127-133,233), replicas bit-identical after two steps."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_trainer_worker, args=(world, port, ret), nprocs=world, join=True)
    assert [ret.get(r) for r in range(world)] == ["ok"] * world


def test_trainer_checkpoint_layout(tmp_path):
    """Checkpoint file layout of train_no_amp.py:248-253: {'epoch', 'state_dict' with 'module.' keys, 'optim_dict'}."""
    from cwf.trainer import save_checkpoint, load_checkpoint
    from models.clswiseformer.cls_wise_former import get_cls_wise_former
    m = get_cls_wise_former(dataset="brats", _conv_repr=True, _pe_type="fixed")
    opt = torch.optim.Adam(m.parameters(), lr=2e-4, weight_decay=1e-5, amsgrad=True)
    path = str(tmp_path / "model_epoch_0.pth")
    save_checkpoint(path, m, opt, 0)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ck.keys()) == {"epoch", "state_dict", "optim_dict"}
    assert all(k.startswith("module.") for k in ck["state_dict"]) and len(ck["state_dict"]) == 222
    m2 = get_cls_wise_former(dataset="brats", _conv_repr=True, _pe_type="fixed")
    assert load_checkpoint(path, m2) == 0
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    # the reference loads it onto a DDP/DataParallel-wrapped model (train_no_amp.py:133, test_overlap.py:78): same keys
    wrapped = torch.nn.DataParallel(m2)
    assert list(wrapped.state_dict().keys()) == list(ck["state_dict"].keys())
    wrapped.load_state_dict(ck["state_dict"])
