"""Data-parallel gradient synchronisation: one process per GPU, bucketed all-reduce (RCCL over xGMI when the process
group backend is 'nccl'; gloo in the CPU tests) launched from autograd hooks while backward is still running.

Replaces ``nn.parallel.DistributedDataParallel`` as the reference uses it (train_no_amp.py:127-133): the only data-path
collective is the gradient all-reduce (16,824,556 fp32 = 67.3 MB per step).  Samples are independent
(InstanceNorm / LayerNorm only), so rank-local batches need no other exchange.  The per-iteration barrier and the five
scalar all-reduces of train_no_amp.py:216-224 feed a log line that prints local values; they are dropped.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of 67 MB is ~0.8 ms per step, far below the
step time, so a handful of ~16 MB buckets (large messages, few launches) is the right granularity here.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


class GradSync:
    """Bucketed, overlapped gradient averaging.

    usage:  sync = GradSync(model.parameters()); ...; loss.backward(); sync.finish(); optimizer.step()
    Parameters are bucketed in reverse registration order (the decoder, registered late, produces gradients first);
    a bucket is reduced as soon as all of its gradients exist.  ``finish()`` flushes stragglers, waits and writes the
    averaged gradients back."""

    def __init__(self, params, bucket_mb: float = 16.0, group=None, average: bool = True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.average = average
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.buckets: List[List[int]] = []
        cur, cur_bytes, cap = [], 0, int(bucket_mb * 2 ** 20)
        for i in reversed(range(len(self.params))):
            cur.append(i)
            cur_bytes += self.params[i].numel() * 4
            if cur_bytes >= cap:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)
        self.bucket_of = {}
        for b, idxs in enumerate(self.buckets):
            for i in idxs:
                self.bucket_of[i] = b
        self._flat = [None] * len(self.buckets)
        self._pending = [0] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._hooks = []
        if self.world > 1:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.reset()

    def broadcast_parameters(self, tensors, src=0):
        """One-time parameter/buffer broadcast from rank 0 (what the DDP constructor does)."""
        if self.world > 1:
            for t in tensors:
                dist.broadcast(t.data, src, group=self.group)

    def reset(self):
        for b, idxs in enumerate(self.buckets):
            self._pending[b] = len(idxs)
            self._work[b] = None
            self._launched[b] = False

    def _make_hook(self, i):
        def hook(param):
            b = self.bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        if self._launched[b]:
            return
        idxs = self.buckets[b]
        grads = []
        for i in idxs:
            p = self.params[i]
            grads.append((p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1))
        flat = torch.cat(grads)
        self._flat[b] = flat
        self._work[b] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._launched[b] = True

    def finish(self):
        """Wait for all buckets, average, scatter back into .grad.  Call after backward, before optimizer.step()."""
        if self.world == 1:
            return
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)          # parameters that received no gradient this step still take part
        for b, idxs in enumerate(self.buckets):
            self._work[b].wait()
            flat = self._flat[b]
            if self.average:
                flat.div_(self.world)
            off = 0
            for i in idxs:
                p = self.params[i]
                n = p.numel()
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n
        self.reset()

    def remove(self):
        for h in self._hooks:
            h.remove()


def shard_indices(n_items: int, rank: int, world: int, epoch: int = 0, shuffle: bool = True, seed: int = 0):
    """DistributedSampler-equivalent (train_no_amp.py:162,175): a per-epoch permutation padded to a multiple of the
    world size, strided over ranks."""
    g = torch.Generator().manual_seed(seed + epoch)
    order = torch.randperm(n_items, generator=g).tolist() if shuffle else list(range(n_items))
    total = (n_items + world - 1) // world * world
    order += order[: total - n_items]
    return order[rank:total:world]
