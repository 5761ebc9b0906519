"""Training-step harness: the reference hot loop (train_no_amp.py:181-239) without its per-iteration host syncs.

    forward -> softmax_dice + 2x get_separate_loss + 2x get_edge_separate_loss (all weights 1.0, :205-211)
    -> backward -> gradient average over ranks -> Adam(amsgrad) with the poly learning rate (:183,270-273).

Execution modes:
  * eager (default): every kernel is launched from Python (~2000 launches, ~26 ms of host time per step, hidden behind the GPU);
    the three sub-region pipelines run on parallel HIP streams.
  * graph: forward + losses + backward + gradient flattening are captured ONCE into a hipGraph (torch.cuda.graph) and replayed
    per step (host cost ~1 ms).  Shapes are static (fixed patch size, per-sample top-k of fixed k) and the token selection runs
    on device, so nothing in the step needs the host.  Currently slower end to end than eager (the captured branches overlap less).
Multi-GPU gradient averaging (the only data-path collective):
  * default: after backward the flat 67 MB gradient buffer is all-reduced in 4 chunks (a few large messages -- the right shape
    for point-to-point xGMI links; ~1 ms on an 8-GPU ring, <3 % of the step), then the single fused Adam kernel runs.  Backward's
    end synchronises every stream, so this is race-free with the multi-stream regions.
  * overlap_comm=True: bucketed all-reduce launched from autograd hooks while backward is still running (cwf.parallel.GradSync);
    the regions are then kept on one stream (gradients of one bucket would otherwise be produced on different streams).
Checkpoints use the reference layout {'epoch', 'state_dict' with 'module.' prefix, 'optim_dict'} (:248-253)."""
from __future__ import annotations

import torch
import torch.distributed as dist

from .optim import FusedAdam, poly_lr
from .parallel import GradSync


def total_loss(outputs, target, edge):
    from models import criterions
    from utils import tools
    parts = [criterions.softmax_dice(outputs[0], target), tools.get_separate_loss(outputs[1], target),
             tools.get_edge_separate_loss(outputs[2], edge), tools.get_separate_loss(outputs[3], target),
             tools.get_edge_separate_loss(outputs[4], edge)]
    return parts[0] + parts[1] + parts[2] + parts[3] + parts[4], parts


def kernels_backend():
    from .kernels import backend
    return backend()


class Trainer:
    def __init__(self, model, lr=2e-4, weight_decay=1e-5, amsgrad=True, end_epoch=1000, bucket_mb=16.0, use_graph=False,
                 graph_warmup=2, overlap_comm=False, wgrad_async=True):
        self.model = model
        self.init_lr, self.end_epoch = lr, end_epoch
        self.opt = FusedAdam(model.parameters(), lr=lr, weight_decay=weight_decay, amsgrad=amsgrad)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.use_graph = use_graph
        self.sync = None
        # weight gradients on a side stream (they are leaves of backward; the data-gradient chain is its critical path): +3 %.
        # Not with hook-driven bucketed all-reduce (the hooks would read .grad before the side stream is joined).
        self.wgrad_async = bool(wgrad_async) and not overlap_comm and next(model.parameters()).is_cuda
        if self.world > 1 and overlap_comm and not use_graph:
            self.sync = GradSync(model.parameters(), bucket_mb=bucket_mb)
            if hasattr(model, "parallel_regions"):
                model.parallel_regions = False
        if self.world > 1:
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, 0)
        self._graph = None
        self._static = None
        self._eager_steps = 0
        self._graph_warmup = graph_warmup

    # ------------------------------------------------------------------------------------------------
    def _fwd_bwd(self, x, target, edge):
        outputs = self.model(x, None)
        loss, parts = total_loss(outputs, target, edge)
        self.opt.zero_grad(set_to_none=True)
        K = kernels_backend() if self.wgrad_async else None
        if K is not None:
            K.wgrad_async = True
        try:
            loss.backward()
        finally:
            if K is not None:
                K.wgrad_async = False
                K.join_wgrad_stream()      # weight gradients were produced on the side stream
        if self.sync is not None:
            self.sync.finish()            # eager multi-GPU: bucketed all-reduce overlapped with backward
        self.opt.gather_grads()
        return loss.detach(), [p.detach() for p in parts]

    def _allreduce_flat(self):
        if self.world > 1 and self.sync is None:
            flat = self.opt.flat_grad
            works = [dist.all_reduce(c, async_op=True) for c in flat.chunk(4)]
            for w in works:
                w.wait()
            flat.div_(self.world)

    def _capture(self, x, target, edge):
        self._static = (x.clone(), target.clone(), edge.clone())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._static_out = self._fwd_bwd(*self._static)
        self._graph = g

    def step(self, x, target, edge, epoch=0):
        """One optimisation step on a rank-local batch.  Returns (loss, [five parts]) as device tensors (no host sync)."""
        self.opt.param_groups[0]["lr"] = float(poly_lr(self.init_lr, epoch, self.end_epoch))   # plain float: checkpoints stay weights_only-loadable
        if self.use_graph and self._graph is None and self._eager_steps >= self._graph_warmup:
            torch.cuda.synchronize()
            self._capture(x, target, edge)
        if self._graph is not None:
            sx, st, se = self._static
            sx.copy_(x); st.copy_(target); se.copy_(edge)
            self._graph.replay()
            loss, parts = self._static_out
        else:
            loss, parts = self._fwd_bwd(x, target, edge)
            self._eager_steps += 1
        self._allreduce_flat()
        self.opt.advance_host()
        self.opt.launch()
        return loss, parts


def save_checkpoint(path, model, optimizer, epoch):
    """train_no_amp.py:248-253: the reference saves the DDP-wrapped model, hence the 'module.' key prefix."""
    sd = {"module." + k: v for k, v in model.state_dict().items()}
    torch.save({"epoch": epoch, "state_dict": sd, "optim_dict": optimizer.state_dict()}, path)


def load_checkpoint(path, model, map_location="cpu"):
    """train_no_amp.py:147-151 (weights only; the reference never restores 'optim_dict' or 'epoch')."""
    ck = torch.load(path, map_location=map_location, weights_only=True)
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in ck["state_dict"].items()}
    model.load_state_dict(sd)
    return ck.get("epoch", 0)
