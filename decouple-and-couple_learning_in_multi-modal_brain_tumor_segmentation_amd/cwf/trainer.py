"""Training-step harness: the reference hot loop (train_no_amp.py:181-239) without its per-iteration host syncs.

    forward -> softmax_dice + 2x get_separate_loss + 2x get_edge_separate_loss (all weights 1.0, :205-211)
    -> backward -> gradient average over ranks -> Adam(amsgrad) with the poly learning rate (:183,270-273).

Execution modes:
  * eager (default): every kernel is launched from Python (~460 launches, ~16-21 ms of host time per step, hidden behind the GPU's
    ~23 ms); the three sub-regions share each launch (grouped kernels), weight gradients run on a side stream.
  * plan (use_graph=True or "plan"): forward + losses + backward + gradient reduces are captured ONCE (stream capture through
    torch.cuda.graph; shapes are static -- fixed patch size, per-sample top-k of fixed k -- and the token selection and the dropout
    counters live on device, so nothing in the step needs the host) and re-issued per step by the library as a plain LAUNCH LIST
    (csrc/plan.hip: one hipModuleLaunchKernel per node on the capture's stream chains, events for the cross-stream edges; ONE
    Python -> C call per step, or one per gradient phase when data-parallel).  Host cost ~2 ms per step instead of ~16.
  * hipgraph (use_graph="hipgraph"): the same capture replayed with hipGraphLaunch -- slower end to end than eager on this ROCm
    build (~44 us of host time per node); kept as the fallback for graphs the plan cannot express.
Gradients never exist as per-parameter tensors: backward kernels write them into ONE flat buffer laid out in backward-completion
order (cwf.optim.GradSink; the conv weight gradients through one batched split-K reduce per phase), which the fused Adam launch reads.
Multi-GPU gradient averaging (the only data-path collective) is overlapped with backward: the model fires a callback when backward
has passed a cut point (decoder done / everything but the encoder done, ClsWiseFormer.grad_phases); the phase's contiguous slice
(9.8 MB, then 43 MB) is all-reduced (RCCL, summed; Adam reads g / world) on a communication stream that waits for the producing
streams -- the three region streams and the weight-gradient side stream stay in use -- while the encoder's backward (~7 ms) runs;
only the last 14 MB slice is exposed.  In plan mode the cut points are captured as marker nodes on the communication stream; the
launch list is issued up to a marker, the phase's all-reduce is enqueued behind it, and the list continues.  Under hipGraph replay
the collective follows the replay (4 chunks of the flat buffer).
CWF_FORCE_COMM=1 keeps the whole collective path on at world size 1 (a one-rank RCCL group): the way to exercise and profile it on a
one-GPU box.
Checkpoints use the reference layout {'epoch', 'state_dict' with 'module.' prefix, 'optim_dict'} (:248-253)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .optim import FusedAdam, poly_lr


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
    def __init__(self, model, lr=2e-4, weight_decay=1e-5, amsgrad=True, end_epoch=1000, use_graph=False,
                 graph_warmup=2, overlap_comm=True, wgrad_async=True):
        self.model = model
        self.init_lr, self.end_epoch = lr, end_epoch
        phases = model.grad_phases() if hasattr(model, "grad_phases") else None
        self.opt = FusedAdam(model.parameters(), lr=lr, weight_decay=weight_decay, amsgrad=amsgrad, phases=phases)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # collectives on: more than one rank, or a one-rank group with CWF_FORCE_COMM=1 (exercises the RCCL path on a one-GPU box)
        self.comm = self.world > 1 or (dist.is_available() and dist.is_initialized() and os.environ.get("CWF_FORCE_COMM", "0") == "1")
        self.opt.grad_scale = 1.0 / self.world          # the all-reduce SUMS; Adam reads g / world (the DDP average, train_no_amp.py:133)
        if use_graph not in (False, True, None, "plan", "hipgraph"):
            raise ValueError("use_graph must be False, True, 'plan' or 'hipgraph'")
        self.graph_mode = {True: "plan", False: None, None: None}.get(use_graph, use_graph)
        self.use_graph = self.graph_mode is not None
        self.cuda = next(model.parameters()).is_cuda
        # weight gradients on a side stream (they are leaves of backward; the data-gradient chain is its critical path)
        self.wgrad_async = bool(wgrad_async) and self.cuda
        # all-reduce of a phase's slice as soon as backward has passed its cut point (eager; plan mode through captured markers)
        self.overlap_comm = bool(overlap_comm) and self.comm and self.graph_mode != "hipgraph"
        # HIGH priority: its own hardware-queue pool (a default-priority stream can land on the main stream's queue, see
        # cwf.kernels._priority_stream), and a collective should start the moment its slice is final
        self._comm_stream = (torch.cuda.Stream(priority=torch.cuda.Stream.priority_range()[1])
                             if (self.overlap_comm and self.cuda) else None)
        self._works = []
        if self.comm:
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, 0)
        if hasattr(model, "phase_callback"):
            model.phase_callback = self._phase_done
        self._graph = None
        self._plan = None
        self.plan_info = None
        self._static = None
        self._eager_steps = 0
        self._graph_warmup = graph_warmup
        self._in_backward = False
        # The decoder's (full-resolution) weight gradients are queued and released when backward leaves the decoder: they then run
        # beside the backward of the supervision heads / couplers -- 10-20 us kernels behind ~35 us of Python each, where the main
        # stream idles 1.7 ms per step -- instead of competing with the decoder's own HBM-bound data gradients (95.6 -> 96.7
        # volumes/s; neutral while the step was still bound elsewhere).  CWF_DEFER_WGRAD=0 launches them immediately.
        # Level 2 also holds the heads / couplers / decouplers phase's ~40 small weight-gradient launches until backward is in the
        # encoder.  Round 3, launch plan (the host is never the bound): level 2 was best (103.1 -> 104.8 volumes/s with 128 side
        # workgroups) while the eight residual layers' weight gradients still ran on the main stream; with them on the side stream
        # (functional.CarryLink) the SIDE stream ends the step, and it must start earlier: level 1 (0 / 1 / 2: 104.0 / 106.2 / 100.5).
        self.defer_level = int(os.environ.get("CWF_DEFER_WGRAD", "1"))
        self.defer_decoder_wgrad = self.defer_level >= 1

    # ------------------------------------------------------------------------------------------------
    def _phase_done(self, k):
        """Backward has passed cut point k: every gradient of phase k has been produced (conv weight gradients as split-K slabs).
        Reduce the slabs of the layers seen so far into the flat buffer (one launch) and, data-parallel, start the all-reduce of
        the phase's contiguous slice on the communication stream while backward goes on."""
        if not self._in_backward:
            return
        K = kernels_backend()
        if getattr(K, "wgrad_defer", False):
            # the weight gradients queued during the phase that just ended start now.  Decoder (k = 0): beside the GPU-light heads /
            # couplers backward.  Heads / couplers / decouplers (k = 1, level 2): their ~40 small side-stream launches leave the
            # host-bound stretch and are enqueued where the host has slack (the encoder's backward).  The last phase is never held.
            K.wgrad_release()
            K.wgrad_defer = self.wgrad_async and self.defer_level >= 2 and k == 0
        K.wgrad_flush()
        if self.overlap_comm:
            self._allreduce_chunk(k)

    def _allreduce_chunk(self, k):
        lo, hi = self.opt.sink.chunks[k]
        if hi <= lo:
            return
        chunk = self.opt.flat_grad[lo:hi]
        if self._comm_stream is not None:
            K = kernels_backend()
            cs = self._comm_stream
            cs.wait_stream(torch.cuda.current_stream())          # coupler gradients (region streams are joined into this one by autograd)
            for st in K._wg_stream.values():
                cs.wait_stream(st)                               # the batched slab reduce of this phase
            with torch.cuda.stream(cs):
                if _capturing():
                    # plan mode: the cut point becomes a marker node behind both streams; the collective itself is enqueued on this
                    # stream by _run_plan when the launch list reaches the marker
                    K._call("cwf_plan_marker", int(k), K._stream())
                else:
                    self._works.append(dist.all_reduce(chunk, async_op=True))
        elif not _capturing():
            self._works.append(dist.all_reduce(chunk, async_op=True))

    def _fwd_bwd(self, x, target, edge):
        self.opt._ensure()
        sink = self.opt.sink
        sink.begin()
        K = kernels_backend()
        K.wgrad_async = self.wgrad_async       # (from the forward pass on: it prepares weight-gradient operands on the side stream)
        outputs = self.model(x, None)
        loss, parts = total_loss(outputs, target, edge)
        self.opt.zero_grad(set_to_none=True)
        if hasattr(K, "flush_every_default"):
            # graph modes: one reduce per phase, in warm-up too (CWF_PLAN_FLUSH=N: instalments of N layers in plan mode as in eager mode --
            # the descriptor tables are keyed by their rows, so the warm-up steps must already produce the capture's instalments)
            pf = int(os.environ.get("CWF_PLAN_FLUSH", "0"))
            K.flush_every = (pf if (pf > 0 and self.graph_mode == "plan") else (1 << 30)) if self.use_graph else K.flush_every_default
        if hasattr(K, "wgrad_release"):
            K.wgrad_defer = self.wgrad_async and self.defer_decoder_wgrad and hasattr(self.model, "phase_callback")
        self._in_backward = True
        try:
            with sink:
                loss.backward()
                if getattr(K, "wgrad_defer", False):
                    K.wgrad_release()      # (no cut point fired: plain module)
                K.wgrad_flush()            # the last phase (encoder)
        finally:
            self._in_backward = False
            K.wgrad_async = False
            if self.wgrad_async:
                K.join_wgrad_stream()      # weight gradients were produced on the side stream
        self.opt.gather_grads()            # parameters whose gradient came through autograd after all (none on the normal path)
        if self.overlap_comm:
            self._allreduce_chunk(len(sink.chunks) - 1)
            if _capturing() and self._comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self._comm_stream)      # (a capture must end with its forked streams joined)
        return loss.detach(), [p.detach() for p in parts]

    def _finish_comm(self):
        if not self.comm:
            return
        if self.overlap_comm and (self._graph is None or self._plan is not None):
            for w in self._works:
                w.wait()
            self._works = []
            if self._comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self._comm_stream)
        else:                              # graph replay (collectives are not captured): the whole flat buffer after the replay
            works = [dist.all_reduce(c, async_op=True) for c in self.opt.flat_grad.chunk(4)]
            for w in works:
                w.wait()

    def _capture(self, x, target, edge):
        import ctypes
        self._static = (x.clone(), target.clone(), edge.clone())
        want_plan = self.graph_mode == "plan"
        g = torch.cuda.CUDAGraph(keep_graph=True) if want_plan else torch.cuda.CUDAGraph()
        # thread-local capture mode: with a process group alive, RCCL's watchdog thread polls events while this thread captures; in the
        # default (global) mode any such call from another thread invalidates the capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._static_out = self._fwd_bwd(*self._static)
        self._graph = g
        if want_plan:
            K = kernels_backend()
            plan = ctypes.c_void_p()
            rc = K.lib.cwf_plan_create(ctypes.c_void_p(g.raw_cuda_graph()), ctypes.byref(plan))
            if rc == 0:
                info = (ctypes.c_int * 8)()
                K._call("cwf_plan_info", plan, info)
                self._plan = plan
                self.plan_info = dict(zip(("nodes", "kernels", "markers", "streams", "events", "on_stream0", "on_stream1", "on_streams2plus"), list(info)))
            else:
                # a node kind the launch list cannot express (e.g. a copy node): go on EAGERLY -- the step is GPU-bound, eager launches
                # run it as fast as the plan (they only cost more host time), while hipGraphLaunch of the same capture is 30 % slower
                self.plan_info = {"error": int(rc), "detail": (K.lib.cwf_plan_last_error() or b"").decode(), "fallback": "eager"}
                self._graph = None
                self._static = None
                self.use_graph = False
                self.graph_mode = None

    def _run_plan(self):
        import ctypes
        K = kernels_backend()
        nxt, mk = ctypes.c_int(0), ctypes.c_int(-1)
        main = K._stream()
        cs = self._comm_stream
        cs_raw = cs.cuda_stream if cs is not None else None
        pos = 0
        while True:
            K._call("cwf_plan_run", self._plan, main, cs_raw, pos, ctypes.byref(nxt), ctypes.byref(mk))
            if mk.value < 0:
                break
            lo, hi = self.opt.sink.chunks[mk.value]
            if hi > lo and self.comm:
                with torch.cuda.stream(cs):
                    self._works.append(dist.all_reduce(self.opt.flat_grad[lo:hi], async_op=True))
            pos = nxt.value

    def __del__(self):
        try:
            if self._plan is not None:
                kernels_backend().lib.cwf_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    def step(self, x, target, edge, epoch=0):
        """One optimisation step on a rank-local batch.  Returns (loss, [five parts]) as device tensors (no host sync)."""
        self.opt.param_groups[0]["lr"] = float(poly_lr(self.init_lr, epoch, self.end_epoch))   # plain float: checkpoints stay weights_only-loadable
        if self.use_graph and self._graph is None and self._eager_steps >= self._graph_warmup:
            torch.cuda.synchronize()
            self._capture(x, target, edge)
        if self._graph is not None and tuple(x.shape) != tuple(self._static[0].shape):
            raise ValueError("this Trainer captured its step for inputs of shape %s; got %s (a captured step is shape-static: use "
                             "use_graph=False for varying shapes)" % (tuple(self._static[0].shape), tuple(x.shape)))
        if self._graph is not None:
            sx, st, se = self._static
            if sx.data_ptr() != x.data_ptr():
                sx.copy_(x)
            if st.data_ptr() != target.data_ptr():
                st.copy_(target)
            if se.data_ptr() != edge.data_ptr():
                se.copy_(edge)
            if self._plan is not None:
                self._run_plan()
            else:
                self._graph.replay()
            loss, parts = self._static_out
        else:
            loss, parts = self._fwd_bwd(x, target, edge)
            self._eager_steps += 1
        self._finish_comm()
        self.opt.advance_host()
        self.opt.launch()
        return loss, parts


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


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
