"""Fused multi-tensor Adam(amsgrad) -- one HIP launch per optimizer step for all 218 parameters.

Semantics and state layout are those of ``torch.optim.Adam(params, lr, weight_decay, amsgrad=True)`` as the reference
uses it (train_no_amp.py:136,239): L2 decay folded into the gradient, bias-corrected first/second moments, running
max of the second moment; ``state_dict()`` has the same keys ('step', 'exp_avg', 'exp_avg_sq', 'max_exp_avg_sq') so
the checkpoint's 'optim_dict' (train_no_amp.py:252) stays interchangeable.  ``poly_lr`` is
train_no_amp.adjust_learning_rate (:270-273).

Gradients are gathered into ONE persistent flat fp32 buffer (``flat_grad``, 67 MB): the kernel's descriptor table is
then static (safe for hipGraph capture / replay) and data-parallel training all-reduces that single buffer."""
import numpy as np
import torch

from .kernels import backend


def poly_lr(init_lr, epoch, max_epoch, power=0.9):
    return round(init_lr * np.power(1 - epoch / max_epoch, power), 8)


_ACTIVE_SINK = None


def active_sink():
    """The gradient sink backward kernels may write parameter gradients into (None outside Trainer-driven backward passes)."""
    return _ACTIVE_SINK


class GradSink:
    """ONE flat fp32 buffer holding every parameter gradient, laid out in backward-completion order (`phases`: lists of
    parameters; phase k is final once backward has passed the k-th cut point of the model).  While a sink is active the conv
    weight-gradient reduction (cwf_wgrad_reduce_batched) and the coupler Functions write their parameter gradients straight
    into their slices and return None to autograd: no per-parameter gradient tensors, no AccumulateGrad, no concatenation, and
    the data-parallel all-reduce of a phase's contiguous slice can start while backward is still running (cwf.trainer)."""

    def __init__(self, params, phases=None):
        params = [p for p in params if p.requires_grad]
        if phases is None:
            phases = [params]
        known = {id(p) for p in params}
        order = [p for ph in phases for p in ph if id(p) in known]
        if len(order) != len(params) or len({id(p) for p in order}) != len(params):
            raise ValueError("phases must partition the parameter list")
        dev = params[0].device
        self.flat = torch.zeros(sum(p.numel() for p in order), dtype=torch.float32, device=dev)
        self.params = order
        self.views, self.chunks, off = {}, [], 0
        for ph in phases:
            start = off
            for p in ph:
                if id(p) in known:
                    self.views[p.data_ptr()] = self.flat[off:off + p.numel()].view_as(p)
                    off += p.numel()
            self.chunks.append((start, off))
        self.written = set()

    def view(self, p):
        """the slice of `p` (looked up by storage address: autograd may hand a Function a different Python object)"""
        return self.views.get(p.data_ptr())

    def mark(self, p):
        self.written.add(p.data_ptr())

    def begin(self):
        self.written = set()

    def __enter__(self):
        global _ACTIVE_SINK
        self._prev, _ACTIVE_SINK = _ACTIVE_SINK, self
        return self

    def __exit__(self, *a):
        global _ACTIVE_SINK
        _ACTIVE_SINK = self._prev

    def finish(self):
        """After backward: parameters whose gradient came through autograd (.grad) are copied in, parameters that received no
        gradient at all are zeroed (their slices would otherwise keep the previous step's values)."""
        srcs, dsts = [], []
        for p in self.params:
            if p.data_ptr() in self.written:
                continue
            v = self.views[p.data_ptr()]
            if p.grad is not None:
                srcs.append(p.grad); dsts.append(v)
            else:
                v.zero_()
        if srcs:
            torch._foreach_copy_(dsts, srcs)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, phases=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        if len(self.param_groups) != 1:
            raise ValueError("FusedAdam supports one parameter group (the reference uses one, train_no_amp.py:136)")
        self._plist = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        self._phases = phases
        self.sink = None
        self.grad_scale = 1.0
        self.flat_grad = None
        self._table = None
        self._max_n = 0
        self._steps = 0

    # ------------------------------------------------------------------ setup
    def _key(self):
        """Everything the descriptor table points at: parameters AND state tensors (load_state_dict replaces the latter)."""
        k = []
        for p in self._plist:
            st = self.state.get(p) or {}
            k.append((p.data_ptr(),) + tuple(st[n].data_ptr() if n in st else 0 for n in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq")))
        return tuple(k)

    def load_state_dict(self, state_dict):
        """torch semantics; additionally the descriptor table is rebuilt (the moments are new tensors) and the step counter
        continues from the loaded 'step' (train_no_amp.py:252 saves it inside 'optim_dict')."""
        super().load_state_dict(state_dict)
        self._table = None
        steps = [int(float(self.state[p]["step"])) for p in self._plist if p in self.state and "step" in self.state[p]]
        self._steps = max(steps) if steps else 0

    def _ensure(self):
        if self._table is not None and self._table_key == self._key():
            return
        group = self.param_groups[0]
        dev = self._plist[0].device
        total = sum(p.numel() for p in self._plist)
        if self.sink is None or self.sink.flat.numel() != total or self.sink.flat.device != dev or \
                any(self.sink.view(p) is None for p in self._plist):
            self.sink = GradSink(self._plist, self._phases)
            self.flat_grad = self.sink.flat
        rows = []
        for p in self._plist:
            st = self.state[p]
            if not st:
                st["step"] = torch.tensor(float(self._steps))
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if group["amsgrad"]:
                    st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            n = p.numel()
            rows.append([p.data_ptr(), self.sink.view(p).data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                         st["max_exp_avg_sq"].data_ptr() if "max_exp_avg_sq" in st else 0, n])
        self._table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self._table_key = self._key()
        self._max_n = max(r[5] for r in rows)

    # ------------------------------------------------------------------ the three phases of a step
    def gather_grads(self):
        """p.grad -> flat gradient buffer for every parameter the backward kernels did not already write there (all of them
        when no sink was active during backward): one multi-tensor copy; capturable."""
        self._ensure()
        self.sink.finish()
        return self.flat_grad

    def advance_host(self):
        """Host side of a step: the step counter.  Call BEFORE launch()."""
        self._ensure()
        self._steps += 1
        for p in self._plist:
            self.state[p]["step"] += 1

    def launch(self):
        """The single fused kernel.  lr and the step number travel as KERNEL ARGUMENTS (copied at enqueue time; the bias
        corrections are evaluated in double inside cwf_adam_amsgrad): nothing the host rewrites later is read by the GPU, so
        any number of steps may be in flight.  The launch is issued eagerly after a graph replay (it is not captured)."""
        group = self.param_groups[0]
        b1, b2 = group["betas"]
        backend().adam(self._table, len(self._plist), self._max_n, float(group["lr"]), b1, b2, group["eps"], group["weight_decay"],
                       self._steps, group["amsgrad"], hyper_dev=None, grad_scale=self.grad_scale)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._ensure()
        self.sink.begin()               # plain optimizer use: every gradient comes from p.grad
        self.gather_grads()
        self.advance_host()
        self.launch()
        return loss
