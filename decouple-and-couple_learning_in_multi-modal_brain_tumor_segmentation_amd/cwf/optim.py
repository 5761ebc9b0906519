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


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        if len(self.param_groups) != 1:
            raise ValueError("FusedAdam supports one parameter group (the reference uses one, train_no_amp.py:136)")
        self._plist = [p for p in self.param_groups[0]["params"] if p.requires_grad]
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
        if self.flat_grad is None or self.flat_grad.numel() != total or self.flat_grad.device != dev:
            self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        rows, off = [], 0
        for p in self._plist:
            st = self.state[p]
            if not st:
                st["step"] = torch.tensor(float(self._steps))
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if group["amsgrad"]:
                    st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            n = p.numel()
            rows.append([p.data_ptr(), self.flat_grad.data_ptr() + 4 * off, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                         st["max_exp_avg_sq"].data_ptr() if "max_exp_avg_sq" in st else 0, n])
            off += n
        self._table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self._table_key = self._key()
        self._max_n = max(r[5] for r in rows)

    # ------------------------------------------------------------------ the three phases of a step
    def gather_grads(self):
        """p.grad -> flat_grad (one concatenation; capturable).  Parameters without a gradient contribute zeros."""
        self._ensure()
        torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self._plist], out=self.flat_grad)
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
                       self._steps, group["amsgrad"], hyper_dev=None)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.gather_grads()
        self.advance_host()
        self.launch()
        return loss
