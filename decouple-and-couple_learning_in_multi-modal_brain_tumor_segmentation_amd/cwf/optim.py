"""Fused multi-tensor Adam(amsgrad) -- one HIP launch per optimizer step for all 218 parameters.

Semantics and state layout are those of ``torch.optim.Adam(params, lr, weight_decay, amsgrad=True)`` as the reference
uses it (train_no_amp.py:136,239): L2 decay folded into the gradient, bias-corrected first/second moments, running
max of the second moment; ``state_dict()`` has the same keys ('step', 'exp_avg', 'exp_avg_sq', 'max_exp_avg_sq') so
the checkpoint's 'optim_dict' (train_no_amp.py:252) stays interchangeable.  ``poly_lr`` is
train_no_amp.adjust_learning_rate (:270-273)."""
import numpy as np
import torch

from .kernels import backend


def poly_lr(init_lr, epoch, max_epoch, power=0.9):
    return round(init_lr * np.power(1 - epoch / max_epoch, power), 8)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        self._tables = {}

    def _table(self, gi, plist):
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in plist)
        hit = self._tables.get(gi)
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        rows = []
        for p in plist:
            st = self.state[p]
            rows.append([p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                         st["max_exp_avg_sq"].data_ptr() if "max_exp_avg_sq" in st else 0, p.numel()])
        table = torch.tensor(rows, dtype=torch.int64).to(plist[0].device)
        max_n = max(r[5] for r in rows)
        self._tables[gi] = (key, table, max_n)
        return table, max_n

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        K = backend()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            for p in plist:
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    if group["amsgrad"]:
                        st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                st["step"] += 1
            step = int(self.state[plist[0]]["step"])
            table, max_n = self._table(gi, plist)
            b1, b2 = group["betas"]
            K.adam(table, len(plist), max_n, float(group["lr"]), b1, b2, group["eps"], group["weight_decay"], step, group["amsgrad"])
        return loss
