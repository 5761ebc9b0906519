"""Training-step harness: the reference hot loop (train_no_amp.py:181-239) without its per-iteration host syncs.

    forward -> softmax_dice + 2x get_separate_loss + 2x get_edge_separate_loss (all weights 1.0, :205-211)
    -> backward (gradient all-reduce overlapped) -> Adam(amsgrad) with the poly learning rate (:183,270-273).
Checkpoints use the reference layout {'epoch', 'state_dict' with 'module.' prefix, 'optim_dict'} (:248-253)."""
from __future__ import annotations

import torch

from .optim import FusedAdam, poly_lr
from .parallel import GradSync


def total_loss(outputs, target, edge):
    from models import criterions
    from utils import tools
    parts = [criterions.softmax_dice(outputs[0], target), tools.get_separate_loss(outputs[1], target),
             tools.get_edge_separate_loss(outputs[2], edge), tools.get_separate_loss(outputs[3], target),
             tools.get_edge_separate_loss(outputs[4], edge)]
    return parts[0] + parts[1] + parts[2] + parts[3] + parts[4], parts


class Trainer:
    def __init__(self, model, lr=2e-4, weight_decay=1e-5, amsgrad=True, end_epoch=1000, bucket_mb=16.0):
        self.model = model
        self.init_lr, self.end_epoch = lr, end_epoch
        self.opt = FusedAdam(model.parameters(), lr=lr, weight_decay=weight_decay, amsgrad=amsgrad)
        self.sync = GradSync(model.parameters(), bucket_mb=bucket_mb)
        self.sync.broadcast_parameters(list(model.parameters()) + list(model.buffers()))

    def step(self, x, target, edge, epoch=0):
        """One optimisation step on a rank-local batch.  Returns the loss tensors (still on device, no sync)."""
        for g in self.opt.param_groups:
            g["lr"] = poly_lr(self.init_lr, epoch, self.end_epoch)
        outputs = self.model(x, None)
        loss, parts = total_loss(outputs, target, edge)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.sync.finish()
        self.opt.step()
        return loss.detach(), [p.detach() for p in parts]


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
