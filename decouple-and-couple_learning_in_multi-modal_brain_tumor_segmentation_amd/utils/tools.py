"""Drop-in for the hot-path part of the reference ``utils.tools``: the per-sub-region Dice / weighted-CE losses
(tools.py:8-34,112-231) on the fused HIP loss kernels, the collective helper (:37-41) and the integer Dice / IoU metrics
(:44-61,89-109).  ``medpy`` (Hausdorff, tools.py:5) is not required.

Label decoding happens inside the kernel: a 4-class map uses the label as class; a binary map uses
``(posmask >> label) & 1`` -- sub-region k -> {target == k}; edge sets E1={1,5,6,7}, E2={2,5,6,8}, E4={4,5,7,8}
(tools.py:174-218).  Per-sample CE weights and batch-global Dice sums, as in the reference."""
import torch
import torch.distributed as dist

from cwf import functional as CF

REGION_MASKS = {"01": 1 << 1, "02": 1 << 2, "04": 1 << 3}
EDGE_MASKS = {"01": sum(1 << c for c in (1, 5, 6, 7)), "02": sum(1 << c for c in (2, 5, 6, 8)),
              "04": sum(1 << c for c in (4, 5, 7, 8))}


def dice_ce(output, target, num_cls=4, posmask=0):
    """dice_loss(output, onehot) + softmax_weighted_loss(output, onehot)  (tools.py:8-34) in one fused pass."""
    if output.shape[1] != num_cls:
        raise ValueError("expected %d channels, got %d" % (num_cls, output.shape[1]))
    return CF.dice_ce_loss(output, target, posmask)


def _separate(output, target, masks):
    maps = [output[r] for r in ("01", "02", "04")]
    if all(isinstance(m, CF.LazyProb) for m in maps) and len({(m.scale, tuple(m.logit.shape)) for m in maps}) == 1:
        # training-mode outputs of this package's heads: Dice / CE straight from the low-resolution logits, one label read for
        # the three sub-regions, no full-resolution map (cwf_head_loss_*)
        return CF.head_group_loss(maps, target, [masks[r] for r in ("01", "02", "04")])
    return sum(dice_ce(torch.as_tensor(output[r]) if not isinstance(output[r], CF.LazyProb) else output[r].materialize(), target, 2, masks[r])
               for r in ("01", "02", "04"))


def get_separate_loss(output, target):
    """tools.get_separate_loss (tools.py:112-162): three binary problems {target == k}, k = 1, 2, 3."""
    return _separate(output, target, REGION_MASKS)


def get_edge_separate_loss(output, target):
    """tools.get_edge_separate_loss (tools.py:165-231) on edge codes in {0,1,2,4,5,6,7,8}."""
    return _separate(output, target, EDGE_MASKS)


def all_reduce_tensor(tensor, op=dist.ReduceOp.SUM, world_size=1):
    """tools.all_reduce_tensor (tools.py:37-41)."""
    tensor = tensor.clone()
    dist.all_reduce(tensor, op)
    tensor.div_(world_size)
    return tensor


def dice_score(o, t, eps=1e-8):
    """tools.dice_score (tools.py:44-47) on boolean / 0-1 arrays or tensors."""
    num = 2 * (o * t).sum() + eps
    den = o.sum() + t.sum() + eps
    return num / den


def mIOU(o, t, eps=1e-8):
    """tools.mIOU (tools.py:50-53) on boolean arrays or tensors."""
    num = (o * t).sum() + eps
    den = (o | t).sum() + eps
    return num / den


def softmax_mIOU_score(output, target):
    """tools.softmax_mIOU_score (tools.py:56-61): IoU of classes 1, 2, 3 of integer label maps (numpy or torch)."""
    return [mIOU(o=(output == 1), t=(target == 1)), mIOU(o=(output == 2), t=(target == 2)), mIOU(o=(output == 3), t=(target == 3))]


def softmax_output_dice(output, target):
    """tools.softmax_output_dice (tools.py:89-109): [WT, TC, ET] Dice of integer label maps (numpy or torch)."""
    ret = [dice_score(output > 0, target > 0)]
    ret.append(dice_score((output == 1) | (output == 3), (target == 1) | (target == 3)))
    ret.append(dice_score(output == 3, target == 3))
    return ret
