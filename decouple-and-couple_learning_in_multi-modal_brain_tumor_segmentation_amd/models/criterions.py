"""Drop-in for the reference ``models.criterions`` on the hot path: ``softmax_dice`` (criterions.py:49-62), the default
``--criterion`` of every training script (train_no_amp.py:83,138).  One fused HIP pass (cwf_dice_ce_*) replaces
F.one_hot + permute + tools.dice_loss + tools.softmax_weighted_loss; the backward is analytic.

The reference's other criteria (softmax_dice2, sigmoid_dice, Generalized_dice, Dual_focal_loss) are selectable by name
there but never defaulted; they are out of scope and raise here."""
from utils import tools


def softmax_dice(output, target):
    """output: probabilities [B,4,D,H,W]; target: int64 [B,D,H,W] in {0..3}.  Returns a scalar tensor."""
    return tools.dice_ce(output, target, num_cls=4)


def _not_built(name):
    def f(*a, **k):
        raise NotImplementedError("criterions.%s is an unused alternative in the reference and is not built; "
                                  "use softmax_dice" % name)
    f.__name__ = name
    return f


softmax_dice2 = _not_built("softmax_dice2")
sigmoid_dice = _not_built("sigmoid_dice")
Generalized_dice = _not_built("Generalized_dice")
Dual_focal_loss = _not_built("Dual_focal_loss")
