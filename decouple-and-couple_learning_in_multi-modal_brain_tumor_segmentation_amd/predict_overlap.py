"""Sliding-window inference -- counterpart of the hot-path part of the reference ``predict_overlap.py``
(BASELINE.json configs[3]: full 240x240x155 volume on one MI355X).

``tailor_and_concat`` (predict_overlap.py:31-58): eight fixed 128^3 windows over a [B,4,240,240,>=155] volume
(H, W starts {0, 112}, D starts {0, 27}) and hard-overwrite stitching -- INCLUDING the reference's depth-axis quirk
(`y[..., 128:155] = window(27:155)[..., 96:123]`, i.e. voxels 123..149 land at 128..154), reproduced
for parity.  The reference runs the eight windows as eight sequential B=1 forwards; here they are independent samples of
one batch, so they go through the kernels as ONE batch-8 forward.

``validate_softmax`` (predict_overlap.py:103-171) minus file I/O (nibabel / imageio are not hot-path): argmax over
classes and the WT / TC / ET Dice of ``utils.tools.softmax_output_dice``.

``flip_tta`` (N4; predict_simple.py:333-349, predict_cls.py:184-203): the reference's 8-flip test-time augmentation -- the
mean over all subsets of the three spatial axes of ``softmax(model(flip(x))[0]).flip`` (the reference re-applies softmax to
the model's already-normalised output; kept, ``resoftmax=True``).  The eight flipped copies are independent samples, so
they run as batches instead of eight sequential B=1 forwards.
"""
import torch

from utils import tools

WINDOWS = [(0, 0, 0), (0, 112, 0), (112, 0, 0), (112, 112, 0), (0, 0, 27), (0, 112, 27), (112, 0, 27), (112, 112, 27)]


def tailor_and_concat(x, missing_modal, model, target=None, batched=True):
    """x: [B,4,240,240,D>=155].  Returns the stitched probability volume [B,4,240,240,155]."""
    wins = [x[..., a:a + 128, b:b + 128, c:c + 128] for a, b, c in WINDOWS]
    if batched:
        nb = x.shape[0]
        out = model(torch.cat(wins, dim=0), missing_modal)[0]
        if out.is_cuda and out.dtype == torch.float32 and tuple(out.shape[1:]) == (4, 128, 128, 128) and x.shape[-1] >= 155:
            from cwf.kernels import backend
            return backend().stitch_windows(out, nb)          # one launch (cwf_stitch_windows) instead of clone + 8 slice copies
        t = [out[i * nb:(i + 1) * nb] for i in range(8)]
    else:
        t = [model(w, missing_modal)[0] for w in wins]
    y = x.clone()                       # the reference relies on 4 modalities == 4 classes (predict_overlap.py:43)
    y[..., :128, :128, :128] = t[0]
    y[..., :128, 128:240, :128] = t[1][..., :, 16:128, :]
    y[..., 128:240, :128, :128] = t[2][..., 16:128, :, :]
    y[..., 128:240, 128:240, :128] = t[3][..., 16:128, 16:128, :]
    y[..., :128, :128, 128:155] = t[4][..., 96:123]
    y[..., :128, 128:240, 128:155] = t[5][..., :, 16:128, 96:123]
    y[..., 128:240, :128, 128:155] = t[6][..., 16:128, :, 96:123]
    y[..., 128:240, 128:240, 128:155] = t[7][..., 16:128, 16:128, 96:123]
    return y[..., :155]


FLIPS = [(), (2,), (3,), (4,), (2, 3), (2, 4), (3, 4), (2, 3, 4)]          # order of predict_simple.py:333-347


@torch.no_grad()
def flip_tta(x, missing_modal, forward, resoftmax=True, batch=8):
    """x [B,4,D,H,W]; forward(xb, missing_modal) -> probabilities [b,C,D,H,W].  Returns the 8-flip average [B,C,D,H,W]."""
    nb = x.shape[0]
    acc = None
    for g0 in range(0, 8, max(1, batch // max(nb, 1))):
        group = FLIPS[g0:g0 + max(1, batch // max(nb, 1))]
        xb = torch.cat([x.flip(dims=f) if f else x for f in group], dim=0)
        out = forward(xb, missing_modal)
        for k, f in enumerate(group):
            o = out[k * nb:(k + 1) * nb]
            if resoftmax:
                o = torch.softmax(o, dim=1)
            o = o.flip(dims=f) if f else o
            acc = o.clone() if acc is None else acc.add_(o)
    return acc / 8.0


@torch.no_grad()
def validate_softmax(x, target, model, deterministic=True, use_TTA=False, with_miou=False):
    """One subject: stitched probabilities -> label map (argmax; class 3 stands for BraTS label 4) -> [WT, TC, ET] Dice.
    ``deterministic`` zeroes the stem dropout that the reference leaves on in eval mode.  ``with_miou`` adds the per-class
    IoU list of tools.softmax_mIOU_score (what predict_simple.py reports next to Dice) as a fourth result."""
    model.eval()
    saved = model.Unet_list.InitConv.dropout
    if deterministic:
        model.Unet_list.InitConv.dropout = 0.0
    try:
        if use_TTA:         # each flipped volume goes through the 8-window stitcher (one batch-8 forward per flip)
            prob = flip_tta(x, None, lambda xb, mm: tailor_and_concat(xb, mm, model), batch=1)
        else:
            prob = tailor_and_concat(x, None, model)
    finally:
        model.Unet_list.InitConv.dropout = saved
    if prob.is_cuda and prob.dtype == torch.float32 and prob.dim() == 5 and prob.shape[1] == 4:
        from cwf.kernels import backend                     # argmax + WT/TC/ET counts in one launch (cwf_argmax_dice)
        tgt = None if target is None else target[..., :155].long()
        if with_miou and tgt is not None:
            seg, d, iou = backend().argmax_dice(prob, tgt, miou=True)          # argmax + Dice + IoU counts, still one launch
            return seg, prob, [d[0], d[1], d[2]], [iou[0], iou[1], iou[2]]
        seg, d = backend().argmax_dice(prob, tgt)
        res = (seg, prob, (None if d is None else [d[0], d[1], d[2]]))
        return res + (None,) if with_miou else res
    seg = prob.argmax(1)
    dice = tools.softmax_output_dice(seg, target[..., :155]) if target is not None else None
    if with_miou:
        return seg, prob, dice, (tools.softmax_mIOU_score(seg, target[..., :155]) if target is not None else None)
    return seg, prob, dice
