"""Gradient accuracy of precision configurations against the reference's float64 gradients (tests/golden/model_{64,128}.npz) and,
elementwise, against the first configuration run:   python tools/grad_err.py bf16x3:-:- bf16x3:bf16:- bf16x3:bf16:bf16"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "decouple-and-couple_learning_in_multi-modal_brain_tumor_segmentation_amd")); sys.path.insert(0, REPO)
import numpy as np, torch
from cwf import kernels
from oracle import reference_model as rm
from utils import synthetic as syn
from models.clswiseformer.cls_wise_former import get_cls_wise_former
from models import criterions
from utils import tools
BASE = {}
for cfg in sys.argv[1:]:
    mode, wg, dg = [None if v == "-" else v for v in cfg.split(":")]
    kernels.set_precision(mode, wg, dg)
    for tag, size in (("64", (64,) * 3), ("128", (128,) * 3)):
        g = np.load(os.path.join(REPO, "tests", "golden", "model_%s.npz" % tag))
        m = get_cls_wise_former(dataset="brats", _conv_repr=True, _pe_type="fixed")
        m.load_state_dict(syn.det_state_dict(rm.param_shapes()), strict=False)
        m.Unet_list.InitConv.dropout = 0.0
        m = m.to("cuda:0").eval()
        x, target, edge = [t.to("cuda:0") for t in syn.synthetic_batch([0], size)]
        outs = m(x, None)
        loss = criterions.softmax_dice(outs[0], target) + tools.get_separate_loss(outs[1], target) + tools.get_edge_separate_loss(outs[2], edge) + \
            tools.get_separate_loss(outs[3], target) + tools.get_edge_separate_loss(outs[4], edge)
        loss.backward()
        names, l2, noise = list(g["grad_names"]), g["grad_l2_f64"], g["grad_noise_ref32"]
        errs = []
        for n, p in m.named_parameters():
            i = names.index(n)
            if l2[i] > 1e-7:
                errs.append((abs(float(p.grad.double().norm()) - l2[i]) / l2[i], float(noise[i]), n))
        errs.sort(reverse=True)
        full = []
        for key in g.files:
            if key.startswith("grad::"):
                ref = torch.from_numpy(g[key]).double()
                got = dict(m.named_parameters())[key[6:]].grad.double().cpu()
                if float(ref.norm()) > 1e-7:
                    full.append((float((got - ref).norm() / ref.norm()), key[6:]))
        cur = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
        msg = ""
        if tag in BASE:
            d = sorted(((float((cur[n] - b).norm() / b.norm()), n) for n, b in BASE[tag].items() if float(b.norm()) > 1e-12), reverse=True)
            msg = " | elementwise vs first config: worst %.2e %s, median %.2e" % (d[0][0], d[0][1], d[len(d) // 2][0])
        else:
            BASE[tag] = cur
        print("%s size %s: worst grad-norm errs vs f64 %s ; worst full-gradient err %.2e%s" %
              (cfg, tag, ["%.1e(ref32 noise %.0e) %s" % e for e in errs[:3]], max(full)[0], msg), flush=True)
