"""The unit test's 'fwd' case through conv_ws and through the tap-table kernel, both against the emulator.  Diagnostic."""
import os, sys, math
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "decouple-and-couple_learning_in_multi-modal_brain_tumor_segmentation_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from cwf import functional as CF, packing as pk, kernels
from oracle.kernel_emul import EmulBackend
from test_kernels_gpu import rnd, _packed
E = EmulBackend(); DEV = "cuda:0"
hip = kernels.backend()
cin, cout, size, n, prec = 32, 32, (16, 16, 32), 2, sys.argv[1] if len(sys.argv) > 1 else "bf16"
d, h, w_ = size
op = pk.CONV3_S1
x = rnd(n, d, h, w_, cin, seed=31)
w = rnd(cout, cin, 3, 3, 3, seed=32, scale=1.0 / math.sqrt(cin * 27))
b = rnd(cout, seed=33, scale=0.1)
in_scale = rnd(n, cin, seed=34).abs() + 0.5
in_shift = rnd(n, cin, seed=35)
res = rnd(n, d, h, w_, cout, seed=37)
spec = _packed(CF.ConvSpec(op, cin, cout), w, prec)
st_ref = E.new_stats(n, cout, None)
y_ref = E.conv(op, x, None, b, cout, in_scale, in_shift, 0.01, res, None, st_ref, w_ref=w)
for units in (1 << 30, 1, 1 << 30, 1):
    hip.lib.cwf_debug_ws_min_units(units); hip.lib.cwf_debug_ws_x3(1)
    st = hip.new_stats(n, cout, DEV)
    y = hip.conv(op, x.to(DEV), spec.wpk16_f, b.to(DEV), cout, in_scale.to(DEV), in_shift.to(DEV), 0.01, res.to(DEV), None, st, prec=prec)
    torch.cuda.synchronize()
    err = (y.cpu() - y_ref).abs()
    e = err.nan_to_num(99.0).view(n, d // 4, 4, h // 4, 4, w_ // 16, 16, cout).amax(dim=(2, 4, 6, 7))
    bad = (e > 0.05).nonzero().tolist()
    print("min_units", units, "max err", float(err.nan_to_num(99.0).max()), "bad tiles", len(bad), bad[:12])
