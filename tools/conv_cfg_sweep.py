"""Time one generic split-bf16 conv shape under each tile configuration (CWF_FORCE_CFG), to tune choose_cfg.  Diagnostic."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(REPO, "decouple-and-couple_learning_in_multi-modal_brain_tumor_segmentation_amd")); sys.path.insert(0, REPO)
    import torch
    from cwf import functional as CF, packing as pk, kernels
    c, s, n = int(sys.argv[2]), int(sys.argv[3]), 2
    prec = sys.argv[4] if len(sys.argv) > 4 else "bf16x3"
    kernels.set_precision(prec)
    K = kernels.backend()
    x = torch.randn((n, s, s, s, c), device="cuda:0")
    w = torch.nn.Parameter(torch.randn((c, c, 3, 3, 3), device="cuda:0") * 0.05)
    spec = CF.ConvSpec(pk.CONV3_S1, c, c)
    packer = CF.WeightPacker(); packer.add(spec, w); packer.refresh()
    sc = torch.ones((n, c), device="cuda:0"); sh = torch.zeros((n, c), device="cuda:0")
    b = torch.zeros(c, device="cuda:0"); y = torch.empty_like(x); stats = K.new_stats(n, c, "cuda:0")
    run = lambda: K.conv(pk.CONV3_S1, x, spec.packed(False), b, c, sc, sh, 0.0, None, None, stats, out=y)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    print("%-6s %-8s c=%d s=%d  %.1f us  %.0f TFLOP/s" % (prec, os.environ.get("CWF_FORCE_CFG", "auto"), c, s, ms * 1e3, 2.0 * 27 * c * c * n * s ** 3 / ms / 1e9), flush=True)
else:
    precs = sys.argv[1:] or ["bf16x3"]
    for prec in precs:
        for c, s in ((128, 16), (64, 32), (32, 64)):
            for cfg in (None, "4,4,1", "2,4,2", "2,4,4", "4,2,4", "4,1,4", "1,4,4", "1,2,4", "1,1,4"):
                env = dict(os.environ)
                if cfg: env["CWF_FORCE_CFG"] = cfg
                subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(c), str(s), prec], env=env)
