"""bench.py -- training volumes/sec of the ClsWiseFormer hot path on MI355X (BASELINE.json metric).

One "step" = forward + 5 losses + backward + Adam(amsgrad) on a rank-local batch of 2 synthetic 4-modality 128^3
volumes (BASELINE.json configs[1]; for N > 1 the same per-rank batch = weak scaling, gradients averaged over ranks by
cwf.trainer.Trainer's RCCL all-reduce of the flat gradient buffer, see its docstring for what overlaps with backward).
Inputs are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     -- the dominant kernel (3x3x3 conv 16->16 @128^3 x2, conv16s_kernel in the bf16 modes): ALGORITHMIC bytes per launch
                  (x read once + y written once, fp32) / average launch duration measured here with HIP events on the launch stream,
                  against the 8 TB/s HBM peak (fp32 mode: algorithmic FLOPs against the dense fp32 MFMA peak); `traffic` = HBM bytes
                  per launch from the committed rocprofv3 --pmc summary
  cpu_baseline -- the CPU oracle (oracle/reference_model.py, kind "port") timed on this box's host cores, B=1 128^3.
  val_dice / max_rel_logit_err -- the other half of BASELINE.json's metric: WT/TC/ET Dice (tools.softmax_output_dice,
                  reference utils/tools.py:89-109) of the HIP model's argmax map against the CPU oracle's argmax map, and the
                  largest logit deviation relative to the largest |logit|, on the cpu_baseline sample (same weights, eval mode).
"""
import argparse
import json
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # before the first HIP call: see cwf/__init__.py (stream -> hardware-queue multiplexing)
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "decouple-and-couple_learning_in_multi-modal_brain_tumor_segmentation_amd")
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
HBM_PEAK_GBS = 8000.0
FLOP_PER_TRAIN_VOLUME = 1.594e12   # 3 x 531.3 GFLOP per 128^3 volume


def log(msg):
    print("[bench %s] %s" % (time.strftime("%H:%M:%S"), msg), file=sys.stderr, flush=True)


def usable_cores():
    """Host threads this process may really use: CPU affinity capped by the cgroup CPU quota (the GPU boxes expose every
    host core in the affinity mask but grant a share of them; oversubscribing OpenMP threads is catastrophically slow)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


# HBM bytes per launch of the dominant kernel come from the committed rocprofv3 --pmc summary (FETCH_SIZE x2 [gfx950 correction]
# + WRITE_SIZE), not from a constant in this file: profiles/dominant_kernel_traffic.json names the kernel build it was measured
# on (source hash of conv_bf16.hip); a stale or missing entry yields traffic = null.
TRAFFIC_FILE = os.path.join(REPO, "profiles", "dominant_kernel_traffic.json")


def conv16s_source_sha():
    """sha256 (16 hex digits) of the dominant kernel's own source region in conv_bf16.hip (from its header comment to the end of its
    launcher): the committed PMC record stays valid while THAT text is unchanged, whatever else the file gains."""
    import hashlib
    src = open(os.path.join(PKG, "csrc", "conv_bf16.hip"), "rb").read()
    a, b = src.find(b"// conv16s: conv16 with a SLIDING WINDOW"), src.find(b"// Pointwise family")
    return hashlib.sha256(src[a:b] if 0 <= a < b else src).hexdigest()[:16]


def measured_traffic(precision):
    try:
        rec = json.load(open(TRAFFIC_FILE))[precision]
        if rec.get("conv16s_sha16") != conv16s_source_sha():
            return None, "stale: %s was measured on another build of conv16s_kernel" % rec.get("source")
        return float(rec["hbm_bytes_per_launch"]), rec.get("source")
    except Exception as e:          # missing file / precision: report null rather than a guess
        return None, "no committed PMC summary (%s)" % type(e).__name__
MFMA_PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0 / 3.0, "bf16": 2500.0}   # dense peaks; bf16x3 issues 3 MFMAs per product


def dominant_kernel_roofline(dev, precision, iters=20):
    """The step's dominant kernel: 3x3x3 conv, 16->16 channels, 128^3, batch 2 (all eight full-resolution convs of the
    encoder/decoder blocks and their data gradients run it), with the fused InstanceNorm+ReLU prologue and the statistics
    epilogue it has inside EnBlock / DeBlock.  fp32 mode: conv_mfma_kernel<4,1,4> (exact f32 MFMA, MFMA-bound: AI 108 F/B
    > ridge 19.6).  bf16x3 / bf16 modes: conv16_kernel (split-bf16 MFMA operands, fp32 storage): HBM-bound by design --
    algorithmic bytes = x read once + y written once in fp32 = 2*N*V*16*4 B per launch."""
    from cwf import functional as CF, packing as pk
    from cwf.kernels import backend
    K = backend()
    n, s, c = 2, 128, 16
    x = torch.randn((n, s, s, s, c), device=dev)
    w = torch.nn.Parameter(torch.randn((c, c, 3, 3, 3), device=dev) * 0.05)
    b = torch.zeros(c, device=dev)
    spec = CF.ConvSpec(pk.CONV3_S1, c, c)
    packer = CF.WeightPacker()
    packer.add(spec, w)
    packer.refresh()
    sc = torch.ones((n, c), device=dev)
    sh = torch.zeros((n, c), device=dev)
    y = torch.empty_like(x)
    stats = K.new_stats(n, c, dev)
    for _ in range(3):
        K.conv(pk.CONV3_S1, x, spec.packed(False), b, c, sc, sh, 0.0, None, None, stats, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        K.conv(pk.CONV3_S1, x, spec.packed(False), b, c, sc, sh, 0.0, None, None, stats, out=y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * 27 * c * c * n * s ** 3
    alg_bytes = 2.0 * n * s ** 3 * c * 4          # read x once + write y once, fp32
    tflops = flops / (ms * 1e-3) / 1e12
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    if precision == "fp32":
        out = {"kernel": "conv_mfma_kernel<4,1,4> (v_mfma_f32_16x16x4_f32) 3x3x3 16->16 @128^3 x2", "bound": "mfma",
               "achieved": round(tflops, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / FP32_MFMA_PEAK_TFLOPS, 4)}
    else:
        out = {"kernel": "conv16s_kernel<%s> (v_mfma_f32_16x16x32_bf16, fp32 storage) 3x3x3 16->16 @128^3 x2" % precision, "bound": "hbm",
               "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
    traffic, traffic_src = measured_traffic(precision)
    out.update({"traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(ms, 4),
                "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_flops_per_launch": flops,
                "algorithmic_tflops": round(tflops, 2), "frac_of_mfma_peak_for_mode": round(tflops / MFMA_PEAK_TFLOPS[precision], 4)})
    return out


def val_dice_vs_cpu(dev, state, x, precision):
    """GPU argmax vs CPU-oracle argmax on identical inputs and weights (eval mode, stem dropout off)."""
    from oracle import reference_model as rm
    from models.clswiseformer.cls_wise_former import get_cls_wise_former
    from utils import tools
    m = get_cls_wise_former(dataset="brats", _conv_repr=True, _pe_type="fixed")
    m.load_state_dict(state, strict=False)
    m.Unet_list.InitConv.dropout = 0.0
    m = m.to(dev).eval()
    m.collect_aux = True
    with torch.no_grad():
        t0 = time.time()
        ref, aux = rm.forward(state, x, return_aux=True)
        cpu_fwd = time.time() - t0
        out = m(x.to(dev), None)
        seg_gpu = out[0].argmax(1).cpu()
        logits = m.aux["logits"].float().cpu()
    seg_cpu = ref[0].argmax(1)
    dice = [float(d) for d in tools.softmax_output_dice(seg_gpu.numpy(), seg_cpu.numpy())]
    rel = float((logits - aux["logits"]).abs().max() / aux["logits"].abs().max())
    return {"WT": round(dice[0], 6), "TC": round(dice[1], 6), "ET": round(dice[2], 6),
            "argmax_mismatch_voxels": int((seg_gpu != seg_cpu).sum()), "voxels": int(seg_cpu.numel()),
            "sample": "B=1 4x128^3 synthetic volume, generator-defined weights, eval mode, %s vs fp32 CPU oracle" % precision}, rel, cpu_fwd


def cpu_baseline(steps=3, dev=None, precision="bf16x3"):
    """The reference train step (fwd + 5 losses + bwd + Adam amsgrad) as restated in oracle/reference_model.py, on the host
    cores of this box: B=1, 128^3, fp32, 1 warm-up + `steps` timed steps (bounded sample of the same workload)."""
    from oracle import reference_model as rm
    from utils import synthetic as syn
    cores = usable_cores()
    torch.set_num_threads(cores)
    log("cpu_baseline: %d host threads" % cores)
    state = syn.det_state_dict(rm.param_shapes())
    x, target, edge = syn.synthetic_batch([0], (128, 128, 128))
    extra = {}
    if dev is not None:
        extra["val_dice"], extra["max_rel_logit_err"], fwd_s = val_dice_vs_cpu(dev, state, x, precision)
        log("val_dice leg done (CPU oracle forward %.1f s): %s, max rel logit err %.2e" % (fwd_s, extra["val_dice"], extra["max_rel_logit_err"]))
    tr = rm.CpuTrainer(state)
    t0 = time.time()
    tr.step(x, target, edge)
    log("cpu_baseline: warm-up step %.1f s" % (time.time() - t0))
    t0 = time.time()
    for i in range(steps):
        tr.step(x, target, edge)
        log("cpu_baseline: step %d done" % i)
    dt = (time.time() - t0) / steps
    return {"value": round(1.0 / dt, 4), "unit": "volumes/s", "cores": cores, "kind": "port",
            "sample": "%d timed steps (+1 warm-up) of B=1 4x128^3 fwd+5 losses+bwd+Adam(amsgrad), fp32, torch CPU ops" % steps,
            "sec_per_step": round(dt, 3)}, extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (100 x 23 ms: a timed region long enough for 1 Hz GPU-busy samplers)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2, help="rank-local batch (independent B=1 samples, SURVEY F2)")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--mode", default="plan", choices=["plan", "eager", "hipgraph"],
                    help="plan: the step is captured once and re-issued by the library as a launch list (csrc/plan.hip: one plain launch per "
                         "node, ~2 ms of host time per step); eager: every kernel launched from Python (~14-16 ms of host time per step); "
                         "hipgraph: the captured graph replayed with hipGraphLaunch (~44 us of host time per node on ROCm 7.2)")
    ap.add_argument("--graph", action="store_true", help="alias of --mode hipgraph (kept for older command lines)")
    ap.add_argument("--no-wgrad-async", action="store_true", help="keep the weight gradients on the main stream (A/B)")
    ap.add_argument("--precision", default="bf16x3", choices=["fp32", "bf16x3", "bf16"],
                    help="MFMA operand form of the conv family (storage and accumulation are fp32 in every mode)")
    ap.add_argument("--wgrad-precision", default="bf16", choices=["bf16x3", "bf16", "fp32"],
                    help="operand form of the weight-gradient kernels (default: single bf16 products, fp32 accumulate -- gradient norms stay "
                         "within the fp32 reference's own noise of the float64 gradients, tests/test_model_gpu.py)")
    ap.add_argument("--dgrad-precision", default="bf16", choices=["bf16x3", "bf16", "fp32"], help="operand form of the data-gradient kernels")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # Rehearsal switch (not used by the driver): CWF_BENCH_REHEARSE=1 runs the N > 1 code path with all ranks on cuda:0 over
    # gloo, for a one-GPU box -- the numbers mean nothing, the control flow (broadcast, all-reduce, barriers, MAX) is what runs.
    rehearse = os.environ.get("CWF_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    # CWF_FORCE_COMM=1 under a one-rank launcher (RANK set): a one-rank RCCL group, so that the collective path (comm stream,
    # per-phase all-reduce) runs and can be profiled on a one-GPU box.  The printed line is then marked "comm_forced".
    forced = world == 1 and "RANK" in os.environ and os.environ.get("CWF_FORCE_COMM", "0") in ("1", "init")
    use_dist = world > 1 or forced
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            opts = dist.ProcessGroupNCCL.Options()
            opts.is_high_priority_stream = True          # RCCL's own stream in the high-priority queue pool, off the compute streams' queues
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), pg_options=opts)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from models.clswiseformer.cls_wise_former import get_cls_wise_former
    from cwf.trainer import Trainer
    from cwf import kernels
    if args.precision == "fp32":
        args.wgrad_precision = args.dgrad_precision = "fp32"
    kernels.set_precision(args.precision, args.wgrad_precision, args.dgrad_precision)
    from utils import synthetic as syn

    torch.manual_seed(1000 + rank)                       # train_no_amp.py:85 seed, per-rank streams
    model = get_cls_wise_former(dataset="brats", _conv_repr=True, _pe_type="fixed").to(dev).train()   # random init, dropout ON
    trainer = Trainer(model, lr=2e-4, weight_decay=1e-5, amsgrad=True, end_epoch=1000, use_graph={"plan": "plan", "eager": False, "hipgraph": "hipgraph"}["hipgraph" if args.graph else args.mode],
                      wgrad_async=not args.no_wgrad_async)
    size = (args.size,) * 3
    idx = [rank * args.batch + i for i in range(args.batch)]
    x, target, edge = syn.synthetic_batch(idx, size)
    x, target, edge = x.to(dev), target.to(dev), edge.to(dev)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    log("rank %d: model + inputs ready" % rank)
    for i in range(args.warmup):
        trainer.step(x, target, edge, epoch=0)
        torch.cuda.synchronize()
        log("rank %d: warm-up step %d done" % (rank, i))
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, _ = trainer.step(x, target, edge, epoch=0)
    host_dt = time.perf_counter() - t0                   # host enqueue time (no sync): ~ dt means launch-bound
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss)
    # host cost of issuing one step, UNTHROTTLED: with the GPU idle at the start the launch queue never fills, so this is what the
    # host spends (the figure above saturates at the GPU's step time once the host runs ahead and blocks on a full queue)
    host_ms = []
    for _ in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        trainer.step(x, target, edge, epoch=0)
        host_ms.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    host_ms = sorted(host_ms)[len(host_ms) // 2]
    log("rank %d: timed %d steps in %.3f s" % (rank, args.steps, dt))

    if rank == 0:
        vols = world * args.batch * args.steps
        value = vols / dt
        out = {
            "metric": "training volumes/sec (4x128^3)", "value": round(value, 3), "unit": "volumes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x3": "bf16x3 forward (split-bf16 MFMA operands hi.hi+hi.lo+lo.hi: logits within 1e-3 of the fp32 reference)",
                      "bf16": "bf16 forward"}[args.precision] +
                     ("" if args.precision == "fp32" else " / %s data-gradient / %s weight-gradient MFMA operands; f32 accumulate, f32 storage, f32 master weights"
                      % (args.dgrad_precision, args.wgrad_precision)), "data": "synthetic",
            "config": {"workload": "%s: %dxMI355X, batch %d per GPU, 4-modality %d^3 synthetic BraTS patches; "
                                   "fwd + softmax_dice + 4 sub-region/edge losses + bwd + Adam(amsgrad)%s; random-init weights, dropout on"
                                   % ("configs[1]" if world == 1 else ("configs[2]" if world == 8 else "configs[1] per GPU, data-parallel"),
                                      world, args.batch, args.size, "" if world == 1 else " + gradient all-reduce (RCCL)"),
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world, "precision": args.precision, "dgrad_precision": args.dgrad_precision,
                       "wgrad_precision": args.wgrad_precision, "mode": ("hipgraph" if args.graph else args.mode) if trainer._graph is not None or args.mode == "eager" else "eager (not captured)",
                       "plan": trainer.plan_info},
            "final_loss": round(final_loss, 5), "host_enqueue_ms_per_step": round(host_ms, 2), "host_ms_per_step_in_timed_region": round(host_dt / args.steps * 1e3, 2),
            "end_to_end": {"tflops": round(value * FLOP_PER_TRAIN_VOLUME * (args.size / 128.0) ** 3 / 1e12, 2),
                           "frac_mfma_peak_for_mode": round(value * FLOP_PER_TRAIN_VOLUME * (args.size / 128.0) ** 3 / 1e12 / world / MFMA_PEAK_TFLOPS[args.precision], 4)},
        }
        out["roofline"] = dominant_kernel_roofline(dev, args.precision)
        log("roofline leg done: %s" % json.dumps(out["roofline"]))
        if forced:
            out["comm_forced"] = True                    # one-rank RCCL group: the collective path ran (CWF_FORCE_COMM=1)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], extra = cpu_baseline(args.cpu_steps, dev, args.precision)
            out.update(extra)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
