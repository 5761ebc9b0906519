"""N2 -- training harness with the command line of the reference's train_no_amp.py (flags :27-107, loop :110-263), rebuilt
around cwf.trainer.Trainer.  Launch one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train_no_amp.py --batch_size 2

What is kept from the reference: flag names and defaults, seeding (:123-126), poly learning rate re-evaluated every
iteration from the epoch (:182, :266-269), the five-term loss (:205-210), rank-`print_rank` logging, the checkpoint cadence
(:243-246) and layout {'epoch','state_dict' ('module.' keys),'optim_dict'} (:248-262), weights-only resume (:147-151).
What differs, deliberately: no per-iteration `.cpu()` / `.item()` / barrier + five scalar all-reduces on the critical path
(:191-204,:216-224) -- the log line is produced every `--log_every` iterations from device tensors copied asynchronously;
the gradient exchange is the Trainer's flat all-reduce; data comes from utils.data (the reference's `data/` package is absent).
"""
import argparse
import logging
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # before the first HIP call: see cwf/__init__.py (stream -> hardware-queue multiplexing)
import random
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)


def _bool(v):
    """The reference declares its switches with type=bool (any non-empty string is True, train_no_amp.py:79,87,103);
    here 'false'/'0'/'no' mean False."""
    return str(v).strip().lower() not in ("", "0", "false", "no", "off")


def build_parser():
    local_time = time.strftime("%Y%m%d %H%M%S", time.localtime())
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    # basic information (train_no_amp.py:30-42)
    p.add_argument("--user", default="bitgroup_21_concat", type=str)
    p.add_argument("--experiment", default="clswiseformer_21_concat", type=str)
    p.add_argument("--date", default=local_time.split(" ")[0], type=str)
    p.add_argument("--description", default="cls_wise,training on train.txt!", type=str)
    p.add_argument("--project_root", default="cls_wise_concat", type=str)
    # dataset information (:45-73)
    p.add_argument("--root", default="dataset", type=str)
    p.add_argument("--train_dir", default="MICCAI_BraTS_2018_Data_Training", type=str)
    p.add_argument("--valid_dir", default="Valid", type=str)
    p.add_argument("--mode", default="train", type=str)
    p.add_argument("--train_file", default="train.txt", type=str)
    p.add_argument("--valid_file", default="valid.txt", type=str)
    p.add_argument("--dataset", default="brats", type=str)
    p.add_argument("--input_C", default=4, type=int)
    p.add_argument("--input_H", default=240, type=int)
    p.add_argument("--input_W", default=240, type=int)
    p.add_argument("--input_D", default=160, type=int)
    p.add_argument("--crop_H", default=128, type=int)
    p.add_argument("--crop_W", default=128, type=int)
    p.add_argument("--crop_D", default=128, type=int)
    p.add_argument("--output_D", default=155, type=int)
    # training information (:76-105)
    p.add_argument("--lr", default=0.0002, type=float)
    p.add_argument("--weight_decay", default=1e-5, type=float)
    p.add_argument("--amsgrad", default=True, type=_bool)
    p.add_argument("--criterion", default="softmax_dice", type=str)
    p.add_argument("--num_class", default=4, type=int)
    p.add_argument("--seed", default=1000, type=int)
    p.add_argument("--no_cuda", default=False, type=_bool)
    p.add_argument("--gpu", default="0,1,2,3", type=str)
    p.add_argument("--num_workers", default=8, type=int)
    p.add_argument("--batch_size", default=1, type=int)
    p.add_argument("--start_epoch", default=0, type=int)
    p.add_argument("--end_epoch", default=1000, type=int)
    p.add_argument("--save_freq", default=50, type=int)
    p.add_argument("--resume", default="", type=str)
    p.add_argument("--load", default=True, type=_bool)
    p.add_argument("--local_rank", default=int(os.environ.get("LOCAL_RANK", 0)), type=int)
    p.add_argument("--print_rank", default=0, type=int)
    # additions of this implementation
    p.add_argument("--synthetic", default=0, type=int, help="train on N generator-defined subjects instead of files under --root")
    p.add_argument("--precision", default="bf16x3", choices=("fp32", "bf16x3", "bf16"))
    p.add_argument("--step_mode", default="eager", choices=("eager", "plan", "hipgraph"),
                   help="plan: the step (fixed crop size and batch) is captured once and re-issued by the library as a launch list "
                        "(cwf.trainer: ~2 ms of host time per step instead of ~8-14); the last, smaller batch of an epoch would not fit "
                        "the capture, so it is dropped in this mode")
    p.add_argument("--log_every", default=10, type=int)
    p.add_argument("--max_iters", default=0, type=int, help="stop after this many iterations (smoke runs); 0 = no limit")
    p.add_argument("--backend", default=None, type=str, help="torch.distributed backend (default: nccl = RCCL on GPU, gloo on CPU)")
    return p


def should_save(epoch, end_epoch, save_freq):
    """train_no_amp.py:243-246 (epochs are 0-based in the loop, 1-based in the test)."""
    e = epoch + 1
    conds = [e % int(save_freq) == 0]
    for back in (1, 2, 3):
        if end_epoch - back > 0:
            conds.append(e % int(end_epoch - back) == 0)
    return any(conds)


def make_dataset(args):
    from utils import data
    crop = (args.crop_H, args.crop_W, args.crop_D)
    if args.synthetic > 0:
        return data.SyntheticBraTS(args.synthetic, crop, args.seed)
    root = os.path.join(args.root, args.train_dir)
    lst = os.path.join(root, args.train_file)
    return data.NpzBraTS(root, lst if os.path.isfile(lst) else None, crop, args.seed, train=(args.mode == "train"))


def main(argv=None):
    args = build_parser().parse_args(argv)
    from cwf import kernels
    from cwf.parallel import shard_indices
    from cwf.trainer import Trainer, load_checkpoint, save_checkpoint
    from models.clswiseformer.cls_wise_former import get_cls_wise_former

    use_cuda = torch.cuda.is_available() and not args.no_cuda
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend or ("nccl" if use_cuda else "gloo"))
    device = torch.device("cuda", args.local_rank) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    is_printer = rank == args.print_rank

    torch.manual_seed(args.seed); random.seed(args.seed); np.random.seed(args.seed)        # :123-126
    if use_cuda:
        torch.cuda.manual_seed(args.seed)
        kernels.set_precision(args.precision)

    log = logging.getLogger("cwf.train")
    if is_printer and not log.handlers:
        log.setLevel(logging.INFO)
        fmt = logging.Formatter("%(asctime)s ===> %(message)s", datefmt="%Y-%m-%d %H:%M:%S")
        log_dir = os.path.join(args.project_root, "log")
        os.makedirs(log_dir, exist_ok=True)
        for h in (logging.StreamHandler(), logging.FileHandler(os.path.join(log_dir, args.experiment + args.date + ".txt"))):
            h.setFormatter(fmt); log.addHandler(h)
        for k, v in sorted(vars(args).items()):
            log.info("%s=%s", k, v)

    model = get_cls_wise_former(dataset=args.dataset, _conv_repr=True, _pe_type="fixed", gpu=args.local_rank).to(device)
    model.train()
    if os.path.isfile(args.resume) and args.load:
        load_checkpoint(args.resume, model)
        if is_printer:
            log.info("loaded checkpoint %s, training from epoch %d", args.resume, args.start_epoch)
    elif is_printer:
        log.info("re-training!!!")
    trainer = Trainer(model, lr=args.lr, weight_decay=args.weight_decay, amsgrad=args.amsgrad, end_epoch=args.end_epoch,
                      use_graph={"eager": False, "plan": "plan", "hipgraph": "hipgraph"}[args.step_mode] if use_cuda else False)

    ckpt_dir = os.path.join(args.project_root, "checkpoint", args.experiment + args.date)
    if is_printer:
        os.makedirs(ckpt_dir, exist_ok=True)
    ds = make_dataset(args)
    t_start, iters, pending = time.time(), 0, None
    done = False
    for epoch in range(args.start_epoch, args.end_epoch):
        if hasattr(ds, "set_epoch"):
            ds.set_epoch(epoch)
        mine = shard_indices(len(ds), rank, world, epoch=epoch, shuffle=True, seed=args.seed)      # DistributedSampler semantics
        loader = torch.utils.data.DataLoader(torch.utils.data.Subset(ds, mine), batch_size=args.batch_size, shuffle=False,
                                             drop_last=(args.step_mode != "eager" and use_cuda), num_workers=args.num_workers if use_cuda else 0, pin_memory=use_cuda)
        for i, (x, target, edge, _missing) in enumerate(loader):
            x, target, edge = (t.to(device, non_blocking=True) for t in (x, target, edge))
            loss, parts = trainer.step(x, target, edge, epoch)
            iters += 1
            if is_printer and (iters % max(args.log_every, 1) == 0 or iters == 1):
                if pending is not None:                      # print the PREVIOUS snapshot: its copy has long finished
                    ev, host, tag = pending
                    ev.synchronize() if ev is not None else None
                    v = host.tolist()
                    log.info("Epoch: %d_Iter:%d  loss: %.5f || end_loss: %.5f || s_loss:%.4f || edge_loss:%.4f || mid_s_loss:%.4f || mid_edge_loss:%.4f ||",
                             tag[0], tag[1], v[0], v[1], v[2], v[3], v[4], v[5])
                snap = torch.stack([loss] + list(parts)).float()
                if use_cuda:
                    host = torch.empty(6, dtype=torch.float32).pin_memory()
                    host.copy_(snap, non_blocking=True)
                    ev = torch.cuda.Event(); ev.record()
                else:
                    host, ev = snap.clone(), None
                pending = (ev, host, (epoch, i))
            if args.max_iters and iters >= args.max_iters:
                done = True
                break
        if is_printer and should_save(epoch, args.end_epoch, args.save_freq):
            save_checkpoint(os.path.join(ckpt_dir, "model_epoch_%d.pth" % epoch), model, trainer.opt, epoch)
        if done:
            break
    if is_printer:
        if pending is not None:
            ev, host, tag = pending
            ev.synchronize() if ev is not None else None
            log.info("Epoch: %d_Iter:%d  loss: %.5f (last logged)", tag[0], tag[1], host.tolist()[0])
        save_checkpoint(os.path.join(ckpt_dir, "model_epoch_last.pth"), model, trainer.opt, args.end_epoch)       # :255-262
        log.info("The total training time is %.2f hours", (time.time() - t_start) / 3600.0)
        log.info("----------------------------------The training process finished!-----------------------------------")
    if world > 1 and dist.is_initialized():
        dist.barrier()
    return 0


if __name__ == "__main__":
    sys.exit(main())
