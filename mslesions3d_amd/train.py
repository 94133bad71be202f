"""Training entry point — the flags of the reference's ``lesions3d/train.py:27-64`` and the flow of its
``example()`` (``train.py:128-188``) without Lightning / wandb / MONAI:

    python -m mslesions3d_amd.train -d DATA -dn NAME -b 2 -me 2 -ld LOGS

dataset -> LSSD3D(n_classes + 1, input_channels=1, ...) -> loop {training_step-equivalent fused step, validation at
epoch end} -> metrics as JSONL with the reference's scalar names -> top-3 checkpoints by ``avg_val_loss`` ->
early stopping on the validation loss (patience 5) -> stop at ``max_iterations`` / ``max_epochs``.

Data parallel (BASELINE north_star; the reference is single-GPU, train.py:182 ``devices=1``):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m mslesions3d_amd.train ...
"""
import argparse
import json
import os
from os.path import join as pjoin

import torch


def build_parser():
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('-d', '--dataset_path', type=str, default=r'../data/artificial_dataset')
    p.add_argument('-dn', '--dataset_name', type=str, default="#3k_64_n1-5_s6-14")
    p.add_argument('-su', '--subject', type=str, default=None)
    p.add_argument('-p', '--percentage', type=float, default=1.)
    p.add_argument('--n_classes', type=int, default=1)
    p.add_argument('-b', '--batch_size', type=int, default=8)
    p.add_argument('-lr', '--learning_rate', type=float, default=0.001)
    p.add_argument('-sr', '--scheduler', type=str, default="CosineAnnealingLR")
    p.add_argument('-th', '--threshold', type=float, default=[0.1, 0.2], nargs='+')
    p.add_argument('-pl', '--prediction_layers', type=str, default="3 5 7")
    p.add_argument('-cfg', '--base_network_config', type=str, default="mobilenet")
    p.add_argument('-sc', '--scales', type=json.loads, default="{}")
    p.add_argument('-bpl', '--boxes_per_location', type=int, default=2)
    p.add_argument('-minos', '--min_object_size', type=int, default=6)
    p.add_argument('-maxos', '--max_object_size', type=int, default=14)
    p.add_argument('--alpha', type=float, default=1.)
    p.add_argument('-a', '--augmentations', type=str, nargs='*', default=[])
    p.add_argument('-ld', '--logdir', type=str, default=r'../logs/artificial_dataset')
    p.add_argument('-nw', '--num_workers', type=int, default=0)
    p.add_argument('-wm', '--width_mult', type=float, default=1.)
    p.add_argument('-en', '--experiment_name', type=str, default="multiple_subjects_64")
    p.add_argument('-me', '--max_epochs', type=int, default=None)
    p.add_argument('-mi', '--max_iterations', type=int, default=4000)
    p.add_argument('-cp', '--checkpoint', type=str, default=None)
    p.add_argument('-rs', '--seed', type=int, default=970205)
    p.add_argument('-es', '--early_stopping', type=int, default=1)
    p.add_argument('-cm', '--compute_metric_every_n_epochs', type=int, default=1)
    p.add_argument('-coms', '--comments', type=str, default="")
    # not in the reference's CLI: the loss variants it keeps as commented code (MultiBoxLoss docstring); default off
    p.add_argument('--hard_negative_mining', action='store_true')
    p.add_argument('--smooth_l1', action='store_true')
    p.add_argument('--focal_loss', action='store_true')
    # build-side extension (the reference trains in fp32): activations and activation gradients stored as bf16
    p.add_argument('--dtype', choices=["f32", "bf16"], default="f32")
    return p


def _dist_env():
    """(world, rank, local rank) of a `python -m torch.distributed.run` launch; (1, 0, 0) for a plain run."""
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def dist_barrier():
    import torch.distributed as dist
    dist.barrier()


def _mean_over_ranks(sums, count, device, on):
    """[sum of per-batch values ...] / number of batches, over every rank's validation shard: one all-reduce, so every
    rank holds bit-identical averages (early stopping and checkpoint ranking then need no broadcast)."""
    import torch.distributed as dist
    t = torch.tensor(list(sums) + [float(count)], dtype=torch.float64, device=device)
    if on:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    n = max(float(t[-1].item()), 1.0)
    return [float(v) / n for v in t[:-1].tolist()]


def example(args):
    """The reference's ``example()`` (train.py:128-188) as a data-parallel loop: under
    ``python -m torch.distributed.run --nproc-per-node N -m mslesions3d_amd.train ...`` every rank owns one GPU, trains on
    its shard of the cases (``datasets.ShardSampler``; BatchNorm statistics and the loss normaliser per replica), exchanges
    gradients through ``FusedTrainer``'s bucketed RCCL all-reduce overlapped with the backward pass, validates its shard of
    the validation cases, and rank 0 writes metrics and checkpoints.  N = 1 is the reference's single-GPU run."""
    from .datasets import ExampleDataset, select_augmentations
    from .ssd3d import LSSD3D
    from .trainer import FusedTrainer
    world, rank, local = _dist_env()
    dp = world > 1
    if dp:  # join the job BEFORE the first GPU call of this process (RCCL binds the communicator to the device)
        from .parallel import broadcast_model, init_distributed
        backend = os.environ.get("MSL_DP_BACKEND", "nccl")  # gloo: functional rehearsal with ranks sharing a GPU
        if backend != "nccl":
            local = local % max(torch.cuda.device_count(), 1)
        init_distributed(backend, rank=rank, world_size=world, device=torch.device("cuda", local) if backend == "nccl" else None)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(args.seed)  # same seed on every rank: identical initial weights even before the broadcast
    layers = [int(x) for x in args.prediction_layers.split()]
    aspect_ratios = {l: [1.] for l in layers}
    scales = {int(k): v for k, v in args.scales.items()}
    augmentations = select_augmentations(args.augmentations)  # train.py:132-145 (flip rotate90 translate scale)
    dataset = ExampleDataset(n_classes=args.n_classes, subject=args.subject, percentage=args.percentage,
                             num_workers=args.num_workers, batch_size=args.batch_size, data_dir=args.dataset_path,
                             dataset_name=args.dataset_name, augmentations=augmentations, random_state=970205,
                             rank=rank, world_size=world)
    dataset.setup(stage="fit")
    input_size = tuple(dataset.train_dataset[0]["img"].shape)[1:]
    threshold = args.threshold if len(args.threshold) > 1 else [args.threshold[0]]
    if args.checkpoint:
        model = LSSD3D.load_from_checkpoint(args.checkpoint)
    else:
        model = LSSD3D(n_classes=args.n_classes + 1, input_channels=1, lr=args.learning_rate, width_mult=args.width_mult,
                       scheduler=args.scheduler, batch_size=args.batch_size, comments=args.comments, input_size=input_size,
                       compute_metric_every_n_epochs=args.compute_metric_every_n_epochs, use_wandb=False,
                       aspect_ratios=aspect_ratios, scales=scales, alpha=args.alpha, threshold=threshold,
                       min_object_size=args.min_object_size, max_object_size=args.max_object_size,
                       base_network_config=args.base_network_config, boxes_per_location=args.boxes_per_location,
                       hard_negative_mining=args.hard_negative_mining, smooth_l1=args.smooth_l1,
                       focal_loss=args.focal_loss)
    model.init()
    model = model.to(dev)
    model.compute_dtype = args.dtype
    if dp:  # rank 0's weights and BatchNorm buffers everywhere (one flat broadcast of the parameter arena)
        model._ensure_device_state(dev)
        model._engine.ensure_arena(dev)
        broadcast_model(model)
    trainer = FusedTrainer(model)
    first_epoch = 0
    best, bad_epochs = [], 0
    if args.checkpoint:
        # resume_from_checkpoint (train.py:185): weights above; here the optimiser moments / step count, the scheduler phase,
        # the epoch counter and the loop's own bookkeeping (top-3 list, early-stopping counter).  Shuffle order and
        # augmentation draws are functions of (seed, epoch, subject), so the resumed run sees what the interrupted one would
        # have seen from epoch `epoch + 1` on.  A run continues exactly only from a `last.ckpt` (written every epoch); a
        # top-3 checkpoint restarts from ITS epoch.
        ckpt = LSSD3D.read_checkpoint(args.checkpoint)
        if ckpt.get("optimizer_states"):
            trainer.load_state_dict({"optimizer": ckpt["optimizer_states"][0],
                                     "scheduler": (ckpt.get("lr_schedulers") or [None])[0]})
            first_epoch = int(ckpt.get("epoch", -1)) + 1
        loop = ckpt.get("loop_state") or {}
        best = [(float(v), str(p)) for v, p in loop.get("best", [])]
        bad_epochs = int(loop.get("bad_epochs", 0))
    logdir = pjoin(args.logdir, args.experiment_name)
    log = shard_log = None
    if rank == 0:
        os.makedirs(logdir, exist_ok=True)
        log = open(pjoin(logdir, "metrics.jsonl"), "a")
    if dp:  # every rank's own shard: which subjects it trained on and what they cost (rank 0's losses are in metrics.jsonl too)
        dist_barrier()
        shard_log = open(pjoin(logdir, f"shard_rank{rank}.jsonl"), "a")
    done = False
    max_epochs = args.max_epochs if args.max_epochs else 10 ** 9
    max_iters = -1 if args.max_epochs else args.max_iterations
    for epoch in range(first_epoch, max_epochs):
        model.current_epoch = epoch
        model.train()
        dataset.set_epoch(epoch)
        for batch in dataset.train_dataloader():
            out = trainer.step(batch["img"].to(dev), batch["boxes"], batch["labels"])
            if shard_log is not None:
                shard_log.write(json.dumps({"step": model.global_step, "epoch": epoch, "subjects": list(batch["subject"]),
                                            "total_loss/training": out["loss"]}) + "\n")
            if log is not None:  # rank 0's shard losses (the reference logs one process's batch)
                log.write(json.dumps({"step": model.global_step, "epoch": epoch, "total_loss/training": out["loss"],
                                      "confidence_loss/training": out["conf"], "localization_loss/training": out["loc"],
                                      "hp_metric/lr": trainer.sch.get_last_lr()[1] if trainer.sch else model.lr}) + "\n")
            if 0 < max_iters <= model.global_step:
                done = True
                break
        if trainer.sch is not None:
            # Lightning steps a scheduler returned by configure_optimizers once per epoch (interval="epoch") on top of
            # the manual per-step call inside training_step: one extra cosine step per epoch
            trainer.sch.step()
        model.eval()
        vals = [model.validation_step(b, i) for i, b in enumerate(dataset.test_dataloader())]
        keys = ("val_total_loss", "val_conf_loss", "val_loc_loss")
        sums = [float(sum(float(v["log"][k]) for v in vals)) for k in keys]
        # (decided from the epoch, not from this rank's batches: every rank must contribute a vector of the same length)
        with_metrics = epoch % max(int(model.compute_metric_every_n_epochs), 1) == 0 and all("metrics_50" in v["log"] for v in vals)
        mkeys = [(tag, key, m) for tag, key in (("0.1", "metrics_10"), ("0.5", "metrics_50"))
                 for m in ("mAP", "precision", "recall", "f1_score")] if with_metrics else []
        sums += [float(sum(float(v["log"][key][m]) for v in vals)) for _, key, m in mkeys]
        avg = _mean_over_ranks(sums, len(vals), dev, dp)
        rec = {"step": model.global_step, "epoch": epoch, "avg_val_loss": avg[0],
               "total_loss/validation": avg[0], "confidence_loss/validation": avg[1],
               "localization_loss/validation": avg[2]}
        for (tag, _, m), v in zip(mkeys, avg[3:]):
            rec[f"{m}/validation_IoU_{tag}"] = v
        # ModelCheckpoint(monitor="avg_val_loss", save_top_k=3, mode="min")  (train.py:171-176); every rank ranks the same
        # all-reduced value, rank 0 touches the files
        path = pjoin(logdir, f"checkpoint-epoch={epoch:03d}-avg_val_loss={rec['avg_val_loss']:.4f}.ckpt")
        best.append((rec["avg_val_loss"], path))
        best.sort()
        keep, drop = best[:3], best[3:]
        # EarlyStopping('total_loss/validation', patience=5)  (train.py:180)
        bad_epochs = 0 if rec["avg_val_loss"] <= keep[0][0] else bad_epochs + 1
        loop_state = {"best": [[v, p] for v, p in keep], "bad_epochs": bad_epochs}
        if rank == 0:
            log.write(json.dumps(rec) + "\n")
            log.flush()
            print(rec)
            if (rec["avg_val_loss"], path) in keep:
                model.save_checkpoint(path, trainer, loop_state=loop_state)
            model.save_checkpoint(pjoin(logdir, "last.ckpt"), trainer, loop_state=loop_state)  # exact continuation point
            for _, p in drop:
                if os.path.exists(p):
                    os.remove(p)
        best = keep
        if done or (args.early_stopping and bad_epochs >= 5):
            break
    if log is not None:
        log.close()
    if shard_log is not None:
        shard_log.close()
    if dp:
        import torch.distributed as dist
        if os.environ.get("MSL_DP_CHECK_REPLICAS", "1") == "1":  # the data-parallel invariant, once, at the end of the run
            flat = model._engine.arena.flat
            ref = flat.detach().clone()
            dist.broadcast(ref, src=0)
            same = torch.tensor([1.0 if torch.equal(ref, flat) else 0.0], device=dev)
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            if float(same.item()) != 1.0:
                raise RuntimeError("data-parallel replicas diverged: the gradient all-reduce is not reaching every parameter")
        dist.barrier()
        dist.destroy_process_group()
    return model


if __name__ == "__main__":
    example(build_parser().parse_args())
