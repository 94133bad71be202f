"""Training entry point — the flags of the reference's ``lesions3d/train.py:27-64`` and the flow of its
``example()`` (``train.py:128-188``) without Lightning / wandb / MONAI:

    python -m mslesions3d_amd.train -d DATA -dn NAME -b 2 -me 2 -ld LOGS

dataset -> LSSD3D(n_classes + 1, input_channels=1, ...) -> loop {training_step-equivalent fused step, validation at
epoch end} -> metrics as JSONL with the reference's scalar names -> top-3 checkpoints by ``avg_val_loss`` ->
early stopping on the validation loss (patience 5) -> stop at ``max_iterations`` / ``max_epochs``.
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


def example(args):
    from .datasets import ExampleDataset, select_augmentations
    from .ssd3d import LSSD3D
    from .trainer import FusedTrainer
    torch.manual_seed(args.seed)
    layers = [int(x) for x in args.prediction_layers.split()]
    aspect_ratios = {l: [1.] for l in layers}
    scales = {int(k): v for k, v in args.scales.items()}
    augmentations = select_augmentations(args.augmentations)  # train.py:132-145 (flip rotate90 translate scale)
    dataset = ExampleDataset(n_classes=args.n_classes, subject=args.subject, percentage=args.percentage,
                             num_workers=args.num_workers, batch_size=args.batch_size, data_dir=args.dataset_path,
                             dataset_name=args.dataset_name, augmentations=augmentations, random_state=970205)
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
    model = model.to("cuda")
    model.compute_dtype = args.dtype
    trainer = FusedTrainer(model)
    first_epoch = 0
    if args.checkpoint:
        # resume_from_checkpoint (train.py:185): weights above, here the optimiser moments / step count, the scheduler
        # phase and the epoch counter, so that the resumed run continues the interrupted one bit for bit
        ckpt = LSSD3D.read_checkpoint(args.checkpoint)
        if ckpt.get("optimizer_states"):
            trainer.load_state_dict({"optimizer": ckpt["optimizer_states"][0],
                                     "scheduler": (ckpt.get("lr_schedulers") or [None])[0]})
            first_epoch = int(ckpt.get("epoch", -1)) + 1
    logdir = pjoin(args.logdir, args.experiment_name)
    os.makedirs(logdir, exist_ok=True)
    log = open(pjoin(logdir, "metrics.jsonl"), "a")
    best, bad_epochs, done = [], 0, False
    max_epochs = args.max_epochs if args.max_epochs else 10 ** 9
    max_iters = -1 if args.max_epochs else args.max_iterations
    for epoch in range(first_epoch, max_epochs):
        model.current_epoch = epoch
        model.train()
        for batch in dataset.train_dataloader():
            out = trainer.step(batch["img"].to("cuda"), batch["boxes"], batch["labels"])
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
        avg = {k: float(torch.stack([v["log"][k] for v in vals]).mean()) for k in ("val_total_loss", "val_conf_loss", "val_loc_loss")}
        rec = {"step": model.global_step, "epoch": epoch, "avg_val_loss": avg["val_total_loss"],
               "total_loss/validation": avg["val_total_loss"], "confidence_loss/validation": avg["val_conf_loss"],
               "localization_loss/validation": avg["val_loc_loss"]}
        if vals and "metrics_50" in vals[0]["log"]:
            for tag, key in (("0.1", "metrics_10"), ("0.5", "metrics_50")):
                for m in ("mAP", "precision", "recall", "f1_score"):
                    rec[f"{m}/validation_IoU_{tag}"] = float(sum(float(v["log"][key][m]) for v in vals) / len(vals))
        log.write(json.dumps(rec) + "\n")
        log.flush()
        print(rec)
        # ModelCheckpoint(monitor="avg_val_loss", save_top_k=3, mode="min")  (train.py:171-176)
        path = pjoin(logdir, f"checkpoint-epoch={epoch:03d}-avg_val_loss={rec['avg_val_loss']:.4f}.ckpt")
        best.append((rec["avg_val_loss"], path))
        best.sort()
        if (rec["avg_val_loss"], path) in best[:3]:
            model.save_checkpoint(path, trainer)
        for _, p in best[3:]:
            if os.path.exists(p):
                os.remove(p)
        best = best[:3]
        # EarlyStopping('total_loss/validation', patience=5)  (train.py:180)
        bad_epochs = 0 if rec["avg_val_loss"] <= best[0][0] else bad_epochs + 1
        if done or (args.early_stopping and bad_epochs >= 5):
            break
    log.close()
    return model


if __name__ == "__main__":
    example(build_parser().parse_args())
