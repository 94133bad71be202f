"""Prediction entry point — flags of the reference's ``lesions3d/predict.py:29-44`` and the flow of
``predict_example`` (``predict.py:235-281``): load checkpoint -> predict every subject -> write
``sub-XXXX_preds.json`` (``{j+1: [box_frac(6), box_voxel(6), label, score]}``, ``predict.py:149,222-232``) and
``sub-XXXX_preds.csv`` (``label_id,score``) -> per-subject mAP at IoU 0.5 and 0.1 (``predict.py:87-152``).
NIfTI overlays are not written (nibabel absent; out of scope).

    python -m mslesions3d_amd.predict -d DATA -dn NAME -m CKPT -o OUT
"""
import argparse
import json
import collections
import os
from os.path import join as pjoin

import numpy as np
import torch


def build_parser():
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('-d', '--dataset_path', type=str, default=r'../data/artificial_dataset')
    p.add_argument('-dn', '--dataset_name', type=str, default=None)
    p.add_argument('-m', '--model_path', type=str, default=r'model_final.ckpt')
    p.add_argument('-p', '--percentage', type=float, default=1.)
    p.add_argument('-su', '--subject', type=str, default=None)
    p.add_argument('-c', '--n_classes', type=int, default=1)
    p.add_argument('-nw', '--num_workers', type=int, default=0)
    p.add_argument('-ps', '--predict_subset', type=str, choices=['train', 'validation', 'test', 'all'], default='train')
    p.add_argument('-sc', '--min_score', type=float, default=0.5)
    p.add_argument('-k', '--top_k', type=int, default=100)
    p.add_argument('-o', '--output_dir', type=str, default=r"../data/predictions/")
    # not in the reference's CLI: bf16 activations for the eval forward (fp32 is the reference's precision)
    p.add_argument('--dtype', type=str, choices=['f32', 'bf16'], default='f32')
    return p


def save_predictions(subject, img_shape, boxes, labels, scores, min_score, output_dir):
    """predict.py:155-232 without the NIfTI overlay; file contents byte-identical to the reference's on the same detections
    (tests/golden/preds/*): the voxel box is the fp32 product ``clip(box, 0, 1) * shape`` truncated to int, and the CSV's
    score column holds what pandas prints for the 0-dim tensors the reference stores there (``tensor(0.9700)``)."""
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 6)
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    shape2 = np.asarray(tuple(img_shape) * 2, dtype=np.float32)
    infos, scores_map = {}, []
    for j in range(boxes.shape[0]):
        score = float(scores[j])
        scores_map.append((j + 1, scores[j]))
        if score < min_score or int(labels[j]) == 0:
            continue
        frac = [float(v) for v in boxes[j]]
        vox = (np.clip(boxes[j], np.float32(0), np.float32(1)) * shape2).astype(int).tolist()
        infos[j + 1] = (frac, vox, int(labels[j]), score)
    with open(pjoin(output_dir, f"sub-{subject}_preds.json"), "w") as f:
        json.dump(infos, f)
    with open(pjoin(output_dir, f"sub-{subject}_preds.csv"), "w") as f:
        f.write(",label_id,score\n")
        for i, (lid, sc) in enumerate(scores_map):
            f.write(f"{i},{lid},{str(torch.tensor(sc))}\n")


def gather_detections(records, world, rank):
    """SURVEY section 8(e), inference: replicas only - every rank predicts its share of the subjects, no collective on the
    data path; the per-subject detections (a few KB of plain lists each) are gathered on rank 0 for the metrics.
    ``records``: list of (position in the data set, subject, record dict).  Returns the merged list in data-set order on rank 0,
    None elsewhere."""
    if world == 1:
        return sorted(records, key=lambda r: r[0])
    import torch.distributed as dist
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(records, bucket, dst=0)
    if rank != 0:
        return None
    seen, merged = set(), []
    for part in bucket:
        for rec in part:
            if rec[0] not in seen:  # (the wrap-around padding of the shards predicts a few subjects twice)
                seen.add(rec[0])
                merged.append(rec)
    return sorted(merged, key=lambda r: r[0])


def predict_example(args):
    """predict.py:235-281.  Under ``python -m torch.distributed.run --nproc-per-node N -m mslesions3d_amd.predict ...`` the
    subjects are dealt round-robin over N replicas (one GPU each) and rank 0 writes the files and the metrics."""
    from .datasets import ExampleDataset, ShardSampler
    from .ssd3d import LSSD3D
    from .utils import calculate_mAP
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if world > 1:
        from .parallel import init_distributed
        backend = os.environ.get("MSL_DP_BACKEND", "nccl")
        if backend != "nccl":
            local = local % max(torch.cuda.device_count(), 1)
        init_distributed(backend, rank=rank, world_size=world, device=torch.device("cuda", local) if backend == "nccl" else None)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
    dataset = ExampleDataset(n_classes=args.n_classes, batch_size=1, num_workers=args.num_workers, subject=args.subject,
                             percentage=args.percentage, data_dir=args.dataset_path, dataset_name=args.dataset_name)
    dataset.setup(stage="predict_train" if args.predict_subset == "train" else "predict")
    model = LSSD3D.load_from_checkpoint(args.model_path, min_score=args.min_score).to(dev).eval()
    model.top_k, model.min_score = args.top_k, args.min_score  # predict.py:259-260
    model.compute_dtype = getattr(args, "dtype", "f32")
    ds = dataset.predict_dataset
    mine = ShardSampler(len(ds), rank, world, shuffle=False).indices().tolist()
    records = []
    queued = collections.deque()  # (position, batch) of the passes in flight: predict_batches yields results in order

    def feed():
        for pos in mine:
            batch = collate([ds[pos]])
            queued.append((pos, batch))
            yield batch

    for boxes, labels, scores in model.predict_batches(feed(), depth=2):
        pos, batch = queued.popleft()
        records.append((pos, batch["subject"][0],
                        {"shape": tuple(batch["img"].shape[2:]), "boxes": boxes[0].cpu().numpy().tolist(),
                         "labels": labels[0].cpu().numpy().tolist(), "scores": scores[0].cpu().numpy().tolist(),
                         "gt_boxes": batch["boxes"][0].numpy().tolist(), "gt_labels": batch["labels"][0].numpy().tolist()}))
    merged = gather_detections(records, world, rank)
    metrics = {"0.5": {}, "0.1": {}}
    if merged is not None:
        for _, subj, r in merged:
            save_predictions(subj, r["shape"], np.asarray(r["boxes"], np.float32), np.asarray(r["labels"]),
                             np.asarray(r["scores"], np.float32), args.min_score, args.output_dir)
            det_b = [torch.tensor(r["boxes"], dtype=torch.float32).reshape(-1, 6)]
            det_l = [torch.tensor(r["labels"], dtype=torch.long)]
            det_s = [torch.tensor(r["scores"], dtype=torch.float32)]
            gt_b = [torch.tensor(r["gt_boxes"], dtype=torch.float32).reshape(-1, 6)]
            gt_l = [torch.tensor(r["gt_labels"], dtype=torch.long)]
            dif = [torch.zeros(len(l), dtype=torch.bool) for l in gt_l]
            for iou in (0.5, 0.1):
                d = calculate_mAP(det_b, det_l, det_s, gt_b, gt_l, dif, min_overlap=iou, return_detail=True)
                metrics[str(iou)][subj] = {k: float(d[k]) for k in ("mAP", "precision", "recall", "f1_score")}
        for iou, m in metrics.items():
            with open(pjoin(args.output_dir, f"aa_metrics_per_subject_(min_IoU={iou}).json"), "w") as f:
                json.dump(m, f, indent=4)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return metrics


def collate(samples):
    from .datasets import collate_fn
    return collate_fn(samples)


if __name__ == "__main__":
    predict_example(build_parser().parse_args())
