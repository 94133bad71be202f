"""Prediction entry point — flags of the reference's ``lesions3d/predict.py:29-44`` and the flow of
``predict_example`` (``predict.py:235-281``): load checkpoint -> predict every subject -> write
``sub-XXXX_preds.json`` (``{j+1: [box_frac(6), box_voxel(6), label, score]}``, ``predict.py:149,222-232``) and
``sub-XXXX_preds.csv`` (``label_id,score``) -> per-subject mAP at IoU 0.5 and 0.1 (``predict.py:87-152``).
NIfTI overlays are not written (nibabel absent; out of scope).

    python -m mslesions3d_amd.predict -d DATA -dn NAME -m CKPT -o OUT
"""
import argparse
import json
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


def predict_example(args):
    from .datasets import ExampleDataset
    from .ssd3d import LSSD3D
    from .utils import calculate_mAP
    os.makedirs(args.output_dir, exist_ok=True)
    dataset = ExampleDataset(n_classes=args.n_classes, batch_size=1, num_workers=args.num_workers, subject=args.subject,
                             percentage=args.percentage, data_dir=args.dataset_path, dataset_name=args.dataset_name)
    dataset.setup(stage="predict_train" if args.predict_subset == "train" else "predict")
    model = LSSD3D.load_from_checkpoint(args.model_path, min_score=args.min_score).to("cuda").eval()
    model.top_k, model.min_score = args.top_k, args.min_score  # predict.py:259-260
    model.compute_dtype = getattr(args, "dtype", "f32")
    metrics = {"0.5": {}, "0.1": {}}
    for batch in dataset.predict_dataloader():
        boxes, labels, scores = model.predict_step(batch, 0)
        subj = batch["subject"][0]
        shape = tuple(batch["img"].shape[2:])
        save_predictions(subj, shape, boxes[0].cpu().numpy(), labels[0].cpu().numpy(), scores[0].cpu().numpy(),
                         args.min_score, args.output_dir)
        dif = [torch.zeros(len(l), dtype=torch.bool) for l in batch["labels"]]
        for iou in (0.5, 0.1):
            d = calculate_mAP(boxes, labels, scores, batch["boxes"], batch["labels"], dif, min_overlap=iou, return_detail=True)
            metrics[str(iou)][subj] = {k: float(d[k]) for k in ("mAP", "precision", "recall", "f1_score")}
    for iou, m in metrics.items():
        with open(pjoin(args.output_dir, f"aa_metrics_per_subject_(min_IoU={iou}).json"), "w") as f:
            json.dump(m, f, indent=4)
    return metrics


if __name__ == "__main__":
    predict_example(build_parser().parse_args())
