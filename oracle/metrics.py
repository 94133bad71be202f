"""Oracle detection metrics (test infrastructure; see oracle/__init__.py).

Restates reference ``lesions3d/utils.py:157-396`` (``compute_metrics_per_class``, ``calculate_mAP``)
in numpy.  Two classes only (background + lesion): the reference hard-codes
``n_classes = len(label_map) = 2`` (utils.py:27-29, :260).
"""
import numpy as np

from . import boxes as B
import torch


def _iou_1_to_many(box, others):
    return B.iou_matrix(torch.from_numpy(box[None]), torch.from_numpy(others))[0].numpy()


def class_metrics(det_img, det_boxes, det_scores, true_img, true_boxes, true_difficult, min_overlap):
    """utils.py:157-239.  Detections in stable descending score order; each is matched to the
    max-IoU (first max) ground truth of its image; TP iff IoU > min_overlap (strict), GT easy and not
    yet claimed; a match to a 'difficult' GT counts as neither TP nor FP."""
    order = np.lexsort((np.arange(det_scores.shape[0]), -det_scores.astype(np.float64)))
    det_img, det_boxes, det_scores = det_img[order], det_boxes[order], det_scores[order]
    claimed = np.zeros(true_boxes.shape[0], dtype=np.uint8)
    tp = np.zeros(det_boxes.shape[0], dtype=np.float32)
    fp = np.zeros(det_boxes.shape[0], dtype=np.float32)
    for d in range(det_boxes.shape[0]):
        same = np.nonzero(true_img == det_img[d])[0]
        if same.size == 0:
            fp[d] = 1
            continue
        ov = _iou_1_to_many(det_boxes[d], true_boxes[same])
        mx = ov.max()
        j = same[int(np.nonzero(ov == mx)[0][0])] if not np.isnan(mx) else same[int(np.argmax(ov))]
        if mx > min_overlap:
            if not true_difficult[j]:
                if claimed[j] == 0:
                    tp[d] = 1
                    claimed[j] = 1
                else:
                    fp[d] = 1
        else:
            fp[d] = 1
    vols = np.array([(b[3] - b[0]) * (b[4] - b[1]) * (b[5] - b[2]) for b, dif in zip(true_boxes, true_difficult) if not dif],
                    dtype=np.float32)
    return tp, fp, claimed, det_scores, vols[claimed == 1], vols[claimed == 0]


def calculate_map(det_boxes, det_labels, det_scores, true_boxes, true_labels, true_difficulties, min_overlap=0.5):
    """utils.py:242-396 with ``return_detail=True``; inputs are lists (one entry per image) of numpy
    arrays.  Returns the detail dict with numpy / float values."""
    n_img = len(det_boxes)
    assert n_img == len(det_labels) == len(det_scores) == len(true_boxes) == len(true_labels) == len(true_difficulties)
    t_img = np.concatenate([np.full(len(true_labels[i]), i, dtype=np.int64) for i in range(n_img)])
    t_box = np.concatenate(true_boxes).astype(np.float32).reshape(-1, 6)
    t_lab = np.concatenate(true_labels)
    t_dif = np.concatenate(true_difficulties).astype(bool)
    d_img = np.concatenate([np.full(len(det_labels[i]), i, dtype=np.int64) for i in range(n_img)])
    d_box = np.concatenate(det_boxes).astype(np.float32).reshape(-1, 6)
    d_lab = np.concatenate(det_labels)
    d_sco = np.concatenate(det_scores).astype(np.float32)

    c = 1
    tsel, dsel = t_lab == c, d_lab == c
    n_easy = int((~t_dif[tsel]).sum())
    if dsel.sum() == 0:  # utils.py:308-309 + :370-380 (KeyError branch)
        vols = np.array([(b[3] - b[0]) * (b[4] - b[1]) * (b[5] - b[2]) for b in t_box], dtype=np.float32)
        return {"APs": 0.0, "mAP": 0.0, "precision": 0.0, "recall": 0.0, "f1_score": 0.0, "sorted_det_scores": {},
                "TP": np.zeros(0, np.float32), "FP": np.zeros(0, np.float32), "n_true_boxes": n_easy,
                "found_boxes_volumes_per_class": np.zeros(0, np.float32), "not_found_boxes_volumes_per_class": vols}
    tp, fp, claimed, sorted_scores, found, not_found = class_metrics(
        d_img[dsel], d_box[dsel], d_sco[dsel], t_img[tsel], t_box[tsel], t_dif[tsel], min_overlap)
    fn = 1 - claimed.astype(np.float32)
    tps = np.float32(tp.sum())
    recall = tps / (tps + np.float32(fn.sum()))  # utils.py:324
    precision = tps / (tps + np.float32(fp.sum()))  # utils.py:325
    with np.errstate(invalid="ignore", divide="ignore"):
        f1 = (2 * precision * recall) / (precision + recall)  # utils.py:326
    ctp, cfp = np.cumsum(tp, dtype=np.float32), np.cumsum(fp, dtype=np.float32)
    cprec = ctp / (ctp + cfp + np.float32(1e-10))
    with np.errstate(invalid="ignore", divide="ignore"):
        crec = ctp / np.float32(n_easy)
    thr = torch.arange(start=0, end=1.1, step=.1).tolist()  # utils.py:335 (same float values)
    prec = np.zeros(len(thr), dtype=np.float32)
    for i, t in enumerate(thr):
        m = crec >= t
        prec[i] = cprec[m].max() if m.any() else 0.0
    ap = float(prec.mean(dtype=np.float32))
    return {"APs": ap, "mAP": ap, "precision": float(precision), "recall": float(recall), "f1_score": float(f1),
            "sorted_det_scores": {1: sorted_scores}, "TP": tp, "FP": fp, "n_true_boxes": int(claimed.shape[0]),
            "found_boxes_volumes_per_class": found, "not_found_boxes_volumes_per_class": not_found}
