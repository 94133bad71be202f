"""Box utilities and detection metrics — host-side mirror of the reference's ``lesions3d/utils.py:25-396``.

* Box transforms / IoU (``utils.py:42-149``) run as HIP kernels (``csrc/multibox.hip``) on GPU tensors; there is
  no CPU path for them.
* ``calculate_mAP`` (``utils.py:157-396``) is host-side bookkeeping in the reference (a Python loop per
  detection) and stays host-side here: detections are tiny ragged lists, copied to the host once.
"""
import numpy as np
import torch

from . import _lib
from ._lib import ptr

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")

# Label map (utils.py:25-30)
voc_labels = tuple(["lesion"])
label_map = {k: v + 1 for v, k in enumerate(voc_labels)}
label_map['background'] = 0
rev_label_map = {v: k for k, v in label_map.items()}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _gpu(t, what):
    if not t.is_cuda:
        raise _lib.HipKernelError(f"{what}: expected a tensor on the HIP device (no CPU fallback)")
    return t.contiguous().float()


def _transform(boxes, priors, op, what):
    boxes = _gpu(boxes, what)
    if priors is not None:
        priors = _gpu(priors, what)
        assert priors.shape == boxes.shape
    out = torch.empty_like(boxes)
    _lib.call("msl_box_transform", ptr(boxes), ptr(priors), ptr(out), boxes.shape[0], op, _stream())
    return out


def cxcycz_to_xyz(cxcycz):
    """utils.py:42-51."""
    return _transform(cxcycz, None, 0, "cxcycz_to_xyz")


def xyz_to_cxcycz(xy):
    """utils.py:92-102."""
    return _transform(xy, None, 1, "xyz_to_cxcycz")


def cxcycz_to_gcxgcygcz(cxcycz, priors_cxcycz):
    """utils.py:71-89."""
    return _transform(cxcycz, priors_cxcycz, 2, "cxcycz_to_gcxgcygcz")


def gcxgcygcz_to_cxcycz(gcxgcygcz, priors_cxcycz):
    """utils.py:54-68."""
    return _transform(gcxgcygcz, priors_cxcycz, 3, "gcxgcygcz_to_cxcycz")


def _pairwise(set_1, set_2, inter_only, what):
    a, b = _gpu(set_1, what), _gpu(set_2, what)
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    _lib.call("msl_iou_matrix", ptr(a), ptr(b), ptr(out), a.shape[0], b.shape[0], inter_only, _stream())
    return out


def find_intersection3d(set_1, set_2):
    """utils.py:105-122."""
    return _pairwise(set_1, set_2, 1, "find_intersection3d")


def find_jaccard_overlap3d(set_1, set_2):
    """utils.py:125-149."""
    return _pairwise(set_1, set_2, 0, "find_jaccard_overlap3d")


def volume(box):
    """utils.py:152-154."""
    return (box[3] - box[0]) * (box[4] - box[1]) * (box[5] - box[2])


# ----------------------------------------------------------------------------------------------------------
# metrics (host side)

def _np(t, dtype=None):
    a = t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    return a.astype(dtype) if dtype is not None else a


def _iou_one_to_many(box, others):
    """fp32, same operation order as utils.py:105-149."""
    lo = np.maximum(box[None, :3], others[:, :3])
    hi = np.minimum(box[None, 3:], others[:, 3:])
    ext = np.clip(hi - lo, 0, None).astype(np.float32)
    inter = ext[:, 0] * ext[:, 1] * ext[:, 2]
    va = (box[3] - box[0]) * (box[4] - box[1]) * (box[5] - box[2])
    vb = (others[:, 3] - others[:, 0]) * (others[:, 4] - others[:, 1]) * (others[:, 5] - others[:, 2])
    with np.errstate(invalid="ignore", divide="ignore"):
        return inter / (va + vb - inter)


def compute_metrics_per_class(det_class_images, det_class_boxes, det_class_scores, true_class_images, true_class_boxes,
                              true_class_difficulties, min_overlap):
    """utils.py:157-239 (numpy arrays in, numpy arrays out).  Detections are visited in stable descending score
    order; a detection is a true positive iff its best-IoU (first max) ground truth in the same image has
    IoU > min_overlap, is not 'difficult' and is not claimed yet."""
    order = np.lexsort((np.arange(det_class_scores.shape[0]), -det_class_scores.astype(np.float64)))
    det_class_images, det_class_boxes, det_class_scores = det_class_images[order], det_class_boxes[order], det_class_scores[order]
    detected = np.zeros(true_class_boxes.shape[0], dtype=np.uint8)
    tp = np.zeros(det_class_boxes.shape[0], dtype=np.float32)
    fp = np.zeros(det_class_boxes.shape[0], dtype=np.float32)
    for d in range(det_class_boxes.shape[0]):
        same = np.nonzero(true_class_images == det_class_images[d])[0]
        if same.size == 0:
            fp[d] = 1
            continue
        ov = _iou_one_to_many(det_class_boxes[d], true_class_boxes[same])
        ind = int(np.argmax(ov))  # first maximum
        if ov[ind] > min_overlap:
            if not true_class_difficulties[same[ind]]:
                if detected[same[ind]] == 0:
                    tp[d] = 1
                    detected[same[ind]] = 1
                else:
                    fp[d] = 1
        else:
            fp[d] = 1
    vols = np.array([volume(b) for b, dif in zip(true_class_boxes, true_class_difficulties) if not dif], dtype=np.float32)
    return tp, fp, detected, det_class_scores, vols[detected == 1], vols[detected == 0]


def calculate_mAP(det_boxes, det_labels, det_scores, true_boxes, true_labels, true_difficulties, min_overlap=0.5,
                  return_detail=False):
    """utils.py:242-396.  Inputs: lists (one entry per image) of tensors (any device) or arrays.
    Returns ``(APs, mAP)`` or, with ``return_detail``, the reference's detail dict (torch CPU tensors / floats)."""
    assert len(det_boxes) == len(det_labels) == len(det_scores) == len(true_boxes) == len(true_labels) == len(true_difficulties)
    n_classes = len(label_map)
    n_img = len(true_labels)
    t_img = np.concatenate([np.full(len(true_labels[i]), i, dtype=np.int64) for i in range(n_img)])
    t_box = np.concatenate([_np(b, np.float32).reshape(-1, 6) for b in true_boxes])
    t_lab = np.concatenate([_np(l, np.int64) for l in true_labels])
    t_dif = np.concatenate([_np(d).astype(bool) for d in true_difficulties])
    assert t_img.shape[0] == t_box.shape[0] == t_lab.shape[0]
    d_img = np.concatenate([np.full(len(det_labels[i]), i, dtype=np.int64) for i in range(n_img)])
    d_box = np.concatenate([_np(b, np.float32).reshape(-1, 6) for b in det_boxes])
    d_lab = np.concatenate([_np(l, np.int64) for l in det_labels])
    d_sco = np.concatenate([_np(s, np.float32) for s in det_scores])
    assert d_img.shape[0] == d_box.shape[0] == d_lab.shape[0] == d_sco.shape[0]

    average_precisions = np.zeros(n_classes - 1, dtype=np.float32)
    per = {}
    n_easy = 0
    for c in range(1, n_classes):
        ts, ds = t_lab == c, d_lab == c
        n_easy = int((~t_dif[ts]).sum())
        if ds.sum() == 0:
            continue
        tp, fp, detected, sorted_scores, found, not_found = compute_metrics_per_class(
            d_img[ds], d_box[ds], d_sco[ds], t_img[ts], t_box[ts], t_dif[ts], min_overlap)
        fn = np.float32((1 - detected.astype(np.float32)).sum())
        tps = np.float32(tp.sum())
        with np.errstate(invalid="ignore", divide="ignore"):
            recall = tps / (tps + fn)
            precision = tps / (tps + np.float32(fp.sum()))
            f1 = (2 * precision * recall) / (precision + recall)
            ctp, cfp = np.cumsum(tp, dtype=np.float32), np.cumsum(fp, dtype=np.float32)
            cprec = ctp / (ctp + cfp + np.float32(1e-10))
            crec = ctp / np.float32(n_easy)
        thresholds = torch.arange(start=0, end=1.1, step=.1).tolist()
        precs = np.zeros(len(thresholds), dtype=np.float32)
        for i, t in enumerate(thresholds):
            above = crec >= t
            precs[i] = cprec[above].max() if above.any() else 0.
        average_precisions[c - 1] = precs.mean(dtype=np.float32)
        per[c] = dict(tp=tp, fp=fp, detected=detected, scores=sorted_scores, found=found, not_found=not_found,
                      recall=float(recall), precision=float(precision), f1=float(f1))

    mean_average_precision = float(average_precisions.mean())
    aps = {rev_label_map[c + 1]: float(v) for c, v in enumerate(average_precisions.tolist())}
    if not return_detail:
        return aps, mean_average_precision
    T = torch.from_numpy
    if 1 in per:  # n_classes == 2 in the reference (utils.py:359-369)
        p = per[1]
        return {"APs": aps[rev_label_map[1]], "mAP": mean_average_precision, "precision": p["precision"],
                "recall": p["recall"], "f1_score": p["f1"], "sorted_det_scores": {1: T(p["scores"])},
                "TP": T(p["tp"]), "FP": T(p["fp"]), "n_true_boxes": int(p["detected"].shape[0]),
                "found_boxes_volumes_per_class": T(p["found"]), "not_found_boxes_volumes_per_class": T(p["not_found"])}
    vols = np.array([volume(b) for b in t_box], dtype=np.float32)  # utils.py:370-380: nothing detected
    return {"APs": 0., "mAP": mean_average_precision, "precision": 0., "recall": 0., "f1_score": 0.,
            "sorted_det_scores": {}, "TP": torch.zeros(0), "FP": torch.zeros(0), "n_true_boxes": n_easy,
            "found_boxes_volumes_per_class": torch.zeros(0), "not_found_boxes_volumes_per_class": T(vols)}
