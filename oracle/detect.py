"""Oracle detection decoding + 3D NMS (test infrastructure; see oracle/__init__.py).

Restates reference ``lesions3d/ssd3d.py:344-460`` (``LSSD3D.detect_objects``).  The sort is made
explicitly STABLE (descending score, ties by ascending prior index) — what torch's CPU sort was
observed to do — so keep-lists are well defined.
"""
import numpy as np
import torch

from . import boxes as B


def stable_desc_order(scores):
    """Permutation sorting ``scores`` descending, equal scores by ascending original index."""
    s = np.asarray(scores)
    return np.lexsort((np.arange(s.shape[0]), -s.astype(np.float64)))


def nms_keep(boxes_sorted, max_overlap):
    """Greedy suppression over score-sorted boxes.  ssd3d.py:407-426: for each not-yet-suppressed box,
    suppress every box whose IoU with it is STRICTLY greater than ``max_overlap`` (all indices, not only
    later ones), then un-suppress the box itself.  Returns the boolean keep mask."""
    n = boxes_sorted.shape[0]
    iou = B.iou_matrix(boxes_sorted, boxes_sorted).numpy()
    suppress = np.zeros(n, dtype=bool)
    for i in range(n):
        if suppress[i]:
            continue
        suppress |= iou[i] > max_overlap  # NaN compares False
        suppress[i] = False
    return ~suppress


def detect_objects(pred_locs, pred_scores, priors_c, min_score, max_overlap, top_k, return_prior_index=False):
    """-> lists (len N) of boxes (k,6) f32 corner-form fractional, labels (k,) i64, scores (k,) f32
    [+ prior index (k,) i64, -1 for the placeholder].  ssd3d.py:360-460."""
    n, p, n_classes = pred_scores.shape
    assert p == priors_c.size(0) == pred_locs.size(1)  # ssd3d.py:370
    probs = torch.softmax(pred_scores, dim=2)  # ssd3d.py:363
    out_b, out_l, out_s, out_i = [], [], [], []
    for i in range(n):
        decoded = B.cxcycz_to_xyz(B.decode(pred_locs[i], priors_c))  # ssd3d.py:373
        ib, il, isc, ii = [], [], [], []
        for c in range(1, n_classes):
            cs = probs[i, :, c]
            cand = torch.nonzero(cs > min_score).view(-1)  # strict >   ssd3d.py:388
            if cand.numel() == 0:
                continue
            order = torch.from_numpy(stable_desc_order(cs[cand].numpy()))  # ssd3d.py:397
            cand = cand[order][: min(10 * top_k, cand.numel())]  # ssd3d.py:401-403
            keep = torch.from_numpy(nms_keep(decoded[cand], max_overlap))
            kept = cand[keep]
            ib.append(decoded[kept])
            il.append(torch.full((kept.numel(),), c, dtype=torch.long))
            isc.append(cs[kept])
            ii.append(kept)
        if not ib:  # ssd3d.py:437-440 placeholder
            ib = [torch.tensor([[0., 0., 0., 1., 1., 1.]])]
            il = [torch.zeros(1, dtype=torch.long)]
            isc = [torch.zeros(1)]
            ii = [torch.full((1,), -1, dtype=torch.long)]
        ib, il, isc, ii = torch.cat(ib), torch.cat(il), torch.cat(isc), torch.cat(ii)
        if isc.numel() > top_k:  # ssd3d.py:449-453
            order = torch.from_numpy(stable_desc_order(isc.numpy()))[:top_k]
            ib, il, isc, ii = ib[order], il[order], isc[order], ii[order]
        out_b.append(ib)
        out_l.append(il)
        out_s.append(isc)
        out_i.append(ii)
    if return_prior_index:
        return out_b, out_l, out_s, out_i
    return out_b, out_l, out_s
