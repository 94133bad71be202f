"""Oracle prior<->object matching, target encoding and MultiBox loss
(test infrastructure; see oracle/__init__.py).

Restates reference ``lesions3d/ssd3d.py:741-941`` (``MultiBoxLoss``).  Tie-break rules are written
out explicitly instead of being inherited from ``torch.max``:

* best object per prior / best prior per object: the FIRST index attaining the maximum
  (``Tensor.max(dim)`` on CPU; the 100-object chunking of ssd3d.py:786-839 followed by a
  first-max over chunks is the same thing as one global first-max);
* force-match ``object_for_each_prior[prior_for_each_object] = arange(n)`` (ssd3d.py:865) with
  duplicate indices: the LAST (highest-numbered) object wins.
"""
import torch

from . import boxes as B


def _first_argmax(vals, dim):
    """Index of the first element equal to the max along ``dim`` (explicit tie rule)."""
    mx = vals.max(dim=dim, keepdim=True).values
    n = vals.size(dim)
    shape = [1, 1]
    shape[dim] = n
    idx = torch.arange(n).view(shape).expand_as(vals)
    first = torch.where(vals == mx, idx, torch.full_like(idx, n)).min(dim=dim).values
    return mx.squeeze(dim), first


def normalize_threshold(threshold):
    """ssd3d.py:762-773: float or 1-list -> hard; 2-list -> soft band [lo, hi)."""
    if isinstance(threshold, list):
        if len(threshold) == 1:
            return "hard", float(threshold[0]), None
        assert len(threshold) == 2
        return "soft", float(threshold[0]), float(threshold[1])
    if isinstance(threshold, float):
        return "hard", threshold, None
    raise Exception("Type error. Expected float or list of floats for threshold")


def match_image(gt_boxes, gt_labels, priors_c, threshold):
    """One image.  Returns (true_class (P,) int64 in {-1,0,label}, matched object (P,) int64,
    overlap after force-match (P,) f32, true_locs (P,6) f32).  ssd3d.py:851-887."""
    mode, lo, hi = normalize_threshold(threshold)
    priors_xyz = B.cxcycz_to_xyz(priors_c)
    iou = B.iou_matrix(gt_boxes, priors_xyz)  # (n, P)   ssd3d.py:798
    overlap, obj = _first_argmax(iou, 0)  # ssd3d.py:801,833-837
    _, prior_for_obj = _first_argmax(iou, 1)  # ssd3d.py:812
    obj = obj.clone()
    overlap = overlap.clone()
    for o in range(gt_boxes.size(0)):  # ascending => last writer wins   ssd3d.py:865,868
        obj[prior_for_obj[o]] = o
        overlap[prior_for_obj[o]] = 1.0
    label = gt_labels[obj].clone()  # ssd3d.py:871
    if mode == "hard":
        label[overlap < lo] = 0  # ssd3d.py:877
    else:
        label[overlap < lo] = 0  # ssd3d.py:879
        label[(overlap >= lo) & (overlap < hi)] = -1  # ssd3d.py:880-881
    true_locs = B.encode(B.xyz_to_cxcycz(gt_boxes[obj]), priors_c)  # ssd3d.py:887
    return label, obj, overlap, true_locs


def match_batch(boxes, labels, priors_c, threshold):
    """ssd3d.py:847-888.  Images without objects keep all-zero targets (ssd3d.py:854-855)."""
    n, p = len(boxes), priors_c.size(0)
    true_locs = torch.zeros((n, p, 6), dtype=torch.float32)
    true_classes = torch.zeros((n, p), dtype=torch.long)
    matched = torch.zeros((n, p), dtype=torch.long)
    for i in range(n):
        if boxes[i].size(0) == 0:
            continue
        true_classes[i], matched[i], _, true_locs[i] = match_image(boxes[i], labels[i], priors_c, threshold)
    return true_classes, true_locs, matched


def focal_binary(logit_fg, is_pos, gamma=2.0, weight=0.25):
    """The loss the reference names in its commented line ssd3d.py:760: MONAI ``FocalLoss(reduction="none", gamma=2,
    to_onehot_y=True, include_background=False, weight=0.25)``.  With two classes and the background channel
    excluded that is a sigmoid focal loss on the foreground logit alone:  weight * (1 - p_t)^gamma * BCE(x, t).
    MONAI is not installed here and the reference holds no output of it: PARITY UNPINNED (published formula only)."""
    t = is_pos.to(logit_fg.dtype)
    bce = torch.nn.functional.binary_cross_entropy_with_logits(logit_fg, t, reduction="none")
    invprobs = torch.nn.functional.logsigmoid(-logit_fg * (t * 2 - 1))  # log(1 - p_t)
    return weight * (invprobs * gamma).exp() * bce


def multibox_loss(pred_locs, pred_scores, boxes, labels, priors_c, threshold, hard_negative_mining=False,
                  smooth_l1=False, focal=False, neg_pos_ratio=3):
    """-> (conf_loss, loc_loss) scalars.  ssd3d.py:890-941 live code path: plain L1 mean over
    positives x 6 (``nn.L1Loss``, ssd3d.py:758,896); cross entropy of every prior with ignored (-1)
    priors re-targeted to class 0 and then zeroed (ssd3d.py:913-917); all negatives + positives summed and
    divided by the number of positives (ssd3d.py:924-933; hard-negative mining is commented out).

    Optional variants (all off = the live path above):
    ``hard_negative_mining``: the commented recipe ssd3d.py:926-932 - per image, only the ``neg_pos_ratio * n_positives``
    largest entries of ``conf_loss_neg`` (positives zeroed) are summed; equal losses keep index order (stable sort).
    ``smooth_l1``: ``nn.SmoothL1Loss()`` (beta 1, mean) in place of ``nn.L1Loss()`` (the attribute's name, ssd3d.py:758).
    ``focal``: ``focal_binary`` in place of the cross entropy (two classes only)."""
    n, p, n_classes = pred_scores.shape
    assert p == priors_c.size(0) == pred_locs.size(1)
    true_classes, true_locs, _ = match_batch(boxes, labels, priors_c, threshold)
    positive = true_classes > 0
    if smooth_l1:
        loc_loss = torch.nn.functional.smooth_l1_loss(pred_locs[positive], true_locs[positive])
    else:
        loc_loss = (pred_locs[positive] - true_locs[positive]).abs().mean()
    if focal:
        assert n_classes == 2, "the focal variant is defined for background + one class"
        ce = focal_binary(pred_scores[..., 1], positive)
    else:
        target = true_classes.clamp(min=0).view(-1)
        ce = torch.nn.functional.cross_entropy(pred_scores.reshape(-1, n_classes), target, reduction="none").view(n, p)
    ce = torch.where(true_classes < 0, torch.zeros_like(ce), ce)
    neg = ce.clone()
    neg[positive] = 0.0
    if hard_negative_mining:
        n_hard = neg_pos_ratio * positive.sum(dim=1)  # (N)  ssd3d.py:907-908
        neg_sorted, _ = neg.sort(dim=1, descending=True, stable=True)  # ssd3d.py:926
        ranks = torch.arange(p).unsqueeze(0).expand_as(neg_sorted)  # ssd3d.py:927
        hard = ranks < n_hard.unsqueeze(1)  # ssd3d.py:928
        conf_loss = (neg_sorted[hard].sum() + ce[positive].sum()) / positive.sum().float()  # ssd3d.py:929,932
    else:
        conf_loss = (neg.sum() + ce[positive].sum()) / positive.sum().float()
    if torch.isnan(loc_loss):  # ssd3d.py:938-940 (empty-GT batch)
        raise Exception("Loss is NaN")
    return conf_loss, loc_loss
