"""Input definitions shared by make_golden.py (drives the reference) and the tests (drive the oracle
and the HIP path).  Pure data construction; no reference code involved."""
import numpy as np
import torch

from . import detinit

SIZE_C64 = (64, 64, 64)
FIVE_SCALES = (1, 2, 3, 5, 7)  # `--prediction_layers "1 2 3 5 7"`
P_C64 = 1168


def matching_cases():
    """name -> dict(boxes, labels, threshold, head_seed).  All at 64^3 (P=1168), n_classes=2."""
    cases = {}
    b, l = detinit.make_gt(1, 2, SIZE_C64)
    cases["soft_random"] = dict(boxes=b, labels=l, threshold=[0.1, 0.2], head_seed=11)
    b, l = detinit.make_gt(2, 3, SIZE_C64)
    cases["hard_float"] = dict(boxes=b, labels=l, threshold=0.5, head_seed=12)
    cases["hard_list1"] = dict(boxes=b, labels=l, threshold=[0.3], head_seed=13)
    # two identical GT boxes (same best prior -> last writer wins) + one more, and a second image
    dup = torch.tensor([[0.25, 0.25, 0.25, 0.36, 0.36, 0.36],
                        [0.25, 0.25, 0.25, 0.36, 0.36, 0.36],
                        [0.60, 0.10, 0.40, 0.70, 0.22, 0.52]], dtype=torch.float32)
    b2, l2 = detinit.make_gt(3, 1, SIZE_C64)
    cases["duplicate_gt"] = dict(boxes=[dup, b2[0]], labels=[torch.ones(3, dtype=torch.long), l2[0]],
                                 threshold=[0.1, 0.2], head_seed=14)
    # an image with zero objects inside a non-empty batch (ssd3d.py:854-855)
    b3, l3 = detinit.make_gt(4, 2, SIZE_C64)
    cases["empty_image"] = dict(boxes=[b3[0], torch.zeros((0, 6)), b3[1]],
                                labels=[l3[0], torch.zeros((0,), dtype=torch.long), l3[1]],
                                threshold=[0.1, 0.2], head_seed=15)
    # chunk boundary of ssd3d.py:857 (100 objects per chunk): 100, 101 and 230 objects
    for n_obj, seed in ((100, 5), (101, 6), (230, 7)):
        bb, ll = detinit.make_gt(seed, 1, SIZE_C64, n_obj_range=(n_obj, n_obj), edge_range=(4, 12))
        cases[f"many_{n_obj}"] = dict(boxes=bb, labels=ll, threshold=[0.1, 0.2], head_seed=20 + seed)
    # boxes aligned exactly with priors / shifted so IoU lands inside the [0.1,0.2) band and on ties
    aligned = torch.tensor([[0.0625 - 0.046875, 0.0625 - 0.046875, 0.0625 - 0.046875,
                             0.0625 + 0.046875, 0.0625 + 0.046875, 0.0625 + 0.046875],
                            [0.50, 0.50, 0.50, 0.59375, 0.59375, 0.59375],   # centred between 8 priors: 8-way tie
                            [0.30, 0.31, 0.32, 0.34, 0.38, 0.37]], dtype=torch.float32)
    cases["aligned_ties"] = dict(boxes=[aligned], labels=[torch.ones(3, dtype=torch.long)],
                                 threshold=[0.1, 0.2], head_seed=31)
    return cases


def detect_cases():
    """name -> dict(head_seed | quantized, n, min_score, max_overlap, top_k)."""
    return {
        "ms05_k100": dict(head_seed=41, n=2, min_score=0.5, max_overlap=0.5, top_k=100, quantized=False),
        "ms03_k10": dict(head_seed=42, n=2, min_score=0.3, max_overlap=0.5, top_k=10, quantized=False),
        "ms00_k100": dict(head_seed=43, n=1, min_score=0.0, max_overlap=0.3, top_k=100, quantized=False),
        "none_found": dict(head_seed=44, n=2, min_score=0.9999999, max_overlap=0.5, top_k=100, quantized=False),
        "quant_ties": dict(head_seed=45, n=2, min_score=0.3, max_overlap=0.45, top_k=50, quantized=True),
    }


def detect_inputs(case, p=P_C64):
    locs, scores = detinit.make_head_outputs(case["head_seed"], case["n"], p, loc_std=0.6, score_std=2.5)
    if case["quantized"]:
        # class-0 logit 0, class-1 logit on a coarse grid: many EXACT score ties, no near-ties, so the
        # stable-sort / first-index rules are what decides the keep-list (on CPU and on the GPU alike)
        scores = scores.clone()
        scores[..., 0] = 0.0
        scores[..., 1] = torch.round(scores[..., 1] * 2) / 2
        locs = torch.round(locs * 4) / 4
    return locs, scores


def multiclass_gt(seed, n, size, n_fg=2):
    """Ground truth with foreground labels 1..n_fg (deterministic, alternating from a per-image offset): the
    3-class fixtures (ssd3d.py:132 head width, :384 class loop)."""
    boxes, labels = detinit.make_gt(seed, n, size, n_obj_range=(2, 5))
    labels = [1 + (torch.arange(len(l)) + i) % n_fg for i, l in enumerate(labels)]
    return boxes, [l.to(torch.long) for l in labels]


def multiclass_detect_cases():
    """3-class detect_objects settings: the class loop (ssd3d.py:384), and with total > top_k the cross-class
    re-sort (ssd3d.py:449-453)."""
    return {
        "mc_ms03_k10": dict(head_seed=71, n=2, min_score=0.3, max_overlap=0.5, top_k=10),    # re-sort across classes
        "mc_ms04_k100": dict(head_seed=72, n=2, min_score=0.4, max_overlap=0.45, top_k=100),  # no truncation
        "mc_ms02_k25": dict(head_seed=73, n=1, min_score=0.2, max_overlap=0.3, top_k=25),
        "mc_one_class_empty": dict(head_seed=74, n=2, min_score=0.5, max_overlap=0.5, top_k=20, kill_class=1),
    }


def multiclass_detect_inputs(case, p=P_C64, n_classes=3):
    locs, scores = detinit.make_head_outputs(case["head_seed"], case["n"], p, n_classes=n_classes, loc_std=0.6, score_std=2.5)
    if "kill_class" in case:  # no candidate of that class anywhere: the class loop skips it (ssd3d.py:391-392)
        scores = scores.clone()
        scores[..., case["kill_class"]] = -30.0
    return locs, scores


def boxmath_inputs():
    rs = np.random.RandomState(77)
    lo = rs.uniform(0, 0.8, (24, 3)).astype(np.float32)
    ext = rs.uniform(0.02, 0.3, (24, 3)).astype(np.float32)
    a = np.concatenate([lo, lo + ext], 1)
    a[3, 3:] = a[3, :3]  # degenerate: zero volume
    a[4] = a[5]  # identical pair
    a[6, 3] = a[6, 0]  # zero extent on one axis
    lo2 = rs.uniform(0, 0.8, (40, 3)).astype(np.float32)
    ext2 = rs.uniform(0.02, 0.3, (40, 3)).astype(np.float32)
    b = np.concatenate([lo2, lo2 + ext2], 1)
    b[0] = a[3]  # two degenerate boxes meet -> 0/0
    b[1] = a[7]
    g = (rs.randn(40, 6) * 0.7).astype(np.float32)
    return torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(g)


def map_cases():
    """Detections / ground truth for calculate_mAP: name -> dict of lists of arrays."""
    rs = np.random.RandomState(99)
    cases = {}

    def rnd_boxes(n):
        lo = rs.uniform(0, 0.8, (n, 3)).astype(np.float32)
        return np.concatenate([lo, lo + rs.uniform(0.05, 0.2, (n, 3)).astype(np.float32)], 1)

    tb = [rnd_boxes(4), rnd_boxes(2), rnd_boxes(3)]
    db = []
    for t in tb:
        jit = t + rs.uniform(-0.02, 0.02, t.shape).astype(np.float32)
        db.append(np.concatenate([jit, jit[:1] + 0.01, rnd_boxes(3)], 0))  # hits, a duplicate hit, misses
    ds = [rs.uniform(0.3, 1.0, len(d)).astype(np.float32) for d in db]
    ds[0][1] = ds[0][0]  # a score tie
    cases["mixed"] = dict(det_boxes=db, det_labels=[np.ones(len(d), np.int64) for d in db], det_scores=ds,
                          true_boxes=tb, true_labels=[np.ones(len(t), np.int64) for t in tb])
    # nothing detected: the placeholder of ssd3d.py:437-440 (label 0) in every image
    ph = np.array([[0., 0., 0., 1., 1., 1.]], np.float32)
    cases["no_detections"] = dict(det_boxes=[ph, ph, ph], det_labels=[np.zeros(1, np.int64)] * 3,
                                  det_scores=[np.zeros(1, np.float32)] * 3,
                                  true_boxes=tb, true_labels=[np.ones(len(t), np.int64) for t in tb])
    # an image without ground truth but with detections -> all false positives
    cases["image_without_gt"] = dict(det_boxes=db[:2], det_labels=[np.ones(len(d), np.int64) for d in db[:2]],
                                     det_scores=ds[:2], true_boxes=[tb[0], np.zeros((0, 6), np.float32)],
                                     true_labels=[np.ones(4, np.int64), np.zeros(0, np.int64)])
    return cases
