"""Inputs of the data-path fixtures (shared by the minting script and the tests; no reference code involved)."""
import numpy as np
import torch


def generator_configs():
    return {
        "s32": dict(image_size=(32, 32, 32), object_size=(4, 9), num_objects=(1, 5), random_seed=0, indices=(0, 1, 7)),
        "s64": dict(image_size=(64, 64, 64), object_size=(6, 14), num_objects=(1, 5), random_seed=0, indices=(0, 3)),
        "s48x40x36": dict(image_size=(48, 40, 36), object_size=(5, 11), num_objects=(2, 4), random_seed=11, indices=(2,)),
    }


def segmentation_cases():
    """name -> (segmentation (D,H,W) float64, n_classes)."""
    cases = {}
    s = np.zeros((16, 16, 16))
    s[2:6, 3:9, 4:8] = 1
    s[8:12, 8:12, 8:12] = 1
    cases["two_cubes"] = (s, 1)
    s = np.zeros((16, 16, 16))
    s[2:6, 2:6, 2:6] = 1
    s[5:9, 5:9, 5:9] = 1  # overlaps the first cube: ONE connected component
    s[9:12, 2:5, 2:5] = 1  # face-adjacent along axis 0 to nothing, separate
    cases["merged_cubes"] = (s, 1)
    s = np.zeros((16, 16, 16))
    s[2:6, 3:9, 4:5] = 1  # thickness 1 along the last axis -> zero volume -> dropped (utils.py:476-481)
    s[8:12, 8:12, 8:12] = 1
    s[13:14, 1:3, 1:3] = 1  # thickness 1 along the first axis -> dropped
    cases["zero_thickness"] = (s, 1)
    s = np.zeros((12, 14, 10))
    s[1:4, 1:5, 1:4] = 1
    s[4:5, 1:2, 1:2] = 1  # touches the first object only diagonally-free? face neighbour along axis 0 -> same component
    s[6:10, 6:12, 2:8] = 2
    s[1:3, 8:12, 6:9] = 2
    cases["two_classes_noncube"] = (s, 2)
    # (an all-background mask makes the reference's converter raise - FloatTensor([]) / FloatTensor(6), utils.py:472 - so
    # there is no reference output to pin; boxes_from_segmentation returns empty tensors there)
    return cases


def prediction_cases():
    g = torch.Generator().manual_seed(5)
    lo = torch.rand((7, 3), generator=g) * 0.7
    boxes = torch.cat([lo, lo + 0.05 + torch.rand((7, 3), generator=g) * 0.25], 1)
    boxes[2, 0] = -0.03   # unclamped detections leave [0, 1] (ssd3d.py:373 decodes without clamping)
    boxes[4, 5] = 1.04
    scores = torch.tensor([0.97, 0.91, 0.73, 0.5, 0.4999, 0.31, 0.12])
    labels = torch.tensor([1, 1, 1, 1, 1, 0, 1])
    cases = {"mixed": dict(subject="0007", img_shape=(64, 64, 64), boxes=boxes, labels=labels, scores=scores, min_score=0.5),
             "noncube": dict(subject="0012", img_shape=(48, 64, 40), boxes=boxes[:4].clone(), labels=labels[:4].clone(),
                             scores=scores[:4].clone(), min_score=0.3),
             # the no-detection placeholder of ssd3d.py:437-440
             "placeholder": dict(subject="0003", img_shape=(32, 32, 32), boxes=torch.tensor([[0., 0., 0., 1., 1., 1.]]),
                                 labels=torch.tensor([0]), scores=torch.tensor([0.]), min_score=0.5)}
    return cases
