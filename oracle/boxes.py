"""Oracle box math (test infrastructure; see oracle/__init__.py).

Restates reference ``lesions3d/utils.py:42-154``.  Operation ORDER is part of the contract
(fp32, no reassociation), because the matched indices and NMS keep-lists derived from these
values must be bit-exact.
"""
import torch


def cxcycz_to_xyz(c):
    """centre-size (cx,cy,cz,w,h,d) -> corners (lo3, hi3).  utils.py:42-51: ``c - size/2``, ``c + size/2``."""
    half = c[:, 3:] / 2
    return torch.cat([c[:, :3] - half, c[:, :3] + half], dim=1)


def xyz_to_cxcycz(b):
    """corners -> centre-size.  utils.py:92-102: centre ``(hi + lo)/2``, size ``hi - lo``."""
    return torch.cat([(b[:, 3:] + b[:, :3]) / 2, b[:, 3:] - b[:, :3]], dim=1)


def encode(c, priors_c):
    """utils.py:71-89: g_c = (c - c_p) / (s_p / 10);  g_s = log(s / s_p) * 5."""
    return torch.cat([(c[:, :3] - priors_c[:, :3]) / (priors_c[:, 3:] / 10),
                      torch.log(c[:, 3:] / priors_c[:, 3:]) * 5], dim=1)


def decode(g, priors_c):
    """utils.py:54-68: c = g_c * s_p / 10 + c_p  (multiply first, then divide);  s = exp(g_s / 5) * s_p."""
    return torch.cat([g[:, :3] * priors_c[:, 3:] / 10 + priors_c[:, :3],
                      torch.exp(g[:, 3:] / 5) * priors_c[:, 3:]], dim=1)


def intersection(a, b):
    """utils.py:105-122: clamp(min(hi) - max(lo), 0), product of the three extents in axis order 0,1,2."""
    lo = torch.maximum(a[:, None, :3], b[None, :, :3])
    hi = torch.minimum(a[:, None, 3:], b[None, :, 3:])
    ext = (hi - lo).clamp(min=0)
    return ext[..., 0] * ext[..., 1] * ext[..., 2]


def box_volume(b):
    """utils.py:138-143: (x1-x0)*(y1-y0)*(z1-z0), left to right."""
    return (b[:, 3] - b[:, 0]) * (b[:, 4] - b[:, 1]) * (b[:, 5] - b[:, 2])


def iou_matrix(a, b):
    """utils.py:125-149: inter / ((vol_a + vol_b) - inter); 0/0 -> NaN for two degenerate boxes."""
    inter = intersection(a, b)
    union = box_volume(a)[:, None] + box_volume(b)[None, :] - inter
    return inter / union
