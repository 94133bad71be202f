"""Deterministic, torch-RNG-independent weights and inputs for the golden vectors.

The fixtures under tests/golden/ do not carry the ~3.8 MB of weights: both the minting script
(make_golden.py, which drives the reference) and the tests (which drive the oracle and the HIP
path) rebuild them from here.  numpy's MT19937 ``RandomState`` stream is stable across numpy
versions.
"""
import zlib

import numpy as np
import torch


def _rs(seed, name):
    return np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31 - 1))


def fill_state_dict(state_dict, seed=1234):
    """Returns a new dict with the same keys/shapes/dtypes, filled deterministically per key name."""
    out = {}
    for k, v in state_dict.items():
        rs = _rs(seed, k)
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            out[k] = torch.from_numpy(rs.uniform(0.5, 1.5, shape).astype(np.float32))
        elif k.endswith("running_mean"):
            out[k] = torch.from_numpy(rs.uniform(-0.1, 0.1, shape).astype(np.float32))
        elif k == "rescale_factors":
            out[k] = torch.full(shape, 20.0)
        elif v.dim() == 1 and k.endswith(".weight"):  # BN gamma
            out[k] = torch.from_numpy(rs.uniform(0.5, 1.5, shape).astype(np.float32))
        elif k.endswith(".bias"):
            out[k] = torch.from_numpy(rs.uniform(-0.1, 0.1, shape).astype(np.float32))
        else:  # conv weight (Cout, Cin/groups, 3|1, ., .)
            fan_in = int(np.prod(shape[1:]))
            b = (3.0 / fan_in) ** 0.5
            out[k] = torch.from_numpy(rs.uniform(-b, b, shape).astype(np.float32))
    return out


def make_volume_batch(seed, n, c, size):
    """Noise volumes with a few bright cubes, roughly normalised (shape-compatible with the
    reference's synthetic data recipe; exact recipe parity is not needed for arithmetic parity)."""
    rs = _rs(seed, "volume")
    x = rs.rand(n, c, *size).astype(np.float32)
    for i in range(n):
        for _ in range(3):
            e = int(rs.randint(6, 14))
            o = [int(rs.randint(0, s - e)) for s in size]
            x[i, :, o[0]:o[0] + e, o[1]:o[1] + e, o[2]:o[2] + e] += 0.4
    x = np.clip(x, 0, 1)
    x = (x - x.mean()) / x.std()
    return torch.from_numpy(x.astype(np.float32))


def make_gt(seed, n, size, n_obj_range=(1, 5), edge_range=(6, 14)):
    """Ragged ground truth: list of (n_i,6) f32 corner boxes in fractional coords + (n_i,) i64 labels.
    Boxes follow the reference's convention (utils.py:500): inclusive min/max voxel index / size."""
    rs = _rs(seed, "gt")
    boxes, labels = [], []
    for _ in range(n):
        k = int(rs.randint(n_obj_range[0], n_obj_range[1] + 1))
        b = []
        for _ in range(k):
            e = int(rs.randint(edge_range[0], edge_range[1]))
            o = [int(rs.randint(0, s - e)) for s in size]
            b.append([o[0] / size[0], o[1] / size[1], o[2] / size[2],
                      (o[0] + e - 1) / size[0], (o[1] + e - 1) / size[1], (o[2] + e - 1) / size[2]])
        boxes.append(torch.tensor(b, dtype=torch.float32).view(-1, 6))
        labels.append(torch.ones(k, dtype=torch.long))
    return boxes, labels


def make_head_outputs(seed, n, p, n_classes=2, loc_std=0.5, score_std=2.0):
    rs = _rs(seed, "heads")
    locs = torch.from_numpy((rs.randn(n, p, 6) * loc_std).astype(np.float32))
    scores = torch.from_numpy((rs.randn(n, p, n_classes) * score_std).astype(np.float32))
    return locs, scores
