"""Oracle prior (anchor) generation (test infrastructure; see oracle/__init__.py).

Restates reference ``lesions3d/ssd3d.py:286-342`` (``LSSD3D.create_prior_boxes``) and the scale
rule of ``ssd3d.py:228-234``.  Vectorised, float64 then rounded once to float32 — exactly what the
reference's Python-float lists -> ``torch.FloatTensor`` do.
"""
import numpy as np
import torch


def default_scales(feature_ids, input_size, min_object_size=6, max_object_size=14):
    """ssd3d.py:228-234: linspace(min/size0, max/size0, n_layers) keyed by feature index."""
    vals = np.linspace(min_object_size / input_size[0], max_object_size / input_size[0], len(feature_ids))
    return {f: float(v) for f, v in zip(feature_ids, vals)}


def feature_map_dims(input_size, cube=None, last_feature=7):
    """Spatial dims after each backbone feature (what ssd3d.py:102-110 measures with a dummy pass).

    k3/p1 convs: out = floor((in - 1) / stride) + 1.  Strides follow mobilenet.py:13-20 truncated at
    ``last_feature`` (ssd3d.py:65-75); stem stride is (2,2,2) for cubes else (1,2,2) (ssd3d.py:60).
    """
    if cube is None:
        cube = input_size[0] == input_size[1] == input_size[2]
    strides = [(2, 2, 2) if cube else (1, 2, 2)]
    chans = [32]
    for c, n, s in [[64, 1, 2], [128, 2, 2], [256, 2, 2], [512, 6, 2], [1024, 2, 1]]:
        for i in range(n):
            if len(strides) - 1 == last_feature:
                break
            st = s if i == 0 else 1
            strides.append((st, st, st))
            chans.append(c)
    dims = {}
    cur = tuple(input_size)
    for i, st in enumerate(strides):
        cur = tuple((d - 1) // s + 1 for d, s in zip(cur, st))
        dims[i] = cur
    return dims, chans


def make_priors(fmap_dims, scales, boxes_per_location=2):
    """(P,6) float32 centre-size priors.

    ssd3d.py:302-337.  Flat order: feature maps in key order, then array axes (i,j,k) row-major, then
    the ``boxes_per_location`` sizes.  NOTE the reference's axis swap (ssd3d.py:307-309):
    centre = [ (j+.5)/d1, (i+.5)/d0, (k+.5)/d2 ].  Sizes: s, then s + s/div for div=1..bpl-1
    (ssd3d.py:330-331).  Finally clamp to [0,1] on the whole tensor (ssd3d.py:337).
    """
    rows = []
    for f in fmap_dims:
        d0, d1, d2 = fmap_dims[f]
        s = float(scales[f])
        i, j, k = np.meshgrid(np.arange(d0), np.arange(d1), np.arange(d2), indexing="ij")
        cx = (j.astype(np.float64) + 0.5) / d1
        cy = (i.astype(np.float64) + 0.5) / d0
        cz = (k.astype(np.float64) + 0.5) / d2
        sizes = [s] + [s + s / div for div in range(1, boxes_per_location)]
        per_loc = []
        for sz in sizes:
            per_loc.append(np.stack([cx, cy, cz, np.full_like(cx, sz), np.full_like(cx, sz), np.full_like(cx, sz)], axis=-1))
        rows.append(np.stack(per_loc, axis=3).reshape(-1, 6))
    pri = torch.from_numpy(np.concatenate(rows, axis=0)).to(torch.float32)
    return pri.clamp_(0, 1)
