"""Synthetic volumes with cube 'lesions' — the recipe of the reference's ``generate_artificial_dataset.py:63-111``
(uniform noise, ``n+1`` random cubes brightened by 0.4, clipped to [0,1]) followed by what its data module does
(``datasets.py:403`` NormalizeIntensity(nonzero=True); ``utils.py:450-513`` boxes = inclusive min/max voxel index
of each object divided by the image size, label 1).

Two generators with the same distribution:
* ``make_case`` (numpy, seeded exactly like the reference: ``np.random.seed(seed + idx)``, same draw order) —
  for files / host pipelines;
* ``make_batch_on_device`` (torch on the GPU) — for throughput runs, so inputs are resident in HBM and no PCIe
  traffic sits in the timed region.  Ground-truth boxes are the cube extents themselves (the reference gets them
  from connected components of the mask, which merges touching cubes; for the loss kernels that difference is
  immaterial and it is stated wherever these numbers are reported).
Cube edges keep the reference's 6-14 voxel range at every volume size (``train.py:30`` dataset naming
``s6-14``), matching the default prior scales (``min_object_size=6, max_object_size=14``).
"""
import numpy as np
import torch


def generate_volume(idx, image_size=(64, 64, 64), num_objects=(1, 5), object_size=(6, 14), random_seed=0):
    """generate_artificial_dataset.py:63-111 for ``n_classes = 1``, noise on: the float64 volume and mask the reference
    hands to ``nib.save`` (same seed, same draw order: pinned bit for bit by tests/golden/datapath.npz) + the cube list."""
    image_size = tuple(image_size)
    lo, hi = sorted(object_size)
    np.random.seed(random_seed + idx)
    data = np.random.rand(*image_size)
    mask = np.zeros_like(data)
    n_objects = np.random.randint(*num_objects)
    cubes = []
    for _ in range(n_objects + 1):
        size = np.random.randint(lo, hi)
        np.random.randint(0, 1)  # selected_class draw of the reference (n_classes = 1)
        tl = [np.random.randint(0, image_size[i] - size) for i in range(len(image_size))]
        sl = tuple(slice(t, t + size) for t in tl)
        data[sl] = data[sl] + 0.4
        data = data.clip(0, 1)
        mask[sl] = 1
        cubes.append((tl, size))
    return data, mask, cubes


def make_case(idx, image_size=(64, 64, 64), num_objects=(1, 5), object_size=(6, 14), random_seed=0):
    """-> (image float32 (D,H,W) normalised, mask uint8, boxes (n,6) float32 fractional, labels (n,) int64)."""
    data, mask, cubes = generate_volume(idx, image_size, num_objects, object_size, random_seed)
    boxes = [[tl[0] / image_size[0], tl[1] / image_size[1], tl[2] / image_size[2],
              (tl[0] + size - 1) / image_size[0], (tl[1] + size - 1) / image_size[1],
              (tl[2] + size - 1) / image_size[2]] for tl, size in cubes]
    img = data.astype(np.float32)
    nz = img != 0
    img[nz] = (img[nz] - img[nz].mean()) / img[nz].std()
    return img, mask.astype(np.uint8), np.asarray(boxes, dtype=np.float32), np.ones(len(boxes), dtype=np.int64)


def make_batch_on_device(n, image_size, device, channels=1, num_objects=(1, 5), object_size=(6, 14), seed=0):
    """-> images (n,channels,D,H,W) fp32 on ``device``, boxes list[(k,6)], labels list[(k,)] (on ``device``)."""
    g = torch.Generator(device=device).manual_seed(seed)
    rs = np.random.RandomState(seed)
    D, H, W = image_size
    imgs = torch.rand((n, channels, D, H, W), generator=g, device=device, dtype=torch.float32)
    boxes, labels = [], []
    for i in range(n):
        k = int(rs.randint(*num_objects)) + 1
        b = []
        for _ in range(k):
            e = int(rs.randint(object_size[0], object_size[1]))
            o = [int(rs.randint(0, s - e)) for s in image_size]
            for c in range(channels):  # extra channels: same cubes at a different contrast (build-defined)
                imgs[i, c, o[0]:o[0] + e, o[1]:o[1] + e, o[2]:o[2] + e] += 0.4 / (1 + c)
            b.append([o[0] / D, o[1] / H, o[2] / W, (o[0] + e - 1) / D, (o[1] + e - 1) / H, (o[2] + e - 1) / W])
        boxes.append(torch.tensor(b, dtype=torch.float32, device=device))
        labels.append(torch.ones(k, dtype=torch.int64, device=device))
    imgs.clamp_(0, 1)
    mean = imgs.mean(dim=(2, 3, 4), keepdim=True)
    std = imgs.std(dim=(2, 3, 4), keepdim=True, unbiased=False)
    return (imgs - mean) / std, boxes, labels
