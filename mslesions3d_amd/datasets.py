"""MONAI-free synthetic data path — host-side mirror of the parts of the reference's
``lesions3d/datasets.py:50-95,359-485`` and ``generate_artificial_dataset.py`` that the example pipeline uses.

Same directory layout (``<root>/multiple_objects/one_class/<name>/{images,labels}/sub-XXXX_{image,seg}.*``), same
80/20 split (``train_test_split(random_state=970205)``), same per-sample pipeline (add channel ->
NormalizeIntensity(nonzero=True) -> boxes from connected components of the mask, ``utils.py:450-513``) and the same
batch dict from ``collate_fn`` (``datasets.py:86-94``).  Files are ``.npy`` because nibabel is not installed here
(``.nii.gz`` is read when nibabel is importable).  This is data plumbing in front of the hot path, on the host, as in
the reference; MONAI's exact NormalizeIntensity arithmetic is not pinned (MONAI absent) — population mean/std over
the non-zero voxels is used.
"""
import os
from os.path import join as pjoin

import numpy as np
import torch
from scipy.ndimage import label as cc_label
import zlib

from torch.utils.data import DataLoader, Dataset, Sampler

from .synth import generate_volume, make_case  # noqa: F401


def generate_artificial_dataset(output_dir, dataset_name, num_images=20, image_size=(64, 64, 64), num_objects=(1, 5),
                                object_size=(6, 14), random_seed=0):
    """generate_artificial_dataset.py:63-111 (n_classes = 1, noise on), one file pair per case."""
    root = pjoin(output_dir, "multiple_objects", "one_class", dataset_name)
    os.makedirs(pjoin(root, "images"), exist_ok=True)
    os.makedirs(pjoin(root, "labels"), exist_ok=True)
    for idx in range(num_images):
        data, mask, _ = generate_volume(idx, image_size, num_objects, object_size, random_seed)
        np.save(pjoin(root, "images", f"sub-{str(idx).zfill(4)}_image.npy"), data.astype(np.float32))
        np.save(pjoin(root, "labels", f"sub-{str(idx).zfill(4)}_seg.npy"), mask.astype(np.uint8))
    return root


def boxes_from_segmentation(seg, n_classes=1):
    """BoundingBoxesGeneratord, 'classes' mode (utils.py:450-513): per class, connected components -> inclusive
    [min, max] voxel index per axis / image size; zero-volume boxes dropped."""
    seg = np.squeeze(seg)
    size = np.array(seg.shape * 2, dtype=np.float32)
    boxes, labels = [], []
    for c in range(1, n_classes + 1):
        comp, n = cc_label(seg == c)
        for k in range(1, n + 1):
            idx = np.where(comp == k)
            b = [idx[0].min(), idx[1].min(), idx[2].min(), idx[0].max(), idx[1].max(), idx[2].max()]
            boxes.append(b)
            labels.append(c)
    boxes = torch.from_numpy(np.asarray(boxes, dtype=np.float32).reshape(-1, 6) / size)
    labels = torch.tensor(labels, dtype=torch.long)
    if boxes.numel():
        keep = ((boxes[:, 3] - boxes[:, 0]) * (boxes[:, 4] - boxes[:, 1]) * (boxes[:, 5] - boxes[:, 2])) != 0
        boxes, labels = boxes[keep], labels[keep]
    return boxes, labels


# ---- augmentations (reference train.py:132-145 -> datasets.py:99-122 registry; train pipeline only) -----------------
# The reference applies MONAI's RandFlipd / RandRotate90d / RandAffined to the (image, segmentation) pair BEFORE the boxes
# are extracted, so boxes always follow from the transformed mask.  Same here, on numpy arrays with a channel axis in
# front.  flip / rotate90 are index permutations (exact; MONAI calls the same flip / rot90).  translate / scale are the
# two RandAffined uses: resampling with bilinear (image) / nearest (mask) interpolation and reflection padding.  MONAI is
# absent, so its random-number stream and its affine grid convention are NOT pinned (documented in DESIGN.md); the
# transforms are drawn from ``np.random.RandomState(seed)``.

def _aug_flip(img, seg, rs, spatial_axis=(0, 1, 2), prob=0.1):
    if rs.rand() >= prob:
        return img, seg
    ax = tuple(a + 1 for a in ((spatial_axis,) if np.isscalar(spatial_axis) else spatial_axis))
    return np.flip(img, ax), np.flip(seg, ax)


def _aug_rotate90(img, seg, rs, spatial_axes=(0, 1), prob=0.1, max_k=3):
    if rs.rand() >= prob:
        return img, seg
    k = int(rs.randint(max_k)) + 1
    ax = tuple(a + 1 for a in spatial_axes)
    return np.rot90(img, k, ax), np.rot90(seg, k, ax)


def _rand_range(rs, rng, n=3):
    """MONAI's per-axis parameter draw: a (lo, hi) pair draws uniform(lo, hi), a number f draws uniform(-f, f);
    axes beyond the given entries get 0."""
    out = []
    for f in tuple(rng)[:n]:
        out.append(rs.uniform(f[0], f[1]) if isinstance(f, (tuple, list)) else rs.uniform(-f, f))
    return out + [0.0] * (n - len(out))


def _aug_affine(img, seg, rs, mode=("bilinear", "nearest"), translate_range=None, scale_range=None,
                padding_mode="reflection", prob=0.1):
    from scipy.ndimage import affine_transform
    if rs.rand() >= prob:
        return img, seg
    shift = _rand_range(rs, translate_range) if translate_range is not None else [0.0] * 3
    zoom = [1.0 + v for v in _rand_range(rs, scale_range)] if scale_range is not None else [1.0] * 3
    pad = {"reflection": "reflect", "border": "nearest", "zeros": "constant"}[padding_mode]
    centre = (np.array(img.shape[1:], dtype=np.float64) - 1) / 2
    mat = np.diag(zoom)
    off = centre - mat @ centre + np.array(shift, dtype=np.float64)  # output voxel o samples input voxel M o + off
    outs = []
    for a, m in ((img, mode[0]), (seg, mode[1])):
        order = 1 if m == "bilinear" else 0
        outs.append(np.stack([affine_transform(c.astype(np.float32), mat, offset=off, order=order, mode=pad) for c in a]).astype(a.dtype))
    return outs[0], outs[1]


AUGMENTATIONS = {"flip": _aug_flip, "rotate90": _aug_rotate90, "affine": _aug_affine}

# train.py:132-143: the names the CLI accepts and the parameters the reference binds to them
REFERENCE_AUGMENTATIONS = [("flip", {"spatial_axis": (0, 1, 2), "prob": .5}),
                           ("rotate90", {"spatial_axes": (1, 2), "prob": .5}),
                           ("rotate90", {"spatial_axes": (0, 1), "prob": .5}),
                           ("rotate90", {"spatial_axes": (0, 2), "prob": .5}),
                           ("translate", {"mode": ("bilinear", "nearest"), "translate_range": (-3, 3), "prob": .7}),
                           ("scale", {"mode": ("bilinear", "nearest"), "scale_range": (0.15, 0.15, 0.15),
                                      "padding_mode": "reflection", "prob": .7})]


def select_augmentations(names):
    """train.py:145: keep the reference's entries whose name was asked for; translate / scale both become 'affine'."""
    unknown = set(names) - {n for n, _ in REFERENCE_AUGMENTATIONS}
    if unknown:
        raise ValueError(f"unknown augmentation(s) {sorted(unknown)}; known: flip rotate90 translate scale")
    return [(n.replace("translate", "affine").replace("scale", "affine"), kw) for n, kw in REFERENCE_AUGMENTATIONS if n in names]


def _load(path_noext):
    if os.path.exists(path_noext + ".npy"):
        return np.load(path_noext + ".npy")
    import nibabel as nib  # only when .nii.gz data is supplied
    return np.asarray(nib.load(path_noext + ".nii.gz").dataobj)


class ShardSampler(Sampler):
    """Data-parallel shard of one epoch (BASELINE north_star: "data-parallel training shards synthetic volumes across the
    8 GPUs"): every rank draws the SAME permutation of the cases from (seed, epoch), pads it by wrapping around to a
    multiple of the world size - every rank runs the same number of steps, each of which contains collectives - and takes
    every world-th entry from its rank on.  Shards of one epoch are disjoint (up to the wrap-around padding) and cover the
    data set; ``set_epoch`` re-deals them.  world = 1 is the plain seeded shuffle, so a resumed run (train.py --checkpoint)
    sees the order the interrupted one would have seen."""

    def __init__(self, n, rank=0, world=1, shuffle=True, seed=0):
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world {world}")
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, rank, world, shuffle, seed, 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def indices(self):
        order = (np.random.RandomState((self.seed * 1000003 + self.epoch) % (2 ** 31 - 1)).permutation(self.n) if self.shuffle
                 else np.arange(self.n))
        if self.n == 0:
            return order
        per = -(-self.n // self.world)
        order = np.resize(order, per * self.world)  # wrap-around padding
        return order[self.rank::self.world]

    def __iter__(self):
        return iter(self.indices().tolist())

    def __len__(self):
        return -(-self.n // self.world) if self.n else 0


class _Cases(Dataset):
    def __init__(self, root, subjects, n_classes, augmentations=None, seed=0):
        self.root, self.subjects, self.n_classes = root, subjects, n_classes
        self.augmentations = list(augmentations or [])
        for t in self.augmentations:
            if (t if isinstance(t, str) else t[0]) not in AUGMENTATIONS:
                raise ValueError(f"unknown transform {t!r}")
        # augmentation randomness is drawn per SAMPLE from (seed, epoch, subject): DataLoader workers are forked copies of
        # this object, so a generator stored here would hand every worker - and every epoch - the same stream
        self.seed, self.epoch = seed, 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def sample_rng(self, i):
        key = f"{self.seed}:{self.epoch}:{self.subjects[i]}".encode()
        return np.random.RandomState(zlib.crc32(key) & 0x7FFFFFFF)

    def __len__(self):
        return len(self.subjects)

    def __getitem__(self, i):
        s = self.subjects[i]
        img = _load(pjoin(self.root, "images", f"sub-{s}_image")).astype(np.float32)
        seg = _load(pjoin(self.root, "labels", f"sub-{s}_seg"))
        nz = img != 0
        if nz.any():
            std = img[nz].std()
            img[nz] = (img[nz] - img[nz].mean()) / (std if std != 0 else 1.0)
        img, seg = img[None], np.asarray(seg)[None]  # add_channel
        rs = self.sample_rng(i) if self.augmentations else None
        for t in self.augmentations:  # between normalizeintensity and bounding_boxes_generator (datasets.py:417-430)
            name, kw = (t, {}) if isinstance(t, str) else t
            img, seg = AUGMENTATIONS[name](img, seg, rs, **kw)
        img = np.ascontiguousarray(img)
        boxes, labels = boxes_from_segmentation(seg, self.n_classes)
        return {"img": torch.from_numpy(img), "boxes": boxes, "labels": labels, "seg": [boxes, labels], "subject": s,
                "img_meta_dict": {"affine": np.eye(4)}, "seg_meta_dict": {}, "img_transforms": [], "seg_transforms": []}


def collate_fn(batch):
    """datasets.py:50-95: images stacked, ragged boxes / labels kept as lists."""
    boxes = [b["boxes"] for b in batch]
    labels = [b["labels"] for b in batch]
    return {"img": torch.stack([b["img"] for b in batch], 0), "seg": [boxes, labels], "boxes": boxes, "labels": labels,
            "subject": [b["subject"] for b in batch], "img_meta_dict": [b["img_meta_dict"] for b in batch],
            "seg_meta_dict": [b["seg_meta_dict"] for b in batch], "img_transforms": [b["img_transforms"] for b in batch],
            "seg_transforms": [b["seg_transforms"] for b in batch]}


class ExampleDataset:
    """datasets.py:359-485 surface: ``setup(stage)``, ``train_dataloader()``, ``test_dataloader()``,
    ``predict_dataloader()``, ``train_dataset`` / ``test_dataset``."""

    def __init__(self, n_classes=1, objects="multiple", percentage=1., augmentations=None, batch_size=8, num_workers=0,
                 verbose=False, random_state=970205, cache=False, subject=None,
                 data_dir="../data/artificial_dataset", dataset_name=None, rank=0, world_size=1):
        """``rank`` / ``world_size`` (not in the reference, which is single-GPU): this process's data-parallel shard of
        the train and validation cases (``ShardSampler``)."""
        assert n_classes == 1 or n_classes == 2
        d = data_dir + "/multiple_objects" if objects == "multiple" else data_dir
        d = pjoin(d, "one_class") if n_classes == 1 else pjoin(d, "double_class")
        self.data_dir = d if dataset_name is None else pjoin(d, dataset_name)
        self.batch_size, self.num_workers, self.random_state = batch_size, num_workers, random_state
        self.n_classes, self.subject, self.percentage = n_classes, subject, percentage
        self.augmentations = augmentations
        self.rank, self.world_size, self.epoch = rank, world_size, 0
        subs = sorted(s.replace("sub-", "")[:4] for s in os.listdir(pjoin(self.data_dir, "images")) if "sub-" in s)
        self.subjects_list = subs[:int(percentage * len(subs))] if percentage > 0 else subs
        self.train_dataset = self.test_dataset = self.predict_dataset = None

    def setup(self, stage=None):
        from sklearn.model_selection import train_test_split
        if self.subject is not None:
            train, test = [self.subject], [self.subject]
        else:
            train, test = train_test_split(self.subjects_list, test_size=0.2, random_state=self.random_state)
        self.train_dataset = _Cases(self.data_dir, train, self.n_classes, self.augmentations, self.random_state)
        self.test_dataset = _Cases(self.data_dir, test, self.n_classes)
        self.predict_dataset = _Cases(self.data_dir, train if stage == "predict_train" else test, self.n_classes)

    def set_epoch(self, epoch):
        """Call before ``train_dataloader()`` of every epoch: re-deals the shards and the augmentation draws."""
        self.epoch = int(epoch)
        if self.train_dataset is not None:
            self.train_dataset.set_epoch(epoch)

    def _loader(self, ds, shuffle, bs=None, shard=False):
        sampler = None
        if shard:
            sampler = ShardSampler(len(ds), self.rank, self.world_size, shuffle, self.random_state)
            sampler.set_epoch(self.epoch)
        return DataLoader(ds, batch_size=bs or self.batch_size, shuffle=shuffle if sampler is None else False, sampler=sampler,
                          num_workers=self.num_workers, collate_fn=collate_fn, drop_last=False)

    def train_dataloader(self):
        return self._loader(self.train_dataset, True, shard=True)

    def test_dataloader(self):
        return self._loader(self.test_dataset, False, shard=self.world_size > 1)

    def predict_dataloader(self):
        return self._loader(self.predict_dataset, False, 1)
