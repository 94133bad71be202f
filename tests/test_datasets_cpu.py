"""Host-side synthetic data path: generator recipe, box extraction, split and collate contract."""
import numpy as np
import torch

from mslesions3d_amd import datasets as DS
from mslesions3d_amd.synth import make_case


def test_generator_layout_and_boxes(tmp_path):
    root = DS.generate_artificial_dataset(str(tmp_path), "toy", num_images=10, image_size=(32, 32, 32), object_size=(4, 9))
    ds = DS.ExampleDataset(n_classes=1, batch_size=2, data_dir=str(tmp_path), dataset_name="toy")
    assert ds.data_dir == root and len(ds.subjects_list) == 10
    ds.setup("fit")
    assert len(ds.train_dataset) == 8 and len(ds.test_dataset) == 2  # 80/20, random_state 970205
    batch = next(iter(ds.train_dataloader()))
    assert set(batch) == {"img", "seg", "boxes", "labels", "subject", "img_meta_dict", "seg_meta_dict", "img_transforms", "seg_transforms"}
    assert tuple(batch["img"].shape) == (2, 1, 32, 32, 32) and batch["img"].dtype == torch.float32
    for b, l in zip(batch["boxes"], batch["labels"]):
        assert b.shape[1] == 6 and b.dtype == torch.float32 and l.dtype == torch.int64 and bool((l == 1).all())
        assert bool((b[:, 3:] > b[:, :3]).all()) and float(b.min()) >= 0 and float(b.max()) < 1
    nz = batch["img"][0][batch["img"][0] != 0]
    assert abs(float(nz.mean())) < 1e-3 and abs(float(nz.std(unbiased=False)) - 1) < 1e-3


def test_boxes_follow_inclusive_voxel_convention():
    seg = np.zeros((16, 16, 16), np.uint8)
    seg[2:6, 3:9, 4:5] = 1  # thickness 1 along the last axis -> zero volume -> dropped (utils.py:476-481)
    seg[8:12, 8:12, 8:12] = 1
    b, l = DS.boxes_from_segmentation(seg, 1)
    assert b.shape[0] == 1 and l.tolist() == [1]
    assert torch.allclose(b[0], torch.tensor([8, 8, 8, 11, 11, 11], dtype=torch.float32) / 16)


def test_make_case_is_seeded_like_the_reference():
    a = make_case(3, (24, 24, 24))
    b = make_case(3, (24, 24, 24))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    assert not np.array_equal(a[0], make_case(4, (24, 24, 24))[0])


# ---- fixtures minted from the reference's own code (tests/golden/make_golden_data.py) ------------------------------------
import hashlib  # noqa: E402
import os  # noqa: E402

import pytest  # noqa: E402

from mslesions3d_amd import predict as PR  # noqa: E402
from mslesions3d_amd.synth import generate_volume  # noqa: E402
from tests.golden import cases_data  # noqa: E402
from tests.util import GOLDEN_DIR, golden  # noqa: E402


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


@pytest.mark.parametrize("tag", list(cases_data.generator_configs().keys()))
def test_generator_reproduces_the_reference_volumes_bit_for_bit(tag):
    """generate_artificial_dataset.py:63-111: same seed -> the same float64 volume and mask, and boxes_from_segmentation on
    that mask -> the boxes of the reference's BoundingBoxesGeneratord (utils.py:440-483), float32 bit for bit."""
    g = golden("datapath")
    cfg = cases_data.generator_configs()[tag]
    for idx in cfg["indices"]:
        data, mask, cubes = generate_volume(idx, cfg["image_size"], cfg["num_objects"], cfg["object_size"], cfg["random_seed"])
        k = f"{tag}__{idx}"
        assert data.dtype == np.float64 and mask.dtype == np.float64
        assert np.array_equal(_sha(data), g[f"{k}__data_sha256"]), k
        assert np.array_equal(_sha(mask), g[f"{k}__mask_sha256"]), k
        assert np.array_equal(data.reshape(-1)[::997], g[f"{k}__data_s997"])
        b, l = DS.boxes_from_segmentation(mask[None], 1)
        assert np.array_equal(b.numpy(), g[f"{k}__boxes"]) and np.array_equal(l.numpy(), g[f"{k}__labels"])


@pytest.mark.parametrize("name", list(cases_data.segmentation_cases().keys()))
def test_boxes_from_segmentation_matches_the_reference(name):
    g = golden("datapath")
    seg, ncls = cases_data.segmentation_cases()[name]
    b, l = DS.boxes_from_segmentation(seg[None], ncls)
    assert b.dtype == torch.float32 and l.dtype == torch.int64
    assert np.array_equal(b.numpy().reshape(-1, 6), g[f"seg__{name}__boxes"]), (b, g[f"seg__{name}__boxes"])
    assert np.array_equal(l.numpy(), g[f"seg__{name}__labels"])


def test_boxes_from_an_empty_mask():
    b, l = DS.boxes_from_segmentation(np.zeros((1, 8, 8, 8)), 1)  # the reference raises here (utils.py:472): no vector
    assert tuple(b.shape) == (0, 6) and l.numel() == 0


@pytest.mark.parametrize("name", list(cases_data.prediction_cases().keys()))
def test_prediction_files_match_the_reference_byte_for_byte(name, tmp_path):
    """predict.py:155-232 (save_predictions_example, save_images off) on fixed detections."""
    c = cases_data.prediction_cases()[name]
    PR.save_predictions(c["subject"], c["img_shape"], c["boxes"].numpy(), c["labels"].numpy(), c["scores"].numpy(),
                        c["min_score"], str(tmp_path))
    for ext in ("json", "csv"):
        ours = open(tmp_path / f"sub-{c['subject']}_preds.{ext}").read()
        ref = open(os.path.join(GOLDEN_DIR, "preds", f"{name}__sub-{c['subject']}_preds.{ext}")).read()
        assert ours == ref, (ext, ours, ref)


# ---- augmentations (train.py:132-145) ------------------------------------------------------------------------------------
def _aug_case():
    img = np.random.RandomState(0).rand(1, 16, 16, 16).astype(np.float32)
    seg = np.zeros((1, 16, 16, 16), np.uint8)
    seg[0, 2:6, 3:9, 4:8] = 1
    seg[0, 9:13, 10:14, 1:4] = 1
    return img, seg


def test_flip_and_rotate90_move_the_boxes_with_the_mask():
    img, seg = _aug_case()
    b0, _ = DS.boxes_from_segmentation(seg, 1)
    rs = np.random.RandomState(1)
    fi, fs = DS.AUGMENTATIONS["flip"](img, seg, rs, spatial_axis=(0, 1, 2), prob=1.0)
    assert np.array_equal(fi, img[:, ::-1, ::-1, ::-1])
    b1, _ = DS.boxes_from_segmentation(fs, 1)
    # inclusive voxel boxes: lo' = (n-1-hi)/n, hi' = (n-1-lo)/n ; component order may change -> compare as sets
    exp = torch.cat([(15 / 16) - b0[:, 3:], (15 / 16) - b0[:, :3]], 1)
    assert sorted(map(tuple, b1.tolist())) == sorted(map(tuple, exp.tolist()))
    ri, rseg = DS.AUGMENTATIONS["rotate90"](img, seg, np.random.RandomState(3), spatial_axes=(1, 2), prob=1.0)
    k = int(np.random.RandomState(3).randint(3)) + 1 if False else None  # (k is drawn after the probability draw)
    assert ri.shape == img.shape and rseg.sum() == seg.sum()
    assert any(np.array_equal(ri, np.rot90(img, kk, (2, 3))) for kk in (1, 2, 3))
    b2, _ = DS.boxes_from_segmentation(rseg, 1)
    assert b2.shape == b0.shape and torch.allclose(b2[:, 0].sort()[0], b0[:, 0].sort()[0])  # axis 0 untouched
    # prob = 0 -> identity, and nothing but the probability is drawn
    a, b = DS.AUGMENTATIONS["flip"](img, seg, rs, prob=0.0)
    assert a is img and b is seg


def test_affine_translate_shifts_the_mask_and_scale_keeps_the_shape():
    img, seg = _aug_case()

    class FixedShift(np.random.RandomState):
        def uniform(self, lo=0.0, hi=1.0, size=None):
            return 2.0

    ti, ts = DS._aug_affine(img, seg, FixedShift(0), translate_range=(3, 3, 3), prob=1.0)
    # output voxel o samples input voxel o + 2 -> content moves by -2 along every axis
    assert np.array_equal(ts[0, 0:4, 1:7, 2:6], seg[0, 2:6, 3:9, 4:8]) and ts.dtype == seg.dtype
    np.testing.assert_allclose(ti[0, 2:10, 2:10, 2:10], img[0, 4:12, 4:12, 4:12], rtol=1e-6)
    si, ss = DS._aug_affine(img, seg, np.random.RandomState(2), scale_range=(0.15, 0.15, 0.15), padding_mode="reflection", prob=1.0)
    assert si.shape == img.shape and ss.shape == seg.shape and set(np.unique(ss)) <= {0, 1}


def test_train_dataset_applies_the_selected_augmentations(tmp_path):
    DS.generate_artificial_dataset(str(tmp_path), "toy", num_images=5, image_size=(24, 24, 24), object_size=(4, 8))
    aug = DS.select_augmentations(["flip", "rotate90", "translate"])
    assert [n for n, _ in aug] == ["flip", "rotate90", "rotate90", "rotate90", "affine"]
    with pytest.raises(ValueError):
        DS.select_augmentations(["zoom"])
    plain = DS.ExampleDataset(n_classes=1, batch_size=1, data_dir=str(tmp_path), dataset_name="toy")
    plain.setup("fit")
    augd = DS.ExampleDataset(n_classes=1, batch_size=1, data_dir=str(tmp_path), dataset_name="toy", augmentations=aug)
    augd.setup("fit")
    changed = 0
    for i in range(len(plain.train_dataset)):
        a, b = plain.train_dataset[i], augd.train_dataset[i]
        assert a["img"].shape == b["img"].shape and b["boxes"].shape[1] == 6
        changed += int(not torch.equal(a["img"], b["img"]))
    assert changed > 0
    for i in range(len(plain.test_dataset)):  # the test pipeline is never augmented (datasets.py:417)
        assert torch.equal(plain.test_dataset[i]["img"], augd.test_dataset[i]["img"])


# ---- data-parallel shards and per-sample augmentation draws --------------------------------------------------------------
@pytest.mark.parametrize("n,world", [(8, 2), (13, 4), (5, 8), (16, 1)])
def test_shard_sampler_deals_disjoint_covering_shards(n, world):
    shards = []
    for r in range(world):
        s = DS.ShardSampler(n, r, world, shuffle=True, seed=970205)
        s.set_epoch(3)
        shards.append(list(s))
        assert len(shards[-1]) == len(s) == -(-n // world)   # every rank runs the same number of steps
    flat = [i for sh in shards for i in sh]
    assert set(flat) == set(range(n))                          # the shards cover the data set
    assert len(flat) - len(set(flat)) == (-n) % world          # duplicates only from the wrap-around padding
    if n >= world:
        for a in range(world):
            for b in range(a + 1, world):
                assert not (set(shards[a]) & set(shards[b])) or (-n) % world
    again = DS.ShardSampler(n, 0, world, shuffle=True, seed=970205)
    again.set_epoch(4)
    if n > 2:
        assert list(again) != shards[0] or n <= world          # another epoch, another deal
    again.set_epoch(3)
    assert list(again) == shards[0]                            # a resumed run sees the same order
    with pytest.raises(ValueError):
        DS.ShardSampler(4, 2, 2)


def test_augmentation_draws_differ_across_workers_and_epochs(tmp_path):
    """with num_workers > 0 every DataLoader worker is a forked copy of the data set, re-created each epoch -
    a generator stored on the data set hands every worker and every epoch the same flips.  The draws are now a function of
    (seed, epoch, subject): different per sample, different per epoch, identical for a re-run of the same epoch."""
    DS.generate_artificial_dataset(str(tmp_path), "toy", num_images=10, image_size=(16, 16, 16), object_size=(3, 6))
    aug = DS.select_augmentations(["flip", "rotate90"])
    d = DS.ExampleDataset(n_classes=1, batch_size=1, num_workers=2, data_dir=str(tmp_path), dataset_name="toy", augmentations=aug)
    d.setup("fit")

    def epoch(e):
        d.set_epoch(e)
        return {b["subject"][0]: b["img"].clone() for b in d.train_dataloader()}
    e0, e0b, e1 = epoch(0), epoch(0), epoch(1)
    assert set(e0) == set(e1) == set(e0b) and len(e0) == 8
    assert all(torch.equal(e0[s], e0b[s]) for s in e0)               # deterministic per (seed, epoch, subject)
    assert sum(not torch.equal(e0[s], e1[s]) for s in e0) >= 3       # another epoch draws differently
    # the draws of two samples of one epoch are independent streams (not one stream replayed per worker)
    ds = d.train_dataset
    ds.set_epoch(0)
    firsts = {tuple(ds.sample_rng(i).randint(0, 1 << 30, 4)) for i in range(len(ds))}
    assert len(firsts) == len(ds)
