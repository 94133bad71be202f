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
