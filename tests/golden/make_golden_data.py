"""Mint the data-path fixtures by running the REFERENCE's own code (build container only).

    python -m tests.golden.make_golden_data

* ``generate_artificial_dataset.generate_image`` (generate_artificial_dataset.py:63-111) for a few seeds / sizes: the
  volume and mask it hands to ``nib.save`` are captured by an inert in-memory ``nibabel`` (sha256 + samples stored);
* ``utils.BoundingBoxesGeneratord(segmentation_mode="classes").converter`` (utils.py:440-483) on those masks and on hand
  cases (touching cubes, a zero-thickness object, two classes);
* ``predict.save_predictions_example`` (predict.py:155-232) on fixed detections: the ``sub-*_preds.json`` / ``.csv`` it writes.

MONAI / Lightning / wandb / nibabel are absent here and none is on these code paths' arithmetic; they are replaced by inert
stand-ins (``_ref_loader`` + the catch-all modules below).  Only numbers and the two small output files are stored.
"""
import hashlib
import importlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

from . import _ref_loader, cases_data

OUT = os.path.dirname(os.path.abspath(__file__))


class _Dummy(object):
    def __init__(self, *a, **k):
        pass


class _AnyModule(types.ModuleType):
    """``from m import AnyName`` yields an inert class."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy


SAVED = {}


def install():
    _ref_loader.install_standins()
    box_area = sys.modules["monai.data"].box_area
    for name in ("monai.data", "monai.transforms", "pytorch_lightning.loggers", "pytorch_lightning.callbacks"):
        sys.modules[name] = _AnyModule(name)
    sys.modules["monai.data"].box_area = box_area
    sys.modules["monai.transforms"].transform = sys.modules["monai.transforms.transform"]
    sys.modules["monai.transforms"].inverse = sys.modules["monai.transforms.inverse"]

    class Nifti1Image(object):
        def __init__(self, data, affine=None):
            self.data, self.affine = data, affine

    nib = types.ModuleType("nibabel")
    nib.Nifti1Image = Nifti1Image
    nib.save = lambda img, path: SAVED.__setitem__(os.path.basename(path), np.array(img.data, copy=True))
    sys.modules["nibabel"] = nib
    if _ref_loader.REFERENCE_DIR not in sys.path:
        sys.path.insert(0, _ref_loader.REFERENCE_DIR)
    import matplotlib
    matplotlib.use("Agg")


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def gen_generator_and_boxes():
    import utils as ref_utils
    out = {}
    gen = None
    tmp = tempfile.mkdtemp()
    for tag, cfg in cases_data.generator_configs().items():
        argv = ["generate_artificial_dataset.py", "--output_dir", tmp, "--image_size", *map(str, cfg["image_size"]),
                "--object_size", *map(str, cfg["object_size"]), "--num_objects", *map(str, cfg["num_objects"]),
                "--random_seed", str(cfg["random_seed"])]
        old = sys.argv
        sys.argv = argv
        try:
            gen = importlib.reload(gen) if gen is not None else importlib.import_module("generate_artificial_dataset")
        finally:
            sys.argv = old
        conv = ref_utils.BoundingBoxesGeneratord(keys=["seg"], segmentation_mode="classes", n_classes=1)
        for idx in cfg["indices"]:
            SAVED.clear()
            gen.generate_image(gen.image_dir, gen.seg_dir, idx, 1)
            data = SAVED[f"sub-{str(idx).zfill(4)}_image.nii.gz"]
            mask = SAVED[f"sub-{str(idx).zfill(4)}_seg.nii.gz"]
            assert data.dtype == np.float64 and mask.dtype == np.float64
            k = f"{tag}__{idx}"
            out[f"{k}__data_sha256"] = sha(data)
            out[f"{k}__mask_sha256"] = sha(mask)
            out[f"{k}__data_s997"] = data.reshape(-1)[::997]
            out[f"{k}__mask_sum"] = np.float64(mask.sum())
            b, l = conv.converter(mask[None])  # the data module adds the channel axis first (datasets.py:404)
            out[f"{k}__boxes"] = b.numpy()
            out[f"{k}__labels"] = l.numpy()
    for name, (seg, ncls) in cases_data.segmentation_cases().items():
        conv = ref_utils.BoundingBoxesGeneratord(keys=["seg"], segmentation_mode="classes", n_classes=ncls)
        b, l = conv.converter(seg[None].copy())
        out[f"seg__{name}__boxes"] = b.numpy().reshape(-1, 6)
        out[f"seg__{name}__labels"] = l.numpy()
    np.savez_compressed(os.path.join(OUT, "datapath.npz"), **out)
    print("wrote datapath.npz", len(out), "arrays")


def gen_predictions():
    old = sys.argv
    sys.argv = ["predict.py"]
    try:
        import predict as ref_predict
    finally:
        sys.argv = old
    outdir = os.path.join(OUT, "preds")
    os.makedirs(outdir, exist_ok=True)
    for name, c in cases_data.prediction_cases().items():
        loader = [{"img_meta_dict": [{"affine": torch.eye(4)[None]}], "subject": [c["subject"]],
                   "img": torch.zeros((1, 1) + c["img_shape"]), "boxes": [torch.zeros((2, 6))]}]
        tmp = tempfile.mkdtemp()
        ref_predict.save_predictions_example(loader, [c["boxes"].clone()], [c["labels"].clone()], [c["scores"].clone()],
                                             min_score=c["min_score"], output_dir=tmp, save_images=False)
        for ext in ("json", "csv"):
            src = os.path.join(tmp, f"sub-{c['subject']}_preds.{ext}")
            with open(src) as f, open(os.path.join(outdir, f"{name}__sub-{c['subject']}_preds.{ext}"), "w") as g:
                g.write(f.read())
        print("wrote preds fixture", name)


if __name__ == "__main__":
    install()
    gen_generator_and_boxes()
    gen_predictions()
