"""Load the reference's hot-path modules (ssd3d, mobilenet, utils) in THIS container only.

Test infrastructure for minting golden vectors.  The reference imports
pytorch_lightning / monai / wandb, none of which is installed; none of them is on the
arithmetic path (all arithmetic is stock torch CPU ops), so they are replaced by inert
in-memory modules before import.  Nothing from /root/reference is copied; this file is never
shipped to / used on the GPU box (tests read the committed fixtures only).
"""
import sys
import types

import torch
import torch.nn as nn

REFERENCE_DIR = "/root/reference/lesions3d"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_standins():
    class LightningModule(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self.current_epoch = 0
            self.global_step = 0

        @property
        def device(self):
            return torch.device("cpu")

        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        def lr_schedulers(self):
            return None

    class LightningDataModule(object):
        pass

    _mod("pytorch_lightning", LightningModule=LightningModule, LightningDataModule=LightningDataModule)
    _mod("wandb")

    class _Dummy(object):
        def __init__(self, *a, **k):
            pass

    class MapTransform(object):
        def __init__(self, *a, **k):
            pass

    class InvertibleTransform(object):
        def __init__(self, *a, **k):
            pass

    _mod("monai")
    _mod("monai.losses", FocalLoss=_Dummy)
    _mod("monai.networks")
    _mod("monai.networks.blocks", Convolution=_Dummy)
    _mod("monai.config", KeysCollection=object)
    _mod("monai.config.type_definitions", NdarrayOrTensor=object)
    _mod("monai.transforms")
    _mod("monai.transforms.transform", MapTransform=MapTransform)
    _mod("monai.transforms.inverse", InvertibleTransform=InvertibleTransform)
    _mod("monai.data", box_area=lambda b: (b[:, 3] - b[:, 0]) * (b[:, 4] - b[:, 1]) * (b[:, 5] - b[:, 2]))


def load_reference():
    """Returns the reference modules (ssd3d, mobilenet, utils) with untouched arithmetic."""
    install_standins()
    if REFERENCE_DIR not in sys.path:
        sys.path.insert(0, REFERENCE_DIR)
    import matplotlib
    matplotlib.use("Agg")
    import ssd3d as ref_ssd3d
    import mobilenet as ref_mobilenet
    import utils as ref_utils
    return ref_ssd3d, ref_mobilenet, ref_utils
