"""Backbone contract — host-side mirror of the reference's ``lesions3d/base_network.py``.

``base_network.py:64-126`` defines the API every backbone of the detector follows (``.features``,
``.aspect_ratios``, ``.init()``, ``.forward(image) -> {feature index: tensor}``,
``.get_feature_map_infos(input_size, device)``).  The MobileNet backbone that the reference actually uses
(``ssd3d.MobileNetBase``) implements it on HIP kernels.  The alternative ``ConvNetBase`` is built from
``monai.networks.blocks.Convolution`` (not vendored, version unpinned) and is unreachable from ``LSSD3D`` in the
reference (typo at ssd3d.py:281), so only its configuration tables and API shape are mirrored here: SURVEY.md §2
row 5 marks its arithmetic out of scope ("parity unpinned").
"""
import torch.nn as nn


def get_n_params(model):
    """base_network.py:9-16."""
    return sum(p.numel() for p in model.parameters())


config_no_maxpool = [
    # out_channel, stride, padding
    (32, (1, 1, 1), (1, 1, 1)), (32, (1, 1, 1), (1, 1, 1)),
    (64, (2, 2, 2), (1, 1, 1)), (64, (1, 1, 1), (1, 1, 1)),
    (128, (2, 2, 2), (1, 1, 1)), (128, (1, 1, 1), (1, 1, 1)),
    (256, (2, 2, 2), (1, 1, 1)), (256, (1, 1, 1), (1, 1, 1)),
]
config_maxpool_simple = [
    (32, (1, 1, 1), (1, 1, 1)), (32, (1, 1, 1), (1, 1, 1)), ('maxpool3d', (2, 2, 2), (1, 1, 1)),
    (64, (1, 1, 1), (1, 1, 1)), ('maxpool3d', (2, 2, 2), (1, 1, 1)),
    (128, (1, 1, 1), (1, 1, 1)), ('maxpool3d', (2, 2, 2), (1, 1, 1)),
    (256, (1, 1, 1), (1, 1, 1)),
]
config_maxpool_double = [
    (32, (1, 1, 1), (1, 1, 1)), (32, (1, 1, 1), (1, 1, 1)), ('maxpool3d', (2, 2, 2), (1, 1, 1)),
    (64, (1, 1, 1), (1, 1, 1)), (64, (1, 1, 1), (1, 1, 1)), ('maxpool3d', (2, 2, 2), (1, 1, 1)),
    (128, (1, 1, 1), (1, 1, 1)), (128, (1, 1, 1), (1, 1, 1)), ('maxpool3d', (2, 2, 2), (1, 1, 1)),
    (256, (1, 1, 1), (1, 1, 1)),
]

CONVNET_CONFIGS = {
    "convnet_strides": config_no_maxpool,
    "convnet_maxpool_simple": config_maxpool_simple,
    "convnet_maxpool_double": config_maxpool_double,
}


class ConvNetBase(nn.Module):
    """API shape of base_network.py:64-126.  Construction records the configuration; the MONAI-block
    arithmetic is out of scope (see module docstring), so ``forward`` raises."""

    def __init__(self, aspect_ratios, config="convnet_maxpool_double", in_channels=1):
        super(ConvNetBase, self).__init__()
        self.config = CONVNET_CONFIGS[config]
        self.aspect_ratios = aspect_ratios
        self.in_channels = in_channels
        self.features = nn.Sequential()

    def init(self):
        pass

    def forward(self, image):
        raise NotImplementedError("ConvNetBase relies on monai.networks.blocks.Convolution and is unreachable in the "
                                  "reference (ssd3d.py:281); use base_network_config='mobilenet'")

    def get_feature_map_infos(self, input_size, device):
        dims, chans, cur, c = {}, [], tuple(input_size), self.in_channels
        for i, (out_channels, stride, padding) in enumerate(self.config):
            if i > max(self.aspect_ratios.keys()):
                break
            cur = tuple((d + 2 * p - 3) // s + 1 for d, s, p in zip(cur, stride, padding))
            c = c if isinstance(out_channels, str) else out_channels
            dims[i] = cur
            chans.append(c)
        return dims, chans
