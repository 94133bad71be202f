"""Oracle network: MobileNet-3D backbone + SSD heads on stock torch CPU ops
(test infrastructure; see oracle/__init__.py).

Restates reference ``lesions3d/mobilenet.py:13-49`` (config, conv_bn, Block),
``ssd3d.py:47-110`` (MobileNetBase), ``ssd3d.py:113-169`` (PredictionConvolutions) and
``ssd3d.py:177-263`` (LSSD3D construction + forward).  Module/attribute names are chosen so the
``state_dict`` keys are identical to the reference's, hence weights can be moved
between the reference, this oracle and the HIP product with ``load_state_dict``.
The arithmetic is the same third-party torch CPU code the reference itself calls.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .priors import default_scales, feature_map_dims, make_priors

BACKBONE_CFG = [32, [64, 1, 2], [128, 2, 2], [256, 2, 2], [512, 6, 2], [1024, 2, 1]]  # mobilenet.py:13-20
FEATURE_IDS = (3, 5, 7)  # ssd3d.py:25


class _SepBlock(nn.Module):
    """mobilenet.py:34-49: depthwise k3 (groups=C) + BN + ReLU, pointwise k1 + BN + ReLU."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv3d(cin, cin, 3, stride=stride, padding=1, groups=cin, bias=False)
        self.bn1 = nn.BatchNorm3d(cin)
        self.conv2 = nn.Conv3d(cin, cout, 1, bias=False)
        self.bn2 = nn.BatchNorm3d(cout)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(x)))


class _Backbone(nn.Module):
    """ssd3d.py:47-100."""

    def __init__(self, in_channels, cube, last_feature=7):
        super().__init__()
        stem_stride = (2, 2, 2) if cube else (1, 2, 2)  # ssd3d.py:60
        feats = [nn.Sequential(nn.Conv3d(in_channels, BACKBONE_CFG[0], 3, stride=stem_stride, padding=1, bias=False),
                               nn.BatchNorm3d(BACKBONE_CFG[0]), nn.ReLU(inplace=True))]
        cin = BACKBONE_CFG[0]
        for c, n, s in BACKBONE_CFG[1:]:
            for i in range(n):
                if len(feats) - 1 == last_feature:  # ssd3d.py:66-72 truncation
                    break
                feats.append(_SepBlock(cin, c, s if i == 0 else 1))
                cin = c
        self.features = nn.Sequential(*feats)

    def forward(self, x, keep=FEATURE_IDS):
        out = {}
        for i, f in enumerate(self.features):
            x = f(x)
            if i in keep:
                out[i] = x
        return out


class _Heads(nn.Module):
    """ssd3d.py:113-169."""

    def __init__(self, n_classes, chans, n_boxes=2):
        super().__init__()
        self.n_classes = n_classes
        loc, cl = [], []
        for c in chans:  # creation order loc,cl per scale matters for seeded-init parity (ssd3d.py:129-132)
            loc.append(nn.Conv3d(c, n_boxes * 6, 3, padding=1))
            cl.append(nn.Conv3d(c, n_boxes * n_classes, 3, padding=1))
        self.loc_convs = nn.ModuleList(loc)
        self.cl_convs = nn.ModuleList(cl)

    def forward(self, feats):
        keys = list(feats.keys())
        n = feats[keys[0]].size(0)
        locs, scores = [], []
        for key, lc, cc in zip(keys, self.loc_convs, self.cl_convs):
            locs.append(lc(feats[key]).permute(0, 2, 3, 4, 1).reshape(n, -1, 6))
            scores.append(cc(feats[key]).permute(0, 2, 3, 4, 1).reshape(n, -1, self.n_classes))
        return torch.cat(locs, 1), torch.cat(scores, 1)


class OracleSSD3D(nn.Module):
    """ssd3d.py:172-263 without Lightning.  ``emulate_reference_init`` reproduces SURVEY §0.2-2: the
    reference's constructor runs three train-mode dummy passes (``torch.randn`` input) through the
    backbone (ssd3d.py:238,270,293 -> :102-110), consuming RNG and touching BN running stats."""

    def __init__(self, n_classes=2, input_channels=1, input_size=(64, 64, 64), min_object_size=6,
                 max_object_size=14, emulate_reference_init=True, feature_ids=FEATURE_IDS):
        """``feature_ids``: the keys of the reference's ``aspect_ratios`` argument (ssd3d.py:204-205; train.py:131
        ``--prediction_layers``), default 3 5 7."""
        super().__init__()
        self.feature_ids = FEATURE_IDS = tuple(feature_ids)
        self.n_classes = n_classes
        self.input_size = tuple(input_size)
        self.input_channels = input_channels
        cube = input_size[0] == input_size[1] == input_size[2]
        self.base = _Backbone(input_channels, cube)
        if emulate_reference_init:
            self._dummy_pass()  # ssd3d.py:270
        dims, chans = feature_map_dims(input_size, cube)
        self.pred_convs = _Heads(n_classes, [chans[f] for f in FEATURE_IDS])
        if emulate_reference_init:
            self._dummy_pass()  # ssd3d.py:238
        self.rescale_factors = nn.Parameter(torch.full((1, chans[FEATURE_IDS[0]], 1, 1, 1), 20.0))  # ssd3d.py:240-241 (unused)
        self.scales = default_scales(FEATURE_IDS, input_size, min_object_size, max_object_size)
        if emulate_reference_init:
            self._dummy_pass()  # ssd3d.py:293
        self.priors_cxcycz = make_priors({f: dims[f] for f in FEATURE_IDS}, self.scales)

    def _dummy_pass(self):
        x = torch.randn((1, self.input_channels, *self.input_size))
        for layer in self.base.features:
            x = layer(x)

    def forward(self, image):
        locs, scores = self.pred_convs(self.base(image, keep=self.feature_ids))
        if torch.isnan(scores).any() or torch.isnan(locs).any():  # ssd3d.py:258-261
            raise Exception("NaN in SSD forward")
        return locs, scores
