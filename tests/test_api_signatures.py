"""The drop-in boundary as data: every public class / function of the path keeps the reference's name,
parameter order, parameter kinds and defaults.  ``tests/golden/signatures.json`` is minted from the reference's own objects
with ``inspect.signature`` (tests/golden/make_golden.py::gen_signatures); this test compares the product with it and
executes INTEGRATION.md recipe A (the product's modules under the reference's module names).  CPU only."""
import importlib
import inspect
import json
import os
import sys

import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "signatures.json")
with open(GOLDEN) as f:
    TABLE = json.load(f)

# Deliberate, documented differences (each is an ADDITION that keeps every reference call working):
EXTRA_OK = {
    # keyword-only switches for the loss variants the reference keeps as commented code
    "ssd3d.MultiBoxLoss.__init__": {"hard_negative_mining", "smooth_l1", "focal"},
    "ssd3d.LSSD3D.__init__": {"hard_negative_mining", "smooth_l1", "focal_loss"},
    # oracle-facing extra output, default off
    "ssd3d.LSSD3D.detect_objects": {"return_prior_index"},
    # build-side: priors are built without touching the RNG unless the constructor asks for the reference's draw
    "ssd3d.LSSD3D.create_prior_boxes": {"_draw"},
}
# Lightning passes batch_idx positionally; the reference's training_step omits it (ssd3d.py:467) and validation_step
# requires it (ssd3d.py:533): the product accepts both spellings with a default, which is a superset.
DEFAULT_ADDED_OK = {"ssd3d.LSSD3D.training_step": {"batch_idx"}, "ssd3d.LSSD3D.validation_step": {"batch_idx"},
                    "ssd3d.LSSD3D.predict_step": {"batch_idx"}}


def _resolve(key):
    mod, *path = key.split(".")
    obj = importlib.import_module(f"mslesions3d_amd.{mod}")
    for p in path:
        obj = getattr(obj, p)
    return obj


def _rows(fn):
    rows = []
    for prm in inspect.signature(fn).parameters.values():
        d = prm.default
        if d is inspect.Parameter.empty:
            d = "<required>"
        elif isinstance(d, tuple):
            d = list(d)
        elif not isinstance(d, (int, float, str, bool, type(None), list, dict)):
            d = repr(d)
        rows.append([prm.name, prm.kind.name, d])
    return rows


@pytest.mark.parametrize("key", sorted(TABLE["signatures"].keys()))
def test_signature_matches_reference(key):
    ref = TABLE["signatures"][key]
    mine = _rows(_resolve(key))
    extra = EXTRA_OK.get(key, set())
    added_default = DEFAULT_ADDED_OK.get(key, set())
    mine_core = [r for r in mine if r[0] not in extra]
    for r in mine:
        if r[0] in extra:
            assert r[2] != "<required>", f"{key}: added parameter {r[0]} must have a default"
    ref_names = [r[0] for r in ref]
    got = [r for r in mine_core if r[0] in ref_names]
    assert [r[0] for r in got] == ref_names, f"{key}: parameter order {[r[0] for r in mine_core]} vs reference {ref_names}"
    for (name, kind, default), (n2, k2, d2) in zip(ref, got):
        assert kind == k2, f"{key}.{name}: kind {k2} vs reference {kind}"
        if name in added_default and default == "<required>":
            continue
        assert default == d2, f"{key}.{name}: default {d2!r} vs reference {default!r}"
    for r in mine_core:
        if r[0] not in ref_names:
            assert r[0] in added_default and r[2] != "<required>", f"{key}: parameter {r[0]} is not in the reference"


def test_constants_match_reference():
    from mslesions3d_amd import base_network, mobilenet, ssd3d
    c = TABLE["constants"]
    norm = lambda v: json.loads(json.dumps(v, default=list))
    assert norm(mobilenet.MOBILENET_CONFIGS) == c["mobilenet.MOBILENET_CONFIGS"]
    assert norm(base_network.CONVNET_CONFIGS) == c["base_network.CONVNET_CONFIGS"]
    assert norm({str(k): v for k, v in ssd3d.ASPECT_RATIOS.items()}) == c["ssd3d.ASPECT_RATIOS"]


def test_integration_recipe_a_module_aliases():
    """INTEGRATION.md recipe A: the reference's scripts do ``from ssd3d import *`` / ``from mobilenet import *`` /
    ``from base_network import *`` / ``from utils import *`` - the product's modules under those names must provide every
    hot-path name the scripts use (train.py:154-160, predict.py:257-263, model_insight.py:143-166, eval.py)."""
    import mslesions3d_amd.base_network
    import mslesions3d_amd.mobilenet
    import mslesions3d_amd.ssd3d
    import mslesions3d_amd.utils
    saved = {k: sys.modules.get(k) for k in ("ssd3d", "mobilenet", "base_network", "utils")}
    try:
        sys.modules["ssd3d"] = mslesions3d_amd.ssd3d
        sys.modules["mobilenet"] = mslesions3d_amd.mobilenet
        sys.modules["base_network"] = mslesions3d_amd.base_network
        sys.modules["utils"] = mslesions3d_amd.utils
        ns = {}
        exec("from ssd3d import *\nfrom mobilenet import *\nfrom base_network import *\nfrom utils import *", ns)
        for name in ("LSSD3D", "MultiBoxLoss", "MobileNetBase", "PredictionConvolutions", "Block", "conv_bn", "MOBILENET_CONFIGS",
                     "CONVNET_CONFIGS", "ConvNetBase", "get_n_params", "calculate_mAP", "find_jaccard_overlap3d", "cxcycz_to_xyz",
                     "gcxgcygcz_to_cxcycz", "cxcycz_to_gcxgcygcz", "xyz_to_cxcycz", "find_intersection3d", "volume"):
            assert name in ns, name
        # the constructor call of train.py:154-160, keyword for keyword (CPU: no kernel is launched by construction)
        m = ns["LSSD3D"](n_classes=2, input_channels=1, lr=1e-3, width_mult=1., scheduler="CosineAnnealingLR", batch_size=2,
                         comments="", input_size=(64, 64, 64), compute_metric_every_n_epochs=5, use_wandb=False,
                         aspect_ratios={3: [1.], 5: [1.], 7: [1.]}, scales={}, alpha=1., threshold=[0.1, 0.2], min_object_size=6,
                         max_object_size=14, base_network_config="mobilenet", boxes_per_location=2)
        m.init()
        # (the priors are produced by msl_make_priors on the HIP device: there is no CPU fallback, so without a GPU the
        # attribute stays None until the model meets its first CUDA tensor - tests/test_gpu_model.py checks them bit for bit)
        assert m.priors_cxcycz is None or tuple(m.priors_cxcycz.shape) == (1168, 6)
        assert list(m.state_dict().keys())[0] == "rescale_factors" and len(m.state_dict()) == 103
        assert ns["get_n_params"](m) == sum(p.numel() for p in m.parameters())
        for hook in ("training_step", "validation_step", "predict_step", "configure_optimizers", "load_from_checkpoint", "log",
                     "lr_schedulers", "detect_objects", "create_prior_boxes"):
            assert callable(getattr(m, hook)), hook
        assert m.top_k == 100 and m.min_score == 0.5  # predict.py:259-260 overwrite these attributes
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
