"""Host-side product code that needs no GPU: the detection metric and the checkpoint format."""
import os
import pickle

import numpy as np
import pytest
import torch

from tests.golden import cases
from tests.util import golden


@pytest.mark.parametrize("name", list(cases.map_cases().keys()))
@pytest.mark.parametrize("ov", [0.1, 0.5])
def test_product_calculate_map_matches_reference_golden(name, ov):
    """``mslesions3d_amd.utils.calculate_mAP`` (what ``LSSD3D._metrics`` and ``predict.py`` call) against the outputs of the
    reference's ``calculate_mAP(return_detail=True)`` (utils.py:242-396) on the same detections (map.npz)."""
    from mslesions3d_amd.utils import calculate_mAP
    g = golden("map")
    c = cases.map_cases()[name]
    T = lambda xs: [torch.from_numpy(np.asarray(x)) for x in xs]
    dif = [torch.zeros(len(x), dtype=torch.bool) for x in c["true_labels"]]
    d = calculate_mAP(T(c["det_boxes"]), T(c["det_labels"]), T(c["det_scores"]), T(c["true_boxes"]), T(c["true_labels"]), dif,
                      min_overlap=ov, return_detail=True)
    tag = f"{name}__{ov}"
    for k in ("APs", "mAP", "precision", "recall", "f1_score", "n_true_boxes"):
        np.testing.assert_allclose(float(d[k]), float(g[f"{tag}__{k}"]), rtol=1e-6, equal_nan=True)
    for k in ("TP", "FP", "found_boxes_volumes_per_class", "not_found_boxes_volumes_per_class"):
        np.testing.assert_allclose(np.asarray(d[k], np.float32), g[f"{tag}__{k}"], rtol=1e-6)
    # the two-value form (utils.py:382-383) agrees with the detail dict
    aps, m = calculate_mAP(T(c["det_boxes"]), T(c["det_labels"]), T(c["det_scores"]), T(c["true_boxes"]), T(c["true_labels"]),
                           dif, min_overlap=ov)
    np.testing.assert_allclose(m, float(g[f"{tag}__mAP"]), rtol=1e-6)
    assert list(aps.keys()) == ["lesion"]


def _small_model(**kw):
    from mslesions3d_amd.ssd3d import LSSD3D
    torch.manual_seed(3)
    return LSSD3D(n_classes=2, input_channels=1, input_size=(64, 64, 64), threshold=[0.1, 0.2], lr=1e-3, **kw)


def test_checkpoint_is_weights_only_loadable(tmp_path):
    """Own checkpoints hold tensors and plain Python types only (numpy scalars cast), in Lightning's key layout, and load
    through ``torch.load(weights_only=True)`` — there is no other loader."""
    from mslesions3d_amd.ssd3d import LSSD3D
    m = _small_model(scales={3: np.float64(0.1), 5: np.float32(0.2), 7: 0.3}, min_score=np.float64(0.25))
    m.current_epoch, m.global_step = 3, 17
    p = str(tmp_path / "own.ckpt")
    m.save_checkpoint(p)
    raw = torch.load(p, map_location="cpu", weights_only=True)
    assert set(raw.keys()) >= {"state_dict", "hyper_parameters", "epoch", "global_step"}
    assert type(raw["hyper_parameters"]["scales"][3]) is float and type(raw["hyper_parameters"]["min_score"]) is float
    m2 = LSSD3D.load_from_checkpoint(p, top_k=7)
    assert (m2.current_epoch, m2.global_step, m2.top_k) == (3, 17, 7)
    assert m2.scales == {3: 0.1, 5: float(np.float32(0.2)), 7: 0.3}
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


class _Evil:
    def __reduce__(self):
        return (os.system, ("echo checkpoint code execution > /dev/null",))


def test_checkpoint_with_pickled_object_is_refused(tmp_path):
    """A file that needs arbitrary unpickling (what a reference Lightning .ckpt — or a hostile file — is) is refused with a
    clear error; nothing retries it with ``weights_only=False``."""
    from mslesions3d_amd.ssd3d import LSSD3D
    m = _small_model()
    p = str(tmp_path / "evil.ckpt")
    with open(p, "wb") as f:
        pickle.dump({"state_dict": {k: v for k, v in m.state_dict().items()}, "hyper_parameters": {"x": _Evil()}}, f)
    with pytest.raises(RuntimeError, match="weights_only=True"):
        LSSD3D.load_from_checkpoint(p)
    import inspect
    import mslesions3d_amd.ssd3d as S
    assert "weights_only=False" not in inspect.getsource(S)


def test_softmax_restatement_is_bit_exact(tmp_path):
    """csrc/softmax_exp.h (the text detect.hip compiles for the GPU) built for the host with g++ and compared with
    ``torch.softmax`` bit for bit: the NMS candidate order hangs on the last bit of these probabilities (ssd3d.py:363,
    :397), so the kernel restates ATen's CPU algorithm (Sleef expf, sum in class order, multiply by the reciprocal)."""
    import ctypes
    import subprocess
    src = tmp_path / "sm.cpp"
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mslesions3d_amd", "csrc", "softmax_exp.h")
    src.write_text('#define MSL_FN static inline\n#include "%s"\nextern "C" void run(const float* x, int n, int ncls, float* out) {\n'
                   '  for (int i = 0; i < n; ++i) msl_softmax_foreground(x + (long)i * ncls, ncls, out + (long)i * (ncls - 1), 1);\n}\n' % hdr)
    so = tmp_path / "sm.so"
    subprocess.run(["g++", "-O2", "-std=c++20", "-ffp-contract=off", "-mfma", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    lib = ctypes.CDLL(str(so))
    rs = np.random.RandomState(3)
    for ncls, std in ((2, 2.5), (2, 12.0), (3, 2.5), (5, 1.0), (2, 60.0)):
        x = (rs.randn(300000, ncls) * std).astype(np.float32)
        out = np.empty((x.shape[0], ncls - 1), np.float32)
        lib.run(x.ctypes.data_as(ctypes.c_void_p), x.shape[0], ncls, out.ctypes.data_as(ctypes.c_void_p))
        ref = torch.softmax(torch.from_numpy(x), dim=1).numpy()[:, 1:]
        assert np.array_equal(out.view(np.uint32), np.ascontiguousarray(ref).view(np.uint32)), (ncls, std)
    # three-dimensional input as detect_objects sees it (N, P, ncls), softmax over dim 2
    x = (rs.randn(2, 9344, 2) * 2.5).astype(np.float32)
    out = np.empty((2 * 9344, 1), np.float32)
    lib.run(x.ctypes.data_as(ctypes.c_void_p), 2 * 9344, 2, out.ctypes.data_as(ctypes.c_void_p))
    ref = torch.softmax(torch.from_numpy(x), dim=2).numpy()[..., 1].reshape(-1, 1)
    assert np.array_equal(out.view(np.uint32), np.ascontiguousarray(ref).view(np.uint32))


def test_stop_event_fusion_of_launch_programs():
    """_lib._fuse_stop_events: an event record whose predecessor on the same stream launched k >= 1 kernels becomes an
    msl_arm_stop_event(ev, k - 1) in front of that entry; never across a hook, a wait, a launch-free entry, a timed launch,
    or when a second record follows the same launch; the wait of the other lane then refers to the launch entry."""
    from mslesions3d_amd import _lib

    def fn(name):
        f = lambda *a: 0
        f.__name__ = name
        return f

    rec, wait = fn("msl_event_record"), fn("msl_stream_wait_event")
    k1, k2, k0 = fn("msl_pwconv_fwd"), fn("msl_head_conv_fwd"), fn("msl_fill_u32")
    A, B = 0x1000, 0x2000  # streams
    prog = _lib.Program()
    entries = [
        ((k1, (1, 2, A), "a"), 1),        # 0: fused with record of ev 11
        ((k1, (1, 2, B), "b"), 1),        # 1: other stream in between: does not matter
        ((rec, (11, A), "event"), 0),     # -> arm(11, 0) in front of entry 0
        ((wait, (B, 11), "event"), 0),
        ((k2, (3, 4, A), "c"), 2),        # two launches: arm(12, 1)
        ((rec, (12, A), "event"), 0),
        ((rec, (13, A), "event"), 0),     # second record behind the same launch: stays a record
        ((k0, (5, 6, A), "d"), 0),        # launches nothing (a memset): the record behind it stays
        ((rec, (14, A), "event"), 0),
        ((k1, (1, 2, A), "timed"), 1),    # timed launch: stays
        ((rec, (15, A), "event"), 0),
        ((k1, (1, 2, A), "e"), 1),
        ((None, (lambda: None), "hook"), 0),
        ((rec, (16, A), "event"), 0),     # across a hook: stays
        ((k1, (1, 2, A), "f"), 1),
        ((wait, (A, 11), "event"), 0),    # a wait between launch and record: stays
        ((rec, (17, A), "event"), 0),
    ]
    for (f, a, t), n in entries:
        prog.append((f, a, t) if f is not None else (None, a, t))
        prog.nl.append(n)
    flat = _lib._fuse_stop_events(prog, ("timed",))
    names = [(e[0].__name__ if e[0] is not None else "hook", e[1] if e[0] is not None and e[0].__name__ in ("msl_arm_stop_event", "msl_event_record") else None) for e in flat]
    arms = [a for nme, a in names if nme == "msl_arm_stop_event"]
    recs = [a[0] for nme, a in names if nme == "msl_event_record"]
    assert arms == [(11, 0), (12, 1)]
    assert recs == [13, 14, 15, 16, 17]
    assert names[0][0] == "msl_arm_stop_event" and names[1][0] == "msl_pwconv_fwd"          # arm sits right in front of its launch
    i12 = names.index(("msl_arm_stop_event", (12, 1)))
    assert names[i12 + 1][0] == "msl_head_conv_fwd"
    assert flat[0][3] == A and flat[i12][3] == A                                              # the arm runs on the launch's lane
    saved = _lib.STOP_EVENT_FORKS
    try:
        _lib.STOP_EVENT_FORKS = False
        assert [e[0].__name__ for e in _lib._fuse_stop_events(prog, ()) if e[0] is not None] == [f.__name__ for (f, _, _), _ in entries if f is not None]
    finally:
        _lib.STOP_EVENT_FORKS = saved
    plain = list(prog)  # a program without launch counts (not from the recorder): never fused
    assert not any(e[0] is not None and e[0].__name__ == "msl_arm_stop_event" for e in _lib._fuse_stop_events(plain, ()))
