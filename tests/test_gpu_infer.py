"""BASELINE configs[3] as a checked path: 192^3 volumes, batch 2, the predict.py flow (eval forward -> detect_objects with the 3-D
NMS kernel -> calculate_mAP at IoU 0.1 / 0.5; reference ssd3d.py:344-460,692-702, predict.py:87-112) scored on synthetic cases:
the HIP fp32 path must give the oracle's detections and hence its mAP exactly; the bf16 activation path (a build-side
extension) within a stated margin."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import detect as OD
from oracle import metrics as OM
from oracle.network import OracleSSD3D
from tests.golden import detinit

pytestmark = pytest.mark.gpu
DEV = "cuda"
SIZE, BATCH, CASES = (192, 192, 192), 2, 8
KW = dict(min_score=0.3, max_overlap=0.3, top_k=50)
BF16_MAP_MARGIN = 0.10   # |mAP(bf16) - mAP(fp32 oracle)| at either IoU, absolute (printed by the test)


def _cases():
    from mslesions3d_amd.synth import make_case
    imgs, boxes, labels = [], [], []
    for i in range(CASES):
        img, _, b, l = make_case(700 + i, SIZE)
        imgs.append(torch.from_numpy(img)[None])
        boxes.append(torch.from_numpy(b))
        labels.append(torch.from_numpy(l))
    return torch.stack(imgs), boxes, labels


def _model(weights):
    from mslesions3d_amd.ssd3d import LSSD3D
    from mslesions3d_amd.synth import make_batch_on_device
    from mslesions3d_amd.trainer import FusedTrainer
    m = LSSD3D(n_classes=2, input_channels=1, input_size=SIZE, threshold=[0.1, 0.2], lr=1e-3)
    m.load_state_dict(detinit.fill_state_dict(m.state_dict(), 1234))
    m = m.to(DEV)
    if weights == "trained":  # a short run of the fused step, so that scores and running statistics mean something
        m.train()
        tr = FusedTrainer(m)
        for s in range(3):  # (a longer run of the live loss drives every prior to background: no detections left to compare)
            x, b, l = make_batch_on_device(BATCH, SIZE, torch.device(DEV), 1, seed=4000 + s % 32)
            tr.step(x, b, l, sync=False)
        torch.cuda.synchronize()
    return m.eval()


def _map(det, gt_b, gt_l, fn, oracle=False):
    dif = [torch.zeros(len(l), dtype=torch.bool) for l in gt_l]
    out = {}
    for iou in (0.1, 0.5):
        if oracle:  # oracle/metrics.py works on numpy arrays and always returns the detail dict
            npy = lambda xs: [np.asarray(x) for x in xs]
            d = fn(npy(det[0]), npy(det[1]), npy(det[2]), npy(gt_b), npy(gt_l), npy(dif), min_overlap=iou)
        else:
            d = fn(det[0], det[1], det[2], gt_b, gt_l, dif, min_overlap=iou, return_detail=True)
        out[iou] = {k: float(d[k]) for k in ("mAP", "precision", "recall", "f1_score")}
    return out


@pytest.mark.parametrize("weights", ["detinit", "trained"])  # deterministic init / + 3 fused optimisation steps
def test_map_on_synthetic_192_equals_the_oracle_and_bf16_is_close(weights):
    from mslesions3d_amd.utils import calculate_mAP
    x, gt_b, gt_l = _cases()
    m = _model(weights)
    om = OracleSSD3D(2, 1, SIZE, emulate_reference_init=False)
    om.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    om.eval()
    m.min_score, m.max_overlap, m.top_k = KW["min_score"], KW["max_overlap"], KW["top_k"]
    hip = {"f32": ([], [], [], []), "bf16": ([], [], [], [])}
    orc = ([], [], [], [])
    for c in range(0, CASES, BATCH):
        xb = x[c:c + BATCH]
        with torch.no_grad():
            ol, osc = om(xb)
        od = OD.detect_objects(ol, osc, om.priors_cxcycz, KW["min_score"], KW["max_overlap"], KW["top_k"], return_prior_index=True)
        for k in range(4):
            orc[k].extend(od[k])
        for dt in ("f32", "bf16"):
            m.compute_dtype = dt
            with torch.no_grad():
                locs, scores = m(xb.to(DEV))
                d = m.detect_objects(locs, scores, return_prior_index=True, **KW)
            for k in range(4):
                hip[dt][k].extend(t.cpu() for t in d[k])
    # fp32: the detections ARE the oracle's (keep-lists bit-exact, scores bit-equal given equal logits to 1e-4 ...)
    same = sum(int(torch.equal(a, b)) for a, b in zip(hip["f32"][3], orc[3]))
    print(f"[{weights}] fp32 keep-lists identical to the oracle's: {same} of {CASES}")
    assert same == CASES
    # An untrained / briefly trained network hits none of the generator's cubes (mAP 0 on both sides: a trivial equality), so the
    # metric is ALSO scored against a pseudo ground truth that some detections do hit: per image, the oracle's detections of
    # rank 2, 9 and 24 jittered by 2 % of their extent, plus one box nothing overlaps - true and false positives, ranks, ties.
    rs = np.random.RandomState(5)
    pseudo_b, pseudo_l = [], []
    for i in range(CASES):
        ob = orc[0][i]
        pick = [ob[k] for k in sorted({min(k, len(ob) - 1) for k in (2, 9, 24)})]
        boxes = []
        for b in pick:
            ext = (b[3:] - b[:3]).abs()
            jit = torch.from_numpy(rs.uniform(-0.02, 0.02, 6).astype(np.float32)) * torch.cat([ext, ext])
            boxes.append(b + jit)
        boxes.append(torch.tensor([0.90, 0.90, 0.90, 0.97, 0.97, 0.97]))
        pseudo_b.append(torch.stack(boxes))
        pseudo_l.append(torch.ones(len(boxes), dtype=torch.long))
    ref_p = _map(orc, pseudo_b, pseudo_l, OM.calculate_map, oracle=True)
    got_p = _map(hip["f32"], pseudo_b, pseudo_l, calculate_mAP)
    b16_p = _map(hip["bf16"], pseudo_b, pseudo_l, calculate_mAP)
    assert ref_p[0.5]["mAP"] > 0.05 and ref_p[0.1]["recall"] > 0.3, ref_p   # the pseudo ground truth is really being hit
    for iou in (0.1, 0.5):
        for k in ("mAP", "precision", "recall", "f1_score"):
            assert got_p[iou][k] == pytest.approx(ref_p[iou][k], rel=1e-6, abs=1e-9, nan_ok=True), (iou, k, got_p[iou], ref_p[iou])
        assert abs(b16_p[iou]["mAP"] - ref_p[iou]["mAP"]) <= BF16_MAP_MARGIN, (iou, b16_p[iou], ref_p[iou])
    print(f"[{weights}] pseudo ground truth mAP@0.1 / @0.5: oracle {ref_p[0.1]['mAP']:.4f} / {ref_p[0.5]['mAP']:.4f}, HIP fp32 "
          f"{got_p[0.1]['mAP']:.4f} / {got_p[0.5]['mAP']:.4f}, HIP bf16 {b16_p[0.1]['mAP']:.4f} / {b16_p[0.5]['mAP']:.4f}")
    ref = _map(orc, gt_b, gt_l, OM.calculate_map, oracle=True)
    got = _map(hip["f32"], gt_b, gt_l, calculate_mAP)
    for iou in (0.1, 0.5):
        for k in ("mAP", "precision", "recall", "f1_score"):
            assert got[iou][k] == pytest.approx(ref[iou][k], rel=1e-6, abs=1e-9, nan_ok=True), (iou, k, got[iou], ref[iou])
    b16 = _map(hip["bf16"], gt_b, gt_l, calculate_mAP)
    same16 = sum(int(a.shape == b.shape and torch.equal(a, b)) for a, b in zip(hip["bf16"][3], orc[3]))
    print(f"[{weights}] mAP@0.1 / @0.5: oracle {ref[0.1]['mAP']:.4f} / {ref[0.5]['mAP']:.4f}, HIP fp32 {got[0.1]['mAP']:.4f} / "
          f"{got[0.5]['mAP']:.4f}, HIP bf16 {b16[0.1]['mAP']:.4f} / {b16[0.5]['mAP']:.4f}; bf16 keep-lists identical: {same16} of {CASES}")
    for iou in (0.1, 0.5):
        assert abs(b16[iou]["mAP"] - ref[iou]["mAP"]) <= BF16_MAP_MARGIN, (iou, b16[iou], ref[iou])
        assert abs(b16[iou]["recall"] - ref[iou]["recall"]) <= 0.15, (iou, b16[iou], ref[iou])


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_predict_batches_pipeline_equals_predict_step(dtype):
    """LSSD3D.predict_batches (what predict.py and `bench.py --mode infer` loop over): with 1, 2 or 3 batches in flight the
    results - per image boxes, labels, scores, in order - are exactly those of predict_step batch by batch, on batches that
    differ (the pinned landing zones and the staging buffer are reused across passes) and for host and device inputs."""
    from mslesions3d_amd.synth import make_batch_on_device
    m = _model("trained")
    m.compute_dtype = dtype
    m.min_score, m.max_overlap, m.top_k = 0.05, KW["max_overlap"], KW["top_k"]
    xs = [make_batch_on_device(BATCH, SIZE, torch.device(DEV), 1, seed=8100 + k)[0] for k in range(5)]
    ref = [m.predict_step({"img": x}, k) for k, x in enumerate(xs)]
    assert sum(len(b) for r in ref for b in r[0]) > 0, "no detections: nothing compared"
    assert len({tuple(len(b) for b in r[0]) for r in ref}) > 1 or not all(torch.equal(ref[0][2][0], r[2][0]) for r in ref[1:]), \
        "the batches must differ"
    for depth, host in [(1, False), (2, False), (3, False), (2, True), (8, False)]:
        batches = [{"img": x.cpu() if host else x} for x in xs]
        got = list(m.predict_batches(iter(batches), depth=depth))
        assert len(got) == len(ref)
        for r, g in zip(ref, got):
            for u, v in zip(r, g):
                assert len(u) == len(v) == BATCH
                for a, b in zip(u, v):
                    assert torch.equal(a, b), (depth, host)
    assert list(m.predict_batches(iter([]))) == []
    m.train()  # not in eval mode: falls back to predict_step batch by batch (which runs the eager path)
    m.eval()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_bench_infer_line(dtype):
    """`python bench.py --mode infer [--dtype bf16]`: ONE JSON line for configs[3] - volumes/s, kept boxes/s, mAP on synthetic
    cases, the roofline of its longest launch and the oracle's predict path as cpu_baseline (fp32: keep-lists equal)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", "infer", "--dtype", dtype, "--steps", "10", "--warmup", "3",
                        "--train-steps", "40", "--cpu-steps", "2", "--map-cases", "4"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["unit"] == "volumes/s" and d["n_gpus"] == 1 and d["dtype"] == dtype and d["steps"] == 10 and d["value"] > 0
    assert "192^3" in d["config"]["workload"] and d["config"]["priors"] == 31536 and d["config"]["global_batch"] == 2
    assert abs(d["value"] - 2 * 10 / (d["ms_per_step"] * 10 * 1e-3)) <= 0.01 * d["value"] and d["boxes_per_s"] >= 0
    assert set(d["mAP_synthetic"]["IoU"]) == {"0.1", "0.5"}
    rf = d["roofline"]
    # eval mode runs stem + block-1 depthwise as one launch that is bound by the fp32 matrix pipe, not by HBM
    assert rf["bound"] == "mfma" and rf["peak"] == 157.3 and rf["unit"] == "TFLOP/s" and 0.02 < rf["frac"] < 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["launches_timed"] == 5  # (the event pair rides on every 2nd pass)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "volumes/s" and cb["value"] > 0
    if dtype == "f32":
        assert cb["keep_lists_equal_oracle"] is True
