"""Mint the golden vectors by running the REFERENCE's own code (build container only).

    cd /root/repo && python -m tests.golden.make_golden

Imports ``lesions3d/{ssd3d,mobilenet,utils}.py`` from /root/reference through
``_ref_loader`` (Lightning/MONAI/wandb replaced by inert stand-ins; the arithmetic — stock torch
2.10 CPU ops — is untouched) and records inputs->outputs for every hot-path row of SURVEY.md §8(a).
Only numbers are written (``tests/golden/*.npz``); no reference source travels.
"""
import hashlib
import os
import sys

import numpy as np
import torch

from . import _ref_loader, cases, detinit  # noqa

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrs.items()})


def build_ref_model(ref, n_classes=2, input_channels=1, input_size=(64, 64, 64), threshold=None, seed=1234, **kw):
    torch.manual_seed(0)
    threshold = [0.1, 0.2] if threshold is None else threshold
    m = ref.LSSD3D(n_classes=n_classes, input_channels=input_channels, input_size=input_size,
                   threshold=threshold, **kw)
    m.load_state_dict(detinit.fill_state_dict(m.state_dict(), seed))
    return m


def capture_loss_locals(loss_fn, *args):
    """Run MultiBoxLoss.forward and grab its local ``true_classes`` / ``true_locs`` at return."""
    grabbed = {}

    def tracer(frame, event, arg):
        if frame.f_code.co_name == "forward" and "true_classes" in frame.f_code.co_varnames:
            def local(frame, event, arg):
                if event == "return":
                    grabbed["true_classes"] = frame.f_locals["true_classes"].clone()
                    grabbed["true_locs"] = frame.f_locals["true_locs"].clone()
                return local
            return local
        return None

    sys.settrace(tracer)
    try:
        out = loss_fn(*args)
    finally:
        sys.settrace(None)
    return out, grabbed


def gen_priors(ref):
    out = {}
    for tag, size in (("64", (64, 64, 64)), ("128", (128, 128, 128)), ("192", (192, 192, 192)), ("48x64x64", (48, 64, 64))):
        torch.manual_seed(0)
        m = ref.LSSD3D(n_classes=2, input_channels=1, input_size=size, threshold=[0.1, 0.2])
        p = m.priors_cxcycz.contiguous().numpy()
        out[f"n_{tag}"] = np.int64(p.shape[0])
        out[f"sha256_{tag}"] = np.frombuffer(hashlib.sha256(p.tobytes()).digest(), dtype=np.uint8)
        out[f"head_{tag}"] = p[:4]
        out[f"tail_{tag}"] = p[-4:]
        out[f"stride97_{tag}"] = p[::97]
        out[f"scales_{tag}"] = np.array([m.scales[k] for k in (3, 5, 7)], dtype=np.float64)
        fm, ch = m.base.get_feature_map_infos(size, "cpu")
        out[f"fmap_dims_{tag}"] = np.array([fm[i] for i in range(8)], dtype=np.int64)
        out[f"fmap_chans_{tag}"] = np.array(ch, dtype=np.int64)
        if tag == "64":
            out["full_64"] = p
        # state-dict inventory
        if tag == "64":
            sd = m.state_dict()
            out["sd_keys"] = np.array(list(sd.keys()))
            out["sd_numel"] = np.array([v.numel() for v in sd.values()], dtype=np.int64)
    save("priors.npz", **out)


def gen_boxmath(ref_utils):
    a, b, g = cases.boxmath_inputs()
    ac = ref_utils.xyz_to_cxcycz(a)
    bc = ref_utils.xyz_to_cxcycz(b)
    bc_pos = bc.clone()
    bc_pos[:, 3:] = bc_pos[:, 3:].clamp(min=1e-3)
    save("boxmath.npz",
         xyz_to_cxcycz=ac.numpy(), cxcycz_to_xyz=ref_utils.cxcycz_to_xyz(ac).numpy(),
         encode=ref_utils.cxcycz_to_gcxgcygcz(ref_utils.xyz_to_cxcycz(b[2:26]), bc_pos[14:38]).numpy(),
         decode=ref_utils.gcxgcygcz_to_cxcycz(g, bc_pos).numpy(),
         intersection=ref_utils.find_intersection3d(a, b).numpy(),
         iou=ref_utils.find_jaccard_overlap3d(a, b).numpy(),
         iou_self=ref_utils.find_jaccard_overlap3d(a, a).numpy())


def gen_matching(ref):
    m = build_ref_model(ref)
    out = {}
    for name, c in cases.matching_cases().items():
        loss_fn = ref.MultiBoxLoss(m.priors_cxcycz, threshold=c["threshold"], alpha=1.0)
        locs, scores = detinit.make_head_outputs(c["head_seed"], len(c["boxes"]), cases.P_C64)
        locs.requires_grad_(True)
        scores.requires_grad_(True)
        (conf, loc), g = capture_loss_locals(loss_fn, locs, scores, c["boxes"], c["labels"])
        (conf + loc).backward()
        out[f"{name}__true_classes"] = g["true_classes"].numpy().astype(np.int8)
        out[f"{name}__true_locs"] = g["true_locs"].numpy()
        out[f"{name}__conf"] = np.float32(conf.item())
        out[f"{name}__loc"] = np.float32(loc.item())
        out[f"{name}__dlocs_nz"] = locs.grad.numpy()[g["true_classes"].numpy() > 0]
        out[f"{name}__dscores_s17"] = scores.grad.numpy().reshape(-1)[::17]
    save("matching.npz", **out)


def gen_network(ref, tag, n, c_in, size, sample_stride):
    m = build_ref_model(ref, input_channels=c_in, input_size=size)
    x = detinit.make_volume_batch(5, n, c_in, size)
    boxes, labels = detinit.make_gt(8, n, size)
    out = {}
    # eval-mode forward first (running stats from detinit)
    m.eval()
    with torch.no_grad():
        le, se = m(x)
    out["eval_locs"] = le.numpy().reshape(-1)[::sample_stride]
    out["eval_scores"] = se.numpy().reshape(-1)[::sample_stride]
    # train-mode forward + loss + backward
    m.train()
    locs, scores = m(x)
    conf, loc = m.loss_fn(locs, scores, boxes, labels)
    (conf + m.loss_fn.alpha * loc).backward()
    out["train_locs"] = locs.detach().numpy().reshape(-1)[::sample_stride]
    out["train_scores"] = scores.detach().numpy().reshape(-1)[::sample_stride]
    if tag == "c64":
        out["train_locs_full"] = locs.detach().numpy()
        out["train_scores_full"] = scores.detach().numpy()
    out["conf"] = np.float32(conf.item())
    out["loc"] = np.float32(loc.item())
    names, gnorm, ghead = [], [], []
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        gnorm.append(p.grad.double().norm().item())
        ghead.append(p.grad.reshape(-1)[:4].numpy().copy() if p.grad.numel() >= 4 else np.resize(p.grad.reshape(-1).numpy(), 4))
    out["grad_names"] = np.array(names)
    out["grad_norm"] = np.array(gnorm)
    out["grad_head4"] = np.stack(ghead)
    sd = m.state_dict()
    for k in ("base.features.0.1", "base.features.1.bn1", "base.features.4.bn2", "base.features.7.bn2"):
        out[f"rm__{k}"] = sd[k + ".running_mean"].numpy().copy()
        out[f"rv__{k}"] = sd[k + ".running_var"].numpy().copy()
        out[f"nbt__{k}"] = sd[k + ".num_batches_tracked"].numpy().copy()
    # two optimiser steps, driven the way Lightning's automatic optimisation drives the reference: the optimizer calls a
    # closure that runs the reference's OWN training_step (which steps the scheduler itself, ssd3d.py:527-529) and the
    # backward pass, then applies the update (ssd3d.py:704-722).  use_wandb=True routes its logging to the inert
    # self.log; current_epoch = 1 skips the periodic mAP branch (ssd3d.py:497).
    m2 = build_ref_model(ref, input_channels=c_in, input_size=size, lr=1e-3, use_wandb=True)
    m2.train()
    m2.current_epoch = 1
    # the reference's configure_optimizers passes `verbose=` to the scheduler, which torch 2.10 no longer
    # accepts; the same optimizer/scheduler are built here from its parameter groups
    biases = [p for k, p in m2.named_parameters() if k.endswith(".bias")]
    others = [p for k, p in m2.named_parameters() if not k.endswith(".bias")]
    opt = torch.optim.Adam([{"params": biases, "lr": 2 * m2.lr}, {"params": others}], lr=m2.lr, weight_decay=0.0005)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=40)
    m2.lr_schedulers = lambda: sch
    losses = []
    for step in range(2):
        xs = detinit.make_volume_batch(50 + step, n, c_in, size)
        bs, ls = detinit.make_gt(60 + step, n, size)
        batch = {"img": xs, "boxes": bs, "labels": ls, "subject": [str(i) for i in range(n)]}

        def closure():
            opt.zero_grad(set_to_none=True)
            res = m2.training_step(batch)
            res["loss"].backward()
            losses.append([res["loss"].item(), res["log"]["train_conf_loss"].item(), res["log"]["train_loc_loss"].item()])
            return res["loss"]

        opt.step(closure)
    out["adam_losses"] = np.array(losses, dtype=np.float64)
    out["adam_lr"] = np.array(sch.get_last_lr(), dtype=np.float64)
    pn, ph = [], []
    for k, p in m2.named_parameters():
        pn.append(p.detach().double().norm().item())
        ph.append(np.resize(p.detach().reshape(-1)[:4].numpy(), 4))
    out["adam_param_norm"] = np.array(pn)
    out["adam_param_head4"] = np.stack(ph)
    out["adam_param_names"] = np.array([k for k, _ in m2.named_parameters()])
    save(f"network_{tag}.npz", **out)


class _stable_sort:
    """``Tensor.sort(descending=True)`` without ``stable=`` leaves the order of EXACTLY equal scores
    unspecified (torch's CPU introsort is not stable beyond 16 elements), so the reference's keep-list on
    tied scores is implementation-defined.  For the one fixture built from exact ties (``quant_ties``) the
    reference is run with the sort forced stable — one of its valid outcomes, and the rule the oracle and
    the HIP kernels implement (ties by ascending prior index)."""

    def __enter__(self):
        self.orig = torch.Tensor.sort
        orig = self.orig
        torch.Tensor.sort = lambda t, *a, **k: orig(t, *a, **dict(k, stable=True))

    def __exit__(self, *exc):
        torch.Tensor.sort = self.orig


class _nullctx:
    def __enter__(self):
        pass

    def __exit__(self, *exc):
        pass


def gen_detect(ref):
    m = build_ref_model(ref)
    out = {}
    for name, c in cases.detect_cases().items():
        locs, scores = cases.detect_inputs(c)
        with torch.no_grad(), (_stable_sort() if c["quantized"] else _nullctx()):
            b, l, s = m.detect_objects(locs, scores, c["min_score"], c["max_overlap"], c["top_k"])
        for i in range(c["n"]):
            out[f"{name}__boxes_{i}"] = b[i].numpy()
            out[f"{name}__labels_{i}"] = l[i].numpy()
            out[f"{name}__scores_{i}"] = s[i].numpy()
    save("detect.npz", **out)


def gen_map(ref_utils):
    out = {}
    for name, c in cases.map_cases().items():
        t = lambda xs, dt: [torch.from_numpy(np.asarray(x)).to(dt) for x in xs]
        for ov in (0.1, 0.5):
            d = ref_utils.calculate_mAP(t(c["det_boxes"], torch.float32), t(c["det_labels"], torch.long),
                                        t(c["det_scores"], torch.float32), t(c["true_boxes"], torch.float32),
                                        t(c["true_labels"], torch.long),
                                        [torch.zeros(len(x), dtype=torch.bool) for x in c["true_labels"]],
                                        min_overlap=ov, return_detail=True)
            tag = f"{name}__{ov}"
            for k in ("APs", "mAP", "precision", "recall", "f1_score", "n_true_boxes"):
                out[f"{tag}__{k}"] = np.float64(float(d[k]))
            for k in ("TP", "FP", "found_boxes_volumes_per_class", "not_found_boxes_volumes_per_class"):
                out[f"{tag}__{k}"] = np.asarray(d[k].numpy(), dtype=np.float32)
    save("map.npz", **out)


def gen_multiclass(ref):
    """n_classes = 3 (background + two lesion classes): head width (ssd3d.py:132), label gather and per-class confidence loss
    (ssd3d.py:871-933), the class loop of detect_objects (ssd3d.py:384) and its cross-class top-k re-sort (:449-453)."""
    size, n = cases.SIZE_C64, 2
    m = build_ref_model(ref, n_classes=3, input_size=size)
    x = detinit.make_volume_batch(5, n, 1, size)
    boxes, labels = cases.multiclass_gt(8, n, size)
    out = {"gt_labels_0": labels[0].numpy(), "gt_labels_1": labels[1].numpy()}
    m.eval()
    with torch.no_grad():
        le, se = m(x)
        out["eval_locs"], out["eval_scores"] = le.numpy(), se.numpy()
        b, l, s = m.detect_objects(le, se, 0.34, 0.5, 20)  # the untrained net's softmax sits near 1/3
        for i in range(n):
            out[f"e2e__boxes_{i}"], out[f"e2e__labels_{i}"], out[f"e2e__scores_{i}"] = b[i].numpy(), l[i].numpy(), s[i].numpy()
    m.train()
    locs, scores = m(x)
    (conf, loc), g = capture_loss_locals(m.loss_fn, locs, scores, boxes, labels)
    (conf + m.loss_fn.alpha * loc).backward()
    out["train_locs"], out["train_scores"] = locs.detach().numpy(), scores.detach().numpy()
    out["true_classes"] = g["true_classes"].numpy().astype(np.int8)
    out["true_locs"] = g["true_locs"].numpy()
    out["conf"], out["loc"] = np.float32(conf.item()), np.float32(loc.item())
    names, gnorm, ghead = [], [], []
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        gnorm.append(p.grad.double().norm().item())
        ghead.append(np.resize(p.grad.reshape(-1)[:4].numpy(), 4))
    out["grad_names"], out["grad_norm"], out["grad_head4"] = np.array(names), np.array(gnorm), np.stack(ghead)
    # loss on fixed head outputs (dL/dlocs, dL/dscores of the reference's autograd)
    hl, hs = detinit.make_head_outputs(81, n, cases.P_C64, n_classes=3)
    hl.requires_grad_(True)
    hs.requires_grad_(True)
    (c2, l2), g2 = capture_loss_locals(ref.MultiBoxLoss(m.priors_cxcycz, threshold=[0.1, 0.2], alpha=1.0), hl, hs, boxes, labels)
    (c2 + l2).backward()
    out["heads__conf"], out["heads__loc"] = np.float32(c2.item()), np.float32(l2.item())
    out["heads__true_classes"] = g2["true_classes"].numpy().astype(np.int8)
    out["heads__dlocs_nz"] = hl.grad.numpy()[g2["true_classes"].numpy() > 0]
    out["heads__dscores"] = hs.grad.numpy()
    for name, c in cases.multiclass_detect_cases().items():
        dl, ds = cases.multiclass_detect_inputs(c)
        with torch.no_grad():
            b, l, s = m.detect_objects(dl, ds, c["min_score"], c["max_overlap"], c["top_k"])
        for i in range(c["n"]):
            out[f"{name}__boxes_{i}"], out[f"{name}__labels_{i}"], out[f"{name}__scores_{i}"] = b[i].numpy(), l[i].numpy(), s[i].numpy()
    save("multiclass.npz", **out)


def gen_fivescale(ref):
    """`--prediction_layers "1 2 3 5 7"` (train.py:131 -> aspect_ratios keys, ssd3d.py:204): five prediction scales."""
    size, n = cases.SIZE_C64, 2
    ar = {l: [1.] for l in cases.FIVE_SCALES}
    m = build_ref_model(ref, input_size=size, aspect_ratios=ar)
    x = detinit.make_volume_batch(5, n, 1, size)
    boxes, labels = detinit.make_gt(8, n, size)
    out = {"priors_n": np.int64(m.priors_cxcycz.shape[0]),
           "priors_sha256": np.frombuffer(hashlib.sha256(m.priors_cxcycz.contiguous().numpy().tobytes()).digest(), dtype=np.uint8),
           "scales": np.array([m.scales[k] for k in cases.FIVE_SCALES], dtype=np.float64),
           "sd_keys": np.array(list(m.state_dict().keys()))}
    m.eval()
    with torch.no_grad():
        le, se = m(x)
        out["eval_locs"], out["eval_scores"] = le.numpy().reshape(-1)[::7], se.numpy().reshape(-1)[::7]
        b, l, s = m.detect_objects(le, se, 0.5, 0.5, 30)
        for i in range(n):
            out[f"e2e__labels_{i}"], out[f"e2e__scores_{i}"] = l[i].numpy(), s[i].numpy()
    m.train()
    locs, scores = m(x)
    conf, loc = m.loss_fn(locs, scores, boxes, labels)
    (conf + m.loss_fn.alpha * loc).backward()
    out["train_locs"], out["train_scores"] = locs.detach().numpy().reshape(-1)[::7], scores.detach().numpy().reshape(-1)[::7]
    out["conf"], out["loc"] = np.float32(conf.item()), np.float32(loc.item())
    names, gnorm = [], []
    for k, p in m.named_parameters():
        if p.grad is not None:
            names.append(k)
            gnorm.append(p.grad.double().norm().item())
    out["grad_names"], out["grad_norm"] = np.array(names), np.array(gnorm)
    save("fivescale.npz", **out)


def gen_signatures(ref_ssd3d, ref_mobilenet, ref_utils):
    """The drop-in boundary as data: names, parameter order, kinds and defaults of every public class /
    function on the path, read off the reference's own objects with ``inspect.signature`` -> tests/golden/signatures.json."""
    import inspect
    import json
    import base_network as ref_base  # on sys.path through _ref_loader

    def sig(fn):
        rows = []
        for prm in inspect.signature(fn).parameters.values():
            d = prm.default
            if d is inspect.Parameter.empty:
                d = "<required>"
            elif not isinstance(d, (int, float, str, bool, type(None), list, dict, tuple)):
                d = repr(d)
            rows.append([prm.name, prm.kind.name, list(d) if isinstance(d, tuple) else d])
        return rows

    table = {}
    classes = {"ssd3d": (ref_ssd3d, ["MobileNetBase", "PredictionConvolutions", "LSSD3D", "MultiBoxLoss"]),
               "mobilenet": (ref_mobilenet, ["Block"]),
               "base_network": (ref_base, ["ConvNetBase"])}
    methods = {"MobileNetBase": ["__init__", "init", "forward", "get_feature_map_infos"],
               "PredictionConvolutions": ["__init__", "init", "forward"],
               "LSSD3D": ["__init__", "forward", "init", "create_prior_boxes", "detect_objects", "training_step", "validation_step",
                          "predict_step", "configure_optimizers"],
               "MultiBoxLoss": ["__init__", "forward"], "Block": ["__init__", "forward"],
               "ConvNetBase": ["__init__", "init", "forward", "get_feature_map_infos"]}
    for mod, (module, names) in classes.items():
        for cname in names:
            cls = getattr(module, cname)
            for meth in methods[cname]:
                table[f"{mod}.{cname}.{meth}"] = sig(getattr(cls, meth))
    functions = {"mobilenet": (ref_mobilenet, ["conv_bn"]), "base_network": (ref_base, ["get_n_params"]),
                 "utils": (ref_utils, ["cxcycz_to_xyz", "gcxgcygcz_to_cxcycz", "cxcycz_to_gcxgcygcz", "xyz_to_cxcycz", "find_intersection3d",
                                       "find_jaccard_overlap3d", "volume", "compute_metrics_per_class", "calculate_mAP"])}
    for mod, (module, names) in functions.items():
        for fname in names:
            table[f"{mod}.{fname}"] = sig(getattr(module, fname))
    consts = {"mobilenet.MOBILENET_CONFIGS": ref_mobilenet.MOBILENET_CONFIGS, "base_network.CONVNET_CONFIGS": ref_base.CONVNET_CONFIGS,
              "ssd3d.ASPECT_RATIOS": getattr(ref_ssd3d, "ASPECT_RATIOS", None)}
    path = os.path.join(OUT, "signatures.json")
    with open(path, "w") as f:
        json.dump({"signatures": table, "constants": json.loads(json.dumps(consts, default=list))}, f, indent=1, sort_keys=True)
    print("wrote signatures.json", len(table), "callables")


def main():
    ref_ssd3d, ref_mobilenet, ref_utils = _ref_loader.load_reference()
    only = sys.argv[1:]
    jobs = [("priors", lambda: gen_priors(ref_ssd3d)), ("boxmath", lambda: gen_boxmath(ref_utils)),
            ("matching", lambda: gen_matching(ref_ssd3d)),
            ("network_c64", lambda: gen_network(ref_ssd3d, "c64", 2, 1, (64, 64, 64), 7)),
            ("network_a128", lambda: gen_network(ref_ssd3d, "a128", 4, 1, (128, 128, 128), 61)),
            ("network_a2_2ch64", lambda: gen_network(ref_ssd3d, "a2_2ch64", 2, 2, (64, 64, 64), 7)),
            # BASELINE configs[4]: multi-modal 2-channel (T1 + FLAIR) 128^3, batch 4
            ("network_a2_2ch128", lambda: gen_network(ref_ssd3d, "a2_2ch128", 4, 2, (128, 128, 128), 61)),
            ("network_noncube", lambda: gen_network(ref_ssd3d, "noncube", 2, 1, (48, 64, 64), 7)),
            ("detect", lambda: gen_detect(ref_ssd3d)), ("map", lambda: gen_map(ref_utils)),
            ("multiclass", lambda: gen_multiclass(ref_ssd3d)), ("fivescale", lambda: gen_fivescale(ref_ssd3d)),
            ("signatures", lambda: gen_signatures(ref_ssd3d, ref_mobilenet, ref_utils))]
    for name, job in jobs:
        if not only or name in only:
            job()


if __name__ == "__main__":
    main()
