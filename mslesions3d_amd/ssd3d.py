"""3D SSD lesion detector on MI355X — host-side mirror of the reference's ``lesions3d/ssd3d.py``.

Same classes, constructor arguments, attributes, method names, return types, ``state_dict`` keys and error
behaviour as the reference; the arithmetic runs in the hand-written HIP kernels of
``mslesions3d_amd/csrc`` through the C ABI of ``include/mslesions3d_hip.h``.  PyTorch is used for device memory,
streams, autograd bookkeeping and (in ``parallel.py``) RCCL — not for arithmetic.  There is no CPU fallback.

Lightning is not required: ``LSSD3D`` is a plain ``nn.Module`` that keeps the LightningModule surface the
reference's scripts use (``training_step`` / ``validation_step`` / ``predict_step`` / ``configure_optimizers`` /
``load_from_checkpoint`` / ``log``), and ``mslesions3d_amd.trainer`` provides the minimal loop.
"""
import math
import os
import warnings

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import ptr
from .base_network import CONVNET_CONFIGS, ConvNetBase  # noqa: F401  (re-exported like the reference)
from .engine import Engine
from .mobilenet import MOBILENET_CONFIGS, Block, conv_bn
from .utils import *  # noqa: F401,F403  (the reference does `from utils import *`, ssd3d.py:14)
from .utils import calculate_mAP

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")

ASPECT_RATIOS = {3: [1.], 5: [1.], 7: [1]}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _conv_out(d, s):
    return (d - 1) // s + 1


class MobileNetBase(nn.Module):
    """ssd3d.py:47-110."""

    def __init__(self, config="mobilenet", in_channels=1, width_mult=1., cube=False, aspect_ratios=None):
        super(MobileNetBase, self).__init__()
        if aspect_ratios is None:
            aspect_ratios = ASPECT_RATIOS
        self.aspect_ratios = aspect_ratios
        self.in_channels = in_channels
        self.config = MOBILENET_CONFIGS[config]
        input_channel = int(self.config[0] * width_mult)
        cfg = self.config[1:]
        first_stride = (1, 2, 2) if not cube else (2, 2, 2)  # ssd3d.py:60
        features = [conv_bn(in_channels, input_channel, first_stride)]
        for c, n, s in cfg:  # ssd3d.py:65-75 (truncated after the last feature that feeds a head)
            if len(features) - 1 == max(self.aspect_ratios.keys()):
                break
            output_channel = int(c * width_mult)
            for i in range(n):
                if len(features) - 1 == max(aspect_ratios.keys()):
                    break
                stride = s if i == 0 else 1
                features.append(Block(input_channel, output_channel, stride))
                input_channel = output_channel
        self.features = nn.Sequential(*features)

    def init(self):
        """ssd3d.py:80-84 — iterates ``children()`` (an ``nn.Sequential``), never a Conv3d: a no-op in the
        reference, kept a no-op here."""
        for c in self.children():
            if isinstance(c, nn.Conv3d):
                nn.init.kaiming_uniform_(c.weight)
                nn.init.constant_(c.bias, 0.)

    def forward(self, image):
        """Stand-alone backbone pass -> {feature index: activation}.  (Inside LSSD3D the fused executor is used.)"""
        out = image
        out_features = {}
        for i, feat in enumerate(self.features):
            out = feat(out)
            if i in self.aspect_ratios:
                out_features[i] = out
                if torch.isnan(out).sum() > 0:  # ssd3d.py:95-98
                    raise Exception("Yesssss this NaN error again in the base network")
        return out_features

    def feature_map_dims(self, input_size):
        dims, chans, cur = {}, [], tuple(input_size)
        for i, layer in enumerate(self.features):
            if i == 0:
                stride, c = tuple(layer[0].stride), layer[0].out_channels
            else:
                stride, c = tuple(layer.conv1.stride), layer.conv2.out_channels
            cur = tuple(_conv_out(d, s) for d, s in zip(cur, stride))
            dims[i] = cur
            chans.append(c)
        return dims, chans

    def get_feature_map_infos(self, input_size, device):
        """ssd3d.py:102-110.  The reference measures the shapes with a train-mode dummy pass of ``torch.randn``;
        the shapes are computed in closed form here, and the same amount of RNG is drawn so that seeded
        construction yields the same weights as the reference (the heads are initialised after the first draw).
        The dummy pass's side effect on the BatchNorm running statistics is reproduced by
        ``LSSD3D.replay_reference_init_side_effects()`` when a GPU is present."""
        torch.randn((1, self.in_channels, *input_size))
        return self.feature_map_dims(input_size)


class PredictionConvolutions(nn.Module):
    """ssd3d.py:113-169."""

    def __init__(self, n_classes, width_mult, aspect_ratios, features_n_channels, boxes_per_location=2):
        super(PredictionConvolutions, self).__init__()
        self.n_classes = n_classes
        n_boxes = {feat: len(aspect_ratios[feat]) + boxes_per_location - 1 for feat in aspect_ratios}
        if any(v != 2 for v in n_boxes.values()):
            raise NotImplementedError("the HIP head kernel is built for 2 priors per location "
                                      "(the reference hard-codes boxes_per_location = 2, ssd3d.py:213)")
        self.feature_ids = list(aspect_ratios.keys())
        loc_convs, cl_convs = [], []
        for f in aspect_ratios:
            f_n_channels = int(features_n_channels[f] * width_mult)
            loc_convs.append(nn.Conv3d(f_n_channels, n_boxes[f] * 6, kernel_size=3, padding=1))
            cl_convs.append(nn.Conv3d(f_n_channels, n_boxes[f] * n_classes, kernel_size=3, padding=1))
        self.loc_convs = nn.ModuleList(loc_convs)
        self.cl_convs = nn.ModuleList(cl_convs)

    def init(self):
        """ssd3d.py:137-141 — a no-op for the same reason as MobileNetBase.init."""
        for c in self.children():
            if isinstance(c, nn.Conv3d):
                nn.init.kaiming_uniform_(c.weight)
                nn.init.constant_(c.bias, 0.)

    def forward(self, feats):
        """feats: {feature index: activation (N,C,D,H,W)} -> locs (N,P,6), classes_scores (N,P,n_classes)."""
        L = _lib.load()
        keys = list(feats.keys())
        first = feats[keys[0]]
        if not first.is_cuda:
            raise _lib.HipKernelError("PredictionConvolutions runs on the HIP device only (no CPU fallback)")
        N = first.size(0)
        P = sum(2 * feats[k].shape[2] * feats[k].shape[3] * feats[k].shape[4] for k in keys)
        locs = torch.empty((N, P, 6), dtype=torch.float32, device=first.device)
        scores = torch.empty((N, P, self.n_classes), dtype=torch.float32, device=first.device)
        off = 0
        st = _stream()
        for i, key in enumerate(keys):
            f = feats[key].float()
            _, C, D, H, W = f.shape
            pad = torch.zeros((N, C, D + 2, H + 2, W + 2), dtype=torch.float32, device=f.device)
            pad[:, :, 1:-1, 1:-1, 1:-1] = f
            ne = L.msl_head_packed_weight_elems(C, self.n_classes)
            Wf = torch.empty(ne, dtype=torch.float32, device=f.device)
            Wb = torch.empty(ne, dtype=torch.float32, device=f.device)
            ws = torch.empty(max(L.msl_head_fwd_workspace_bytes(N, C, D, H, W, self.n_classes) // 4, 1),
                             dtype=torch.float32, device=f.device)
            lc, cc = self.loc_convs[i], self.cl_convs[i]
            _lib.call("msl_head_pack_weights", ptr(lc.weight), ptr(cc.weight), ptr(Wf), ptr(Wb), C, self.n_classes, st)
            _lib.call("msl_head_conv_fwd", ptr(pad), ptr(Wf), ptr(lc.bias), ptr(cc.bias), ptr(locs), ptr(scores), ptr(ws),
                      N, C, D, H, W, P, off, self.n_classes, st)
            off += 2 * D * H * W
        return locs, scores


class _SSDFunction(torch.autograd.Function):
    """Whole-network autograd node: forward = Engine.forward, backward = Engine.backward."""

    @staticmethod
    def forward(ctx, model, need_grad, image, *params):
        eng = model._engine
        locs, scores = eng.forward(image, training=model.training, need_grad=need_grad)
        pl = eng.plan_for(image, need_grad)
        ctx.model, ctx.plan, ctx.gen, ctx.names = model, pl, pl.generation, model._param_names
        return locs.clone(), scores.clone()

    @staticmethod
    def backward(ctx, dlocs, dscores):
        eng, pl = ctx.model._engine, ctx.plan
        if pl.generation != ctx.gen:
            raise RuntimeError("the activations of this forward pass were overwritten by a later forward pass "
                               "with the same shape; call backward() before running the model again")
        eng.backward(pl, dlocs, dscores)
        gv = eng.arena.grad_views
        return (None, None, None) + tuple(gv[n].clone() if n in gv else None for n in ctx.names)


class LSSD3D(nn.Module):
    """The SSD 3D network (ssd3d.py:172-738): MobileNet-3D base + prediction convolutions + priors + loss."""

    def __init__(self,
                 n_classes,
                 input_channels=3,
                 input_size=(64, 64, 64),
                 threshold=0.5,  # threshold for box matching in MultiBoxLoss
                 alpha=1.,
                 lr=1.3e-5,
                 base_network_config="mobilenet",
                 width_mult=1.,
                 min_score=0.5,
                 max_overlap=0.5,  # for box matching
                 min_overlap=0.5,  # for evaluation metrics
                 top_k=100,
                 scheduler="CosineAnnealingLR",
                 use_wandb=False,
                 batch_size=8,
                 compute_metric_every_n_epochs=1,
                 comments="",
                 aspect_ratios={},
                 min_object_size=6,
                 max_object_size=14,
                 scales={},
                 boxes_per_location=2,
                 *,
                 hard_negative_mining=False,  # the three loss variants the reference keeps as commented code
                 smooth_l1=False,             # (MultiBoxLoss docstring); default off = the reference's live loss
                 focal_loss=False,
                 ):
        super(LSSD3D, self).__init__()
        self._loss_variants = dict(hard_negative_mining=hard_negative_mining, smooth_l1=smooth_l1, focal=focal_loss)
        if aspect_ratios == {}:
            aspect_ratios = ASPECT_RATIOS
        self.hparams = dict(n_classes=n_classes, input_channels=input_channels, input_size=tuple(input_size),
                            threshold=threshold, alpha=alpha, lr=lr, base_network_config=base_network_config,
                            width_mult=width_mult, min_score=min_score, max_overlap=max_overlap,
                            min_overlap=min_overlap, top_k=top_k, scheduler=scheduler, use_wandb=use_wandb,
                            batch_size=batch_size, compute_metric_every_n_epochs=compute_metric_every_n_epochs,
                            comments=comments, aspect_ratios=aspect_ratios, min_object_size=min_object_size,
                            max_object_size=max_object_size, scales=scales, boxes_per_location=boxes_per_location)
        # only non-default variants enter the hyper-parameters: default checkpoints keep the reference's key set
        self.hparams.update({k: v for k, v in dict(hard_negative_mining=hard_negative_mining, smooth_l1=smooth_l1,
                                                   focal_loss=focal_loss).items() if v})
        self.base_network_config = base_network_config
        self.cube = input_size[0] == input_size[1] == input_size[2]
        self.input_size = tuple(input_size)
        self.input_channels = input_channels
        self.width_mult = width_mult
        self.aspect_ratios = aspect_ratios
        self.boxes_per_location = 2  # ssd3d.py:213 (the constructor argument is ignored by the reference)
        self.n_classes = n_classes
        self._make_base_and_prediction_layers()
        self.lr = lr
        self.min_score = min_score
        self.max_overlap = max_overlap
        self.min_overlap = min_overlap
        self.top_k = top_k
        self.scheduler = scheduler
        self.use_wandb = use_wandb
        self.batch_size = batch_size
        self.compute_metric_every_n_epochs = compute_metric_every_n_epochs
        self.comments = comments
        self.alpha = alpha
        self.threshold = threshold

        if scales == {}:  # ssd3d.py:228-234
            self.scales = {layer: scale for layer, scale in zip(self.aspect_ratios.keys(),
                                                                np.linspace(min_object_size / input_size[0],
                                                                            max_object_size / input_size[0],
                                                                            len(self.aspect_ratios)))}
        else:
            self.scales = scales

        features_n_channels = self.base.get_feature_map_infos(self.input_size, self.device)[1]  # ssd3d.py:238
        n_channel_rescale = int(features_n_channels[min(self.aspect_ratios.keys())] * self.width_mult)
        self.rescale_factors = nn.Parameter(torch.FloatTensor(1, n_channel_rescale, 1, 1, 1))  # unused in forward
        nn.init.constant_(self.rescale_factors, 20)

        # Lightning-surface state
        self.current_epoch = 0
        self.global_step = 0
        self.logged = {}
        self._scheduler = None
        self._engine = Engine(self)
        self._param_names = [n for n, _ in self.named_parameters()]
        self._det_ws = {}
        self._pred_programs = {}
        # "f32" (the reference's precision) or "bf16": eval-mode passes keep the activations in HBM as bf16 (fp32 weights,
        # BatchNorm vectors and accumulators; pointwise + head convolutions on bf16 MFMA) - BASELINE configs[3]
        self.compute_dtype = "f32"
        self.use_predict_programs = True  # predict_step replays a recorded launch program (see _predict_replay)

        # Prior boxes (ssd3d.py:244).  The draw below keeps RNG parity with the reference's third dummy pass.
        self.base.get_feature_map_infos(self.input_size, self.device)
        self.priors_cxcycz = self.create_prior_boxes(_draw=False) if torch.cuda.is_available() else None
        self.loss_fn = MultiBoxLoss(self.priors_cxcycz, threshold=threshold, alpha=alpha, **self._loss_variants)

    # -- Lightning surface ------------------------------------------------------------------------------
    @property
    def device(self):
        return next(self.parameters()).device

    def log(self, name, value, *args, **kwargs):
        self.logged[name] = float(value)

    def lr_schedulers(self):
        return self._scheduler

    def save_hyperparameters(self, *a, **k):
        pass

    # -- construction ---------------------------------------------------------------------------------------
    def _make_base_and_prediction_layers(self):
        if 'mobilenet' in self.base_network_config:
            if self.width_mult != 1.:
                raise NotImplementedError("width_mult != 1 crashes in the reference (ssd3d.py:130, SURVEY §0.2-8)")
            self.base = MobileNetBase(config=self.base_network_config, in_channels=self.input_channels,
                                      width_mult=self.width_mult, cube=self.cube, aspect_ratios=self.aspect_ratios)
            features_n_channels = self.base.get_feature_map_infos(self.input_size, None)[1]  # ssd3d.py:270
            self.pred_convs = PredictionConvolutions(self.n_classes, width_mult=self.width_mult,
                                                     aspect_ratios=self.aspect_ratios,
                                                     features_n_channels=features_n_channels,
                                                     boxes_per_location=self.boxes_per_location)
        elif 'convnet' in self.base_network_config:
            raise NotImplementedError("the 'convnet' backbone is unreachable in the reference (ssd3d.py:281) and needs "
                                      "MONAI's Convolution block; only 'mobilenet' is built")
        else:
            raise Exception(
                f"Unknown base network name. Expected 'mobilenet' or 'convnet' but got {self.base_network_config}")

    def create_prior_boxes(self, per_feature_map=False, _draw=True):
        """ssd3d.py:286-342 on the GPU: (P,6) centre-size priors, fp64 arithmetic rounded once to fp32, clamped
        to [0,1]; row order scale-major, then (i,j,k) row-major, then the 2 sizes; centre = ((j+.5)/d1,
        (i+.5)/d0, (k+.5)/d2) exactly like the reference."""
        fmd = (self.base.get_feature_map_infos(self.input_size, self.device) if _draw
               else self.base.feature_map_dims(self.input_size))[0]
        if not torch.cuda.is_available():
            raise _lib.HipKernelError("create_prior_boxes needs the HIP device (no CPU fallback)")
        dev = torch.device("cuda", torch.cuda.current_device())
        feats = list(self.aspect_ratios.keys())
        P = sum(fmd[f][0] * fmd[f][1] * fmd[f][2] * self.boxes_per_location for f in feats)
        out = torch.empty((P, 6), dtype=torch.float32, device=dev)
        off, ranges = 0, {}
        for f in feats:
            d0, d1, d2 = fmd[f]
            _lib.call("msl_make_priors", ptr(out), off, d0, d1, d2, float(self.scales[f]), self.boxes_per_location,
                      _stream())
            ranges[f] = (off, off + d0 * d1 * d2 * self.boxes_per_location)
            off = ranges[f][1]
        if not per_feature_map:
            return out
        host = out.cpu()
        return {f: host[lo:hi].tolist() for f, (lo, hi) in ranges.items()}

    def _ensure_device_state(self, dev):
        if self.priors_cxcycz is None or self.priors_cxcycz.device != dev:
            with torch.cuda.device(dev):
                self.priors_cxcycz = self.create_prior_boxes(_draw=False)
            self.loss_fn.set_priors(self.priors_cxcycz)

    def init(self):
        print("[INFO] Initializing model weights")
        self.base.init()
        self.pred_convs.init()

    def replay_reference_init_side_effects(self):
        """the reference's constructor pushes three train-mode ``randn`` batches through the
        backbone, leaving every BatchNorm with num_batches_tracked == 3 and nudged running statistics.  Call
        this once (after moving the model to the GPU) to reproduce that state with the HIP kernels."""
        was_training = self.training
        self.train()
        with torch.no_grad():
            for _ in range(3):
                x = torch.randn((1, self.input_channels, *self.input_size)).to(self.device)
                self._engine.forward(x, training=True, need_grad=False)
        self.train(was_training)

    # -- forward ------------------------------------------------------------------------------------------
    def forward(self, image):
        """ssd3d.py:248-263: image (N,C,D,H,W) -> locs (N,P,6), classes_scores (N,P,n_classes)."""
        if not image.is_cuda:
            raise _lib.HipKernelError("mslesions3d_amd runs on the HIP device only (no CPU fallback): move the model "
                                      "and the input to 'cuda'")
        self._ensure_device_state(image.device)
        self._engine.ensure_arena(image.device)
        params = [p for _, p in self.named_parameters()]
        need_grad = self.training and torch.is_grad_enabled()
        if self.compute_dtype not in ("f32", "bf16"):
            raise ValueError(f"compute_dtype must be 'f32' or 'bf16', got {self.compute_dtype!r}")
        locs, classes_scores = _SSDFunction.apply(self, need_grad, image, *params)
        self._engine.check_nan(self._engine.plan_for(image, need_grad))
        return locs, classes_scores

    # -- detection ------------------------------------------------------------------------------------------
    def _detect_workspace(self, N, P, ncls, top_k, dev):
        key = (N, P, ncls, top_k, dev)
        ws = self._det_ws.get(key)
        if ws is None:
            cap, k1 = 10 * top_k, ncls - 1
            wn = (cap + 63) // 64
            f32, i32, i64 = torch.float32, torch.int32, torch.int64
            ws = dict(probs=torch.empty((N, k1, P), dtype=f32, device=dev),
                      boxes=torch.empty((N, P, 6), dtype=f32, device=dev),
                      sorted_idx=torch.zeros((N, k1, cap), dtype=i32, device=dev),
                      ncand=torch.zeros(N * k1, dtype=i32, device=dev),
                      mask=torch.zeros((N, k1, cap, wn), dtype=i64, device=dev),
                      keep=torch.zeros((N, k1, wn), dtype=i64, device=dev),
                      nkept=torch.zeros(N * k1, dtype=i32, device=dev),
                      tmp_s=torch.empty((N, k1 * cap), dtype=f32, device=dev),
                      tmp_r=torch.empty((N, k1 * cap), dtype=i32, device=dev),
                      sel=torch.zeros(int(_lib.load().msl_detect_select_ws_ints(N, P, ncls)), dtype=i32, device=dev),
                      # the four output buffers are views of ONE allocation [labels | prior indices | boxes | scores], so
                      # that a batch's detections leave the workspace with one device copy instead of four
                      out_all=torch.empty(N * top_k * 44, dtype=torch.uint8, device=dev),
                      # per-image detection counts + one spare int: predict_step's eval plan keeps its NaN flag there, so that
                      # counts and flag leave the device with ONE copy
                      oc=torch.zeros(N + 1, dtype=i32, device=dev),
                      # pinned landing zone of the per-image detection counts (+ one slot for a NaN flag): filled by async
                      # copies, read after ONE stream synchronisation (a blocking 4-byte device-to-host copy costs 50-100 us)
                      host=torch.empty(N + 1, dtype=i32).pin_memory())
            ws.update(self._detect_out_views(ws["out_all"], N, top_k))
            self._det_ws[key] = ws
        return ws

    @staticmethod
    def _detect_out_views(buf, N, top_k):
        """(labels i64, prior indices i64, boxes f32 x 6, scores f32) views of a [N * top_k * 44]-byte buffer."""
        n = N * top_k
        return dict(ol=buf[:8 * n].view(torch.int64).view(N, top_k), op=buf[8 * n:16 * n].view(torch.int64).view(N, top_k),
                    ob=buf[16 * n:40 * n].view(torch.float32).view(N, top_k, 6), os=buf[40 * n:44 * n].view(torch.float32).view(N, top_k))

    def detect_objects(self, predicted_locs, predicted_scores, min_score, max_overlap, top_k, return_prior_index=False):
        """ssd3d.py:344-460.  -> three lists of length N: boxes (k,6) corner-form fractional, labels (k,) int64,
        scores (k,) fp32.  Equal scores are ordered by ascending prior index (stable sort)."""
        if not predicted_locs.is_cuda:
            raise _lib.HipKernelError("detect_objects runs on the HIP device only (no CPU fallback)")
        self._ensure_device_state(predicted_locs.device)
        N = predicted_locs.size(0)
        P = self.priors_cxcycz.size(0)
        assert P == predicted_locs.size(1) == predicted_scores.size(1)  # ssd3d.py:370
        ncls = predicted_scores.size(2)
        locs = predicted_locs.detach().contiguous().float()
        scores = predicted_scores.detach().contiguous().float()
        w = self._detect_workspace(N, P, ncls, int(top_k), locs.device)
        self._detect_launch(locs, scores, w, min_score, max_overlap, top_k)
        return self._detect_collect(w, N, return_prior_index)

    def _detect_launch(self, locs, scores, w, min_score, max_overlap, top_k):
        N, P, ncls = locs.size(0), locs.size(1), scores.size(2)
        _lib.call("msl_detect_objects", ptr(locs), ptr(scores), ptr(self.priors_cxcycz), N, P, ncls, float(min_score),
                  float(max_overlap), int(top_k), ptr(w["probs"]), ptr(w["boxes"]), ptr(w["sorted_idx"]), ptr(w["ncand"]),
                  ptr(w["mask"]), ptr(w["keep"]), ptr(w["nkept"]), ptr(w["tmp_s"]), ptr(w["tmp_r"]), ptr(w["sel"]), ptr(w["ob"]),
                  ptr(w["os"]), ptr(w["ol"]), ptr(w["op"]), ptr(w["oc"]), _stream())

    @staticmethod
    def _detect_collect_begin(w, N, nan_flag=None, slot=0):
        """Enqueue what takes a batch's detections out of the workspace - the counts (and, if given, the forward pass's NaN
        flag) to pinned memory, one device copy of the four output buffers (the workspace is reused by the next batch) -
        and record an event behind it.  ``slot``: which pinned landing zone (a pipelined caller keeps several batches in
        flight).  -> handle for ``_detect_collect_end``; no host synchronisation here."""
        hosts = w.setdefault("hosts", {0: w["host"]})
        host = hosts.get(slot)
        if host is None:
            host = hosts[slot] = torch.empty_like(w["host"]).pin_memory()
        if nan_flag is not None and nan_flag.data_ptr() == w["oc"][N:].data_ptr():
            host.copy_(w["oc"], non_blocking=True)  # counts and the plan's NaN flag (its home is oc[N]) in one copy
        else:
            host[:N].copy_(w["oc"][:N], non_blocking=True)
            if nan_flag is not None:
                host[N:].copy_(nan_flag, non_blocking=True)
        v = LSSD3D._detect_out_views(w["out_all"].clone(), *w["ol"].shape)  # one copy for all four outputs
        evs = w.setdefault("events", {})
        ev = evs.get(slot)
        if ev is None:
            ev = evs[slot] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(w["oc"].device))
        return {"views": v, "host": host, "event": ev, "N": N, "flag": nan_flag is not None}

    @staticmethod
    def _detect_collect_end(h, return_prior_index=False):
        """The host half: ONE synchronisation (on the handle's event), then the per-image lists."""
        N, host, v = h["N"], h["host"], h["views"]
        ob, ol, os_ = v["ob"], v["ol"], v["os"]
        op = v["op"] if return_prior_index else None
        h["event"].synchronize()  # the only host sync
        counts = host[:N].tolist()
        boxes = [ob[i, :counts[i]] for i in range(N)]
        labels = [ol[i, :counts[i]] for i in range(N)]
        dscores = [os_[i, :counts[i]] for i in range(N)]
        out = (boxes, labels, dscores, [op[i, :counts[i]] for i in range(N)]) if return_prior_index else (boxes, labels, dscores)
        if h["flag"]:
            return out + (int(host[N]),)
        return out

    @staticmethod
    def _detect_collect(w, N, return_prior_index=False, nan_flag=None):
        """Detections out of the workspace: one clone per output buffer, the counts (and, if given, the forward pass's NaN
        flag) through pinned memory, ONE host synchronisation.  With ``nan_flag`` the flag's value is returned as the last
        element."""
        return LSSD3D._detect_collect_end(LSSD3D._detect_collect_begin(w, N, nan_flag), return_prior_index)

    # -- steps ------------------------------------------------------------------------------------------------
    def _gt_warnings(self, gt_boxes, subjects):
        for i, subj_boxes in enumerate(gt_boxes):  # ssd3d.py:481-490
            if subj_boxes.numel() == 0:
                continue
            sizes = (subj_boxes[:, 3:] - subj_boxes[:, :3]).cpu()
            for axis in (0, 1, 2):
                negatives = int((sizes[:, axis] < 0).sum())
                zeros = int((sizes[:, axis] == 0).sum())
                if negatives > 0:
                    warnings.warn(f"Given boxes has invalid values (subject {subjects[i]}). The box size must "
                                  f"be non-negative but got {negatives} boxes with negative sizes.")
                if zeros > 0:
                    warnings.warn(f"Given boxes has invalid values (subject {subjects[i]}). The box size must "
                                  f"be non-zero but got {zeros} boxes with size of zero.")

    def _metrics(self, predicted_locs, predicted_scores, gt_boxes, gt_labels):
        det_boxes, det_labels, det_scores = self.detect_objects(predicted_locs, predicted_scores, self.min_score,
                                                                self.max_overlap, self.top_k)
        if predicted_locs.size(1) <= 500:  # ssd3d.py:504-515
            raise NotImplementedError
        dif = [torch.zeros(l.size(0), dtype=torch.bool) for l in gt_labels]
        m10 = calculate_mAP(det_boxes, det_labels, det_scores, gt_boxes, gt_labels, dif, min_overlap=0.1, return_detail=True)
        m50 = calculate_mAP(det_boxes, det_labels, det_scores, gt_boxes, gt_labels, dif, min_overlap=0.5, return_detail=True)
        return m10, m50

    def training_step(self, batch, batch_idx=None):
        """ssd3d.py:467-531."""
        dev = self.device
        images = batch["img"].to(dev)
        gt_boxes = [b.to(dev) for b in batch["boxes"]]
        gt_labels = [l.to(dev) for l in batch["labels"]]
        predicted_locs, predicted_scores = self(images)
        self._gt_warnings(gt_boxes, batch.get("subject", list(range(len(gt_boxes)))))
        conf_loss, loc_loss = self.loss_fn(predicted_locs, predicted_scores, gt_boxes, gt_labels)
        loss = conf_loss + self.loss_fn.alpha * loc_loss
        logs = {"train_total_loss": loss, "train_conf_loss": conf_loss, "train_loc_loss": loc_loss}
        if self.current_epoch % (self.compute_metric_every_n_epochs * 2) == 0:
            with torch.no_grad():
                logs["metrics_10"], logs["metrics_50"] = self._metrics(predicted_locs, predicted_scores, gt_boxes, gt_labels)
        self.log('total_loss/training', loss.item())
        self.log('confidence_loss/training', conf_loss.item())
        self.log('localization_loss/training', loc_loss.item())
        sch = self.lr_schedulers()
        if sch is not None:
            sch.step()
        return {'loss': loss, "log": logs}

    def validation_step(self, batch, batch_idx=None):
        """ssd3d.py:533-586."""
        dev = self.device
        images = batch["img"].to(dev)
        gt_boxes, gt_labels = batch["seg"] if "seg" in batch else (batch["boxes"], batch["labels"])
        gt_boxes = [b.to(dev) for b in gt_boxes]
        gt_labels = [l.to(dev) for l in gt_labels]
        with torch.no_grad():
            predicted_locs, predicted_scores = self(images)
            self._gt_warnings(gt_boxes, batch.get("subject", list(range(len(gt_boxes)))))
            conf_loss, loc_loss = self.loss_fn(predicted_locs, predicted_scores, gt_boxes, gt_labels)
            loss = conf_loss + self.loss_fn.alpha * loc_loss
            logs = {"val_total_loss": loss, "val_conf_loss": conf_loss, "val_loc_loss": loc_loss}
            if self.current_epoch % self.compute_metric_every_n_epochs == 0:
                m10, m50 = self._metrics(predicted_locs, predicted_scores, gt_boxes, gt_labels)
                m50["mAP"] = torch.FloatTensor([m50["mAP"]])
                logs["metrics_10"], logs["metrics_50"] = m10, m50
        return {'val_loss': loss, "log": logs}

    def predict_step(self, batch, batch_idx: int = 0, dataloader_idx: int = None):
        """ssd3d.py:692-702."""
        with torch.no_grad():
            if self.use_predict_programs and not self.training:
                return self._predict_replay(batch["img"])
            predicted_locs, predicted_scores = self(batch["img"].to(self.device))
            return self.detect_objects(predicted_locs, predicted_scores, min_score=self.min_score,
                                       max_overlap=self.max_overlap, top_k=self.top_k)

    def _predict_replay(self, img):
        """predict_step without Python between the launches: the first batch of a shape runs eval forward + decode + NMS
        through the executor and records its C-ABI calls on a persistent input buffer; later batches are copied into that
        buffer and the launch program is replayed natively (same kernels, same arguments).  One host sync per batch."""
        return self._predict_finish(self._predict_enqueue(img))

    def _predict_enqueue(self, img, slot=0):
        """The device half of ``_predict_replay``: stage the batch, replay (or record) the launch program, enqueue the
        copies that take the detections out -> handle for ``_predict_finish``.  No host synchronisation."""
        dev = self.device
        x = img  # a host tensor goes straight into the staging buffer below (one host-to-device copy, no device-side second copy)
        self._ensure_device_state(dev)
        eng = self._engine
        eng.ensure_arena(dev)
        key = (x.shape, _stream(), self.min_score, self.max_overlap, self.top_k, id(eng.arena), self.compute_dtype)
        ent = self._pred_programs.get(key)
        if ent is None:
            buf = torch.empty(x.shape, dtype=torch.float32, device=dev)
            buf.copy_(x, non_blocking=True)
            # the eval plan's NaN flag moves into the spare int behind the detection counts (one device-to-host copy per pass
            # instead of two); every launch recorded below takes the flag's address from the plan
            w = self._detect_workspace(x.size(0), self.priors_cxcycz.size(0), self.n_classes, int(self.top_k), dev)
            eng.plan_for(buf, False).nan_flag = w["oc"][x.size(0):]
            _lib.start_recording()
            try:
                locs, scores = eng.forward(buf, training=False, need_grad=False)
                assert (locs.size(0), locs.size(1), scores.size(2)) == (x.size(0), self.priors_cxcycz.size(0), self.n_classes)
                self._detect_launch(locs, scores, w, self.min_score, self.max_overlap, self.top_k)
            finally:
                prog = _lib.stop_recording()
            ent = self._pred_programs[key] = {"buf": buf, "ws": w, "plan": eng.plan_for(buf, False), "prog": prog,
                                              "compiled": _lib.compile_program(prog, set())}
        else:
            if not (x.is_cuda and x.data_ptr() == ent["buf"].data_ptr()):  # (a caller may fill predict_input_buffer() itself)
                ent["buf"].copy_(x, non_blocking=True)
            if eng.prof is not None and eng.prof_tags:  # bench.py's roofline leg: HIP-event pairs around the tagged launches
                tags = frozenset(eng.prof_tags)
                timed = ent.setdefault("timed", {})
                if tags not in timed:
                    timed[tags] = _lib.compile_program(ent["prog"], tags)
                _lib.replay_native(timed[tags], eng.prof)
            else:
                _lib.replay_native(ent["compiled"], None)
        return self._detect_collect_begin(ent["ws"], x.size(0), nan_flag=ent["plan"].nan_flag, slot=slot)

    def _predict_finish(self, handle):
        *out, flag = self._detect_collect_end(handle)
        self._engine.raise_on_nan_flag(flag)  # (the forward pass's NaN flag came over with the detection counts)
        return tuple(out)

    def predict_batches(self, batches, depth=2):
        """``predict_step`` over an iterable of batches (dicts with "img"), as a generator of its results in order, with
        ``depth`` batches in flight: batch k + 1 is staged and its launch program enqueued BEFORE the host waits for batch
        k's detections, so the device never idles through the host's synchronisation, list building and next enqueue
        (~0.1 ms of a 0.5 ms pass at 192^3 x 2).  The passes share one stream, hence one set of activation buffers: only
        the pinned landing zones of the counts exist per slot.  ``depth=1`` is predict_step batch by batch.  Same kernels,
        same results (tests/test_gpu_infer.py)."""
        if self.training or not self.use_predict_programs:
            for k, batch in enumerate(batches):
                yield self.predict_step(batch, k)
            return
        depth = max(1, int(depth))
        inflight = []
        with torch.no_grad():
            for k, batch in enumerate(batches):
                inflight.append(self._predict_enqueue(batch["img"], slot=k % depth))
                if len(inflight) >= depth:
                    yield self._predict_finish(inflight.pop(0))
            while inflight:
                yield self._predict_finish(inflight.pop(0))

    def predict_input_buffer(self, shape):
        """The persistent device buffer ``predict_step`` stages batches of ``shape`` in (None before the first batch of that
        shape).  A data loader that writes its batch straight into it (``buf.copy_(host_batch, non_blocking=True)``) and
        passes the buffer itself to ``predict_step`` saves the device-to-device copy of the batch."""
        for key, ent in self._pred_programs.items():
            if tuple(key[0]) == tuple(shape) and key[-1] == self.compute_dtype:
                return ent["buf"]
        return None

    def configure_optimizers(self):
        """ssd3d.py:704-722: Adam(weight_decay 5e-4), '.bias' parameters at 2*lr, cosine annealing T_max=40."""
        from .optim import CosineAnnealingLR, FusedAdam
        optimizer = FusedAdam(self, lr=self.lr, weight_decay=0.0005)
        if self.scheduler != "none":
            scheduler = CosineAnnealingLR(optimizer, T_max=40)
            return [optimizer], [scheduler]
        return optimizer

    def compute_parameters_median_size(self):
        with torch.no_grad():
            return sum(abs(p).sum() for p in self.parameters())

    # -- checkpoints (Lightning-compatible dict) ------------------------------------------------------
    @staticmethod
    def _plain(v):
        """Hyper-parameters as plain Python types (numpy scalars cast), so that the file loads with
        ``torch.load(weights_only=True)`` — the only loader this package uses on a checkpoint."""
        if isinstance(v, dict):
            return {LSSD3D._plain(k): LSSD3D._plain(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return type(v)(LSSD3D._plain(x) for x in v)
        if isinstance(v, np.generic):
            return v.item()
        if isinstance(v, np.ndarray):
            return v.tolist()
        return v

    def save_checkpoint(self, path, optimizer=None, scheduler=None, loop_state=None):
        """Lightning's key layout (``state_dict`` / ``hyper_parameters`` / ``epoch`` / ``global_step`` /
        ``optimizer_states`` / ``lr_schedulers``, reference train.py:171-176,185) with tensors and plain types only.
        ``optimizer`` may be a FusedAdam or a FusedTrainer (then its scheduler is saved too)."""
        ckpt = {"state_dict": {k: v.detach().cpu() for k, v in self.state_dict().items()},
                "hyper_parameters": self._plain(dict(self.hparams)), "epoch": int(self.current_epoch),
                "global_step": int(self.global_step)}
        if optimizer is not None and hasattr(optimizer, "opt"):  # a FusedTrainer
            scheduler = optimizer.sch if scheduler is None else scheduler
            optimizer = optimizer.opt
        if optimizer is not None:
            ckpt["optimizer_states"] = [self._plain(optimizer.state_dict())]
        if scheduler is not None:
            ckpt["lr_schedulers"] = [self._plain(scheduler.state_dict())]
        if loop_state is not None:  # train.py's own bookkeeping (top-3 list, early-stopping counter): plain types
            ckpt["loop_state"] = self._plain(loop_state)
        tmp = f"{path}.tmp{os.getpid()}"
        torch.save(ckpt, tmp)
        os.replace(tmp, path)  # a reader (or a crash) never sees a half-written checkpoint

    @staticmethod
    def read_checkpoint(path, map_location="cpu"):
        """The one place a checkpoint file is opened: ``weights_only=True``, no fallback.  A Lightning ``.ckpt`` written by
        the reference pickles arbitrary objects (callbacks, ``AttributeDict``) and is refused here: re-export it on a
        trusted machine as ``{"state_dict": ..., "hyper_parameters": {plain types}}``."""
        try:
            return torch.load(path, map_location=map_location, weights_only=True)
        except Exception as e:  # noqa: BLE001 - any unpickling refusal
            raise RuntimeError(
                f"{path}: not loadable with torch.load(weights_only=True) ({type(e).__name__}: {str(e).splitlines()[0][:200]}). "
                "mslesions3d_amd never unpickles arbitrary objects from a checkpoint; re-export the file as a dict of "
                "tensors and plain Python types (state_dict + hyper_parameters).") from e

    @classmethod
    def load_from_checkpoint(cls, path, map_location="cpu", **overrides):
        ckpt = cls.read_checkpoint(path, map_location)
        hp = dict(ckpt.get("hyper_parameters", {}))
        hp.update(overrides)
        hp = {k: v for k, v in hp.items() if k in cls.__init__.__code__.co_varnames}
        if "aspect_ratios" in hp:  # JSON-ish round trips may have turned the feature indices into strings
            hp["aspect_ratios"] = {int(k): v for k, v in hp["aspect_ratios"].items()}
        model = cls(**hp)
        model.load_state_dict(ckpt["state_dict"])
        model.current_epoch = int(ckpt.get("epoch", 0))
        model.global_step = int(ckpt.get("global_step", 0))
        return model


class _MultiBoxLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loss_mod, locs, scores, gt_boxes, gt_labels, obj_off, total_objects):
        st = loss_mod._state(locs.shape[0], locs.shape[1], scores.shape[2], total_objects, locs.device)
        locs_c, scores_c = locs.contiguous().float(), scores.contiguous().float()
        loss_mod._run_forward(st, locs_c, scores_c, gt_boxes, gt_labels, obj_off, total_objects)
        ctx.loss_mod, ctx.st = loss_mod, st
        ctx.save_for_backward(locs_c, scores_c)
        out = st["loss_out"]
        return out[0].clone(), out[1].clone()

    @staticmethod
    def backward(ctx, g_conf, g_loc):
        locs, scores = ctx.saved_tensors
        st = ctx.st
        st["upstream"].copy_(torch.stack([g_conf.reshape(()), g_loc.reshape(())]))
        N, P, ncls = locs.shape[0], locs.shape[1], scores.shape[2]
        lm = ctx.loss_mod
        if lm.variant_flags:  # forward + backward of the variant in one call (the mined mask is rebuilt)
            _lib.call("msl_multibox_loss_var", ptr(locs), ptr(scores), ptr(st["true_classes"]), ptr(st["true_locs"]),
                      ptr(st["ws"]), ptr(lm._var_ws(st, N, P)), ptr(st["loss_out"]), ptr(st["upstream"]), ptr(st["dlocs"]),
                      ptr(st["dscores"]), None, N, P, ncls, lm.variant_flags, int(lm.neg_pos_ratio), _stream())
        else:
            _lib.call("msl_multibox_loss_bwd", ptr(locs), ptr(scores), ptr(st["true_classes"]), ptr(st["true_locs"]),
                      ptr(st["loss_out"]), ptr(st["upstream"]), ptr(st["dlocs"]), ptr(st["dscores"]), N, P, ncls, _stream())
        return None, st["dlocs"].clone(), st["dscores"].clone(), None, None, None, None


class MultiBoxLoss(nn.Module):
    """The MultiBox loss (ssd3d.py:741-941): prior<->object matching + target encoding + confidence (cross
    entropy over all non-ignored priors / number of positives) + localisation (mean |.| over positives) loss.

    ``hard_negative_mining`` / ``smooth_l1`` / ``focal`` (keyword-only, default off = the reference's live code) switch on
    the variants the reference keeps as commented code: the mining recipe of ssd3d.py:926-932 with ``neg_pos_ratio``,
    ``nn.SmoothL1Loss`` for the attribute it calls ``smooth_l1`` (ssd3d.py:758), MONAI's ``FocalLoss`` of ssd3d.py:760."""

    def __init__(self, priors_cxcycz, threshold=0.5, neg_pos_ratio=3, alpha=1., *, hard_negative_mining=False,
                 smooth_l1=False, focal=False):
        super(MultiBoxLoss, self).__init__()
        self.priors_cxcycz = priors_cxcycz
        self.priors_xyz = None
        self.threshold = threshold
        self.neg_pos_ratio = neg_pos_ratio
        self.alpha = alpha
        self.variant_flags = (1 if hard_negative_mining else 0) | (2 if smooth_l1 else 0) | (4 if focal else 0)
        if type(self.threshold) == list:  # ssd3d.py:762-773
            if len(self.threshold) == 1:
                self.thresholding_mode = "hard"
                self.threshold = self.threshold[0]
            else:
                self.thresholding_mode = "soft"
                assert (len(self.threshold) == 2)
        elif type(self.threshold) == float:
            self.thresholding_mode = "hard"
        else:
            raise Exception(
                f"Type error. Expected float or list of floats for threshold but got {type(self.threshold)}")
        self._states = {}

    def set_priors(self, priors_cxcycz):
        self.priors_cxcycz = priors_cxcycz
        self.priors_xyz = None

    def _state(self, N, P, ncls, total_objects, dev):
        key = (N, P, ncls, dev)
        st = self._states.get(key)
        if st is None:
            f32, i32, i64 = torch.float32, torch.int32, torch.int64
            L = _lib.load()
            st = dict(overlap=torch.zeros((N, P), dtype=f32, device=dev), obj=torch.zeros((N, P), dtype=i32, device=dev),
                      prior_for_obj=torch.zeros(4096, dtype=i32, device=dev),
                      true_classes=torch.zeros((N, P), dtype=i64, device=dev),
                      true_locs=torch.zeros((N, P, 6), dtype=f32, device=dev),
                      matched=torch.zeros((N, P), dtype=i64, device=dev),
                      ws=torch.zeros(L.msl_multibox_loss_workspace_bytes() // 8, dtype=torch.float64, device=dev),
                      loss_out=torch.zeros(3, dtype=f32, device=dev), upstream=torch.ones(2, dtype=f32, device=dev),
                      dlocs=torch.zeros((N, P, 6), dtype=f32, device=dev),
                      dscores=torch.zeros((N, P, ncls), dtype=f32, device=dev),
                      # fused hot-loop form (msl_multibox_loss_pack): positives counter of the matching + loss partials
                      npos=torch.zeros(1, dtype=i32, device=dev),
                      pack_parts=torch.zeros(2 * L.msl_multibox_loss_pack_num_partials(N, P), dtype=torch.float64, device=dev))
            self._states[key] = st
        if st["prior_for_obj"].numel() < total_objects:
            st["prior_for_obj"] = torch.zeros(2 * total_objects, dtype=torch.int32, device=dev)
        return st

    @staticmethod
    def _var_ws(st, N, P):
        """Scratch of the loss variants (mined mask + per-image sums), allocated on first use."""
        if "var_ws" not in st:
            nbytes = _lib.load().msl_multibox_loss_var_workspace_bytes(N, P)
            st["var_ws"] = torch.zeros((nbytes + 7) // 8, dtype=torch.float64, device=st["ws"].device)
        return st["var_ws"]

    @staticmethod
    def pack_targets(boxes, labels, dev):
        """list[(n_i,6)], list[(n_i,)] -> concatenated boxes (T,6) f32, labels (T,) i64, offsets (N+1,) i32, T."""
        sizes = [int(b.shape[0]) for b in boxes]
        offs = np.zeros(len(sizes) + 1, dtype=np.int32)
        offs[1:] = np.cumsum(sizes)
        T = int(offs[-1])
        if T > 0:
            gb = torch.cat([b.reshape(-1, 6) for b in boxes], 0).to(device=dev, dtype=torch.float32).contiguous()
            gl = torch.cat([l.reshape(-1) for l in labels], 0).to(device=dev, dtype=torch.int64).contiguous()
        else:
            gb = torch.zeros((1, 6), dtype=torch.float32, device=dev)
            gl = torch.zeros(1, dtype=torch.int64, device=dev)
        return gb, gl, torch.from_numpy(offs).to(dev), T

    def _thresholds(self):
        if self.thresholding_mode == "hard":
            return float(self.threshold), 0.0, 0
        return float(self.threshold[0]), float(self.threshold[1]), 1

    def _run_match(self, st, N, gt_boxes, gt_labels, obj_off, T, stream=None, count=False):
        """Matching + target encoding only (depends on the ground truth and the priors, not on the network).  ``count``: also
        leave the number of positive priors in ``st["npos"]`` (what ``_run_loss_pack`` divides by)."""
        P = self.priors_cxcycz.size(0)
        lo, hi, soft = self._thresholds()
        args = [ptr(gt_boxes), ptr(gt_labels), ptr(obj_off), T, ptr(self.priors_cxcycz), N, P, lo, hi, soft, ptr(st["overlap"]),
                ptr(st["obj"]), ptr(st["prior_for_obj"]), ptr(st["true_classes"]), ptr(st["true_locs"]), ptr(st["matched"])]
        if count:
            _lib.call("msl_multibox_match_count", *args, ptr(st["npos"]), _stream() if stream is None else stream)
        else:
            _lib.call("msl_multibox_match", *args, _stream() if stream is None else stream)

    def can_pack(self, n_scales):
        """Is the one-launch loss + gradient + head-gradient-image form available (the live loss, at most four scales)?"""
        return not self.variant_flags and n_scales <= 4

    def _run_loss_pack(self, st, locs, scores, upstream, nan_flag, dO, dims, prior_off):
        """Loss terms, their gradients and the zero-haloed head-gradient images in one launch (training hot loop; the targets
        and the positives counter are in ``st``: ``_run_match(count=True)``).  ``dO`` / ``dims`` / ``prior_off``: per scale.  The
        loss VALUES reach ``st["loss_out"]`` with the step's batched gradient reduction (Engine._grad_reduce, kind 4)."""
        import ctypes
        N, P, ncls = locs.shape[0], locs.shape[1], scores.shape[2]
        assert P == self.priors_cxcycz.size(0) == scores.size(1)  # ssd3d.py:845
        n = len(dO)
        key = tuple(ptr(t) for t in dO) + tuple(prior_off)
        if st.get("pack_key") != key:
            I = ctypes.c_int * n
            st["pack_args"] = ((ctypes.c_void_p * n)(*[ptr(t) for t in dO]),) + tuple(I(*[d[a] for d in dims]) for a in range(3)) + (
                I(*prior_off),)
            st["pack_key"] = key
        a = st["pack_args"]  # host arrays: kept alive by the state (a recorded launch program points at them)
        _lib.call("msl_multibox_loss_pack", ptr(locs), ptr(scores), ptr(st["true_classes"]), ptr(st["true_locs"]), ptr(st["npos"]),
                  ptr(upstream), ptr(st["pack_parts"]), ptr(nan_flag), *[ctypes.addressof(x) for x in a], n, N, P, ncls, _stream())

    def _run_forward(self, st, locs, scores, gt_boxes, gt_labels, obj_off, T, with_backward_upstream=None,
                     matched=False, nan_flag=None):
        """Matching + loss.  With ``with_backward_upstream`` (a 2-float device tensor [dL/dconf, dL/dloc]) the loss
        gradient w.r.t. locs / scores is produced in the same call (training hot loop); ``matched``: the targets
        are already in ``st`` (the trainer runs the matching beside the forward pass)."""
        N, P, ncls = locs.shape[0], locs.shape[1], scores.shape[2]
        assert P == self.priors_cxcycz.size(0) == scores.size(1)  # ssd3d.py:845
        sm = _stream()
        if not matched:
            self._run_match(st, N, gt_boxes, gt_labels, obj_off, T)
        if self.variant_flags:
            if int(self.neg_pos_ratio) != self.neg_pos_ratio or self.neg_pos_ratio < 0:
                raise ValueError("neg_pos_ratio must be a non-negative integer")
            up = with_backward_upstream
            _lib.call("msl_multibox_loss_var", ptr(locs), ptr(scores), ptr(st["true_classes"]), ptr(st["true_locs"]),
                      ptr(st["ws"]), ptr(self._var_ws(st, N, P)), ptr(st["loss_out"]), ptr(up),
                      ptr(st["dlocs"]) if up is not None else None, ptr(st["dscores"]) if up is not None else None,
                      ptr(nan_flag), N, P, ncls, self.variant_flags, int(self.neg_pos_ratio), sm)
            return
        if with_backward_upstream is not None:
            _lib.call("msl_multibox_loss_fwd_bwd", ptr(locs), ptr(scores), ptr(st["true_classes"]), ptr(st["true_locs"]),
                      ptr(st["ws"]), ptr(st["loss_out"]), ptr(with_backward_upstream), ptr(st["dlocs"]), ptr(st["dscores"]),
                      ptr(nan_flag), N, P, ncls, sm)
            return
        _lib.call("msl_multibox_loss_fwd", ptr(locs), ptr(scores), ptr(st["true_classes"]), ptr(st["true_locs"]),
                  ptr(st["ws"]), ptr(st["loss_out"]), N, P, ncls, sm)

    def match(self, boxes, labels, n_classes=2):
        """Matching only -> (true_classes (N,P) int64, true_locs (N,P,6), matched object (N,P) int64)."""
        dev = self.priors_cxcycz.device
        gb, gl, off, T = self.pack_targets(boxes, labels, dev)
        N, P = len(boxes), self.priors_cxcycz.size(0)
        st = self._state(N, P, n_classes, T, dev)
        lo, hi, soft = self._thresholds()
        _lib.call("msl_multibox_match", ptr(gb), ptr(gl), ptr(off), T, ptr(self.priors_cxcycz), N, P, lo, hi, soft,
                  ptr(st["overlap"]), ptr(st["obj"]), ptr(st["prior_for_obj"]), ptr(st["true_classes"]),
                  ptr(st["true_locs"]), ptr(st["matched"]), _stream())
        return st["true_classes"].clone(), st["true_locs"].clone(), st["matched"].clone()

    def forward(self, predicted_locs, predicted_scores, boxes, labels):
        """-> (conf_loss, loc_loss) scalars (ssd3d.py:775-941)."""
        if not predicted_locs.is_cuda:
            raise _lib.HipKernelError("MultiBoxLoss runs on the HIP device only (no CPU fallback)")
        if self.priors_cxcycz is None:
            raise RuntimeError("MultiBoxLoss has no priors (model constructed without a GPU and never run)")
        gb, gl, off, T = self.pack_targets(boxes, labels, predicted_locs.device)
        conf_loss, loc_loss = _MultiBoxLossFunction.apply(self, predicted_locs, predicted_scores, gb, gl, off, T)
        if torch.isnan(loc_loss):  # ssd3d.py:938-940 (batch without a single positive prior)
            raise Exception("Loss is NaN")
        return conf_loss, loc_loss
