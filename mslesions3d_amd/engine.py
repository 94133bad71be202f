"""Static executor for the 3D-SSD network on one MI355X.

The network of ``lesions3d/ssd3d.py:248-263`` (MobileNet-3D backbone ``ssd3d.py:47-100`` /
``mobilenet.py:26-49`` + prediction heads ``ssd3d.py:113-169``) is a fixed chain, so instead of a tracing
compiler the executor lays the whole step out once per input shape ("plan"): every activation, gradient,
BatchNorm vector and workspace is a resident HBM buffer (the 128^3 x 4 plan is ~0.8 GB of 288 GB), and
forward / backward are fixed sequences of C-ABI kernel launches on the current HIP stream (capturable into a
HIP graph: no host synchronisation, no allocation inside).

Data convention: conv kernels write RAW outputs + fp64 statistic partials; ``msl_bn_finalize`` turns them into
a per-channel (scale, shift); the next kernel applies relu(x*scale+shift) while loading.  Only the three
feature maps the heads read are materialised (in a zero-haloed layout).
"""
import os

import torch

from . import _lib
from ._lib import ptr

BN_ROWS = 8  # scale, shift, mean, invstd, c1, c2, cC, cE (the last two: msl_bn_bwd_finalize_coef)


def conv_out(d, s):
    return (d - 1) // s + 1


class ParamArena:
    """All parameters of the model in one flat fp32 HBM buffer, ordered by backward completion (heads first,
    stem last, gradient-less parameters at the end), with flat grad / Adam-moment twins.  ``nn.Parameter``s
    become views of the arena, so ``state_dict`` / ``load_state_dict`` keep working with the reference's keys."""

    def __init__(self, model, device):
        named = list(model.named_parameters())
        self.no_grad_names = {"rescale_factors"}  # unused in forward (ssd3d.py:251-254): grad stays None

        def order_key(item):
            name = item[0]
            if name in self.no_grad_names:
                return (2, 0, name)
            if name.startswith("pred_convs"):
                return (0, 0, name)
            idx = int(name.split(".")[2])  # base.features.<idx>....
            return (1, -idx, name)

        ordered = sorted(named, key=order_key)
        self.names = [n for n, _ in ordered]
        total = sum(p.numel() for _, p in ordered)
        self.n_trainable = sum(p.numel() for n, p in ordered if n not in self.no_grad_names)
        self.flat = torch.empty(total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.n_trainable, dtype=torch.float32, device=device)
        self.is_bias = torch.zeros(self.n_trainable, dtype=torch.uint8, device=device)
        self.offsets = {}
        self.views = {}
        self.grad_views = {}
        off = 0
        with torch.no_grad():
            for name, p in ordered:
                n = p.numel()
                view = self.flat[off:off + n].view(p.shape)
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = view
                self.offsets[name] = (off, n)
                self.views[name] = view
                if name not in self.no_grad_names:
                    self.grad_views[name] = self.grad[off:off + n].view(p.shape)
                    if name.endswith(".bias"):  # ssd3d.py:709 (BN betas match the rule too)
                        self.is_bias[off:off + n] = 1
                off += n
        self.params = {name: p for name, p in ordered}
        # ownership fingerprint (see owns): every parameter slot and every sub-module slot of the model at build time
        self._model = model
        self._slots = []
        for prefix, mod in model.named_modules():
            for key, q in mod._parameters.items():
                if q is not None:
                    name = f"{prefix}.{key}" if prefix else key
                    self._slots.append((mod._parameters, key, q, self.views[name].data_ptr()))
        self._mods = [(parent._modules, name, child) for parent in model.modules()
                      for name, child in parent._modules.items() if child is not None]

    def owns(self, model):
        """Are the model's parameters still the views of this arena?  Called on every step / batch, so it must not walk
        ``named_parameters()`` (~100 us of Python): it re-checks the slots recorded at build time - a parameter object
        replaced in its module, a parameter whose storage moved (``.to()``, ``p.data = ...``), a sub-module swapped."""
        if model is not self._model:
            return False
        for d, k, q, ptr in self._slots:
            cur = d.get(k)
            if cur is not q or cur.data_ptr() != ptr:
                return False
        for d, k, child in self._mods:
            if d.get(k) is not child:
                return False
        return True

    def bucket_ranges(self, n_buckets):
        """Contiguous [lo, hi) ranges of the flat gradient that complete in this order during backward."""
        bounds = [0]
        # Split on parameter boundaries.  The LAST bucket completes with the last backward kernel, so its all-reduce is
        # the only one whose latency is exposed in front of the optimiser: keep it tiny (the final ~2 % of the
        # gradient: the first blocks and the stem, a latency-only exchange); the buckets before it share the rest
        # equally and overlap the remaining backward kernels.
        tail = 0.98 if n_buckets > 1 else 1.0
        targets = [self.n_trainable * tail * k / max(n_buckets - 1, 1) for k in range(1, n_buckets)]
        acc = 0
        for name in self.names:
            if name in self.no_grad_names:
                continue
            acc = self.offsets[name][0] + self.offsets[name][1]
            if len(bounds) <= len(targets) and acc >= targets[len(bounds) - 1] and acc < self.n_trainable:
                bounds.append(acc)
        bounds.append(self.n_trainable)
        return [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1) if bounds[i + 1] > bounds[i]]


class Plan:
    """Buffers + static shape data for one (batch, input size, mode)."""

    def __init__(self, engine, N, in_dims, device, need_grad):
        L = _lib.load()
        m = engine.model
        self.N, self.in_dims, self.need_grad = N, tuple(in_dims), need_grad
        f32 = dict(dtype=torch.float32, device=device)
        specs = engine.layer_specs
        self.dims = []  # output dims per feature index
        cur = tuple(in_dims)
        for sp in specs:
            cur = tuple(conv_out(d, s) for d, s in zip(cur, sp["stride"]))
            self.dims.append(cur)
        self.y = []   # raw conv output of feature i (stem conv / block pointwise)
        self.z = [None]  # raw depthwise output of block i
        self.bn_y = []   # (6, C) BatchNorm vectors for y[i]
        self.bn_z = [None]
        part_elems = 0
        ncls = m.n_classes
        for i, sp in enumerate(specs):
            D, H, W = self.dims[i]
            S = D * H * W
            self.y.append(torch.empty((N, sp["cout"], D, H, W), **f32))
            self.bn_y.append(torch.zeros((BN_ROWS, sp["cout"]), **f32))
            if i == 0:
                part_elems = max(part_elems, 2 * sp["cout"] * L.msl_stem_conv_fwd_num_partials(N, D, H, W))
            else:
                pd, ph, pw = self.dims[i - 1]
                self.z.append(torch.empty((N, sp["cin"], D, H, W), **f32))
                self.bn_z.append(torch.zeros((BN_ROWS, sp["cin"]), **f32))
                part_elems = max(part_elems,
                                 2 * sp["cin"] * L.msl_dwconv_fwd_num_partials(N, sp["cin"], pd, ph, pw, sp["stride"][0]),
                                 2 * sp["cout"] * L.msl_pwconv_fwd_num_partials(N, sp["cin"], sp["cout"], S))
                if need_grad:
                    part_elems = max(part_elems,
                                     sp["cin"] * 27 * L.msl_dwconv_bwd_weight_num_partials(N, sp["cin"], pd, ph, pw, sp["stride"][0]),
                                     2 * sp["cin"] * L.msl_bn_relu_bwd_num_partials(N, S),
                                     2 * sp["cin"] * max(L.msl_dwconv_bwd_data_bnreduce_num_partials(N, sp["cin"], pd, ph, pw), 0))
            if need_grad:
                part_elems = max(part_elems, 2 * sp["cout"] * L.msl_bn_relu_bwd_num_partials(N, S))
        self.partials = torch.empty(max(part_elems, 1), dtype=torch.float64, device=device)
        # one statistics buffer per BatchNorm: the consumer folds them in its prologue while the finalize kernel
        # (running stats + vectors for backward) reads them on a side stream
        f64 = dict(dtype=torch.float64, device=device)
        self.np_y, self.np_z, self.part_y, self.part_z = [], [None], [], [None]
        for i, sp in enumerate(specs):
            D, H, W = self.dims[i]
            if i == 0:
                self.np_y.append(L.msl_stem_conv_fwd_num_partials(N, D, H, W))
            else:
                pd, ph, pw = self.dims[i - 1]
                self.np_z.append(L.msl_dwconv_fwd_num_partials(N, sp["cin"], pd, ph, pw, sp["stride"][0]))
                self.part_z.append(torch.empty(2 * sp["cin"] * self.np_z[i], **f64))
                self.np_y.append(L.msl_pwconv_fwd_num_partials(N, sp["cin"], sp["cout"], D * H * W))
            self.part_y.append(torch.empty(2 * sp["cout"] * self.np_y[i], **f64))

        # heads
        self.feat_ids = list(m.aspect_ratios.keys())
        self.prior_off, off = {}, 0
        for f in self.feat_ids:
            self.prior_off[f] = off
            D, H, W = self.dims[f]
            off += D * H * W * m.boxes_per_location
        self.P = off
        self.fpad, self.Wf, self.Wb, self.head_ws = {}, {}, {}, {}
        for f in self.feat_ids:
            C = specs[f]["cout"]
            D, H, W = self.dims[f]
            self.fpad[f] = torch.zeros((N, C, D + 2, H + 2, W + 2), **f32)
            ne = L.msl_head_packed_weight_elems(C, ncls)
            self.Wf[f] = torch.empty(ne, **f32)
            self.Wb[f] = torch.empty(ne, **f32)
            ws = max(L.msl_head_fwd_workspace_bytes(N, C, D, H, W, ncls),
                     L.msl_head_bwd_weight_workspace_bytes(N, C, D, H, W, ncls) if need_grad else 0)
            self.head_ws[f] = torch.empty(max(ws // 4, 1), **f32)
        self.locs = torch.empty((N, self.P, 6), **f32)
        self.scores = torch.empty((N, self.P, ncls), **f32)
        self.nan_flag = torch.zeros(1, dtype=torch.int32, device=device)

        if need_grad:
            self.g_y = [torch.empty_like(t) for t in self.y]
            self.g_z = [None] + [torch.empty_like(t) for t in self.z[1:]]
            mt16 = 16 * ((12 + 2 * ncls + 15) // 16)
            self.dO = {f: torch.zeros((N, mt16) + tuple(d + 2 for d in self.dims[f]), **f32) for f in self.feat_ids}
            self.ws_stem = torch.empty(max(L.msl_stem_conv_bwd_weight_workspace_bytes(specs[0]["cin"]) // 4, 1), **f32)
            # Deferred gradient reduction (Engine._grad_reduce): every weight-gradient kernel leaves its partial sums in a
            # buffer of ITS OWN (they all stay live until the one batched reduction in front of the optimiser):
            #   pw_slabs[i]  fp32 [nslabs][Cout][Cin] (None: the kernel covers the whole position range and writes dW itself)
            #   dw_part[i]   fp64 [Cin*27][NP]
            self.pw_nslabs, self.pw_slabs, self.dw_np, self.dw_part = [0], [None], [0], [None]
            for i in range(1, len(specs)):
                D, H, W = self.dims[i]
                pd, ph, pw = self.dims[i - 1]
                ns = L.msl_pwconv_bwd_weight_nslabs(N, specs[i]["cin"], specs[i]["cout"], D * H * W)
                self.pw_nslabs.append(ns)
                self.pw_slabs.append(torch.empty(ns * specs[i]["cin"] * specs[i]["cout"], **f32) if ns > 1 else None)
                npd = L.msl_dwconv_bwd_weight_num_partials(N, specs[i]["cin"], pd, ph, pw, specs[i]["stride"][0])
                self.dw_np.append(npd)
                self.dw_part.append(torch.empty(specs[i]["cin"] * 27 * npd, dtype=torch.float64, device=device))
            # blocks whose whole pointwise backward is one pass (csrc/pwfused.hip): per block (slabs [NP][Cout][Cin], fp64
            # statistics partials [2][Cin][NP], NP)
            self.pw_fused = {}
            for i in range(1, len(specs)):
                D, H, W = self.dims[i]
                nf = L.msl_pwconv_bwd_fused_num_partials(N, specs[i]["cin"], specs[i]["cout"], D * H * W)
                if nf > 0:
                    self.pw_fused[i] = (torch.empty(nf * specs[i]["cin"] * specs[i]["cout"], **f32),
                                        torch.empty(2 * specs[i]["cin"] * nf, dtype=torch.float64, device=device), nf)
            # big stride-2 blocks whose depthwise weight gradient comes with their bwd-data pass: fp64 partials [Cin*27][NP]
            self.dw_fused_part = {}
            for i in range(2, len(specs)):
                pd, ph, pw = self.dims[i - 1]
                if specs[i]["stride"][0] == 2 and N * pd * ph * pw > 65536:
                    npw = L.msl_dwconv_s2_bwd_bnreduce_bww_num_partials(N, specs[i]["cin"], pd, ph, pw)
                    if npw > 0 and 2 * specs[i]["cin"] * npw <= self.partials.numel():
                        self.dw_fused_part[i] = (torch.empty(specs[i]["cin"] * 27 * npw, dtype=torch.float64, device=device), npw)
            self.head_nslabs = {f: L.msl_head_conv_bwd_weight_nslabs(N, specs[f]["cout"], *self.dims[f], ncls) for f in self.feat_ids}
            self.stem_nslabs = L.msl_stem_conv_bwd_weight_nslabs(N, *self.in_dims, *specs[0]["stride"])
            self.grad_tables = {}
            # fused stem backward (block 1 is a stride-2 depthwise layer fed by a 32-channel stem that is not a head
            # feature): its dL/d(stem activation) is never materialised (Engine.backward)
            self.fused_stem_np = -1
            if (len(specs) > 1 and specs[0]["cout"] == 32 and specs[1]["cin"] == 32 and tuple(specs[1]["stride"]) == (2, 2, 2)
                    and 0 not in self.feat_ids):
                np_f = L.msl_dwconv_s2_bwd_bnreduce_bww_num_partials(N, 32, *self.dims[0])
                small = self.y[0].numel() * 4 < (1 << 32)  # the fused kernel uses 32-bit byte offsets
                if np_f > 0 and small and 2 * 32 * np_f <= self.partials.numel():
                    self.fused_stem_np = np_f
                    self.partials_wf = torch.empty(32 * 27 * np_f, dtype=torch.float64, device=device)
                    self.w1_taps_t = torch.empty((27, 32), **f32)
        self.events = {}
        self.saved_input = None
        self.generation = 0
        self.dw_in_link = set()  # blocks whose depthwise weight gradient a channel link writes (no partials to fold)
        self.pw_fused_used = set()  # blocks whose pointwise weight-gradient slabs came with the fused pointwise backward
        self.dw_fused_used = set()  # blocks whose depthwise weight-gradient partials came with their bwd-data pass


class Engine:
    """Schedule: the main stream carries the dependency chain (backbone forward; activation-gradient chain in
    backward).  Two side streams take what is off that chain and too small to fill 256 CUs on its own:
    * ``heads``  — forward: the head convolutions of scales 3 and 5 run beside backbone blocks 4-7; backward: their
      weight / data gradients run beside the backward of blocks 7..4;
    * ``wgrad``  — every pointwise / depthwise weight gradient runs beside the next layer's data-gradient chain.
    Forks and joins are hipEvents recorded through the C ABI, so the schedule survives launch-program replay."""

    def __init__(self, model):
        self.model = model
        # ---- schedule options: plain attributes (defaults = what the measurements of rounds 1-3 chose; probes and
        #      `bench.py --opt name=value` set them; the library reads no environment variable for any of them) ------------
        self.multi_stream = True
        self.fold_bn = False      # fold EVERY BatchNorm into its consumers (slower: big layers have thousands of partials)
        # fold only the BatchNorms with at most this many statistics partials per channel into their depthwise / pointwise
        # consumers (one finalize launch less on the chain each; a folded stem BatchNorm - 1024 partials - costs block 1's
        # depthwise what the launch saves)
        self.fold_np_max = 512     # (65536 - the stem's 1024 partials too - gains another 2 us per step but costs the roofline
        #                             kernel, block 1's depthwise forward, 4.5 us: the launch stays)
        self.fold_np_max_pw = 64   # block 1's depthwise layer emits 64 partials per channel: folded by its pointwise consumer
        self.fold_bf16 = True     # the same for the bf16 step
        # bf16 path: head convolutions on the fp32 kernels from an fp32 feature copy ("f32": measured faster at every size
        # tried, and no second rounding of the head operands) or on the bf16 MFMA kernel from a channels-last bf16 copy
        # ("bf16", inference only)
        self.bf16_heads = "f32"
        self.bf16_materialize_beside = True  # bf16 pass: a scale's fp32 feature copy on the heads stream with its head convolution
        # set by a caller (FusedTrainer) while it records a step whose heads stream it orders itself, every step, behind the
        # previous step's optimiser: the forward pass then has no "fwd_start" record at the head of the chain
        self.prologue_presynced = False
        self.early_loss_fork = True  # fp32 backward: the heads stream is released by the loss launch itself (its stop event)
        self.eval_multi_stream_bf16 = True  # bf16 inference: heads of the earlier scales on the heads stream, as in fp32
        self.fold_bf16_feats = True  # bf16 pass: the feature maps' BatchNorms folded into the copy / the next depthwise layer too
        # 0: every block's weight gradients on the wgrad stream; 1: odd blocks on the heads stream (idle once the head
        # gradients are done); 2: three ways, the third on a stream of its own
        self.split_wgrad = 1
        self.wgrad_on_heads = None   # explicit set of blocks whose weight gradients go to the heads stream (probes)
        # an event record costs the chain ~6 us: the weight gradients of several blocks can share one (a set of block indices)
        # at the price of starting later - measured slower every time (the side streams are as critical as the chain)
        self.wgrad_record_at = None
        self.early_pw_bww = False    # start the pointwise weight gradient when dL/dy is final (one more record: slower)
        # the weight gradients of block i are enqueued after the chain launches of block i - wgrad_lag (host order only)
        self.wgrad_lag = 1
        # pointwise weight gradients of the tail blocks in ONE launch (single process only).  Paid while a fork cost the chain
        # 5-7 us; with stop-event forks (~1 us) four separate launches that start as their dL/dy arrive are 1.5 % faster
        self.batch_tail_pw = False
        self.batch_head_gpack = True  # head-gradient images of all scales in one launch
        self.prologue_on_side = True  # NaN-flag reset + head weight packing on the heads stream instead of the chain
        # the batched BatchNorm finalize (running statistics + backward vectors of the folded layers: nothing in the forward
        # pass reads them) on the weight-gradient stream, forked behind the last pointwise convolution and joined at the end
        # of the pass: beside the last head convolution instead of 9 us on the chain (a fork costs ~1 us since the stop events).
        # True: the fp32 pass (-0.4 % same-box); "all": the bf16 pass too (+0.4 %: its list is short); False: on the chain
        self.finalize_on_side = True
        self.side_stream_priority = 0
        self.fuse_stem = True     # block-1 / stem backward without materialising dL/d(stem activation)
        # eval mode: stem + block-1 depthwise convolution in one pass, the stem activation never in HBM (csrc/stemdw.hip)
        self.fuse_stem_eval = True
        self.channel_link = True  # per-channel backward links of the tail blocks in one launch each (csrc/chanlink.hip)
        self.fuse_pw_bwd = True   # whole pointwise backward of the big early block in one pass (csrc/pwfused.hip)
        self.fuse_dw_bww = True   # depthwise weight gradient of a big stride-2 block inside its bwd-data pass
        # a channel link also produces the depthwise weight gradient while it has at most this many waves per channel (with
        # more it is bound by instruction issue; the stride-2 forms above 4 waves also spill)
        self.link_bww_max_waves = 4
        self.link_bww_max_waves_s1 = 4
        self.extra = {}
        self.side = {}
        self.arena = None
        self.plans = {}
        self.prof, self.prof_tags = None, None
        feats = model.base.features
        specs = []
        for i, f in enumerate(feats):
            if i == 0:
                conv = f[0]
                specs.append(dict(kind="stem", cin=conv.in_channels, cout=conv.out_channels, stride=tuple(conv.stride)))
            else:
                specs.append(dict(kind="block", cin=f.conv1.in_channels, cout=f.conv2.out_channels,
                                  stride=tuple(f.conv1.stride)))
        self.layer_specs = specs

    # ------------------------------------------------------------------------------------------------
    def ensure_arena(self, device):
        if self.arena is None or self.arena.flat.device != device or not self.arena.owns(self.model):
            self.arena = ParamArena(self.model, device)
            self.plans = {}
        return self.arena

    def plan_for(self, x, need_grad):
        if getattr(self.model, "compute_dtype", "f32") == "bf16":
            return self._plan_bf16(x, need_grad)
        key = (x.shape[0], tuple(x.shape[2:]), x.device, need_grad)
        p = self.plans.get(key)
        if p is None:
            p = Plan(self, x.shape[0], x.shape[2:], x.device, need_grad)
            self.plans[key] = p
        return p

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def side_streams(self, device):
        """(heads, wgrad) torch streams for ``device`` (created once)."""
        key = (device.type, device.index)
        if key not in self.side:
            pr = self.side_stream_priority
            # (restricting the side streams to a CU subset with hipExtStreamCreateWithCUMask was measured 8-9 % slower for
            # every mask - half, quarter, three quarters of the chip - so they are ordinary streams)
            self.side[key] = (torch.cuda.Stream(device=device, priority=pr), torch.cuda.Stream(device=device, priority=pr))
        return self.side[key]

    def extra_stream(self, device):
        key = (device.type, device.index)
        if key not in self.extra:
            self.extra[key] = torch.cuda.Stream(device=device, priority=self.side_stream_priority)
        return self.extra[key]

    @staticmethod
    def _event(pl, name):
        """The plan's event ``name``.  Events that only order this GPU's own streams are created without the
        system-scope fence when ``_lib.DEVICE_SCOPE_EVENTS`` is set; the ``bucket_*`` events, which a communication
        stream (RCCL: peer-to-peer traffic) waits for, always keep it."""
        ev = pl.events.get(name)
        if ev is None:
            ev = pl.events[name] = _lib.new_event(device_only=not name.startswith("bucket_"))
        return ev

    @classmethod
    def _record(cls, pl, name, stream):
        ev = cls._event(pl, name)
        _lib.call("msl_event_record", ev, stream, tag="event")
        return ev

    @staticmethod
    def _wait(stream, ev):
        _lib.call("msl_stream_wait_event", stream, ev, tag="event")

    @classmethod
    def _fork(cls, pl, name, src, dst):
        """dst stream waits for everything enqueued on src so far."""
        ev = cls._event(pl, name)
        _lib.call("msl_event_record", ev, src, tag="event")
        _lib.call("msl_stream_wait_event", dst, ev, tag="event")

    # -- optional per-launch HIP-event timing (bench.py's roofline leg) -------------------------------------
    def start_profile(self, tags=None):
        """Record a HIP event pair (on the launch stream) around every tagged launch; tags=None -> all."""
        self.prof, self.prof_tags = {}, (None if tags is None else set(tags))

    def prof_all(self):
        """True while every launch is being timed (the trainer then uses the eager path, which tags every launch)."""
        return self.prof is not None and self.prof_tags is None

    def stop_profile(self):
        """-> {tag: [ms, ...]} (synchronises)."""
        torch.cuda.synchronize()
        out = {}
        for t, evs in (self.prof or {}).items():
            out[t] = [(_lib.elapsed_ms(e[1], e[2]) if e[0] == "c" else e[0].elapsed_time(e[1])) for e in evs]
            for e in evs:  # the native replay creates a fresh timed pair per tagged launch and replay: release them
                if e[0] == "c":
                    _lib.destroy_event(e[1])
                    _lib.destroy_event(e[2])
        self.prof = None
        return out

    def _k(self, tag, name, *args):
        if self.prof is None or (self.prof_tags is not None and tag not in self.prof_tags):
            _lib.call(name, *args, tag=tag)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call(name, *args, tag=tag)
        e1.record()
        self.prof.setdefault(tag, []).append((e0, e1))

    @staticmethod
    def _hook(cb, tag):
        if cb is not None:
            cb(tag)
            _lib.record_hook(lambda: cb(tag), tag=f"hook:{tag}")

    def _bn_fwd(self, bn, vec, partials, NP, count, training, st):
        C = vec.shape[1]
        if training:
            mom = 0.1 if bn.momentum is None else bn.momentum
            _lib.call("msl_bn_finalize", ptr(partials), NP, float(count), ptr(bn.weight), ptr(bn.bias),
                      ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked), mom, bn.eps,
                      ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]), C, st)
        else:
            _lib.call("msl_bn_eval_affine", ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
                      bn.eps, ptr(vec[0]), ptr(vec[1]), C, st)

    # ------------------------------------------------------------------------------------------------
    def forward(self, x, training, need_grad, want_features=False, nan_check=True, after_block=None):
        """x (N,Cin,D,H,W) fp32 on the GPU -> (locs (N,P,6), scores (N,P,n_classes)) [+ dict of feature maps].
        ``after_block``: {block index: callable} - side work (the trainer's target matching) to enqueue once that block's
        launches have been issued."""
        if not x.is_cuda:
            raise _lib.HipKernelError("mslesions3d_amd runs on the HIP device only (no CPU fallback): move the "
                                      "model and the input to 'cuda'")
        L = _lib.load()
        m = self.model
        if getattr(m, "compute_dtype", "f32") == "bf16":
            return self._forward_bf16(x, training=training, need_grad=need_grad, want_features=want_features,
                                      nan_check=nan_check, after_block=after_block)
        x = x.contiguous().float()
        self.ensure_arena(x.device)
        pl = self.plan_for(x, need_grad)
        pl.generation += 1
        pl.saved_input = x
        pl.trained_mode = training
        st = self._stream()
        stH = self.side_streams(x.device)[0].cuda_stream if self.multi_stream else st
        N = pl.N
        feats = m.base.features
        specs = self.layer_specs
        ev_pack = None
        if not training:
            # eval mode: (scale, shift) of all BatchNorms depend on parameters and running statistics only - one launch.  It
            # is the FIRST launch of the pass, in front of the prologue's fork: the fork's record then rides on it as a stop
            # event instead of being a packet of its own at the head of the chain (the heads stream still starts behind
            # everything the chain held before this pass)
            every = [(feats[0][1], pl.bn_y[0], pl.part_y[0], pl.np_y[0], 1.0)]
            for i in range(1, len(specs)):
                every += [(feats[i].bn1, pl.bn_z[i], pl.part_z[i], pl.np_z[i], 1.0),
                          (feats[i].bn2, pl.bn_y[i], pl.part_y[i], pl.np_y[i], 1.0)]
            self._finalize_all(pl, every, st, eval_mode=True)
        if self.multi_stream and self.prologue_on_side:
            # the NaN-flag reset and the MFMA-fragment copies of the head weights are needed 200 us into the pass (first head
            # convolution) and at its end (loss / NaN checks): on the heads stream they cost the dependency chain nothing
            # (in front of the stem they were a memset + a launch + two dispatch gaps, ~20 us).  The heads stream first waits
            # for everything the chain has done so far (the previous step's optimiser wrote the weights).
            if not self.prologue_presynced:  # (a caller that orders the heads stream itself: FusedTrainer)
                self._fork(pl, "fwd_start", st, stH)
            _lib.call("msl_fill_u32", ptr(pl.nan_flag), 0, 1, stH)
            self._pack_head_weights(pl, stH)
            ev_pack = self._record(pl, "head_pack_done", stH)
        else:
            _lib.call("msl_fill_u32", ptr(pl.nan_flag), 0, 1, st)
            self._pack_head_weights(pl, st)  # MFMA-fragment copies of the head weights, all scales, once per pass

        fold_max = (1 << 30) if self.fold_bn else self.fold_np_max
        folds = lambda NP: training and NP <= fold_max  # is the BatchNorm with NP partials folded into its consumers?
        # a depthwise output's only consumer is the pointwise GEMM of its block (one fold per wave, 32-64 channels wide):
        # a threshold of its own (MSL_FOLD_NP_MAX_PW; block 1's depthwise emits 64 partials)
        fold_max_pw = (1 << 30) if self.fold_bn else self.fold_np_max_pw
        folds_z = lambda NP: training and NP <= fold_max_pw
        deferred = []
        bn_layers = []  # (bn module, vector buffer, partials, NP, element count) of every BatchNorm, in order
        ev_bn_done = None

        def finalize_later(bn, vec, part, NP, count, name):
            """BatchNorm finalize (running stats + vectors for backward): all layers in ONE launch after the last
            block — nothing in the forward pass reads its outputs."""
            bn_layers.append((bn, vec, part, NP, count))

        def flush():
            nonlocal deferred
            for fn in deferred:
                fn()
            deferred = []

        # stem (features[0] = Conv3d + BN + ReLU)
        D, H, W = pl.in_dims
        sd, sh, sw = specs[0]["stride"]
        stem_dw = self._stem_dw_eval(specs, training, N, D, H, W)
        if stem_dw:
            self._k("stem_fwd", "msl_stem_dw_fwd_eval", ptr(x), ptr(feats[0][0].weight), ptr(pl.bn_y[0][0]), ptr(pl.bn_y[0][1]),
                    ptr(feats[1].conv1.weight), ptr(pl.z[1]), N, specs[0]["cin"], D, H, W, st)
        else:
            self._k("stem_fwd", "msl_stem_conv_fwd", ptr(x), ptr(feats[0][0].weight), ptr(pl.y[0]),
                    ptr(pl.part_y[0]) if training else None, N, specs[0]["cin"], D, H, W, sd, sh, sw, st)
        pl.stem_dw_eval = stem_dw
        od, oh, ow = pl.dims[0]
        def bn_done(bn, vec, part, NP, count, name, folded=None):
            if not training:
                return  # done above, for all layers at once
            if folds(NP) if folded is None else folded:
                finalize_later(bn, vec, part, NP, count, name)
            else:
                self._bn_fwd(bn, vec, part, NP, count, training, st)

        bn_done(feats[0][1], pl.bn_y[0], pl.part_y[0], pl.np_y[0], N * od * oh * ow, "stat_y0")
        out_feats = {}
        for i in range(1, len(specs)):
            sp, blk = specs[i], feats[i]
            pd, ph, pw = pl.dims[i - 1]
            D, H, W = pl.dims[i]
            S = D * H * W
            s = sp["stride"][0]
            bn_prev = feats[0][1] if i == 1 else feats[i - 1].bn2
            if i == 1 and stem_dw:
                pass  # z1 came with the stem
            elif folds(pl.np_y[i - 1]):
                self._k(f"dw_fwd{i}", "msl_dwconv_fwd_fold", ptr(pl.y[i - 1]), ptr(pl.part_y[i - 1]), pl.np_y[i - 1],
                        float(N * pd * ph * pw), ptr(bn_prev.weight), ptr(bn_prev.bias), bn_prev.eps, ptr(blk.conv1.weight),
                        ptr(pl.z[i]), ptr(pl.part_z[i]), N, sp["cin"], pd, ph, pw, s, st)
            else:
                self._k(f"dw_fwd{i}", "msl_dwconv_fwd", ptr(pl.y[i - 1]), ptr(pl.bn_y[i - 1][0]), ptr(pl.bn_y[i - 1][1]),
                        ptr(blk.conv1.weight), ptr(pl.z[i]), ptr(pl.part_z[i]) if training else None, N, sp["cin"], pd, ph,
                        pw, s, 0, st)
            flush()
            bn_done(blk.bn1, pl.bn_z[i], pl.part_z[i], pl.np_z[i], N * S, f"stat_z{i}", folded=folds_z(pl.np_z[i]))
            if folds_z(pl.np_z[i]):
                self._k(f"pw_fwd{i}", "msl_pwconv_fwd_fold", ptr(pl.z[i]), ptr(pl.part_z[i]), pl.np_z[i], float(N * S),
                        ptr(blk.bn1.weight), ptr(blk.bn1.bias), blk.bn1.eps, ptr(blk.conv2.weight), ptr(pl.y[i]),
                        ptr(pl.part_y[i]), N, sp["cin"], sp["cout"], S, st)
            else:
                self._k(f"pw_fwd{i}", "msl_pwconv_fwd", ptr(pl.z[i]), ptr(pl.bn_z[i][0]), ptr(pl.bn_z[i][1]),
                        ptr(blk.conv2.weight), ptr(pl.y[i]), ptr(pl.part_y[i]) if training else None, N, sp["cin"],
                        sp["cout"], S, st)
            flush()
            bn_done(blk.bn2, pl.bn_y[i], pl.part_y[i], pl.np_y[i], N * S, f"stat_y{i}")
            ev_feat = None  # the event a side stream waits for at this block, shared by everything forked here
            if i in pl.fpad:
                plain = None
                if want_features:
                    plain = torch.empty((N, sp["cout"], D, H, W), dtype=torch.float32, device=x.device)
                    out_feats[i] = plain

                def materialize(s_, i=i, sp=sp, blk=blk, plain=plain, D=D, H=H, W=W, S=S):
                    if folds(pl.np_y[i]):
                        self._k(f"materialize{i}", "msl_bn_relu_materialize_fold", ptr(pl.y[i]), ptr(pl.part_y[i]), pl.np_y[i],
                                float(N * S), ptr(blk.bn2.weight), ptr(blk.bn2.bias), blk.bn2.eps, ptr(plain), ptr(pl.fpad[i]),
                                N, sp["cout"], D, H, W, s_)
                    else:
                        self._k(f"materialize{i}", "msl_bn_relu_materialize", ptr(pl.y[i]), ptr(pl.bn_y[i][0]),
                                ptr(pl.bn_y[i][1]), ptr(plain), ptr(pl.fpad[i]), N, sp["cout"], D, H, W, s_)
                # this scale's head convolution only needs the feature map: run it beside the remaining blocks - and with it
                # the zero-haloed activation copy it reads (nothing on the chain reads that copy: the next depthwise layer
                # re-creates the activation from the raw output itself), when the BatchNorm is folded from the partials
                # (an unfolded one is finalised on the chain first, so its vectors are ready on either stream)
                last = i == len(specs) - 1
                if self.multi_stream and not last:
                    ev = ev_feat = self._record(pl, f"fwd_feat{i}", st)
                    deferred.append(lambda ev=ev, i=i, materialize=materialize: (self._wait(stH, ev), materialize(stH),
                                                                               self._head_forward(pl, i, stH)))
                else:
                    if last and bn_layers and self.multi_stream and self.finalize_on_side:
                        ev_bn_done = self._finalize_all_beside(pl, bn_layers, st)
                    materialize(st)
                    if ev_pack is not None:  # this scale's convolution runs on the chain: the packed weights come from stH
                        self._wait(st, ev_pack)
                        ev_pack = None
                    self._head_forward(pl, i, st)
            if after_block and i in after_block:
                after_block[i](ev_feat)  # (side work forked after this block waits for the same event: one record, not two)
        flush()
        if ev_bn_done is not None:
            self._wait(st, ev_bn_done)
        elif bn_layers:
            # running statistics and backward vectors of the folded BatchNorms: one launch
            self._finalize_all(pl, bn_layers, st)
        if self.multi_stream:
            self._fork(pl, "fwd_heads_done", stH, st)
        if nan_check:  # the fused training step lets the loss kernel set the flag instead (it reads both tensors anyway)
            _lib.call("msl_nan_flag2", ptr(pl.locs), pl.locs.numel(), 1, ptr(pl.scores), pl.scores.numel(), 2, ptr(pl.nan_flag), st)
        if want_features:
            return pl.locs, pl.scores, out_feats
        return pl.locs, pl.scores

    # ------------------------------------------------------------------------------------------------
    def _plan_bf16(self, x, need_grad=False):
        """Buffers of the bf16 activation path for this input shape: inference (BASELINE configs[3]) and, with
        ``need_grad``, the training step (configs[2])."""
        key = ("bf16", x.shape[0], tuple(x.shape[2:]), x.device, need_grad)
        pl = self.plans.get(key)
        if pl is not None:
            return pl
        L = _lib.load()
        m, specs = self.model, self.layer_specs

        class _P:
            pass
        pl = _P()
        pl.bf16 = True
        N, dev = x.shape[0], x.device
        pl.N, pl.in_dims = N, tuple(x.shape[2:])
        pl.dims, cur = [], tuple(x.shape[2:])
        for sp in specs:
            cur = tuple(conv_out(d, s) for d, s in zip(cur, sp["stride"]))
            pl.dims.append(cur)
        bf = dict(dtype=torch.bfloat16, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        f64 = dict(dtype=torch.float64, device=dev)
        ncls = m.n_classes
        pl.y = [torch.empty((N, sp["cout"]) + pl.dims[i], **bf) for i, sp in enumerate(specs)]
        pl.z = [None] + [torch.empty((N, specs[i]["cin"]) + pl.dims[i], **bf) for i in range(1, len(specs))]
        pl.bn_y = [torch.zeros((BN_ROWS, sp["cout"]), **f32) for sp in specs]
        pl.bn_z = [None] + [torch.zeros((BN_ROWS, specs[i]["cin"]), **f32) for i in range(1, len(specs))]
        # statistics partials of every BatchNorm (training): counts of the bf16 kernels
        pl.np_y, pl.np_z = [], [None]
        for i, sp in enumerate(specs):
            D, H, W = pl.dims[i]
            if i == 0:
                pl.np_y.append(L.msl_stem_conv_fwd_num_partials(N, D, H, W))
            else:
                pd, ph, pw = pl.dims[i - 1]
                pl.np_z.append(L.msl_dwconv_fwd_bf16_num_partials(N, sp["cin"], pd, ph, pw, sp["stride"][0]))
                pl.np_y.append(L.msl_pwconv_fwd_bf16_num_partials(N, D * H * W))
        pl.part_y = [torch.empty(2 * sp["cout"] * pl.np_y[i], **f64) for i, sp in enumerate(specs)]
        pl.part_z = [None] + [torch.empty(2 * specs[i]["cin"] * pl.np_z[i], **f64) for i in range(1, len(specs))]
        pl.feat_ids = list(m.aspect_ratios.keys())
        pl.prior_off, off = {}, 0
        for f in pl.feat_ids:
            pl.prior_off[f] = off
            D, H, W = pl.dims[f]
            off += D * H * W * m.boxes_per_location
        pl.P = off
        pl.f32_heads = need_grad or self.bf16_heads != "bf16"
        if pl.f32_heads:  # head convolutions on the fp32 kernels (faster at these sizes; the training step always)
            pl.fpad = {f: torch.zeros((N, specs[f]["cout"]) + tuple(d + 2 for d in pl.dims[f]), **f32) for f in pl.feat_ids}
            if not need_grad:
                pl.Wf, pl.Wb, pl.head_ws = {}, {}, {}
                for f in pl.feat_ids:
                    C = specs[f]["cout"]
                    ne = L.msl_head_packed_weight_elems(C, ncls)
                    pl.Wf[f], pl.Wb[f] = torch.empty(ne, **f32), torch.empty(ne, **f32)
                    pl.head_ws[f] = torch.empty(max(L.msl_head_fwd_workspace_bytes(N, C, *pl.dims[f], ncls) // 4, 1), **f32)
        else:
            pl.fpad_cl = {f: torch.zeros((N,) + tuple(d + 2 for d in pl.dims[f]) + (specs[f]["cout"],), **bf) for f in pl.feat_ids}
            pl.Wp = {f: torch.empty(L.msl_head_packed_weight_bf16_elems(specs[f]["cout"]), **bf) for f in pl.feat_ids}
        pl.locs = torch.empty((N, pl.P, 6), **f32)
        pl.scores = torch.empty((N, pl.P, ncls), **f32)
        pl.nan_flag = torch.zeros(1, dtype=torch.int32, device=dev)
        pl.events, pl.generation, pl.saved_input, pl.need_grad, pl.trained_mode = {}, 0, None, need_grad, False
        if need_grad:
            pl.g_y = [torch.empty_like(t) for t in pl.y]
            pl.g_z = [None] + [torch.empty_like(t) for t in pl.z[1:]]
            mt16 = 16 * ((12 + 2 * ncls + 15) // 16)
            pl.dO = {f: torch.zeros((N, mt16) + tuple(d + 2 for d in pl.dims[f]), **f32) for f in pl.feat_ids}
            pl.Wf, pl.Wb, pl.head_ws = {}, {}, {}
            for f in pl.feat_ids:
                C = specs[f]["cout"]
                ne = L.msl_head_packed_weight_elems(C, ncls)
                pl.Wf[f], pl.Wb[f] = torch.empty(ne, **f32), torch.empty(ne, **f32)
                ws = max(L.msl_head_fwd_workspace_bytes(N, C, *pl.dims[f], ncls),
                         L.msl_head_bwd_weight_workspace_bytes(N, C, *pl.dims[f], ncls))
                pl.head_ws[f] = torch.empty(max(ws // 4, 1), **f32)
            pl.head_nslabs = {f: L.msl_head_conv_bwd_weight_nslabs(N, specs[f]["cout"], *pl.dims[f], ncls) for f in pl.feat_ids}
            pl.ws_stem = torch.empty(max(L.msl_stem_conv_bwd_weight_workspace_bytes(specs[0]["cin"]) // 4, 1), **f32)
            pl.stem_nslabs = L.msl_stem_conv_bwd_weight_nslabs(N, *pl.in_dims, *specs[0]["stride"])
            pl.pw_nslabs, pl.pw_slabs, pl.dw_np, pl.dw_part = [0], [None], [0], [None]
            bnp = 0
            for i in range(1, len(specs)):
                D, H, W = pl.dims[i]
                pd, ph, pw = pl.dims[i - 1]
                ns = L.msl_pwconv_bwd_weight_bf16_nslabs(N, specs[i]["cin"], specs[i]["cout"], D * H * W)
                if ns < 1:
                    raise _lib.HipKernelError(f"msl_pwconv_bwd_weight_bf16_nslabs failed for block {i}")
                pl.pw_nslabs.append(ns)
                pl.pw_slabs.append(torch.empty(ns * specs[i]["cin"] * specs[i]["cout"], **f32) if ns > 1 else None)
                pl.dw_np.append(pl.np_z[i])
                pl.dw_part.append(torch.empty(specs[i]["cin"] * 27 * pl.np_z[i], **f64))
                bnp = max(bnp, 2 * specs[i]["cout"] * L.msl_bn_relu_bwd_bf16_num_partials(N, D * H * W),
                          2 * specs[i]["cin"] * L.msl_bn_relu_bwd_bf16_num_partials(N, D * H * W),
                          2 * specs[i]["cin"] * max(L.msl_dwconv_bwd_data_bnreduce_num_partials(N, specs[i]["cin"], pd, ph, pw), 0))
            d0 = pl.dims[0]
            bnp = max(bnp, 2 * specs[0]["cout"] * L.msl_bn_relu_bwd_bf16_num_partials(N, d0[0] * d0[1] * d0[2]))
            pl.fused_stem_np = -1
            if (len(specs) > 1 and specs[0]["cout"] == 32 and specs[1]["cin"] == 32 and tuple(specs[1]["stride"]) == (2, 2, 2)
                    and 0 not in pl.feat_ids and self.fuse_stem):
                # fused stem backward on bf16 storage: dL/d(stem activation) is never stored (as in the fp32 step)
                np_f = L.msl_dwconv_s2_bwd_bnreduce_bww_num_partials(N, 32, *pl.dims[0])
                if np_f > 0 and pl.y[0].numel() * 2 < (1 << 32):
                    pl.fused_stem_np = np_f
                    pl.partials_wf = torch.empty(32 * 27 * np_f, **f64)
                    pl.w1_taps_t = torch.empty((27, 32), **f32)
                    bnp = max(bnp, 2 * 32 * np_f)
                    pl.g_y[0] = None  # never materialised
            pl.partials = torch.empty(bnp, **f64)
            pl.grad_tables = {}
            pl.dw_in_link = set()
        self.plans[key] = pl
        return pl

    def _forward_bf16(self, x, training=False, need_grad=False, want_features=False, nan_check=True, after_block=None):
        """Forward with bf16 activations in HBM (csrc/bf16.hip + the bf16 head kernel): fp32 input volume, fp32 weights and
        BatchNorm vectors / statistics, bf16 everything in between, fp32 locs / scores out.  One stream (the head
        convolutions are issued in line)."""
        m, specs, feats = self.model, self.layer_specs, self.model.base.features
        x = x.contiguous().float()
        self.ensure_arena(x.device)
        pl = self._plan_bf16(x, need_grad)
        pl.generation += 1
        pl.saved_input, pl.trained_mode = x, training
        st = self._stream()
        # heads of the earlier scales beside the backbone: the training step, and inference too when the heads run on the fp32
        # kernels (eval_multi_stream_bf16; the bf16 inference pass used to be one stream: 86 us of feature copies and head
        # convolutions on the chain at 192^3 x 2)
        ms = self.multi_stream and (need_grad or (self.eval_multi_stream_bf16 and pl.f32_heads))
        stH = self.side_streams(x.device)[0].cuda_stream if ms else st
        N = pl.N
        ncls = m.n_classes
        side_prologue = ms and self.prologue_on_side and pl.f32_heads  # as in the fp32 forward: off the dependency chain
        ev_pack = None
        if not training:  # first launch of the pass, in front of the prologue's fork (see the fp32 pass)
            every = [(feats[0][1], pl.bn_y[0], pl.part_y[0], 1, 1.0)]
            for i in range(1, len(specs)):
                every += [(feats[i].bn1, pl.bn_z[i], pl.part_z[i], 1, 1.0), (feats[i].bn2, pl.bn_y[i], pl.part_y[i], 1, 1.0)]
            self._finalize_all(pl, every, st, eval_mode=True)
        if side_prologue:
            if not self.prologue_presynced:
                self._fork(pl, "fwd_start", st, stH)
            _lib.call("msl_fill_u32", ptr(pl.nan_flag), 0, 1, stH)
            self._pack_head_weights(pl, stH)
            ev_pack = self._record(pl, "head_pack_done", stH)
        else:
            _lib.call("msl_fill_u32", ptr(pl.nan_flag), 0, 1, st)
        if pl.f32_heads and not side_prologue:  # MFMA-fragment copies of the head weights (forward and, in training, bwd-data)
            self._pack_head_weights(pl, st)
        part = (lambda t: ptr(t)) if training else (lambda t: None)
        L = _lib.load()
        later = []  # BatchNorms folded into their consumer: running statistics + backward vectors in ONE launch at the end
        folded_feats = set()  # feature maps among them
        ev_bn_done = None
        D, H, W = pl.in_dims
        stem_dw = self._stem_dw_eval(specs, training, N, D, H, W)
        if stem_dw:
            self._k("stem_fwd", "msl_stem_dw_fwd_eval_bf16", ptr(x), ptr(feats[0][0].weight), ptr(pl.bn_y[0][0]),
                    ptr(pl.bn_y[0][1]), ptr(feats[1].conv1.weight), ptr(pl.z[1]), N, specs[0]["cin"], D, H, W, st)
        else:
            self._k("stem_fwd", "msl_stem_conv_fwd_bf16", ptr(x), ptr(feats[0][0].weight), ptr(pl.y[0]), part(pl.part_y[0]), N,
                    specs[0]["cin"], D, H, W, *specs[0]["stride"], st)
        pl.stem_dw_eval = stem_dw
        od, oh, ow = pl.dims[0]
        out_feats = {}
        for i in range(1, len(specs)):
            sp, blk = specs[i], feats[i]
            pd, ph, pw = pl.dims[i - 1]
            D, H, W = pl.dims[i]
            S = D * H * W
            bn_prev = feats[0][1] if i == 1 else feats[i - 1].bn2
            cnt_prev = N * pd * ph * pw
            # the consumer rebuilds (scale, shift) from the producer's partials when they are few (<= fold_np_max: every wave /
            # workgroup repeats the sum) - no finalize launch between the two; feature maps need the vectors anyway
            # (a feature map whose fp32 copy was made from the partials - folded_feats - is folded here too)
            fold_y = (training and self.fold_bf16 and pl.np_y[i - 1] <= max(64, self.fold_np_max)
                      and ((i - 1) not in pl.feat_ids or (i - 1) in folded_feats)
                      and L.msl_dwconv_wave_num_partials(N, sp["cin"], pd, ph, pw, sp["stride"][0]) > 0)
            if training and not fold_y and (i - 1) in folded_feats:  # the copy folded, this consumer cannot: finalize after all
                self._bn_fwd(bn_prev, pl.bn_y[i - 1], pl.part_y[i - 1], pl.np_y[i - 1], cnt_prev, True, st)
                later.pop([id(e[1]) for e in later].index(id(pl.bn_y[i - 1])))
            elif training and not fold_y and (i - 1) not in pl.feat_ids:  # (a feature map's vectors exist already)
                self._bn_fwd(bn_prev, pl.bn_y[i - 1], pl.part_y[i - 1], pl.np_y[i - 1], cnt_prev, True, st)
            if i == 1 and stem_dw:
                pass  # z1 came with the stem
            elif fold_y:
                if (i - 1) not in folded_feats:
                    later.append((bn_prev, pl.bn_y[i - 1], pl.part_y[i - 1], pl.np_y[i - 1], cnt_prev))
                self._k(f"dw_fwd{i}", "msl_dwconv_fwd_wave_bf16_fold", ptr(pl.y[i - 1]), ptr(pl.part_y[i - 1]), pl.np_y[i - 1],
                        float(cnt_prev), ptr(bn_prev.weight), ptr(bn_prev.bias), bn_prev.eps, ptr(blk.conv1.weight), ptr(pl.z[i]),
                        ptr(pl.part_z[i]), N, sp["cin"], pd, ph, pw, sp["stride"][0], st)
            else:
                self._k(f"dw_fwd{i}", "msl_dwconv_fwd_bf16", ptr(pl.y[i - 1]), ptr(pl.bn_y[i - 1][0]), ptr(pl.bn_y[i - 1][1]),
                        ptr(blk.conv1.weight), ptr(pl.z[i]), part(pl.part_z[i]), N, sp["cin"], pd, ph, pw, sp["stride"][0], st)
            fold_z = training and self.fold_bf16 and pl.np_z[i] <= 64 and sp["cin"] <= 1024
            if fold_z:
                later.append((blk.bn1, pl.bn_z[i], pl.part_z[i], pl.np_z[i], N * S))
                self._k(f"pw_fwd{i}", "msl_pwconv_fwd_bf16_fold", ptr(pl.z[i]), ptr(pl.part_z[i]), pl.np_z[i], float(N * S),
                        ptr(blk.bn1.weight), ptr(blk.bn1.bias), blk.bn1.eps, ptr(blk.conv2.weight), ptr(pl.y[i]),
                        ptr(pl.part_y[i]), N, sp["cin"], sp["cout"], S, st)
            else:
                if training:
                    self._bn_fwd(blk.bn1, pl.bn_z[i], pl.part_z[i], pl.np_z[i], N * S, True, st)
                self._k(f"pw_fwd{i}", "msl_pwconv_fwd_bf16", ptr(pl.z[i]), ptr(pl.bn_z[i][0]), ptr(pl.bn_z[i][1]),
                        ptr(blk.conv2.weight), ptr(pl.y[i]), part(pl.part_y[i]), N, sp["cin"], sp["cout"], S, st)
            if after_block and i in after_block:
                after_block[i](None)
            if training and later and ms and self.finalize_on_side == "all" and i == len(specs) - 1:
                ev_bn_done = self._finalize_all_beside(pl, later, st)  # as in the fp32 pass (bf16: measured +0.4 %, so opt-in)
            # a feature map's BatchNorm: folded into the fp32 copy (and into the next depthwise layer) from the partials when
            # they are few - no finalize launch on the chain; else finalised here (the copy below reads the vectors)
            fold_feat = (training and i in pl.feat_ids and pl.f32_heads and self.fold_bf16 and self.fold_bf16_feats
                         and pl.np_y[i] <= max(64, self.fold_np_max) and not want_features)
            if fold_feat:
                folded_feats.add(i)
                later.append((blk.bn2, pl.bn_y[i], pl.part_y[i], pl.np_y[i], N * S))
            elif training and i in pl.feat_ids:
                self._bn_fwd(blk.bn2, pl.bn_y[i], pl.part_y[i], pl.np_y[i], N * S, True, st)
            if i in pl.feat_ids and pl.f32_heads:
                # the fp32 zero-haloed copy only feeds this scale's head convolution: it goes to the heads stream with it
                # (as in the fp32 pass; 6-8 us per scale off the chain) unless the caller wants the feature map back
                beside = ms and i != len(specs) - 1
                on_side = beside and not want_features and self.bf16_materialize_beside
                if on_side:
                    self._fork(pl, f"fwd_feat{i}", st, stH)
                if fold_feat:
                    self._k(f"materialize{i}", "msl_bn_relu_materialize_bf16_pad32_fold", ptr(pl.y[i]), ptr(pl.part_y[i]),
                            pl.np_y[i], float(N * S), ptr(blk.bn2.weight), ptr(blk.bn2.bias), blk.bn2.eps, ptr(pl.fpad[i]), N,
                            sp["cout"], D, H, W, stH if on_side else st)
                else:
                    self._k(f"materialize{i}", "msl_bn_relu_materialize_bf16_pad32", ptr(pl.y[i]), ptr(pl.bn_y[i][0]),
                            ptr(pl.bn_y[i][1]), ptr(pl.fpad[i]), N, sp["cout"], D, H, W, stH if on_side else st)
                if want_features:
                    out_feats[i] = pl.fpad[i][:, :, 1:-1, 1:-1, 1:-1].clone()
                if beside:  # beside the next blocks, on the heads stream
                    if not on_side:
                        self._fork(pl, f"fwd_feat{i}", st, stH)
                    self._head_forward(pl, i, stH)
                else:
                    if ev_pack is not None:  # the packed weights come from the heads stream
                        self._wait(st, ev_pack)
                        ev_pack = None
                    self._head_forward(pl, i, st)
            elif i in pl.feat_ids:
                plain = None
                if want_features:
                    plain = out_feats[i] = torch.empty((N, sp["cout"], D, H, W), dtype=torch.float32, device=x.device)
                self._k(f"materialize{i}", "msl_bn_relu_materialize_bf16", ptr(pl.y[i]), ptr(pl.bn_y[i][0]), ptr(pl.bn_y[i][1]),
                        ptr(plain), ptr(pl.fpad_cl[i]), N, sp["cout"], D, H, W, st)
                k = pl.feat_ids.index(i)
                lc, cc = m.pred_convs.loc_convs[k], m.pred_convs.cl_convs[k]
                self._k(f"head_pack{i}", "msl_head_pack_weights_bf16", ptr(lc.weight), ptr(cc.weight), ptr(pl.Wp[i]), sp["cout"],
                        ncls, st)
                self._k(f"head_fwd{i}", "msl_head_conv_fwd_bf16", ptr(pl.fpad_cl[i]), ptr(pl.Wp[i]), ptr(lc.bias), ptr(cc.bias),
                        ptr(pl.locs), ptr(pl.scores), N, sp["cout"], D, H, W, pl.P, pl.prior_off[i], ncls, st)
        if ev_bn_done is not None:
            self._wait(st, ev_bn_done)
        elif training and later:
            self._finalize_all(pl, later, st)
        if ms:
            self._fork(pl, "fwd_heads_done", stH, st)
        if nan_check:
            _lib.call("msl_nan_flag2", ptr(pl.locs), pl.locs.numel(), 1, ptr(pl.scores), pl.scores.numel(), 2, ptr(pl.nan_flag), st)
        if want_features:
            return pl.locs, pl.scores, out_feats
        return pl.locs, pl.scores

    @staticmethod
    def _bn_bwd_bf16_fused(N, S):
        """One launch (the channel's data held in registers) instead of reduce + finalize + apply?"""
        return (N * S <= 32768 and S % 8 == 0) or N * S <= 4096

    def _bn_bwd_bf16(self, g, y, vec, bn_name, N, C, S, pl, st, pre_np=None):
        """In place: g (bf16, = dL/d relu(bn(y))) becomes dL/dy; dgamma / dbeta into the gradient arena.
        ``pre_np``: the producer of g already left that many reduce partials per channel in pl.partials."""
        L = _lib.load()
        gv = self.arena.grad_views
        if pre_np is None and self._bn_bwd_bf16_fused(N, S):
            self._k("bn_bwd_fused:" + bn_name, "msl_bn_relu_bwd_fused_bf16", ptr(g), ptr(y), ptr(vec), ptr(gv[bn_name + ".weight"]),
                    ptr(gv[bn_name + ".bias"]), ptr(g), N, C, S, st)
            return
        NP = pre_np if pre_np is not None else L.msl_bn_relu_bwd_bf16_num_partials(N, S)
        if pre_np is None:
            self._k("bn_bwd_reduce:" + bn_name, "msl_bn_relu_bwd_reduce_bf16", ptr(g), ptr(y), ptr(vec[0]), ptr(vec[1]), ptr(vec[2]),
                    ptr(vec[3]), ptr(pl.partials), N, C, S, st)
        _lib.call("msl_bn_bwd_finalize", ptr(pl.partials), NP, float(N * S), ptr(gv[bn_name + ".weight"]),
                  ptr(gv[bn_name + ".bias"]), ptr(vec[4]), ptr(vec[5]), C, st)
        self._k("bn_bwd_apply:" + bn_name, "msl_bn_relu_bwd_apply_bf16", ptr(g), ptr(y), ptr(vec), ptr(g), N, C, S, st)

    def _backward_bf16(self, pl, dlocs, dscores, on_bucket_ready=None):
        """Backward of the bf16 training step: activation gradients are bf16 tensors, weight gradients fp32 partial sums
        folded by the batched reduction.  Same three-stream schedule as the fp32 step (dependency chain on the main stream,
        head gradients of the earlier scales on the heads stream, weight gradients alternating over the two side streams);
        a data-parallel reducer gets each bucket as soon as the launches producing it are enqueued (as in the fp32 step)."""
        m, specs, feats = self.model, self.layer_specs, self.model.base.features
        gv = self.arena.grad_views
        st = self._stream()
        ms = self.multi_stream
        if ms:
            sH, sW = self.side_streams(pl.locs.device)
            stH, stW = sH.cuda_stream, sW.cuda_stream
        else:
            stH = stW = st
        N, ncls = pl.N, m.n_classes
        prepacked = dlocs is None  # the loss kernel wrote the head-gradient images itself (MultiBoxLoss._run_loss_pack)
        if not prepacked:
            dlocs, dscores = dlocs.contiguous(), dscores.contiguous()
        last = len(specs) - 1
        L = _lib.load()
        pre_np, linked = None, False
        if last not in pl.feat_ids:
            raise RuntimeError("the last backbone feature must feed a head")
        wanted = getattr(on_bucket_ready, "stages", None)
        groups = self._bucket_groups(on_bucket_ready) if wanted else {}
        report = self._reporter(pl, on_bucket_ready, groups, st, stH, stW, stW, ms)

        def head(f, s_data, s_weight, done=None):
            C = specs[f]["cout"]
            D, H, W = pl.dims[f]
            if not packed:
                self._k(f"head_gpack{f}", "msl_head_grad_pack", ptr(dlocs), ptr(dscores), ptr(pl.dO[f]), N, D, H, W, pl.P,
                        pl.prior_off[f], ncls, s_data)
            self._k(f"head_bwd{f}", "msl_head_conv_bwd_data_bf16", ptr(pl.dO[f]), ptr(pl.Wb[f]), ptr(pl.g_y[f]), N, C, D, H, W,
                    ncls, s_data)
            if done is not None:
                self._record(pl, done, s_data)
            if s_weight != s_data:
                self._wait(s_weight, self._record(pl, f"head_dO{f}", s_data))
            self._k(f"head_bww{f}", "msl_head_conv_bwd_weight", ptr(pl.dO[f]), ptr(pl.fpad[f]), None, None, None, None,
                    ptr(pl.head_ws[f]), N, C, D, H, W, ncls, s_weight)

        side_feats = [f for f in pl.feat_ids if f != last] if ms else []
        packed = prepacked or (bool(side_feats) and self.batch_head_gpack)
        if packed and not prepacked:
            self._head_gpack_batch(pl, dlocs, dscores, st)
        if side_feats:
            self._fork(pl, "bwd_loss_ready", st, stH)
        head(last, st, stW)  # its data gradient starts the chain
        for f in reversed(side_feats) if ms else [f for f in pl.feat_ids if f != last]:
            head(f, stH, stH, done=f"head_done{f}" if ms else None)
        report("heads", join_heads=True)
        for i in range(last, 0, -1):
            sp = specs[i]
            D, H, W = pl.dims[i]
            S = D * H * W
            pd, ph, pw = pl.dims[i - 1]
            s = sp["stride"][0]
            name = f"base.features.{i}"
            if not linked:  # (a channel link of block i+1 has already turned g_y[i] into dL/dy_i)
                self._bn_bwd_bf16(pl.g_y[i], pl.y[i], pl.bn_y[i], name + ".bn2", N, sp["cout"], S, pl, st, pre_np=pre_np)
            pre_np, linked = None, False
            self._k(f"pw_bwd{i}", "msl_pwconv_bwd_data_bf16", ptr(pl.g_y[i]), ptr(feats[i].conv2.weight), ptr(pl.g_z[i]), N,
                    sp["cin"], sp["cout"], S, st)
            accumulate = 1 if (i - 1) in pl.feat_ids else 0  # the heads already wrote their share
            Sp = pd * ph * pw
            fused_stem = i == 1 and pl.fused_stem_np > 0
            big_producer = s == 2 and pw % 4 == 0 and not self._bn_bwd_bf16_fused(N, Sp)
            # the per-channel link of the tail blocks in one launch, as in the fp32 step (csrc/chanlink.hip on bf16 storage)
            link_nw = L.msl_block_bwd_channel_link_supported(N, pd, ph, pw, s) if self.channel_link else 0
            link = link_nw > 0 and not fused_stem and not big_producer and i >= 2
            link_bww = link and (link_nw <= self.link_bww_max_waves or (s == 1 and link_nw <= self.link_bww_max_waves_s1))
            if not link:
                self._bn_bwd_bf16(pl.g_z[i], pl.z[i], pl.bn_z[i], name + ".bn1", N, sp["cin"], S, pl, st)
            if accumulate and (i - 1) in side_feats:
                self._wait(st, pl.events[f"head_done{i - 1}"])
            if link:
                prev = f"base.features.{i - 1}.bn2"
                self._k(f"link{i}", "msl_block_bwd_channel_link_bf16", ptr(pl.g_z[i]), ptr(pl.z[i]), ptr(pl.bn_z[i]),
                        ptr(feats[i].conv1.weight), ptr(pl.y[i - 1]), ptr(pl.bn_y[i - 1]), ptr(pl.g_y[i - 1]),
                        ptr(gv[name + ".bn1.weight"]), ptr(gv[name + ".bn1.bias"]), ptr(gv[prev + ".weight"]),
                        ptr(gv[prev + ".bias"]), ptr(gv[name + ".conv1.weight"]) if link_bww else None, N, sp["cin"], pd, ph,
                        pw, s, accumulate, st)
                linked = True
                if link_bww:
                    pl.dw_in_link.add(i)
            sX = st
            if ms:  # both weight gradients of the block on a side stream, once dL/dy_i and dL/dz_i are final
                sX = stW if i % 2 else stH
                self._wait(sX, self._record(pl, f"dz{i}", st))
            if link:
                pass
            elif fused_stem:
                # one pass over (dL/dz_1, y_0): the stem's BatchNorm-backward sums + this block's depthwise weight gradient;
                # the stem weight gradient below rebuilds dL/d(stem activation) from dL/dz_1 on the fly
                self._k("dw_bwd1", "msl_dwconv_s2_bwd_bnreduce_bww_bf16", ptr(pl.g_z[1]), ptr(feats[1].conv1.weight), ptr(pl.y[0]),
                        ptr(pl.bn_y[0]), ptr(pl.partials), ptr(pl.partials_wf), ptr(pl.w1_taps_t), N, 32, pd, ph, pw, st)
                pre_np = pl.fused_stem_np
            elif s == 2 and pw % 4 == 0 and not self._bn_bwd_bf16_fused(N, Sp):
                # big producer layer: emit the BatchNorm-backward partials of y_{i-1} while its gradient is in registers
                pre_np = L.msl_dwconv_bwd_data_bnreduce_num_partials(N, sp["cin"], pd, ph, pw)
                self._k(f"dw_bwd{i}", "msl_dwconv_bwd_data_s2_patch_bf16", ptr(pl.g_z[i]), ptr(feats[i].conv1.weight),
                        ptr(pl.g_y[i - 1]), ptr(pl.y[i - 1]), ptr(pl.bn_y[i - 1]), ptr(pl.partials), N, sp["cin"], pd, ph, pw,
                        accumulate, st)
            else:
                self._k(f"dw_bwd{i}", "msl_dwconv_bwd_data_bf16", ptr(pl.g_z[i]), ptr(feats[i].conv1.weight), ptr(pl.g_y[i - 1]),
                        N, sp["cin"], pd, ph, pw, s, accumulate, st)
            out = pl.pw_slabs[i] if pl.pw_nslabs[i] > 1 else gv[name + ".conv2.weight"]
            self._k(f"pw_bww{i}", "msl_pwconv_bwd_weight_slabs_bf16", ptr(pl.g_y[i]), ptr(pl.z[i]), ptr(pl.bn_z[i][0]),
                    ptr(pl.bn_z[i][1]), ptr(out), N, sp["cin"], sp["cout"], S, sX)
            if not fused_stem and not link_bww:  # (its partials came with the fused pass: pl.partials_wf / the link wrote dW)
                self._k(f"dw_bww{i}", "msl_dwconv_bwd_weight_bf16", ptr(pl.g_z[i]), ptr(pl.y[i - 1]), ptr(pl.bn_y[i - 1][0]),
                        ptr(pl.bn_y[i - 1][1]), ptr(pl.dw_part[i]), N, sp["cin"], pd, ph, pw, s, sX)
            report(i, join_heads=True)  # (odd blocks put their weight gradients on the heads stream)
        # stem: BatchNorm-backward sums, then the weight gradient with the BatchNorm backward applied on load
        od, oh, ow = pl.dims[0]
        S0 = od * oh * ow
        D, H, W = pl.in_dims
        vec = pl.bn_y[0]
        NP = pre_np if pre_np is not None else L.msl_bn_relu_bwd_bf16_num_partials(N, S0)
        if pre_np is None:
            self._k("bn_bwd_reduce:stem", "msl_bn_relu_bwd_reduce_bf16", ptr(pl.g_y[0]), ptr(pl.y[0]), ptr(vec[0]), ptr(vec[1]),
                    ptr(vec[2]), ptr(vec[3]), ptr(pl.partials), N, specs[0]["cout"], S0, st)
        if pl.fused_stem_np > 0:
            _lib.call("msl_bn_bwd_finalize_coef", ptr(pl.partials), NP, float(N * S0), ptr(gv["base.features.0.1.weight"]),
                      ptr(gv["base.features.0.1.bias"]), ptr(vec), specs[0]["cout"], st)
            self._k("stem_bww", "msl_stem_conv_bwd_weight_fused_bf16", ptr(pl.g_z[1]), ptr(pl.w1_taps_t), ptr(pl.y[0]), ptr(vec),
                    ptr(pl.saved_input), None, ptr(pl.ws_stem), N, specs[0]["cin"], D, H, W, *specs[0]["stride"], st)
        else:
            _lib.call("msl_bn_bwd_finalize", ptr(pl.partials), NP, float(N * S0), ptr(gv["base.features.0.1.weight"]),
                      ptr(gv["base.features.0.1.bias"]), ptr(vec[4]), ptr(vec[5]), specs[0]["cout"], st)
            self._k("stem_bww", "msl_stem_conv_bwd_weight_bnapply_bf16", ptr(pl.g_y[0]), ptr(pl.y[0]), ptr(vec), ptr(pl.saved_input),
                    None, ptr(pl.ws_stem), N, specs[0]["cin"], D, H, W, *specs[0]["stride"], st)
        if ms:  # every gradient is complete once the side streams have been joined
            self._fork(pl, "bwd_join_w", stW, st)
            self._fork(pl, "bwd_join_h", stH, st)
        if not groups:  # single process: ONE reduction launch for every layer's partial sums, in front of the optimiser
            self._grad_reduce(pl, "all", None, st)
        report(0)

    def _stem_dw_eval(self, specs, training, N, D, H, W):
        """Eval mode: may the stem and block 1's depthwise convolution run as one launch (csrc/stemdw.hip)?"""
        if training or not self.fuse_stem_eval or len(specs) < 2:
            return False
        if tuple(specs[0]["stride"]) != (2, 2, 2) or tuple(specs[1]["stride"]) != (2, 2, 2) or specs[0]["cout"] != 32:
            return False
        return bool(_lib.load().msl_stem_dw_fwd_eval_supported(N, specs[0]["cin"], D, H, W))

    def _finalize_all_beside(self, pl, bn_layers, st):
        """_finalize_all on the weight-gradient stream, ordered behind what ``st`` holds so far (the last pointwise
        convolution: every layer's partial sums exist) -> the event the chain waits for at the end of the pass."""
        stW = self.side_streams(pl.locs.device)[1].cuda_stream
        self._fork(pl, "fwd_bn_parts", st, stW)
        self._finalize_all(pl, bn_layers, stW)
        return self._record(pl, "fwd_bn_done", stW)

    def _finalize_all(self, pl, bn_layers, st, eval_mode=False):
        """One launch for the running statistics and backward vectors of the listed BatchNorms (table built once per
        plan and mode); ``eval_mode``: one launch for their eval-mode (scale, shift) instead."""
        import ctypes
        L = _lib.load()
        key = (eval_mode,) + tuple((ptr(bn.weight), ptr(bn.running_mean)) for bn, *_ in bn_layers)
        if eval_mode:
            if getattr(pl, "bn_eval_table_key", None) != key:
                keep = (getattr(pl, "bn_table", None), getattr(pl, "bn_table_key", None), getattr(pl, "bn_table_n", None),
                        getattr(pl, "bn_table_channels", None))
                pl.bn_table_key = None
                self._finalize_all(pl, bn_layers, None)  # builds pl.bn_table for this list (no launch: st is None)
                pl.bn_eval_table, pl.bn_eval_n, pl.bn_eval_channels, pl.bn_eval_table_key = pl.bn_table, pl.bn_table_n, pl.bn_table_channels, key
                pl.bn_table, pl.bn_table_key, pl.bn_table_n, pl.bn_table_channels = keep
            _lib.call("msl_bn_eval_affine_batch", ptr(pl.bn_eval_table), pl.bn_eval_n, pl.bn_eval_channels, st, tag="bn_eval_all")
            return
        if getattr(pl, "bn_table_key", None) != key:
            esz = L.msl_bn_finalize_entry_bytes()
            host = (ctypes.c_ubyte * (esz * len(bn_layers)))()
            first = 0
            for k, (bn, vec, part, NP, count) in enumerate(bn_layers):
                C = vec.shape[1]
                mom = 0.1 if bn.momentum is None else bn.momentum
                _lib.check(L.msl_bn_finalize_table_set(ctypes.addressof(host), k, first, ptr(part), NP, float(count),
                                                       ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean),
                                                       ptr(bn.running_var), ptr(bn.num_batches_tracked), mom, bn.eps,
                                                       ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]), C),
                           "msl_bn_finalize_table_set")
                first += C
            pl.bn_table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(vec.device)
            pl.bn_table_key, pl.bn_table_n, pl.bn_table_channels = key, len(bn_layers), first
        if st is None:
            return
        _lib.call("msl_bn_finalize_batch", ptr(pl.bn_table), pl.bn_table_n, pl.bn_table_channels, st, tag="bn_finalize_all")

    def _pw_bww_batch(self, pl, rows, N, st):
        """msl_pwconv_bwd_weight_slabs for several tail blocks in one launch; rows = (dy, z, scale, shift, out, cin, cout, S)."""
        import ctypes
        key = tuple(rows)
        if getattr(pl, "pwb_key", None) != key:
            n = len(rows)
            P, I = ctypes.c_void_p * n, ctypes.c_int * n
            pl.pwb_args = tuple(P(*[r[c] for r in rows]) for c in range(5)) + tuple(I(*[r[c] for r in rows]) for c in range(5, 8))
            pl.pwb_key = key
        a = pl.pwb_args  # host arrays: kept alive by the plan (a recorded launch program points at them)
        self._k("pw_bww_tail", "msl_pwconv_bwd_weight_slabs_batch", *[ctypes.addressof(x) for x in a], len(rows), N, st)

    def _pack_head_weights(self, pl, st):
        import ctypes
        m = self.model
        key = tuple(ptr(c.weight) for c in m.pred_convs.loc_convs) + tuple(ptr(c.weight) for c in m.pred_convs.cl_convs)
        if getattr(pl, "pack_key", None) != key:
            n = len(pl.feat_ids)
            P = ctypes.c_void_p * n
            pl.pack_args = (P(*[ptr(c.weight) for c in m.pred_convs.loc_convs]), P(*[ptr(c.weight) for c in m.pred_convs.cl_convs]),
                            P(*[ptr(pl.Wf[f]) for f in pl.feat_ids]), P(*[ptr(pl.Wb[f]) for f in pl.feat_ids]),
                            (ctypes.c_int * n)(*[self.layer_specs[f]["cout"] for f in pl.feat_ids]))
            pl.pack_key = key
        a = pl.pack_args  # host arrays: kept alive by the plan (a recorded launch program points at them)
        n = len(pl.feat_ids)
        for g0 in range(0, n, 4):  # the batch entry holds four scales (`--prediction_layers "1 2 3 5 7"`: two launches)
            self._k("head_pack", "msl_head_pack_weights_batch", *[ctypes.addressof(a[c]) + 8 * g0 for c in range(4)],
                    ctypes.addressof(a[4]) + 4 * g0, min(4, n - g0), m.n_classes, st)

    def _head_forward(self, pl, f, st):
        m = self.model
        ncls = m.n_classes
        k = pl.feat_ids.index(f)
        lc, cc = m.pred_convs.loc_convs[k], m.pred_convs.cl_convs[k]
        C = self.layer_specs[f]["cout"]
        D, H, W = pl.dims[f]
        self._k(f"head_fwd{f}", "msl_head_conv_fwd", ptr(pl.fpad[f]), ptr(pl.Wf[f]), ptr(lc.bias), ptr(cc.bias), ptr(pl.locs),
                ptr(pl.scores), ptr(pl.head_ws[f]), pl.N, C, D, H, W, pl.P, pl.prior_off[f], ncls, st)

    # ------------------------------------------------------------------------------------------------
    def _bn_bwd(self, g, y, vec, bn_name, count, N, C, S, pl, st, pre_np=None, apply=True, coef=False, partials=None):
        """In place: g (= dL/d relu(bn(y))) becomes dL/dy; writes dgamma/dbeta into the gradient arena.
        ``pre_np``: the producer of g already emitted the reduce partials (pl.partials, that many per channel);
        ``apply=False``: the consumer applies the BatchNorm backward itself while loading (only c1/c2 are produced)."""
        L = _lib.load()
        gv = self.arena.grad_views
        part = pl.partials if partials is None else partials  # (``partials``: where the producer left its pre_np sums)
        if pre_np is not None and coef:
            _lib.call("msl_bn_bwd_finalize_coef", ptr(part), pre_np, float(count), ptr(gv[bn_name + ".weight"]),
                      ptr(gv[bn_name + ".bias"]), ptr(vec), C, st)
            return
        if pre_np is not None and apply and pre_np <= 256:
            self._k("bn_bwd_apply:" + bn_name, "msl_bn_relu_bwd_finalize_apply", ptr(part), pre_np, float(count), ptr(g),
                    ptr(y), ptr(vec), ptr(gv[bn_name + ".weight"]), ptr(gv[bn_name + ".bias"]), ptr(g), N, C, S, st)
            return
        if pre_np is not None:
            _lib.call("msl_bn_bwd_finalize", ptr(pl.partials), pre_np, float(count), ptr(gv[bn_name + ".weight"]),
                      ptr(gv[bn_name + ".bias"]), ptr(vec[4]), ptr(vec[5]), C, st)
            if apply:
                self._k("bn_bwd_apply:" + bn_name, "msl_bn_relu_bwd_apply", ptr(g), ptr(y), ptr(vec[0]), ptr(vec[1]), ptr(vec[2]),
                        ptr(vec[3]), ptr(vec[4]), ptr(vec[5]), ptr(g), N, C, S, st)
            return
        if N * S <= 65536:  # small per-channel data: reduce + finalize + apply in one launch
            self._k("bn_bwd_fused:" + bn_name, "msl_bn_relu_bwd_fused", ptr(g), ptr(y), ptr(vec[0]), ptr(vec[1]),
                    ptr(vec[2]), ptr(vec[3]), ptr(gv[bn_name + ".weight"]), ptr(gv[bn_name + ".bias"]), ptr(g), N, C, S, st)
            return
        NP = L.msl_bn_relu_bwd_num_partials(N, S)
        self._k("bn_bwd_reduce:" + bn_name, "msl_bn_relu_bwd_reduce", ptr(g), ptr(y), ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]),
                  ptr(pl.partials), N, C, S, st)
        if NP <= 256:
            self._k("bn_bwd_apply:" + bn_name, "msl_bn_relu_bwd_finalize_apply", ptr(pl.partials), NP, float(count), ptr(g),
                    ptr(y), ptr(vec), ptr(gv[bn_name + ".weight"]), ptr(gv[bn_name + ".bias"]), ptr(g), N, C, S, st)
            return
        _lib.call("msl_bn_bwd_finalize", ptr(pl.partials), NP, float(count), ptr(gv[bn_name + ".weight"]),
                  ptr(gv[bn_name + ".bias"]), ptr(vec[4]), ptr(vec[5]), C, st)
        self._k("bn_bwd_apply:" + bn_name, "msl_bn_relu_bwd_apply", ptr(g), ptr(y), ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]),
                  ptr(vec[4]), ptr(vec[5]), ptr(g), N, C, S, st)

    def _head_gpack_batch(self, pl, dlocs, dscores, st):
        """msl_head_grad_pack of every scale in one launch."""
        import ctypes
        key = (ptr(dlocs), ptr(dscores)) + tuple(ptr(pl.dO[f]) for f in pl.feat_ids)
        if getattr(pl, "gpack_key", None) != key:
            n = len(pl.feat_ids)
            I = ctypes.c_int * n
            pl.gpack_args = ((ctypes.c_void_p * n)(*[ptr(pl.dO[f]) for f in pl.feat_ids]),) + tuple(
                I(*[pl.dims[f][a] for f in pl.feat_ids]) for a in range(3)) + (I(*[pl.prior_off[f] for f in pl.feat_ids]),)
            pl.gpack_key = key
        a = pl.gpack_args  # host arrays: kept alive by the plan (a recorded launch program points at them)
        n = len(pl.feat_ids)
        for g0 in range(0, n, 4):  # four scales per launch (msl_head_grad_pack_batch)
            self._k("head_gpack_all", "msl_head_grad_pack_batch", ptr(dlocs), ptr(dscores), ctypes.addressof(a[0]) + 8 * g0,
                    *[ctypes.addressof(x) + 4 * g0 for x in a[1:]], min(4, n - g0), pl.N, pl.P, self.model.n_classes, st)

    def _head_backward(self, pl, f, dlocs, dscores, st, data_done_event=None, data=True, weight=True, packed=False):
        m, gv, ncls = self.model, self.arena.grad_views, self.model.n_classes
        k = pl.feat_ids.index(f)
        C = self.layer_specs[f]["cout"]
        D, H, W = pl.dims[f]
        pre = f"pred_convs.loc_convs.{k}", f"pred_convs.cl_convs.{k}"
        if data:
            if not packed:
                self._k(f"head_gpack{f}", "msl_head_grad_pack", ptr(dlocs), ptr(dscores), ptr(pl.dO[f]), pl.N, D, H, W, pl.P,
                        pl.prior_off[f], ncls, st)
            self._k(f"head_bwd{f}", "msl_head_conv_bwd_data", ptr(pl.dO[f]), ptr(pl.Wb[f]), ptr(pl.g_y[f]), pl.N, C, D, H, W, ncls, st)
            if data_done_event is not None:  # the activation-gradient chain only waits for the data gradient
                _lib.call("msl_event_record", data_done_event, st, tag="event")
        if not weight:
            return
        # weight and bias slabs stay in head_ws[f] (folded by the batched gradient reduction)
        self._k(f"head_bww{f}", "msl_head_conv_bwd_weight", ptr(pl.dO[f]), ptr(pl.fpad[f]), None, None, None, None,
                ptr(pl.head_ws[f]), pl.N, C, D, H, W, ncls, st)

    def _reporter(self, pl, on_bucket_ready, groups, st, stH, stW, stX, ms):
        """-> report(stage, join_heads=False): tell the data-parallel reducer that every launch producing the gradients of
        ``stage`` ('heads', 7, ..., 0) has been enqueued - after making the stream the exchange will run on wait for the
        streams that produced them and folding the stage's partial sums there."""
        wanted = getattr(on_bucket_ready, "stages", None)

        def report(stage, join_heads=False):
            if on_bucket_ready is None or (wanted is not None and stage not in wanted):
                return
            comm = getattr(on_bucket_ready, "comm_stream", None)
            if comm is not None and stage == 0 and getattr(on_bucket_ready, "final_on_main", False):
                # every stream has just been joined into the chain: the last bucket is exchanged right here.  Collectives
                # of one communicator must not overlap each other, so the chain first waits for the communication stream
                # (whose buckets were issued ~0.3 ms ago: the wait is normally satisfied on arrival)
                self._fork(pl, "bucket_tail", comm.cuda_stream, st)
            elif comm is not None:
                # the exchange has a stream of its own: IT waits for the streams that produced the bucket, the
                # dependency chain never does (joining the weight-gradient stream into the chain three times per step
                # cost 0.19 ms of a 1.0 ms step: tools/probes/dp_host_cost.py)
                dst = comm.cuda_stream
                self._fork(pl, f"bucket_m{stage}", st, dst)
                if ms:
                    if dst != stW:
                        self._fork(pl, f"bucket_w{stage}", stW, dst)
                    if join_heads and dst != stH:
                        self._fork(pl, f"bucket_h{stage}", stH, dst)
                    if stX != stW and dst != stX:
                        self._fork(pl, f"bucket_x{stage}", stX, dst)
            elif ms:  # the exchange follows the main stream: bring the side streams' work in first
                self._fork(pl, f"bucket_w{stage}", stW, st)
                if join_heads:
                    self._fork(pl, f"bucket_h{stage}", stH, st)
                if stX != stW:
                    self._fork(pl, f"bucket_x{stage}", stX, st)
            if stage in groups:  # fold the partial sums of this stage's buckets where their exchange will run
                comm2 = getattr(on_bucket_ready, "comm_stream", None)
                on_main = comm2 is None or (stage == 0 and getattr(on_bucket_ready, "final_on_main", False))
                self._grad_reduce(pl, stage, groups[stage], st if on_main else comm2.cuda_stream)
            # the waits above are part of the launch program; the reducer must not repeat them at the torch level - told per
            # call (eager and replayed alike), never through state that outlives the call (a flag left set
            # by an fp32 step made a later bf16 step on the same reducer skip its wait)
            presync = comm is not None and hasattr(on_bucket_ready, "presynced")

            def fire(tag, presync=presync):
                if presync:
                    on_bucket_ready.presynced = True
                try:
                    on_bucket_ready(tag)
                finally:
                    if presync:
                        on_bucket_ready.presynced = False
            if getattr(on_bucket_ready, "native", False):
                on_bucket_ready(stage)  # issues C-ABI calls itself (recorded like any other launch): no Python hook
            else:
                self._hook(fire, stage)
        return report

    def backward(self, pl, dlocs, dscores, on_bucket_ready=None):
        """Given dL/dlocs, dL/dscores, fill the flat gradient arena.  ``on_bucket_ready(stage)`` is called right
        after the launches that complete a gradient stage ('heads', 7, ..., 0) have been enqueued; if it has a
        ``stages`` attribute only those stages are reported (and the side streams are joined first)."""
        if not pl.need_grad or not pl.trained_mode:
            raise RuntimeError("backward needs a train-mode forward made with gradients enabled")
        if getattr(pl, "bf16", False):
            return self._backward_bf16(pl, dlocs, dscores, on_bucket_ready)
        m = self.model
        gv = self.arena.grad_views
        st = self._stream()
        ms = self.multi_stream
        if ms:
            sH, sW = self.side_streams(pl.locs.device)
            stH, stW = sH.cuda_stream, sW.cuda_stream
            stX = self.extra_stream(pl.locs.device).cuda_stream if self.split_wgrad > 1 else stW
        else:
            stH = stW = stX = st
        N = pl.N
        specs = self.layer_specs
        feats = m.base.features
        prepacked = dlocs is None  # the loss kernel wrote the head-gradient images itself (MultiBoxLoss._run_loss_pack)
        if not prepacked:
            dlocs = dlocs.contiguous()
            dscores = dscores.contiguous()
        wanted = getattr(on_bucket_ready, "stages", None)
        # data parallel: the partial sums of a gradient bucket are folded when the bucket's last stage has been enqueued
        # (stage -> parameter names); single process: one reduction at the very end
        groups = self._bucket_groups(on_bucket_ready) if wanted else {}

        report = self._reporter(pl, on_bucket_ready, groups, st, stH, stW, stX, ms)

        # heads: the last scale feeds the chain immediately (main stream); the earlier scales are only needed when
        # the chain reaches their feature map, so they run on the heads stream beside blocks 7..4
        last = len(specs) - 1
        side_feats = [f for f in pl.feat_ids if f != last] if ms else []
        packed = prepacked or (bool(side_feats) and self.batch_head_gpack)
        if packed and not prepacked:
            self._head_gpack_batch(pl, dlocs, dscores, st)
        L = _lib.load()
        pre_np = None  # set when the producer of the next activation gradient also produced its BatchNorm partials
        linked = False  # set when a channel link already applied the BatchNorm backward of the next layer's output gradient
        pending = []  # side-stream launches, issued one layer late so that the chain's launches always go first
        sinks = []    # (layer, closure) of the weight-gradient launches still to be issued
        tail = []     # blocks whose pointwise weight gradients share one launch (deepest first), and their arguments
        if ms and not groups and self.batch_tail_pw:
            for i in range(last, 0, -1):
                D_, H_, W_ = pl.dims[i]
                if N * D_ * H_ * W_ <= 4096 and L.msl_pwconv_bwd_weight_batchable(N, specs[i]["cin"], specs[i]["cout"], D_ * H_ * W_) == 1:
                    tail.append(i)
            tail = tail[:4] if len(tail) >= 2 else []
        tail_args = []
        ev_loss = None
        if side_feats and self.early_loss_fork:
            # the earlier scales' gradients (heads stream) need the head-gradient images - written by the launch in front
            # of this point (the loss kernel, or the batched pack), whose stop event this record becomes: no packet on the
            # chain, and the heads stream starts one launch earlier than behind the chain's own head bwd-data
            self._fork(pl, "bwd_loss_ready", st, stH)
        for f in pl.feat_ids:  # the chain's own scale first: its data gradient starts the backward chain
            if f not in side_feats:
                if ms:
                    self._head_backward(pl, f, dlocs, dscores, st, weight=False, packed=packed)
                    ev = ev_loss = self._record(pl, f"head_dO{f}", st)
                    pending.append(lambda f=f, ev=ev: (self._wait(stW, ev),
                                                       self._head_backward(pl, f, dlocs, dscores, stW, data=False)))
                else:
                    self._head_backward(pl, f, dlocs, dscores, st, packed=packed)
        if side_feats and not self.early_loss_fork:
            # (sharing the record behind the chain's own head bwd-data instead - one launch later, one record fewer - was
            # measured 1-2 % SLOWER: the heads stream's gradients are as critical as the chain here)
            self._fork(pl, "bwd_loss_ready", st, stH)
        for f in reversed(side_feats):  # the deeper scale is needed first
            pending.append(lambda f=f: self._head_backward(pl, f, dlocs, dscores, stH, self._event(pl, f"head_done{f}"), packed=packed))
        if wanted is not None and "heads" in wanted:
            for fn in pending:
                fn()
            pending = []
        report("heads", join_heads=True)
        for i in range(last, 0, -1):
            sp = specs[i]
            D, H, W = pl.dims[i]
            S = D * H * W
            pd, ph, pw = pl.dims[i - 1]
            s = sp["stride"][0]
            name = f"base.features.{i}"
            if i not in pl.fpad and i == last:
                raise RuntimeError("the last backbone feature must feed a head")
            # y_i = pw(relu(bn1(z_i))): BN2 backward, then the two GEMMs; z_i = dw(relu(bn(y_{i-1}))): BN1 backward,
            # then the depthwise gradients.  The host enqueues the dependency chain (main stream) FIRST and the two
            # weight gradients (wgrad stream, waiting on events recorded in the chain) afterwards: the chain is made of
            # ~10 us kernels, so any launch queued in front of its next link shows up as idle time.
            # big early block whose producer left the BatchNorm2-backward sums: BatchNorm backward of y_i (applied on load),
            # bwd-data GEMM, BatchNorm1-backward sums of z_i and the pointwise weight gradient in ONE pass (csrc/pwfused.hip)
            pw_fused = (self.fuse_pw_bwd and not linked and pre_np is not None and i in pl.pw_fused
                        and 2 * sp["cout"] * pre_np <= pl.partials.numel())
            z_np = None
            if pw_fused:
                slabs_f, part_f, z_np = pl.pw_fused[i]
                self._k(f"pw_bwd_fused{i}", "msl_pwconv_bwd_fused", ptr(pl.g_y[i]), ptr(pl.y[i]), ptr(pl.bn_y[i]), ptr(pl.partials),
                        pre_np, float(N * S), ptr(gv[name + ".bn2.weight"]), ptr(gv[name + ".bn2.bias"]),
                        ptr(feats[i].conv2.weight), ptr(pl.z[i]), ptr(pl.bn_z[i]), ptr(pl.g_z[i]), ptr(part_f), ptr(slabs_f), N,
                        sp["cin"], sp["cout"], S, st)
                pl.pw_fused_used.add(i)
            elif not linked:  # (a channel link of block i+1 has already turned g_y[i] into dL/dy_i)
                self._bn_bwd(pl.g_y[i], pl.y[i], pl.bn_y[i], name + ".bn2", N * S, N, sp["cout"], S, pl, st, pre_np=pre_np)
            pre_np, linked = None, False
            # dL/dy_i is final here: the pointwise weight gradient may start two or three chain kernels before dL/dz_i is
            rec_here = self.wgrad_record_at is None or i in self.wgrad_record_at or i == 1
            ev_dy = self._record(pl, f"dy{i}", st) if ms and self.early_pw_bww and rec_here and not pw_fused else None
            if not pw_fused:
                self._k(f"pw_bwd{i}", "msl_pwconv_bwd_data", ptr(pl.g_y[i]), ptr(feats[i].conv2.weight), ptr(pl.g_z[i]), N,
                        sp["cin"], sp["cout"], S, st)
            accumulate = 1 if (i - 1) in pl.fpad else 0  # the heads already wrote their share
            Sp = pd * ph * pw
            np_red = L.msl_dwconv_bwd_data_bnreduce_num_partials(N, sp["cin"], pd, ph, pw) if (s == 2 and N * Sp > 65536) else -1
            fused_stem = i == 1 and self.fuse_stem and pl.fused_stem_np > 0
            # tail of the network: BatchNorm1 backward of z_i, depthwise bwd-data (+ the heads' share) and BatchNorm2 backward
            # of y_{i-1} are all per channel and a channel's population fits one workgroup: ONE launch (csrc/chanlink.hip)
            link_nw = L.msl_block_bwd_channel_link_supported(N, pd, ph, pw, s) if self.channel_link else 0
            link = link_nw > 0 and not fused_stem and np_red <= 0 and not pw_fused
            # (with <= 4 waves per channel the link takes the depthwise weight gradient along: no launch for it below)
            link_bww = link and (link_nw <= self.link_bww_max_waves or (s == 1 and link_nw <= self.link_bww_max_waves_s1))
            # a block whose pointwise weight gradient rides in the tail's batched launch (issued with the shallowest of them) and
            # whose depthwise weight gradient the link produces has nothing waiting for dL/dz_i: no event record on the chain
            idle_sink = (link_bww and i in tail and i != tail[-1]) or (pw_fused and fused_stem)
            rec_here = rec_here and not idle_sink
            if not link:
                self._bn_bwd(pl.g_z[i], pl.z[i], pl.bn_z[i], name + ".bn1", N * S, N, sp["cin"], S, pl, st, pre_np=z_np,
                             partials=pl.pw_fused[i][1] if pw_fused else None)
                ev_dz = self._record(pl, f"dz{i}", st) if ms and rec_here else None
            if accumulate and (i - 1) in side_feats:
                self._wait(st, pl.events[f"head_done{i - 1}"])
            # big producer layers: emit the BatchNorm-backward partials of y_{i-1} while its gradient is in registers
            ev_red = None
            dw_fused = False
            if link:
                prev = "base.features.0.1" if i == 1 else f"base.features.{i - 1}.bn2"
                self._k(f"link{i}", "msl_block_bwd_channel_link", ptr(pl.g_z[i]), ptr(pl.z[i]), ptr(pl.bn_z[i]),
                        ptr(feats[i].conv1.weight), ptr(pl.y[i - 1]), ptr(pl.bn_y[i - 1]), ptr(pl.g_y[i - 1]),
                        ptr(gv[name + ".bn1.weight"]), ptr(gv[name + ".bn1.bias"]), ptr(gv[prev + ".weight"]),
                        ptr(gv[prev + ".bias"]), ptr(gv[name + ".conv1.weight"]) if link_bww else None, N, sp["cin"], pd, ph,
                        pw, s, accumulate, st)
                if link_bww:
                    pl.dw_in_link.add(i)
                ev_dz = self._record(pl, f"dz{i}", st) if ms and rec_here else None
                linked = True
            elif fused_stem:
                # one pass over (dL/dz_1, y_0): BatchNorm-backward sums of the stem + the depthwise weight gradient;
                # the stem weight gradient below rebuilds dL/d(stem activation) from dL/dz_1 on the fly
                vp = pl.bn_y[0]
                self._k("dw_bwd1", "msl_dwconv_s2_bwd_bnreduce_bww", ptr(pl.g_z[1]), ptr(feats[1].conv1.weight), ptr(pl.y[0]),
                        ptr(vp[0]), ptr(vp[1]), ptr(vp[2]), ptr(vp[3]), ptr(pl.partials), ptr(pl.partials_wf), ptr(pl.w1_taps_t),
                        N, 32, pd, ph, pw, st)
                pre_np = pl.fused_stem_np
            elif np_red > 0 and self.fuse_dw_bww and i in pl.dw_fused_part:
                # big stride-2 producer layer: bwd-data, the BatchNorm-backward sums of y_{i-1} AND this block's depthwise
                # weight gradient in one pass over (dL/dz_i, y_{i-1}) - the weight-gradient launch on the side stream (the same
                # 38 MB read again, 35 us beside the chain at block 2) disappears
                vp = pl.bn_y[i - 1]
                part_w, np_w = pl.dw_fused_part[i]
                self._k(f"dw_bwd{i}", "msl_dwconv_s2_bwd_data_bnreduce_bww", ptr(pl.g_z[i]), ptr(feats[i].conv1.weight),
                        ptr(pl.g_y[i - 1]), ptr(pl.y[i - 1]), ptr(vp[0]), ptr(vp[1]), ptr(vp[2]), ptr(vp[3]), ptr(pl.partials),
                        ptr(part_w), N, sp["cin"], pd, ph, pw, accumulate, st)
                pre_np = np_w
                dw_fused = True
                pl.dw_fused_used.add(i)
            elif np_red > 0 and 2 * sp["cin"] * np_red <= pl.partials.numel():
                vp = pl.bn_y[i - 1]
                self._k(f"dw_bwd{i}", "msl_dwconv_bwd_data_bnreduce", ptr(pl.g_z[i]), ptr(feats[i].conv1.weight),
                        ptr(pl.g_y[i - 1]), ptr(pl.y[i - 1]), ptr(vp[0]), ptr(vp[1]), ptr(vp[2]), ptr(vp[3]), ptr(pl.partials),
                        N, sp["cin"], pd, ph, pw, s, accumulate, st)
                pre_np = np_red
            else:
                self._k(f"dw_bwd{i}", "msl_dwconv_bwd_data", ptr(pl.g_z[i]), ptr(feats[i].conv1.weight), ptr(pl.g_y[i - 1]),
                        N, sp["cin"], pd, ph, pw, s, accumulate, st)
            def wgrads(ev_dz, i=i, sp=sp, S=S, pd=pd, ph=ph, pw=pw, s=s, name=name,
                       fused_stem=fused_stem or link_bww or dw_fused, ev_red=ev_red, ev_dy=ev_dy, idle_sink=idle_sink,
                       pw_fused=pw_fused):
                # split_wgrad: the heads stream is idle once the head gradients are done - odd blocks go there
                streams = [stW, stH, stX]
                sX = streams[i % (self.split_wgrad + 1)] if ms else st
                if ms and self.wgrad_on_heads is not None:  # experiment knob: explicit list of blocks for the heads stream
                    sX = stH if i in self.wgrad_on_heads else stW
                if ms and not idle_sink:  # the pointwise gradient needs dL/dy_i, the depthwise one dL/dz_i
                    self._wait(sX, ev_dy if ev_dy is not None else ev_dz)
                # partial sums only: slabs / fp64 partials stay in this layer's own buffers until Engine._grad_reduce
                out = pl.pw_slabs[i] if pl.pw_nslabs[i] > 1 else gv[name + ".conv2.weight"]
                if pw_fused:
                    pass  # (its slabs came with the fused pointwise backward)
                elif i in tail:
                    tail_args.append((ptr(pl.g_y[i]), ptr(pl.z[i]), ptr(pl.bn_z[i][0]), ptr(pl.bn_z[i][1]), ptr(out), sp["cin"],
                                      sp["cout"], S))
                    if i == tail[-1]:  # the shallowest: every dL/dy of the group is final (same chain, earlier)
                        self._pw_bww_batch(pl, tail_args, N, sX)
                else:
                    self._k(f"pw_bww{i}", "msl_pwconv_bwd_weight_slabs", ptr(pl.g_y[i]), ptr(pl.z[i]), ptr(pl.bn_z[i][0]),
                            ptr(pl.bn_z[i][1]), ptr(out), N, sp["cin"], sp["cout"], S, sX)
                if ms and ev_dy is not None and not idle_sink:
                    self._wait(sX, ev_dz)
                if fused_stem:
                    return  # its partials came with the fused stem backward pass (pl.partials_wf) / the channel link wrote dW
                self._k(f"dw_bww{i}", "msl_dwconv_bwd_weight", ptr(pl.g_z[i]), ptr(pl.y[i - 1]), ptr(pl.bn_y[i - 1][0]),
                        ptr(pl.bn_y[i - 1][1]), None, ptr(pl.dw_part[i]), N, sp["cin"], pd, ph, pw, s, sX)

            # issue what earlier layers left for the side streams (the head gradients the chain waits for: one layer late;
            # the weight gradients, which nothing in the step waits for: wgrad_lag layers late), then queue this layer's
            for fn in pending:
                fn()
            pending = []
            if ms:
                sinks.append([i, wgrads, ev_dz if not idle_sink else "none"])
                if ev_dz is not None:  # blocks that did not record wait for the next record of the chain instead
                    for ent in sinks:
                        if ent[2] is None:
                            ent[2] = ev_dz
                while sinks and sinks[0][2] is not None and sinks[0][0] >= i + self.wgrad_lag:
                    ent = sinks.pop(0)
                    ent[1](ent[2])
            else:
                wgrads(None)
            if wanted is not None and i in wanted:
                if any(e is None for _, _, e in sinks):
                    raise RuntimeError("MSL_WGRAD_RECORD_AT must contain every block that completes a gradient bucket")
                for _, fn, e in sinks:
                    fn(e)
                sinks = []
            report(i, join_heads=self.split_wgrad > 0)
        # stem
        od, oh, ow = pl.dims[0]
        S0 = od * oh * ow
        D, H, W = pl.in_dims
        sd, sh, sw = specs[0]["stride"]
        if pre_np is not None and specs[0]["cout"] == 32:
            # reduce came with the depthwise bwd-data above; the apply is fused into the stem weight gradient
            fused = self.fuse_stem and pl.fused_stem_np > 0
            self._bn_bwd(pl.g_y[0], pl.y[0], pl.bn_y[0], "base.features.0.1", N * S0, N, 32, S0, pl, st, pre_np=pre_np,
                         apply=False, coef=fused)
            if fused:
                self._k("stem_bww", "msl_stem_conv_bwd_weight_fused", ptr(pl.g_z[1]), ptr(pl.w1_taps_t), ptr(pl.y[0]),
                        ptr(pl.bn_y[0]), ptr(pl.saved_input), None, ptr(pl.ws_stem), N,
                        specs[0]["cin"], D, H, W, sd, sh, sw, st)
            else:
                self._k("stem_bww", "msl_stem_conv_bwd_weight_bnapply", ptr(pl.g_y[0]), ptr(pl.y[0]), ptr(pl.bn_y[0]),
                        ptr(pl.saved_input), None, ptr(pl.ws_stem), N, specs[0]["cin"], D, H, W, sd, sh, sw, st)
        else:
            if not linked:  # (tiny inputs: block 1's channel link has already applied the stem's BatchNorm backward)
                self._bn_bwd(pl.g_y[0], pl.y[0], pl.bn_y[0], "base.features.0.1", N * S0, N, specs[0]["cout"], S0, pl, st,
                             pre_np=pre_np)
            self._k("stem_bww", "msl_stem_conv_bwd_weight", ptr(pl.g_y[0]), ptr(pl.saved_input),
                    None, ptr(pl.ws_stem), N, specs[0]["cin"], D, H, W, sd, sh, sw, st)
        for fn in pending:
            fn()
        for _, fn, e in sinks:
            fn(e)
        if ms:  # every gradient is complete once the side streams have been joined
            self._fork(pl, "bwd_join_w", stW, st)
            self._fork(pl, "bwd_join_h", stH, st)
            if stX != stW:
                self._fork(pl, "bwd_join_x", stX, st)
        if not groups:  # single process: ONE reduction launch for every layer's partial sums, right in front of the optimiser
            self._grad_reduce(pl, "all", None, st)
        report(0)

    def _bucket_groups(self, reducer):
        """stage -> set of parameter names whose gradient bucket completes at that stage (GradBucketReducer layout)."""
        groups = {}
        for stage, buckets in reducer.trigger.items():
            names = set()
            for k in buckets:
                lo, hi = reducer.ranges[k]
                names |= {n for n, (off, cnt) in self.arena.offsets.items() if n not in self.arena.no_grad_names and lo <= off < hi}
            groups[stage] = names
        return groups

    def _grad_reduce(self, pl, key, names, st):
        """One launch that folds the partial weight-gradient sums (pointwise slabs, depthwise fp64 partials, head and stem
        slabs) of the parameters in ``names`` (None: all) into the gradient arena (table built once per plan and key)."""
        import ctypes
        L = _lib.load()
        fold = getattr(pl, "loss_fold", None)
        tkey = (key, None if fold is None else ptr(fold[0]))
        ent = pl.grad_tables.get(tkey)
        if ent is None:
            gv = self.arena.grad_views
            specs, m = self.layer_specs, self.model
            ncls = m.n_classes
            rows = []  # (kind, src, dst, dst2, nslabs, count, stride, p0, p1, p2)
            want = lambda n: names is None or n in names
            for k, f in enumerate(pl.feat_ids):
                lw, cw = f"pred_convs.loc_convs.{k}.weight", f"pred_convs.cl_convs.{k}.weight"
                if want(lw) or want(cw):
                    if not (want(lw) and want(cw)):
                        raise RuntimeError("a gradient bucket boundary separates the two head convolutions of one scale")
                    C = specs[f]["cout"]
                    mt = (12 + 2 * ncls + 15) // 16
                    slab = (C // 16) * 27 * mt * 256
                    ns = pl.head_nslabs[f]
                    rows.append((3, pl.head_ws[f], gv[lw], gv[cw], ns, slab, slab, C, mt, 12 + 2 * ncls))
                    bias = pl.head_ws[f][ns * slab:]  # [ns][16*mt] row sums of dO: rows 0-11 loc, 12.. cls
                    rows.append((0, bias, gv[lw[:-6] + "bias"], None, ns, 12, 16 * mt, 0, 0, 0))
                    rows.append((0, bias[12:], gv[cw[:-6] + "bias"], None, ns, 2 * ncls, 16 * mt, 0, 0, 0))
            fused = self.fuse_stem and pl.fused_stem_np > 0
            for i in range(len(specs) - 1, 0, -1):
                name = f"base.features.{i}"
                cnt = specs[i]["cin"] * specs[i]["cout"]
                if want(name + ".conv2.weight") and i in getattr(pl, "pw_fused_used", ()):
                    rows.append((0, pl.pw_fused[i][0], gv[name + ".conv2.weight"], None, pl.pw_fused[i][2], cnt, cnt, 0, 0, 0))
                elif want(name + ".conv2.weight") and pl.pw_nslabs[i] > 1:
                    rows.append((0, pl.pw_slabs[i], gv[name + ".conv2.weight"], None, pl.pw_nslabs[i], cnt, cnt, 0, 0, 0))
                if want(name + ".conv1.weight") and i not in getattr(pl, "dw_in_link", ()):  # (a channel link wrote it)
                    if i == 1 and fused:
                        rows.append((1, pl.partials_wf, gv[name + ".conv1.weight"], None, pl.fused_stem_np, specs[i]["cin"] * 27, 0, 0, 0, 0))
                    elif i in getattr(pl, "dw_fused_used", ()):
                        part_w, np_w = pl.dw_fused_part[i]
                        rows.append((1, part_w, gv[name + ".conv1.weight"], None, np_w, specs[i]["cin"] * 27, 0, 0, 0, 0))
                    else:
                        rows.append((1, pl.dw_part[i], gv[name + ".conv1.weight"], None, pl.dw_np[i], specs[i]["cin"] * 27, 0, 0, 0, 0))
            if want("base.features.0.0.weight"):
                K = specs[0]["cin"] * 27
                nt = (K + 31) // 32
                rows.append((2, pl.ws_stem, gv["base.features.0.0.weight"], None, pl.stem_nslabs, 1024 * nt, 1024 * nt, K, nt, 0))
            fold = getattr(pl, "loss_fold", None)
            if fold is not None and key in ("all", 0):  # the loss values of MultiBoxLoss._run_loss_pack ride along (kind 4)
                parts, nparts, loss_out, npos = fold
                rows.append((4, parts, loss_out, npos, nparts, 1, 0, 0, 0, 0))
            esz = L.msl_grad_reduce_entry_bytes()
            host = (ctypes.c_ubyte * (esz * max(len(rows), 1)))()
            first = 0
            owner = []  # table entry of every workgroup
            for k, (kind, src, dst, dst2, ns, cnt, stride, p0, p1, p2) in enumerate(rows):
                nb = L.msl_grad_reduce_table_set(ctypes.addressof(host), k, first, kind, ptr(src), ptr(dst), ptr(dst2), ns, cnt,
                                                 stride, p0, p1, p2)
                if nb < 0:
                    raise _lib.HipKernelError(f"msl_grad_reduce_table_set failed for entry {k} (kind {kind})")
                first += nb
                owner += [k] * nb
            dev = self.arena.grad.device
            table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev) if rows else None
            ent = pl.grad_tables[tkey] = (table, len(rows), first, torch.tensor(owner, dtype=torch.int32, device=dev))
        table, n, blocks, owner = ent
        if n:
            self._k(f"grad_reduce:{key}", "msl_grad_reduce_batch_indexed", ptr(table), n, ptr(owner), blocks, st)

    def check_nan(self, pl):
        """One host sync: raise like ssd3d.py:258-261 if the forward produced NaN."""
        self.raise_on_nan_flag(int(pl.nan_flag.item()))

    @staticmethod
    def raise_on_nan_flag(flag):
        if flag & 2:
            raise Exception("Oh no not this NaN error again... (forward SSD), CLASSES_SCORES is nan!")
        if flag & 1:
            raise Exception("Oh no not this NaN error again... (forward SSD), LOCS is nan!")
