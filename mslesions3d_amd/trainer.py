"""Minimal training / prediction loops replacing PyTorch-Lightning's ``Trainer`` (reference ``train.py:171-188``,
``predict.py:257-263``) for the one model of this repository.

``FusedTrainer.step`` is the hot loop of the headline metric (training volumes/s): forward -> matching ->
MultiBox loss -> closed-form loss gradient -> backward -> (RCCL bucketed all-reduce, overlapped) -> fused Adam
-> cosine LR step, as one fixed sequence of C-ABI launches with no host synchronisation inside
(ssd3d.py:467-531 + :704-722).  ``LSSD3D.training_step`` + ``loss.backward()`` + ``optimizer.step()`` is the
API-compatible (autograd) route through the same kernels.
"""
import collections
import os

import torch

from . import _lib
from ._lib import ptr
from .optim import CosineAnnealingLR, FusedAdam
from .parallel import GradBucketReducer
from .ssd3d import MultiBoxLoss


class _StagedOptimizer:
    """Single-process counterpart of the data-parallel reducer: a gradient bucket that is complete long before the end of the
    backward pass (70 % of the parameters after the heads and block 7, 98.5 % after block 3) is folded
    (Engine._grad_reduce) and stepped by Adam right then, on the heads stream beside the remaining backward kernels; the
    end of the step only has the last 1.5 % (blocks 2-1 + stem) to fold and step.  Same kernels on sub-ranges of the same
    flat buffers (Adam is elementwise, every reduction keeps its fixed order): bit-identical parameters.  Engine.backward
    drives it exactly as it drives GradBucketReducer (``trigger`` / ``ranges`` / ``stages`` / ``comm_stream``)."""
    native = True         # Engine._reporter calls it directly: the launches below are recorded, no Python hook
    final_on_main = True  # the last bucket on the chain, everything else on comm_stream

    def __init__(self, reducer, opt, comm_stream):
        self.trigger, self.ranges = reducer.trigger, reducer.ranges
        self.stages = set(self.trigger.keys())
        self.opt, self.comm_stream = opt, comm_stream
        self.main_stream = None  # raw handle of the chain's stream, set by the trainer before each backward pass

    def __call__(self, stage):
        arena, opt = self.opt._ensure(), self.opt
        st = self.main_stream if stage == 0 else self.comm_stream.cuda_stream
        for k in self.trigger.get(stage, []):
            lo, hi = self.ranges[k]
            _lib.call("msl_adam_step", ptr(arena.flat[lo:]), ptr(arena.grad[lo:]), ptr(opt.exp_avg[lo:]), ptr(opt.exp_avg_sq[lo:]),
                      ptr(opt.hp), ptr(arena.is_bias[lo:]), hi - lo, st, tag=f"adam_bucket{k}")


class FusedTrainer:
    def __init__(self, model, lr=None, n_buckets=3, process_group=None):
        self.model = model
        self.opt = FusedAdam(model, lr=model.lr if lr is None else lr, weight_decay=0.0005)
        self.sch = CosineAnnealingLR(self.opt, T_max=40) if model.scheduler != "none" else None
        self.n_buckets, self.group = n_buckets, process_group
        self.reducer = None
        self.last_plan = None
        # Replay recorded launch programs after the first step on a set of buffers (natively, two issuing threads).
        # A HIP-graph capture of the same program works but replays at 1.76 ms/step against 1.01 ms (ROCm 7.2, round 2:
        # hipGraphLaunch of the 127-node three-stream graph alone holds the host for 1.1 ms), so there is no graph path.
        self.use_programs = True
        # the upload of the optimiser's hyper-parameters waits for the previous step through an event of the launch program
        # (a stop event of the Adam launch) instead of a torch wait_stream, which put a record at the head of the chain
        self.hp_wait_event = True
        self._step_done = None  # event recorded behind every step's optimiser launch (any plan)
        self.presync_prologue = True  # no "fwd_start" record in the step's program: the trainer orders the heads stream itself
        # single process: fold and step each gradient bucket as soon as it is complete (_StagedOptimizer) instead of one
        # reduction + one Adam launch over everything at the end of the step.  Bit-identical, and measured SLOWER (same-box
        # A/B: fp32 0.667 -> 0.684 ms, bf16 0.646 -> 0.669): the 14 us it takes off the end of the chain cost less than making
        # the heads stream wait, twice per backward pass, for everything the other streams hold - off by default
        self.staged_adam = False
        self._stager = None
        self._programs = collections.OrderedDict()  # LRU, at most max_programs entries (each pins its input tensors)
        self.max_programs = 16
        self._staging = {}         # (image shape, target capacity) -> persistent input buffers (see _stage)
        self._stream = None        # the step runs on its own stream (graph capture needs a non-default one)
        # block after which the target matching is enqueued (a block that forks the heads stream anyway shares its event;
        # A/B after block 4 / 5 at the end of round 3: equal within noise)
        self.match_after = 4
        # which stream carries the overlapped gradient collectives: "heads" | "wgrad" | "own" (see _reducer)
        self.dp_comm_stream = "heads"
        self.main_stream_priority = -1  # the dependency chain must not queue behind the side streams' bulk work

    def _reducer(self, arena):
        if self.reducer is None or self.reducer.arena is not arena:
            self.reducer = GradBucketReducer(arena, self.n_buckets, self.group)
            # which stream carries the overlapped collectives (self.dp_comm_stream = heads | wgrad | own).  A fifth HIP stream
            # per process is not free on this runtime: in the one-rank rehearsal a stream of their own costs 3.9 % of the
            # step, the wgrad stream 3.0 %, the heads stream 0.7 % (it is idle between the head gradients and the odd
            # blocks' weight gradients, and a bucket has to wait for that stream's work anyway)
            which = self.dp_comm_stream
            eng = self.model._engine
            if self.reducer.comm_stream is not None and eng.multi_stream and which in ("heads", "wgrad"):
                sH, sW = eng.side_streams(arena.grad.device)
                self.reducer.comm_stream = sH if which == "heads" else sW
        return self.reducer

    def _eager_step(self, images, gt_boxes, gt_labels, obj_off, total_objects, red):
        """Issue every launch of one step through the Python executor (and record it if a recorder is active)."""
        m, eng, lf = self.model, self.model._engine, self.model.loss_fn
        dev = images.device
        N, P, ncls = images.shape[0], m.priors_cxcycz.shape[0], m.n_classes
        st = lf._state(N, P, ncls, total_objects, dev)
        main = torch.cuda.current_stream().cuda_stream
        after = None
        if eng.multi_stream:
            # the matching only needs the ground truth: it runs on a side stream beside the forward pass.  Not beside the
            # stem / block 1 (bandwidth-bound kernels that a concurrent 37 k-prior arg-max slows down: the stem forward
            # takes 49 us beside it, 38 alone) but beside the small launches from block `match_after` on, on the
            # weight-gradient stream, which is idle during the forward pass (MSL_MATCH_AFTER=0: at the start, heads stream).
            # A/B at 128^3 x 4 (tools/probes/r02_match_after.sh): after block 0 / 1 / 2 / 3 / 4 / 5 ->
            # 0.870 / 0.869 / 0.863 / 0.861-0.866 / 0.858 / 0.861 ms (bf16 0.793 / - / 0.794 / 0.787 / 0.785 / -)
            pl0 = eng.plan_for(images, True)
            k = min(self.match_after, len(eng.layer_specs) - 1)  # (the hook of a block that does not exist would never fire)
            sM = eng.side_streams(dev)[1 if k > 0 else 0].cuda_stream

            def run_match(ev=None):
                if ev is not None:  # the block forks the heads stream anyway: its event serves both (one record fewer)
                    eng._wait(sM, ev)
                else:
                    eng._fork(pl0, "match_start", main, sM)
                lf._run_match(st, N, gt_boxes, gt_labels, obj_off, total_objects, stream=sM, count=True)
            if k > 0:
                after = {k: run_match}
            else:
                run_match()
        # single process, three streams: the optimiser steps bucket by bucket during the backward pass (its hyper-parameters
        # must then be in place before the first bucket: uploaded here, in front of the forward pass)
        stager = None
        if self.staged_adam and eng.multi_stream and not red.active and len(red.ranges) > 1:
            if self._stager is None or self._stager.ranges is not red.ranges:
                self._stager = _StagedOptimizer(red, self.opt, eng.side_streams(dev)[0])
            stager = self._stager
            stager.main_stream = main
            self.opt._ensure()
            self.opt.prepare_step(grad_scale=1.0)
        # The heads stream's prologue (NaN-flag reset, head-weight packing) has to follow the previous step's optimiser.  The
        # trainer orders it itself - here through torch, in every replayed step through the "step_done" event of the previous
        # step's program (see step_packed) - so the recorded forward pass carries no event record at the head of the chain
        presync = self.presync_prologue and eng.multi_stream and eng.prologue_on_side
        if presync:
            eng.side_streams(dev)[0].wait_stream(torch.cuda.current_stream(dev))
        eng.prologue_presynced = presync
        try:
            locs, scores = eng.forward(images, training=True, need_grad=True, nan_check=False, after_block=after)
        finally:
            eng.prologue_presynced = False
        pl = eng.plan_for(images, True)
        if eng.multi_stream:
            eng._fork(pl, "match_done", sM, main)
        if "upstream_alpha" not in st or st["upstream_alpha_value"] != float(lf.alpha):
            st["upstream_alpha"] = torch.tensor([1.0, float(lf.alpha)], dtype=torch.float32, device=dev)  # loss = conf + alpha*loc
            st["upstream_alpha_value"] = float(lf.alpha)
        if eng.multi_stream and lf.can_pack(len(pl.feat_ids)):
            # loss terms + their gradients + the head-gradient images in ONE launch (the matching left the number of positives);
            # the loss values are folded by the batched gradient reduction at the end of the step
            lf._run_loss_pack(st, locs, scores, st["upstream_alpha"], pl.nan_flag, [pl.dO[f] for f in pl.feat_ids],
                              [pl.dims[f] for f in pl.feat_ids], [pl.prior_off[f] for f in pl.feat_ids])
            pl.loss_fold = (st["pack_parts"], st["pack_parts"].numel() // 2, st["loss_out"], st["npos"])
            eng.backward(pl, None, None, on_bucket_ready=stager or red)
        else:
            pl.loss_fold = None
            lf._run_forward(st, locs, scores, gt_boxes, gt_labels, obj_off, total_objects,
                            with_backward_upstream=st["upstream_alpha"], matched=eng.multi_stream, nan_flag=pl.nan_flag)
            eng.backward(pl, st["dlocs"], st["dscores"], on_bucket_ready=stager or red)
        if stager is None:
            scale = red.finish()
            if red.active:
                _lib.record_hook(red.finish, tag="hook:finish")
            self.opt.step(grad_scale=scale, gather_autograd_grads=False)
        if eng.multi_stream and eng.prologue_on_side:
            # "the optimiser has read its hyper-parameter vector": what the NEXT step's upload of that vector waits for.  Part
            # of the launch program (behind the Adam launch it is a stop event: no packet of its own), so that the next step
            # need not put a record at the head of the chain
            # ONE event per trainer, whatever plan (input shape) a step ran on: the wait of the next step must mean "the most
            # recent step", also when steps of different shapes alternate
            if self._step_done is None:
                self._step_done = _lib.new_event(device_only=True)
            _lib.call("msl_event_record", self._step_done, main, tag="event")
        return pl, st

    def _stage(self, images, gt_boxes, gt_labels, obj_off, total_objects):
        """Copy one batch into the persistent input buffers of its (shape, target-capacity) class and return those.

        A launch program is a list of recorded device pointers, so it can only be replayed on the buffers it was
        recorded on: a training loop that hands over freshly allocated tensors every step would otherwise record (and pin)
        a new program per step.  Target rows beyond the batch's objects are never read: the matching kernels walk
        ``obj_off`` (images) and only the per-object arg-max launches one workgroup per row, whose surplus results
        nobody consumes."""
        cap = 8
        while cap < total_objects:
            cap *= 2
        key = (tuple(images.shape), cap, images.device)
        buf = self._staging.get(key)
        if buf is None:
            dev = images.device
            buf = self._staging[key] = (torch.empty(images.shape, dtype=torch.float32, device=dev),
                                        torch.zeros((cap, 6), dtype=torch.float32, device=dev),
                                        torch.ones(cap, dtype=torch.int64, device=dev),
                                        torch.zeros(images.shape[0] + 1, dtype=torch.int32, device=dev))
        x, gb, gl, off = buf
        x.copy_(images, non_blocking=True)
        if total_objects > 0:
            gb[:total_objects].copy_(gt_boxes[:total_objects], non_blocking=True)
            gl[:total_objects].copy_(gt_labels[:total_objects], non_blocking=True)
        off.copy_(obj_off, non_blocking=True)
        return x, gb, gl, off, cap

    def step_packed(self, images, gt_boxes, gt_labels, obj_off, total_objects, sync=True, resident=False, fence=True):
        """One optimisation step on already packed targets (see MultiBoxLoss.pack_targets).  With ``sync=False``
        nothing is read back: the returned dict holds the device tensor ``loss_out`` = [conf, loc, n_positives].

        The first step on a given set of buffers runs through the Python executor and records its launches;
        later steps replay that launch program (same kernels, same arguments, no Python in between).  By default the
        batch is first copied into persistent input buffers (``_stage``), so every step of a training loop replays one
        program; ``resident=True`` promises that the caller re-uses a small fixed set of input tensors (bench.py's
        resident pool) and skips the copy: the programs are then keyed on those tensors (LRU-bounded).

        The step runs on the trainer's own stream.  With ``fence=True`` (default) that stream first waits for the caller's
        current stream (inputs produced there) and the caller's stream afterwards waits for the step (results readable there):
        two cross-queue hand-offs of ~16 us each per step, during which the GPU idles.  ``fence=False`` (with ``sync=False``
        and ``resident=True``; replayed steps only) enqueues the step straight behind the previous one: consecutive steps are
        ordered by the trainer's stream itself, and the CALLER must order anything else against it - ``trainer.fence()`` or
        ``torch.cuda.synchronize()`` - before it reads parameters / ``loss_out`` or rewrites the resident inputs."""
        m = self.model
        if not images.is_cuda:
            raise _lib.HipKernelError("FusedTrainer runs on the HIP device only (no CPU fallback)")
        dev = images.device
        m._ensure_device_state(dev)
        eng = m._engine
        arena = eng.ensure_arena(dev)
        red = self._reducer(arena)
        if self._stream is None or self._stream.device != dev:
            # high priority: the dependency chain must not queue behind the bulk weight-gradient work of the side streams
            self._stream = torch.cuda.Stream(device=dev, priority=self.main_stream_priority)
        caller = torch.cuda.current_stream(dev)
        unfenced = (not fence) and resident and not sync and self.use_programs  # decided for good once the program is known
        if not unfenced:
            self._stream.wait_stream(caller)  # inputs produced on the caller's stream
        # ssd3d.py:527-529: the reference steps the scheduler INSIDE training_step, which Lightning's automatic
        # optimisation runs inside the optimiser closure, i.e. before Adam applies the update: update k uses the
        # learning rate after k scheduler steps
        if self.sch is not None:
            self.sch.step()
        with torch.cuda.stream(self._stream):
            stream = self._stream.cuda_stream
            if not resident and self.use_programs:
                images, gt_boxes, gt_labels, obj_off, total_objects = self._stage(images, gt_boxes, gt_labels, obj_off,
                                                                                  total_objects)
            st0 = m.loss_fn._state(images.shape[0], m.priors_cxcycz.shape[0], m.n_classes, total_objects, dev)
            key = (st0["prior_for_obj"].data_ptr(), images.data_ptr(), tuple(images.shape), gt_boxes.data_ptr(),
                   gt_labels.data_ptr(), obj_off.data_ptr(), total_objects, stream, id(arena), float(m.loss_fn.alpha))
            entry = self._programs.get(key) if self.use_programs else None
            if entry is not None:
                self._programs.move_to_end(key)
            elif unfenced:  # first step on these buffers: they may have just been written on the caller's stream
                self._stream.wait_stream(caller)
                unfenced = False
            if entry is None or eng.prof_all():
                if self.use_programs:
                    _lib.start_recording()
                try:
                    pl, st = self._eager_step(images, gt_boxes, gt_labels, obj_off, total_objects, red)
                finally:
                    prog = _lib.stop_recording() if self.use_programs else None
                if self.use_programs:
                    # keep the tensors the program points at alive for as long as the program exists
                    self._programs[key] = {"prog": prog, "plan": pl, "state": st,
                                           "keep": (images, gt_boxes, gt_labels, obj_off)}
                    while len(self._programs) > self.max_programs:
                        self._programs.popitem(last=False)
            else:
                prog, pl, st = entry["prog"], entry["plan"], entry["state"]
                pl.generation += 1
                pl.saved_input, pl.trained_mode = images, True
                if eng.multi_stream and eng.prologue_on_side:
                    # the optimiser's hyper-parameter vector (this step's learning rate) is read by the last kernel of the
                    # step: copy it on the heads stream, which the chain joins before the optimiser anyway, instead of in
                    # front of the stem (the stream first waits for the previous step's optimiser, which still reads it)
                    sH = eng.side_streams(dev)[0]
                    done = self._step_done if self.hp_wait_event else None
                    if done is not None:  # recorded by the previous step's program, right behind its optimiser launch
                        eng._wait(sH.cuda_stream, done)
                    else:
                        sH.wait_stream(self._stream)
                    if not unfenced:  # a fenced step: whatever the caller's stream holds (the chain waits for it above)
                        sH.wait_stream(caller)
                    with torch.cuda.stream(sH):
                        self.opt.prepare_step(grad_scale=1.0 / red.world)
                else:
                    self.opt.prepare_step(grad_scale=1.0 / red.world)
                if eng.prof is not None:
                    tags = frozenset(eng.prof_tags)
                    segs = entry.setdefault("native_timed", {}).get(tags)
                    if segs is None:
                        segs = entry["native_timed"][tags] = _lib.compile_program(prog, tags, stream, dev.index or 0)
                    _lib.replay_native(segs, eng.prof)
                else:
                    if "native" not in entry:
                        entry["native"] = _lib.compile_program(prog, (), stream, dev.index or 0)
                    _lib.replay_native(entry["native"])
        if not unfenced:
            caller.wait_stream(self._stream)
        self.last_plan = pl
        m.global_step += 1
        out = {"loss_out": st["loss_out"]}
        if sync:
            eng.check_nan(pl)
            conf, loc, npos = st["loss_out"].tolist()
            if loc != loc:  # ssd3d.py:938-940
                raise Exception("Loss is NaN")
            out.update(conf=conf, loc=loc, loss=conf + float(m.loss_fn.alpha) * loc, n_positives=int(npos))
        return out

    def fence(self):
        """Order the caller's current stream behind every step enqueued so far (after ``step_packed(..., fence=False)``)."""
        if self._stream is not None:
            torch.cuda.current_stream(self._stream.device).wait_stream(self._stream)

    def step(self, images, boxes, labels, sync=True):
        gb, gl, off, T = MultiBoxLoss.pack_targets(boxes, labels, images.device)
        return self.step_packed(images, gb, gl, off, T, sync=sync)

    def state_dict(self):
        return {"optimizer": self.opt.state_dict(), "scheduler": None if self.sch is None else self.sch.state_dict()}

    def load_state_dict(self, sd):
        self.opt.load_state_dict(sd["optimizer"])
        if self.sch is not None and sd.get("scheduler") is not None:
            self.sch.load_state_dict(sd["scheduler"])
