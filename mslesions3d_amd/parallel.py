"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference is single-GPU (``train.py:182`` ``devices=1``; the only multi-GPU trace is an ``nn.DataParallel``
demo, ``mobilenet.py:175``), so this is new design: each rank runs the whole step on its own
shard of volumes (BatchNorm statistics and the loss normaliser stay per replica, as with DDP), and the only
exchange is a SUM all-reduce of the 949 808 fp32 gradients (3.8 MB), averaged inside the fused Adam
(gradient scale 1/world).  The flat gradient arena is ordered by backward completion (heads first, stem
last), so it is cut into a few contiguous buckets; bucket k is all-reduced on a side stream as soon as the
launches that produce it have been enqueued, overlapping the remaining backward kernels.  3.8 MB is
latency-bound on xGMI: few large buckets, not many small ones.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend="nccl", rank=None, world_size=None, device=None, timeout_s=None):
    """One process per GPU: join the job's process group (``nccl`` = RCCL over xGMI on ROCm).

    Failure handling: a collective that a peer never joins must not hang the job forever.  The group
    gets a finite timeout (``MSL_DP_TIMEOUT_S``, default 180 s; torch's NCCL default is 10 min) and torch's watchdog is left
    in its tear-down mode (``TORCH_NCCL_ASYNC_ERROR_HANDLING=1``): on a timeout or an RCCL error the watchdog aborts the
    communicator and the process exits with the error instead of blocking in a stream wait.  No elasticity: the launcher
    restarts the job from the last checkpoint (train.py --checkpoint restores optimiser, scheduler and epoch)."""
    import datetime
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else world_size
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
    timeout = datetime.timedelta(seconds=float(os.environ.get("MSL_DP_TIMEOUT_S", "180") if timeout_s is None else timeout_s))
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world_size, timeout=timeout, **kw)
    return dist.group.WORLD


class GradBucketReducer:
    def __init__(self, arena, n_buckets=3, process_group=None):
        self.arena = arena
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # active: buckets are exchanged.  MSL_DP_REHEARSE=1 runs the whole exchange path (side-stream joins, RCCL launches,
        # waits) in a one-rank group too: the host and launch cost of data parallelism measured on a single GPU
        self.active = self.world > 1 or (os.environ.get("MSL_DP_REHEARSE") == "1" and dist.is_available()
                                         and dist.is_initialized())
        self.ranges = arena.bucket_ranges(n_buckets)
        # stage after which a bucket is complete: 'heads', 7, 6, ..., 0 (see Engine.backward)
        stage_of = {}
        for name in arena.names:
            if name in arena.no_grad_names:
                continue
            lo, n = arena.offsets[name]
            stage = "heads" if name.startswith("pred_convs") else int(name.split(".")[2])
            for k, (blo, bhi) in enumerate(self.ranges):
                if lo < bhi and lo + n > blo:
                    prev = stage_of.get(k)
                    # later stage = smaller feature index; 'heads' is the earliest
                    if prev is None or prev == "heads" or (stage != "heads" and stage < prev):
                        stage_of[k] = stage
        self.trigger = {}
        for k, stg in stage_of.items():
            self.trigger.setdefault(stg, []).append(k)
        self.comm_stream = torch.cuda.Stream() if self.active and arena.grad.is_cuda else None
        self.pending = []
        self.final_on_main = True  # stage 0 (everything joined) is exchanged on the caller's stream, see on_stage
        self.presynced = False  # set by a caller that already made comm_stream wait for the bucket's producers
        # Engine.backward only reports (and joins its side streams for) the stages that complete a bucket
        self.stages = set(self.trigger.keys()) if self.active else set()

    def __call__(self, stage):
        self.on_stage(stage)

    def on_stage(self, stage):
        """Engine.backward calls this right after enqueueing the kernels of ``stage``."""
        if not self.active:
            return
        for k in self.trigger.get(stage, []):
            lo, hi = self.ranges[k]
            view = self.arena.grad[lo:hi]
            if self.comm_stream is not None and stage == 0 and self.final_on_main:
                # the last bucket (a few KB: first blocks + stem) is complete when every stream has been joined into the
                # chain, and the optimiser is next: exchange it on the chain's own stream - no cross-queue hand-off
                # (~16 us each way) in front of Adam.  One communicator = one collective at a time: the earlier buckets on
                # comm_stream must have finished (the engine records that wait in the launch program)
                if not self.presynced:
                    torch.cuda.current_stream().wait_stream(self.comm_stream)
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=False)
            elif self.comm_stream is not None:
                if not self.presynced:
                    self.comm_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.comm_stream):
                    # stream-ordered form: RCCL's stream waits for comm_stream and comm_stream for the collective, the host
                    # for nothing.  async_op=True + Work.wait() gives the same ordering but made the whole step 3.3x
                    # slower on ROCm 7.2 / torch 2.10 (3.4 vs 1.06 ms; tools/probes/dp_host_cost.py)
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=False)
            else:  # CPU tensors (gloo tests)
                self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Make the compute stream wait for every bucket; returns the gradient scale for the optimiser."""
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        return 1.0 / self.world


def broadcast_model(model, src=0, group=None):
    """Initial weights + BatchNorm buffers from rank ``src`` (one flat broadcast for the arena)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if dist.get_world_size(group) == 1 and os.environ.get("MSL_DP_REHEARSE") != "1":
        return
    arena = getattr(model._engine, "arena", None)
    if arena is not None:
        dist.broadcast(arena.flat, src=src, group=group)
    else:
        for p in model.parameters():
            dist.broadcast(p.data, src=src, group=group)
    for b in model.buffers():
        dist.broadcast(b, src=src, group=group)
