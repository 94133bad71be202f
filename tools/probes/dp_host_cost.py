"""Where does the host time of the data-parallel path go?  One-rank RCCL group on one GPU (MSL_DP_REHEARSE=1), the
reducer's calls wrapped with host timers.  Usage: MSL_DP_REHEARSE=1 python tools/probes/dp_host_cost.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
MODE = os.environ.get("PROBE_MODE", "full")  # full | no_ar (exchange path without the collective) | init_only
os.environ["MSL_DP_REHEARSE"] = "0" if MODE == "init_only" else "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29578")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)

from mslesions3d_amd import parallel  # noqa: E402
from mslesions3d_amd.ssd3d import LSSD3D, MultiBoxLoss  # noqa: E402
from mslesions3d_amd.synth import make_batch_on_device  # noqa: E402
from mslesions3d_amd.trainer import FusedTrainer  # noqa: E402

acc = {}


def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        acc[name + "#"] = acc.get(name + "#", 0) + 1
        return r
    return w


parallel.GradBucketReducer.on_stage = timed("on_stage", parallel.GradBucketReducer.on_stage)
parallel.GradBucketReducer.finish = timed("finish", parallel.GradBucketReducer.finish)
class _Done:
    def wait(self):
        return True


_real_ar = dist.all_reduce
_dummy = None


def _variant_ar(t, op=None, group=None, async_op=True):
    """PROBE_AR=dummy: reduce a private buffer instead of the arena slice; PROBE_ASYNC=0: blocking-API form."""
    global _dummy
    if os.environ.get("PROBE_AR") == "dummy":
        if _dummy is None:
            _dummy = torch.zeros(400000, device=t.device)
        t = _dummy[:t.numel()]
    forced = os.environ.get("PROBE_ASYNC")  # unset: what the reducer asked for
    if forced == "0" or (forced is None and not async_op):
        _real_ar(t, op=op, group=group, async_op=False)
        return _Done()
    return _real_ar(t, op=op, group=group, async_op=True)


_ar = _variant_ar if MODE != "no_ar" else (lambda *a, **k: _Done())
dist.all_reduce = timed("all_reduce", _ar)
parallel.dist.all_reduce = dist.all_reduce

if MODE == "ar_only":  # the collective alone: 3 x 1.3 MB in-place SUM per iteration on a side stream
    buf = torch.zeros(949808, device=dev)
    cs = torch.cuda.Stream()
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            ws = []
            for k in range(3):
                cs.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(cs):
                    ws.append(_real_ar(buf[k * 300000:(k + 1) * 300000], op=dist.ReduceOp.SUM, async_op=True))
            for w_ in ws:
                w_.wait()
            torch.cuda.current_stream().wait_stream(cs)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f"mode ar_only: 3 all-reduces: host {th * 10:.3f} ms/iter, wall {(time.perf_counter() - t0) * 10:.3f} ms/iter")
    dist.destroy_process_group()
    sys.exit(0)

torch.manual_seed(1)
model = LSSD3D(n_classes=2, input_channels=1, input_size=(128,) * 3, threshold=[0.1, 0.2], alpha=1.0, lr=1e-3, batch_size=4).to(dev).train()
model._ensure_device_state(dev)
model._engine.ensure_arena(dev)
tr = FusedTrainer(model, n_buckets=int(os.environ.get("PROBE_BUCKETS", "3")))
x, boxes, labels = make_batch_on_device(4, (128,) * 3, dev, 1, seed=3)
packed = (x,) + MultiBoxLoss.pack_targets(boxes, labels, dev)
for _ in range(8):
    tr.step_packed(*packed, sync=False)
torch.cuda.synchronize()
acc.clear()
n = int(os.environ.get("PROBE_STEPS", "100"))
t0 = time.perf_counter()
for _ in range(n):
    tr.step_packed(*packed, sync=False)
th = time.perf_counter() - t0
torch.cuda.synchronize()
tw = time.perf_counter() - t0
print(f"mode {MODE} {os.environ.get('PROBE_AR', '')} async={os.environ.get('PROBE_ASYNC', 'as-requested')} buckets={os.environ.get('PROBE_BUCKETS', '3')}: host {th / n * 1e3:.3f} ms/step, wall {tw / n * 1e3:.3f} ms/step")
for k in sorted(k for k in acc if not k.endswith("#")):
    print(f"  {k:12s} {acc[k] / n * 1e3:8.3f} ms/step  ({acc[k + '#'] / n:.1f} calls/step)")
dist.destroy_process_group()
