"""Composition of the recorded launch program of one training step (GPU box): calls per entry point / stream, and the
host cost of replaying it into idle queues."""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mslesions3d_amd import _lib  # noqa: E402
from mslesions3d_amd.ssd3d import LSSD3D, MultiBoxLoss  # noqa: E402
from mslesions3d_amd.synth import make_batch_on_device  # noqa: E402
from mslesions3d_amd.trainer import FusedTrainer  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
model = LSSD3D(n_classes=2, input_channels=1, input_size=(size,) * 3, threshold=[0.1, 0.2], lr=1e-3).to(dev).train()
tr = FusedTrainer(model)
x, b, l = make_batch_on_device(4, (size,) * 3, dev, 1, seed=1)
packed = (x,) + MultiBoxLoss.pack_targets(b, l, dev)
for _ in range(3):
    tr.step_packed(*packed, sync=False, resident=True)
torch.cuda.synchronize()
entry = list(tr._programs.values())[-1]
prog = entry["prog"]
names = collections.Counter(fn.__name__ if fn is not None else "hook" for fn, _, _ in prog)
streams = collections.Counter()
for fn, args, tag in prog:
    if fn is None:
        continue
    n = fn.__name__
    st = args[1] if n == "msl_event_record" else args[0] if n == "msl_stream_wait_event" else args[-1]
    streams[(st, "event" if "event" in n else "launch")] += 1
print(f"{len(prog)} calls per step")
for k, v in names.most_common():
    print(f"  {v:4d} {k}")
for k, v in sorted(streams.items(), key=lambda kv: str(kv[0])):
    print("  stream", hex(k[0] or 0), k[1], v)
# host cost of one replay into idle queues, and split launches / events
comp = entry["native"]
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.opt.prepare_step(1.0)
    _lib.replay_native(comp)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"replay host {1e3 * (t1 - t0):.3f} ms, until idle {1e3 * (t2 - t0):.3f} ms")
