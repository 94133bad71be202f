"""Every launch of the recorded training step, re-issued ALONE (50 back to back between one event pair, the step's own
buffers): kernel + dependent-dispatch time without the other streams.  Compare with the in-step rocprofv3 durations
(tools/prof_step.sh) to see what concurrency costs each kernel.  GPU box: python tools/probes/alone.py [size]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
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
prog = list(tr._programs.values())[-1]["prog"]
main = tr._stream.cuda_stream
rows, total = [], {}
with torch.cuda.stream(tr._stream):
    for fn, args, tag in prog:
        if fn is None or "event" in fn.__name__ or fn.__name__ == "msl_adam_step":
            continue
        a = list(args)
        a[-1] = main  # everything on one stream
        for _ in range(3):
            fn(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn(*a)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        on_main = args[-1] == main
        rows.append((tag, us, on_main))
tot_main = sum(u for _, u, m in rows if m)
tot_side = sum(u for _, u, m in rows if not m)
for tag, us, m in rows:
    print(f"{'main' if m else 'side'} {us:8.1f} us  {tag}")
print(f"sum alone: main-stream launches {tot_main:.0f} us, side-stream launches {tot_side:.0f} us")
