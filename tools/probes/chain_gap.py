"""Idle time between two launches of the chain inside the replayed training step, WITHOUT a profiler: HIP-event pairs around
the two tagged launches; gap = elapsed(end of first, start of second).  Usage: chain_gap.py <dtype> <tag_a> <tag_b>"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mslesions3d_amd import _lib  # noqa: E402
from mslesions3d_amd.ssd3d import LSSD3D, MultiBoxLoss  # noqa: E402
from mslesions3d_amd.synth import make_batch_on_device  # noqa: E402
from mslesions3d_amd.trainer import FusedTrainer  # noqa: E402

dtype, tag_a, tag_b = sys.argv[1], sys.argv[2], sys.argv[3]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = LSSD3D(n_classes=2, input_channels=1, input_size=(128,) * 3, threshold=[0.1, 0.2], lr=1e-3).to(dev).train()
model.compute_dtype = dtype
tr = FusedTrainer(model)
x, b, l = make_batch_on_device(4, (128,) * 3, dev, 1, seed=1)
packed = (x,) + MultiBoxLoss.pack_targets(b, l, dev)
for _ in range(6):
    tr.step_packed(*packed, sync=False, resident=True, fence=False)
torch.cuda.synchronize()
eng = model._engine
eng.prof, eng.prof_tags = {}, {tag_a, tag_b}
for _ in range(30):
    tr.step_packed(*packed, sync=False, resident=True, fence=False)
torch.cuda.synchronize()
A, B = eng.prof[tag_a], eng.prof[tag_b]
gaps = [_lib.elapsed_ms(a[2], b_[1]) * 1e3 for a, b_ in zip(A, B)][5:]
da = [_lib.elapsed_ms(a[1], a[2]) * 1e3 for a in A][5:]
db = [_lib.elapsed_ms(b_[1], b_[2]) * 1e3 for b_ in B][5:]
print(f"{dtype}: {tag_a} {sum(da) / len(da):.1f} us -> gap {sum(gaps) / len(gaps):.1f} us (min {min(gaps):.1f}, max {max(gaps):.1f}) -> {tag_b} {sum(db) / len(db):.1f} us")
eng.prof = None
