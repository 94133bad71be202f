"""Timing experiments (WRONG RESULTS where noted, probe only): what do the side streams cost / buy?

  full step (three streams) | everything on one stream | weight gradients skipped (wrong results)

The product engine has no switch for skipping work; the probe wraps the launch helper and drops every launch whose tag names
a weight gradient.  Usage (GPU box): python tools/probes/skip_wgrad.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from mslesions3d_amd import engine as eng_mod  # noqa: E402
from mslesions3d_amd.ssd3d import LSSD3D, MultiBoxLoss  # noqa: E402
from mslesions3d_amd.synth import make_batch_on_device  # noqa: E402
from mslesions3d_amd.trainer import FusedTrainer  # noqa: E402

SKIP = ("pw_bww", "dw_bww", "head_bww", "grad_reduce")


def run(skip=False, multi_stream=True):
    orig = eng_mod.Engine._k

    def _k(self, tag, name, *args):
        if skip and tag.startswith(SKIP):
            return
        orig(self, tag, name, *args)

    eng_mod.Engine._k = _k
    try:
        dev = torch.device("cuda", 0)
        torch.manual_seed(0)
        model = LSSD3D(n_classes=2, input_channels=1, input_size=(128,) * 3, threshold=[0.1, 0.2], lr=1e-3).to(dev).train()
        model._engine.multi_stream = multi_stream
        tr = FusedTrainer(model)
        x, b, l = make_batch_on_device(4, (128,) * 3, dev, 1, seed=1)
        packed = (x,) + MultiBoxLoss.pack_targets(b, l, dev)
        for _ in range(10):
            tr.step_packed(*packed, sync=False, resident=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            tr.step_packed(*packed, sync=False, resident=True)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 200 * 1e3
    finally:
        eng_mod.Engine._k = orig


if __name__ == "__main__":
    print(f"full step {run():.3f} ms; one stream {run(multi_stream=False):.3f} ms; "
          f"weight gradients skipped (wrong results) {run(skip=True):.3f} ms; "
          f"one stream, weight gradients skipped {run(skip=True, multi_stream=False):.3f} ms")
