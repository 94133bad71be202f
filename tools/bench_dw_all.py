"""Depthwise forward of all seven blocks at config A (128^3, batch 4), each kernel alone on the GPU, rotating over several
input buffers so that the Infinity Cache does not hold the input: SURVEY 8(d)'s aggregate  222.71 MB / sum(t) / 8 TB/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mslesions3d_amd import _lib  # noqa: E402
from mslesions3d_amd._lib import ptr  # noqa: E402

L = _lib.load()
st = torch.cuda.current_stream().cuda_stream
N = 4
layers = [(32, 64, 2), (64, 32, 2), (128, 16, 1), (128, 16, 2), (256, 8, 1), (256, 8, 2), (512, 4, 1)]
total_us, total_mb = 0.0, 0.0
for i, (C, D, s) in enumerate(layers, 1):
    OD = (D - 1) // s + 1
    nbuf = max(2, min(8, int(600e6 // (N * C * D ** 3 * 4))))
    xs = [torch.randn(N, C, D, D, D, device="cuda") for _ in range(nbuf)]
    w = torch.randn(C, 27, device="cuda")
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    y = torch.empty(N, C, OD, OD, OD, device="cuda")
    NP = L.msl_dwconv_fwd_num_partials(N, C, D, D, D, s)
    part = torch.empty(2 * C * max(NP, 1), dtype=torch.float64, device="cuda")

    def run(k):
        _lib.call("msl_dwconv_fwd", ptr(xs[k % nbuf]), ptr(sc), ptr(sh), ptr(w), ptr(y), ptr(part), N, C, D, D, D, s, 0, st)
    for k in range(10):
        run(k)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 100
    a.record()
    for k in range(reps):
        run(k)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / reps * 1e3
    mb = 4e-6 * (N * C * (D ** 3 + OD ** 3) + C * 27)
    total_us += us
    total_mb += mb
    print(f"block {i}: C={C:3d} {D:2d}^3 -> {OD:2d}^3 (variant {L.msl_dwconv_fwd_variant(N, C, D, D, D, s)}): {us:6.1f} us, {mb:7.2f} MB, "
          f"{mb / us:5.2f} TB/s", flush=True)
print(f"all seven: {total_us:.1f} us for {total_mb:.2f} MB -> {total_mb / total_us:.2f} TB/s = {total_mb / total_us / 8:.3f} of 8 TB/s "
      f"(back-to-back launches on one stream: each time includes the ~1.8 us dispatch)")
