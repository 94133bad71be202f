"""Micro-benchmark of the stem forward / backward-weight kernels at config A (128^3, batch 4)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mslesions3d_amd import _lib  # noqa: E402
from mslesions3d_amd._lib import ptr  # noqa: E402

L = _lib.load()
N, D = 4, 128
x = torch.randn(N, 1, D, D, D, device="cuda")
w = torch.randn(32, 27, device="cuda")
y = torch.empty(N, 32, D // 2, D // 2, D // 2, device="cuda")
NP = L.msl_stem_conv_fwd_num_partials(N, D // 2, D // 2, D // 2)
part = torch.empty(2 * 32 * NP, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


t = timeit(lambda: _lib.call("msl_stem_conv_fwd", ptr(x), ptr(w), ptr(y), ptr(part), N, 1, D, D, D, 2, 2, 2, st))
mb = 4e-6 * (x.numel() + y.numel())
print(f"stem fwd (MSL_STEM_DEBUG={os.environ.get('MSL_STEM_DEBUG', '0')}): {t:.1f} us, {mb:.0f} MB -> {mb / t * 1e-6 * 1e6 / 1e6:.2f} TB/s")
