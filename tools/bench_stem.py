"""Micro-benchmark of the stem forward / backward-weight kernels at config A (128^3, batch 4)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mslesions3d_amd import _lib  # noqa: E402
from mslesions3d_amd._lib import ptr  # noqa: E402

L = _lib.load()
N, D = 4, 128
CIN = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # input channels
x = torch.randn(N, CIN, D, D, D, device="cuda")
w = torch.randn(32, 27 * CIN, device="cuda")
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


t = timeit(lambda: _lib.call("msl_stem_conv_fwd", ptr(x), ptr(w), ptr(y), ptr(part), N, CIN, D, D, D, 2, 2, 2, st))
mb = 4e-6 * (x.numel() + y.numel())
print(f"stem fwd: {t:.1f} us, {mb:.0f} MB -> {mb / t:.2f} TB/s (x is cache-resident here: one buffer)")

# fused stem backward (kernel B): dz (N,32,32^3), tap-major w1, y raw, 8-row BN vector block
dz = torch.randn(N, 32, D // 4, D // 4, D // 4, device="cuda")
w1t = torch.randn(27, 32, device="cuda")
vec = torch.rand(8, 32, device="cuda") + 0.5
dw = torch.empty(32, 27 * CIN, device="cuda")
ws = torch.empty(L.msl_stem_conv_bwd_weight_workspace_bytes(CIN) // 4, device="cuda")
t = timeit(lambda: _lib.call("msl_stem_conv_bwd_weight_fused", ptr(dz), ptr(w1t), ptr(y), ptr(vec), ptr(x), ptr(dw), ptr(ws),
                             N, CIN, D, D, D, 2, 2, 2, st))
print(f"stem bwd-weight fused: {t:.1f} us")
g = torch.randn_like(y)
t = timeit(lambda: _lib.call("msl_stem_conv_bwd_weight_bnapply", ptr(g), ptr(y), ptr(vec), ptr(x), ptr(dw), ptr(ws),
                             N, CIN, D, D, D, 2, 2, 2, st))
print(f"stem bwd-weight bnapply (materialised gradient): {t:.1f} us")
