"""Micro-benchmark of the three head kernels at the 16^3 scale of config A (N = 4, C = 128, two classes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mslesions3d_amd import _lib  # noqa: E402
from mslesions3d_amd._lib import ptr  # noqa: E402

L = _lib.load()
N, C, D, ncls = int(os.environ.get("HEAD_N", "4")), int(os.environ.get("HEAD_C", "128")), int(os.environ.get("HEAD_D", "16")), 2
dev = "cuda"
S = D ** 3
pad = torch.zeros((N, C, D + 2, D + 2, D + 2), device=dev)
pad[:, :, 1:-1, 1:-1, 1:-1] = torch.randn((N, C, D, D, D), device=dev)
lw, cw = torch.randn(12, C, 3, 3, 3, device=dev) * 0.02, torch.randn(4, C, 3, 3, 3, device=dev) * 0.02
lb, cb = torch.zeros(12, device=dev), torch.zeros(4, device=dev)
ne = L.msl_head_packed_weight_elems(C, ncls)
Wf, Wb = torch.empty(ne, device=dev), torch.empty(ne, device=dev)
st = torch.cuda.current_stream().cuda_stream
_lib.call("msl_head_pack_weights", ptr(lw), ptr(cw), ptr(Wf), ptr(Wb), C, ncls, st)
ws = torch.empty(max(L.msl_head_fwd_workspace_bytes(N, C, D, D, D, ncls), L.msl_head_bwd_weight_workspace_bytes(N, C, D, D, D, ncls)) // 4 + 1, device=dev)
P = 2 * S
locs, scores = torch.empty((N, P, 6), device=dev), torch.empty((N, P, ncls), device=dev)
dO = torch.randn((N, 16, D + 2, D + 2, D + 2), device=dev)
ga = torch.empty((N, C, D, D, D), device=dev)


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


gf = 2.0 * N * S * C * 27 * 16 / 1e9
t = timeit(lambda: _lib.call("msl_head_conv_fwd", ptr(pad), ptr(Wf), ptr(lb), ptr(cb), ptr(locs), ptr(scores), ptr(ws), N, C, D, D, D, P, 0, ncls, st))
print(f"head fwd      {D}^3 x{N} C={C}: {t:6.1f} us  {gf / t * 1e3:6.1f} TFLOP/s")
if os.environ.get("HEAD_FWD_ONLY") == "1":
    sys.exit(0)
t = timeit(lambda: _lib.call("msl_head_conv_bwd_data", ptr(dO), ptr(Wb), ptr(ga), N, C, D, D, D, ncls, st))
print(f"head bwd-data {D}^3 x{N} C={C}: {t:6.1f} us  {gf / t * 1e3:6.1f} TFLOP/s")
t = timeit(lambda: _lib.call("msl_head_conv_bwd_weight", ptr(dO), ptr(pad), None, None, None, None, ptr(ws), N, C, D, D, D, ncls, st))
print(f"head bwd-wgt  {D}^3 x{N} C={C}: {t:6.1f} us  {gf / t * 1e3:6.1f} TFLOP/s")
