"""Micro-benchmark of the pointwise (1x1x1) kernels at the seven block shapes of config A (128^3, batch 4):
forward (input affine + BN statistics), forward without statistics, backward-data.
Usage (GPU box): python tools/bench_pw.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mslesions3d_amd import _lib  # noqa: E402
from mslesions3d_amd._lib import ptr  # noqa: E402

L = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
N = 4
shapes = [(32, 64, 32 ** 3), (64, 128, 16 ** 3), (128, 128, 16 ** 3), (128, 256, 8 ** 3), (256, 256, 8 ** 3),
          (256, 512, 4 ** 3), (512, 512, 4 ** 3)]


def timeit(fn, reps=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for cin, cout, S in shapes:
    z = torch.randn(N, cin, S, device=dev)
    w = torch.randn(cout, cin, device=dev) / cin ** 0.5
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
    y = torch.empty(N, cout, S, device=dev)
    NP = L.msl_pwconv_fwd_num_partials(N, cin, cout, S)
    part = torch.empty(2 * cout * NP, dtype=torch.float64, device=dev)
    dy = torch.randn(N, cout, S, device=dev)
    g = torch.empty(N, cin, S, device=dev)
    t_fwd = timeit(lambda: _lib.call("msl_pwconv_fwd", ptr(z), ptr(sc), ptr(sh), ptr(w), ptr(y), ptr(part), N, cin, cout, S, st))
    t_nostat = timeit(lambda: _lib.call("msl_pwconv_fwd", ptr(z), ptr(sc), ptr(sh), ptr(w), ptr(y), None, N, cin, cout, S, st))
    t_plain = timeit(lambda: _lib.call("msl_pwconv_fwd", ptr(z), None, None, ptr(w), ptr(y), None, N, cin, cout, S, st))
    t_bwd = timeit(lambda: _lib.call("msl_pwconv_bwd_data", ptr(dy), ptr(w), ptr(g), N, cin, cout, S, st))
    mb = 4e-6 * N * S * (cin + cout)
    print(f"cin {cin:4d} cout {cout:4d} S {S:6d} NP {NP:5d} | fwd {t_fwd:6.1f} us  no-stats {t_nostat:6.1f}  no-affine {t_plain:6.1f} | "
          f"bwd-data {t_bwd:6.1f} us | {mb:6.1f} MB -> {mb / 5e6 * 1e6:5.1f} us at 5 TB/s", flush=True)
