"""Eval-mode stem + block-1 depthwise convolution: the fused launch (csrc/stemdw.hip) against the two separate ones, at the
192^3 x 2 shape of BASELINE configs[3] (HEAD_N / HEAD_D to change it)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mslesions3d_amd import _lib  # noqa: E402
from mslesions3d_amd._lib import ptr  # noqa: E402

L = _lib.load()
N, D = int(os.environ.get("SDW_N", "2")), int(os.environ.get("SDW_D", "192"))
dev = "cuda"
x = torch.randn((N, 1, D, D, D), device=dev)
w = torch.randn((32, 27), device=dev) * 0.2
wd = torch.randn((32, 27), device=dev) * 0.2
sc, sh = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.3
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for bf16 in (False, True):
    dt = torch.bfloat16 if bf16 else torch.float32
    sfx = "_bf16" if bf16 else ""
    y = torch.empty((N, 32, D // 2, D // 2, D // 2), dtype=dt, device=dev)
    z = torch.empty((N, 32, D // 4, D // 4, D // 4), dtype=dt, device=dev)
    z2 = torch.empty_like(z)

    def two():
        _lib.call("msl_stem_conv_fwd" + sfx, ptr(x), ptr(w), ptr(y), None, N, 1, D, D, D, 2, 2, 2, st)
        if bf16:
            _lib.call("msl_dwconv_fwd_bf16", ptr(y), ptr(sc), ptr(sh), ptr(wd), ptr(z), None, N, 32, D // 2, D // 2, D // 2, 2, st)
        else:
            _lib.call("msl_dwconv_fwd", ptr(y), ptr(sc), ptr(sh), ptr(wd), ptr(z), None, N, 32, D // 2, D // 2, D // 2, 2, 0, st)

    def one():
        _lib.call("msl_stem_dw_fwd_eval" + sfx, ptr(x), ptr(w), ptr(sc), ptr(sh), ptr(wd), ptr(z2), N, 1, D, D, D, st)

    t2, t1 = timeit(two), timeit(one)
    print(f"{'bf16' if bf16 else 'fp32'} {D}^3 x{N}: stem + depthwise {t2:7.1f} us, fused {t1:7.1f} us, equal {torch.equal(z, z2)}")
