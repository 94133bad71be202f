"""Micro-benchmark of the dominant kernel (depthwise 3x3x3 stride-2 forward of block 1, config A) in isolation."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mslesions3d_amd import _lib
from mslesions3d_amd._lib import ptr
L = _lib.load()
N, C, D = 4, 32, 64
x = torch.randn(N, C, D, D, D, device="cuda")
w = torch.randn(C, 27, device="cuda")
sc = torch.rand(C, device="cuda") + 0.5
sh = torch.randn(C, device="cuda") * 0.1
y = torch.empty(N, C, D // 2, D // 2, D // 2, device="cuda")
NP = L.msl_dwconv_fwd_num_partials(N, C, D, D, D, 2)
part = torch.empty(2 * C * max(NP, 4096), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
alg = 4.0 * (N * C * (D ** 3 + (D // 2) ** 3) + C * 27)
for rep in range(3):
    evs = []
    for i in range(60):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("msl_dwconv_fwd", ptr(x), ptr(sc), ptr(sh), ptr(w), ptr(y), ptr(part), N, C, D, D, D, 2, 0, st)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs[10:])
    med = t[len(t) // 2]
    print(f"median {med*1e3:.1f} us  min {t[0]*1e3:.1f} us  -> {alg/med/1e6:.0f} GB/s algorithmic ({alg/med/1e6/8000:.3f} of 8 TB/s)")
