"""How far ahead of the GPU can the host enqueue?  Block the stream with a finite spin kernel (torch.cuda._sleep), then
time every one of 4000 tiny launches (msl_fill_u32 through the C ABI): the index where the per-launch host time jumps
from ~3 us to the GPU's pace is the number of launches the runtime lets a stream hold."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mslesions3d_amd import _lib  # noqa: E402
from mslesions3d_amd._lib import ptr  # noqa: E402

L = _lib.load()
buf = torch.zeros(64, dtype=torch.int32, device="cuda")
for nstreams in (1, 3):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    torch.cuda.synchronize()
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(int(2.0e9 * 0.02))  # ~20 ms at ~2 GHz: finite
    t = []
    n = 4000
    t0 = time.perf_counter()
    for i in range(n):
        L.msl_fill_u32(ptr(buf), 0, 1, streams[i % nstreams].cuda_stream)
        t.append(time.perf_counter())
    torch.cuda.synchronize()
    dt = [(t[i] - (t[i - 1] if i else t0)) * 1e6 for i in range(n)]
    knee = next((i for i in range(8, n) if sum(dt[i:i + 8]) / 8 > 5 * (sum(dt[:64]) / 64 + 1.0)), None)
    print(f"{nstreams} stream(s): first 64 launches {sum(dt[:64]) / 64:.2f} us each; knee at launch {knee}; "
          f"total host {1e3 * (t[-1] - t0):.2f} ms; slowest single enqueue {max(dt):.0f} us at {dt.index(max(dt))}", flush=True)
    for lo in (0, 256, 512, 1024, 2048, 3072):
        seg = dt[lo:lo + 256]
        print(f"   launches {lo:4d}-{lo + 255}: mean {sum(seg) / len(seg):6.2f} us  max {max(seg):7.1f} us")
