"""Per-kernel dispatch floor: N dependent launches of a trivial kernel on one stream (msl_fill_u32 of one word),
through ctypes and through the native program runner.  Usage (GPU box): python tools/bench_launch.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mslesions3d_amd import _lib  # noqa: E402
from mslesions3d_amd._lib import ptr  # noqa: E402

L = _lib.load()
buf = torch.zeros(16, dtype=torch.int32, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    st = s.cuda_stream
    for n in (1000, 4000):
        for _ in range(100):
            _lib.call("msl_fill_u32", ptr(buf), 0, 1, st)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record(s)
        for _ in range(n):
            _lib.call("msl_fill_u32", ptr(buf), 0, 1, st)
        b.record(s)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"{n} launches: host {1e6 * (t1 - t0) / n:.2f} us/launch, GPU {1e3 * a.elapsed_time(b) / n:.2f} us/launch", flush=True)
    # recorded once, replayed natively (no Python between launches)
    _lib.start_recording()
    for _ in range(1000):
        _lib.call("msl_fill_u32", ptr(buf), 0, 1, st)
    prog = _lib.stop_recording()
    comp = _lib.compile_program(prog, set())
    for _ in range(3):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record(s)
        _lib.replay_native(comp, None)
        b.record(s)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"native replay of 1000: host {1e3 * (t1 - t0):.2f} us/launch, GPU {a.elapsed_time(b):.2f} us/launch", flush=True)
    # GPU-side floor: queue the launches behind ~20 ms of other work so that the host is far ahead
    big = torch.randn(8192, 8192, device="cuda")
    for n in (1000,):
        torch.cuda.synchronize()
        for _ in range(8):
            big @ big
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        _lib.replay_native(comp, None)
        b.record(s)
        torch.cuda.synchronize()
        print(f"queued behind other work: GPU {a.elapsed_time(b):.2f} us per dependent trivial kernel", flush=True)
    # with an event record + a wait on another stream's event between kernels
    s2 = torch.cuda.Stream()
    ev = _lib.new_event()
    _lib.start_recording()
    for _ in range(500):
        _lib.call("msl_fill_u32", ptr(buf), 0, 1, st)
        _lib.call("msl_event_record", ev, st, tag="event")
        _lib.call("msl_stream_wait_event", s2.cuda_stream, ev, tag="event")
        _lib.call("msl_fill_u32", ptr(buf[8:]), 0, 1, s2.cuda_stream)
        _lib.call("msl_event_record", ev, s2.cuda_stream, tag="event")
        _lib.call("msl_stream_wait_event", st, ev, tag="event")
    comp2 = _lib.compile_program(_lib.stop_recording(), set())
    torch.cuda.synchronize()
    for _ in range(8):
        big @ big
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record(s)
    _lib.replay_native(comp2, None)
    b.record(s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"ping-pong between two streams (500 round trips): GPU {2 * a.elapsed_time(b):.2f} us per round trip, "
          f"host {1e6 * (t1 - t0) / 500:.2f} us per round trip (2 launches + 2 records + 2 waits)", flush=True)
