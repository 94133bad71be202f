import time, torch
dev = torch.device("cuda", 0)
big = torch.randn(8192, 8192, device=dev)
hp = torch.zeros(8, device=dev)
pin = torch.zeros(8, pin_memory=True)
torch.cuda.synchronize()
for name, fn in (("pageable", lambda: hp.copy_(torch.tensor([1.0] * 8), non_blocking=True)),
                 ("pinned", lambda: hp.copy_(pin, non_blocking=True))):
    for _ in range(2):
        torch.cuda.synchronize()
        for _ in range(6):
            big @ big          # ~15 ms of GPU work queued
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name}: copy_ call returned after {1e3 * (t1 - t0):.3f} ms; GPU drained after {1e3 * (t2 - t0):.3f} ms")
