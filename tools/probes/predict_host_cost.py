"""Host time of one inference pass at 192^3 x 2: enqueue (stage + replay the launch program + request the copies) and finish
(event wait excluded: measured after a device synchronise; counts -> lists)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mslesions3d_amd.ssd3d import LSSD3D
from mslesions3d_amd.synth import make_batch_on_device
dev = torch.device("cuda", 0)
size = (192,) * 3
torch.manual_seed(970205)
m = LSSD3D(n_classes=2, input_channels=1, input_size=size, threshold=[0.1, 0.2]).to(dev).eval()
m.min_score, m.max_overlap, m.top_k = 0.3, 0.3, 50
x = make_batch_on_device(2, size, dev, 1, seed=3)[0]
for _ in range(3):
    m.predict_step({"img": x})
buf = m.predict_input_buffer(x.shape)
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
hs = [m._predict_enqueue(buf, slot=k % 2) for k in range(n)]  # (two slots, as predict_batches: the landing zones are overwritten - timing only)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
out = [m._predict_finish(h) for h in hs[-2:]]

t3 = time.perf_counter()
print(f"enqueue {1e6 * (t1 - t0) / n:.0f} us/pass (host), device drain {1e6 * (t2 - t0) / n:.0f} us/pass wall, finish {1e6 * (t3 - t2) / 2:.0f} us/pass (host)")
