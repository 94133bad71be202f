"""2000 fused training steps on a pool of 4 synthetic batches: losses stay finite and go down, device memory is flat,
the launch-program cache holds one program per batch buffer."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mslesions3d_amd.ssd3d import LSSD3D, MultiBoxLoss
from mslesions3d_amd.synth import make_batch_on_device
from mslesions3d_amd.trainer import FusedTrainer
dev = torch.device("cuda", 0)
size = (128,) * 3
torch.manual_seed(970205)
m = LSSD3D(n_classes=2, input_channels=1, input_size=size, threshold=[0.1, 0.2], alpha=1.0, lr=1e-3, batch_size=4).to(dev).train()
tr = FusedTrainer(m)
pool = []
for k in range(4):
    x, b, l = make_batch_on_device(4, size, dev, 1, seed=k)
    pool.append((x,) + MultiBoxLoss.pack_targets(b, l, dev))
hist = []
for s in range(2000):
    x, gb, gl, off, T = pool[s % 4]
    out = tr.step_packed(x, gb, gl, off, T, sync=(s % 250 == 0 or s == 1999))
    if s % 250 == 0 or s == 1999:
        hist.append((s, out["conf"], out["loc"], torch.cuda.memory_allocated() / 2**20))
        print(f"step {s}: conf {out['conf']:.4f} loc {out['loc']:.4f} allocated {hist[-1][3]:.0f} MiB programs {len(tr._programs)}", flush=True)
m._engine.check_nan(tr.last_plan)
assert all(c == c and l == l for _, c, l, _ in hist)
assert hist[-1][1] + hist[-1][2] < 0.5 * (hist[0][1] + hist[0][2]), "loss did not go down"
assert abs(hist[-1][3] - hist[1][3]) < 1.0, "device memory grew"
print("soak ok")
