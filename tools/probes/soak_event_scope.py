"""2000 fused training steps (128^3 x 4, pool of 4 synthetic batches, unfenced replay) twice from the same initial state:
once with ordinary fork / join events and records, once with device-scope events (_lib.DEVICE_SCOPE_EVENTS,
hipEventDisableSystemFence) and stop-event forks (_lib.STOP_EVENT_FORKS).
The step is deterministic, so the two runs must end at bit-identical parameters and loss histories; a consumer reading stale
data behind a device-scope event would break that.  Also fp32 vs bf16 activations."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mslesions3d_amd import _lib
from mslesions3d_amd.ssd3d import LSSD3D, MultiBoxLoss
from mslesions3d_amd.synth import make_batch_on_device
from mslesions3d_amd.trainer import FusedTrainer
dev = torch.device("cuda", 0)
size = (128,) * 3
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000


def run(flag, dtype):
    _lib.DEVICE_SCOPE_EVENTS = _lib.STOP_EVENT_FORKS = flag
    torch.manual_seed(970205)
    m = LSSD3D(n_classes=2, input_channels=1, input_size=size, threshold=[0.1, 0.2], alpha=1.0, lr=1e-3, batch_size=4).to(dev).train()
    m.compute_dtype = dtype
    tr = FusedTrainer(m)
    pool = []
    for k in range(4):
        x, b, l = make_batch_on_device(4, size, dev, 1, seed=k)
        pool.append((x,) + MultiBoxLoss.pack_targets(b, l, dev))
    hist = []
    for s in range(steps):
        x, gb, gl, off, T = pool[s % 4]
        rd = s % 100 == 0 or s == steps - 1
        out = tr.step_packed(x, gb, gl, off, T, sync=rd)
        if rd:
            hist.append((out["conf"], out["loc"]))
    torch.cuda.synchronize()
    return hist, torch.cat([q.detach().reshape(-1) for q in m.parameters()]).cpu()


for dtype in ("f32", "bf16"):
    a = run(False, dtype)
    b = run(True, dtype)
    same = a[0] == b[0] and torch.equal(a[1], b[1])
    print(f"{dtype}: {steps} steps, loss {a[0][0]} -> {a[0][-1]}; cheap forks bit-identical to plain events: {same}", flush=True)
    assert same
print("soak_event_scope ok")
