import sys, time, torch
sys.path.insert(0, "/root/repo")
from mslesions3d_amd.ssd3d import LSSD3D
from mslesions3d_amd import _lib
from mslesions3d_amd.synth import make_batch_on_device
dev = torch.device("cuda", 0)
size = (192,) * 3
torch.manual_seed(970205)
m = LSSD3D(n_classes=2, input_channels=1, input_size=size, threshold=[0.1, 0.2], batch_size=2).to(dev).eval()
x, _, _ = make_batch_on_device(2, size, dev, 1, seed=3)
for _ in range(3):
    m.predict_step({"img": x})
ent = list(m._pred_programs.values())[0]
def t(fn, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
print("copy_:  host %.3f ms, drained %.3f ms" % t(lambda: ent["buf"].copy_(x)))
print("replay: host %.3f ms, drained %.3f ms" % t(lambda: _lib.replay_native(ent["compiled"], None)))
print("collect: host %.3f ms" % t(lambda: m._detect_collect(ent["ws"], 2))[0])
print("predict_step: host %.3f ms" % t(lambda: m.predict_step({"img": x}))[0])
# the detect call alone
import ctypes
