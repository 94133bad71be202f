"""Inference path of BASELINE configs[3] (192^3, batch 2) in fp32: eval forward + decode + 3-D NMS (predict_step),
timed, then checked against the CPU oracle on the same volumes (keep-lists bit-exact, boxes within 1e-4; the check runs
last: the oracle's CPU worker threads would otherwise compete with the timed host loop).
Usage (GPU box): python tools/bench_infer.py [--size 192] [--batch 2] [--no-oracle]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=192)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--no-oracle", action="store_true")
args = ap.parse_args()

from mslesions3d_amd.ssd3d import LSSD3D  # noqa: E402
from mslesions3d_amd.synth import make_batch_on_device  # noqa: E402

dev = torch.device("cuda", 0)
size = (args.size,) * 3
torch.manual_seed(970205)
model = LSSD3D(n_classes=2, input_channels=1, input_size=size, threshold=[0.1, 0.2], batch_size=args.batch).to(dev)
# a few training steps so that the running statistics and the scores are not the initial ones
from mslesions3d_amd.trainer import FusedTrainer  # noqa: E402
tr = FusedTrainer(model)
x, boxes, labels = make_batch_on_device(args.batch, size, dev, 1, seed=3)
for _ in range(3):
    tr.step(x, boxes, labels)
model.eval()
kw = dict(min_score=0.3, max_overlap=0.3, top_k=50)
with torch.no_grad():
    locs, scores = model(x)
    det = model.detect_objects(locs, scores, return_prior_index=True, **kw)
torch.cuda.synchronize()
print(f"{args.size}^3 batch {args.batch}: priors {locs.shape[1]}, detections per volume {[len(b) for b in det[0]]}")

for _ in range(3):
    model.predict_step({"img": x})
torch.cuda.synchronize()
t0 = time.perf_counter()
nb = 0
for _ in range(args.steps):
    out = model.predict_step({"img": x})
    nb += sum(len(b) for b in out[0])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"predict_step: {dt / args.steps * 1e3:.2f} ms per batch of {args.batch} -> {args.batch * args.steps / dt:.0f} volumes/s, "
      f"{nb / dt:.0f} boxes/s (fp32, host sync per batch for the detection counts)")

if not args.no_oracle:
    from oracle import detect as odet  # noqa: E402  (checker only)
    from oracle.network import OracleSSD3D  # noqa: E402
    om = OracleSSD3D(2, 1, size, emulate_reference_init=False)
    om.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    om.eval()
    t0 = time.perf_counter()
    with torch.no_grad():
        ol, osc = om(x.cpu())
        ob, olab, oscore, oprior = odet.detect_objects(ol, osc, om.priors_cxcycz, return_prior_index=True, **kw)
    print(f"oracle forward+detect on the CPU: {time.perf_counter() - t0:.1f} s")
    print(f"locs max err {float((locs.cpu() - ol).abs().max()):.2e}, scores max err {float((scores.cpu() - osc).abs().max()):.2e}")
    for i in range(args.batch):
        assert torch.equal(det[3][i].cpu().long(), torch.as_tensor(oprior[i]).long()), f"keep-list differs in volume {i}"
        err = (det[0][i].cpu() - torch.as_tensor(ob[i])).abs().max().item() if len(ob[i]) else 0.0
        assert err <= 1e-4, err
    print("parity: NMS keep-lists bit-exact, boxes within 1e-4")
