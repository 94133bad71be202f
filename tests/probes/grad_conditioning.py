"""Is a HIP-vs-oracle gradient difference a bug or fp32 conditioning?  Run the oracle in fp64 as ground truth and compare both
fp32 implementations (oracle fp32 on the CPU, HIP kernels) against it, per parameter tensor."""
import sys, os, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.golden import detinit
from tests.util import oracle_model
from oracle import multibox as OMB
from mslesions3d_amd.ssd3d import LSSD3D
DEV = "cuda"
n, cin, size = 2, 1, (128, 128, 128)
x = detinit.make_volume_batch(5, n, cin, size)
boxes, labels = detinit.make_gt(8, n, size)
def oracle_grads(dtype):
    o = oracle_model(cin, size).to(dtype).train()
    ol, osc = o(x.to(dtype))
    oc, olc = OMB.multibox_loss(ol, osc, [b.to(dtype) for b in boxes], labels, o.priors_cxcycz.to(dtype), [0.1, 0.2])
    (oc + olc).backward()
    return {k: p.grad.double() for k, p in o.named_parameters() if p.grad is not None}
g64, g32 = oracle_grads(torch.float64), oracle_grads(torch.float32)
m = LSSD3D(n_classes=2, input_channels=cin, input_size=size, threshold=[0.1, 0.2])
m.load_state_dict(detinit.fill_state_dict(m.state_dict(), 1234))
m = m.to(DEV).train()
m._engine.multi_stream = os.environ.get("MS", "1") == "1"
m._engine.fold_np_max = int(os.environ.get("FOLD", "32"))
l, s = m(x.to(DEV))
c, lc = m.loss_fn(l, s, [b.to(DEV) for b in boxes], [t.to(DEV) for t in labels])
(c + lc).backward()
gh = {k: p.grad.double().cpu() for k, p in m.named_parameters() if p.grad is not None}
rows = []
for k in g64:
    ref = g64[k]
    e32 = float((g32[k] - ref).abs().max() / ref.abs().max())
    eh = float((gh[k] - ref).abs().max() / ref.abs().max())
    rows.append((max(e32, eh), k, e32, eh))
rows.sort(reverse=True)
print("worst tensors: max|err| / max|ref| against the fp64 oracle   [oracle fp32 | HIP fp32]")
for _, k, e32, eh in rows[:8]:
    print(f"  {k:40s} {e32:.2e} | {eh:.2e}")
print("HIP worse than 3x the fp32 oracle's own error on:", [k for _, k, e32, eh in rows if eh > 3 * e32 + 1e-6][:10])

k = "base.features.5.conv2.weight"
d = (gh[k] - g64[k]).abs().view(256, 256)
print("error structure of", k, ": max per-row top5", torch.topk(d.max(1).values, 5), "\nmax per-col top5", torch.topk(d.max(0).values, 5))
print("ref max", float(g64[k].abs().max()), "rows with err>1e-4*max:", int((d.max(1).values > 1e-4 * g64[k].abs().max()).sum()),
      "cols:", int((d.max(0).values > 1e-4 * g64[k].abs().max()).sum()))
for kk in ("base.features.5.bn2.weight", "base.features.5.bn2.bias", "base.features.5.bn1.weight", "base.features.5.bn1.bias", "base.features.6.conv1.weight", "base.features.6.bn1.weight"):
    dd = (gh[kk] - g64[kk]).abs().view(-1)
    print(kk, "max err", float(dd.max() / g64[kk].abs().max()), "argmax", int(dd.argmax()))

pl = m._engine.plan_for(x.to(DEV), True)
v = pl.bn_y[5].cpu()
ratio = (v[2].abs() * v[3])
print("block-5 bn2: |mean|*invstd per channel: median %.2f, max %.2f at channel %d; channel 136: mean %.4g invstd %.4g ratio %.2f scale %.4g shift %.4g"
      % (float(ratio.median()), float(ratio.max()), int(ratio.argmax()), float(v[2][136]), float(v[3][136]), float(ratio[136]), float(v[0][136]), float(v[1][136])))
y5 = pl.y[5][:, 136].double().cpu()
bn = y5 * float(v[0][136]) + float(v[1][136])
print("channel 136: fraction of |bn(y)| < 1e-4:", float((bn.abs() < 1e-4).double().mean()), " min |bn|", float(bn.abs().min()), " positive fraction", float((bn > 0).double().mean()))
