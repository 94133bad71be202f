"""Whole-model forward / loss / every gradient element against CPU autograd of the oracle at unusual shapes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.golden import detinit
from tests.util import oracle_model
from oracle import multibox as OMB
from mslesions3d_amd.ssd3d import LSSD3D
DEV = "cuda"
def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
SHAPES = ((1, 1, (128, 96, 160)), (2, 1, (128, 128, 128))) if os.environ.get("BIG") else ((1, 1, (64, 64, 64)), (3, 1, (32, 64, 96)), (1, 2, (96, 64, 32)), (5, 1, (64, 64, 64)), (2, 3, (64, 96, 64)),
                     (1, 1, (128, 96, 160)))
for n, cin, size in SHAPES:
    m = LSSD3D(n_classes=2, input_channels=cin, input_size=size, threshold=[0.1, 0.2])
    m.load_state_dict(detinit.fill_state_dict(m.state_dict(), 1234))
    m = m.to(DEV).train()
    m._engine.fuse_stem = os.environ.get("FUSE", "1") == "1"
    o = oracle_model(cin, size).train()
    x = detinit.make_volume_batch(5, n, cin, size)
    boxes, labels = detinit.make_gt(8, n, size)
    ol, osc = o(x)
    oc, olc = OMB.multibox_loss(ol, osc, boxes, labels, o.priors_cxcycz, [0.1, 0.2])
    (oc + olc).backward()
    l, s = m(x.to(DEV))
    c, lc = m.loss_fn(l, s, [b.to(DEV) for b in boxes], [t.to(DEV) for t in labels])
    (c + lc).backward()
    og = dict((k, p.grad) for k, p in o.named_parameters())
    worst = sorted(((relerr(p.grad, og[k]), k) for k, p in m.named_parameters() if og[k] is not None), reverse=True)[:2]
    ok = relerr(l, ol) < 1e-4 and relerr(s, osc) < 1e-4 and abs(c.item() - oc.item()) < 1e-4 * abs(oc.item()) and worst[0][0] < 2e-3
    print(f"N={n} cin={cin} size={size}: locs {relerr(l, ol):.1e} scores {relerr(s, osc):.1e} conf {c.item():.6f}/{oc.item():.6f} "
          f"loc {lc.item():.6f}/{olc.item():.6f} worst grads {[(f'{e:.1e}', k) for e, k in worst]} {'OK' if ok else '<-- MISMATCH'}", flush=True)
