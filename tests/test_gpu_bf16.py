"""bf16 activation path (BASELINE configs[3]: 192^3, batch 2, bf16 inference; a build-side extension - the reference is
fp32).  Kernel level: each bf16 kernel against stock torch fp32 ops applied to the SAME bf16-rounded inputs
(only the accumulation order and the final rounding differ).  Model level: the bf16 eval forward + decode + 3-D NMS
against the fp32 CPU oracle - a tolerance test; the keep-list disagreement is REPORTED, not hidden."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mslesions3d_amd import _lib
from mslesions3d_amd._lib import ptr
from oracle import detect as OD
from tests.golden import detinit
from tests.test_gpu_kernels import K, close, rnd, st  # noqa: F401  (K keeps device copies alive)
from tests.test_gpu_kernels import _release_kept  # noqa: F401  (autouse fixture)

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF_EPS = 2.0 ** -8  # half a bf16 ulp, relative


def bfr(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("cin,dims,stride", [(1, (16, 16, 16), (2, 2, 2)), (2, (10, 12, 20), (1, 2, 2)), (1, (9, 11, 13), (2, 2, 2))])
def test_stem_fwd_bf16(cin, dims, stride):
    L = _lib.load()
    N = 2
    x, w = rnd(N, cin, *dims, seed=1), rnd(32, cin, 3, 3, 3, seed=2, scale=0.3)
    ref = F.conv3d(x, w, stride=stride, padding=1)
    y = torch.zeros(ref.shape, dtype=torch.bfloat16, device=DEV)
    od, oh, ow = ref.shape[2:]
    NP = L.msl_stem_conv_fwd_num_partials(N, od, oh, ow)
    part = torch.zeros(2 * 32 * NP, dtype=torch.float64, device=DEV)
    _lib.call("msl_stem_conv_fwd_bf16", ptr(K(x)), ptr(K(w)), ptr(y), ptr(part), N, cin, *dims, *stride, st())
    close(y.float(), ref, 2 * BF_EPS, 1e-5, "stem fwd bf16")
    assert torch.equal(y.float().cpu(), bfr(y.float().cpu()))
    p = part.view(2, 32, NP).sum(-1).cpu()
    close(p[0], ref.double().sum((0, 2, 3, 4)), 1e-5, 1e-3, "stem sum (from the fp32 accumulators)")


@pytest.mark.parametrize("N,C,dims,stride", [(1, 4, (10, 40, 40), 2), (2, 8, (9, 24, 24), 1), (1, 3, (7, 6, 6), 1),
                                             (2, 16, (12, 12, 12), 2), (1, 2, (5, 7, 9), 2), (1, 2, (3, 96, 96), 2)])
@pytest.mark.parametrize("affine", [True, False])
def test_dw_fwd_bf16(N, C, dims, stride, affine):
    L = _lib.load()
    x = bfr(rnd(N, C, *dims, seed=4))
    w = rnd(C, 1, 3, 3, 3, seed=5, scale=0.4)
    sc, sh = rnd(C, seed=6).abs() + 0.5, rnd(C, seed=7, scale=0.3)
    a = torch.relu(x * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)) if affine else x
    ref = F.conv3d(a, w, stride=stride, padding=1, groups=C)
    y = torch.full(ref.shape, float("nan"), dtype=torch.bfloat16, device=DEV)
    NP = L.msl_dwconv_fwd_bf16_num_partials(N, C, *dims, stride)
    part = torch.zeros(2 * C * NP, dtype=torch.float64, device=DEV)
    _lib.call("msl_dwconv_fwd_bf16", ptr(K(x.to(torch.bfloat16))), ptr(K(sc)) if affine else None, ptr(K(sh)) if affine else None,
              ptr(K(w)), ptr(y), ptr(part), N, C, *dims, stride, st())
    close(y.float(), ref, 2 * BF_EPS, 1e-5, "dw fwd bf16")
    p = part.view(2, C, NP).sum(-1).cpu()
    close(p[0], ref.double().sum((0, 2, 3, 4)), 1e-4, 1e-3, "dw sum")
    close(p[1], (ref.double() ** 2).sum((0, 2, 3, 4)), 1e-4, 1e-3, "dw sumsq")


@pytest.mark.parametrize("N,Cin,Cout,S", [(2, 32, 64, 1000), (1, 64, 128, 64), (2, 128, 128, 130), (1, 512, 512, 8),
                                          (1, 256, 512, 27), (2, 128, 256, 1728), (1, 32, 96, 200)])
def test_pw_fwd_bf16(N, Cin, Cout, S):
    """Pointwise GEMM on v_mfma_f32_32x32x16_bf16: operands rounded to bf16 (activation after its fp32 affine + ReLU,
    weights), fp32 accumulation -> equals the fp32 product of the rounded operands up to summation order."""
    L = _lib.load()
    z = bfr(rnd(N, Cin, S, seed=10))
    w = rnd(Cout, Cin, seed=11) / Cin ** 0.5
    sc, sh = rnd(Cin, seed=12).abs() + 0.5, rnd(Cin, seed=13, scale=0.3)
    a = bfr(torch.relu(z * sc.view(1, -1, 1) + sh.view(1, -1, 1)))
    ref = torch.einsum("oc,ncs->nos", bfr(w).double(), a.double()).float()
    y = torch.full(ref.shape, float("nan"), dtype=torch.bfloat16, device=DEV)
    NP = L.msl_pwconv_fwd_bf16_num_partials(N, S)
    part = torch.zeros(2 * Cout * NP, dtype=torch.float64, device=DEV)
    _lib.call("msl_pwconv_fwd_bf16", ptr(K(z.to(torch.bfloat16))), ptr(K(sc)), ptr(K(sh)), ptr(K(w)), ptr(y), ptr(part), N, Cin,
              Cout, S, st())
    # the activation operand is rounded to bf16 AFTER its fp32 affine (an fma here, a multiply + add in torch): a value next
    # to a rounding boundary may land one bf16 ulp apart, i.e. one product term moves by 0.4 % -> absolute slack
    close(y.float(), ref, 2 * BF_EPS, 4e-3 * float(ref.abs().max()), "pw fwd bf16")
    p = part.view(2, Cout, NP).sum(-1).cpu()
    close(p[0], ref.double().sum((0, 2)), 1e-3, 1e-2, "pw sum")


@pytest.mark.parametrize("N,C,dims", [(2, 128, (8, 8, 8)), (1, 32, (3, 5, 6)), (2, 256, (6, 6, 6)), (1, 64, (12, 12, 12))])
def test_materialize_and_head_fwd_bf16(N, C, dims):
    L = _lib.load()
    ncls = 2
    yraw = bfr(rnd(N, C, *dims, seed=30))
    sc, sh = rnd(C, seed=31).abs() + 0.5, rnd(C, seed=32, scale=0.3)
    act = torch.relu(yraw * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1))
    lw = rnd(12, C, 3, 3, 3, seed=33) / (27 * C) ** 0.5
    cw = rnd(2 * ncls, C, 3, 3, 3, seed=34) / (27 * C) ** 0.5
    lb, cb = rnd(12, seed=35, scale=0.1), rnd(2 * ncls, seed=36, scale=0.1)
    S = dims[0] * dims[1] * dims[2]
    pad = torch.zeros((N,) + tuple(d + 2 for d in dims) + (C,), dtype=torch.bfloat16, device=DEV)
    plain = torch.full((N, C) + dims, float("nan"), device=DEV)
    _lib.call("msl_bn_relu_materialize_bf16", ptr(K(yraw.to(torch.bfloat16))), ptr(K(sc)), ptr(K(sh)), ptr(plain), ptr(pad), N, C,
              *dims, st())
    close(plain, act, 1e-6, 1e-6, "materialised activation (fp32 copy)")
    inner = pad[:, 1:-1, 1:-1, 1:-1, :].permute(0, 4, 1, 2, 3).float()
    close(inner, act, BF_EPS, 1e-6, "channels-last bf16 copy (rounded once, from the fp32 fma)")
    a_b = inner.cpu()  # the head reference uses exactly what the kernel reads
    rl = F.conv3d(a_b.double(), bfr(lw).double(), lb.double(), padding=1).permute(0, 2, 3, 4, 1).reshape(N, -1, 6).float()
    rc = F.conv3d(a_b.double(), bfr(cw).double(), cb.double(), padding=1).permute(0, 2, 3, 4, 1).reshape(N, -1, ncls).float()
    assert float(pad[:, 0].abs().max()) == 0 and float(pad[:, :, :, -1].abs().max()) == 0, "halo must stay zero"
    Wp = torch.empty(L.msl_head_packed_weight_bf16_elems(C), dtype=torch.bfloat16, device=DEV)
    _lib.call("msl_head_pack_weights_bf16", ptr(K(lw)), ptr(K(cw)), ptr(Wp), C, ncls, st())
    off, Ptot = 10, 2 * S + 14
    locs = torch.full((N, Ptot, 6), 7.0, device=DEV)
    scores = torch.full((N, Ptot, ncls), 7.0, device=DEV)
    _lib.call("msl_head_conv_fwd_bf16", ptr(pad), ptr(Wp), ptr(K(lb)), ptr(K(cb)), ptr(locs), ptr(scores), N, C, *dims, Ptot, off,
              ncls, st())
    close(locs[:, off:off + 2 * S], rl, 1e-4, 1e-4, "head locs bf16")
    close(scores[:, off:off + 2 * S], rc, 1e-4, 1e-4, "head scores bf16")
    assert bool((locs[:, :off] == 7).all()) and bool((locs[:, off + 2 * S:] == 7).all())


def _models(size, seed=1234):
    from mslesions3d_amd.ssd3d import LSSD3D
    from oracle.network import OracleSSD3D
    m = LSSD3D(n_classes=2, input_channels=1, input_size=size, threshold=[0.1, 0.2])
    m.load_state_dict(detinit.fill_state_dict(m.state_dict(), seed))
    m = m.to(DEV).eval()
    om = OracleSSD3D(2, 1, size, emulate_reference_init=False)
    om.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    return m, om.eval()


@pytest.mark.parametrize("size,n", [((64, 64, 64), 2), ((192, 192, 192), 2)])
def test_bf16_inference_against_the_fp32_oracle(size, n):
    """BASELINE configs[3] (192^3, batch 2, bf16 inference) - and 64^3 as the small case - against the fp32 CPU oracle.
    Tolerances (bf16 has 8 significant bits; 15 conv layers): locs / scores within 3e-2 of the tensor's max magnitude,
    softmax probabilities within 2e-2; detections: every oracle detection has a bf16 detection of IoU >= 0.5 on >= 90 % of
    the kept boxes; the exact keep-list disagreement is printed."""
    m, om = _models(size)
    x = detinit.make_volume_batch(9, n, 1, size)
    with torch.no_grad():
        ol, osc = om(x)
        m.compute_dtype = "f32"
        fl, fs = (t.clone() for t in m(x.to(DEV)))
        m.compute_dtype = "bf16"
        bl, bs = (t.clone() for t in m(x.to(DEV)))
    assert float((fl.cpu() - ol).abs().max()) <= 1e-4 * float(ol.abs().max()) + 1e-6
    el = float((bl.cpu() - ol).abs().max() / ol.abs().max())
    es = float((bs.cpu() - osc).abs().max() / osc.abs().max())
    ep = float((torch.softmax(bs.cpu(), 2) - torch.softmax(osc, 2)).abs().max())
    print(f"[bf16 {size[0]}^3 x{n}] max |locs - fp32 oracle| / max |locs| = {el:.2e}, scores {es:.2e}, softmax prob {ep:.2e}")
    assert el <= 3e-2 and es <= 3e-2 and ep <= 2e-2
    kw = dict(min_score=0.3, max_overlap=0.3, top_k=50)
    with torch.no_grad():
        b, l, s, pi = m.detect_objects(bl, bs, return_prior_index=True, **kw)
    ob, olab, oscore, oi = OD.detect_objects(ol, osc, om.priors_cxcycz, 0.3, 0.3, 50, return_prior_index=True)
    from mslesions3d_amd.utils import _iou_one_to_many
    for i in range(n):
        same = len(set(pi[i].cpu().tolist()) & set(oi[i].tolist()))
        hit = 0
        for k in range(len(ob[i])):
            if int(olab[i][k]) == 0:
                hit += int(len(b[i]) == 1 and int(l[i][0]) == 0)
                continue
            iou = _iou_one_to_many(ob[i][k].numpy(), b[i].cpu().numpy().reshape(-1, 6))
            hit += int(np.nanmax(iou) >= 0.5) if len(iou) else 0
        print(f"[bf16 {size[0]}^3 image {i}] oracle keeps {len(oi[i])}, bf16 keeps {len(pi[i])}, identical prior indices {same}, "
              f"oracle detections matched at IoU >= 0.5: {hit}")
        assert hit >= 0.9 * len(ob[i])


def test_bf16_predict_step_replays_and_training_refuses():
    m, _ = _models((64, 64, 64))
    m.compute_dtype = "bf16"
    m.min_score, m.max_overlap, m.top_k = 0.3, 0.3, 20
    outs = []
    for seed in (11, 12, 11):
        outs.append(m.predict_step({"img": detinit.make_volume_batch(seed, 2, 1, (64, 64, 64))}))
    for a, b in zip(outs[0], outs[2]):  # replayed program on the same input: identical detections
        for u, v in zip(a, b):
            assert torch.equal(u, v)
    assert len(m._pred_programs) == 1
    m.train()
    with pytest.raises(NotImplementedError, match="bf16"):
        m(detinit.make_volume_batch(5, 2, 1, (64, 64, 64)).to(DEV))
