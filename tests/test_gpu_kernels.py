"""Per-kernel parity on a real MI355X: every C-ABI entry point against stock torch fp32 ops on the CPU
(the same third-party arithmetic the reference calls).  Tolerances are stated per test; integer outputs exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mslesions3d_amd import _lib
from mslesions3d_amd._lib import ptr

pytestmark = pytest.mark.gpu
DEV = "cuda"


def st():
    return torch.cuda.current_stream().cuda_stream


_KEEP = []


def K(t):
    """Move to the GPU and keep the tensor alive until the end of the test (a temporary passed as
    ``ptr(K(x))`` would be freed — and its block re-used — before the kernel runs)."""
    d = t.detach().to(DEV).contiguous()
    _KEEP.append(d)
    return d


@pytest.fixture(autouse=True)
def _release_kept():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(a, b, rtol, atol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs()
    lim = atol + rtol * b.abs()
    bad = err > lim
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} elements off; max abs err {err.max().item():.3e}, "
                           f"max |ref| {b.abs().max().item():.3e}, worst idx {int(err.argmax())}")


def stats_from_partials(part, C, NP):
    p = part[:2 * C * NP].view(2, C, NP).sum(-1).cpu()
    return p[0], p[1]


BF_EPS = 2.0 ** -8  # half a bf16 ulp, relative


def affine_act(x, sc, sh):
    return torch.relu(x * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1))


# ------------------------------------------------------------------------------------------------- stem
@pytest.mark.parametrize("cin,dims,stride", [(1, (16, 16, 16), (2, 2, 2)), (2, (10, 12, 20), (1, 2, 2)),
                                             (1, (9, 11, 13), (2, 2, 2)), (3, (8, 8, 8), (2, 2, 2)),
                                             # W % 4 == 0, OW % 32 == 0, Cin <= 2: the row-staged kernel (whole and half
                                             # chunks, first / last planes and rows, depth stride 1, two input channels)
                                             (1, (8, 12, 64), (2, 2, 2)), (1, (6, 10, 128), (2, 2, 2)), (2, (5, 8, 192), (1, 2, 2)),
                                             (1, (3, 3, 256), (2, 2, 2)), (2, (7, 9, 320), (2, 2, 2))])
def test_stem_fwd(cin, dims, stride):
    L = _lib.load()
    N = 2
    x, w = rnd(N, cin, *dims, seed=1), rnd(32, cin, 3, 3, 3, seed=2, scale=0.3)
    ref = F.conv3d(x, w, stride=stride, padding=1)
    y = torch.empty(ref.shape, device=DEV)
    od, oh, ow = ref.shape[2:]
    NP = L.msl_stem_conv_fwd_num_partials(N, od, oh, ow)
    part = torch.zeros(2 * 32 * NP, dtype=torch.float64, device=DEV)
    _lib.call("msl_stem_conv_fwd", ptr(K(x)), ptr(K(w)), ptr(y), ptr(part), N, cin, *dims, *stride, st())
    close(y, ref, 1e-5, 1e-5, "stem fwd")
    s, q = stats_from_partials(part, 32, NP)
    close(s, ref.double().sum((0, 2, 3, 4)), 1e-5, 1e-3, "stem sum")
    close(q, (ref.double() ** 2).sum((0, 2, 3, 4)), 1e-5, 1e-3, "stem sumsq")


@pytest.mark.parametrize("cin,dims,stride", [(1, (16, 16, 16), (2, 2, 2)), (2, (6, 12, 20), (1, 2, 2)),
                                             (1, (8, 8, 140), (2, 2, 2))])
def test_stem_bwd_weight(cin, dims, stride):
    L = _lib.load()
    N = 2
    x = rnd(N, cin, *dims, seed=1)
    w = rnd(32, cin, 3, 3, 3, seed=2, scale=0.3).requires_grad_(True)
    out = F.conv3d(x, w, stride=stride, padding=1)
    dy = rnd(*out.shape, seed=3)
    out.backward(dy)
    dw = torch.empty((32, cin, 3, 3, 3), device=DEV)
    ws = torch.empty(L.msl_stem_conv_bwd_weight_workspace_bytes(cin) // 4, device=DEV)
    _lib.call("msl_stem_conv_bwd_weight", ptr(K(dy)), ptr(K(x)), ptr(dw), ptr(ws), N, cin, *dims, *stride, st())
    close(dw, w.grad, 1e-4, 1e-3, "stem bwd weight")


# ------------------------------------------------------------------------------------------------- depthwise
DW_CASES = [  # (N, C, dims, stride, expected variant)
    (1, 4, (10, 40, 40), 2, 1), (1, 3, (9, 40, 48), 1, 1), (1, 2, (7, 96, 96), 2, 1),
    (2, 16, (32, 16, 32), 2, 2), (1, 4, (6, 12, 16), 1, 2), (2, 8, (12, 24, 24), 2, 2),
    (1, 2, (6, 64, 64), 2, 3),
    # register-marching wave kernel, stride 2 (8^2 / 16^2 / 32^2 planes): shared and split planes, odd depth, ragged slab
    (2, 8, (16, 16, 16), 2, 3), (2, 64, (8, 8, 8), 2, 3), (2, 16, (32, 32, 32), 2, 3), (1, 4, (7, 16, 16), 2, 3),
    (1, 8, (5, 8, 8), 2, 3), (1, 2, (18, 32, 32), 2, 3), (3, 8, (2, 8, 8), 2, 3),
    # register-marching wave kernel (stride 1, 4^2 / 8^2 / 16^2 planes): every slab length, ragged last slab
    (2, 8, (16, 16, 16), 1, 3), (2, 64, (4, 4, 4), 1, 3), (1, 8, (5, 8, 8), 1, 3), (1, 4, (3, 16, 16), 1, 3),
    (2, 16, (9, 4, 4), 1, 3), (1, 8, (20, 8, 8), 1, 3), (1, 2, (1, 16, 16), 1, 3), (3, 32, (2, 4, 4), 1, 3),
    (2, 4, (6, 6, 6), 1, 0), (2, 4, (4, 4, 4), 2, 0), (1, 4, (5, 7, 9), 2, 0),
]


@pytest.mark.parametrize("N,C,dims,stride,variant", DW_CASES)
@pytest.mark.parametrize("affine", [True, False])
def test_dw_fwd(N, C, dims, stride, variant, affine):
    L = _lib.load()
    assert L.msl_dwconv_fwd_variant(N, C, *dims, stride) == variant
    x, w = rnd(N, C, *dims, seed=4), rnd(C, 1, 3, 3, 3, seed=5, scale=0.4)
    sc, sh = rnd(C, seed=6).abs() + 0.5, rnd(C, seed=7, scale=0.3)
    a = affine_act(x, sc, sh) if affine else x
    ref = F.conv3d(a, w, stride=stride, padding=1, groups=C)
    for force_naive in ([0, 1] if variant != 0 else [0]):
        y = torch.full(ref.shape, float("nan"), device=DEV)
        NP = L.msl_dwconv_fwd_num_partials(N, C, *dims, stride) if not force_naive else 4096
        part = torch.zeros(2 * C * max(NP, 4096), dtype=torch.float64, device=DEV)
        _lib.call("msl_dwconv_fwd", ptr(K(x)), ptr(K(sc)) if affine else None, ptr(K(sh)) if affine else None,
                  ptr(K(w)), ptr(y), ptr(part), N, C, *dims, stride, force_naive, st())
        close(y, ref, 1e-5, 1e-5, f"dw fwd (naive={force_naive})")
        if not force_naive:
            s, q = stats_from_partials(part, C, NP)
            close(s, ref.double().sum((0, 2, 3, 4)), 1e-5, 1e-3, "dw sum")
            close(q, (ref.double() ** 2).sum((0, 2, 3, 4)), 1e-5, 1e-3, "dw sumsq")


@pytest.mark.parametrize("N,C,dims,stride", [(2, 8, (10, 96, 96), 2), (1, 16, (10, 48, 48), 2), (1, 4, (7, 24, 24), 2),
                                             (1, 3, (5, 12, 12), 2), (1, 2, (6, 20, 40), 2), (2, 3, (2, 6, 8), 2),
                                             (1, 2, (17, 10, 256), 2), (1, 5, (9, 96, 24), 2),
                                             (2, 8, (9, 24, 24), 1), (1, 16, (12, 12, 12), 1), (1, 3, (5, 7, 20), 1),
                                             (1, 2, (3, 33, 96), 1), (2, 4, (2, 5, 8), 1), (1, 2, (6, 9, 256), 1)])
@pytest.mark.parametrize("affine", [True, False])
def test_dw_fwd_eval_rows(N, C, dims, stride, affine):
    """Statistics-free (eval-mode) forward on dw_s2_rows_eval_kernel / dw_s1_rows_eval_kernel: planes that are not powers of
    two (the 192^3 inference maps 96^2 ... 12^2), rows of 24 / 12 / 6 / 3 / 10 / 2 / 5 / 64 lanes, ragged row groups and
    slabs, odd sizes."""
    L = _lib.load()
    assert L.msl_dwconv_fwd_eval_rows_ok(N, C, *dims, stride) == 1
    assert L.msl_dwconv_fwd_eval_rows_ok(1, 2, 6, 64, 64, 2) == 0 and L.msl_dwconv_fwd_eval_rows_ok(1, 2, 6, 12, 6, 1) == 0
    x, w = rnd(N, C, *dims, seed=4), rnd(C, 1, 3, 3, 3, seed=5, scale=0.4)
    sc, sh = rnd(C, seed=6).abs() + 0.5, rnd(C, seed=7, scale=0.3)
    a = affine_act(x, sc, sh) if affine else x
    ref = F.conv3d(a, w, stride=stride, padding=1, groups=C)
    y = torch.full(ref.shape, float("nan"), device=DEV)
    _lib.call("msl_dwconv_fwd", ptr(K(x)), ptr(K(sc)) if affine else None, ptr(K(sh)) if affine else None,
              ptr(K(w)), ptr(y), None, N, C, *dims, stride, 0, st())
    close(y, ref, 1e-5, 1e-5, "dw fwd (eval rows)")
    # the kernel a training forward of the same shape runs (with statistics) agrees to rounding
    NP = L.msl_dwconv_fwd_num_partials(N, C, *dims, stride)
    part = torch.zeros(2 * C * max(NP, 4096), dtype=torch.float64, device=DEV)
    y2 = torch.full(ref.shape, float("nan"), device=DEV)
    _lib.call("msl_dwconv_fwd", ptr(K(x)), ptr(K(sc)) if affine else None, ptr(K(sh)) if affine else None,
              ptr(K(w)), ptr(y2), ptr(part), N, C, *dims, stride, 0, st())
    close(y2, y, 1e-6, 1e-6, "eval rows vs the training-mode kernel")


@pytest.mark.parametrize("N,C,dims,stride", [(2, 512, (6, 6, 6), 1), (2, 7, (6, 6, 6), 2), (1, 5, (3, 5, 7), 1), (3, 3, (8, 8, 6), 2),
                                             (1, 2, (1, 1, 1), 1), (1, 9, (5, 6, 14), 1)])
@pytest.mark.parametrize("affine", [True, False])
def test_dw_fwd_eval_small_maps(N, C, dims, stride, affine):
    """Statistics-free forward of maps of at most 512 voxels that no wave / rows kernel takes (the 6^3 map of a 192^3 volume):
    dw_small_eval_kernel equals the one-output-per-thread fallback bit for bit and torch's convolution to rounding."""
    L = _lib.load()
    x, w = rnd(N, C, *dims, seed=4), rnd(C, 1, 3, 3, 3, seed=5, scale=0.4)
    sc, sh = rnd(C, seed=6).abs() + 0.5, rnd(C, seed=7, scale=0.3)
    a = affine_act(x, sc, sh) if affine else x
    ref = F.conv3d(a, w, stride=stride, padding=1, groups=C)
    y = torch.full(ref.shape, float("nan"), device=DEV)
    y_naive = torch.full(ref.shape, float("nan"), device=DEV)
    for out, naive in ((y, 0), (y_naive, 1)):
        _lib.call("msl_dwconv_fwd", ptr(K(x)), ptr(K(sc)) if affine else None, ptr(K(sh)) if affine else None,
                  ptr(K(w)), ptr(out), None, N, C, *dims, stride, naive, st())
    close(y, ref, 1e-5, 1e-5, "dw fwd (small map)")
    assert torch.equal(y, y_naive)


@pytest.mark.parametrize("N,C,dims,in_np,stride", [(2, 8, (16, 16, 16), 6, 1), (2, 8, (8, 8, 8), 100, 1),
                                                   (2, 32, (4, 4, 4), 70, 1), (1, 16, (5, 4, 4), 9, 1),
                                                   (2, 4, (6, 12, 16), 70, 1), (2, 16, (8, 8, 8), 70, 2),
                                                   (2, 4, (8, 16, 16), 5, 2), (1, 2, (8, 32, 32), 130, 2),
                                                   (1, 2, (6, 64, 64), 70, 2), (1, 2, (5, 96, 96), 70, 2)])
def test_dw_fwd_fold_matches_explicit_affine(N, C, dims, in_np, stride):
    """The in-kernel BatchNorm fold (serial for NP <= 64, wave tree above) gives the bits of finalize + explicit vectors."""
    L = _lib.load()
    x, w = rnd(N, C, *dims, seed=14), rnd(C, 1, 3, 3, 3, seed=15, scale=0.4)
    gamma, beta = rnd(C, seed=16).abs() + 0.5, rnd(C, seed=17, scale=0.3)
    count = 1000.0
    g = torch.Generator().manual_seed(18)
    ps = torch.randn(C, in_np, generator=g, dtype=torch.float64) * 3.0
    pq = ps.abs() * 2.0 + torch.rand(C, in_np, generator=g, dtype=torch.float64) * 40.0 + 20.0
    part_in = torch.stack([ps, pq]).contiguous().to(DEV)
    vec = torch.zeros(4 * C, device=DEV)  # scale | shift | mean | invstd
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
    _lib.call("msl_bn_finalize", ptr(part_in), in_np, count, ptr(K(gamma)), ptr(K(beta)), ptr(rm), ptr(rv), ptr(nbt), 0.1,
              1e-5, ptr(vec), ptr(vec[C:]), ptr(vec[2 * C:]), ptr(vec[3 * C:]), C, st())
    NP = L.msl_dwconv_fwd_num_partials(N, C, *dims, stride)
    out = []
    for fold in (False, True):
        y = torch.full((N, C) + tuple((d - 1) // stride + 1 for d in dims), float("nan"), device=DEV)
        part = torch.zeros(2 * C * NP, dtype=torch.float64, device=DEV)
        if fold:
            _lib.call("msl_dwconv_fwd_fold", ptr(K(x)), ptr(part_in), in_np, count, ptr(K(gamma)), ptr(K(beta)), 1e-5,
                      ptr(K(w)), ptr(y), ptr(part), N, C, *dims, stride, st())
        else:
            _lib.call("msl_dwconv_fwd", ptr(K(x)), ptr(vec), ptr(vec[C:]), ptr(K(w)), ptr(y), ptr(part), N, C, *dims,
                      stride, 0, st())
        out.append((y, part))
    torch.cuda.synchronize()
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    sc, sh = vec[:C].cpu(), vec[C:2 * C].cpu()
    ref = F.conv3d(affine_act(x, sc, sh), w, stride=stride, padding=1, groups=C)
    close(out[1][0], ref, 1e-5, 1e-5, "dw fwd fold")


@pytest.mark.parametrize("N,C,dims,stride", [(2, 4, (8, 16, 16), 2), (2, 4, (8, 16, 16), 1), (1, 3, (5, 7, 9), 2),
                                              (1, 3, (5, 7, 9), 1), (1, 2, (9, 12, 20), 2),
                                              # LDS-tiled bwd-weight: streamed planes (1/4 items per thread), resident
                                              # slabs with several channels per workgroup, odd depth
                                              (1, 4, (10, 40, 40), 2), (1, 3, (9, 40, 48), 1), (1, 2, (7, 96, 96), 2),
                                              (2, 64, (8, 8, 8), 2), (2, 64, (4, 4, 4), 1), (2, 16, (32, 32, 32), 2),
                                              (1, 8, (5, 8, 8), 1), (2, 8, (16, 16, 16), 1), (1, 4, (6, 12, 16), 1),
                                              (2, 16, (32, 16, 32), 2), (1, 8, (20, 8, 8), 1),
                                              # weight gradient on the wave kernels: split planes, odd depth, one plane
                                              (1, 2, (6, 64, 64), 2), (1, 4, (7, 16, 16), 2), (3, 8, (2, 8, 8), 2),
                                              (1, 2, (1, 16, 16), 1), (2, 16, (9, 4, 4), 1), (1, 2, (18, 32, 32), 2)])
def test_dw_bwd(N, C, dims, stride):
    L = _lib.load()
    x = rnd(N, C, *dims, seed=4)
    w = rnd(C, 1, 3, 3, 3, seed=5, scale=0.4).requires_grad_(True)
    sc, sh = rnd(C, seed=6).abs() + 0.5, rnd(C, seed=7, scale=0.3)
    a = affine_act(x, sc, sh).requires_grad_(True)
    out = F.conv3d(a, w, stride=stride, padding=1, groups=C)
    dy = rnd(*out.shape, seed=8)
    out.backward(dy)
    g = torch.full(x.shape, float("nan"), device=DEV)
    _lib.call("msl_dwconv_bwd_data", ptr(K(dy)), ptr(K(w.detach())), ptr(g), N, C, *dims, stride, 0, st())
    close(g, a.grad, 1e-5, 1e-5, "dw bwd data")
    base = rnd(*x.shape, seed=9).to(DEV)
    g2 = base.clone()
    _lib.call("msl_dwconv_bwd_data", ptr(K(dy)), ptr(K(w.detach())), ptr(g2), N, C, *dims, stride, 1, st())
    close(g2 - base, a.grad, 1e-5, 1e-5, "dw bwd data (accumulate)")
    NP = L.msl_dwconv_bwd_weight_num_partials(N, C, *dims, stride)
    part = torch.zeros(C * 27 * NP, dtype=torch.float64, device=DEV)
    dw = torch.empty((C, 27), device=DEV)
    _lib.call("msl_dwconv_bwd_weight", ptr(K(dy)), ptr(K(x)), ptr(K(sc)), ptr(K(sh)), ptr(dw), ptr(part),
              N, C, *dims, stride, st())
    close(dw.view(C, 1, 3, 3, 3), w.grad, 1e-4, 1e-4, "dw bwd weight")


def test_fused_bn_backward_of_the_stem_tail():
    """dw bwd-data emitting the BatchNorm-backward partials + stem bwd-weight applying the BatchNorm backward on load
    == the unfused chain (bwd-data, reduce, finalize, apply, bwd-weight) on CPU autograd."""
    L = _lib.load()
    N, cin, dims = 2, 1, (12, 16, 24)
    x = rnd(N, cin, *dims, seed=1)
    w0 = rnd(32, cin, 3, 3, 3, seed=2, scale=0.3).requires_grad_(True)
    gamma = (rnd(32, seed=3).abs() + 0.5).requires_grad_(True)
    beta = rnd(32, seed=4, scale=0.2).requires_grad_(True)
    w1 = rnd(32, 1, 3, 3, 3, seed=5, scale=0.4)
    y0 = F.conv3d(x, w0, stride=2, padding=1)
    a0 = torch.relu(F.batch_norm(y0, None, None, gamma, beta, True, 0.1, 1e-5))
    z1 = F.conv3d(a0, w1, stride=2, padding=1, groups=32)
    dz = rnd(*z1.shape, seed=6)
    z1.backward(dz)
    od, oh, ow = y0.shape[2:]
    S0 = od * oh * ow
    # forward statistics of y0 -> vec rows 0..3
    yd = y0.detach().double()
    part = torch.stack([yd.sum((0, 2, 3, 4)), (yd ** 2).sum((0, 2, 3, 4))]).view(2, 32, 1).contiguous().to(DEV)
    vec = torch.zeros((6, 32), device=DEV)
    _lib.call("msl_bn_finalize", ptr(part), 1, float(N * S0), ptr(K(gamma)), ptr(K(beta)), None, None, None, 0.1, 1e-5,
              ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]), 32, st())
    g = torch.full(y0.shape, float("nan"), device=DEV)
    NP = L.msl_dwconv_bwd_data_bnreduce_num_partials(N, 32, od, oh, ow)
    bp = torch.zeros(2 * 32 * NP, dtype=torch.float64, device=DEV)
    y0d = K(y0)
    _lib.call("msl_dwconv_bwd_data_bnreduce", ptr(K(dz)), ptr(K(w1)), ptr(g), ptr(y0d), ptr(vec[0]), ptr(vec[1]), ptr(vec[2]),
              ptr(vec[3]), ptr(bp), N, 32, od, oh, ow, 2, 0, st())
    dgam, dbet = torch.empty(32, device=DEV), torch.empty(32, device=DEV)
    _lib.call("msl_bn_bwd_finalize", ptr(bp), NP, float(N * S0), ptr(dgam), ptr(dbet), ptr(vec[4]), ptr(vec[5]), 32, st())
    close(dgam, gamma.grad, 1e-4, 1e-5, "dgamma from fused reduce")
    close(dbet, beta.grad, 1e-4, 1e-5, "dbeta from fused reduce")
    dw = torch.empty((32, cin, 3, 3, 3), device=DEV)
    ws = torch.empty(L.msl_stem_conv_bwd_weight_workspace_bytes(cin) // 4, device=DEV)
    _lib.call("msl_stem_conv_bwd_weight_bnapply", ptr(g), ptr(y0d), ptr(vec), ptr(K(x)), ptr(dw), ptr(ws), N, cin, *dims, 2, 2, 2, st())
    close(dw, w0.grad, 2e-4, 1e-4, "stem dW with fused BN apply")


@pytest.mark.parametrize("cin,dims", [(1, (12, 16, 24)), (2, (16, 16, 16)), (1, (10, 14, 72)),
                                      # W == 128, H % 16 == 0: the tile-staged kernel (stem_bww_tile_kernel) - one / odd /
                                      # even plane counts, both parities at every border, two channels, and 768 tiles on
                                      # 512 workgroups (two tiles per workgroup, a quarter of the workgroups idle)
                                      (1, (2, 16, 128)), (1, (5, 48, 128)), (2, (6, 32, 128)), (1, (96, 64, 128))])
def test_stem_backward_without_materialising_the_activation_gradient(cin, dims):
    """One pass over (dL/dz_1, y_0) for the BatchNorm sums and the depthwise weight gradient, then the stem weight
    gradient rebuilding dL/d(stem activation) from dL/dz_1 on the fly == CPU autograd through
    conv -> BN -> ReLU -> depthwise conv (odd output sizes included: 5 x 7 x 36 from 10 x 14 x 72)."""
    _fused_stem_case(cin, dims)


def _fused_stem_case(cin, dims):
    L = _lib.load()
    N = 2
    x = rnd(N, cin, *dims, seed=1)
    w0 = rnd(32, cin, 3, 3, 3, seed=2, scale=0.3).requires_grad_(True)
    gamma = (rnd(32, seed=3).abs() + 0.5).requires_grad_(True)
    beta = rnd(32, seed=4, scale=0.2).requires_grad_(True)
    w1 = rnd(32, 1, 3, 3, 3, seed=5, scale=0.4).requires_grad_(True)
    y0 = F.conv3d(x, w0, stride=2, padding=1)
    a0 = torch.relu(F.batch_norm(y0, None, None, gamma, beta, True, 0.1, 1e-5))
    z1 = F.conv3d(a0, w1, stride=2, padding=1, groups=32)
    dz = rnd(*z1.shape, seed=6)
    z1.backward(dz)
    od, oh, ow = y0.shape[2:]
    S0 = od * oh * ow
    yd = y0.detach().double()
    part = torch.stack([yd.sum((0, 2, 3, 4)), (yd ** 2).sum((0, 2, 3, 4))]).view(2, 32, 1).contiguous().to(DEV)
    vec = torch.zeros((8, 32), device=DEV)
    _lib.call("msl_bn_finalize", ptr(part), 1, float(N * S0), ptr(K(gamma)), ptr(K(beta)), None, None, None, 0.1, 1e-5,
              ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]), 32, st())
    NP = L.msl_dwconv_s2_bwd_bnreduce_bww_num_partials(N, 32, od, oh, ow)
    assert NP > 0
    bp = torch.full((2 * 32 * NP,), float("nan"), dtype=torch.float64, device=DEV)
    wp = torch.full((32 * 27 * NP,), float("nan"), dtype=torch.float64, device=DEV)
    y0d, dzd, w1d = K(y0), K(dz), K(w1)
    w1t = torch.full((27, 32), float("nan"), device=DEV)
    _lib.call("msl_dwconv_s2_bwd_bnreduce_bww", ptr(dzd), ptr(w1d), ptr(y0d), ptr(vec[0]), ptr(vec[1]), ptr(vec[2]),
              ptr(vec[3]), ptr(bp), ptr(wp), ptr(w1t), N, 32, od, oh, ow, st())
    assert torch.equal(w1t.cpu(), w1.detach().view(32, 27).t())
    dgam, dbet = torch.empty(32, device=DEV), torch.empty(32, device=DEV)
    _lib.call("msl_bn_bwd_finalize_coef", ptr(bp), NP, float(N * S0), ptr(dgam), ptr(dbet), ptr(vec), 32, st())
    close(dgam, gamma.grad, 1e-4, 1e-5, "dgamma")
    close(dbet, beta.grad, 1e-4, 1e-5, "dbeta")
    dw1 = torch.empty((32, 1, 3, 3, 3), device=DEV)
    _lib.call("msl_dwconv_bwd_weight_finalize", ptr(wp), NP, ptr(dw1), 32, st())
    close(dw1, w1.grad, 1e-4, 1e-4, "depthwise dW from the fused pass")
    dw = torch.empty((32, cin, 3, 3, 3), device=DEV)
    ws = torch.empty(L.msl_stem_conv_bwd_weight_workspace_bytes(cin) // 4, device=DEV)
    _lib.call("msl_stem_conv_bwd_weight_fused", ptr(dzd), ptr(w1t), ptr(y0d), ptr(vec), ptr(K(x)), ptr(dw), ptr(ws), N, cin,
              *dims, 2, 2, 2, st())
    # fp32 accumulation over N * OD * OH * OW positions: the absolute bar scales with the tensor's magnitude
    close(dw, w0.grad, 2e-4, max(1e-4, 2e-6 * float(w0.grad.abs().max())), "stem dW with the gradient rebuilt on the fly")


# ------------------------------------------------------------------------------------------------- pointwise
@pytest.mark.parametrize("N,Cin,Cout,S", [(2, 32, 64, 1000), (1, 64, 128, 64), (2, 128, 128, 130), (1, 512, 512, 8),
                                          (1, 256, 512, 27),
                                          # S % 32 == 0: the wave-autonomous weight-gradient kernel (whole range in one
                                          # workgroup / split over slabs / odd chunk counts per wave / 32-row tiles)
                                          (4, 512, 512, 64), (4, 256, 512, 64), (2, 128, 256, 512), (3, 64, 128, 4096),
                                          (2, 32, 64, 32768), (1, 32, 96, 96), (1, 64, 32, 32), (3, 96, 160, 1120),
                                          # the column-strip kernels of blocks 1-3 (forward and bwd-data)
                                          (2, 128, 128, 4096), (4, 64, 128, 4096), (1, 32, 64, 65536)])
def test_pw_fwd_bwd(N, Cin, Cout, S):
    L = _lib.load()
    z = rnd(N, Cin, S, seed=10)
    w = (rnd(Cout, Cin, seed=11) / Cin ** 0.5).requires_grad_(True)
    sc, sh = rnd(Cin, seed=12).abs() + 0.5, rnd(Cin, seed=13, scale=0.3)
    a = torch.relu(z * sc.view(1, -1, 1) + sh.view(1, -1, 1)).requires_grad_(True)
    ref = torch.einsum("oc,ncs->nos", w, a)
    y = torch.full(ref.shape, float("nan"), device=DEV)
    NP = L.msl_pwconv_fwd_num_partials(N, Cin, Cout, S)
    part = torch.zeros(2 * Cout * NP, dtype=torch.float64, device=DEV)
    _lib.call("msl_pwconv_fwd", ptr(K(z)), ptr(K(sc)), ptr(K(sh)), ptr(K(w.detach())), ptr(y), ptr(part),
              N, Cin, Cout, S, st())
    close(y, ref, 1e-5, 1e-5, "pw fwd")
    s, q = stats_from_partials(part, Cout, NP)
    close(s, ref.double().sum((0, 2)), 1e-5, 1e-3, "pw sum")
    close(q, (ref.double() ** 2).sum((0, 2)), 1e-5, 1e-3, "pw sumsq")
    dy = rnd(*ref.shape, seed=14)
    ref.backward(dy)
    g = torch.full(z.shape, float("nan"), device=DEV)
    _lib.call("msl_pwconv_bwd_data", ptr(K(dy)), ptr(K(w.detach())), ptr(g), N, Cin, Cout, S, st())
    close(g, a.grad, 1e-5, 1e-5, "pw bwd data")
    ws = torch.empty(max(L.msl_pwconv_bwd_weight_workspace_bytes(N, Cin, Cout, S) // 4, 1), device=DEV)
    dw = torch.full((Cout, Cin), float("nan"), device=DEV)
    _lib.call("msl_pwconv_bwd_weight", ptr(K(dy)), ptr(K(z)), ptr(K(sc)), ptr(K(sh)), ptr(dw), ptr(ws),
              N, Cin, Cout, S, st())
    atol = max(1e-4, 5e-6 * float(w.grad.abs().max()))  # long sums (up to 65 536 products) cancel: error scales with max |dW|
    close(dw, w.grad, 1e-4, atol, "pw bwd weight")
    if Cin % 32 == 0 and Cout % 32 == 0 and (Cout % 64 == 0 or S % 32 == 0):
        # the slab form the training step uses + the batched reduction (kind 0); bit-identical to the stand-alone entry
        ns = L.msl_pwconv_bwd_weight_nslabs(N, Cin, Cout, S)
        slabs = torch.full((ns, Cout, Cin), float("nan"), device=DEV)
        _lib.call("msl_pwconv_bwd_weight_slabs", ptr(K(dy)), ptr(K(z)), ptr(K(sc)), ptr(K(sh)), ptr(slabs), N, Cin, Cout, S, st())
        if ns == 1:
            assert torch.equal(slabs[0], dw)
        else:
            out = torch.full((Cout, Cin), float("nan"), device=DEV)
            grad_reduce([(0, slabs, out, None, ns, Cout * Cin, Cout * Cin, 0, 0, 0)])
            close(out, dw, 1e-5, atol, "batched slab reduction vs the stand-alone one (different, fixed, summation orders)")
            out2 = torch.full((Cout, Cin), float("nan"), device=DEV)
            grad_reduce([(0, slabs, out2, None, ns, Cout * Cin, Cout * Cin, 0, 0, 0)])
            assert torch.equal(out, out2), "the reduction must be run-to-run bit-identical"
        dw2 = torch.full((Cout, Cin), float("nan"), device=DEV)  # without the input affine
        _lib.call("msl_pwconv_bwd_weight", ptr(K(dy)), ptr(K(z)), None, None, ptr(dw2), ptr(ws), N, Cin, Cout, S, st())
        close(dw2, torch.einsum("nos,ncs->oc", dy.double(), z.double()).float(), 1e-4, atol, "pw bwd weight, no affine")


def test_pw_bww_batch_is_the_single_launches_bit_for_bit():
    """msl_pwconv_bwd_weight_slabs_batch (the tail blocks' pointwise weight gradients in one launch) writes exactly the
    slabs of the per-layer launches: the 128^3 x 4 shapes of blocks 7..4 (two with a position split, two without)."""
    import ctypes
    L = _lib.load()
    N = 4
    shapes = [(512, 512, 64), (256, 512, 64), (256, 256, 512), (128, 256, 512)]  # (Cin, Cout, S)
    rows, single = [], []
    for q, (Cin, Cout, S) in enumerate(shapes):
        assert L.msl_pwconv_bwd_weight_batchable(N, Cin, Cout, S) == 1
        dy, z = K(rnd(N, Cout, S, seed=40 + q)), K(rnd(N, Cin, S, seed=50 + q))
        sc, sh = K(rnd(Cin, seed=60 + q).abs() + 0.5), K(rnd(Cin, seed=70 + q, scale=0.3))
        ns = L.msl_pwconv_bwd_weight_nslabs(N, Cin, Cout, S)
        ref = torch.full((ns, Cout, Cin), float("nan"), device=DEV)
        _lib.call("msl_pwconv_bwd_weight_slabs", ptr(dy), ptr(z), ptr(sc), ptr(sh), ptr(ref), N, Cin, Cout, S, st())
        out = torch.full((ns, Cout, Cin), float("nan"), device=DEV)
        rows.append((dy, z, sc, sh, out, Cin, Cout, S))
        single.append(ref)
    for n in (4, 2):  # the whole group, and a shorter one
        for r in rows:
            r[4].fill_(float("nan"))
        P, I = ctypes.c_void_p * n, ctypes.c_int * n
        arrs = [P(*[ptr(r[c]) for r in rows[:n]]) for c in range(5)] + [I(*[r[c] for r in rows[:n]]) for c in range(5, 8)]
        _lib.call("msl_pwconv_bwd_weight_slabs_batch", *[ctypes.addressof(a) for a in arrs], n, N, st())
        torch.cuda.synchronize()
        for q in range(n):
            assert torch.equal(rows[q][4], single[q]), f"layer {q} of a batch of {n}"
    assert L.msl_pwconv_bwd_weight_batchable(N, 32, 96, 96) == 0  # 32-row tiles: not in the batch form
    bad = (ctypes.c_int * 1)(96)
    one = [(ctypes.c_void_p * 1)(ptr(rows[0][c])) for c in range(5)]
    rc = L.msl_pwconv_bwd_weight_slabs_batch(*[ctypes.addressof(a) for a in one], ctypes.addressof((ctypes.c_int * 1)(32)),
                                             ctypes.addressof(bad), ctypes.addressof((ctypes.c_int * 1)(96)), 1, 1, st())
    assert rc != 0



def grad_reduce(rows):
    """rows of (kind, src, dst, dst2, nslabs, count, stride, p0, p1, p2) -> one msl_grad_reduce_batch launch."""
    import ctypes
    L = _lib.load()
    esz = L.msl_grad_reduce_entry_bytes()
    host = (ctypes.c_ubyte * (esz * len(rows)))()
    first = 0
    for k, (kind, src, dst, dst2, ns, cnt, stride, p0, p1, p2) in enumerate(rows):
        nb = L.msl_grad_reduce_table_set(ctypes.addressof(host), k, first, kind, ptr(src), ptr(dst), ptr(dst2), ns, cnt, stride,
                                         p0, p1, p2)
        assert nb in ((cnt + 31) // 32, (cnt + 1023) // 1024)
        first += nb
    table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(DEV)
    _lib.call("msl_grad_reduce_batch", ptr(table), len(rows), first, st())
    torch.cuda.synchronize()


def test_grad_reduce_batch_all_kinds_in_one_launch():
    """The batched gradient reduction against plain sums: fp32 slabs, fp64 partials, the padded stem image and the head
    slab layout, several entries (ragged counts, 1..70 slabs) in ONE launch; canaries around every destination."""
    g = torch.Generator().manual_seed(5)
    rows, checks = [], []

    def dst(n):
        t = torch.full((n + 64,), float("nan"), device=DEV)
        return t, t[32:32 + n]

    for ns, cnt in ((1, 5), (7, 1000), (70, 333), (64, 2048), (3, 4096), (16, 1028), (9, 12)):  # kind 0
        pad = 4 if cnt % 8 == 0 else 3  # stride > count; multiples of 4 take the few-slab vector form when ns <= 16
        src = torch.randn((ns, cnt + pad), generator=g).to(DEV)
        full, view = dst(cnt)
        rows.append((0, src, view, None, ns, cnt, cnt + pad, 0, 0, 0))
        checks.append((full, view, src[:, :cnt].double().sum(0).float(), 1e-5))
    for NP, cnt in ((1, 27), (33, 864), (300, 100)):  # kind 1: [count][NP] fp64
        src = torch.randn((cnt, NP), generator=g, dtype=torch.float64).to(DEV)
        full, view = dst(cnt)
        rows.append((1, src, view, None, NP, cnt, 0, 0, 0, 0))
        checks.append((full, view, src.sum(1).float(), 1e-7))
    for cin in (1, 2):  # kind 2: stem image [32][32*NT] -> [32][K]
        Kk, NT = cin * 27, (cin * 27 + 31) // 32
        src = torch.randn((9, 32, 32 * NT), generator=g).to(DEV)
        full, view = dst(32 * Kk)
        rows.append((2, src, view, None, 9, 1024 * NT, 1024 * NT, Kk, NT, 0))
        checks.append((full, view, src.double().sum(0)[:, :Kk].reshape(-1).float(), 1e-5))
    C, ncls, ns = 32, 2, 5  # kind 3: head slabs [ns][C/16][27*MT][16 co][16 ci] -> loc (12,C,27) / cls (4,C,27)
    src = torch.randn((ns, C // 16, 27, 16, 16), generator=g).to(DEV)
    fl, vl = dst(12 * C * 27)
    fc, vc = dst(2 * ncls * C * 27)
    rows.append((3, src, vl, vc, ns, (C // 16) * 27 * 256, (C // 16) * 27 * 256, C, 1, 12 + 2 * ncls))
    tot = src.double().sum(0)  # [ct][tap][co][cil]
    w = tot.permute(2, 0, 3, 1).reshape(16, C, 27).float()  # [co][ci = ct*16 + cil][tap]
    checks.append((fl, vl, w[:12].reshape(-1), 1e-5))
    checks.append((fc, vc, w[12:16].reshape(-1), 1e-5))
    grad_reduce(rows)
    for full, view, ref, tol in checks:
        close(view, ref, tol, tol, "grad reduce")
        assert torch.isnan(full[:32]).all() and torch.isnan(full[32 + view.numel():]).all(), "wrote outside its range"


# ------------------------------------------------------------------------------------------------- batch norm
def test_bn_forward_backward():
    L = _lib.load()
    N, C, dims = 3, 8, (4, 6, 8)
    S = dims[0] * dims[1] * dims[2]
    y = (rnd(N, C, *dims, seed=20) * 2 + 1).requires_grad_(True)
    gamma, beta = (rnd(C, seed=21).abs() + 0.5).requires_grad_(True), rnd(C, seed=22, scale=0.2).requires_grad_(True)
    rm, rv = rnd(C, seed=23, scale=0.1), rnd(C, seed=24).abs() + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    a = torch.relu(F.batch_norm(y, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5))
    g = rnd(*a.shape, seed=25)
    a.backward(g)
    # forward: partials from the generic fallback path of the depthwise kernel are not needed — build them here
    yd = y.detach().double()
    part = torch.stack([yd.sum((0, 2, 3, 4)), (yd ** 2).sum((0, 2, 3, 4))]).view(2, C, 1).contiguous().to(DEV)
    vec = torch.zeros((6, C), device=DEV)
    rm_d, rv_d = rm.to(DEV), rv.to(DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    _lib.call("msl_bn_finalize", ptr(part), 1, float(N * S), ptr(K(gamma.detach())), ptr(K(beta.detach())),
              ptr(rm_d), ptr(rv_d), ptr(nbt), 0.1, 1e-5, ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]), C, st())
    close(rm_d, rm_ref, 1e-6, 1e-7, "running_mean")
    close(rv_d, rv_ref, 1e-6, 1e-7, "running_var")
    assert int(nbt) == 1
    out = torch.empty(y.shape, device=DEV)
    pad = torch.zeros((N, C) + tuple(d + 2 for d in dims), device=DEV)
    _lib.call("msl_bn_relu_materialize", ptr(K(y.detach())), ptr(vec[0]), ptr(vec[1]), ptr(out), ptr(pad), N, C, *dims, st())
    close(out, a, 1e-5, 1e-5, "bn+relu")
    close(pad[:, :, 1:-1, 1:-1, 1:-1], a, 1e-5, 1e-5, "bn+relu padded")
    halo = pad.clone()
    halo[:, :, 1:-1, 1:-1, 1:-1] = 0
    assert float(halo.abs().max()) == 0.0
    # backward
    NP = L.msl_bn_relu_bwd_num_partials(N, S)
    bp = torch.zeros(2 * C * NP, dtype=torch.float64, device=DEV)
    gd, yd32 = g.to(DEV), y.detach().to(DEV)
    _lib.call("msl_bn_relu_bwd_reduce", ptr(gd), ptr(yd32), ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]), ptr(bp), N, C, S, st())
    dgam, dbet = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    _lib.call("msl_bn_bwd_finalize", ptr(bp), NP, float(N * S), ptr(dgam), ptr(dbet), ptr(vec[4]), ptr(vec[5]), C, st())
    _lib.call("msl_bn_relu_bwd_apply", ptr(gd), ptr(yd32), ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]), ptr(vec[4]),
              ptr(vec[5]), ptr(gd), N, C, S, st())
    close(dgam, gamma.grad, 1e-4, 1e-5, "dgamma")
    close(dbet, beta.grad, 1e-4, 1e-5, "dbeta")
    close(gd, y.grad, 1e-4, 1e-5, "bn bwd dy")
    # single-launch variant
    g2 = K(g)
    dg2, db2 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    _lib.call("msl_bn_relu_bwd_fused", ptr(g2), ptr(yd32), ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]), ptr(dg2), ptr(db2),
              ptr(g2), N, C, S, st())
    close(dg2, gamma.grad, 1e-4, 1e-5, "fused dgamma")
    close(db2, beta.grad, 1e-4, 1e-5, "fused dbeta")
    close(g2, y.grad, 1e-4, 1e-5, "fused bn bwd dy")
    # eval affine
    _lib.call("msl_bn_eval_affine", ptr(K(gamma.detach())), ptr(K(beta.detach())), ptr(rm_d), ptr(rv_d), 1e-5,
              ptr(vec[0]), ptr(vec[1]), C, st())
    ref_eval = F.batch_norm(y.detach(), rm_ref, rv_ref, gamma.detach(), beta.detach(), False, 0.1, 1e-5)
    close(y.detach() * vec[0].cpu().view(1, -1, 1, 1, 1) + vec[1].cpu().view(1, -1, 1, 1, 1), ref_eval, 1e-5, 1e-5, "eval affine")


# ------------------------------------------------------------------------------------------------- heads
@pytest.mark.parametrize("N,C,dims,ncls", [(2, 128, (8, 8, 8), 2), (2, 32, (3, 5, 6), 2), (1, 256, (4, 4, 4), 2),
                                           (2, 64, (2, 2, 2), 2), (1, 32, (4, 4, 4), 3),
                                           # the LDS-staged kernels: W = H in {16, 8, 4}, with and without a channel
                                           # split of the forward pass, several tiles per workgroup in bwd-data, odd
                                           # block counts per workgroup in the weight gradient, D != W
                                           (2, 128, (16, 16, 16), 2), (4, 256, (8, 8, 8), 2), (4, 512, (4, 4, 4), 2),
                                           (1, 16, (16, 16, 16), 2), (3, 48, (8, 8, 8), 2), (1, 32, (8, 4, 4), 2),
                                           (5, 16, (12, 8, 8), 2), (1, 32, (16, 16, 16), 3),
                                           # the 192^3 inference maps: the forward kernel tiles them with 8 x 8 / 4 x 4 x 4 blocks
                                           # (bwd: register-fed kernels)
                                           (2, 128, (24, 24, 24), 2), (2, 256, (12, 12, 12), 2), (1, 16, (12, 12, 12), 2),
                                           (1, 32, (5, 24, 24), 2),
                                           # tiled forward on non-cubic maps: 16-wide blocks (W % 16, H % 4), 8 x 8 blocks, 4^3 blocks
                                           (1, 32, (3, 8, 32), 2), (2, 16, (2, 16, 24), 2), (1, 48, (4, 8, 12), 2),
                                           (1, 32, (3, 12, 48), 2), (1, 16, (8, 12, 4), 2)])
def test_heads_fwd_bwd(N, C, dims, ncls):
    L = _lib.load()
    a = torch.relu(rnd(N, C, *dims, seed=30)).requires_grad_(True)
    lw = (rnd(12, C, 3, 3, 3, seed=31) / (27 * C) ** 0.5).requires_grad_(True)
    cw = (rnd(2 * ncls, C, 3, 3, 3, seed=32) / (27 * C) ** 0.5).requires_grad_(True)
    lb, cb = rnd(12, seed=33, scale=0.1).requires_grad_(True), rnd(2 * ncls, seed=34, scale=0.1).requires_grad_(True)
    S = dims[0] * dims[1] * dims[2]
    rl = F.conv3d(a, lw, lb, padding=1).permute(0, 2, 3, 4, 1).reshape(N, -1, 6)
    rc = F.conv3d(a, cw, cb, padding=1).permute(0, 2, 3, 4, 1).reshape(N, -1, ncls)
    off, Ptot = 10, 2 * S + 14
    pad = torch.zeros((N, C) + tuple(d + 2 for d in dims), device=DEV)
    pad[:, :, 1:-1, 1:-1, 1:-1] = a.detach().to(DEV)
    ne = L.msl_head_packed_weight_elems(C, ncls)
    Wf, Wb = torch.empty(ne, device=DEV), torch.empty(ne, device=DEV)
    lwd, cwd, lbd, cbd = (t.detach().to(DEV) for t in (lw, cw, lb, cb))
    _lib.call("msl_head_pack_weights", ptr(lwd), ptr(cwd), ptr(Wf), ptr(Wb), C, ncls, st())
    ws = torch.empty(max(L.msl_head_fwd_workspace_bytes(N, C, *dims, ncls), L.msl_head_bwd_weight_workspace_bytes(N, C, *dims, ncls)) // 4 + 1, device=DEV)
    locs = torch.full((N, Ptot, 6), 7.0, device=DEV)
    scores = torch.full((N, Ptot, ncls), 7.0, device=DEV)
    _lib.call("msl_head_conv_fwd", ptr(pad), ptr(Wf), ptr(lbd), ptr(cbd), ptr(locs), ptr(scores), ptr(ws), N, C, *dims, Ptot,
              off, ncls, st())
    close(locs[:, off:off + 2 * S], rl, 1e-4, 1e-5, "head locs")
    close(scores[:, off:off + 2 * S], rc, 1e-4, 1e-5, "head scores")
    assert bool((locs[:, :off] == 7).all()) and bool((locs[:, off + 2 * S:] == 7).all())
    # backward
    dl, dc = rnd(*rl.shape, seed=35), rnd(*rc.shape, seed=36)
    (rl * dl).sum().backward(retain_graph=True)
    (rc * dc).sum().backward()
    dlf = torch.zeros((N, Ptot, 6), device=DEV)
    dcf = torch.zeros((N, Ptot, ncls), device=DEV)
    dlf[:, off:off + 2 * S] = dl.to(DEV)
    dcf[:, off:off + 2 * S] = dc.to(DEV)
    mt16 = 16 * ((12 + 2 * ncls + 15) // 16)
    dO = torch.zeros((N, mt16) + tuple(d + 2 for d in dims), device=DEV)
    _lib.call("msl_head_grad_pack", ptr(dlf), ptr(dcf), ptr(dO), N, *dims, Ptot, off, ncls, st())
    ga = torch.full(a.shape, float("nan"), device=DEV)
    _lib.call("msl_head_conv_bwd_data", ptr(dO), ptr(Wb), ptr(ga), N, C, *dims, ncls, st())
    close(ga, a.grad, 1e-4, 1e-5, "head bwd data")
    glw, gcw = torch.full(lw.shape, float("nan"), device=DEV), torch.full(cw.shape, float("nan"), device=DEV)
    glb, gcb = torch.empty(12, device=DEV), torch.empty(2 * ncls, device=DEV)
    _lib.call("msl_head_conv_bwd_weight", ptr(dO), ptr(pad), ptr(glw), ptr(gcw), ptr(glb), ptr(gcb), ptr(ws), N, C, *dims, ncls, st())
    # sums over N * S products (27 648 at 24^3 x 2): the absolute error scales with the largest gradient element
    atol_w = max(1e-4, 2e-6 * float(max(lw.grad.abs().max(), cw.grad.abs().max())))
    close(glw, lw.grad, 1e-4, atol_w, "head dW loc")
    close(gcw, cw.grad, 1e-4, atol_w, "head dW cls")
    close(glb, lb.grad, 1e-4, 1e-4, "head db loc")
    close(gcb, cb.grad, 1e-4, 1e-4, "head db cls")


@pytest.mark.parametrize("ncls", [2, 3])
def test_head_grad_pack_batch_is_the_single_launches_bit_for_bit(ncls):
    """msl_head_grad_pack_batch (every scale's head gradient image in one launch) == one msl_head_grad_pack per scale."""
    import ctypes
    L = _lib.load()
    N = 3
    dims = [(16, 16, 16), (8, 8, 8), (4, 4, 4), (5, 6, 7)]
    offs, Ptot = [], 3
    for d in dims:
        offs.append(Ptot)
        Ptot += 2 * d[0] * d[1] * d[2]
    Ptot += 5
    dl, dc = K(rnd(N, Ptot, 6, seed=80)), K(rnd(N, Ptot, ncls, seed=81))
    mt16 = 16 * ((12 + 2 * ncls + 15) // 16)
    ref = [torch.zeros((N, mt16) + tuple(x + 2 for x in d), device=DEV) for d in dims]
    for r, d, o in zip(ref, dims, offs):
        _lib.call("msl_head_grad_pack", ptr(dl), ptr(dc), ptr(r), N, *d, Ptot, o, ncls, st())
    for n in (4, 3, 1):
        out = [torch.zeros_like(r) for r in ref[:n]]
        I = ctypes.c_int * n
        arrs = [(ctypes.c_void_p * n)(*[ptr(o) for o in out])] + [I(*[d[a] for d in dims[:n]]) for a in range(3)] + [I(*offs[:n])]
        _lib.call("msl_head_grad_pack_batch", ptr(dl), ptr(dc), *[ctypes.addressof(a) for a in arrs], n, N, Ptot, ncls, st())
        torch.cuda.synchronize()
        for q in range(n):
            assert torch.equal(out[q], ref[q]), f"scale {q} of {n}"
    assert L.msl_head_grad_pack_batch(ptr(dl), ptr(dc), ctypes.addressof(arrs[0]), ctypes.addressof(arrs[1]), ctypes.addressof(arrs[2]),
                                      ctypes.addressof(arrs[3]), ctypes.addressof(arrs[4]), 5, N, Ptot, ncls, st()) != 0


# ------------------------------------------------------------------------------------------------- adam
def test_adam_matches_torch():
    n = 10007
    p0, g1, g2 = rnd(n, seed=40), rnd(n, seed=41), rnd(n, seed=42)
    isb = (torch.arange(n) % 7 == 0)
    pa, pb = p0[isb].clone().requires_grad_(True), p0[~isb].clone().requires_grad_(True)
    opt = torch.optim.Adam([{"params": [pa], "lr": 2e-3}, {"params": [pb]}], lr=1e-3, weight_decay=0.0005)
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    import math
    for t, g in enumerate((g1, g2), start=1):
        pa.grad, pb.grad = g[isb].clone(), g[~isb].clone()
        opt.step()
        bc1, bc2 = 1 - 0.9 ** t, 1 - 0.999 ** t
        hp = torch.tensor([2e-3 / bc1, 1e-3 / bc1, math.sqrt(bc2), 0.9, 0.999, 1e-8, 0.0005, 1.0], device=DEV)
        _lib.call("msl_adam_step", ptr(p), ptr(K(g)), ptr(m), ptr(v), ptr(hp), ptr(K(isb.to(torch.uint8))), n, st())
    ref = p0.clone()
    ref[isb], ref[~isb] = pa.detach(), pb.detach()
    close(p, ref, 1e-6, 1e-7, "adam params after 2 steps")


def test_nan_flag():
    x = torch.zeros(100000, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.call("msl_nan_flag", ptr(x), x.numel(), ptr(flag), 1, st())
    assert int(flag) == 0
    x[77777] = float("nan")
    _lib.call("msl_nan_flag", ptr(x), x.numel(), ptr(flag), 2, st())
    assert int(flag) == 2


# ------------------------------------------------------------------------------------------------- random shapes
def _rng_shapes(seed, n, gen):
    import random
    r = random.Random(seed)
    return [gen(r) for _ in range(n)]


@pytest.mark.parametrize("case", _rng_shapes(101, 8, lambda r: (r.choice([1, 2, 3]), r.choice([32, 64, 128, 256, 512]),
                                                                r.choice([32, 64, 128, 256]), r.randint(1, 300))),
                         ids=lambda c: "N%d_K%d_M%d_S%d" % c)
def test_pw_random_shapes(case):
    """Wave-autonomous pointwise kernels at seeded random (N, Cin, Cout, S): ragged column counts, every K-split width."""
    N, Cin, Cout, S = case
    L = _lib.load()
    z = rnd(N, Cin, S, seed=20)
    w = rnd(Cout, Cin, seed=21) / Cin ** 0.5
    sc, sh = rnd(Cin, seed=22).abs() + 0.5, rnd(Cin, seed=23, scale=0.3)
    a = torch.relu(z * sc.view(1, -1, 1) + sh.view(1, -1, 1))
    ref = torch.einsum("oc,ncs->nos", w, a)
    y = torch.full(ref.shape, float("nan"), device=DEV)
    NP = L.msl_pwconv_fwd_num_partials(N, Cin, Cout, S)
    part = torch.full((2 * Cout * NP,), float("nan"), dtype=torch.float64, device=DEV)
    _lib.call("msl_pwconv_fwd", ptr(K(z)), ptr(K(sc)), ptr(K(sh)), ptr(K(w)), ptr(y), ptr(part), N, Cin, Cout, S, st())
    close(y, ref, 1e-5, 1e-5, "pw fwd")
    s, q = stats_from_partials(part, Cout, NP)
    close(s, ref.double().sum((0, 2)), 1e-5, 1e-3, "pw sum")
    close(q, (ref.double() ** 2).sum((0, 2)), 1e-5, 1e-3, "pw sumsq")
    dy = rnd(*ref.shape, seed=24)
    g = torch.full(z.shape, float("nan"), device=DEV)
    _lib.call("msl_pwconv_bwd_data", ptr(K(dy)), ptr(K(w)), ptr(g), N, Cin, Cout, S, st())
    close(g, torch.einsum("oc,nos->ncs", w, dy), 1e-5, 1e-5, "pw bwd data")


@pytest.mark.parametrize("case", _rng_shapes(202, 8, lambda r: (r.choice([1, 1, 2, 3]),
                                                                (r.randint(3, 20), r.randint(3, 24), r.randint(3, 150)),
                                                                r.choice([(2, 2, 2), (1, 2, 2)]))),
                         ids=lambda c: "cin%d_%dx%dx%d_s%d" % (c[0], *c[1], c[2][0]))
def test_stem_fwd_random_shapes(case):
    """MFMA stem forward at seeded random volumes: odd sizes, rows shorter and longer than one 64-column chunk, idle
    waves (fewer chunks than wave slots), both stride patterns."""
    cin, dims, stride = case
    L = _lib.load()
    N = 2
    x, w = rnd(N, cin, *dims, seed=30), rnd(32, cin, 3, 3, 3, seed=31, scale=0.3)
    ref = F.conv3d(x, w, stride=stride, padding=1)
    y = torch.full(ref.shape, float("nan"), device=DEV)
    od, oh, ow = ref.shape[2:]
    NP = L.msl_stem_conv_fwd_num_partials(N, od, oh, ow)
    part = torch.full((2 * 32 * NP,), float("nan"), dtype=torch.float64, device=DEV)
    _lib.call("msl_stem_conv_fwd", ptr(K(x)), ptr(K(w)), ptr(y), ptr(part), N, cin, *dims, *stride, st())
    close(y, ref, 1e-5, 1e-5, "stem fwd")
    s, q = stats_from_partials(part, 32, NP)
    close(s, ref.double().sum((0, 2, 3, 4)), 1e-5, 1e-3, "stem sum")
    close(q, (ref.double() ** 2).sum((0, 2, 3, 4)), 1e-5, 1e-3, "stem sumsq")


@pytest.mark.parametrize("case", _rng_shapes(303, 6, lambda r: (r.choice([1, 1, 2]),
                                                                (2 * r.randint(2, 10), 2 * r.randint(2, 10), 8 * r.randint(1, 20)))),
                         ids=lambda c: "cin%d_%dx%dx%d" % (c[0], *c[1]))
def test_fused_stem_backward_random_shapes(case):
    """Both halves of the stem backward that never stores dL/d(stem activation), at seeded random even volumes (the fused
    pass needs the stem's output rows to be a multiple of 4 wide; other shapes take the materialising path)."""
    _fused_stem_case(*case)


# ------------------------------------------------------------------------------------------------- per-channel backward link
@pytest.mark.parametrize("n,c,dims,stride,acc", [
    (4, 512, (4, 4, 4), 1, 1),     # block 7 at 128^3 x 4: one wave per channel, one quad per lane
    (4, 256, (8, 8, 8), 2, 0),     # block 6: stride 2, four waves, 2 input quads per thread
    (4, 256, (8, 8, 8), 1, 1),     # block 5 (+ the heads' share of the 8^3 feature map)
    (4, 128, (16, 16, 16), 2, 1),  # block 4: sixteen waves per channel, 4 input quads per thread
    (4, 128, (16, 16, 16), 1, 0),  # block 3: sixteen waves, 124 KB of LDS
    (1, 40, (4, 4, 4), 1, 0),      # ragged: fewer quads than lanes
    (3, 24, (4, 8, 16), 1, 1),     # odd batch, non-cubic: masked quads
    (2, 16, (8, 4, 8), 2, 1),      # stride 2 non-cubic
    (5, 8, (8, 8, 16), 2, 0),      # five images: the last thread's quads are masked
    (2, 8, (16, 8, 16), 1, 1),     # eight waves
    (3, 8, (2, 4, 8), 2, 0),       # OD = 1
])
def test_block_bwd_channel_link_matches_autograd(n, c, dims, stride, acc):
    """msl_block_bwd_channel_link against torch autograd of relu(bn1(dwconv(relu(bn2(y))))) (mobilenet.py:43-47 backward,
    train-mode BatchNorm) + the heads' share of dL/d relu(bn2(y)): dL/dz, dL/dy, dgamma / dbeta of both BatchNorms."""
    L = _lib.load()
    D, H, W = dims
    assert L.msl_block_bwd_channel_link_supported(n, D, H, W, stride) in (1, 4, 8, 16)
    y = rnd(n, c, D, H, W, seed=1).requires_grad_(True)
    bn2, bn1 = torch.nn.BatchNorm3d(c), torch.nn.BatchNorm3d(c)
    conv = torch.nn.Conv3d(c, c, 3, stride=stride, padding=1, groups=c, bias=False)
    with torch.no_grad():
        for i, bn in enumerate((bn2, bn1)):
            bn.weight.copy_(torch.rand(c, generator=torch.Generator().manual_seed(10 + i)) + 0.5)
            bn.bias.copy_(rnd(c, seed=20 + i, scale=0.3))
        conv.weight.copy_(rnd(c, 1, 3, 3, 3, seed=3, scale=0.3))
    a_prev = torch.relu(bn2(y))
    z = conv(a_prev)
    z.retain_grad()
    a = torch.relu(bn1(z))
    G = rnd(*a.shape, seed=4)
    Hs = rnd(*a_prev.shape, seed=5) if acc else torch.zeros_like(a_prev)
    ((a * G).sum() + (a_prev * Hs).sum()).backward()

    def vec(x, bn):  # [scale, shift, mean, invstd] of a train-mode BatchNorm on x
        mean = x.mean(dim=(0, 2, 3, 4))
        var = x.var(dim=(0, 2, 3, 4), unbiased=False)
        inv = 1.0 / torch.sqrt(var + bn.eps)
        sc = bn.weight * inv
        return torch.stack([sc, bn.bias - mean * sc, mean, inv]).detach()
    g_z, g_y = K(G), K(Hs if acc else torch.full_like(Hs, 7.0))  # (without accumulate the old content must be ignored)
    outs = [K(torch.zeros(c)) for _ in range(4)]
    dw = K(torch.full((c, 27), 5.0))
    _lib.call("msl_block_bwd_channel_link", ptr(g_z), ptr(K(z)), ptr(K(vec(z, bn1))), ptr(K(conv.weight)), ptr(K(y)),
              ptr(K(vec(y, bn2))), ptr(g_y), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), ptr(outs[3]), ptr(dw), n, c, D, H, W, stride,
              acc, st())
    close(dw, conv.weight.grad.view(c, 27), 2e-4, 2e-5 * float(conv.weight.grad.abs().max()), "depthwise weight gradient")
    close(g_z, z.grad, 2e-4, 2e-5 * float(z.grad.abs().max()), "dL/dz")
    close(g_y, y.grad, 2e-4, 2e-5 * float(y.grad.abs().max()), "dL/dy")
    close(outs[0], bn1.weight.grad, 2e-4, 1e-5 * float(bn1.weight.grad.abs().max()), "dgamma bn1")
    close(outs[1], bn1.bias.grad, 2e-4, 1e-5 * float(bn1.bias.grad.abs().max()), "dbeta bn1")
    close(outs[2], bn2.weight.grad, 2e-4, 1e-5 * float(bn2.weight.grad.abs().max()), "dgamma bn2")
    close(outs[3], bn2.bias.grad, 2e-4, 1e-5 * float(bn2.bias.grad.abs().max()), "dbeta bn2")
    # run-to-run bit-identical (fixed summation order, no atomics)
    # ... and the form without the weight gradient gives the same activation gradients
    g_z2, g_y2 = K(G), K(Hs if acc else torch.full_like(Hs, -3.0))
    _lib.call("msl_block_bwd_channel_link", ptr(g_z2), ptr(K(z)), ptr(K(vec(z, bn1))), ptr(K(conv.weight)), ptr(K(y)),
              ptr(K(vec(y, bn2))), ptr(g_y2), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), ptr(outs[3]), None, n, c, D, H, W, stride,
              acc, st())
    # (another template instantiation: the compiler may contract the BatchNorm arithmetic differently - last-bit agreement only)
    assert torch.equal(g_z, g_z2)
    close(g_y2, g_y, 1e-5, 1e-6 * float(g_y.abs().max()), "dL/dy without the weight gradient")
    dw2 = K(torch.zeros((c, 27)))
    g_z3, g_y3 = K(G), K(Hs if acc else torch.full_like(Hs, 1.0))
    _lib.call("msl_block_bwd_channel_link", ptr(g_z3), ptr(K(z)), ptr(K(vec(z, bn1))), ptr(K(conv.weight)), ptr(K(y)),
              ptr(K(vec(y, bn2))), ptr(g_y3), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), ptr(outs[3]), ptr(dw2), n, c, D, H, W, stride,
              acc, st())
    assert torch.equal(dw, dw2) and torch.equal(g_y, g_y3)


def test_block_bwd_channel_link_refuses_what_it_cannot_hold():
    L = _lib.load()
    assert L.msl_block_bwd_channel_link_supported(4, 32, 32, 32, 2) == 0   # 32 768 elements per channel
    assert L.msl_block_bwd_channel_link_supported(4, 6, 6, 6, 1) == 0      # not a power of two (a 192^3 tail)
    assert L.msl_block_bwd_channel_link_supported(4, 12, 12, 12, 1) == 0
    assert L.msl_block_bwd_channel_link_supported(2, 4, 4, 4, 2) == 0      # OW = 2
    x = K(torch.zeros(2 * 8 * 216))
    rc = L.msl_block_bwd_channel_link(ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), None, 2, 8, 6, 6,
                                      6, 1, 0, st())
    assert rc == -2


# ------------------------------------------------------------------------------------------------- fused pointwise backward
@pytest.mark.parametrize("n,s_dims", [(2, (32, 32, 32)), (4, (16, 32, 32))])
def test_pwconv_bwd_fused_matches_autograd(n, s_dims):
    """msl_pwconv_bwd_fused (block 1's whole pointwise backward in one pass) against torch autograd of
    relu(bn2(conv1x1(relu(bn1(z))))) with train-mode BatchNorms (mobilenet.py:45-47): dL/d relu(bn1(z)), the BatchNorm1-backward
    sums of z, dgamma / dbeta of bn2 and the 1x1x1 weight gradient; the BatchNorm2-backward sums come in as partials."""
    L = _lib.load()
    cin, cout = 32, 64
    S = s_dims[0] * s_dims[1] * s_dims[2]
    NP = L.msl_pwconv_bwd_fused_num_partials(n, cin, cout, S)
    assert NP == 256
    assert L.msl_pwconv_bwd_fused_num_partials(n, 64, 128, S) == 0 and L.msl_pwconv_bwd_fused_num_partials(1, cin, cout, 4096) == 0
    z = rnd(n, cin, *s_dims, seed=1).requires_grad_(True)
    bn1, bn2 = torch.nn.BatchNorm3d(cin), torch.nn.BatchNorm3d(cout)
    conv = torch.nn.Conv3d(cin, cout, 1, bias=False)
    with torch.no_grad():
        for i, bn in enumerate((bn1, bn2)):
            bn.weight.copy_(torch.rand(bn.num_features, generator=torch.Generator().manual_seed(10 + i)) + 0.5)
            bn.bias.copy_(rnd(bn.num_features, seed=20 + i, scale=0.3))
        conv.weight.copy_(rnd(cout, cin, 1, 1, 1, seed=3, scale=0.2))
    a1 = torch.relu(bn1(z))
    a1.retain_grad()
    y = conv(a1)
    y.retain_grad()
    out = torch.relu(bn2(y))
    G = rnd(*out.shape, seed=4)
    (out * G).sum().backward()

    def vec(x, bn):
        mean, var = x.mean(dim=(0, 2, 3, 4)), x.var(dim=(0, 2, 3, 4), unbiased=False)
        inv = 1.0 / torch.sqrt(var + bn.eps)
        sc = bn.weight * inv
        return torch.stack([sc, bn.bias - mean * sc, mean, inv]).detach()
    vy, vz = vec(y, bn2), vec(z, bn1)
    # the BatchNorm2-backward sums as a producer would leave them: [2][cout][np] partials (here: split over 5 parts)
    yd = y.detach()
    gm = torch.where(yd * vy[0].view(1, -1, 1, 1, 1) + vy[1].view(1, -1, 1, 1, 1) > 0, G, torch.zeros_like(G)).double()
    xh = ((yd - vy[2].view(1, -1, 1, 1, 1)) * vy[3].view(1, -1, 1, 1, 1)).double()
    ynp = 5
    parts = torch.zeros(2, cout, ynp, dtype=torch.float64)
    flat_g, flat_x = gm.transpose(0, 1).reshape(cout, -1), (gm * xh).transpose(0, 1).reshape(cout, -1)
    for p_, (a, b) in enumerate(zip(flat_g.chunk(ynp, dim=1), flat_x.chunk(ynp, dim=1))):
        parts[0, :, p_], parts[1, :, p_] = a.sum(1), b.sum(1)
    gz = K(torch.zeros(n, cin, *s_dims))
    zpart = K(torch.zeros(2 * cin * NP, dtype=torch.float64))
    slabs = K(torch.zeros(NP * cout * cin))
    dgam, dbet = K(torch.zeros(cout)), K(torch.zeros(cout))
    _lib.call("msl_pwconv_bwd_fused", ptr(K(G)), ptr(K(y)), ptr(K(vy)), ptr(K(parts)), ynp, float(n * S), ptr(dgam), ptr(dbet),
              ptr(K(conv.weight)), ptr(K(z)), ptr(K(vz)), ptr(gz), ptr(zpart), ptr(slabs), n, cin, cout, S, st())
    close(gz, a1.grad, 2e-4, 2e-5 * float(a1.grad.abs().max()), "dL/d relu(bn1(z))")
    close(dgam, bn2.weight.grad, 2e-4, 1e-5 * float(bn2.weight.grad.abs().max()), "dgamma bn2")
    close(dbet, bn2.bias.grad, 2e-4, 1e-5 * float(bn2.bias.grad.abs().max()), "dbeta bn2")
    dw = slabs.view(NP, cout, cin).double().sum(0)
    close(dw, conv.weight.grad.view(cout, cin), 3e-4, 3e-5 * float(conv.weight.grad.abs().max()), "1x1x1 weight gradient")
    zs = zpart.view(2, cin, NP).sum(-1).cpu()
    close(zs[0], bn1.bias.grad, 3e-4, 1e-5 * float(bn1.bias.grad.abs().max()), "sum gm of bn1 (= dbeta)")
    close(zs[1], bn1.weight.grad, 3e-4, 1e-5 * float(bn1.weight.grad.abs().max()), "sum gm * xhat of bn1 (= dgamma)")
    # run-to-run bit-identical
    gz2, zpart2, slabs2 = K(torch.zeros_like(gz)), K(torch.zeros_like(zpart)), K(torch.zeros_like(slabs))
    _lib.call("msl_pwconv_bwd_fused", ptr(K(G)), ptr(K(y)), ptr(K(vy)), ptr(K(parts)), ynp, float(n * S), ptr(dgam), ptr(dbet),
              ptr(K(conv.weight)), ptr(K(z)), ptr(K(vz)), ptr(gz2), ptr(zpart2), ptr(slabs2), n, cin, cout, S, st())
    assert torch.equal(gz, gz2) and torch.equal(zpart, zpart2) and torch.equal(slabs, slabs2)


# ------------------------------------------------------------------------------------------------- dw bwd-data + BN sums + dW in one pass
@pytest.mark.parametrize("n,c,dims,acc", [(2, 16, (8, 8, 8), 0), (1, 8, (10, 14, 20), 1), (3, 4, (6, 6, 36), 0), (2, 64, (32, 32, 32), 0)])
def test_dw_s2_bwd_data_bnreduce_bww_matches_autograd(n, c, dims, acc):
    """msl_dwconv_s2_bwd_data_bnreduce_bww: input gradient of a stride-2 depthwise layer (+ the heads' share), the
    BatchNorm-backward sums of the producer layer and the depthwise weight gradient in ONE pass, against torch autograd of
    dwconv(relu(bn(y))) (mobilenet.py:44 backward; odd output extents included)."""
    L = _lib.load()
    D, H, W = dims
    y = rnd(n, c, D, H, W, seed=1).requires_grad_(True)
    bn = torch.nn.BatchNorm3d(c)
    conv = torch.nn.Conv3d(c, c, 3, stride=2, padding=1, groups=c, bias=False)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=torch.Generator().manual_seed(3)) + 0.5)
        bn.bias.copy_(rnd(c, seed=4, scale=0.3))
        conv.weight.copy_(rnd(c, 1, 3, 3, 3, seed=5, scale=0.3))
    a = torch.relu(bn(y))
    a.retain_grad()
    z = conv(a)
    dz = rnd(*z.shape, seed=6)
    hs = rnd(*a.shape, seed=7) if acc else torch.zeros_like(a)
    ((z * dz).sum() + (a * hs).sum()).backward()
    mean, var = y.detach().mean(dim=(0, 2, 3, 4)), y.detach().var(dim=(0, 2, 3, 4), unbiased=False)
    inv = 1.0 / torch.sqrt(var + bn.eps)
    sc = (bn.weight * inv).detach()
    vec = K(torch.stack([sc, (bn.bias - mean * sc).detach(), mean, inv]))
    NP = L.msl_dwconv_s2_bwd_bnreduce_bww_num_partials(n, c, D, H, W)
    assert NP > 0
    bp = K(torch.full((2 * c * NP,), float("nan"), dtype=torch.float64))
    wp = K(torch.full((c * 27 * NP,), float("nan"), dtype=torch.float64))
    g_in = K(hs if acc else torch.full_like(hs, 9.0))
    _lib.call("msl_dwconv_s2_bwd_data_bnreduce_bww", ptr(K(dz)), ptr(K(conv.weight)), ptr(g_in), ptr(K(y)), ptr(vec[0]), ptr(vec[1]),
              ptr(vec[2]), ptr(vec[3]), ptr(bp), ptr(wp), n, c, D, H, W, acc, st())
    close(g_in, a.grad, 1e-4, 1e-5 * float(a.grad.abs().max()), "dL/d relu(bn(y))")
    sums = bp.view(2, c, NP).sum(-1).cpu()
    close(sums[0], bn.bias.grad, 2e-4, 1e-5 * float(bn.bias.grad.abs().max()), "sum gm (= dbeta)")
    close(sums[1], bn.weight.grad, 2e-4, 1e-5 * float(bn.weight.grad.abs().max()), "sum gm * xhat (= dgamma)")
    dw = wp.view(c * 27, NP).sum(-1).float().view(c, 1, 3, 3, 3)
    close(dw, conv.weight.grad, 2e-4, 1e-5 * float(conv.weight.grad.abs().max()), "depthwise weight gradient")


# ------------------------------------------------------------------------------------------------- fused eval stem + depthwise
@pytest.mark.parametrize("N,cin,dims", [(2, 1, (64, 64, 64)),      # AW 32, four output rows per workgroup
                                        (1, 1, (32, 48, 192)),     # AW 96 (the 192^3 inference rows), three rows, 1.5 chunks per row
                                        (1, 2, (16, 24, 128)),     # two input channels, AW 64
                                        (1, 1, (8, 8, 64)),        # two output rows per workgroup, two output planes
                                        (3, 1, (12, 16, 64))])     # odd plane count per segment, several images
@pytest.mark.parametrize("bf16", [False, True])
def test_stem_dw_eval_fused_equals_the_two_launches(N, cin, dims, bf16):
    """msl_stem_dw_fwd_eval[_bf16] (stem + BatchNorm affine + ReLU + block-1 depthwise convolution in one pass, the stem
    activation never in HBM) returns what msl_stem_conv_fwd + msl_dwconv_fwd return, bit for bit, and the torch composition
    within fp32 / bf16 tolerance."""
    L = _lib.load()
    assert L.msl_stem_dw_fwd_eval_supported(N, cin, *dims) == 1
    x = rnd(N, cin, *dims, seed=1)
    w = rnd(32, cin, 3, 3, 3, seed=2) / (27 * cin) ** 0.5
    wd = rnd(32, 1, 3, 3, 3, seed=3) / 27 ** 0.5
    sc, sh = rnd(32, seed=4).abs() + 0.5, rnd(32, seed=5, scale=0.3)
    a_dims = tuple(d // 2 for d in dims)
    z_dims = tuple(d // 4 for d in dims)
    dt = torch.bfloat16 if bf16 else torch.float32
    sfx = "_bf16" if bf16 else ""
    y = torch.empty((N, 32) + a_dims, dtype=dt, device=DEV)
    _lib.call("msl_stem_conv_fwd" + sfx, ptr(K(x)), ptr(K(w)), ptr(y), None, N, cin, *dims, 2, 2, 2, st())
    z_ref = torch.full((N, 32) + z_dims, float("nan"), dtype=dt, device=DEV)
    if bf16:
        _lib.call("msl_dwconv_fwd_bf16", ptr(y), ptr(K(sc)), ptr(K(sh)), ptr(K(wd)), ptr(z_ref), None, N, 32, *a_dims, 2, st())
    else:
        _lib.call("msl_dwconv_fwd", ptr(y), ptr(K(sc)), ptr(K(sh)), ptr(K(wd)), ptr(z_ref), None, N, 32, *a_dims, 2, 0, st())
    z = torch.full((N, 32) + z_dims, float("nan"), dtype=dt, device=DEV)
    _lib.call("msl_stem_dw_fwd_eval" + sfx, ptr(K(x)), ptr(K(w)), ptr(K(sc)), ptr(K(sh)), ptr(K(wd)), ptr(z), N, cin, *dims, st())
    assert not bool(torch.isnan(z.float()).any())
    assert torch.equal(z, z_ref), float((z.float() - z_ref.float()).abs().max())
    y0 = F.conv3d(x, w, stride=2, padding=1)
    ref = F.conv3d(affine_act(y0, sc, sh), wd, stride=2, padding=1, groups=32)
    if bf16:
        close(z.float(), ref, 4 * BF_EPS, 2e-2, "fused stem+dw bf16")
    else:
        close(z, ref, 1e-4, 1e-5, "fused stem+dw")


def test_stem_dw_eval_unsupported_shapes_are_refused():
    L = _lib.load()
    for N, cin, dims in [(1, 1, (30, 32, 64)), (1, 3, (32, 32, 64)), (1, 1, (32, 32, 96)), (1, 1, (32, 32, 256)), (0, 1, (32, 32, 64))]:
        assert L.msl_stem_dw_fwd_eval_supported(N, cin, *dims) == 0, (N, cin, dims)
    z = torch.zeros(1, device=DEV)
    with pytest.raises(_lib.HipKernelError):
        _lib.call("msl_stem_dw_fwd_eval", ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), 1, 1, 30, 32, 64, st())
