"""CPU emulation of the bf16 activation path's STORAGE semantics (test infrastructure, next to the fp32 oracle).

The bf16 path (mslesions3d_amd/csrc/bf16.hip, DESIGN "bf16") is a build-side extension: the reference is fp32 everywhere,
so there is no reference output to pin it to.  What can be pinned is that the HIP kernels compute exactly the arithmetic the
design states - fp32 math on values that were rounded to bf16 at these points:

  forward   every raw convolution output y / z is stored as bf16 (its BatchNorm statistics come from the UNROUNDED
            fp32 accumulators); the pointwise GEMM takes bf16 operands (activation rounded after its fp32 affine + ReLU,
            weights rounded), the depthwise and stem convolutions fp32 operands; the head convolutions fp32 operands in
            the training step (fp32 kernels on an fp32 feature copy), bf16 operands in the inference path;
  backward  every activation gradient that goes to HBM (BatchNorm-backward outputs, conv bwd-data outputs, the head
            bwd-data output, their sums) is stored as bf16; weight gradients and all reductions are fp32 / fp64.

This file restates that with stock torch CPU ops (manual forward + backward over the oracle's parameters), so the HIP step
can be compared against it tightly (only summation order differs), while the distance to the fp32 oracle is reported as
the cost of the storage format."""
import torch
import torch.nn.functional as F


def bf(t):
    return t.to(torch.bfloat16).float()


def bfg(t):
    """rounding of an activation GRADIENT on its way to HBM"""
    return bf(t)


def _bn_vectors(raw, gamma, beta, eps=1e-5):
    dims = (0, 2, 3, 4)
    mean = raw.double().mean(dims)
    var = (raw.double() ** 2).mean(dims) - mean ** 2
    invstd = 1.0 / torch.sqrt(var.clamp(min=0) + eps)
    scale = (gamma.double() * invstd).float()
    shift = (beta.double() - mean * gamma.double() * invstd).float()
    return scale, shift, mean.float(), invstd.float()


def _act(yb, scale, shift):
    return torch.relu(yb * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1))


def _bn_bwd(g, yb, vec, gamma):
    """g = dL/d relu(bn(y)) (bf16 values) -> (dL/dy in fp32, dgamma, dbeta); the kernels' algebra (bf16.hip:427)."""
    scale, shift, mean, invstd = vec
    v = lambda t: t.view(1, -1, 1, 1, 1)
    pre = yb * v(scale) + v(shift)
    gm = (g * (pre > 0)).double()
    xhat = ((yb - v(mean)) * v(invstd)).double()
    cnt = g.numel() / g.shape[1]
    dbeta = gm.sum((0, 2, 3, 4))
    dgamma = (gm * xhat).sum((0, 2, 3, 4))
    dy = v(scale).double() * (gm - v(dbeta) / cnt - xhat * v(dgamma) / cnt)
    return dy.float(), dgamma.float(), dbeta.float()


def emulated_step(om, x, dlocs=None, dscores=None, f32_heads=True, fused_stem=True):
    """Forward (train mode, batch statistics) of the oracle model ``om`` under bf16 storage; with upstream gradients also
    the backward.  Returns (locs, scores, grads) with grads keyed like ``om.named_parameters()`` (None without upstream).
    ``f32_heads``: the training step's head convolutions run on the fp32 kernels (fp32 feature copy, unrounded weights);
    False: the bf16 head kernel of the inference path (bf16 feature copy and weights).  ``fused_stem``: the gradient of the stem
    activation is rebuilt on the fly from dL/dz of block 1 and never stored, hence never rounded (cubic inputs)."""
    feats = om.base.features
    nblk = len(feats)
    N = x.shape[0]
    P = {k: p.detach() for k, p in om.named_parameters()}
    yb, zb, vy, vz = [None] * nblk, [None] * nblk, [None] * nblk, [None] * nblk
    stem_stride = feats[0][0].stride
    raw = F.conv3d(x, P["base.features.0.0.weight"], stride=stem_stride, padding=1)
    vy[0] = _bn_vectors(raw, P["base.features.0.1.weight"], P["base.features.0.1.bias"])
    yb[0] = bf(raw)
    for i in range(1, nblk):
        n = f"base.features.{i}"
        s = feats[i].conv1.stride
        a = _act(yb[i - 1], vy[i - 1][0], vy[i - 1][1])                         # fp32 operand of the depthwise conv
        raw = F.conv3d(a, P[n + ".conv1.weight"], stride=s, padding=1, groups=a.shape[1])
        vz[i] = _bn_vectors(raw, P[n + ".bn1.weight"], P[n + ".bn1.bias"])
        zb[i] = bf(raw)
        a = bf(_act(zb[i], vz[i][0], vz[i][1]))                                 # bf16 operand of the pointwise GEMM
        raw = F.conv3d(a, bf(P[n + ".conv2.weight"]))
        vy[i] = _bn_vectors(raw, P[n + ".bn2.weight"], P[n + ".bn2.bias"])
        yb[i] = bf(raw)
    fids = [3, 5, 7]
    fmap, locs, scores = {}, [], []
    ncls = om.n_classes
    for k, f in enumerate(fids):
        hr = (lambda t: t) if f32_heads else bf
        fmap[f] = hr(_act(yb[f], vy[f][0], vy[f][1]))                           # the materialised feature copy
        lw, cw = P[f"pred_convs.loc_convs.{k}.weight"], P[f"pred_convs.cl_convs.{k}.weight"]
        lo = F.conv3d(fmap[f], hr(lw), P[f"pred_convs.loc_convs.{k}.bias"], padding=1)
        sc = F.conv3d(fmap[f], hr(cw), P[f"pred_convs.cl_convs.{k}.bias"], padding=1)
        locs.append(lo.permute(0, 2, 3, 4, 1).reshape(N, -1, 6))
        scores.append(sc.permute(0, 2, 3, 4, 1).reshape(N, -1, ncls))
    locs, scores = torch.cat(locs, 1), torch.cat(scores, 1)
    if dlocs is None:
        return locs, scores, None

    G = {}
    gy = [None] * nblk
    off = 0
    for k, f in enumerate(fids):
        D, H, W = fmap[f].shape[2:]
        cnt = D * H * W * 2
        dl = dlocs[:, off:off + cnt].reshape(N, D, H, W, 12).permute(0, 4, 1, 2, 3).contiguous()
        dc = dscores[:, off:off + cnt].reshape(N, D, H, W, 2 * ncls).permute(0, 4, 1, 2, 3).contiguous()
        off += cnt
        lw, cw = P[f"pred_convs.loc_convs.{k}.weight"], P[f"pred_convs.cl_convs.{k}.weight"]
        # bwd-data: fp32 dO x fp32 weights; weight gradient: fp32 dO x the feature copy
        gy[f] = bfg(torch.nn.grad.conv3d_input(fmap[f].shape, lw, dl, padding=1) + torch.nn.grad.conv3d_input(fmap[f].shape, cw, dc, padding=1))
        G[f"pred_convs.loc_convs.{k}.weight"] = torch.nn.grad.conv3d_weight(fmap[f], lw.shape, dl, padding=1)
        G[f"pred_convs.cl_convs.{k}.weight"] = torch.nn.grad.conv3d_weight(fmap[f], cw.shape, dc, padding=1)
        G[f"pred_convs.loc_convs.{k}.bias"] = dl.sum((0, 2, 3, 4))
        G[f"pred_convs.cl_convs.{k}.bias"] = dc.sum((0, 2, 3, 4))
    for i in range(nblk - 1, 0, -1):
        n = f"base.features.{i}"
        s = feats[i].conv1.stride
        dy, G[n + ".bn2.weight"], G[n + ".bn2.bias"] = _bn_bwd(gy[i], yb[i], vy[i], P[n + ".bn2.weight"])
        dy = bfg(dy)
        a = bf(_act(zb[i], vz[i][0], vz[i][1]))
        w2 = P[n + ".conv2.weight"]
        G[n + ".conv2.weight"] = torch.einsum("nodhw,ncdhw->oc", dy.double(), a.double()).float().view(w2.shape)
        gz = bfg(torch.einsum("oc,nodhw->ncdhw", bf(w2).view(w2.shape[0], w2.shape[1]), dy))
        dz, G[n + ".bn1.weight"], G[n + ".bn1.bias"] = _bn_bwd(gz, zb[i], vz[i], P[n + ".bn1.weight"])
        dz = bfg(dz)
        a_in = _act(yb[i - 1], vy[i - 1][0], vy[i - 1][1])
        w1 = P[n + ".conv1.weight"]
        G[n + ".conv1.weight"] = torch.nn.grad.conv3d_weight(a_in, w1.shape, dz, stride=s, padding=1, groups=a_in.shape[1])
        gin = torch.nn.grad.conv3d_input(a_in.shape, w1, dz, stride=s, padding=1, groups=a_in.shape[1])
        gy[i - 1] = gin if (i == 1 and fused_stem) else bfg(gin if gy[i - 1] is None else gy[i - 1] + gin)
    # stem: the weight gradient applies the BatchNorm backward on load, in fp32 (nothing is rounded in between)
    dy, G["base.features.0.1.weight"], G["base.features.0.1.bias"] = _bn_bwd(gy[0], yb[0], vy[0], P["base.features.0.1.weight"])
    w0 = P["base.features.0.0.weight"]
    G["base.features.0.0.weight"] = torch.nn.grad.conv3d_weight(x, w0.shape, dy, stride=stem_stride, padding=1)
    return locs, scores, G
