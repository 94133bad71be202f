"""MobileNet-3D building blocks — host-side mirror of the reference's ``lesions3d/mobilenet.py:13-49``.

The modules below are parameter containers with the reference's attribute names (so ``state_dict`` keys and
PyTorch's seeded default initialisation are identical).  The arithmetic runs in the HIP kernels of
``mslesions3d_amd/csrc``, driven by ``engine.Engine``; the ``forward`` methods here run the SAME kernels for a
stand-alone layer (used by tests and by callers that compose layers by hand) and have no CPU path.
"""
import torch
import torch.nn as nn

from . import _lib
from ._lib import ptr

config_mobilenet = [32,
                    # channel, n, stride
                    [64, 1, (2, 2, 2)],
                    [128, 2, (2, 2, 2)],
                    [256, 2, (2, 2, 2)],
                    [512, 6, (2, 2, 2)],
                    [1024, 2, (1, 1, 1)],
                    ]

MOBILENET_CONFIGS = {"mobilenet": config_mobilenet}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(x):
    if not x.is_cuda:
        raise _lib.HipKernelError("mslesions3d_amd layers run on the HIP device only (no CPU fallback)")


def _bn_vectors(bn, partials, num_partials, count, C, device, training):
    vec = torch.empty((4, C), dtype=torch.float32, device=device)
    if training:
        _lib.call("msl_bn_finalize", ptr(partials), num_partials, float(count), ptr(bn.weight), ptr(bn.bias),
                  ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked),
                  0.1 if bn.momentum is None else bn.momentum, bn.eps, ptr(vec[0]), ptr(vec[1]), ptr(vec[2]),
                  ptr(vec[3]), C, _stream())
    else:
        _lib.call("msl_bn_eval_affine", ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
                  bn.eps, ptr(vec[0]), ptr(vec[1]), C, _stream())
    return vec


def _materialize(y, vec):
    N, C, D, H, W = y.shape
    out = torch.empty_like(y)
    _lib.call("msl_bn_relu_materialize", ptr(y), ptr(vec[0]), ptr(vec[1]), ptr(out), None, N, C, D, H, W, _stream())
    return out


class ConvBN(nn.Sequential):
    """``conv_bn`` of mobilenet.py:26-31: Conv3d(k3, p1, no bias) + BatchNorm3d + ReLU (stem of the backbone)."""

    def forward(self, x):
        _require_cuda(x)
        L = _lib.load()
        conv, bn = self[0], self[1]
        x = x.contiguous().float()
        N, Cin, D, H, W = x.shape
        sd, sh, sw = conv.stride
        od, oh, ow = (D - 1) // sd + 1, (H - 1) // sh + 1, (W - 1) // sw + 1
        y = torch.empty((N, conv.out_channels, od, oh, ow), dtype=torch.float32, device=x.device)
        NP = L.msl_stem_conv_fwd_num_partials(N, od, oh, ow)
        part = torch.empty(2 * conv.out_channels * NP, dtype=torch.float64, device=x.device)
        _lib.call("msl_stem_conv_fwd", ptr(x), ptr(conv.weight), ptr(y), ptr(part) if self.training else None, N, Cin,
                  D, H, W, sd, sh, sw, _stream())
        vec = _bn_vectors(bn, part, NP, N * od * oh * ow, conv.out_channels, x.device, self.training)
        return _materialize(y, vec)


def conv_bn(inp, oup, stride):
    if oup != 32:
        raise NotImplementedError("the HIP stem kernel is built for the reference's 32 output channels "
                                  "(width_mult != 1 crashes in the reference as well, SURVEY §0.2-8)")
    return ConvBN(
        nn.Conv3d(inp, oup, kernel_size=3, stride=stride, padding=(1, 1, 1), bias=False),
        nn.BatchNorm3d(oup),
        nn.ReLU(inplace=True),
    )


class Block(nn.Module):
    """Depthwise conv + Pointwise conv (mobilenet.py:34-49)."""

    def __init__(self, in_planes, out_planes, stride=1):
        super(Block, self).__init__()
        self.conv1 = nn.Conv3d(in_planes, in_planes, kernel_size=3, stride=stride, padding=1, groups=in_planes, bias=False)
        self.bn1 = nn.BatchNorm3d(in_planes)
        self.conv2 = nn.Conv3d(in_planes, out_planes, kernel_size=1, stride=1, padding=0, bias=False)
        self.bn2 = nn.BatchNorm3d(out_planes)

    def forward(self, x):
        """Stand-alone use on an ACTIVATION tensor x (N,C,D,H,W).  Inside LSSD3D the engine chains the raw
        tensors instead and never materialises the intermediate activations."""
        _require_cuda(x)
        L = _lib.load()
        x = x.contiguous().float()
        N, C, D, H, W = x.shape
        s = self.conv1.stride[0]
        od, oh, ow = (D - 1) // s + 1, (H - 1) // s + 1, (W - 1) // s + 1
        S = od * oh * ow
        Cout = self.conv2.out_channels
        st = _stream()
        z = torch.empty((N, C, od, oh, ow), dtype=torch.float32, device=x.device)
        NP = L.msl_dwconv_fwd_num_partials(N, C, D, H, W, s)
        part = torch.empty(2 * max(C * NP, Cout * L.msl_pwconv_fwd_num_partials(N, C, Cout, S)), dtype=torch.float64, device=x.device)
        pp = ptr(part) if self.training else None
        _lib.call("msl_dwconv_fwd", ptr(x), None, None, ptr(self.conv1.weight), ptr(z), pp, N, C, D, H, W, s, 0, st)
        v1 = _bn_vectors(self.bn1, part, NP, N * S, C, x.device, self.training)
        y = torch.empty((N, Cout, od, oh, ow), dtype=torch.float32, device=x.device)
        _lib.call("msl_pwconv_fwd", ptr(z), ptr(v1[0]), ptr(v1[1]), ptr(self.conv2.weight), ptr(y), pp, N, C, Cout, S, st)
        v2 = _bn_vectors(self.bn2, part, L.msl_pwconv_fwd_num_partials(N, C, Cout, S), N * S, Cout, x.device, self.training)
        out = _materialize(y, v2)
        if torch.isnan(out).sum() > 0:  # mobilenet.py:46-48
            raise Exception("NaN Loss in MobileNet Block")
        return out
