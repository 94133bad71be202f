"""Per-kernel fp64 check at the exact shapes of block 5 in the 128^3 x 2 configuration (N=2, 256 -> 256 channels, S=512)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mslesions3d_amd import _lib
from mslesions3d_amd._lib import ptr
L = _lib.load()
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max())
for N, Cin, Cout, S in ((2, 256, 256, 512), (2, 128, 256, 512), (4, 256, 256, 512), (2, 256, 512, 64)):
    z = torch.randn(N, Cin, S) * 2
    sc, sh = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.3
    dy = torch.randn(N, Cout, S)
    w = torch.randn(Cout, Cin) / Cin ** 0.5
    a = torch.relu(z.double() * sc.double().view(1, -1, 1) + sh.double().view(1, -1, 1))
    dw_ref = torch.einsum("nos,ncs->oc", dy.double(), a)
    g_ref = torch.einsum("oc,nos->ncs", w.double(), dy.double())
    zd, scd, shd, dyd, wd = (t.to(dev).contiguous() for t in (z, sc, sh, dy, w))
    dw = torch.empty(Cout, Cin, device=dev)
    ws = torch.empty(max(L.msl_pwconv_bwd_weight_workspace_bytes(N, Cin, Cout, S) // 4, 1), device=dev)
    _lib.call("msl_pwconv_bwd_weight", ptr(dyd), ptr(zd), ptr(scd), ptr(shd), ptr(dw), ptr(ws), N, Cin, Cout, S, st)
    g = torch.empty(N, Cin, S, device=dev)
    _lib.call("msl_pwconv_bwd_data", ptr(dyd), ptr(wd), ptr(g), N, Cin, Cout, S, st)
    # fp32 CPU for scale
    dw32 = torch.einsum("nos,ncs->oc", dy, torch.relu(z * sc.view(1, -1, 1) + sh.view(1, -1, 1)))
    print(f"N={N} Cin={Cin} Cout={Cout} S={S}: pw_bwd_weight err {rel(dw, dw_ref):.2e} (torch fp32 einsum {rel(dw32, dw_ref):.2e}); "
          f"pw_bwd_data err {rel(g, g_ref):.2e}")
# BatchNorm backward fused, C=256, S=512, N=2
N, C, S = 2, 256, 512
y = torch.randn(N, C, S) * 1.5 + 0.3
gin = torch.randn(N, C, S)
gamma, beta = torch.rand(C) + 0.5, torch.randn(C) * 0.2
yd64 = y.double().requires_grad_(True)
mean = yd64.mean((0, 2), keepdim=True); var = yd64.var((0, 2), unbiased=False, keepdim=True)
out = torch.relu((yd64 - mean) / torch.sqrt(var + 1e-5) * gamma.double().view(1, -1, 1) + beta.double().view(1, -1, 1))
out.backward(gin.double())
part = torch.stack([y.double().sum((0, 2)), (y.double() ** 2).sum((0, 2))]).view(2, C, 1).contiguous().to(dev)
vec = torch.zeros(8, C, device=dev)
gam_d, bet_d = gamma.to(dev), beta.to(dev)
_lib.call("msl_bn_finalize", ptr(part), 1, float(N * S), ptr(gam_d), ptr(bet_d), None, None, None, 0.1, 1e-5, ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]), C, st)
gd, ydv = gin.to(dev).contiguous(), y.to(dev).contiguous()
dgam, dbet, dyv = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(N, C, S, device=dev)
_lib.call("msl_bn_relu_bwd_fused", ptr(gd), ptr(ydv), ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]), ptr(dgam), ptr(dbet), ptr(dyv), N, C, S, st)
print(f"bn_relu_bwd_fused N={N} C={C} S={S}: dy err {rel(dyv, yd64.grad):.2e}")
