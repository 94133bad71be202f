import torch, torch.nn.functional as F, sys
sys.path.insert(0, '.')
from mslesions3d_amd import _lib
from mslesions3d_amd._lib import ptr
L = _lib.load()
torch.manual_seed(0)
N, cin, dims, stride = 2, 1, (16,16,16), (2,2,2)
x = torch.randn(N, cin, *dims); w = torch.randn(32, cin, 3,3,3)*0.3
ref = F.conv3d(x, w, stride=stride, padding=1)
y = torch.zeros(ref.shape, device='cuda')
od,oh,ow = ref.shape[2:]
NP = L.msl_stem_conv_fwd_num_partials(N, od, oh, ow)
part = torch.zeros(2*32*NP, dtype=torch.float64, device='cuda')
_lib.call("msl_stem_conv_fwd", ptr(x.cuda()), ptr(w.cuda()), ptr(y), ptr(part), N, cin, *dims, *stride, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
err = (y.cpu()-ref).abs()
print("per (n,c) max err:")
print(err.amax((2,3,4)))
bad = err > 1e-3
print("bad per channel", bad.sum((0,2,3,4)))
n,c = 0, int(bad.sum((0,2,3,4)).argmax())
print("channel", c, "bad positions (od,oh,ow) count", bad[n,c].sum().item())
idx = bad[n,c].nonzero()[:10]
print(idx)
for i in idx[:5]:
    print(tuple(i.tolist()), y[n,c][tuple(i.tolist())].item(), ref[n,c][tuple(i.tolist())].item())
