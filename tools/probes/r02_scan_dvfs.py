"""Is the single-workgroup NMS scan slow because the chip idles at a low clock?  Runs detect_objects alone and beside a big
matmul on another stream; compare detect_scan_kernel / detect_finalize_kernel durations in rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mslesions3d_amd.ssd3d import LSSD3D  # noqa: E402

busy = len(sys.argv) > 1 and sys.argv[1] == "busy"
dev = torch.device("cuda")
size = (192, 192, 192)
torch.manual_seed(0)
m = LSSD3D(n_classes=2, input_channels=1, input_size=size, threshold=[0.1, 0.2]).to(dev).eval()
x = torch.rand((2, 1) + size, device=dev)
with torch.no_grad():
    locs, scores = m(x)
    locs, scores = locs.clone(), scores.clone()
    a = torch.randn(8192, 8192, device=dev)
    side = torch.cuda.Stream()
    for it in range(40):
        if busy:
            with torch.cuda.stream(side):
                for _ in range(3):
                    a @ a
        m.detect_objects(locs, scores, min_score=0.3, max_overlap=0.3, top_k=50)
        torch.cuda.synchronize()
print("done", "busy" if busy else "alone")
