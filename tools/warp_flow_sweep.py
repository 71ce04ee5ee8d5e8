"""warp backward time vs flow statistics (smooth / noisy / large) at B16 C32 96x160."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arflow_amd import _lib
lib = _lib.load()
dev = 'cuda'; g = torch.Generator(device='cuda').manual_seed(0)
B, C, h, w = 16, 32, 96, 160
x1 = torch.randn(B, C, h, w, device=dev, generator=g); x2 = torch.randn(B, C, h, w, device=dev, generator=g)
g2 = torch.empty_like(x2); s = torch.cuda.current_stream().cuda_stream
def smooth(mag, cells):
    coarse = mag * torch.randn(B, 2, max(2, h // cells), max(2, w // cells), device=dev, generator=g)
    return torch.nn.functional.interpolate(coarse, (h, w), mode='bilinear', align_corners=True).contiguous()
cases = {'zero': torch.zeros(B, 2, h, w, device=dev), 'smooth 3px/16': smooth(3, 16), 'smooth 10px/16': smooth(10, 16),
         'smooth 3px/4 (rough)': smooth(3, 4), 'white noise 0.5px': 0.5 * torch.randn(B, 2, h, w, device=dev, generator=g),
         'white noise 2px': 2 * torch.randn(B, 2, h, w, device=dev, generator=g),
         'white noise 8px': 8 * torch.randn(B, 2, h, w, device=dev, generator=g),
         'smooth 3px + noise 1px': smooth(3, 16) + torch.randn(B, 2, h, w, device=dev, generator=g)}
for name, fl in cases.items():
    fl = fl.contiguous(); gfl = torch.empty_like(fl)
    def run(): lib.arflow_warp_bwd(x1.data_ptr(), x2.data_ptr(), fl.data_ptr(), g2.data_ptr(), gfl.data_ptr(), B, C, h, w, h, w, 2 * h * w, 0, 1, 0, s)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print('%-28s %.1f us' % (name, e0.elapsed_time(e1) / 20 * 1e3))
