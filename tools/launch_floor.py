"""Launch floor on the GPU box: a trivial entry point back to back (3.3 us per launch, 2.4 us inside a HIP graph) and the
coarsest correlation level at batch 1..64 (13 us whatever the batch: the per-workgroup latency chain, not the data).
    python tools/launch_floor.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from arflow_amd import _lib
from tools.kbench import timeit, p
lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
fl = torch.randn(1, 2, 8, 8, device='cuda'); out = torch.empty(1, 1, 8, 8, device='cuda')
print('tiny coord_mask launch: %.2f us' % timeit(lambda: lib.arflow_coord_mask(p(fl), p(out), 1, 8, 8, 128, 0, s), 200))
x1 = torch.randn(16, 32, 12, 20, device='cuda'); x2 = torch.randn(16, 32, 12, 20, device='cuda'); o = torch.empty(16, 81, 12, 20, device='cuda')
sg = torch.zeros(16, 3, 12, 20, device='cuda', dtype=torch.int32)
print('corr_fwd 12x20: %.2f us' % timeit(lambda: lib.arflow_corr_fwd(p(x1), p(x2), p(o), p(sg), 16, 32, 12, 20, 4, 0.1, s), 200))
for B in (1, 4, 16, 64):
    x1 = torch.randn(B, 32, 12, 20, device='cuda'); x2 = torch.randn(B, 32, 12, 20, device='cuda'); o = torch.empty(B, 81, 12, 20, device='cuda')
    sg = torch.zeros(B, 3, 12, 20, device='cuda', dtype=torch.int32)
    print('corr_fwd 12x20 B=%d: %.2f us' % (B, timeit(lambda: lib.arflow_corr_fwd(p(x1), p(x2), p(o), p(sg), B, 32, 12, 20, 4, 0.1, s), 200)))
# graph replay of 10 tiny launches
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    ss = st.cuda_stream
    for _ in range(3): lib.arflow_coord_mask(p(fl), p(out), 1, 8, 8, 128, 0, ss)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=st):
        for _ in range(10): lib.arflow_coord_mask(p(fl), p(out), 1, 8, 8, 128, 0, st.cuda_stream)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): g.replay()
e1.record(); torch.cuda.synchronize()
print('graph of 10 tiny launches: %.2f us per launch' % (e0.elapsed_time(e1) * 1e3 / 500))
