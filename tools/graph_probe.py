"""How much of the step is launch gaps?  Captures one full training step (level dropout off: its CPU random
draws cannot be captured) in a HIP graph and compares replay time with eager time.  Probe only."""
import gc
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench  # noqa: F401
from arflow_amd import train_step as TS

name = sys.argv[1] if len(sys.argv) > 1 else 'pwclite_uflow+uflow_loss'
mcfg, lcfg = TS.WORKLOADS[name]
mcfg = dict(mcfg, level_dropout=0.0)
TS.WORKLOADS['probe'] = (mcfg, lcfg)
dev = torch.device('cuda', 0)
step = TS.TrainStep('probe', dev, seed=1234)
step.opt = torch.optim.Adam(step.model.parameters(), lr=1e-4, capturable=True, fused=True)
img = TS.synthetic_pairs(8, 384, 640, frames=mcfg.get('n_frames', 2), device=dev, seed=100)


def timeit(fn, n=20):
    torch.cuda.synchronize(); gc.collect(); gc.disable()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    gc.enable()
    return (time.perf_counter() - t0) / n * 1e3


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(5):
        step(img)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print('eager  %.2f ms/step' % timeit(lambda: step(img)))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step(img)
torch.cuda.synchronize()
print('replay %.2f ms/step' % timeit(g.replay))
