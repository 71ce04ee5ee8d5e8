"""Per-step wall time of a workload (outlier hunting): python tools/step_times.py WORKLOAD [nogc]"""
import gc
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench  # noqa: F401  (points MIOpen at the shipped tuning records)
from arflow_amd.train_step import TrainStep, synthetic_pairs
dev = torch.device('cuda', 0)
step = TrainStep(sys.argv[1], dev, seed=1234)
img = synthetic_pairs(8, 384, 640, frames=step.model_cfg.get('n_frames', 2), device=dev, seed=100)
for _ in range(4):
    step(img)
torch.cuda.synchronize()
if len(sys.argv) > 2 and sys.argv[2] == 'nogc':
    gc.collect()
    gc.disable()
ts = []
for i in range(60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(img)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
s = sorted(ts)
print(sys.argv[1:], 'median %.1f mean %.1f max %.1f; > 1.3x median at steps %s' % (
    s[len(s) // 2], sum(ts) / len(ts), s[-1], [i for i, t in enumerate(ts) if t > 1.3 * s[len(s) // 2]]))
print('mem stats: num_alloc_retries', torch.cuda.memory_stats().get('num_alloc_retries'), 'num_device_alloc', torch.cuda.memory_stats().get('num_device_alloc'))
