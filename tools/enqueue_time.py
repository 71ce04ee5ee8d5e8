"""Host enqueue time vs total time of a step (is the step launch-bound?): python tools/enqueue_time.py WORKLOAD"""
import gc, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from arflow_amd.train_step import TrainStep, synthetic_pairs
dev = torch.device('cuda', 0)
step = TrainStep(sys.argv[1], dev, seed=1234)
img = synthetic_pairs(8, 384, 640, frames=step.model_cfg.get('n_frames', 2), device=dev, seed=100)
for _ in range(5): step(img)
torch.cuda.synchronize(); gc.collect(); gc.disable()
enq = []; tot = []
for i in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step(img)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print(sys.argv[1], 'enqueue ms', ['%.1f' % e for e in enq], 'total', ['%.1f' % t for t in tot])
