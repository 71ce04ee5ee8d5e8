"""Parameter gradients of one training step with the fused level op vs the unfused op chain (same weights, same input)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arflow_amd import functional as AF
from arflow_amd.train_step import TrainStep
dev = torch.device('cuda')
g = torch.Generator().manual_seed(3)
B, H, W, dx, dy = 4, 128, 192, 3, 2
base = torch.rand(B, 3, (H + 16) // 8, (W + 16) // 8, generator=g)
tex = torch.nn.functional.interpolate(base, (H + 16, W + 16), mode='bicubic', align_corners=False).clamp(0, 1)
tex = (tex + 0.1 * torch.rand(B, 3, H + 16, W + 16, generator=g)).clamp(0, 1)
im1 = tex[:, :, 8:8 + H, 8:8 + W]
im2 = tex[:, :, 8 - dy:8 - dy + H, 8 - dx:8 - dx + W]
x = torch.cat([im1, im2], 1).contiguous().to(dev)
res = {}
for fused in (True, False):
    AF._LEVEL_FUSED = fused
    torch.manual_seed(0)
    step = TrainStep('pwclite_uflow+uflow_loss', dev, lr=1e-4, seed=1)
    step.model.level_dropout = 0.0
    m = step.model
    out = m(x, with_bk=True)
    flows = [torch.cat([a, b], 1) for a, b in zip(out['flows_fw'], out['flows_bw'])]
    l = step.loss(flows, x)
    m.zero_grad()
    l[0].backward()
    res[fused] = (float(l[0]), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}, [f.detach().clone() for f in flows])
print('loss fused %.7f unfused %.7f' % (res[True][0], res[False][0]))
for i, (a, b) in enumerate(zip(res[True][2], res[False][2])):
    print('flow level %d max diff %.3e (max |f| %.3e)' % (i, float((a - b).abs().max()), float(b.abs().max())))
worst = []
for n in res[True][1]:
    a, b = res[True][1][n], res[False][1][n]
    worst.append((float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30), n, float(b.abs().max())))
worst.sort(reverse=True)
for w in worst[:12]:
    print('%.3e  %s  (max|g| %.3e)' % w)
