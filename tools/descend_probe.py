import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arflow_amd.train_step import TrainStep
dev = torch.device('cuda')
torch.manual_seed(0)
g = torch.Generator().manual_seed(3)
B, H, W, dx, dy = 4, 128, 192, 3, 2
base = torch.rand(B, 3, (H + 16) // 8, (W + 16) // 8, generator=g)
tex = torch.nn.functional.interpolate(base, (H + 16, W + 16), mode='bicubic', align_corners=False).clamp(0, 1)
tex = (tex + 0.1 * torch.rand(B, 3, H + 16, W + 16, generator=g)).clamp(0, 1)
im1 = tex[:, :, 8:8 + H, 8:8 + W]
im2 = tex[:, :, 8 - dy:8 - dy + H, 8 - dx:8 - dx + W]
x = torch.cat([im1, im2], 1).contiguous().to(dev)
step = TrainStep('pwclite_uflow+uflow_loss', dev, lr=1e-4, seed=1)
step.model.level_dropout = 0.0
losses = [float(step(x)) for _ in range(40)]
print(os.environ.get('ARFLOW_LEVEL_FUSED', '1'), ' '.join('%.4f' % l for l in losses[:6]), '...', ' '.join('%.4f' % l for l in losses[-5:]))
