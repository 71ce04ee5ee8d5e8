import torch
dev='cuda'
for n_mb in (31.5, 63, 126, 252, 1000):
    n=int(n_mb*1e6/4)
    x=torch.randn(n,device=dev); y=torch.empty_like(x)
    for _ in range(5): y.copy_(x)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/50*1e3
    print('copy %.1f MB -> %.1f MB: %.1f us  = %.0f GB/s (read+write)'%(n_mb,n_mb,us, 2*n_mb*1e6/us/1e3))
