#!/usr/bin/env python3
"""Cold-cache variant of tools/kbench.py for the finest-level correlation / warp / featnorm launches: every iteration
works on a DIFFERENT set of buffers (NSETS sets, > 1 GB in total, far beyond the 256 MB Infinity Cache), so inputs come
from HBM as they do inside a training step.  Explains the in-step vs kbench gap of bench.py's per-kernel timings.

    python tools/kbench_cold.py [--sets 8] [--iters 40]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from arflow_amd import _lib  # noqa: E402
from bench import algorithmic_bytes, HBM_PEAK_GBS  # noqa: E402


def p(t):
    return None if t is None else t.data_ptr()


def main():
    if os.environ.get('ARFLOW_LIB_PATH'):
        _lib.LIB_PATH = os.environ['ARFLOW_LIB_PATH']
    ap = argparse.ArgumentParser()
    ap.add_argument('--sets', type=int, default=8)
    ap.add_argument('--iters', type=int, default=40)
    ap.add_argument('--filter', default='')
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device('cuda')
    s = torch.cuda.current_stream().cuda_stream
    B, C, h, w = 16, 32, 96, 160
    g = torch.Generator(device='cuda').manual_seed(0)
    sets = []
    for _ in range(args.sets):
        d = dict(x1=torch.randn(B, C, h, w, device=dev, generator=g), x2=torch.randn(B, C, h, w, device=dev, generator=g),
                 out=torch.empty(B, 81, h, w, device=dev), go=torch.randn(B, 81, h, w, device=dev, generator=g),
                 g1=torch.empty(B, C, h, w, device=dev), g2=torch.empty(B, C, h, w, device=dev),
                 fl=2.0 * torch.randn(B, 2, h, w, device=dev, generator=g), gfl=torch.empty(B, 2, h, w, device=dev),
                 sign=torch.zeros(B, 3, h, w, device=dev, dtype=torch.int32),
                 acc=torch.empty(4 * (2048 + B), device=dev, dtype=torch.float64), st=torch.empty(B, 4, device=dev))
        sets.append(d)

    def timeit(fn):
        for i in range(args.sets):
            fn(sets[i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.iters):
            fn(sets[i % args.sets])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters * 1e3

    n = C * h * w
    ops = [
        ('arflow_corr_fwd', (B, C, h, w, 4, 3), lambda d: lib.arflow_corr_fwd(p(d['x1']), p(d['x2']), p(d['out']), p(d['sign']), B, C, h, w, 4, 0.1, s)),
        ('arflow_corr_bwd', (B, C, h, w, 4, 3), lambda d: lib.arflow_corr_bwd(p(d['go']), None, p(d['sign']), p(d['x1']), p(d['x2']), p(d['g1']), p(d['g2']), B, C, h, w, 4, 0.1, s)),
        ('arflow_warp_fwd', (B, C, h, w), lambda d: lib.arflow_warp_fwd(p(d['x2']), p(d['fl']), p(d['g1']), None, B, C, h, w, h, w, 2 * h * w, 0, 1, 0, s)),
        ('arflow_warp_bwd', (B, C, h, w, True), lambda d: lib.arflow_warp_bwd(p(d['x1']), p(d['x2']), p(d['fl']), p(d['g2']), p(d['gfl']), B, C, h, w, h, w, 2 * h * w, 0, 1, 0, s)),
        ('arflow_featnorm_fwd', (B, n), lambda d: lib.arflow_featnorm_fwd(p(d['x1']), p(d['x2']), p(d['g1']), p(d['g2']), p(d['acc']), p(d['st']), B, n, 0, s)),
        ('arflow_featnorm_bwd', (B, n), lambda d: lib.arflow_featnorm_bwd(p(d['go'][:, :C]), p(d['out'][:, :C]), p(d['x1']), p(d['x2']), p(d['st']), p(d['acc']), p(d['g1']), p(d['g2']), B, n, 0, s)),
    ]
    # fused level launches on the same cold buffer sets
    rows = lib.arflow_level_acc_rows(B, C, h, w, 1)
    for d in sets:
        d['fc'] = 1.5 * torch.randn(B, 2, h // 2, w // 2, device=dev, generator=g)
        d['lacc'] = torch.empty(4 * B * rows, device=dev, dtype=torch.float64)
        d['fu'] = torch.empty(B, 2, h, w, device=dev)
        lib.arflow_level_warp_fwd(p(d['x1']), p(d['x2']), p(d['fc']), 2 * (h // 2) * (w // 2), 1, 1, p(d['fu']), None, 0, p(d['g2']),
                                  p(d['lacc']), B, C, h, w, 0, 1, 0, s)
        lib.arflow_level_corr_fwd(p(d['x1']), p(d['g2']), p(d['lacc']), rows, 0, p(d['out']), 81 * h * w, p(d['g1']), C * h * w,
                                  p(d['sign']), p(d['st']), B, C, h, w, 4, 0.1, s)
    ops += [
        ('arflow_level_warp_fwd', (B, C, h, w, 1), lambda d: lib.arflow_level_warp_fwd(p(d['x1']), p(d['x2']), p(d['fc']), 2 * (h // 2) * (w // 2), 1, 1, p(d['fu']), None, 0, p(d['g2']), p(d['lacc']), B, C, h, w, 0, 1, 0, s)),
        ('arflow_level_corr_fwd', (B, C, h, w, 4, 3), lambda d: lib.arflow_level_corr_fwd(p(d['x1']), p(d['x2']), p(d['lacc']), rows, 0, p(d['out']), 81 * h * w, p(d['g1']), C * h * w, p(d['sign']), p(d['st']), B, C, h, w, 4, 0.1, s)),
        ('arflow_level_corr_bwd', (B, C, h, w, 4, 3), lambda d: lib.arflow_level_corr_bwd(p(d['go']), 81 * h * w, p(d['sign']), p(d['x1']), C * h * w, p(d['x2']), p(d['st']), p(d['g1']), p(d['g2']), B, C, h, w, 4, 0.1, s)),
    ]
    if args.filter:
        ops = [o for o in ops if any(f in o[0] for f in args.filter.split('|'))]
    for name, shape, fn in ops:
        us = timeit(fn)
        nb = algorithmic_bytes(name, shape)
        print('%-22s %-26s %9.1f us %9.1f GB/s  %5.1f%% of HBM peak  (cold)' % (name, list(shape), us, nb / us / 1e3, 100 * nb / us / 1e3 / HBM_PEAK_GBS), flush=True)


if __name__ == '__main__':
    main()
