#!/usr/bin/env python3
"""Timing of the fused level entry points (arflow_level_fwd / arflow_level_bwd) at the pyramid shapes of BASELINE
config 2, warm (back-to-back on the same buffers) and "in-step cold": before every timed call a 600 MB device copy and
an unrelated GEMM run, so the call starts with cold L2 / Infinity Cache / instruction cache like inside a training step.

    python tools/level_bench.py [--iters 20] [--shapes 12x20,24x40,48x80,96x160]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from arflow_amd import _lib  # noqa: E402


def p(t):
    return None if t is None else t.data_ptr()


def main():
    if os.environ.get('ARFLOW_LIB_PATH'):
        _lib.LIB_PATH = os.environ['ARFLOW_LIB_PATH']
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--shapes', default='12x20,24x40,48x80,96x160')
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device('cuda')
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device='cuda').manual_seed(0)
    B, C = args.batch, 32
    big_a = torch.empty(150 * 1024 * 1024, device=dev)
    big_b = torch.empty_like(big_a)
    ma, mb = torch.randn(2048, 2048, device=dev), torch.randn(2048, 2048, device=dev)

    def thrash():
        big_b.copy_(big_a)
        torch.mm(ma, mb)

    def timeit(fn, cold):
        ts = []
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        if not cold:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.iters * 1e3
        for _ in range(args.iters):
            thrash()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    for shp in args.shapes.split(','):
        h, w = map(int, shp.split('x'))
        has_flow = (h, w) != (12, 20)
        x1 = torch.randn(B, C, h, w, device=dev, generator=g) * 0.7 + 0.3
        x2 = torch.randn(B, C, h, w, device=dev, generator=g) * 0.7 + 0.3
        fc = 0.7 * torch.randn(B, 2, h // 2, w // 2, device=dev, generator=g) if has_flow else None
        ctot = 81 + C + 2 + 32
        buf = torch.zeros(B, ctot, h, w, device=dev)
        gbuf = torch.randn(B, ctot, h, w, device=dev, generator=g)
        bs = ctot * h * w
        flow_up = torch.empty(B, 2, h, w, device=dev)
        x2w = torch.empty_like(x2)
        sign = torch.zeros(B, 3, h, w, device=dev, dtype=torch.int32)
        stats = torch.empty(B, 4, device=dev)
        rows = lib.arflow_level_acc_rows(B, C, h, w, int(has_flow))
        acc = torch.empty(4 * B * rows, device=dev, dtype=torch.float64)
        ws = torch.empty(lib.arflow_level_bwd_ws_bytes(B, C, h, w), device=dev, dtype=torch.uint8)
        gx1, gx2 = torch.empty_like(x1), torch.empty_like(x2)
        gfc = torch.empty(B, 2, h // 2, w // 2, device=dev)
        gext = torch.randn(B, 2, h, w, device=dev, generator=g)
        vol, x1n, fslot = buf[:, :81], buf[:, 81:81 + C], buf[:, 81 + C:]

        def fwd():
            _lib.check(lib.arflow_level_fwd(p(x1), p(x2), p(fc), 2 * (h // 2) * (w // 2), int(has_flow), 1,
                                            p(flow_up) if has_flow else None, fslot.data_ptr() if has_flow else None, bs,
                                            p(x2w) if has_flow else None, 0, vol.data_ptr(), bs, x1n.data_ptr(), bs, p(sign),
                                            p(stats), p(acc), B, C, h, w, 4, 0.1, 0, 1, 0, s), 'fwd')

        def bwd():
            _lib.check(lib.arflow_level_bwd(gbuf[:, :81].data_ptr(), bs, p(sign), x1n.data_ptr(), bs, gbuf[:, 81:].data_ptr(), bs,
                                            p(x1), p(x2), p(x2w) if has_flow else None, p(flow_up) if has_flow else None,
                                            2 * h * w, gbuf[:, 81 + C:].data_ptr() if has_flow else None, bs,
                                            p(gext) if has_flow else None, p(stats), 0, p(gx1), p(gx2),
                                            p(gfc) if has_flow else None, int(has_flow), 1, p(ws), B, C, h, w, 4, 0.1, 0, 1, 0,
                                            s), 'bwd')
        fwd()
        torch.cuda.synchronize()
        print('%3dx%-3d  fwd warm %6.1f us  cold %6.1f us   bwd warm %6.1f us  cold %6.1f us' % (
            h, w, timeit(fwd, False), timeit(fwd, True), timeit(bwd, False), timeit(bwd, True)), flush=True)


if __name__ == '__main__':
    main()
