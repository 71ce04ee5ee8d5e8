#!/usr/bin/env python3
"""Development check of the fused level launches against the unfused op chain (same device) + timings.

    python tools/level_check.py [--iters 30]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from arflow_amd import _lib, functional as AF  # noqa: E402


def p(t):
    return None if t is None else t.data_ptr()


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--batch', type=int, default=16)
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device('cuda')
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device='cuda').manual_seed(0)
    B, C = args.batch, 32
    worst = 0.0
    for (h, w) in [(12, 20), (24, 40), (48, 80), (96, 160)]:
        for mode in (0, 1):
            x1 = torch.randn(B, C, h, w, device=dev, generator=g) * 0.7 + 0.3
            x2 = torch.randn(B, C, h, w, device=dev, generator=g) * 0.7 + 0.3
            fc = 1.5 * torch.randn(B, 2, h // 2, w // 2, device=dev, generator=g)
            has_flow = h > 12
            # reference chain with the existing ops
            if has_flow:
                fl = F.interpolate(fc * 2, scale_factor=2, mode='bilinear', align_corners=True)
                x2w_ref = AF.warp(x2, fl, 'zeros', True, AF.NORM_ARFLOW)
            else:
                fl, x2w_ref = None, x2
            y1, y2 = AF.normalize_pair(x1, x2w_ref, 'joint' if mode == 0 else 'avg')
            vol_ref = AF.correlation(y1, y2, 4, 0.1)
            # fused
            rows = lib.arflow_level_acc_rows(B, C, h, w, int(has_flow))
            acc = torch.empty(4 * B * rows, device=dev, dtype=torch.float64)
            ctot = 81 + C + 2
            buf = torch.zeros(B, ctot, h, w, device=dev)
            flow_up = torch.empty(B, 2, h, w, device=dev)
            x2w = torch.empty_like(x2)
            sign = torch.zeros(B, 3, h, w, device=dev, dtype=torch.int32)
            stats = torch.empty(B, 4, device=dev)
            vol, x1n, fslot = buf[:, :81], buf[:, 81:81 + C], buf[:, 81 + C:]

            def k_a():
                if has_flow:
                    _lib.check(lib.arflow_level_warp_fwd(p(x1), p(x2), p(fc), 2 * (h // 2) * (w // 2), 1, 1, p(flow_up),
                                                         fslot.data_ptr(), ctot * h * w, p(x2w), p(acc), B, C, h, w, 0, 1, 0, s), 'warp')
                else:
                    _lib.check(lib.arflow_level_moments(p(x1), p(x2), p(acc), B, C * h * w, s), 'moments')

            def k_b():
                _lib.check(lib.arflow_level_corr_fwd(p(x1), p(x2w) if has_flow else p(x2), p(acc), rows, mode, vol.data_ptr(),
                                                     ctot * h * w, x1n.data_ptr(), ctot * h * w, p(sign), p(stats), B, C, h, w, 4,
                                                     0.1, s), 'corr')
            k_a()
            k_b()
            torch.cuda.synchronize()
            errs = {}
            if has_flow:
                errs['flow_up'] = float((flow_up - fl).abs().max())
                errs['flow_slot'] = float((fslot - fl).abs().max())
                errs['x2w'] = float((x2w - x2w_ref).abs().max())
            errs['x1n'] = float((x1n - y1).abs().max())
            errs['vol'] = float((vol - vol_ref).abs().max())
            worst = max(worst, errs['vol'], errs['x1n'])
            ta, tb = timeit(k_a, args.iters), timeit(k_b, args.iters)
            def k_b0():
                lib.arflow_level_corr_fwd(p(x1), p(x2w) if has_flow else p(x2), p(acc), 0, mode, vol.data_ptr(), ctot * h * w, x1n.data_ptr(), ctot * h * w, p(sign), p(stats), B, C, h, w, 4, 0.1, s)
            def k_b1():
                lib.arflow_level_corr_fwd(p(x1), p(x2w) if has_flow else p(x2), p(acc), rows, mode, vol.data_ptr(), ctot * h * w, None, 0, p(sign), p(stats), B, C, h, w, 4, 0.1, s)
            def k_c():
                lib.arflow_corr_fwd_strided(p(x1), p(x2w) if has_flow else p(x2), vol.data_ptr(), ctot * h * w, p(sign), B, C, h, w, 4, 0.1, s)
            print('     nostats %.1f  nox1n %.1f  plain corr %.1f' % (timeit(k_b0, args.iters), timeit(k_b1, args.iters), timeit(k_c, args.iters)))
            print('%3dx%-3d mode %d  K_A %6.1f us  K_B %6.1f us   %s' % (h, w, mode, ta, tb, ' '.join('%s=%.2e' % kv for kv in errs.items())), flush=True)
    print('worst', worst)
    assert worst < 5e-5


if __name__ == '__main__':
    main()
