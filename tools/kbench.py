#!/usr/bin/env python3
"""Per-kernel microbenchmark on the BASELINE shapes: HIP-event time per launch, algorithmic GB/s and
fraction of the 8 TB/s HBM roofline.  Calls the C ABI directly (no autograd overhead).

    python tools/kbench.py [--iters 50] [--filter corr] [--levels uflow|pwclite]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from arflow_amd import _lib  # noqa: E402
from bench import algorithmic_bytes, HBM_PEAK_GBS  # noqa: E402


def p(t):
    return None if t is None else t.data_ptr()


MANIFEST = []  # (calls in the segment) per timed op, in launch order: tools/pmc_calls.py cuts the counter trace with it
_marker = None


def timeit(fn, iters):
    st = torch.cuda.current_stream()
    if _marker is not None:
        _marker(len(MANIFEST))  # one af_marker_kernel dispatch in front of every op's segment
    MANIFEST.append({'name': None, 'shape': None, 'calls': 5 + iters})
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    if os.environ.get('ARFLOW_LIB_PATH'):  # an alternative build of the library (A/B timing; tools only)
        _lib.LIB_PATH = os.environ['ARFLOW_LIB_PATH']
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--filter', default='')
    ap.add_argument('--levels', default='uflow')
    ap.add_argument('--batch', type=int, default=16, help='model-side batch (2B: both directions stacked)')
    ap.add_argument('--size', type=int, nargs=2, default=[384, 640])
    ap.add_argument('--manifest', default='', help='write the ordered (name, shape, calls) list of the timed ops here')
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device('cuda')
    s = torch.cuda.current_stream().cuda_stream
    global _marker
    _marker = lambda tag: lib.arflow_profile_marker(int(tag), s)
    H0, W0 = args.size
    B2 = args.batch
    if args.levels == 'uflow':
        levels = [(32, H0 // 32, W0 // 32), (32, H0 // 16, W0 // 16), (32, H0 // 8, W0 // 8), (32, H0 // 4, W0 // 4)]
    else:
        levels = [(192, H0 // 64, W0 // 64), (128, H0 // 32, W0 // 32), (96, H0 // 16, W0 // 16),
                  (64, H0 // 8, W0 // 8), (32, H0 // 4, W0 // 4)]
    g = torch.Generator(device='cuda').manual_seed(0)
    rows = []

    def rec(name, shape, us):
        nb = algorithmic_bytes(name, shape)
        gbs = nb / us / 1e3
        rows.append((name, shape, us, gbs))
        MANIFEST[-1].update(name=name, shape=list(shape), us=us)  # the segment timeit() has just run
        print('%-22s %-26s %9.1f us %9.1f GB/s  %5.1f%% of HBM peak' % (name, list(shape), us, gbs, 100 * gbs / HBM_PEAK_GBS), flush=True)

    def want(n):
        return any(f in n for f in args.filter.split('|'))

    for C, h, w in levels:
        x1 = torch.randn(B2, C, h, w, device=dev, generator=g)
        x2 = torch.randn(B2, C, h, w, device=dev, generator=g)
        out = torch.empty(B2, 81, h, w, device=dev)
        go = torch.randn(B2, 81, h, w, device=dev, generator=g)
        g1, g2 = torch.empty_like(x1), torch.empty_like(x2)
        fl = 2.0 * torch.randn(B2, 2, h, w, device=dev, generator=g)
        wout = torch.empty_like(x1)
        gfl = torch.empty_like(fl)
        planes = lib.arflow_corr_sign_planes(C, w, 4)
        sign = torch.zeros(B2, planes, h, w, device=dev, dtype=torch.int32) if planes else None
        if want('corr_fwd') or want('corr_bwd'):
            rec('arflow_corr_fwd', (B2, C, h, w, 4, planes), timeit(lambda: lib.arflow_corr_fwd(p(x1), p(x2), p(out), p(sign), B2, C, h, w, 4, 0.1, s), args.iters))
        if want('corr_bwd'):
            rec('arflow_corr_bwd', (B2, C, h, w, 4, planes or 81), timeit(lambda: lib.arflow_corr_bwd(p(go), None if planes else p(out), p(sign), p(x1), p(x2), p(g1), p(g2), B2, C, h, w, 4, 0.1, s), args.iters))
        if want('warp_fwd'):
            rec('arflow_warp_fwd', (B2, C, h, w), timeit(lambda: lib.arflow_warp_fwd(p(x2), p(fl), p(wout), None, B2, C, h, w, h, w, 2 * h * w, 0, 1, 0, s), args.iters))
        if want('warp_bwd'):
            rec('arflow_warp_bwd', (B2, C, h, w, True), timeit(lambda: lib.arflow_warp_bwd(p(x1), p(x2), p(fl), p(g2), p(gfl), B2, C, h, w, h, w, 2 * h * w, 0, 1, 0, s), args.iters))
        if want('featnorm'):
            n = C * h * w
            acc = torch.empty(4 * (2048 + B2), device=dev, dtype=torch.float64)
            stt = torch.empty(B2, 4, device=dev)
            rec('arflow_featnorm_fwd', (B2, n), timeit(lambda: lib.arflow_featnorm_fwd(p(x1), p(x2), p(g1), p(g2), p(acc), p(stt), B2, n, 0, s), args.iters))
            rec('arflow_featnorm_bwd', (B2, n), timeit(lambda: lib.arflow_featnorm_bwd(p(x1), p(x2), p(x1), p(x2), p(stt), p(acc), p(g1), p(g2), B2, n, 0, s), args.iters))
        if want('level'):
            has_flow = (h, w) != (levels[0][1], levels[0][2])
            fc = 0.7 * torch.randn(B2, 2, h // 2, w // 2, device=dev, generator=g) if has_flow else None
            ctot = 81 + C + 2 + 32
            buf = torch.zeros(B2, ctot, h, w, device=dev)
            gbuf = torch.randn(B2, ctot, h, w, device=dev, generator=g)
            bs = ctot * h * w
            fup, x2w = torch.empty(B2, 2, h, w, device=dev), torch.empty_like(x2)
            lsign = torch.zeros(B2, 3, h, w, device=dev, dtype=torch.int32)
            lstats = torch.empty(B2, 4, device=dev)
            lacc = torch.empty(4 * B2 * lib.arflow_level_acc_rows(B2, C, h, w, int(has_flow)), device=dev, dtype=torch.float64)
            ws = torch.empty(lib.arflow_level_bwd_ws_bytes(B2, C, h, w), device=dev, dtype=torch.uint8)
            gfc = torch.empty(B2, 2, h // 2, w // 2, device=dev)
            gext = torch.randn(B2, 2, h, w, device=dev, generator=g)
            vol, x1n, fslot = buf[:, :81], buf[:, 81:81 + C], buf[:, 81 + C:]
            fk = 2 if has_flow else 0

            def lfwd():
                lib.arflow_level_fwd(p(x1), p(x2), p(fc), 2 * (h // 2) * (w // 2), int(has_flow), 1, p(fup) if has_flow else None,
                                     fslot.data_ptr() if has_flow else None, bs, p(x2w) if has_flow else None, 0, vol.data_ptr(), bs,
                                     x1n.data_ptr(), bs, p(lsign), p(lstats), p(lacc), B2, C, h, w, 4, 0.1, 0, 1, 0, s)

            def lbwd():
                lib.arflow_level_bwd(gbuf[:, :81].data_ptr(), bs, p(lsign), x1n.data_ptr(), bs, gbuf[:, 81:].data_ptr(), bs, p(x1),
                                     p(x2), p(x2w) if has_flow else None, p(fup) if has_flow else None, 2 * h * w,
                                     gbuf[:, 81 + C:].data_ptr() if has_flow else None, bs, p(gext) if has_flow else None,
                                     p(lstats), 0, p(g1), p(g2), p(gfc) if has_flow else None, int(has_flow), 1, p(ws), B2, C, h,
                                     w, 4, 0.1, 0, 1, 0, s)
            # the form the models call (arflow_level_fwd_m): the maps' moments come from the conv epilogue that produced them
            mrows = lib.arflow_bias_act_mom_rows(C, h * w)
            bias0 = torch.zeros(C, device=dev)
            r1 = torch.empty(B2, mrows, 2, device=dev, dtype=torch.float64)
            r2 = torch.empty(B2, mrows, 2, device=dev, dtype=torch.float64)
            lib.arflow_bias_act_fwd_mom(p(x1), p(bias0), p(x1), p(r1), B2, C, h * w, 1.0, s)  # slope 1, bias 0: x unchanged
            lib.arflow_bias_act_fwd_mom(p(x2), p(bias0), p(x2), p(r2), B2, C, h * w, 1.0, s)

            def lfwd_m():
                lib.arflow_level_fwd_m(p(x1), p(x2), p(fc), 2 * (h // 2) * (w // 2), int(has_flow), 1, p(fup) if has_flow else None,
                                       fslot.data_ptr() if has_flow else None, bs, p(x2w) if has_flow else None, 0, vol.data_ptr(), bs,
                                       x1n.data_ptr(), bs, p(lsign), p(lstats), p(lacc), p(r1), mrows, None if has_flow else p(r2),
                                       0 if has_flow else mrows, B2, C, h, w, 4, 0.1, 0, 1, 0, s)
            rec('arflow_level_fwd', (B2, C, h, w, 4, 3, fk), timeit(lfwd, args.iters))
            rec('arflow_level_fwd_m', (B2, C, h, w, 4, 3, fk, 1), timeit(lfwd_m, args.iters))
            rec('arflow_level_bwd', (B2, C, h, w, 4, 3, fk), timeit(lbwd, args.iters))
    # loss side: B = batch/2 image pairs at full resolution, per direction
    B = max(1, B2 // 2)
    im1 = torch.rand(B, 3, H0, W0, device=dev, generator=g)
    im2 = torch.rand(B, 3, H0, W0, device=dev, generator=g)
    mask = torch.ones(B, 1, H0, W0, device=dev)
    fl0 = 2.0 * torch.randn(B, 2, H0, W0, device=dev, generator=g)
    fl2 = 1.0 * torch.randn(B, 2, H0 // 4, W0 // 4, device=dev, generator=g)
    rec3 = torch.empty_like(im1)
    dham = torch.empty(B, 1, H0, W0, device=dev)
    sums = torch.empty(4 * lib.arflow_sums_rows(B, H0, W0), device=dev)  # rows of the largest problem timed below
    gfl0 = torch.empty_like(fl0)
    sm = torch.empty(B, 3, H0 // 4, W0 // 4, device=dev)
    coef = torch.ones(2, device=dev)
    gfl2 = torch.empty_like(fl2)
    one = torch.ones(1, device=dev)
    if want('warp_fwd'):
        rec('arflow_warp_fwd', (B, 3, H0, W0), timeit(lambda: lib.arflow_warp_fwd(p(im2), p(fl0), p(rec3), None, B, 3, H0, W0, H0, W0, 2 * H0 * W0, 0, 1, 1, s), args.iters))
    if want('warp_bwd'):
        rec('arflow_warp_bwd', (B, 3, H0, W0, False), timeit(lambda: lib.arflow_warp_bwd(p(im1), p(im2), p(fl0), None, p(gfl0), B, 3, H0, W0, H0, W0, 2 * H0 * W0, 0, 1, 1, s), args.iters))
    if want('census_fwd'):
        rec('arflow_census_fwd', (B, H0, W0), timeit(lambda: lib.arflow_census_fwd(p(im1), p(im2), p(mask), None, p(dham), p(sums), B, H0, W0, 3, s), args.iters))
    if want('census_bwd'):
        rec('arflow_census_bwd', (B, H0, W0), timeit(lambda: lib.arflow_census_bwd(p(im1), p(im2), p(dham), p(one), p(rec3), B, H0, W0, 3, s), args.iters))
    if want('census_warp'):
        occ = torch.rand(B, 1, H0 // 4, W0 // 4, device=dev, generator=g) * 1.5
        maskw = torch.empty(B, 1, H0, W0, device=dev)
        gr1, gr2 = torch.empty(B, 1, H0, W0, device=dev), torch.empty(B, 1, H0, W0, device=dev)
        rec('arflow_down4_gray', (B, H0, W0), timeit(lambda: lib.arflow_down4_gray(p(im1), p(sm), p(gr1), B, H0, W0, s), args.iters))
        _marker(len(MANIFEST))  # an untimed launch: its own (nameless) segment
        MANIFEST.append({'name': None, 'shape': None, 'calls': 1})
        lib.arflow_down4_gray(p(im2), None, p(gr2), B, H0, W0, s)
        # a smooth flow (x4 bilinear upsample of a 2 px field, what the models emit) next to the white-noise one
        fls = torch.nn.functional.interpolate(fl2 * 2, scale_factor=4, mode='bilinear', align_corners=False).contiguous()
        rec('arflow_census_warp_fwd', (B, H0, W0, 'smooth'), timeit(lambda: lib.arflow_census_warp_fwd(p(gr1), p(gr2), p(fls), 2 * H0 * W0, p(occ), p(maskw), p(dham), p(sums), B, H0, W0, 3, s), args.iters))
        rec('arflow_census_warp_bwd', (B, H0, W0, 'smooth'), timeit(lambda: lib.arflow_census_warp_bwd(p(gr1), p(gr2), p(fls), 2 * H0 * W0, p(dham), p(one), p(gfl0), B, H0, W0, 3, s), args.iters))
        rec('arflow_census_warp_fwd', (B, H0, W0), timeit(lambda: lib.arflow_census_warp_fwd(p(gr1), p(gr2), p(fl0), 2 * H0 * W0, p(occ), p(maskw), p(dham), p(sums), B, H0, W0, 3, s), args.iters))
        rec('arflow_census_warp_bwd', (B, H0, W0), timeit(lambda: lib.arflow_census_warp_bwd(p(gr1), p(gr2), p(fl0), 2 * H0 * W0, p(dham), p(one), p(gfl0), B, H0, W0, 3, s), args.iters))
    if want('pair'):  # UFlowLoss as bench.py runs it: both directions as ONE batch of 2B = --batch samples (sample s = 2 b + direction)
        B2p = B2
        imgs = torch.rand(B2p, 3, H0, W0, device=dev, generator=g)
        smallp = torch.empty(B2p, 3, H0 // 4, W0 // 4, device=dev)
        grayp = torch.empty(B2p, 1, H0, W0, device=dev)
        occp = torch.empty(B2p, 1, H0 // 4, W0 // 4, device=dev)
        fl2p = 1.0 * torch.randn(B2p, 2, H0 // 4, W0 // 4, device=dev, generator=g)
        fl0p = torch.nn.functional.interpolate(fl2p * 4, scale_factor=4, mode='bilinear', align_corners=False).contiguous()
        dhamp = torch.empty(B2p, 1, H0, W0, device=dev)
        maskp = torch.empty(B2p, 1, H0, W0, device=dev)
        sump = torch.empty(4 * lib.arflow_sums_rows(B2p, H0, W0), device=dev)
        sums2 = torch.empty(4 * lib.arflow_sums_rows(B2p, H0 // 4, W0 // 4), device=dev)
        gf0p, gf2p = torch.empty_like(fl0p), torch.empty_like(fl2p)
        sc2, cf2 = torch.ones(2, device=dev), torch.ones(2, device=dev)
        h4, w4 = H0 // 4, W0 // 4
        rec('arflow_down4_gray_z', (B2p, H0, W0), timeit(lambda: lib.arflow_down4_gray_z(p(imgs), p(smallp), p(grayp), p(occp), B2p, H0, W0, s), args.iters))
        # (the range map keeps accumulating over the timed calls -- it is cleared by arflow_down4_gray_z in a step; same work)
        rec('arflow_splat_smooth_fwd', (B2p, 3, h4, w4), timeit(lambda: lib.arflow_splat_smooth_fwd(p(fl2p), p(smallp), p(occp), p(sums2), B2p, h4, w4, 2 * h4 * w4, 1.0, 150.0, 1, 1, 1, 1, s), args.iters))
        rec('arflow_census_warp_pair_fwd', (B2p, H0, W0), timeit(lambda: lib.arflow_census_warp_pair_fwd(p(grayp), p(fl0p), 2 * H0 * W0, p(occp), p(maskp), p(dhamp), p(sump), B2p, H0, W0, 3, s), args.iters))
        rec('arflow_uflow_pair_bwd', (B2p, H0, W0), timeit(lambda: lib.arflow_uflow_pair_bwd(p(grayp), p(fl0p), 2 * H0 * W0, p(dhamp), p(sc2), p(gf0p), B2p, H0, W0, 3, p(fl2p), 2 * h4 * w4, p(smallp), p(cf2), p(gf2p), h4, w4, 1.0, 150.0, 1, 1, 1, s), args.iters))
    if want('photo_fwd'):
        rec('arflow_photo_fwd', (B, 3, H0, W0), timeit(lambda: lib.arflow_photo_fwd(p(im1), p(im2), p(mask), None, p(sums), B, 3, H0, W0, s), args.iters))
    if want('photo_bwd'):
        rec('arflow_photo_bwd', (B, 3, H0, W0), timeit(lambda: lib.arflow_photo_bwd(p(im1), p(im2), p(mask), None, p(coef), p(rec3), B, 3, H0, W0, s), args.iters))
    if want('splat'):
        rec('arflow_splat_map', (B, H0 // 4, W0 // 4), timeit(lambda: lib.arflow_splat_map(p(fl2), p(dham), B, H0 // 4, W0 // 4, 2 * (H0 // 4) * (W0 // 4), 0, s), args.iters))
    if args.levels == 'pwclite':  # unFlowLoss works at full resolution
        if want('smooth_fwd'):
            rec('arflow_smooth_fwd', (B, 3, H0, W0), timeit(lambda: lib.arflow_smooth_fwd(p(fl0), p(im1), p(sums), B, 3, H0, W0, 2 * H0 * W0, 1.0 / 384, 10.0, 1, 0, 0, s), args.iters))
        if want('smooth_bwd'):
            rec('arflow_smooth_bwd', (B, 3, H0, W0), timeit(lambda: lib.arflow_smooth_bwd(p(fl0), p(im1), p(coef), p(gfl0), B, 3, H0, W0, 2 * H0 * W0, 1.0 / 384, 10.0, 1, 0, 0, s), args.iters))
        if want('splat_map'):
            rec('arflow_splat_map', (B, H0, W0), timeit(lambda: lib.arflow_splat_map(p(fl0), p(dham), B, H0, W0, 2 * H0 * W0, 1, s), args.iters))
    if want('smooth_fwd'):
        rec('arflow_smooth_fwd', (B, 3, H0 // 4, W0 // 4), timeit(lambda: lib.arflow_smooth_fwd(p(fl2), p(sm), p(sums), B, 3, H0 // 4, W0 // 4, 2 * (H0 // 4) * (W0 // 4), 1.0, 150.0, 1, 1, 1, s), args.iters))
    if want('smooth_bwd'):
        rec('arflow_smooth_bwd', (B, 3, H0 // 4, W0 // 4), timeit(lambda: lib.arflow_smooth_bwd(p(fl2), p(sm), p(coef), p(gfl2), B, 3, H0 // 4, W0 // 4, 2 * (H0 // 4) * (W0 // 4), 1.0, 150.0, 1, 1, 1, s), args.iters))
    if want('down4'):
        rec('arflow_down4', (B * 3, H0, W0), timeit(lambda: lib.arflow_down4(p(im1), p(sm), B * 3, H0, W0, s), args.iters))
    if want('up4'):
        rec('arflow_up4_clamp_mul', (B, H0 // 4, W0 // 4), timeit(lambda: lib.arflow_up4_clamp_mul(p(sm), p(mask), p(dham), B, H0 // 4, W0 // 4, s), args.iters))
    if args.manifest:
        json.dump(MANIFEST, open(args.manifest, 'w'), indent=1)
    tot_us = sum(r[2] for r in rows)
    tot_b = sum(algorithmic_bytes(r[0], r[1]) for r in rows)
    print(json.dumps({'kernels': len(rows), 'sum_us': tot_us, 'sum_GB': tot_b / 1e9, 'aggregate_GBps': tot_b / tot_us / 1e3,
                      'aggregate_frac_of_hbm_peak': tot_b / tot_us / 1e3 / HBM_PEAK_GBS}))


if __name__ == '__main__':
    main()
