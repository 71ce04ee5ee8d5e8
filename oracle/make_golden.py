"""Generate tests/golden/*.npz by running the REFERENCE (deu439/ARFlow) on CPU.

Run only in the build container, where the reference is mounted read-only at /root/reference:

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [--only ops|losses|models]

Each fixture stores the inputs, the reference outputs and the reference autograd gradients for
one hot-path function, so that tests on the GPU box (where the reference does not exist) can pin
both the CPU oracle (oracle/ops.py, oracle/losses.py) and the HIP kernels.  Nothing from the
reference's source text is stored -- fixtures are arrays only.

Inputs: the analytic ``field`` of SURVEY Appendix A (no RNG) plus seeded-random ragged cases; the
random inputs are saved in the fixture, so a different RNG stream elsewhere does not matter.
"""
import argparse
import os
import sys

import numpy as np
import torch

REF = os.environ.get('ARFLOW_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
# the script's own directory must not shadow the reference's top-level packages (losses/, models/)
sys.path[:] = [p for p in sys.path if os.path.abspath(p or '.') != HERE]
sys.path.insert(0, os.path.dirname(HERE))
from oracle.fixture_common import Cfg, fill_deterministic, loss_cfgs, pool_to_quarter, synth_pair  # noqa: E402


def field(B, C, H, W, a, b, c, d):
    bb = torch.arange(B, dtype=torch.float64).view(B, 1, 1, 1)
    ch = torch.arange(C, dtype=torch.float64).view(1, C, 1, 1)
    y = torch.arange(H, dtype=torch.float64).view(1, 1, H, 1)
    x = torch.arange(W, dtype=torch.float64).view(1, 1, 1, W)
    return torch.sin(a * (bb + 1) + b * (ch + 1) + c * y + d * x).float()


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **{k: (npy(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def grads(out, inputs, gout):
    return torch.autograd.grad(out, inputs, gout, allow_unused=True)


def analytic_inputs():
    B, C, H, W = 2, 8, 12, 20
    return dict(
        x1=field(B, C, H, W, .3, .7, .37, .11), x2=field(B, C, H, W, .5, .2, .13, .29),
        flow=3 * field(B, 2, H, W, .9, 1.3, .21, .17),
        g81=field(B, 81, H, W, .1, .05, .2, .3), gC=field(B, C, H, W, .2, .1, .3, .4),
        im1=(field(B, 3, H, W, .3, .7, .37, .11) + 1) / 2, im2=(field(B, 3, H, W, .5, .2, .13, .29) + 1) / 2)


def gen_ops():
    from models.correlation_native import Correlation
    from utils import warp_utils, uflow_utils, uflow_resampler
    from losses import loss_blocks

    A = analytic_inputs()
    rng = torch.Generator().manual_seed(1234)

    # ---- correlation: analytic + ragged random, several displacements -------------------
    cases = [('analytic', A['x1'], A['x2'], A['g81'], 4)]
    for k, (B, C, H, W, d) in enumerate([(1, 5, 7, 9, 4), (2, 3, 10, 13, 4), (1, 33, 9, 6, 4),
                                         (1, 4, 6, 11, 2), (2, 2, 5, 8, 1), (1, 6, 8, 8, 3)]):
        n = (2 * d + 1) ** 2
        cases.append(('rand%d' % k, torch.randn(B, C, H, W, generator=rng), torch.randn(B, C, H, W, generator=rng),
                      torch.randn(B, n, H, W, generator=rng), d))
    out = {}
    for name, x1, x2, g, d in cases:
        x1 = x1.clone().requires_grad_(True)
        x2 = x2.clone().requires_grad_(True)
        y = Correlation(max_displacement=d)(x1, x2)
        g1, g2 = grads(y, [x1, x2], g)
        out.update({name + '_x1': x1, name + '_x2': x2, name + '_g': g, name + '_d': d,
                    name + '_y': y, name + '_gx1': g1, name + '_gx2': g2})
    out['names'] = np.array([c[0] for c in cases])
    save('corr', **out)

    # ---- warp: flow_warp (pad x align_corners), resample, resampler ---------------------
    out = {}
    wcases = [('analytic', A['x2'], A['flow'], A['gC'])]
    for k, (B, C, H, W, s) in enumerate([(1, 3, 7, 9, 2.5), (2, 5, 10, 13, 4.0), (1, 2, 16, 5, 6.0)]):
        wcases.append(('rand%d' % k, torch.randn(B, C, H, W, generator=rng),
                       s * torch.randn(B, 2, H, W, generator=rng), torch.randn(B, C, H, W, generator=rng)))
    # exact-integer and exact-border coordinates exercise floor / clamp-gradient corner cases
    fl = torch.zeros(1, 2, 6, 8)
    fl[0, 0, :, :4] = 1.0
    fl[0, 0, :, 4:] = -7.0
    fl[0, 1, :3] = 2.0
    fl[0, 1, 3:] = 5.0
    wcases.append(('integer', torch.randn(1, 4, 6, 8, generator=rng), fl, torch.randn(1, 4, 6, 8, generator=rng)))
    for name, x, flow, g in wcases:
        out.update({name + '_x': x, name + '_flow': flow, name + '_g': g})
        for pad in ('zeros', 'border'):
            for ac in (True, False):
                xx = x.clone().requires_grad_(True)
                ff = flow.clone().requires_grad_(True)
                y = warp_utils.flow_warp(xx, ff, pad=pad, align_corners=ac)
                gx, gf = grads(y, [xx, ff], g)
                tag = '%s_%s_%d' % (name, pad, int(ac))
                out.update({tag + '_y': y, tag + '_gx': gx, tag + '_gf': gf})
        xx = x.clone().requires_grad_(True)
        ff = flow.clone().requires_grad_(True)
        coords = uflow_utils.flow_to_warp(ff)
        y = uflow_utils.resample(xx, coords)
        gx, gf = grads(y, [xx, ff], g)
        out.update({name + '_resample_y': y, name + '_resample_gx': gx, name + '_resample_gf': gf,
                    name + '_coords': coords, name + '_mask_invalid': uflow_utils.mask_invalid(coords)})
        out[name + '_resampler_nhwc'] = uflow_resampler.resampler(
            x.permute(0, 2, 3, 1).contiguous(), coords.detach().permute(0, 2, 3, 1).contiguous())
    out['names'] = np.array([c[0] for c in wcases])
    save('warp', **out)

    # ---- splat maps and masks -------------------------------------------------------------
    out = {}
    mcases = [('analytic', A['flow'])]
    for k, (B, H, W, s) in enumerate([(1, 7, 9, 2.0), (2, 12, 10, 5.0)]):
        mcases.append(('rand%d' % k, s * torch.randn(B, 2, H, W, generator=rng)))
    mcases.append(('integer', fl))
    for name, flow in mcases:
        out[name + '_flow'] = flow
        out[name + '_range_map'] = uflow_utils.compute_range_map(flow)
        out[name + '_range_map_wu'] = warp_utils.compute_range_map(flow)
        B, _, H, W = flow.shape
        base = warp_utils.mesh_grid(B, H, W).type_as(flow)
        out[name + '_corr_map'] = warp_utils.get_corresponding_map(base + flow)
        out[name + '_occ_back_02'] = warp_utils.get_occu_mask_backward(flow, th=0.2)
        out[name + '_occ_back_0'] = warp_utils.get_occu_mask_backward(flow, th=0.0)
        out[name + '_occ_bidir'] = warp_utils.get_occu_mask_bidirection(flow, -0.7 * flow.flip(-1))
        out[name + '_occ_bidir_neg'] = warp_utils.get_occu_mask_bidirection(flow, -flow)
        out[name + '_border_mask'] = warp_utils.border_mask(flow)
    out['names'] = np.array([c[0] for c in mcases])
    save('masks', **out)

    # ---- photometric / smoothness blocks --------------------------------------------------
    out = {}
    pcases = [('analytic', A['im1'], A['im2'], A['flow'])]
    for k, (B, H, W) in enumerate([(1, 9, 11), (2, 16, 14)]):
        pcases.append(('rand%d' % k, torch.rand(B, 3, H, W, generator=rng), torch.rand(B, 3, H, W, generator=rng),
                       2 * torch.randn(B, 2, H, W, generator=rng)))
    for name, im1, im2, flow in pcases:
        B, _, H, W = im1.shape
        mask = (torch.rand(B, 1, H, W, generator=rng) > 0.3).float() * torch.rand(B, 1, H, W, generator=rng)
        out.update({name + '_im1': im1, name + '_im2': im2, name + '_flow': flow, name + '_mask': mask})
        a = im1.clone().requires_grad_(True)
        b = im2.clone().requires_grad_(True)
        y = loss_blocks.SSIM(a, b)
        g = torch.rand(y.shape, generator=rng)
        ga, gb = grads(y, [a, b], g)
        out.update({name + '_ssim': y, name + '_ssim_g': g, name + '_ssim_ga': ga, name + '_ssim_gb': gb})
        for md, sd in ((1, False), (3, True)):
            a = im1.clone().requires_grad_(True)
            b = im2.clone().requires_grad_(True)
            dist, tm = loss_blocks.TernaryLoss(a, b, max_distance=md, sum_dist=sd)
            g = torch.rand(dist.shape, generator=rng)
            ga, gb = grads(dist, [a, b], g)
            tag = '%s_ternary_%d_%d' % (name, md, int(sd))
            out.update({tag + '_dist': dist, tag + '_mask': tm, tag + '_g': g, tag + '_ga': ga, tag + '_gb': gb})
        for ps in (7, 3):
            b = im2.clone().requires_grad_(True)
            y = uflow_utils.census_loss(im1, b, mask, patch_size=ps)
            gb, = grads(y, [b], torch.ones(()))
            out.update({'%s_census_%d' % (name, ps): y, '%s_census_%d_gb' % (name, ps): gb})
        y = uflow_utils.census_loss(im1, im2, torch.ones_like(mask))
        out[name + '_census_ones'] = y
        for fn, key in ((lambda f: loss_blocks.smooth_grad_1st(f, im1, 10.), 'sm1_abs'),
                        (lambda f: loss_blocks.smooth_grad_1st(f, im1, 10., penalty='uflow'), 'sm1_uflow'),
                        (lambda f: loss_blocks.smooth_grad_2nd(f, im1, 10.), 'sm2')):
            f = flow.clone().requires_grad_(True)
            y = fn(f)
            gf, = grads(y, [f], torch.ones(()))
            out.update({'%s_%s' % (name, key): y, '%s_%s_gf' % (name, key): gf})
    out['names'] = np.array([c[0] for c in pcases])
    save('photo', **out)

    # ---- resize helpers + feature normalisation -------------------------------------------
    import models.pwclite_uflow as pu
    import models.uflow_model as um
    out = {}
    img = torch.rand(2, 3, 16, 24, generator=rng)
    m = torch.rand(2, 1, 4, 6, generator=rng)
    out.update({'img': img, 'm': m, 'down4': uflow_utils.downsample(img, False, 4.0),
                'up4': uflow_utils.upsample(m, False, 4.0), 'up2_flow': uflow_utils.upsample(m, True),
                'down2_flow': uflow_utils.downsample(img, True)})
    f1 = torch.randn(2, 6, 5, 7, generator=rng) * 2 + 0.5
    f2 = torch.randn(2, 6, 5, 7, generator=rng) * 0.5 - 1.0
    a, b = pu.normalize_features([f1, f2])
    c, d = um.normalize_features([f1, f2], True, True, True, True)
    out.update({'f1': f1, 'f2': f2, 'nj_1': a, 'nj_2': b, 'nu_1': c, 'nu_2': d})
    save('aux', **out)


def gen_losses():
    from losses.uflow_loss import UFlowLoss
    from losses.flow_loss import unFlowLoss
    from losses.fullres_loss import FullResLoss
    rng = torch.Generator().manual_seed(4321)
    img, flows = synth_pair(2, 64, 96, rng)
    out = {'img': img}
    for i, f in enumerate(flows):
        out['flow%d' % i] = f
    uflow, unflow, full = loss_cfgs()
    for group, cls in ((uflow, UFlowLoss), (unflow, unFlowLoss), (full, FullResLoss)):
        for name, cfg in group:
            fl = [f.clone().requires_grad_(True) for f in flows]
            res = cls(cfg)(fl, img)
            g = torch.autograd.grad(res[0], fl, allow_unused=True)
            out[name + '_total'] = res[0]
            out[name + '_warp'] = res[1]
            out[name + '_smooth'] = res[2]
            out[name + '_absflow'] = res[3]
            if len(res) > 4:
                out[name + '_mask1'] = res[4]
            for i, gi in enumerate(g):
                out['%s_g%d' % (name, i)] = gi if gi is not None else torch.zeros_like(flows[i])
    save('losses', **out)


def gen_models():
    import models.pwclite as mp
    import models.pwclite_uflow as mpu
    import models.uflow_model as mum
    from losses.uflow_loss import UFlowLoss
    from losses.flow_loss import unFlowLoss
    rng = torch.Generator().manual_seed(777)
    uflow, unflow, _ = loss_cfgs()

    def run(tag, model, x, with_bk, loss=None):
        fill_deterministic(model)
        model.eval()
        res = model(x, with_bk=with_bk)
        out = {tag + '_nparams': sum(p.numel() for p in model.parameters()),
               tag + '_keys': np.array(list(model.state_dict().keys()))}
        for k in ('flows_fw', 'flows_bw'):
            if k in res:
                for i, f in enumerate(res[k]):
                    out['%s_%s_%d' % (tag, k, i)] = pool_to_quarter(f, x.shape[2])
        if loss is not None:
            flows = [torch.cat([a, b], 1) for a, b in zip(res['flows_fw'], res['flows_bw'])]
            lres = loss(flows, x)
            lres[0].backward()
            out[tag + '_loss'] = lres[0]
            # gradient fingerprints: per-parameter sum and abs-sum
            names, gs, ga = [], [], []
            for n, p in model.named_parameters():
                names.append(n)
                gs.append(float(p.grad.double().sum()) if p.grad is not None else 0.0)
                ga.append(float(p.grad.double().abs().sum()) if p.grad is not None else 0.0)
            out[tag + '_gnames'] = np.array(names)
            out[tag + '_gsum'] = np.array(gs)
            out[tag + '_gabs'] = np.array(ga)
        return out

    x2 = synth_pair(1, 192, 256, rng)[0]
    x3 = torch.cat([x2, synth_pair(1, 192, 256, rng)[0][:, :3]], 1)
    out = {'x3': (x3 * 255).round().to(torch.uint8)}  # 8-bit images: x = uint8/255; x2 = x3[:, :6]
    x3 = out['x3'].float() / 255
    x2 = x3[:, :6].contiguous()
    cfg_unflow6 = Cfg(dict(unflow[0][1]))
    cfg_unflow6['w_scales'] = [1.0, 1.0, 1.0, 1.0, 1.0, 0.0]
    cfg_unflow6['w_sm_scales'] = [1.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    out.update(run('pwclite2', mp.PWCLite(Cfg(upsample=True, n_frames=2, reduce_dense=True)), x2, True,
                   unFlowLoss(cfg_unflow6)))
    out.update(run('pwclite2_dense', mp.PWCLite(Cfg(upsample=True, n_frames=2, reduce_dense=False)), x2, False))
    out.update(run('pwclite3', mp.PWCLite(Cfg(upsample=True, n_frames=3, reduce_dense=True)), x3, True))
    for fn, ac, pad in ((True, True, 'zeros'), (False, False, 'border')):
        cfg = Cfg(level_dropout=0.0, feature_norm=fn, align_corners=ac, warp_pad=pad, n_frames=2, reduce_dense=False)
        out.update(run('pwclite_uflow_%d' % int(fn), mpu.PWCLiteUflow(cfg), x2, True,
                       UFlowLoss(uflow[0][1]) if fn else None))
    out.update(run('pwcflow', mum.PWCFlow(Cfg(level_dropout=0.0, feature_norm=True)), x2, True, UFlowLoss(uflow[1][1])))
    save('models', **out)


def gen_models5():
    """The 5-frame sliding-window pass of PWCLite (models/pwclite.py:274-281: windows (0,1,2), (1,2,3) and, with_bk,
    (2,3,4)) through the REFERENCE model with deterministic weights -> tests/golden/models5.npz."""
    import models.pwclite as mp
    rng = torch.Generator().manual_seed(555)
    frames = torch.cat([synth_pair(1, 128, 192, rng)[0] for _ in range(3)], 1)[:, :15]
    u8 = (frames * 255).round().to(torch.uint8)
    x = u8.float() / 255
    # the window count comes from the INPUT (x.size(1) / 3, models/pwclite.py:261); the estimators are those of the 3-frame model
    model = fill_deterministic(mp.PWCLite(Cfg(upsample=True, n_frames=3, reduce_dense=True))).eval()
    res = model(x, with_bk=True)
    out = {'x5': u8}
    for k in ('flows_fw', 'flows_bw'):
        for w, flows in enumerate(res[k]):
            for i, f in enumerate(flows):
                out['pwclite5_%s_%d_%d' % (k, w, i)] = pool_to_quarter(f, x.shape[2])
    save('models5', **out)


def gen_examples():
    """BASELINE config 1: the reference's example images (examples/img{0,1,2}.png, 1242x375) resized to
    384x640 as the README prescribes, through the reference PWCLite (2- and 3-frame) with deterministic
    weights, on CPU.  The images are data files the reference ships; stored as uint8 arrays."""
    import models.pwclite as mp
    from PIL import Image
    frames = []
    for n in ('img0.png', 'img1.png', 'img2.png'):
        im = Image.open(os.path.join(REF, 'examples', n)).convert('RGB').resize((640, 384), Image.BILINEAR)
        frames.append(torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1))
    u8 = torch.cat(frames, 0).unsqueeze(0)  # [1,9,384,640] uint8
    x = u8.float() / 255
    out = {'frames_u8': u8}
    m2 = fill_deterministic(mp.PWCLite(Cfg(upsample=True, n_frames=2, reduce_dense=True))).eval()
    r2 = m2(x[:, 3:9], with_bk=True)
    m3 = fill_deterministic(mp.PWCLite(Cfg(upsample=True, n_frames=3, reduce_dense=True))).eval()
    r3 = m3(x)
    for tag, r in (('two', r2), ('three', r3)):
        for k in ('flows_fw', 'flows_bw'):
            out['%s_%s_full_pooled8' % (tag, k)] = torch.nn.functional.avg_pool2d(r[k][0], 8)
            out['%s_%s_quarter' % (tag, k)] = r[k][1] if r[k][1].shape[2] == 96 else torch.nn.functional.avg_pool2d(r[k][1], 2)
    save('examples', **out)


def gen_general():
    """The parameter values no shipped config uses but the reference's functions accept: flow_warp(mode='nearest' | 'bicubic')
    (utils/warp_utils.py:83-90), SSIM(md=2) (losses/loss_blocks.py:65-84), TernaryLoss(max_distance=4 / 5)
    (losses/loss_blocks.py:12-62) -- reference outputs and autograd gradients."""
    from utils import warp_utils
    from losses import loss_blocks
    rng = torch.Generator().manual_seed(4321)
    out = {}
    wcases = []
    for k, (B, C, H, W, sc) in enumerate([(2, 3, 9, 13, 3.0), (1, 5, 16, 12, 6.0)]):
        wcases.append(('w%d' % k, torch.randn(B, C, H, W, generator=rng), sc * torch.randn(B, 2, H, W, generator=rng),
                       torch.randn(B, C, H, W, generator=rng)))
    for name, x, flow, g in wcases:
        out.update({name + '_x': x, name + '_flow': flow, name + '_g': g})
        for pad in ('zeros', 'border'):
            for ac in (True, False):
                xx = x.clone().requires_grad_(True)
                y = warp_utils.flow_warp(xx, flow, pad=pad, mode='nearest', align_corners=ac)
                gx, = grads(y, [xx], g)
                tag = '%s_%s_%d' % (name, pad, int(ac))
                out.update({tag + '_y': y, tag + '_gx': gx})
                # mode='bicubic' (round 3): both gradients
                xx, ff = x.clone().requires_grad_(True), flow.clone().requires_grad_(True)
                y = warp_utils.flow_warp(xx, ff, pad=pad, mode='bicubic', align_corners=ac)
                gx, gf = grads(y, [xx, ff], g)
                out.update({tag + '_cub_y': y, tag + '_cub_gx': gx, tag + '_cub_gf': gf})
    out['wnames'] = np.array([c[0] for c in wcases])
    im1, im2 = torch.rand(2, 3, 18, 23, generator=rng), torch.rand(2, 3, 18, 23, generator=rng)
    out.update({'im1': im1, 'im2': im2})
    for md in (2, 3):
        a, b = im1.clone().requires_grad_(True), im2.clone().requires_grad_(True)
        y = loss_blocks.SSIM(a, b, md=md)
        g = torch.randn(y.shape, generator=rng)
        ga, gb = grads(y, [a, b], g)
        out.update({'ssim%d' % md: y, 'ssim%d_g' % md: g, 'ssim%d_ga' % md: ga, 'ssim%d_gb' % md: gb})
    for md, sd in ((4, True), (5, False)):
        a, b = im1.clone().requires_grad_(True), im2.clone().requires_grad_(True)
        dist, mask = loss_blocks.TernaryLoss(a, b, max_distance=md, sum_dist=sd)
        g = torch.randn(dist.shape, generator=rng)
        ga, gb = grads(dist, [a, b], g)
        tag = 'tern%d_%d' % (md, int(sd))
        out.update({tag + '_dist': dist, tag + '_mask': mask, tag + '_g': g, tag + '_ga': ga, tag + '_gb': gb})
    save('general', **out)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='all')
    args = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit('reference not mounted at %s: golden vectors can only be generated in the build container' % REF)
    sys.path.insert(0, REF)
    torch.set_num_threads(4)
    torch.manual_seed(0)
    with torch.enable_grad():
        if args.only in ('all', 'ops'):
            gen_ops()
        if args.only in ('all', 'losses'):
            gen_losses()
        if args.only in ('all', 'models'):
            gen_models()
        if args.only in ('all', 'models5'):
            gen_models5()
        if args.only in ('all', 'examples'):
            gen_examples()
        if args.only in ('all', 'general'):
            gen_general()
