"""Definitions shared by the fixture generator (oracle/make_golden.py) and the tests that replay the
fixtures: loss configurations, the synthetic image/flow pyramid, and the deterministic weight fill.
TEST INFRASTRUCTURE ONLY."""
import torch


class Cfg(dict):
    __getattr__ = dict.__getitem__


def loss_cfgs():
    uflow = [('uflow_o1', Cfg(type='uflow', edge_constant=150, w_smooth=4.0, w_census=1.0, with_bk=True, smooth_order=1)),
             ('uflow_o2', Cfg(type='uflow', edge_constant=150, w_smooth=2.0, w_census=1.0, with_bk=True, smooth_order=2)),
             ('uflow_nobk', Cfg(type='uflow', edge_constant=150, w_smooth=4.0, w_census=1.0, with_bk=False, smooth_order=1))]
    unflow = [('unflow_back', Cfg(type='unflow', w_l1=0.15, w_ssim=0.85, w_ternary=0.0, warp_pad='border', alpha=10,
                                   occ_from_back=True, with_bk=True, w_smooth=75.0,
                                   w_scales=[1.0, 1.0, 1.0, 1.0, 0.0], w_sm_scales=[1.0, 0.0, 0.0, 0.0, 0.0])),
              ('unflow_bidir_2nd', Cfg(type='unflow', w_l1=0.15, w_ssim=0.85, w_ternary=0.0, warp_pad='zeros', alpha=10,
                                        occ_from_back=False, with_bk=True, w_smooth=50.0, smooth_2nd=True,
                                        w_scales=[1.0, 0.5, 1.0, 1.0, 1.0], w_sm_scales=[1.0, 0.5, 0.0, 0.0, 0.0])),
              ('unflow_l1only_nobk', Cfg(type='unflow', w_l1=1.0, w_ssim=0.0, w_ternary=0.0, warp_pad='border', alpha=10,
                                          occ_from_back=True, with_bk=False, w_smooth=10.0,
                                          w_scales=[1.0, 1.0, 0.0, 0.0, 0.0], w_sm_scales=[1.0, 1.0, 0.0, 0.0, 0.0]))]
    full = [('fullres_wang', Cfg(type='fullres', w_l1=0.5, w_ssim=0.0, w_ternary=1.0, ternary_distance=3, warp_pad='zeros',
                                  align_corners=True, occ_type='wang', wang_thr=0.2, with_bk=True, alpha=10, w_smooth=4.0)),
            ('fullres_wang1', Cfg(type='fullres', w_l1=0.0, w_ssim=0.0, w_ternary=1.0, ternary_distance=3, warp_pad='border',
                                   align_corners=False, occ_type='wang1', with_bk=True, alpha=10, w_smooth=4.0)),
            ('fullres_brox', Cfg(type='fullres', w_l1=1.0, w_ssim=0.0, w_ternary=0.5, ternary_distance=1, warp_pad='zeros',
                                  align_corners=True, occ_type='brox', with_bk=True, alpha=10, w_smooth=2.0))]
    return uflow, unflow, full


def synth_pair(B, H, W, rng):
    """Smooth-ish image pair + 5-level flow pyramid (full, 1/2, 1/4, 1/8, 1/16)."""
    base = torch.rand(B, 6, H // 4, W // 4, generator=rng)
    img = torch.nn.functional.interpolate(base, (H, W), mode='bilinear', align_corners=False)
    img = (img + 0.15 * torch.rand(B, 6, H, W, generator=rng)).clamp(0, 1)
    flows = []
    for s in (1, 2, 4, 8, 16):
        f = torch.randn(B, 4, H // s, W // s, generator=rng) * (4.0 / s)
        flows.append(f)
    return img, flows


def fill_deterministic(model):
    """Deterministic, construction-order-independent weights: depends only on key name + shape.

    Used instead of seeded init so that the product's host model (different construction code)
    can be given bit-identical weights without shipping a 9-29 MB state_dict."""
    import zlib
    sd = model.state_dict()
    for key in sorted(sd.keys()):
        t = sd[key]
        n = t.numel()
        h = zlib.crc32(key.encode()) % 1000
        idx = torch.arange(n, dtype=torch.float64)
        if key.endswith('bias'):
            v = 0.02 * torch.sin(0.731 * idx + h)
        else:
            fan_in = t[0].numel() if t.dim() > 1 else 1
            v = torch.sin(0.37 * idx + 0.11 * h) * (1.7 / fan_in) ** 0.5
        sd[key] = v.float().view_as(t)
    model.load_state_dict(sd)
    return model


def pool_to_quarter(f, full_h):
    """Flows finer than 1/4 resolution are stored average-pooled to 1/4 to keep fixtures small (they
    are deterministic bilinear upsamples of the 1/4 flow)."""
    k = f.shape[2] * 4 // full_h
    return torch.nn.functional.avg_pool2d(f, k) if k > 1 else f

