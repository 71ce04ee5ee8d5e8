#!/usr/bin/env python3
"""Benchmark of the ARFlow hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--size H W] [--batch B]

One "step" = one full pass of the hot path over one batch of synthetic image pairs: model forward for
both flow directions (cost-volume correlation + bilinear warp at every pyramid level, HIP kernels),
the unsupervised loss (census / occlusion splat / smoothness, HIP kernels), backward through all of it,
gradient all-reduce (N > 1) and the Adam update.  Default workload = BASELINE.json configs[1]:
384x640 pairs, batch 8 per GPU (weak scaling), fp32.  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.

N > 1: either launched by  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 --master-port P bench.py --gpus N ...  (one rank per GPU, RCCL), or plainly as  python bench.py
--gpus N : it then starts those N ranks itself as child processes and relays rank 0's line.  With fewer
GPUs than ranks the ranks share devices and reduce over gloo (a rehearsal: "oversubscribed": true).
ARFLOW_FORCE_COLLECTIVES=1 runs the bucketed all-reduce path at world size 1 over RCCL.
ARFLOW_GLOBAL_LOSS_NORM=1 normalises the loss by the GLOBAL mask sums (one extra scalar all-reduce per
census/photometric term), the reference's gathered-batch semantics (trainer/uflow_trainer.py:48-54).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



def _use_shipped_miopen_db():
    """Point MIOpen at the tuning records shipped in arflow_amd/miopen_db (found on an MI355X with
    `--miopen-find`).  The convolutions of the host model are MIOpen's (outside the hot
    path); without the records MIOpen's immediate mode falls back to its heuristic solver choice, which is
    ~5 % slower on this network.  The records are copied to a private per-rank directory because MIOpen
    opens its user database read-write.  Set ARFLOW_MIOPEN_DB=off to skip."""
    if os.environ.get('ARFLOW_MIOPEN_DB', 'on') == 'off' or 'MIOPEN_USER_DB_PATH' in os.environ:
        return
    import shutil
    import tempfile
    src = os.path.join(ROOT, 'arflow_amd', 'miopen_db')
    if not os.path.isdir(src):
        return
    dst = os.path.join(tempfile.gettempdir(), 'arflow_miopen_db_%d_r%s' % (os.getuid(), os.environ.get('LOCAL_RANK', '0')))
    try:
        os.makedirs(dst, exist_ok=True)
        for f in os.listdir(src):
            if f.endswith('.txt'):
                shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
        os.environ['MIOPEN_USER_DB_PATH'] = dst
    except OSError:
        pass


_use_shipped_miopen_db()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HOST_GLUE = ('arflow_bias_act_fwd', 'arflow_bias_act_bwd', 'arflow_bias_act_fwd_mom')
COMPOSITE_CALLS = ('arflow_level_fwd', 'arflow_level_fwd_m', 'arflow_level_bwd')  # entry points that launch several kernels back to back
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def algorithmic_bytes(name, shape):
    """Compulsory HBM bytes of one launch (each input read once, each output written once, fp32):
    SURVEY section 8(d).  `shape` is the tuple recorded by arflow_amd.functional for the call."""
    if name in ('arflow_corr_fwd', 'arflow_corr_fwd_strided'):
        B, C, H, W, d = shape[:5]
        act = shape[5] if len(shape) > 5 else 0  # fused LeakyReLU: sign words written per pixel
        return 4 * B * H * W * (2 * C + (2 * d + 1) ** 2 + act)
    if name in ('arflow_corr_bwd', 'arflow_corr_bwd_strided'):
        B, C, H, W, d = shape[:5]
        act = int(shape[5]) if len(shape) > 5 else 0  # fused LeakyReLU: 3 sign words (fast path) or the 81-ch output read per pixel
        return 4 * B * H * W * ((2 * d + 1) ** 2 + act + 4 * C)
    if name == 'arflow_corr_fwd_bf16':  # features at 2 bytes
        B, C, H, W, d = shape[:5]
        return B * H * W * (2 * 2 * C + 4 * (2 * d + 1) ** 2)
    if name == 'arflow_corr_bwd_bf16':
        B, C, H, W, d = shape[:5]
        act = int(shape[5]) if len(shape) > 5 else 0
        return B * H * W * (4 * ((2 * d + 1) ** 2 + act) + 2 * 2 * C + 4 * 2 * C)
    if name == 'arflow_warp_fwd_bf16':
        B, C, H, W = shape
        return B * H * W * (2 * C + 4 * C + 8)
    if name == 'arflow_warp_bwd_bf16':
        B, C, H, W, with_src = shape
        return B * H * W * (4 * C + 2 * C + 16 + (4 * C if with_src else 0))
    if name in ('arflow_level_fwd', 'arflow_level_fwd_m'):  # raw x1, x2 in; x2w (with a flow), volume + sign words, normalised x1 out; flow in/out
        B, C, H, W, d, act, fk = shape[:7]
        flow = {0: 0.0, 1: 2.0, 2: 0.5 + 4.0}[fk]
        return int(4 * B * H * W * (2 * C + (C if fk else 0) + (2 * d + 1) ** 2 + act + C + flow))
    if name == 'arflow_level_bwd':  # gvol + signs, x1n, its direct gradient, x1, x2 (+ x2w, flow) in; gx1, gx2 (+ gflow) out
        B, C, H, W, d, act, fk = shape
        flow = {0: 0.0, 1: 2.0 + 2.0, 2: 2.0 + 4.0 + 0.5}[fk]
        return int(4 * B * H * W * ((2 * d + 1) ** 2 + act + (7 if fk else 6) * C + flow))
    if name == 'arflow_up2_bwd':  # fine flow gradient [B,2,H,W] in, coarse one out
        B, H, W = shape[:3]
        return 4 * B * 2 * (H * W + (H // 2) * (W // 2))
    if name == 'arflow_level_warp_fwd':  # x1 (moments), x2 in; x2w out; coarse flow in, upsampled flow out twice
        B, C, H, W, up = shape
        return 4 * B * H * W * (3 * C + 2) + (4 * B * H * W * 2 + 2 * B * H * W if up else 0)
    if name == 'arflow_level_moments':
        B, n = shape
        return 4 * B * n * 2
    if name == 'arflow_level_corr_fwd':  # raw x1, x2w in; volume (+ sign words) and the normalised first map out
        B, C, H, W, d = shape[:5]
        act = shape[5] if len(shape) > 5 else 0
        return 4 * B * H * W * (3 * C + (2 * d + 1) ** 2 + act)
    if name == 'arflow_level_corr_bwd':
        B, C, H, W, d = shape[:5]
        act = int(shape[5]) if len(shape) > 5 else 0
        return 4 * B * H * W * ((2 * d + 1) ** 2 + act + 4 * C)
    if name == 'arflow_featnorm_fwd':
        B, n = shape
        return 4 * B * n * 4  # two tensors in, two out (the second read of the inputs is not compulsory)
    if name == 'arflow_featnorm_bwd':
        B, n = shape
        return 4 * B * n * 6  # g1, g2, x1, x2 in; gx1, gx2 out
    if name in ('arflow_bias_act_fwd', 'arflow_bias_act_fwd_mom'):
        B, C, hw = shape
        return 4 * B * C * hw * 2
    if name == 'arflow_bias_act_bwd':
        B, C, hw = shape
        return 4 * B * C * hw * 3
    if name == 'arflow_warp_fwd':
        B, C, H, W = shape
        return 4 * B * H * W * (2 * C + 2)
    if name == 'arflow_warp_bwd':
        B, C, H, W, with_src = shape
        return 4 * B * H * W * ((3 * C + 4) if with_src else (2 * C + 4))
    if name == 'arflow_census_fwd':
        B, H, W = shape
        return 4 * B * H * W * (3 + 3 + 1 + 1)
    if name == 'arflow_census_bwd':
        B, H, W = shape
        return 4 * B * H * W * (3 + 3 + 1 + 3)
    if name in ('arflow_census_warp_fwd', 'arflow_census_warp_pair_fwd'):
        B, H, W = shape[:3]  # grey a, grey b, flow 2, range map 1/16 in; mask 1, dham 1 out
        return 4 * B * H * W * (1 + 1 + 2 + 1 + 1) + 4 * B * (H // 4) * (W // 4)
    if name in ('arflow_census_warp_bwd', 'arflow_census_warp_pair_bwd', 'arflow_uflow_pair_bwd'):
        B, H, W = shape[:3]  # grey a, grey b, flow 2, dham 1 in; gflow 2 out
        return 4 * B * H * W * (1 + 1 + 2 + 1 + 2)
    if name == 'arflow_splat_smooth_fwd':  # flow 2 + image 3 in, range map 1 out
        B, Ci, H, W = shape
        return 4 * B * H * W * (2 + Ci + 1)
    if name in ('arflow_down4_gray', 'arflow_down4_gray_z'):
        B, H, W = shape  # image 3 in; grey 1 + 3/16 out
        return 4 * B * H * W * (3 + 1) + 4 * B * 3 * (H // 4) * (W // 4)
    if name == 'arflow_photo_fwd':
        B, C, H, W = shape
        return 4 * B * H * W * (2 * C + 1)
    if name == 'arflow_photo_bwd':
        B, C, H, W = shape
        return 4 * B * H * W * (3 * C + 1)
    if name in ('arflow_smooth_fwd', 'arflow_smooth_bwd'):
        B, Ci, H, W = shape
        return 4 * B * H * W * (2 + Ci + (2 if name.endswith('bwd') else 0))
    if name in ('arflow_splat_map', 'arflow_coord_mask'):
        B, H, W = shape
        return 4 * B * H * W * 3
    if name == 'arflow_occ_bidir':
        B, H, W = shape
        return 4 * B * H * W * 5
    if name == 'arflow_down4':
        P, H, W = shape
        return 4 * P * H * W * (1 + 1 / 16)
    if name == 'arflow_up4_clamp_mul':
        B, h, w = shape
        return 4 * B * h * w * (1 + 16 + 16)
    return 0


# VALU roof.  SPEC: 157.3 TFLOPS fp32 vector (MI355X_MICROARCH.md, chip-level parameters) = 78.6 T v_fma lane-instructions/s
# (2 flop per FMA; v_fma_f32 issues over 2 cycles on a SIMD-32) -- the `peak` of every VALU-bound roofline entry.
# MEASURED: tools/ubench/valu_rate.hip (profiles/r02_valu_rate.log) sustains 55.4 T v_fma/s and 19.2 T v_rsq/s on the whole chip;
# reported next to the spec figure as `peak_measured`.  A kernel's algorithmic VALU work is priced in "v_fma slots": 1 per
# plain fp32 op, TRANS_SLOTS per transcendental (the measured v_fma : v_rsq rate ratio).
VALU_PEAK_SPEC_TSLOTS = 78.6
VALU_PEAK_MEASURED_TSLOTS = 55.4
TRANS_SLOTS = 55.4 / 19.2


def valu_slots(name, shape, trans_slots=None):
    """Algorithmic VALU work of one launch in v_fma-equivalent lane slots (arithmetic the algorithm needs, not the
    instructions the kernel happens to issue): the counterpart of algorithmic_bytes() for the kernels the PMC pass
    shows VALU-bound (profiles/r02_pmc_valu.json).  0 = not modelled (HBM side only)."""
    T = TRANS_SLOTS if trans_slots is None else trans_slots  # trans_slots=1: plain instruction count of the same op model
    if name in ('arflow_census_fwd', 'arflow_census_warp_fwd', 'arflow_census_warp_pair_fwd'):
        B, H, W = shape[:3]  # 49 neighbours x (2 sub, 2 fma, 2 rsq, mul, fma, mul, add, rcp, fma) per image pair
        return B * H * W * 49 * (10 + 3 * T)
    if name in ('arflow_census_bwd', 'arflow_census_warp_bwd', 'arflow_census_warp_pair_bwd', 'arflow_uflow_pair_bwd'):
        B, H, W = shape[:3]  # 48 neighbours x (2 sub, 2 fma, 2 rsq, mul, fma, fma, rcp, 5 mul, add, fma)
        return B * H * W * 48 * (14 + 3 * T)
    if name == 'arflow_photo_fwd':
        B, C, H, W = shape  # per pixel-channel: 2 mask products, 3 products, 5 x 9-tap sums, SSIM closed form (1 rcp)
        return B * C * H * W * (5 + 45 + 30 + T)
    if name == 'arflow_photo_bwd':
        B, C, H, W = shape  # window coefficients (as forward) + 9 windows x 3 fma per pixel
        return B * C * H * W * (5 + 45 + 45 + 2 * T + 27)
    if name in ('arflow_corr_fwd', 'arflow_corr_fwd_strided', 'arflow_level_corr_fwd', 'arflow_level_fwd', 'arflow_level_fwd_m'):
        B, C, H, W, d = shape[:5]
        return B * H * W * (2 * d + 1) ** 2 * C
    if name in ('arflow_corr_bwd', 'arflow_corr_bwd_strided', 'arflow_level_corr_bwd', 'arflow_level_bwd'):
        B, C, H, W, d = shape[:5]
        return B * H * W * (2 * d + 1) ** 2 * C * 2
    if name == 'arflow_warp_fwd':
        B, C, H, W = shape  # coordinates + taps ~60, 4 fma per channel
        return B * H * W * (60 + 4 * C)
    if name == 'arflow_level_warp_fwd':
        B, C, H, W = shape[:4]  # + the moments: 4 per channel
        return B * H * W * (80 + 8 * C)
    if name == 'arflow_warp_bwd':
        B, C, H, W, with_src = shape  # flow gradient: 8 per channel; source gradient: 4 per channel
        return B * H * W * (70 + (12 if with_src else 8) * C)
    return 0


def kernel_keys(name, shape):
    """(kernel symbol | grid threads) keys under which the PMC passes over tools/kbench.py filed this launch."""
    def cdiv(a, b):
        return (a + b - 1) // b

    def grid(tiles, threads):
        return 8 * cdiv(tiles, 8) * threads

    def split(tiles, chunks, cap):  # channel chunks spread over gridDim.y (mirrors the launch code)
        n = 1
        while n * 2 <= chunks and tiles * n * 2 <= cap:
            n *= 2
        return n

    def per_sample(B, n, floats_per_block):  # featnorm.hip blocks_per_sample()
        return max(1, min(cdiv(n, floats_per_block), cdiv(2048, B)))

    keys = []
    if name in ('arflow_census_fwd', 'arflow_census_bwd'):
        B, H, W = shape
        keys = ['census4::%s_kernel<3>|%d' % (name[-3:], grid(cdiv(W, 64) * cdiv(H, 16) * B, 256))]
    elif name in ('arflow_census_warp_fwd', 'arflow_census_warp_bwd'):
        B, H, W = shape[:3]
        keys = ['census_warp::%s_kernel<3>|%d' % (name[-3:], grid(cdiv(W, 64) * cdiv(H, 16) * B, 256))]
    elif name in ('arflow_photo_fwd', 'arflow_photo_bwd'):
        B, C, H, W = shape
        keys = ['photo4::%s_kernel|%d' % (name[-3:], grid(cdiv(W, 64) * cdiv(H, 16) * B, 256))]
    elif name in ('arflow_corr_fwd', 'arflow_corr_fwd_strided'):
        B, C, H, W = shape[:4]
        tiles = cdiv(W, 32) * cdiv(H, 8) * B
        if tiles <= 160 and (C // 4) % 4 == 0 and C // 4 >= 8:
            keys = ['corr_v2::fwd_kernel<2, 4>|%d' % grid(tiles, 768)]
        else:
            keys = ['corr_v2::fwd_kernel<%d, 1>|%d' % (2 if tiles >= 768 else 4, grid(tiles, 192))]
    elif name in ('arflow_corr_bwd', 'arflow_corr_bwd_strided'):
        B, C, H, W = shape[:4]
        act = {0: 0, 3: 2}.get(int(shape[5]) if len(shape) > 5 else 0, 1)
        tiles = cdiv(W, 32) * cdiv(H, 8) * B * 2
        keys = ['corr_v2::bwd_kernel<%d, %d>|%d' % (2 if tiles >= 768 else 3, act,
                                                   grid(tiles, 192) * split(tiles, C // 4, 1024))]
    elif name == 'arflow_warp_fwd':
        B, C, H, W = shape
        tiles = cdiv(W, 32) * cdiv(H, 8) * B
        keys = ['warp_fwd_kernel<%d>|%d' % (3 if C <= 3 else 2, grid(tiles, 256) * split(tiles, C // 4, 2048))]
    elif name == 'arflow_warp_bwd':
        B, C, H, W, with_src = shape
        tiles = cdiv(W, 32) * cdiv(H, 8) * B
        g = grid(tiles, 256) * split(tiles, C // 4, 2048)
        keys = ['warp_bwd_flow_kernel<%d>|%d' % (3 if C <= 3 else 2, g)] + (['lds_scatter::warp_bwd_src_kernel|%d' % g] if with_src else [])
    elif name in ('arflow_featnorm_fwd', 'arflow_featnorm_bwd'):
        B, n = shape
        k1, k2 = ('moment_kernel', 'apply_kernel') if name.endswith('fwd') else ('bwd_sum_kernel', 'bwd_apply_kernel')
        keys = ['%s|%d' % (k1, per_sample(B, n, 256 * 16) * B * 256), '%s|%d' % (k2, per_sample(B, n, 256 * 4) * B * 256)]
    return keys


def _pmc_table(fname):
    path = os.path.join(ROOT, 'profiles', fname)
    return json.load(open(path)) if os.path.exists(path) else None


def _call_key(name, shape):
    return '%s|%s' % (name, ','.join(str(v) for v in shape))


def pmc_traffic(name, shape):
    """(HBM bytes per call, source) from the PMC passes kept under profiles/: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
    separate runs of tools/kbench.py on the same shapes (warm caches), gfx950 correction applied (2 x FETCH_SIZE).  Round-3
    tables are keyed by C-ABI call (tools/pmc_calls.py: an entry point may launch several kernels), older ones by kernel
    symbol (tools/make_traffic.py).  (None, None) when that launch shape was not profiled."""
    t = _pmc_table('r03_pmc_traffic.json')
    k = _call_key(name, shape)
    if t and k in t:
        return t[k]['hbm_bytes'], {'file': 'profiles/r03_pmc_traffic.json', 'commit': t.get('_meta', {}).get('commit'),
                                   'cache_state': 'warm (tools/kbench.py re-runs each call on the same buffers)'}
    keys = kernel_keys(name, shape)
    for fname in ('r02_pmc_traffic.json', 'r01_pmc_traffic.json'):
        table = _pmc_table(fname)
        if table and keys and all(kk in table for kk in keys):
            return sum(table[kk]['hbm_bytes'] for kk in keys), {'file': 'profiles/' + fname, 'commit': 'an earlier round (kernel since unchanged or retuned: see DESIGN.md)',
                                                               'cache_state': 'warm'}
    return None, None


def pmc_valu_busy(name, shape):
    """(share of the chip's SIMD time the call spent issuing VALU instructions, source file) from the PMC pass
    profiles/r03_pmc_valu.json (per call) or r02_pmc_valu.json (per kernel), or (None, None)."""
    t = _pmc_table('r03_pmc_valu.json')
    k = _call_key(name, shape)
    if t and k in t and t[k].get('valu_busy') is not None:
        return t[k]['valu_busy'], 'profiles/r03_pmc_valu.json @ %s' % t.get('_meta', {}).get('commit')
    table = _pmc_table('r02_pmc_valu.json')
    keys = kernel_keys(name, shape)
    vals = [table[kk]['valu_busy'] for kk in keys if table and kk in table and table[kk].get('valu_busy') is not None]
    return (max(vals), 'profiles/r02_pmc_valu.json') if vals else (None, None)


def roof(name, shape, avg_ms):
    """Which roof bounds this launch and how close it runs to it, priced from the SPEC: HBM time = algorithmic bytes / 8 TB/s,
    VALU time = algorithmic slots / 78.6 T v_fma/s (157.3 TFLOPS fp32 vector); the larger one is the bound and `frac` is
    that time over the measured time.  `hbm_frac` is always there (rounds stay comparable), `peak_measured` / `frac_measured`
    price the same work against the ceiling tools/ubench/valu_rate.hip sustains on this chip (55.4 T v_fma/s).  `traffic` and
    `valu_busy_pmc` are PMC table look-ups (tools/kbench.py runs, warm caches), each tagged with its source file and the commit
    it was taken at -- not counters of this run.  Returns the `roofline` object."""
    nbytes, slots = algorithmic_bytes(name, shape), valu_slots(name, shape)
    t = avg_ms * 1e-3
    t_hbm, t_valu = nbytes / (HBM_PEAK_GBS * 1e9), slots / (VALU_PEAK_SPEC_TSLOTS * 1e12)
    busy, busy_src = pmc_valu_busy(name, shape)
    traffic, traffic_src = pmc_traffic(name, shape)
    hbm = {'achieved': nbytes / t / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': t_hbm / t}
    r = {'kernel': name, 'shape': list(shape), 'avg_us': 1e6 * t, 'algorithmic_bytes': nbytes,
         'traffic': traffic, 'traffic_source': traffic_src, 'valu_busy_pmc': busy, 'valu_busy_source': busy_src,
         'hbm_frac': t_hbm / t, 'floor_us': 1e6 * max(t_hbm, t_valu)}
    if slots and t_valu > t_hbm:
        r.update({'bound': 'valu', 'achieved': slots / t / 1e12, 'peak': VALU_PEAK_SPEC_TSLOTS, 'unit': 'T v_fma-slots/s',
                  'frac': t_valu / t, 'peak_measured': VALU_PEAK_MEASURED_TSLOTS,
                  'frac_measured': slots / (VALU_PEAK_MEASURED_TSLOTS * 1e12) / t, 'algorithmic_valu_slots': slots, 'hbm': hbm})
    else:
        r.update({'bound': 'hbm', 'achieved': hbm['achieved'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': hbm['frac'],
                  'peak_measured': 6290.0, 'frac_measured': nbytes / 6.29e12 / t})
    return r


def cpu_baseline(workload, height, width, batch, budget_s=25.0):
    """The oracle (CPU restatement, kind 'port') on this host's cores: the SAME workload shape and batch as the GPU line,
    bounded sample (one warm-up step, then timed steps until ~25 s are spent).  Baseline only -- never the thing shipped."""
    from oracle import losses as OL  # checker / baseline only
    from oracle.host_models import oracle_ops
    from arflow_amd.train_step import WORKLOADS, synthetic_pairs
    from arflow_amd.config import AttrDict
    from arflow_amd.models import get_model
    mcfg, lcfg = WORKLOADS[workload]
    torch.manual_seed(0)
    model = get_model(AttrDict(mcfg))
    model.init_weights()
    model.train()
    loss_cls = {'uflow': OL.UFlowLoss, 'unflow': OL.unFlowLoss, 'fullres': OL.FullResLoss, 'mv': OL.MvLoss}[lcfg['type']]
    loss = loss_cls(AttrDict(lcfg))
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    B = int(batch)
    frames = mcfg.get('n_frames', 2)
    x = synthetic_pairs(B, height, width, frames=frames, device='cpu')
    # the GPU box gives one GPU a 16-core CPU share: more torch threads than that only oversubscribe
    cores = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    times = []
    with oracle_ops(model):
        t_start = time.time()
        for it in range(4):
            t0 = time.time()
            res = model(x, with_bk=True)
            if lcfg['type'] == 'mv':
                out = loss(res['flows_fw'], res['flows_bw'], x)
            else:
                flows = [torch.cat([a, b], 1) for a, b in zip(res['flows_fw'], res['flows_bw'])]
                out = loss(flows, x)
            opt.zero_grad()
            out[0].backward()
            opt.step()
            dt = time.time() - t0
            if it > 0:
                times.append(dt)
            if time.time() - t_start > budget_s and times:
                break
    per_step = sum(times) / len(times)
    return {'value': B / per_step, 'unit': 'image-pairs/s', 'cores': cores, 'kind': 'port', 'batch': B,
            'sample': 'oracle/ (pure-PyTorch CPU restatement) full step (fwd+bwd+Adam) of %s at batch %d, %dx%d, '
                      '%d timed step(s) after 1 warm-up, torch threads=%d' % (workload, B, height, width, len(times), cores)}


def _self_launch(n):
    """`python bench.py --gpus N` from a plain shell: start N ranks as CHILD processes through
    torch.distributed.run (never an exec of this process), relay rank 0's JSON line and the exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1])
        sys.stdout.flush()
    if proc.returncode != 0:
        return proc.returncode
    return 0 if lines else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='pwclite_uflow+uflow_loss')
    ap.add_argument('--size', type=int, nargs=2, default=[384, 640])
    ap.add_argument('--batch', type=int, default=8, help='image pairs per GPU')
    ap.add_argument('--feature-storage', choices=['fp32', 'bf16'], default='fp32',
                    help="opt-in: keep the correlation / warp inputs as bf16 in HBM (fp32 arithmetic); default fp32 = the headline")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--miopen-find', '--miopen-benchmark', dest='miopen_benchmark', action='store_true',
                    help='let MIOpen time every solver per convolution (slow the first time; results go to MIOPEN_USER_DB_PATH)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(_self_launch(args.gpus))  # before anything touches the GPU: the ranks are fresh child processes

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    args.gpus = world
    n_dev = torch.cuda.device_count()  # does not initialise the GPU
    if n_dev == 0 or not torch.cuda.is_available():
        sys.exit('bench.py needs a GPU: the hot path has no CPU fallback')
    # Fewer GPUs than ranks (a rehearsal of the N-rank launch path on a smaller box): ranks share devices and
    # the collectives go over gloo, because RCCL refuses two ranks on one device.  The line says so
    # ("oversubscribed": true) -- such a run checks the launch/reduction path, it is not a scaling number.
    oversub = world > n_dev
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    use_dist = world > 1 or os.environ.get('ARFLOW_FORCE_COLLECTIVES') == '1'
    backend = 'gloo' if oversub else 'nccl'
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=device)
        else:
            dist.init_process_group(backend='gloo')

    from arflow_amd import functional as AF
    from arflow_amd.train_step import TrainStep, synthetic_pairs

    H, W = args.size
    if args.miopen_benchmark:
        torch.backends.cudnn.benchmark = True
    step = TrainStep(args.workload, device, seed=1234, feature_storage=args.feature_storage)
    torch.manual_seed(1000 + rank)  # level-dropout draws differ per rank, like independent workers
    img = synthetic_pairs(args.batch, H, W, frames=step.model_cfg.get('n_frames', 2), device=device, seed=100 + rank)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(img)
    sync()
    if not args.no_kernel_timing:
        AF.start_kernel_timing()
    # HIP events bracket the launches of every 4th timed step (sampling keeps the host-side cost of the
    # event records out of a launch-bound step; the averages are still taken live inside the timed region)
    sampled = [i for i in range(args.steps) if i % 4 == 0]
    # CPython's cyclic collector stops the launching thread for ~75 ms roughly once per 50 steps (the step
    # creates ~10^5 short-lived objects; nothing in it needs cycle collection): collect now, keep it off for
    # the timed steps -- what a training loop would do between epochs
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if not args.no_kernel_timing:
            AF.pause_kernel_timing(i % 4 != 0)
        step(img)
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    timing = AF.stop_kernel_timing() if not args.no_kernel_timing else {}
    finite = bool(torch.isfinite(step.last).item())

    per_rank_ms = [1e3 * elapsed / args.steps]
    if use_dist:
        # gloo reduces host tensors; nccl (RCCL) device tensors
        tdev = device if backend == 'nccl' else 'cpu'
        every = [torch.zeros(1, device=tdev, dtype=torch.float64) for _ in range(dist.get_world_size())]
        dist.all_gather(every, torch.tensor([elapsed], device=tdev, dtype=torch.float64))
        per_rank_ms = [1e3 * float(e.item()) / args.steps for e in every]
        elapsed = max(float(e.item()) for e in every)  # MAX over ranks

    if rank == 0:
        pairs = args.batch * world * args.steps
        line = {
            'metric': 'image-pairs/sec fwd+bwd, PWCLite 384x640 bs=8',
            'value': pairs / elapsed, 'unit': 'image-pairs/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.feature_storage == 'fp32' else 'f32 arithmetic, bf16 storage of the correlation/warp inputs',
            'data': 'synthetic',
            'config': {'workload': '%s, %dx%d pairs, batch %d per GPU, fwd(both directions)+loss+bwd+allreduce+Adam'
                                   % (args.workload, H, W, args.batch),
                       'global_batch': args.batch * world, 'parallelism': 'dp%d' % world,
                       'loss_finite': finite},
            # ranks that took part in the gradient all-reduce and the backend that carried it ('nccl' = RCCL)
            # rccl_ranks counts ranks only when RCCL ('nccl') carried the all-reduce; a gloo rehearsal reports 0 there
            'collective_ranks': dist.get_world_size() if use_dist else 0,
            'rccl_ranks': dist.get_world_size() if (use_dist and backend == 'nccl') else 0,
            'collective_backend': backend if use_dist else None,
            'oversubscribed': bool(oversub),
            'global_loss_norm': os.environ.get('ARFLOW_GLOBAL_LOSS_NORM') == '1',
            'per_rank_ms_per_step': per_rank_ms,
        }
        # dominant hot-path kernel by accumulated device time (HIP events around every launch).  The conv
        # epilogue (bias + LeakyReLU) of the host model also goes through the C ABI but is not on the
        # SURVEY section-8 path: it is reported apart and never taken as the roofline kernel.
        per, glue = {}, {}
        for (name, shape), durs in timing.items():
            (glue if name in HOST_GLUE else per)[(name, shape)] = (sum(durs), len(durs))
        if glue:
            g_ms = sum(v[0] for v in glue.values()) / len(sampled)
            g_bytes = sum(algorithmic_bytes(k[0], k[1]) * v[1] for k, v in glue.items()) / len(sampled)
            line['host_glue'] = {'what': 'fused bias+LeakyReLU conv epilogue of the host model (arflow_bias_act_*)',
                                 'ms_per_step': g_ms, 'algorithmic_GB_per_step': g_bytes / 1e9,
                                 'GBps': g_bytes / (g_ms * 1e-3) / 1e9, 'launches_per_step': sum(v[1] for v in glue.values()) / len(sampled)}
        if per:
            # the roofline KERNEL: the dominant call that is ONE kernel (arflow_level_fwd / _bwd chain 2-5 kernels and are
            # reported, as calls, under hot_path; their largest kernel is smaller than the census one -- profiles/r03_*)
            single = {k: v for k, v in per.items() if k[0] not in COMPOSITE_CALLS} or per
            (name, shape), (tot, n) = max(single.items(), key=lambda kv: kv[1][0])
            # the dominant hot-path kernel against the roof that bounds IT (HBM or VALU, see roof())
            line['roofline'] = roof(name, shape, tot / n)
            line['roofline']['launches_per_step'] = n / len(sampled)
            hot_ms = sum(v[0] for v in per.values()) / len(sampled)
            hot_bytes = sum(algorithmic_bytes(k[0], k[1]) * v[1] for k, v in per.items()) / len(sampled)
            roofs = {k: roof(k[0], k[1], v[0] / v[1]) for k, v in per.items()}
            # aggregate over the HBM-bound launches only (the VALU-bound ones are priced against the VALU ceiling),
            # and the time the whole hot path would take with every launch AT its own roof
            hb = [(k, v) for k, v in per.items() if roofs[k]['bound'] == 'hbm']
            hb_ms = sum(v[0] for _, v in hb) / len(sampled)
            hb_bytes = sum(algorithmic_bytes(k[0], k[1]) * v[1] for k, v in hb) / len(sampled)
            at_roof_ms = sum(roofs[k]['frac'] * v[0] for k, v in per.items()) / len(sampled)
            # the time the path would take with every launch AT the spec roof that bounds it (HBM 8 TB/s or 78.6 T v_fma/s)
            floor_ms = sum(roofs[k]['floor_us'] * 1e-3 * v[1] for k, v in per.items()) / len(sampled)
            line['hot_path'] = {'ms_per_step': hot_ms, 'algorithmic_GB_per_step': hot_bytes / 1e9,
                                'GBps': hot_bytes / (hot_ms * 1e-3) / 1e9,
                                'frac_of_hbm_peak': hot_bytes / (hot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                'hbm_bound_only': {'ms_per_step': hb_ms, 'algorithmic_GB_per_step': hb_bytes / 1e9,
                                                   'frac_of_hbm_peak': (hb_bytes / (hb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if hb_ms else None},
                                'frac_of_own_roofs': at_roof_ms / hot_ms, 'floor_ms': floor_ms,
                                'launches_per_step': sum(v[1] for v in per.values()) / len(sampled),
                                'share_of_step': hot_ms / (1e3 * elapsed / args.steps),
                                'kernels': {('%s%s' % (k[0], list(k[1]))): {'us': 1e3 * v[0] / v[1], 'n': v[1] / len(sampled),
                                                                           'GBps': algorithmic_bytes(k[0], k[1]) / (v[0] / v[1] * 1e-3) / 1e9,
                                                                           'bound': roofs[k]['bound'], 'frac_of_roof': roofs[k]['frac'],
                                                                           'hbm_frac': roofs[k]['hbm_frac'], 'floor_us': roofs[k]['floor_us'],
                                                                           'traffic': roofs[k]['traffic'],
                                                                           'valu_busy_pmc': roofs[k]['valu_busy_pmc']}
                                            for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])}}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line['cpu_baseline'] = cpu_baseline(args.workload, H, W, args.batch)
            except Exception as e:  # never lose the GPU number because the baseline leg failed
                line['cpu_baseline'] = {'value': None, 'error': repr(e)}
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
