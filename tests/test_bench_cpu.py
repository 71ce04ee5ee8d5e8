"""bench.py helpers that do not need a GPU: the algorithmic-byte formulas and the lookup of
the PMC traffic table kept under profiles/."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_bytes_follow_the_survey_formulas():
    import bench
    B, C, H, W = 16, 32, 96, 160
    px = B * H * W
    assert bench.algorithmic_bytes('arflow_corr_fwd', (B, C, H, W, 4)) == 4 * px * (2 * C + 81)
    assert bench.algorithmic_bytes('arflow_corr_fwd', (B, C, H, W, 4, 3)) == 4 * px * (2 * C + 81 + 3)
    assert bench.algorithmic_bytes('arflow_corr_bwd', (B, C, H, W, 4, 0)) == 4 * px * (81 + 4 * C)
    assert bench.algorithmic_bytes('arflow_corr_bwd', (B, C, H, W, 4, 3)) == 4 * px * (81 + 3 + 4 * C)
    assert bench.algorithmic_bytes('arflow_warp_fwd', (B, C, H, W)) == 4 * px * (2 * C + 2)
    assert bench.algorithmic_bytes('arflow_warp_bwd', (B, C, H, W, True)) == 4 * px * (3 * C + 4)
    assert bench.algorithmic_bytes('arflow_warp_bwd', (8, 3, 384, 640, False)) == 4 * 8 * 384 * 640 * (2 * 3 + 4)
    assert bench.algorithmic_bytes('arflow_census_fwd', (8, 384, 640)) == 4 * 8 * 384 * 640 * 8
    assert bench.algorithmic_bytes('arflow_featnorm_fwd', (B, C * H * W)) == 16 * B * C * H * W
    assert bench.algorithmic_bytes('arflow_bias_act_bwd', (B, 128, H * W)) == 12 * B * 128 * H * W
    assert bench.algorithmic_bytes('no_such_entry', (1,)) == 0


def test_pmc_traffic_table_covers_the_benchmark_shapes():
    import bench
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
    table = json.load(open(path))
    assert all(set(v) >= {'fetch_kb', 'write_kb', 'hbm_bytes'} for v in table.values())
    shapes = [('arflow_census_bwd', (8, 384, 640)), ('arflow_census_fwd', (8, 384, 640)),
              ('arflow_corr_fwd', (16, 32, 96, 160, 4, 3)), ('arflow_corr_bwd', (16, 32, 96, 160, 4, 3)),
              ('arflow_warp_fwd', (16, 32, 96, 160)), ('arflow_warp_bwd', (16, 32, 96, 160, True)),
              ('arflow_warp_fwd', (8, 3, 384, 640)), ('arflow_warp_bwd', (8, 3, 384, 640, False)),
              ('arflow_featnorm_fwd', (16, 491520)), ('arflow_featnorm_bwd', (16, 491520))]
    for name, shape in shapes:
        t, src = bench.pmc_traffic(name, shape)
        a = bench.algorithmic_bytes(name, shape)
        assert t is not None and src['file'].startswith('profiles/'), (name, shape)
        assert 0.9 * a <= t <= 2.5 * a, (name, shape, t, a)  # measured HBM traffic within 2.5x of the compulsory bytes
    assert bench.pmc_traffic('arflow_corr_fwd', (1, 5, 9, 11, 4, 0)) == (None, None)  # not a profiled shape


def test_assert_close_rejects_nan():
    """A kernel returning NaN/Inf must fail every parity test: `err > tol` is False for NaN."""
    import pytest
    import torch
    from tests.conftest import assert_close
    ref = torch.ones(4, 5)
    assert_close(ref.clone(), ref, 1e-6, 1e-6)
    for poison in (float('nan'), float('inf'), -float('inf')):
        bad = ref.clone()
        bad[2, 3] = poison
        with pytest.raises(AssertionError):
            assert_close(bad, ref, 1e-6, 1e-6, 'poisoned')
    with pytest.raises(AssertionError):
        assert_close(ref + 1e-3, ref, 1e-6, 1e-6)


def test_valu_slot_model_against_issued_instructions():
    """the hand-counted 'algorithmic v_fma slots' of bench.valu_slots() for the VALU-bound kernels must stay
    below what the kernels actually ISSUE (SQ_INSTS_VALU x 64 lanes, PMC pass profiles/r02_pmc_valu.json) and within a
    stated factor of it: an op-count model that exceeded the issued instructions, or fell under a third of them, would
    make the VALU roofline fraction meaningless."""
    import json
    import os
    import bench
    table = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r02_pmc_valu.json')))
    cases = [('arflow_census_warp_fwd', (8, 384, 640)), ('arflow_census_warp_bwd', (8, 384, 640)),
             ('arflow_census_fwd', (8, 384, 640)), ('arflow_census_bwd', (8, 384, 640)),
             ('arflow_photo_fwd', (8, 3, 384, 640)), ('arflow_photo_bwd', (8, 3, 384, 640))]
    checked = 0
    for name, shape in cases:
        keys = [k for k in bench.kernel_keys(name, shape) if k in table and 'SQ_INSTS_VALU' in table[k]]
        if not keys:
            continue
        issued = sum(table[k]['SQ_INSTS_VALU'] for k in keys) * 64.0
        # transcendentals count TRANS_SLOTS slots but are ONE issued instruction: compare the plain-instruction count
        plain = bench.valu_slots(name, shape, trans_slots=1.0)
        ratio = plain / issued
        assert 0.33 <= ratio <= 1.25, '%s: modelled %.3g slots vs %.3g issued lane-instructions (ratio %.2f)' % (name, plain, issued, ratio)
        checked += 1
    assert checked >= 4
