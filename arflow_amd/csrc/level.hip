// One pyramid level of the PWC decoders in front of its flow estimator as ONE C-ABI call per direction
// (models/pwclite_uflow.py:203-222, models/uflow_model.py:160-198):
//
//     flow   = interpolate(flow_coarse * 2, x2, bilinear)                  (flow_is_coarse)
//     x2w    = flow_warp(x2, flow)   /   resample(x2, flow_to_warp(flow))  (no flow: x2w = x2)
//     x1n, x2n = normalize_features([x1, x2w])
//     vol    = LeakyReLU(corr(x1n, x2n))        -> written with x1n and flow into the decoder's concatenated input
//
// forward  = [level_warp_fwd_kernel | moment_kernel] -> corr_v2::fwd_kernel<.., NORM>       (2 launches, no host gap)
// backward = corr_v2::bwd_kernel<.., NORM> -> featnorm sums + apply (the concatenation's gradient of x1n added on load)
//            -> warp backward (the concatenation's / residual's gradients of the flow added on the way out)
//            -> adjoint of the x2 upsample.
// The reference runs ~25 ATen kernels forward and ~60 backward per level for this.
#include <cstdlib>

#include "common.hpp"
#include "level_internal.hpp"

namespace {
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
constexpr int SLAB_CAP = 768;  // cells per tile slab: windows up to e.g. 16 x 48 (a smooth flow gives ~10 x 36)
struct BwdWs {
  size_t g1, g2, gfl, acc, slab, meta, ovf, qinfo, total;
  int slab_cap;
  bool gather;  // fine level: d/d src of the warp as a gather over the inverse-flow window (warp.hip inv_gather)
};
inline BwdWs bwd_layout(int B, int C, int H, int W) {
  BwdWs w;
  const size_t n = align256(sizeof(float) * (size_t)B * C * H * W);
  w.g1 = 0;
  w.g2 = n;
  w.gfl = 2 * n;
  w.acc = w.gfl + align256(sizeof(float) * (size_t)B * 2 * H * W);
  w.slab = w.acc + align256(sizeof(double) * (size_t)ARFLOW_FEATNORM_ACC_DOUBLES(B));
  // two-pass source gradient of the warp at the fine level: one slab of SLAB_CAP cells per tile and channel
  const size_t tiles = (size_t)((W + 31) / 32) * ((H + 7) / 8) * B;
  static const bool slab_on = [] {
    const char* e = getenv("ARFLOW_WARP_SLAB");
    return e && e[0] == '1';
  }();
  w.slab_cap = (slab_on && tiles >= 768 && tiles / B <= 1024) ? SLAB_CAP : 0;
  w.meta = w.slab + align256(sizeof(float) * tiles * C * (size_t)w.slab_cap);
  w.ovf = w.meta + align256(16 * tiles);
  w.qinfo = w.ovf + align256(sizeof(int) * (size_t)B);
  // OPT-IN (ARFLOW_WARP_GATHER=1).  Built, parity-green, measured in the bench step (B16 C32 96x160): the gather kernel takes
  // 29.6 us against 63 us for the atomic scatter and the correlation backward loses its 31 MB zero-fill (85 -> 79 us), but
  // the flows of the RANDOM-INIT network the bench runs are rough (tools/flow_roughness.py: |f(p) - f(q)| <= 2 px for only 58 %
  // of the tap pairs, tile window boxes 7 rows taller than the tile on average), so 40 % of the pairs fall to the per-pair
  // atomics of the flow-gradient role: 135-368 us for that launch, 365 vs 234 us for the level.  For the smooth flows of a
  // trained network the balance is the other way round.
  static const bool gather_on = [] {
    const char* e = getenv("ARFLOW_WARP_GATHER");
    return e && e[0] == '1';
  }();
  // (below ~768 tiles the two gradient roles share one launch and the atomics are not the bound: the gather form would add
  // a launch)
  w.gather = gather_on && !w.slab_cap && tiles >= 768 && H <= 4095 && W <= 4095;
  w.total = w.qinfo + (w.gather ? align256(sizeof(int) * (size_t)B * H * W) : 0);
  return w;
}
}  // namespace

extern "C" int arflow_level_supported(int C, int W, int max_disp);

extern "C" long arflow_level_bwd_ws_bytes(int B, int C, int H, int W) {
  if (!(B > 0 && C > 0 && H > 0 && W > 0)) return ARFLOW_ESHAPE;
  return (long)bwd_layout(B, C, H, W).total;
}

extern "C" int arflow_level_fwd_m(const float* x1, const float* x2, const float* flow, long flow_bstride, int flow_is_coarse,
                                  int up_align_corners, float* flow_up, float* flow_up2, long flow_up2_bstride, float* x2w,
                                  int norm_mode, float* out, long out_bstride, float* x1n, long x1n_bstride,
                                  unsigned* sign_bits, float* stats, double* acc, const double* x1_rows, int x1_nrows,
                                  const double* x2_rows, int x2_nrows, int B, int C, int H, int W, int max_disp,
                                  float negative_slope, int pad_mode, int align_corners, int coord_norm,
                                  arflow_stream_t stream);
extern "C" int arflow_level_fwd(const float* x1, const float* x2, const float* flow, long flow_bstride, int flow_is_coarse,
                                int up_align_corners, float* flow_up, float* flow_up2, long flow_up2_bstride, float* x2w,
                                int norm_mode, float* out, long out_bstride, float* x1n, long x1n_bstride,
                                unsigned* sign_bits, float* stats, double* acc, int B, int C, int H, int W, int max_disp,
                                float negative_slope, int pad_mode, int align_corners, int coord_norm,
                                arflow_stream_t stream) {
  return arflow_level_fwd_m(x1, x2, flow, flow_bstride, flow_is_coarse, up_align_corners, flow_up, flow_up2, flow_up2_bstride,
                            x2w, norm_mode, out, out_bstride, x1n, x1n_bstride, sign_bits, stats, acc, nullptr, 0, nullptr, 0,
                            B, C, H, W, max_disp, negative_slope, pad_mode, align_corners, coord_norm, stream);
}

// As arflow_level_fwd, with the partial moments of the feature maps taken where the maps were PRODUCED
// (arflow_bias_act_fwd_mom rows, [B][nrows][2] doubles): x1_rows for the first map -- the warp launch then does not read it
// at all -- and, at the level without a warp, x2_rows for the second (no moment pass: the level is one launch).
extern "C" int arflow_level_fwd_m(const float* x1, const float* x2, const float* flow, long flow_bstride, int flow_is_coarse,
                                  int up_align_corners, float* flow_up, float* flow_up2, long flow_up2_bstride, float* x2w,
                                  int norm_mode, float* out, long out_bstride, float* x1n, long x1n_bstride,
                                  unsigned* sign_bits, float* stats, double* acc, const double* x1_rows, int x1_nrows,
                                  const double* x2_rows, int x2_nrows, int B, int C, int H, int W, int max_disp,
                                  float negative_slope, int pad_mode, int align_corners, int coord_norm,
                                  arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE(x1_rows == nullptr || x1_nrows > 0, ARFLOW_ESHAPE);
  AF_REQUIRE(x2_rows == nullptr || (x2_nrows > 0 && flow == nullptr && x1_rows != nullptr), ARFLOW_EPARAM);
  AF_REQUIRE_PTR(x1);
  AF_REQUIRE_PTR(x2);
  AF_REQUIRE_PTR(out);
  AF_REQUIRE_PTR(stats);
  AF_REQUIRE_PTR(acc);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && B <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(arflow_level_supported(C, W, max_disp), ARFLOW_EPARAM);
  AF_REQUIRE(norm_mode == ARFLOW_FEATNORM_JOINT || norm_mode == ARFLOW_FEATNORM_AVG, ARFLOW_EPARAM);
  AF_REQUIRE(out_bstride >= 81L * H * W && out_bstride % 4 == 0, ARFLOW_ESHAPE);
  AF_REQUIRE(x1n == nullptr || (x1n_bstride >= (long)C * H * W && x1n_bstride % 4 == 0), ARFLOW_ESHAPE);
  AF_REQUIRE(negative_slope == 1.0f || sign_bits != nullptr, ARFLOW_ENULL);
  hipStream_t st = (hipStream_t)stream;
  const int rows = arflow_level_acc_rows(B, C, H, W, flow != nullptr);
  if (flow) {
    AF_REQUIRE_PTR(x2w);
    AF_REQUIRE(!flow_is_coarse || (H % 2 == 0 && W % 2 == 0), ARFLOW_ESHAPE);
  }
  if (af_level_small_ok(C, H, W) && max_disp == 4)  // coarse level: one workgroup per sample, ONE launch
    return af_level_small_fwd_launch(x1, x2, flow, flow_bstride, flow_is_coarse, up_align_corners, flow_up, flow_up2,
                                     flow_up2_bstride, x2w, norm_mode, out, out_bstride, x1n, x1n_bstride,
                                     negative_slope == 1.0f ? nullptr : sign_bits, stats, B, C, H, W, negative_slope,
                                     pad_mode, align_corners, coord_norm, st, x1_rows, x1_nrows, x2_rows, x2_nrows);
  if (flow) {
    AF_REQUIRE(pad_mode == ARFLOW_PAD_ZEROS || pad_mode == ARFLOW_PAD_BORDER, ARFLOW_EPARAM);
    AF_REQUIRE(coord_norm == ARFLOW_NORM_ARFLOW || coord_norm == ARFLOW_NORM_UFLOW, ARFLOW_EPARAM);
    if (flow_is_coarse) {
      AF_REQUIRE(H % 2 == 0 && W % 2 == 0, ARFLOW_ESHAPE);
      AF_REQUIRE(flow_bstride >= 2L * (H / 2) * (W / 2), ARFLOW_ESHAPE);
      AF_REQUIRE(flow_up2 == nullptr || flow_up2_bstride >= 2L * H * W, ARFLOW_ESHAPE);
    } else {
      AF_REQUIRE(flow_bstride >= 2L * H * W, ARFLOW_ESHAPE);
    }
    const int rc = af_level_warp_fwd_launch(x1_rows ? nullptr : x1, x2, flow, flow_bstride, flow_is_coarse, up_align_corners,
                                            flow_up, flow_up2, flow_up2_bstride, x2w, acc, B, C, H, W, pad_mode, align_corners,
                                            coord_norm, st);
    if (rc != ARFLOW_OK) return rc;
  } else if (!x2_rows) {
    const int rc = af_featnorm_moments_launch(x1, x2, acc, B, (long)C * H * W, st);
    if (rc != ARFLOW_OK) return rc;
  }
  return af_level_corr_fwd_launch(x1, flow ? x2w : x2, (flow || !x2_rows) ? acc : nullptr, (flow || !x2_rows) ? rows : 0,
                                  norm_mode, out, out_bstride, x1n, x1n_bstride, negative_slope == 1.0f ? nullptr : sign_bits,
                                  stats, B, C, H, W, negative_slope, st, x1_rows, x1_nrows, x2_rows, x2_nrows);
}

extern "C" int arflow_level_bwd(const float* gout, long gout_bstride, const unsigned* sign_bits, const float* x1n,
                                long x1n_bstride, const float* gx1n_direct, long gx1n_direct_bstride, const float* x1,
                                const float* x2, const float* x2w, const float* flow_full, long flow_bstride,
                                const float* gflow_a, long gflow_a_bstride, const float* gflow_b, const float* stats,
                                int norm_mode, float* gx1, float* gx2, float* gflow, int flow_is_coarse,
                                int up_align_corners, void* workspace, int B, int C, int H, int W, int max_disp,
                                float negative_slope, int pad_mode, int align_corners, int coord_norm,
                                arflow_stream_t stream) {
  af_clear_stale_error();
  AF_REQUIRE_PTR(gout);
  AF_REQUIRE_PTR(x1n);
  AF_REQUIRE_PTR(x1);
  AF_REQUIRE_PTR(x2);
  AF_REQUIRE_PTR(stats);
  AF_REQUIRE_PTR(gx1);
  AF_REQUIRE_PTR(gx2);
  AF_REQUIRE_PTR(workspace);
  AF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && B <= 65535 && H <= 65535, ARFLOW_ESHAPE);
  AF_REQUIRE(arflow_level_supported(C, W, max_disp), ARFLOW_EPARAM);
  AF_REQUIRE(norm_mode == ARFLOW_FEATNORM_JOINT || norm_mode == ARFLOW_FEATNORM_AVG, ARFLOW_EPARAM);
  AF_REQUIRE(gout_bstride >= 81L * H * W && gout_bstride % 4 == 0, ARFLOW_ESHAPE);
  AF_REQUIRE(x1n_bstride >= (long)C * H * W && x1n_bstride % 4 == 0, ARFLOW_ESHAPE);
  AF_REQUIRE(gx1n_direct == nullptr || (gx1n_direct_bstride >= (long)C * H * W && gx1n_direct_bstride % 4 == 0), ARFLOW_ESHAPE);
  AF_REQUIRE(negative_slope == 1.0f || sign_bits != nullptr, ARFLOW_ENULL);
  AF_REQUIRE(((size_t)workspace & 255) == 0, ARFLOW_EPARAM);
  hipStream_t st = (hipStream_t)stream;
  const BwdWs ws = bwd_layout(B, C, H, W);
  char* base = (char*)workspace;
  float* g1 = (float*)(base + ws.g1);
  float* g2 = (float*)(base + ws.g2);
  float* gfl = (float*)(base + ws.gfl);
  double* acc = (double*)(base + ws.acc);
  float* slab = ws.slab_cap ? (float*)(base + ws.slab) : nullptr;
  int* ovf = (int*)(base + ws.ovf);
  const bool has_flow = flow_full != nullptr;
  if (has_flow) {
    AF_REQUIRE_PTR(x2w);
    AF_REQUIRE_PTR(gflow);
    AF_REQUIRE(flow_bstride >= 2L * H * W, ARFLOW_ESHAPE);
    AF_REQUIRE(pad_mode == ARFLOW_PAD_ZEROS || pad_mode == ARFLOW_PAD_BORDER, ARFLOW_EPARAM);
    AF_REQUIRE(coord_norm == ARFLOW_NORM_ARFLOW || coord_norm == ARFLOW_NORM_UFLOW, ARFLOW_EPARAM);
    AF_REQUIRE(!flow_is_coarse || (H % 2 == 0 && W % 2 == 0), ARFLOW_ESHAPE);
    AF_REQUIRE(gflow_a == nullptr || gflow_a_bstride >= 2L * H * W, ARFLOW_ESHAPE);
  }
  const float* second = has_flow ? x2w : x2;
  // (the upsample's adjoint through float atomics inside the warp launch was tried for the coarse levels: the channel-split
  // workgroups of a tile all add into the same few coarse cells -- 24x40 backward 48 -> 77 us; the gather kernel stays)
  const bool up_atomic = false;
  const bool gather = has_flow && ws.gather;  // gx2 is then written, not accumulated: no zero-fill
  int rc = af_level_corr_bwd_launch(gout, gout_bstride, sign_bits, x1n, x1n_bstride, second, stats, g1, g2, B, C, H, W,
                                    negative_slope, st, (has_flow && !gather) ? gx2 : nullptr,
                                    has_flow ? (flow_is_coarse ? gfl : gflow) : nullptr, up_atomic ? gflow : nullptr,
                                    slab ? ovf : nullptr);
  if (rc != ARFLOW_OK) return rc;
  if (!has_flow)  // the normalisation's backward (the concatenation's gradient of x1n added on load): its outputs ARE the results
    return af_featnorm_bwd_launch(g1, gx1n_direct, gx1n_direct_bstride, g2, x1, x2, stats, acc, gx1, gx2, B, (long)C * H * W,
                                  norm_mode, st);
  // with a warp: only the two sums of the normalisation's backward; its apply pass is folded into the loads of the
  // warp's two gradient kernels (the gradient of the raw warped map is never written)
  int nrows = 0;
  rc = af_featnorm_bwd_sums_launch(g1, gx1n_direct, gx1n_direct_bstride, g2, x1, x2w, stats, acc, &nrows, B, (long)C * H * W, st);
  if (rc != ARFLOW_OK) return rc;
  rc = af_level_warp_bwd_launch(g2, x2, x2w, flow_full, flow_bstride, gx2, flow_is_coarse ? gfl : gflow, B, C, H, W, pad_mode,
                                align_corners, coord_norm, acc, nrows, stats, norm_mode, g1, gx1n_direct, gx1n_direct_bstride,
                                x1, gx1, gflow_a, gflow_a_bstride, gflow_b, up_atomic ? gflow : nullptr, up_align_corners, slab,
                                base + ws.meta, ovf, ws.slab_cap, gather ? (int*)(base + ws.qinfo) : nullptr, st);
  if (rc != ARFLOW_OK || !flow_is_coarse || up_atomic) return rc;
  return af_up2_bwd_launch(gfl, gflow, B * 2, H, W, up_align_corners, st);
}
