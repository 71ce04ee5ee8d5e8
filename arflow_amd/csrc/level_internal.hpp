// Launchers shared between translation units: the level entry points (level.hip) chain kernels that live in
// warp.hip, corr.hip and featnorm.hip inside ONE C-ABI call.
#pragma once
#include "common.hpp"

int af_featnorm_moments_launch(const float* x1, const float* x2, double* acc, int B, long n, hipStream_t st);
int af_featnorm_bwd_launch(const float* g1, const float* g1b, long g1b_bs, const float* g2, const float* x1, const float* x2,
                           const float* stats, double* acc, float* gx1, float* gx2, int B, long n, int mode, hipStream_t st);
int af_featnorm_bwd_apply_launch(const float* g1, const float* g1b, long g1b_bs, const float* g2, const float* x1,
                                 const float* x2, const float* stats, const double* rows, int nrows, float* gx1, float* gx2,
                                 int B, long n, int mode, hipStream_t st);
int af_level_warp_fwd_launch(const float* x1, const float* x2, const float* flow, long flow_bstride, int flow_is_coarse,
                             int up_align_corners, float* flow_up, float* flow_up2, long flow_up2_bstride, float* x2w,
                             double* acc, int B, int C, int H, int W, int pad_mode, int align_corners, int norm_mode,
                             hipStream_t st);
int af_warp_bwd_launch(const float* gout, const float* src, const float* flow, float* gsrc, float* gflow, int B, int C,
                       int Hs, int Ws, int H, int W, long flow_bstride, int pad_mode, int align_corners, int norm_mode,
                       const float* add1, long add1_bs, const float* add2, hipStream_t st);
int af_level_warp_bwd_launch(const float* g2n, const float* x2, const float* x2w, const float* flow, long flow_bstride,
                             float* gx2, float* gflow, int B, int C, int H, int W, int pad_mode, int align_corners,
                             int norm_mode, const double* rows, int nrows, const float* stats, int featnorm_mode,
                             const float* g1n, const float* gdir, long gdir_bs, const float* x1, float* gx1,
                             const float* add1, long add1_bs, const float* add2, float* gcoarse, int up_align,
                             float* slab, void* slab_meta, int* slab_ovf, int slab_cap, int* qinfo, hipStream_t st);
int af_featnorm_bwd_sums_launch(const float* g1, const float* g1b, long g1b_bs, const float* g2, const float* x1,
                                const float* x2, const float* stats, double* acc, int* nrows, int B, long n, hipStream_t st);
int af_up2_bwd_launch(const float* gfine, float* gcoarse, int planes, int H, int W, int up_align, hipStream_t st);
int af_level_corr_fwd_launch(const float* x1, const float* x2w, const double* acc, int acc_rows, int norm_mode, float* out,
                             long out_bstride, float* x1n, long x1n_bstride, unsigned* sign_bits, float* stats, int B, int C,
                             int H, int W, float negative_slope, hipStream_t st, const double* r1, int n1, const double* r2,
                             int n2);
int af_level_corr_bwd_launch(const float* gout, long gout_bstride, const unsigned* sign_bits, const float* x1n,
                             long x1n_bstride, const float* x2w, const float* stats, float* gx1n, float* gx2n, int B, int C,
                             int H, int W, float negative_slope, hipStream_t st, float* zero_c, float* zero_f,
                             float* zero_fc, int* zero_i);

// coarse levels (H * W <= 1024): one workgroup per sample, one launch per direction (level_small.hip)
bool af_level_small_ok(int C, int H, int W);
int af_level_small_fwd_launch(const float* x1, const float* x2, const float* flow, long flow_bstride, int flow_is_coarse,
                              int up_align_corners, float* flow_up, float* flow_up2, long flow_up2_bstride, float* x2w,
                              int norm_mode, float* out, long out_bstride, float* x1n, long x1n_bstride, unsigned* sign_bits,
                              float* stats, int B, int C, int H, int W, float negative_slope, int pad_mode, int align_corners,
                              int coord_norm, hipStream_t st, const double* r1, int n1, const double* r2, int n2);
