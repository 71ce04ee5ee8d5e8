/*
 * arflow_hip.h -- C ABI of libarflow_hip.so: the MI355X (gfx950) kernels for the ARFlow hot path
 * (cost-volume correlation, bilinear warp, forward-splat occlusion maps, census / SSIM / L1
 * photometric terms, edge-aware smoothness).
 *
 * Drop-in boundary.  The reference's only native boundary for this path is the pybind11 module
 * `correlation_cuda` (models/correlation_package/correlation_cuda.cc:169-172) with
 *     int forward (Tensor& input1, Tensor& input2, Tensor& rInput1, Tensor& rInput2, Tensor& output,
 *                  int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
 *                  int corr_type_multiply)                                  (correlation_cuda.cc:10-16)
 *     int backward(input1, input2, rInput1, rInput2, gradOutput, gradInput1, gradInput2, ...6 ints)
 *                                                                          (correlation_cuda.cc:89-96)
 * arflow_corr_fwd / arflow_corr_bwd replace those two entry points.  Every other function below
 * replaces a pure-PyTorch function of the reference (cited per function); the Python mirror in
 * arflow_amd/ binds them with ctypes (see INTEGRATION.md for the reference-side stub).
 *
 * Conventions (all functions):
 *   - plain pointers to DEVICE memory, fp32, NCHW, contiguous unless a stride argument says otherwise;
 *   - outputs are caller-allocated; outputs that are accumulated into (scatter targets, reduction
 *     sums) are zero-filled by the callee on the same stream (the reference's callee zero-fills too,
 *     correlation_cuda.cc:40-42,112-115);
 *   - `stream` is a hipStream_t (NULL = default stream); the call only enqueues work: no
 *     synchronisation, no allocation, no global state -> re-entrant per stream, capturable in a
 *     hipGraph;
 *   - return 0 on success, ARFLOW_E* (< 0) for argument errors, -(int)hipError_t - 2000 when the
 *     launch failed (the reference returns 0/1 and raises "CUDA call failed",
 *     correlation_cuda.cc:81-83);  arflow_strerror() names a code.
 */
#ifndef ARFLOW_HIP_H
#define ARFLOW_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* arflow_stream_t; /* hipStream_t */

#define ARFLOW_OK 0
#define ARFLOW_ENULL (-1001)   /* required pointer is NULL */
#define ARFLOW_ESHAPE (-1002)  /* non-positive or inconsistent dimension */
#define ARFLOW_EPARAM (-1003)  /* unsupported mode / parameter value */
#define ARFLOW_ELAUNCH_BASE (-2000)

#define ARFLOW_PAD_ZEROS 0
#define ARFLOW_PAD_BORDER 1

/* how flow is turned into grid_sample coordinates */
#define ARFLOW_NORM_ARFLOW 0 /* utils/warp_utils.py:16-23,83-90: 2*(x+u)/(W-1)-1, un-normalised per align_corners */
#define ARFLOW_NORM_UFLOW 1  /* utils/uflow_utils.py:53-77: 2*(x+u)/max(W-1,1)-1, align_corners=True */
#define ARFLOW_NORM_UFLOW_ABS 2 /* same, but the "flow" argument already holds absolute coordinates
                                   (resample(source, coords) with coords = flow_to_warp(flow)) */
#define ARFLOW_COORDS_ABS 2     /* OR into `variant` / `mode` of splat_map / coord_mask: input holds
                                   absolute coordinates instead of a flow */

/* Reduction outputs ("sums") are NOT single scalars: every workgroup of a reduction kernel STORES its partial sums into
 * its own row of ARFLOW_SUM_COLS floats and zero-fills the rows beyond the kernel's grid -- no zero-fill launch, no
 * atomics, and a value that does not depend on the order the workgroups finish in.  A `sums` argument points to
 * arflow_sums_rows(B, H, W) * ARFLOW_SUM_COLS floats (need not be initialised; every row is written by the call);
 * quantity k is  sum over all rows r of  sums[r * ARFLOW_SUM_COLS + k]. */
#define ARFLOW_SUM_COLS 4
int arflow_sums_rows(int B, int H, int W);

int arflow_abi_version(void);
const char* arflow_strerror(int code);
/* HIP keeps ONE "last error" per host thread.  If a code was already pending when an entry point is entered
 * (left by the host framework or an unchecked earlier call) it is not attributed to this library's launch and
 * not dropped either: the first such hipError_t is kept (and reported once on stderr) until the host reads it
 * here.  Returns that hipError_t (0 = none) and clears the slot.  [No reference counterpart: the reference
 * checks cudaGetLastError() once after its launches, correlation_cuda_kernel.cu:383-392.] */
int arflow_take_stale_error(void);
/* Profiling aid: enqueues a one-wave no-op kernel (`af_marker_kernel`) that delimits the dispatches of consecutive calls in
 * a rocprofv3 kernel / counter trace (tools/kbench.py, tools/pmc_calls.py).  [No reference counterpart.] */
int arflow_profile_marker(int tag, arflow_stream_t stream);

/* ---- cost volume ------------------------------------------------------------------------------
 * out[b, i*(2d+1)+j, y, x] = (1/C) sum_c x1[b,c,y,x] * x2[b,c,y+i-d,x+j-d], zero outside.
 * Replaces correlation_cuda.forward (correlation_cuda.cc:10-87), Correlation.forward
 * (models/correlation_native.py:13-23) and compute_cost_volume (models/uflow_model.py:53-92).
 * x1,x2: [B,C,H,W]; out: [B,(2d+1)^2,H,W]; 1 <= max_disp. */
int arflow_corr_fwd(const float* x1, const float* x2, float* out, unsigned* sign_bits, int B, int C, int H,
                    int W, int max_disp, float negative_slope, arflow_stream_t stream);
/* negative_slope: fused LeakyReLU on the cost volume, out = v > 0 ? v : negative_slope * v -- the
 * activation every caller applies right after the correlation (models/pwclite.py:183-184,
 * models/uflow_model.py:180); 1.0f = plain cost volume.
 * sign_bits (nullable): compact record of `v > 0` for the backward, [B, arflow_corr_sign_planes(), H, W]
 * 32-bit words: bit (i % 3) * 9 + j of plane i / 3 belongs to channel i * 9 + j.  Only produced by the
 * max_disp = 4 fast path: asking for it when arflow_corr_sign_planes(C, W, max_disp) == 0 is ARFLOW_EPARAM. */
int arflow_corr_sign_planes(int C, int W, int max_disp); /* 3, or 0 if this shape has no sign_bits */

/* The same with the cost volume at a BATCH STRIDE: out[b] starts at out + b * out_bstride floats (>= (2d+1)^2*H*W,
 * a multiple of 4), channels and rows contiguous inside a sample -- the volume is written straight into channels
 * [k, k+81) of the [B, Ctot, H, W] buffer the caller's decoder reads as its concatenated input
 * (torch.cat([out_corr_relu, x1, flow, ...], 1) at models/pwclite_uflow.py:218-222, models/pwclite.py:187-189,
 * models/uflow_model.py:192-198), so the 81-channel volume is never copied.  Fast path only
 * (arflow_corr_strided_supported() == 1: max_disp 4, W % 4 == 0, C % 4 == 0), else ARFLOW_EPARAM. */
int arflow_corr_strided_supported(int C, int W, int max_disp);
int arflow_corr_fwd_strided(const float* x1, const float* x2, float* out, long out_bstride, unsigned* sign_bits,
                            int B, int C, int H, int W, int max_disp, float negative_slope, arflow_stream_t stream);

/* Gradients of the above (correlation_cuda.backward, correlation_cuda.cc:89-167; kernels
 * correlation_cuda_kernel.cu:116-300).  gx1 / gx2 may be NULL to skip that gradient. */
int arflow_corr_bwd(const float* gout, const float* out, const unsigned* sign_bits, const float* x1,
                    const float* x2, float* gx1, float* gx2, int B, int C, int H, int W, int max_disp,
                    float negative_slope, arflow_stream_t stream);
/* gout (and out, if given) at batch strides: the gradient of the concatenated decoder input is consumed in place. */
int arflow_corr_bwd_strided(const float* gout, long gout_bstride, const float* out, long out_bstride,
                            const unsigned* sign_bits, const float* x1, const float* x2, float* gx1, float* gx2, int B,
                            int C, int H, int W, int max_disp, float negative_slope, arflow_stream_t stream);
/* With negative_slope != 1 the LeakyReLU derivative is selected per element by the forward's sign_bits
 * (12 bytes per pixel instead of re-reading the 324-byte volume) or, when sign_bits is NULL, by the sign
 * of the forward OUTPUT passed as `out` (as torch's in-place leaky_relu backward does; needs
 * negative_slope > 0).  One of the two must be given; with negative_slope == 1 both may be NULL. */

/* ---- feature normalisation in front of the cost volume ------------------------------------------
 * y_i = (x_i - mu) / sqrt(var + 1e-16), one (mu, var) per sample over both tensors:
 *   ARFLOW_FEATNORM_JOINT  normalize_features of models/pwclite_uflow.py:30-38 (moments of the
 *                          channel-concatenated pair, unbiased variance over 2n - 1)
 *   ARFLOW_FEATNORM_AVG    normalize_features of models/uflow_model.py:8-50 as PWCFlow calls it (:167-172):
 *                          per-tensor mean / unbiased variance over (C,H,W), averaged across the two images
 * x1,x2,y1,y2: [B, n] contiguous (n = C*H*W >= 2).  acc: ARFLOW_FEATNORM_ACC_DOUBLES(B) doubles of scratch
 * (need not be initialised: every workgroup of the reduction pass stores its partial sums into its own row).
 * stats: [B,4] floats written by the forward (m1, m2, mu, std) and read by the backward.
 * Backward: gx1 / gx2 (nullable) = gradients w.r.t. x1 / x2 given g1, g2 = gradients w.r.t. y1, y2. */
#define ARFLOW_FEATNORM_JOINT 0
#define ARFLOW_FEATNORM_AVG 1
#define ARFLOW_FEATNORM_ACC_DOUBLES(B) (4 * (2048 + (B)))
int arflow_featnorm_fwd(const float* x1, const float* x2, float* y1, float* y2, double* acc, float* stats,
                        int B, long n, int mode, arflow_stream_t stream);
int arflow_featnorm_bwd(const float* g1, const float* g2, const float* x1, const float* x2,
                        const float* stats, double* acc, float* gx1, float* gx2, int B, long n, int mode,
                        arflow_stream_t stream);

/* ---- pyramid level, fused --------------------------------------------------------------------
 * One level of the PWC decoders in front of its flow estimator,
 *     flow   = interpolate(flow_coarse * 2, x2, bilinear)                 models/pwclite_uflow.py:208
 *     x2w    = flow_warp(x2, flow)  /  resample(x2, flow_to_warp(flow))   models/pwclite_uflow.py:209, uflow_model.py:164
 *     x1n, x2n = normalize_features([x1, x2w])                            models/pwclite_uflow.py:212-213, uflow_model.py:167-172
 *     vol    = LeakyReLU(corr(x1n, x2n))                                  models/pwclite_uflow.py:214-215, uflow_model.py:174-178
 * as TWO launches forward: (1) arflow_level_warp_fwd -- flow upsample + warp + the partial moments of the
 * normalisation (one row of 4 doubles per workgroup: sum x1, sum x1^2, sum x2w, sum x2w^2); at the level without a
 * flow arflow_level_moments takes its place; (2) arflow_level_corr_fwd -- the cost volume of the NORMALISED pair
 * computed from the raw maps (x1 is normalised in registers, the second map's normalisation is an epilogue term),
 * writing the volume, the normalised first map (the decoder concatenates it) and the sign words of the fused LeakyReLU.
 * The normalised second map, the moment / apply passes and the interpolate / mul launches do not exist.
 *
 * acc: 4 * B * arflow_level_acc_rows(B, C, H, W, has_flow) doubles (not initialised by the caller).
 * flow: [B,2,H/2,W/2] with flow_is_coarse (upsampled x2 inside, align flag up_align_corners; flow_up [B,2,H,W] and /
 * or flow_up2 (batch stride flow_up2_bstride) receive it) or [B,2,H,W]; flow_bstride = floats between samples.
 * out / x1n: volume and normalised first map with batch strides (slots of a concatenated buffer); stats: [B,4]
 * (m1, m2, mu, sigma).  Needs the fast-path shapes (arflow_corr_strided_supported). */
/* The level as ONE call per direction (the entry points the host models use; the stage entry points further down
 * remain for callers that keep their own intermediate buffers):
 *   arflow_level_fwd  = [arflow_level_warp_fwd | arflow_level_moments] + arflow_level_corr_fwd, back to back;
 *   arflow_level_bwd  = arflow_level_corr_bwd + the normalisation's backward (gx1n_direct, the gradient the decoder's
 *     concatenation returns for x1n, is added on load) + the warp's backward (gflow_a [batch stride] and gflow_b
 *     [contiguous], further gradients of the upsampled flow, are added on the way out) + the adjoint of the x2 flow
 *     upsample.  gflow: [B,2,H/2,W/2] with flow_is_coarse, else [B,2,H,W]; flow_full: the fine flow the forward
 *     warped with (flow_up of the forward, or the caller's fine flow; NULL at the level without a warp -- then x2w,
 *     gflow*, gflow are ignored and gx2 is the normalisation's second gradient).
 *   workspace: arflow_level_bwd_ws_bytes(B, C, H, W) bytes of scratch, 256-byte aligned; acc as for the stages. */
int arflow_level_supported(int C, int W, int max_disp);
long arflow_level_bwd_ws_bytes(int B, int C, int H, int W);
int arflow_level_fwd(const float* x1, const float* x2, const float* flow, long flow_bstride, int flow_is_coarse,
                     int up_align_corners, float* flow_up, float* flow_up2, long flow_up2_bstride, float* x2w, int norm_mode,
                     float* out, long out_bstride, float* x1n, long x1n_bstride, unsigned* sign_bits, float* stats,
                     double* acc, int B, int C, int H, int W, int max_disp, float negative_slope, int pad_mode,
                     int align_corners, int coord_norm, arflow_stream_t stream);
/* As arflow_level_fwd, with the partial moments of the feature maps taken where the maps were PRODUCED
 * (arflow_bias_act_fwd_mom rows, [B][nrows][2] doubles of (sum, sum of squares)): x1_rows for the first map (the warp launch
 * then does not read it) and, at the level without a warp, x2_rows for the second (no moment pass at all). */
int arflow_level_fwd_m(const float* x1, const float* x2, const float* flow, long flow_bstride, int flow_is_coarse,
                       int up_align_corners, float* flow_up, float* flow_up2, long flow_up2_bstride, float* x2w, int norm_mode,
                       float* out, long out_bstride, float* x1n, long x1n_bstride, unsigned* sign_bits, float* stats,
                       double* acc, const double* x1_rows, int x1_nrows, const double* x2_rows, int x2_nrows, int B, int C, int H,
                       int W, int max_disp, float negative_slope, int pad_mode, int align_corners, int coord_norm,
                       arflow_stream_t stream);
int arflow_level_bwd(const float* gout, long gout_bstride, const unsigned* sign_bits, const float* x1n, long x1n_bstride,
                     const float* gx1n_direct, long gx1n_direct_bstride, const float* x1, const float* x2, const float* x2w,
                     const float* flow_full, long flow_bstride, const float* gflow_a, long gflow_a_bstride,
                     const float* gflow_b, const float* stats, int norm_mode, float* gx1, float* gx2, float* gflow,
                     int flow_is_coarse, int up_align_corners, void* workspace, int B, int C, int H, int W, int max_disp,
                     float negative_slope, int pad_mode, int align_corners, int coord_norm, arflow_stream_t stream);
int arflow_level_acc_rows(int B, int C, int H, int W, int has_flow);
/* Adjoint of the x2 bilinear flow upsample of arflow_level_warp_fwd (flow_up = interpolate(flow * 2, scale 2),
 * models/pwclite.py:178-179, utils/uflow_utils.py:163-180 upsample(is_flow)): gfine [B,2,H,W] -> gcoarse [B,2,H/2,W/2],
 * every element written (no zero-fill needed). */
int arflow_up2_bwd(const float* gfine, float* gcoarse, int B, int H, int W, int up_align_corners, arflow_stream_t stream);
int arflow_level_moments(const float* x1, const float* x2, double* acc, int B, long n, arflow_stream_t stream);
int arflow_level_warp_fwd(const float* x1, const float* x2, const float* flow, long flow_bstride, int flow_is_coarse,
                          int up_align_corners, float* flow_up, float* flow_up2, long flow_up2_bstride, float* x2w,
                          double* acc, int B, int C, int H, int W, int pad_mode, int align_corners, int norm_mode,
                          arflow_stream_t stream);
int arflow_level_corr_fwd(const float* x1, const float* x2w, const double* acc, int acc_rows, int norm_mode, float* out,
                          long out_bstride, float* x1n, long x1n_bstride, unsigned* sign_bits, float* stats, int B, int C,
                          int H, int W, int max_disp, float negative_slope, arflow_stream_t stream);
/* Backward of arflow_level_corr_fwd w.r.t. the NORMALISED maps: gx1n = d/d x1n (correlation part only: the caller adds the
 * gradient the decoder's concatenation returns for x1n), gx2n = d/d x2n; x1n as written by the forward (batch stride),
 * x2w the raw warped map, stats as written by the forward.  The normalisation's own backward is arflow_featnorm_bwd on
 * (gx1n + direct, gx2n, x1, x2w).  sign_bits NULL <=> negative_slope == 1. */
int arflow_level_corr_bwd(const float* gout, long gout_bstride, const unsigned* sign_bits, const float* x1n,
                          long x1n_bstride, const float* x2w, const float* stats, float* gx1n, float* gx2n, int B, int C,
                          int H, int W, int max_disp, float negative_slope, arflow_stream_t stream);

/* ---- conv epilogue of the host models ------------------------------------------------------------
 * y = lrelu(x + bias[c]) for x, y: [B, C, HW] (x == y allowed; bias nullable): the bias add and
 * LeakyReLU(0.1) that follow every convolution of the reference models (models/pwclite.py:10-23,
 * models/uflow_model.py:271-287) in one pass.  Backward: gin = gout * (y > 0 ? 1 : negative_slope) and
 * gbias[c] = sum_{b,hw} gin (gbias nullable, zero-filled here; fp32 atomics). */
int arflow_bias_act_fwd(const float* x, const float* bias, float* y, int B, int C, long HW,
                        float negative_slope, arflow_stream_t stream);
/* As arflow_bias_act_fwd; additionally every workgroup leaves (sum y, sum y^2) of its slice: mom holds
 * [B][arflow_bias_act_mom_rows(C, HW)][2] doubles (every row written by the call) -- the partial moments of
 * normalize_features taken where the feature map is produced (arflow_level_fwd_m consumes them). */
int arflow_bias_act_mom_rows(int C, long HW);
int arflow_bias_act_fwd_mom(const float* x, const float* bias, float* y, double* mom, int B, int C, long HW,
                            float negative_slope, arflow_stream_t stream);
int arflow_bias_act_bwd(const float* gout, const float* y, float* gin, float* gbias, int B, int C, long HW,
                        float negative_slope, arflow_stream_t stream);

/* ---- bilinear warp ----------------------------------------------------------------------------
 * out[b,c,y,x] = bilinear(src[b,c], x + flow[b,0,y,x], y + flow[b,1,y,x]) with torch grid_sample
 * semantics (pad zeros|border, align_corners) after the reference's normalise/un-normalise round
 * trip.  Replaces flow_warp (utils/warp_utils.py:83-90) and resample(flow_to_warp(.))
 * (utils/uflow_utils.py:6-32,53-77).  src: [B,C,Hs,Ws]; flow: [B,2,H,W] with batch stride
 * flow_bstride floats (>= 2*H*W; lets a [B,4,H,W] tensor be addressed as two flows without a copy);
 * out: [B,C,H,W]; valid (nullable): [B,1,H,W] = mask_invalid(flow_to_warp(flow))
 * (utils/uflow_utils.py:35-50). */
int arflow_warp_fwd(const float* src, const float* flow, float* out, float* valid, int B, int C,
                    int Hs, int Ws, int H, int W, long flow_bstride, int pad_mode, int align_corners,
                    int norm_mode, arflow_stream_t stream);

/* gsrc (nullable, zero-filled here, 4-tap atomic scatter) and gflow (nullable, [B,2,H,W]
 * contiguous). */
int arflow_warp_bwd(const float* gout, const float* src, const float* flow, float* gsrc, float* gflow,
                    int B, int C, int Hs, int Ws, int H, int W, long flow_bstride, int pad_mode,
                    int align_corners, int norm_mode, arflow_stream_t stream);

/* ---- forward-splat maps -----------------------------------------------------------------------
 * variant 0: compute_range_map (utils/uflow_utils.py:80-160, utils/warp_utils.py:158-239)
 * variant 1: get_corresponding_map(grid+flow) (utils/warp_utils.py:26-80) -- clamped indices,
 *            weight zeroed when a tap left the image.
 * out [B,1,H,W] is zero-filled here. */
int arflow_splat_map(const float* flow, float* out, int B, int H, int W, long flow_bstride,
                     int variant, arflow_stream_t stream);

/* mode 0: mask_invalid(flow_to_warp(flow)) closed interval (utils/uflow_utils.py:35-50)
 * mode 1: border_mask(flow) open interval (utils/warp_utils.py:119-134) */
int arflow_coord_mask(const float* flow, float* out, int B, int H, int W, long flow_bstride, int mode,
                      arflow_stream_t stream);

/* get_occu_mask_bidirection (utils/warp_utils.py:93-100): 1 where
 * |f12 + warp(f21,f12)|^2 > scale*(|f12|^2+|warp(f21,f12)|^2) + bias. */
int arflow_occ_bidir(const float* flow12, const float* flow21, float* out, int B, int H, int W,
                     long bstride12, long bstride21, float scale, float bias, arflow_stream_t stream);

/* ---- census / ternary ---------------------------------------------------------------------------
 * Soft census distance between image_a and image_b (both [B,3,H,W], values in [0,1]):
 * ham[b,0,y,x] = sum_k sq/(0.1+sq), sq = (t_a,k - t_b,k)^2, t = d/sqrt(0.81+d^2),
 * d = 255*(gray(y+dy,x+dx) - gray(y,x)) over the (2r+1)^2 patch, zero padding
 * (census_transform + soft_hamming, utils/uflow_utils.py:241-279; TernaryLoss,
 * losses/loss_blocks.py:12-62).
 * If mask != NULL the census_loss reduction (utils/uflow_utils.py:282-293) is fused:
 *   sums[0] += sum (|ham|+0.01)^0.4 * pm,  sums[1] += sum pm,  pm = mask with an r-pixel zero border,
 *   and dham[b,0,y,x] = pm * 0.4*(ham+0.01)^-0.6  (d loss-numerator / d ham) is written instead of ham
 *   when dham != NULL.  sums (2 floats) is zero-filled here.  ham / dham may be NULL. */
int arflow_census_fwd(const float* im_a, const float* im_b, const float* mask, float* ham, float* dham,
                      float* sums, int B, int H, int W, int radius, arflow_stream_t stream);

/* g_im_b[b,c,y,x] = scale[0] * d/d im_b ( sum_p gham[p] * ham[p] ); gham: [B,1,H,W];
 * scale: device scalar (nullable = 1).  Gradient flows into image_b only (call with the images
 * swapped for image_a: the distance is symmetric). */
int arflow_census_bwd(const float* im_a, const float* im_b, const float* gham, const float* scale,
                      float* g_im_b, int B, int H, int W, int radius, arflow_stream_t stream);

/* ---- fused photometric direction of UFlowLoss --------------------------------------------------------
 * losses/uflow_loss.py:30-54 in one launch each way:
 *     recons = resample(im_b, flow_to_warp(flow))                        (utils/uflow_utils.py:6-32,53-77)
 *     mask   = upsample(clamp(occ_small,0,1), x4) * mask_invalid(flow_to_warp(flow))
 *                                               (losses/uflow_loss.py:41-48, utils/uflow_utils.py:35-50,163-182)
 *     sums   = census_loss numerator / denominator of (im_a, recons, mask)   (utils/uflow_utils.py:282-293)
 * The census transform sees only rgb_to_grayscale(image) * 255 (utils/uflow_utils.py:227-231,248), and that is
 * linear like the bilinear sample, so the kernels take the GREY planes gray_a, gray_b [B,1,H,W] (x255, from
 * arflow_down4_gray) and sample one plane instead of three; the warped image, its gradient and the validity mask
 * never exist in memory.  Equal to arflow_warp_fwd + arflow_up4_clamp_mul + arflow_census_fwd up to fp32
 * re-association.
 * flow: [B,2,H,W] with batch stride flow_bstride; occ_small: [B,1,H/4,W/4] (the range map of arflow_splat_map;
 * NULL = no occlusion term, mask = validity only); mask_out: [B,1,H,W] or NULL; dham: [B,1,H,W] (saved for the
 * backward); sums: as arflow_census_fwd (zero-filled here).
 * Needs H % 4 == 0 and W % 4 == 0 (arflow_census_warp_supported() == 1), else ARFLOW_ESHAPE. */
int arflow_census_warp_supported(int H, int W);
int arflow_census_warp_fwd(const float* gray_a, const float* gray_b, const float* flow, long flow_bstride,
                           const float* occ_small, float* mask_out, float* dham, float* sums, int B, int H, int W,
                           int radius, arflow_stream_t stream);
/* gflow[B,2,H,W] = scale[0] * d(sums[0]) / d flow through the census distance and the bilinear sample
 * (= arflow_census_bwd followed by arflow_warp_bwd with gsrc = NULL).  scale: device scalar or NULL (= 1). */
int arflow_census_warp_bwd(const float* gray_a, const float* gray_b, const float* flow, long flow_bstride,
                           const float* dham, const float* scale, float* gflow, int B, int H, int W, int radius,
                           arflow_stream_t stream);
/* Both directions of UFlowLoss (losses/uflow_loss.py:30-54 runs them one after the other) in ONE launch each way: the batch
 * holds B2 = 2 x image pairs, sample s = 2 b + direction -- the model's [B,4,H,W] (fw, bw) flow tensor viewed as [2B,2,H,W]
 * and the [B,6,H,W] image pair viewed as [2B,3,H,W].  `gray`: the 2B grey planes (arflow_down4_gray on that view): image a
 * of sample s is plane s, image b plane s ^ 1; occ_small plane s ^ 1 (range map of the partner direction's level-2 flow)
 * masks sample s.  sums: columns (0, 1) = (sum loss, sum mask) of direction 0, (2, 3) of direction 1.  scale2: two
 * factors, one per direction. */
int arflow_census_warp_pair_fwd(const float* gray, const float* flow, long flow_bstride, const float* occ_small,
                                float* mask_out, float* dham, float* sums, int B2, int H, int W, int radius,
                                arflow_stream_t stream);
int arflow_census_warp_pair_bwd(const float* gray, const float* flow, long flow_bstride, const float* dham,
                                const float* scale2, float* gflow, int B2, int H, int W, int radius, arflow_stream_t stream);
/* The whole backward of UFlowLoss as ONE launch: arflow_census_warp_pair_bwd (-> gflow [B2,2,H,W]) and arflow_smooth_bwd of
 * the level-2 flows flow2 [B2,2,h2,w2] with the x1/4 images img2 [B2,3,h2,w2] and the two sum gradients coef2
 * (-> gflow2 [B2,2,h2,w2]) as workgroup roles of one kernel. */
int arflow_uflow_pair_bwd(const float* gray, const float* flow, long flow_bstride, const float* dham, const float* scale2,
                          float* gflow, int B2, int H, int W, int radius, const float* flow2, long flow2_bstride,
                          const float* img2, const float* coef2, float* gflow2, int h2, int w2, float flow_scale, float alpha,
                          int order, int wmode, int penalty, arflow_stream_t stream);
/* gray[B,1,H,W] = rgb_to_grayscale(im) * 255 (utils/uflow_utils.py:227-231) and, if small != NULL,
 * small[B,3,H/4,W/4] = downsample(im, x1/4) as arflow_down4 (losses/uflow_loss.py:59-60) from the same read.
 * im: [B,3,H,W], H % 4 == 0, W % 4 == 0. */
int arflow_down4_gray(const float* im, float* small, float* gray, int B, int H, int W, arflow_stream_t stream);
/* As arflow_down4_gray; additionally clears zero_plane ([B, H/4, W/4] floats, nullable): the accumulation target of the
 * range-map splat that follows (arflow_splat_smooth_fwd with prezeroed = 1), saving its fill launch. */
int arflow_down4_gray_z(const float* im, float* small, float* gray, float* zero_plane, int B, int H, int W,
                        arflow_stream_t stream);
/* compute_range_map(flow) (utils/uflow_utils.py:80-160; arflow_splat_map variant 0) AND the edge-aware smoothness sums of
 * arflow_smooth_fwd(flow, img [B,3,H,W], ...) (losses/uflow_loss.py:62-102) in ONE launch: UFlowLoss needs both of the same
 * level-2 flows.  out: [B,1,H,W] range map (prezeroed != 0: the caller guarantees it arrives zero-filled); sums as for
 * arflow_smooth_fwd.  The smoothness backward is arflow_smooth_bwd. */
int arflow_splat_smooth_fwd(const float* flow, const float* img, float* out, float* sums, int B, int H, int W,
                            long flow_bstride, float flow_scale, float alpha, int order, int wmode, int penalty, int prezeroed,
                            arflow_stream_t stream);


/* ---- SSIM + L1 photometric term (losses/flow_loss.py:13-27, losses/loss_blocks.py:65-84) --------
 * x = recons*mask, y = im*mask, 3x3 un-padded box SSIM, dist = clamp((1-SSIM)/2,0,1).
 * sums[0] += sum |im-recons|*mask  (B*C*H*W terms)
 * sums[1] += sum dist              (B*C*(H-2)*(W-2) terms)
 * sums[2] += sum mask              (B*H*W terms)
 * ssim_map (nullable): [B,C,H-2,W-2].  mask nullable (= ones).  sums: slotted, 3 quantities. */
int arflow_photo_fwd(const float* im, const float* recons, const float* mask, float* ssim_map,
                     float* sums, int B, int C, int H, int W, arflow_stream_t stream);

/* g_recons = coef[0] * d sums[0]/d recons + coef[1] * d sums[1]/d recons, or, when gmap != NULL,
 * coef[0] * dL1 + sum_w gmap[w] * d dist[w] / d recons.  coef: 2 device floats. */
int arflow_photo_bwd(const float* im, const float* recons, const float* mask, const float* gmap,
                     const float* coef, float* g_recons, int B, int C, int H, int W,
                     arflow_stream_t stream);

/* ---- edge-aware smoothness ------------------------------------------------------------------------
 * sums[0] += sum_{b,ch,y,x} wx * pen(Dx flow),  sums[1] += sum wy * pen(Dy flow)
 * order 1: Dx f = f[x+1]-f[x];           wx = exp(-alpha * mean_c |img[x+1]-img[x]|)
 * order 2: Dx f = f[x+2]-2f[x+1]+f[x];   wx from img[x+2]-img[x+1] (wmode 0, smooth_grad_2nd,
 *          losses/loss_blocks.py:112-124) or img[x+2]-img[x] (wmode 1, UFlowLoss smooth_order 2,
 *          losses/uflow_loss.py:81-102)
 * penalty 0: |v| (loss_blocks.py:101-103), 1: sqrt(v^2 + 1e-6) (penalty_uflow :8-9, robust_l1(v^2)
 * utils/uflow_utils.py:337-338).  flow: [B,2,H,W] (batch stride flow_bstride), flow values are
 * multiplied by flow_scale first; img: [B,Ci,H,W].  sums: slotted, 2 quantities. */
int arflow_smooth_fwd(const float* flow, const float* img, float* sums, int B, int Ci, int H, int W,
                      long flow_bstride, float flow_scale, float alpha, int order, int wmode, int penalty,
                      arflow_stream_t stream);

/* gflow ([B,2,H,W] contiguous) = coef[0]*d sums[0]/d flow + coef[1]*d sums[1]/d flow. */
int arflow_smooth_bwd(const float* flow, const float* img, const float* coef, float* gflow, int B, int Ci,
                      int H, int W, long flow_bstride, float flow_scale, float alpha, int order, int wmode,
                      int penalty, arflow_stream_t stream);

/* ---- resize helpers ---------------------------------------------------------------------------------
 * downsample(img, x1/4) of utils/uflow_utils.py:185-204 for H,W multiples of 4 (= mean of the central
 * 2x2 of each 4x4 block); in [B*C,H,W] -> out [B*C,H/4,W/4]. */
int arflow_down4(const float* in, float* out, int planes, int H, int W, arflow_stream_t stream);

/* out[b,0,y,x] = bilinear x4 upsample (align_corners=False, utils/uflow_utils.py:163-182) of
 * clamp(in,0,1) times valid (nullable): the occlusion-mask assembly of losses/uflow_loss.py:41-48.
 * in [B,1,h,w]; valid/out [B,1,4h,4w]. */
int arflow_up4_clamp_mul(const float* in, const float* valid, float* out, int B, int h, int w,
                         arflow_stream_t stream);

/* ---- the rest of the reference's parameter space (no shipped config uses these values; plain kernels) ----------
 * flow_warp(mode='nearest') (utils/warp_utils.py:83-90 -> grid_sample nearest: border clips the coordinate, index =
 * nearbyint, out of range reads 0).  No gradient w.r.t. the flow (grid_sample's nearest mode has none). */
int arflow_warp_nearest_fwd(const float* src, const float* flow, float* out, int B, int C, int Hs, int Ws, int H, int W,
                            long flow_bstride, int pad_mode, int align_corners, int norm_mode, arflow_stream_t stream);
int arflow_warp_nearest_bwd(const float* gout, const float* flow, float* gsrc, int B, int C, int Hs, int Ws, int H, int W,
                            long flow_bstride, int pad_mode, int align_corners, int norm_mode, arflow_stream_t stream);
/* flow_warp(mode='bicubic') (utils/warp_utils.py:83-90 -> grid_sample bicubic, ATen/native/GridSampler.h: A = -0.75, 4 x 4
 * taps at bounded positions, rows first): gradients w.r.t. the source (gsrc, nullable; zero-filled by the call) and the flow
 * (gflow, nullable).  Plain one-thread-per-pixel kernels: no shipped configuration uses this mode. */
int arflow_warp_bicubic_fwd(const float* src, const float* flow, float* out, int B, int C, int Hs, int Ws, int H, int W,
                            long flow_bstride, int pad_mode, int align_corners, int norm_mode, arflow_stream_t stream);
int arflow_warp_bicubic_bwd(const float* gout, const float* src, const float* flow, float* gsrc, float* gflow, int B, int C,
                            int Hs, int Ws, int H, int W, long flow_bstride, int pad_mode, int align_corners, int norm_mode,
                            arflow_stream_t stream);
/* SSIM(x, y, md) distance map of losses/loss_blocks.py:65-84 for any window (2md+1)^2, 1 <= md <= 16:
 * out [B,C,H-2md,W-2md]; arflow_ssim_bwd gives d(sum gmap*out)/dx (call with x and y swapped for d/dy: SSIM is
 * symmetric).  (md = 1 fused with the L1 term and the mask sums is arflow_photo_fwd/bwd.) */
int arflow_ssim_fwd(const float* x, const float* y, float* out, int B, int C, int H, int W, int md, arflow_stream_t stream);
int arflow_ssim_bwd(const float* x, const float* y, const float* gmap, float* gx, int B, int C, int H, int W, int md,
                    arflow_stream_t stream);
/* arflow_census_fwd / arflow_census_bwd accept 1 <= radius <= 16 (TernaryLoss(max_distance), census_loss(patch_size));
 * radius <= 3 runs the tiled kernels. */
/* Correlation with the CUDA extension's full parameter set (correlation_cuda.cc:10-16, correlation_cuda_kernel.cu:41-114):
 *   out[n, tc, oy, ox] = 1/(K^2 C) sum_{j,i in kernel} sum_c p1[c, y1+j, x1+i] * p2[c, y1+j+tj*s2, x1+i+ti*s2],
 *   p = input zero-padded by pad_size, y1 = oy*stride1 + max_disp, tc = (tj+dr)*(2dr+1)+(ti+dr), dr = max_disp/stride2;
 *   output dims as correlation_cuda.cc:31-34 (arflow_corr_general_out_size).  kernel_size odd.
 * arflow_corr_general_bwd is the exact gradient of that forward (the extension's own backward bounds its window with
 * floor divisions, correlation_cuda_kernel.cu:148-151, and over-counts taps when stride1 > 1).
 * (pad_size = max_disp, kernel_size = stride1 = stride2 = 1 is arflow_corr_fwd/bwd.) */
int arflow_corr_general_out_size(int H, int W, int pad_size, int kernel_size, int max_disp, int stride1, int stride2,
                                 int* out_channels, int* out_h, int* out_w);
int arflow_corr_general_fwd(const float* x1, const float* x2, float* out, int B, int C, int H, int W, int pad_size,
                            int kernel_size, int max_disp, int stride1, int stride2, arflow_stream_t stream);
int arflow_corr_general_bwd(const float* gout, const float* x1, const float* x2, float* gx1, float* gx2, int B, int C, int H,
                            int W, int pad_size, int kernel_size, int max_disp, int stride1, int stride2,
                            arflow_stream_t stream);

/* ---- opt-in bf16 STORAGE of the features ----------------------------------------
 * The reference's native correlation dispatches half-precision tensors as well (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 * correlation_cuda_kernel.cu:352,369).  Here: x1 / x2 / src hold bf16 bit patterns (uint16), every product is
 * accumulated in fp32 and every output and gradient is fp32 -- only the bytes read from HBM (and kept for the
 * backward) are halved.  max_disp must be 4.  With negative_slope != 1 the backward takes the LeakyReLU derivative
 * from the sign of the forward output `out`.  Not the default anywhere: callers ask for it explicitly. */
int arflow_corr_fwd_bf16(const unsigned short* x1, const unsigned short* x2, float* out, int B, int C, int H, int W,
                         int max_disp, float negative_slope, arflow_stream_t stream);
int arflow_corr_bwd_bf16(const float* gout, const float* out, const unsigned short* x1, const unsigned short* x2,
                         float* gx1, float* gx2, int B, int C, int H, int W, int max_disp, float negative_slope,
                         arflow_stream_t stream);
int arflow_warp_fwd_bf16(const unsigned short* src, const float* flow, float* out, float* valid, int B, int C, int Hs,
                         int Ws, int H, int W, long flow_bstride, int pad_mode, int align_corners, int norm_mode,
                         arflow_stream_t stream);
int arflow_warp_bwd_bf16(const float* gout, const unsigned short* src, const float* flow, float* gsrc, float* gflow,
                         int B, int C, int Hs, int Ws, int H, int W, long flow_bstride, int pad_mode, int align_corners,
                         int norm_mode, arflow_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ARFLOW_HIP_H */
