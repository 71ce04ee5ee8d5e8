// Pair-symmetric form of the fused census + warp kernels (census_warp.hip) for gfx950.
//
// The soft census distance of utils/uflow_utils.py:241-279 compares, for every pixel p and every offset k of the
// (2R+1)^2 patch, t(p,k) = d / sqrt(0.81 + d^2) with d = grey(p+k) - grey(p), between the two images:
//     ham(p) = sum_k h(p,k),   h = e^2 / (0.1 + e^2),   e = t_a(p,k) - t_b(p,k).
// Seen from the other end of the pair, d(p+k,-k) = -d(p,k), so e flips sign and h(p+k,-k) = h(p,k): every
// UNORDERED pair of pixels needs its two rsq and one rcp ONCE, not twice (the kernels are transcendental/VALU
// bound: 147 transcendentals per pixel in the ordered form).  With K+ = the 24 offsets of the lower half plane
// (ky > 0, or ky = 0 and kx > 0; the centre contributes 0):
//     ham(q) = sum_{k in K+} h(q,k)  +  sum_{k in K+} h(q-k,k)
// -- a pixel evaluates its own 24 "forward" pairs and receives the other 24 from the pixels above / left of it.
// The backward has the same structure: with c = d h / d d_b (odd in d) and w = d loss / d ham,
//     d loss / d grey_b(q) = - sum_{k in K+} (w(q)+w(q+k)) c(q,k) + sum_{k in K+} (w(q-k)+w(q)) c(q-k,k).
//
// Layout.  A workgroup (256 lanes, 4 pixels each) walks DOWN a strip of SW = 56 output columns in chunks of
// CH = 16 rows.  Lanes cover 64 columns (the strip + 4 each side: the senders of the pairs whose receiver is in the
// strip).  Per chunk, for ky = 0..R: every lane evaluates the pairs (p, p+(kx,ky)) of its 4 pixels, keeps them
// for p ("own" role) and writes them as one ds_write_b128 per kx into an LDS plane V[kx][row][col]; after a
// barrier every lane reads the values addressed to ITS pixels from V[kx][row-ky][col-kx] (two aligned
// ds_read_b128 + a compile-time shift).  The rows a chunk needs from the chunk above it (row-ky < 0) come from a
// small carry buffer the previous chunk filled, so only the first chunk of a strip recomputes rows (R of them).
// Pair evaluations per output pixel: 24 * (64/56) * (16n/(16n-R)) ~ 29-31 instead of 48.
//
// The tile of image b is sampled through the flow while it is filled (see census_warp.hip), the mask is evaluated
// per pixel, the backward ends in d loss / d flow.  Every plane access is linear across the wave (16 lanes x 16 B
// per row, rows contiguous): conflict-free.
#include "census_tile.hpp"
#include "taps.hpp"

namespace {
namespace census_sym {
using census4::f32x4;
using census4::read12;
constexpr int NT = 256, SW = 56, LW = 64, CH = 16, PITCH = 128, MAXR = 3;
constexpr int TROWS = CH + MAXR;          // tile rows per chunk (senders + their lower neighbours)
constexpr int VPLANE = CH * LW;           // one kx plane of V
constexpr int GUARD = 4;                  // floats before / after V and the carry (non-output lanes read 1 float4 outside)

__host__ __device__ constexpr int carry_base(int ky) { return (2 * MAXR + 1) * LW * (ky * (ky - 1) / 2); }  // rows before ky
constexpr int CARRY = carry_base(MAXR + 1);  // floats per parity

template <int R>
__device__ __forceinline__ void load_plane(float* __restrict__ tile, const float* __restrict__ g, int H, int W, int cy0,
                                           int tx0 /* image column of tile column 0 */, int tid) {
  constexpr int NR = CH + R, NQ = 18;
  for (int i = tid; i < NR * NQ; i += NT) {
    const int r = i / NQ, q = i - r * NQ;
    const int gy = cy0 + r, gx = tx0 + 4 * q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const float4*>(g + (long)gy * W + gx);
    *reinterpret_cast<float4*>(tile + r * PITCH + 4 * q) = v;
  }
}

__device__ __forceinline__ float sample1(const TapPlan& p, const float (&a)[4]) {
  float r = p.ok[0] ? a[0] * p.w[0] : 0.f;
  r = p.ok[1] ? fmaf(a[1], p.w[1], r) : r;
  r = p.ok[2] ? fmaf(a[2], p.w[2], r) : r;
  r = p.ok[3] ? fmaf(a[3], p.w[3], r) : r;
  return r;
}

// warped grey tile: rows [cy0, cy0+CH+R), tile columns 1 .. 70 (image tx0+1 ..); zero outside the image.
// Rounds of UNR pixels per thread: all flow loads of a round in flight together, then all 4 x UNR taps.
template <int R>
__device__ __forceinline__ void load_warped(float* __restrict__ tile, const float* __restrict__ gsrc,
                                            const float* __restrict__ flow, int H, int W, int cy0, int tx0, int tid) {
  constexpr int NR = CH + R, NC = 70, UNR = 3;
  const long cs = (long)H * W;
#pragma unroll 1
  for (int i0 = tid; i0 < NR * NC; i0 += NT * UNR) {
    float u[UNR], v[UNR];
    int gx[UNR], gy[UNR], dst[UNR];
    bool in[UNR];
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
      const int i = i0 + k * NT;
      const int r = i / NC, c = i - r * NC;
      gy[k] = cy0 + r;
      gx[k] = tx0 + 1 + c;
      dst[k] = i < NR * NC ? r * PITCH + 1 + c : -1;
      in[k] = dst[k] >= 0 && gy[k] >= 0 && gy[k] < H && gx[k] >= 0 && gx[k] < W;
      const long o = in[k] ? (long)gy[k] * W + gx[k] : 0;
      u[k] = flow[o];
      v[k] = flow[o + cs];
    }
    float a[UNR][4];
    TapPlan p[UNR];
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
      const Taps t = make_taps((float)gx[k], (float)gy[k], u[k], v[k], H, W, H, W, ARFLOW_PAD_ZEROS, true, ARFLOW_NORM_UFLOW);
      p[k] = plan_taps(t, H, W);
#pragma unroll
      for (int q = 0; q < 4; ++q) a[k][q] = gsrc[p[k].o[q]];
    }
#pragma unroll
    for (int k = 0; k < UNR; ++k)
      if (dst[k] >= 0) tile[dst[k]] = in[k] ? sample1(p[k], a[k]) : 0.f;
  }
}

__device__ __forceinline__ float up4_clamped(const float* __restrict__ occ, int h, int w, int y, int x) {
  const float sy = fmaxf(0.25f * ((float)y + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.25f * ((float)x + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  auto cl = [](float v) { return fminf(fmaxf(v, 0.f), 1.f); };
  const float v00 = cl(occ[(long)y0 * w + x0]), v01 = cl(occ[(long)y0 * w + x1]);
  const float v10 = cl(occ[(long)y1 * w + x0]), v11 = cl(occ[(long)y1 * w + x1]);
  return (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

__device__ __forceinline__ void st4(float* p, const float (&v)[4]) {
  f32x4 t = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p) = t;
}
// the 4 values addressed to this lane's pixels from a plane row: columns 4g-kx .. 4g-kx+3 (kx is a compile-time
// constant after unrolling: the indices below fold)
__device__ __forceinline__ void ld_shift(const float* rowbase /* column 4g of the source row */, int kx, float (&o)[4]) {
  if (kx == 0) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(rowbase);
    o[0] = t.x, o[1] = t.y, o[2] = t.z, o[3] = t.w;
  } else {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(rowbase + (kx > 0 ? -4 : 0));
    const f32x4 hi = *reinterpret_cast<const f32x4*>(rowbase + (kx > 0 ? 0 : 4));
    const float w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    const int S = kx > 0 ? 4 - kx : -kx;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = w[S + j];
  }
}

struct Lds {
  float* ga;
  float* gb;
  float* gw;  // backward only
  float* V;
  float* carry;  // [2][CARRY]
};

// One ky pass, sender side: evaluate the pairs (p, p + (kx, KY)) of this lane's 4 pixels, accumulate the own
// role, publish the values for the other end of each pair.
template <int R, int KY, bool BWD>
__device__ __forceinline__ void pass_send(const Lds& L, int g, int ly, int parity, const float (&ca)[4], const float (&cb)[4],
                                          const float (&cg)[4], float (&acc)[4]) {
  float wa[12], wb[12], wg[12];
  // the window offset is laundered: without it hipcc hoists the window reads of all four passes above the first
  // barrier (they do not alias V) and keeps ~60 VGPRs per pass alive -- 256 + spills for R = 3
  int woff = (ly + KY) * PITCH + 4 * g;
  asm volatile("" : "+v"(woff));
  read12(L.ga + woff, wa);
  read12(L.gb + woff, wb);
  if (BWD) read12(L.gw + woff, wg);
#pragma unroll
  for (int kx = -R; kx <= R; ++kx) {
    if (KY == 0 && kx <= 0) continue;
    const int kxi = KY == 0 ? kx - 1 : kx + R;
    float v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int k = 4 + p + kx;  // window column of the neighbour
      const float da = wa[k] - ca[p], db = wb[k] - cb[p];
      if (!BWD) {
        const float tb = db * __builtin_amdgcn_rsqf(fmaf(db, db, 0.81f));
        const float e = fmaf(da, __builtin_amdgcn_rsqf(fmaf(da, da, 0.81f)), -tb), sq = e * e;
        v[p] = sq * __builtin_amdgcn_rcpf(0.1f + sq);
        acc[p] += v[p];
      } else {
        const float ua = __builtin_amdgcn_rsqf(fmaf(da, da, 0.81f));
        const float ub = __builtin_amdgcn_rsqf(fmaf(db, db, 0.81f));
        const float e = fmaf(da, ua, -(db * ub));
        const float q = __builtin_amdgcn_rcpf(fmaf(e, e, 0.1f));
        const float hd = ((q * e) * q) * ((ub * ub) * ub);
        v[p] = (wg[k] + cg[p]) * hd;
        acc[p] -= v[p];
      }
    }
    st4(L.V + (kxi * CH + ly) * LW + 4 * g, v);
    if (KY > 0 && ly >= CH - KY)
      st4(L.carry + parity * CARRY + carry_base(KY) + (kxi * KY + (ly - (CH - KY))) * LW + 4 * g, v);
    // two kx (8 independent pixel chains) at a time: left alone the scheduler interleaves all 28 chains of a pass
    // and needs > 256 VGPRs
    if ((kx & 1) == 0) __builtin_amdgcn_sched_barrier(0);
  }
}

// receiver side: the sum of the values of the pairs (q - (kx, KY), q) for this lane's 4 pixels
struct F4 {
  float v[4];
};
template <int R, int KY>
__device__ __forceinline__ F4 pass_recv(const float* base /* column 4g of the source row, plane 0 */, int plane) {
  F4 s = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int kx = -R; kx <= R; ++kx) {
    if (KY == 0 && kx <= 0) continue;
    const int kxi = KY == 0 ? kx - 1 : kx + R;
    float r[4];
    ld_shift(base + kxi * plane, kx, r);
#pragma unroll
    for (int p = 0; p < 4; ++p) s.v[p] += r[p];
  }
  return s;
}

template <int R, int KY, bool BWD>
__device__ __forceinline__ void all_passes(const Lds& L, int g, int ly, int parity, const float (&ca)[4], const float (&cb)[4],
                                           const float (&cg)[4], float (&acc)[4]) {
  pass_send<R, KY, BWD>(L, g, ly, parity, ca, cb, cg, acc);
#ifndef SYM_AB_NO_BARRIER  // A/B timing only: racing exchange -> wrong results
  __syncthreads();
#endif
  {
    const bool from_carry = ly < KY;
    const float* base = from_carry ? L.carry + (parity ^ 1) * CARRY + carry_base(KY) + ly * LW + 4 * g
                                   : L.V + (ly - KY) * LW + 4 * g;
    const F4 r = pass_recv<R, KY>(base, from_carry ? KY * LW : VPLANE);
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] += r.v[p];
  }
#ifndef SYM_AB_NO_BARRIER
  __syncthreads();  // V is rewritten by the next pass / the tiles by the next chunk
#endif
  if constexpr (KY < R) all_passes<R, KY + 1, BWD>(L, g, ly, parity, ca, cb, cg, acc);
}

// strip s of an image: output rows [sy0, sy1), output columns [sx0, sx0+SW)
template <int R, bool BWD>
__global__ __launch_bounds__(NT, 2) void kernel(const float* __restrict__ gray_a, const float* __restrict__ gray_b,
                                            const float* __restrict__ flow, long fbs,
                                            const float* __restrict__ occ_small,  // fwd
                                            float* __restrict__ mask_out,         // fwd
                                            float* __restrict__ dham,             // fwd: out, bwd: in
                                            float* __restrict__ sums, int nrows,  // fwd
                                            const float* __restrict__ scale,      // bwd
                                            float* __restrict__ gflow,            // bwd
                                            int nimg, int H, int W, int strip_h) {
  __shared__ __attribute__((aligned(16))) float lds_ga[TROWS * PITCH];
  __shared__ __attribute__((aligned(16))) float lds_gb[TROWS * PITCH];
  __shared__ __attribute__((aligned(16))) float lds_gw[BWD ? TROWS * PITCH : 4];
  __shared__ __attribute__((aligned(16))) float lds_v[GUARD + (2 * MAXR + 1) * VPLANE + GUARD];
  __shared__ __attribute__((aligned(16))) float lds_c[GUARD + 2 * CARRY + GUARD];
  __shared__ float red[2 * (NT / 64)];
  const int nsx = (W + SW - 1) / SW, nsy = (H + strip_h - 1) / strip_h;
  int stx, sty, b;
  if (!af_tile_of_block(nsx, nsy, nimg, stx, sty, b)) {
    if (!BWD && threadIdx.x == 0) af_store_partial(sums, nrows, 0.f, 0.f, 0.f);  // padding workgroup: its row must be defined
    return;
  }
  const int sx0 = stx * SW, sy0 = sty * strip_h, sy1 = min(sy0 + strip_h, H);
  const long cs = (long)H * W;
  const float* fl = flow + b * fbs;
  const float* ga_p = gray_a + b * cs;
  const float* gb_p = gray_b + b * cs;
  Lds L;
  L.ga = lds_ga, L.gb = lds_gb, L.gw = lds_gw, L.V = lds_v + GUARD, L.carry = lds_c + GUARD;
  const float* occ = (!BWD && occ_small) ? occ_small + (long)b * (H / 4) * (W / 4) : nullptr;
  float part[2] = {0.f, 0.f};
  // tile columns 1 and 70 .. 71 margins of gb are never written by the sampler: zero them once
  for (int i = threadIdx.x; i < TROWS * 2; i += NT) lds_gb[(i >> 1) * PITCH + ((i & 1) ? 71 : 0)] = 0.f;
  int parity = 0;
  for (int cy0 = sy0 - R; cy0 < sy1; cy0 += CH, parity ^= 1) {
    // the lane index is laundered once per chunk: otherwise LICM hoists every lane-dependent address, mask weight
    // and predicate of the ~5000-instruction body out of this loop and keeps > 256 of them alive (spills)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int g = tid & 15, ly = tid >> 4;
    const int x0 = sx0 - 4 + 4 * g;  // image column of this lane's first pixel
    const bool out_col = g >= 1 && g <= SW / 4 && x0 < W;
#ifdef SYM_AB_NO_WARP  // A/B timing only (tools/): plain tile instead of the sampled one -> wrong results
    load_plane<R>(lds_gb, gb_p, H, W, cy0, sx0 - 8, tid);
#else
    load_warped<R>(lds_gb, gb_p, fl, H, W, cy0, sx0 - 8, tid);
#endif
    load_plane<R>(lds_ga, ga_p, H, W, cy0, sx0 - 8, tid);
    if (BWD) load_plane<R>(lds_gw, dham + b * cs, H, W, cy0, sx0 - 8, tid);
    __syncthreads();
    const int y = cy0 + ly;
    const bool out = out_col && y >= sy0 && y < sy1;  // this lane's 4 pixels are outputs of this strip
    // backward: the corner differences of the grey plane at this lane's pixels (gathers in flight during the passes)
    float cdx[4] = {0.f, 0.f, 0.f, 0.f}, cdy[4] = {0.f, 0.f, 0.f, 0.f};
    if (BWD && out) {
      const long o = (long)y * W + x0;
      const float4 fu = *reinterpret_cast<const float4*>(fl + o);
      const float4 fv = *reinterpret_cast<const float4*>(fl + cs + o);
      const float uu[4] = {fu.x, fu.y, fu.z, fu.w}, vv[4] = {fv.x, fv.y, fv.z, fv.w};
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const Taps t = make_taps((float)(x0 + p), (float)y, uu[p], vv[p], H, W, H, W, ARFLOW_PAD_ZEROS, true, ARFLOW_NORM_UFLOW);
        const TapPlan pl = plan_taps(t, H, W);
        float a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = gb_p[pl.o[q]];
        const float nw = pl.ok[0] ? a[0] : 0.f, ne = pl.ok[1] ? a[1] : 0.f;
        const float sw = pl.ok[2] ? a[2] : 0.f, se = pl.ok[3] ? a[3] : 0.f;
        cdx[p] = ((ne - nw) * t.wy0 + (se - sw) * t.wy1) * t.dx;
        cdy[p] = ((sw - nw) * t.wx0 + (se - ne) * t.wx1) * t.dy;
      }
    }
    float ca[4], cb[4], cg[4] = {0.f, 0.f, 0.f, 0.f}, acc[4] = {0.f, 0.f, 0.f, 0.f};
    {
      float w0[12];
      read12(lds_ga + ly * PITCH + 4 * g, w0);
#pragma unroll
      for (int p = 0; p < 4; ++p) ca[p] = w0[4 + p];
      read12(lds_gb + ly * PITCH + 4 * g, w0);
#pragma unroll
      for (int p = 0; p < 4; ++p) cb[p] = w0[4 + p];
      if (BWD) {
        read12(lds_gw + ly * PITCH + 4 * g, w0);
#pragma unroll
        for (int p = 0; p < 4; ++p) cg[p] = w0[4 + p];
      }
    }
    all_passes<R, 0, BWD>(L, g, ly, parity, ca, cb, cg, acc);
    if (out) {
      const long o = (long)y * W + x0;
      if (!BWD) {
        const float4 fu = *reinterpret_cast<const float4*>(fl + o);
        const float4 fv = *reinterpret_cast<const float4*>(fl + cs + o);
        const float uu[4] = {fu.x, fu.y, fu.z, fu.w}, vv[4] = {fv.x, fv.y, fv.z, fv.w};
        float mv[4], dh[4];
        const bool rowin = y >= R && y < H - R;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int xx = x0 + p;
          const float cx = (float)xx + uu[p], cy = (float)y + vv[p];
          const float val = (cx >= 0.f && cx <= (float)(W - 1) && cy >= 0.f && cy <= (float)(H - 1)) ? 1.f : 0.f;
          mv[p] = occ ? up4_clamped(occ, H / 4, W / 4, y, xx) * val : val;
          const float pm = (rowin && xx >= R && xx < W - R) ? mv[p] : 0.f;
          const float lg = __log2f(fabsf(acc[p]) + 0.01f);
          part[0] += exp2f(0.4f * lg) * pm;
          part[1] += pm;
          dh[p] = pm * 0.4f * exp2f(-0.6f * lg);
        }
        if (mask_out) *reinterpret_cast<float4*>(mask_out + (long)b * cs + o) = make_float4(mv[0], mv[1], mv[2], mv[3]);
        *reinterpret_cast<float4*>(dham + (long)b * cs + o) = make_float4(dh[0], dh[1], dh[2], dh[3]);
      } else {
        const float sc = (scale ? scale[0] : 1.f) * (0.1f * -2.f * 0.81f);
        float* gf = gflow + (long)b * 2 * cs + o;
        *reinterpret_cast<float4*>(gf) =
            make_float4(sc * acc[0] * cdx[0], sc * acc[1] * cdx[1], sc * acc[2] * cdx[2], sc * acc[3] * cdx[3]);
        *reinterpret_cast<float4*>(gf + cs) =
            make_float4(sc * acc[0] * cdy[0], sc * acc[1] * cdy[1], sc * acc[2] * cdy[2], sc * acc[3] * cdy[3]);
      }
    }
  }
  if (!BWD) {
    af_block_sum<2>(part, red);
    if (threadIdx.x == 0) af_store_partial(sums, nrows, part[0], part[1], 0.f);
  }
}

}  // namespace census_sym
}  // namespace

// strip height: 16 n - R rows (n chunks, the first one recomputes R rows of its upper neighbour), n chosen so that
// the launch has at least ~3 workgroups per CU
static int census_sym_strip_h(int B, int H, int W, int R) {
  const long per_row_band = (long)B * af_cdiv(W, census_sym::SW);
  int n = 4;
  while (n > 1 && per_row_band * af_cdiv(H, 16 * n - R) < 3 * 256) --n;
  return 16 * n - R;
}

int census_sym_fwd(const float* gray_a, const float* gray_b, const float* flow, long fbs, const float* occ_small,
                   float* mask_out, float* dham, float* sums, int nrows, int B, int H, int W, int radius, hipStream_t st) {
  namespace cs = census_sym;
  const int sh = census_sym_strip_h(B, H, W, radius);
  dim3 g(af_grid_for_tiles((long)af_cdiv(W, cs::SW) * af_cdiv(H, sh) * B));
  switch (radius) {
    case 1: hipLaunchKernelGGL((cs::kernel<1, false>), g, dim3(cs::NT), 0, st, gray_a, gray_b, flow, fbs, occ_small, mask_out, dham, sums, nrows, nullptr, nullptr, B, H, W, sh); break;
    case 2: hipLaunchKernelGGL((cs::kernel<2, false>), g, dim3(cs::NT), 0, st, gray_a, gray_b, flow, fbs, occ_small, mask_out, dham, sums, nrows, nullptr, nullptr, B, H, W, sh); break;
    default: hipLaunchKernelGGL((cs::kernel<3, false>), g, dim3(cs::NT), 0, st, gray_a, gray_b, flow, fbs, occ_small, mask_out, dham, sums, nrows, nullptr, nullptr, B, H, W, sh); break;
  }
  return af_launch_status();
}

int census_sym_bwd(const float* gray_a, const float* gray_b, const float* flow, long fbs, const float* dham,
                   const float* scale, float* gflow, int B, int H, int W, int radius, hipStream_t st) {
  namespace cs = census_sym;
  const int sh = census_sym_strip_h(B, H, W, radius);
  dim3 g(af_grid_for_tiles((long)af_cdiv(W, cs::SW) * af_cdiv(H, sh) * B));
  float* dh = const_cast<float*>(dham);
  switch (radius) {
    case 1: hipLaunchKernelGGL((cs::kernel<1, true>), g, dim3(cs::NT), 0, st, gray_a, gray_b, flow, fbs, nullptr, nullptr, dh, nullptr, 0, scale, gflow, B, H, W, sh); break;
    case 2: hipLaunchKernelGGL((cs::kernel<2, true>), g, dim3(cs::NT), 0, st, gray_a, gray_b, flow, fbs, nullptr, nullptr, dh, nullptr, 0, scale, gflow, B, H, W, sh); break;
    default: hipLaunchKernelGGL((cs::kernel<3, true>), g, dim3(cs::NT), 0, st, gray_a, gray_b, flow, fbs, nullptr, nullptr, dh, nullptr, 0, scale, gflow, B, H, W, sh); break;
  }
  return af_launch_status();
}
