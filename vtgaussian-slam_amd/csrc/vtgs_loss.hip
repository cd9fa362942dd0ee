// vtgs_loss.hip -- SSIM of the mapping loss as two kernels.
//
// Replaces utils/slam_external.py:66-97 (calc_ssim / _ssim): five grouped 11x11 Gaussian convolutions (sigma 1.5,
// zero padding) + element-wise algebra + mean, and their autograd backward (five more convolutions).  The window is an
// outer product (utils/slam_external.py:60-63), so every blur is separable: a workgroup stages a (32+10)^2 patch of one
// channel in LDS, blurs rows then columns for the five moments, evaluates the SSIM map and reduces it to one partial sum
// (fixed order, no atomics).  When a gradient is wanted it also writes three per-pixel derivative maps
//     A = d/dmu1 - 2 mu1 d/ds11 - mu2 d/ds12,  B = d/ds11,  C = d/ds12     (derivatives of the SSIM map)
// and the backward is   dL/dx = g/(C H W) * ( blur(A) + 2 x blur(B) + y blur(C) )   -- three blurs, same tiling.
#include "../../include/vtgs.h"
#include "vtgs_internal.h"

namespace vtgs {

constexpr int kST = 32;                 // output tile edge
constexpr int kSR = 5;                  // window radius
constexpr int kSP = kST + 2 * kSR;      // staged patch edge (42)

// The reference's 1-D window (utils/slam_external.py:54-56: float32(exp(-(x - 5)^2 / 4.5)) / their float32 sum), evaluated once
// on the host.  Round 4: every thread of both kernels used to evaluate it itself -- 11 exponentials and 11 IEEE divisions, a
// fifth of the forward kernel's instructions and over a quarter of the backward's.
// (round 5: both SSIM kernels are bound by vector issue -- ~5300 wave-instructions per tile, 4 cycles each on a 16-lane SIMD -- so
//  the blurs accumulate PAIRS of moments with packed v_pk_fma_f32 / v_pk_mul_f32: the same IEEE operations per component, the
//  same bits, 5 instead of 8 instructions per tap in the row pass and 3 instead of 5 in the column pass)
typedef float f32x2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gauss11(float (&w)[11]) {
  constexpr float k[11] = {0.0010283802403137088f, 0.007598758675158024f, 0.036000773310661316f, 0.1093606948852539f,
                           0.21300554275512695f,   0.2660117447376251f,   0.21300554275512695f,  0.1093606948852539f,
                           0.036000773310661316f,  0.007598758675158024f, 0.0010283802403137088f};
#pragma unroll
  for (int i = 0; i < 11; ++i) w[i] = k[i];
}

__global__ __launch_bounds__(256) void ssim_forward_kernel(const float* __restrict__ img1, const float* __restrict__ img2,
                                                           int C, int H, int W, int tiles_x, int tiles_y,
                                                           float* __restrict__ partial, float* __restrict__ gmaps,
                                                           int ty0 = 0, int rb = 0, int re = 1 << 30) {
  // band form (tile-row multi-GPU partition): tiles_y tile rows starting at ty0 are launched and only the SSIM pixels of
  // rows [rb, re) are summed / get derivative maps; the staged context rows around them are read as they are (the caller
  // has put the neighbours' rows there).  Whole image: ty0 = 0, rb = 0, re >= H.
  // pairs that are blurred together sit together in LDS (one ds_read_b64 = one packed operand, no register shuffling)
  __shared__ float2 pxy[kSP * kSP];                                   // (img1, img2)
  __shared__ float2 hzm[kSP * kST], hzs[kSP * kST];                   // row-blurred (x, y), (x^2, y^2)
  __shared__ float hzc[kSP * kST];                                    // row-blurred x y
  __shared__ float red[4];
#ifdef VTGS_AB_SSIM_PAD                                         // occupancy experiment
  __shared__ float ab_pad[VTGS_AB_SSIM_PAD];
  if (C < 0) { ab_pad[threadIdx.x] = 1.f; __syncthreads(); if (ab_pad[(threadIdx.x + 1) & 255] == 2.f) return; }
#endif
  float w[11];
  gauss11(w);
  const int t = (int)threadIdx.x;
  // Round 5: a workgroup walks tiles b = blockIdx.x, + gridDim.x, ... (the grid is what fits the chip at once, vtgs_ssim_grid) and
  // requests the NEXT tile's patch into registers before it blurs the current one: a tile used to cost ~8 us of which the blurs
  // were ~3 -- the rest was the wait for its own 2 x 1764 pixels with three workgroups per CU to hide it behind.  Same arithmetic
  // per tile, same partial-sum slot per tile: bit-identical results.
  const int ntile = C * tiles_x * tiles_y;
  constexpr int kStage = (kSP * kSP + 255) / 256;                     // 7 staged pixels per thread and image
  float ra[kStage], rd[kStage];
  const int rt = t / kSP, qt = t - rt * kSP;                          // patch position of the thread's first staged pixel
  // (the staging was a third of the kernel's vector instructions: a division, 64-bit address arithmetic and a branch per
  //  pixel.  Now: the position advances by 256 = 6 x 42 + 4, offsets are 32-bit from a uniform plane pointer, and a pixel
  //  outside the image is loaded from offset 0 and replaced by zero -- no branch)
  auto fetch = [&](int bb) {
    const int c_ = bb / (tiles_x * tiles_y), tb_ = bb - c_ * tiles_x * tiles_y;
    const int ty_ = ty0 + tb_ / tiles_x, tx_ = tb_ - (tb_ / tiles_x) * tiles_x;
    const int x0 = tx_ * kST - kSR, y0 = ty_ * kST - kSR;
    const float* __restrict__ p1 = img1 + (size_t)c_ * H * W;
    const float* __restrict__ p2 = img2 + (size_t)c_ * H * W;
    int r = rt, q = qt;
#pragma unroll
    for (int k = 0; k < kStage; ++k) {
      const int gy = y0 + r, gx = x0 + q;
      const bool in = (k < kStage - 1 || t + 256 * k < kSP * kSP) && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;   // zero padding
      const int o = in ? gy * W + gx : 0;
      const float a1 = p1[o], a2 = p2[o];
      ra[k] = in ? a1 : 0.f;
      rd[k] = in ? a2 : 0.f;
      q += 256 - 6 * kSP; r += 6;
      if (q >= kSP) { q -= kSP; r += 1; }
    }
  };
  if ((int)blockIdx.x < ntile) fetch((int)blockIdx.x);
  for (int b = (int)blockIdx.x; b < ntile; b += (int)gridDim.x) {
  const int c = b / (tiles_x * tiles_y), tb = b - c * tiles_x * tiles_y;
  const int ty = ty0 + tb / tiles_x, tx = tb - (tb / tiles_x) * tiles_x;
  const size_t plane = (size_t)c * H * W;
#pragma unroll
  for (int k = 0; k < kStage; ++k) {
    const int i = t + 256 * k;
    if (i < kSP * kSP) pxy[i] = make_float2(ra[k], rd[k]);
  }
  __syncthreads();
  if (b + (int)gridDim.x < ntile) fetch(b + (int)gridDim.x);          // in flight during both passes
  // Both passes slide the window in registers: a work item produces FOUR adjacent outputs from 14 staged values instead of
  // 4 x 11 (the kernel was bound by its LDS reads: 86 K per workgroup).  Every output still accumulates its 11 taps in the
  // same order, so the values are the ones of the one-output-per-item form.
  for (int i = t; i < kSP * (kST / 4); i += 256) {                    // rows: 42 x 32 outputs, 4 per item
    const int r = i / (kST / 4), q = 4 * (i - r * (kST / 4));
    f32x2s v[14], v2[14];
    float vxy[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) {                                    // squares and the product once per staged pixel, not per tap
      const float2 pp = pxy[r * kSP + q + k];
      v[k] = f32x2s{pp.x, pp.y}; v2[k] = v[k] * v[k]; vxy[k] = pp.x * pp.y;
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      f32x2s m = {0.f, 0.f}, sq = {0.f, 0.f};
      float xy = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const f32x2s wk = {w[k], w[k]};
        m = __builtin_elementwise_fma(wk, v[o + k], m);
        sq = __builtin_elementwise_fma(wk, v2[o + k], sq);
        xy = fmaf(w[k], vxy[o + k], xy);
      }
      const int j = r * kST + q + o;
      hzm[j] = make_float2(m.x, m.y); hzs[j] = make_float2(sq.x, sq.y); hzc[j] = xy;
    }
  }
  __syncthreads();
  float acc = 0.f;
  const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
  {                                                                   // columns: 32 x 32 outputs, 4 rows per item
    const int q = t & (kST - 1), r0 = 4 * (t >> 5);                   // 256 items = 8 row groups x 32 columns
    f32x2s hm_[14], hs_[14];
    float hc_[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      const float2 a2 = hzm[(r0 + k) * kST + q], b2 = hzs[(r0 + k) * kST + q];
      hm_[k] = f32x2s{a2.x, a2.y}; hs_[k] = f32x2s{b2.x, b2.y}; hc_[k] = hzc[(r0 + k) * kST + q];
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int r = r0 + o;
      const int gy = ty * kST + r, gx = tx * kST + q;
      f32x2s mm = {0.f, 0.f}, sq = {0.f, 0.f};
      float xy = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const f32x2s wk = {w[k], w[k]};
        mm = __builtin_elementwise_fma(wk, hm_[o + k], mm);
        sq = __builtin_elementwise_fma(wk, hs_[o + k], sq);
        xy = fmaf(w[k], hc_[o + k], xy);
      }
      const float m1 = mm.x, m2 = mm.y, xx = sq.x, yy = sq.y;
      if (gy < H && gx < W && gy >= rb && gy < re) {
        const float s11 = xx - m1 * m1, s22 = yy - m2 * m2, s12 = xy - m1 * m2;
        const float a1 = 2.f * m1 * m2 + c1, a2 = 2.f * s12 + c2, b1 = m1 * m1 + m2 * m2 + c1, b2 = s11 + s22 + c2;
        // two reciprocals per pixel: hardware estimate + one Newton step (< 1 ulp; the IEEE division was ~10 instructions
        // each in a kernel bound by vector issue).  b1, b2 >= c1, c2 > 0.
        float r1 = __builtin_amdgcn_rcpf(b1), r2 = __builtin_amdgcn_rcpf(b2);
        r1 = fmaf(r1, fmaf(-b1, r1, 1.f), r1); r2 = fmaf(r2, fmaf(-b2, r2, 1.f), r2);
        const float ib = r1 * r2;
        const float ssim = a1 * a2 * ib;
        acc += ssim;
        if (gmaps) {
          const float d_mu1 = 2.f * m2 * a2 * ib - ssim * 2.f * m1 * r1;
          const float d_s11 = -ssim * r2, d_s12 = 2.f * a1 * ib;
          const size_t P3 = (size_t)C * H * W;
          float* __restrict__ gm0 = gmaps + plane;                    // uniform bases, 32-bit offsets
          float* __restrict__ gm1 = gm0 + P3;
          float* __restrict__ gm2 = gm1 + P3;
          const int o2 = gy * W + gx;
          gm0[o2] = d_mu1 - 2.f * m1 * d_s11 - m2 * d_s12;
          gm1[o2] = d_s11;
          gm2[o2] = d_s12;
        }
      }
    }
  }
  acc = wave_sum(acc);
  if ((t & 63) == 0) red[t >> 6] = acc;
  __syncthreads();
  if (t == 0) partial[b] = red[0] + red[1] + red[2] + red[3];
  __syncthreads();                                                    // red / hz / the patch are rewritten by the next tile
  }
}

__global__ __launch_bounds__(256) void ssim_backward_kernel(const float* __restrict__ img1, const float* __restrict__ img2,
                                                            const float* __restrict__ gmaps, const float* __restrict__ upstream,
                                                            int C, int H, int W, int tiles_x, int tiles_y,
                                                            float* __restrict__ g_img1, float ssim_coef, float l1_coef,
                                                            const float* __restrict__ l1_weight = nullptr,
                                                            int ty0 = 0, int rb = 0, int re = 1 << 30) {
  // band form: the derivative maps exist on rows [rb, re) only (anything else counts as zero); the gradient is written on
  // rows [rb - 5, re + 5) -- the blur carries it into the neighbours' rows -- and the L1 term belongs to rows [rb, re).
  __shared__ float2 pm01[kSP * kSP];                                  // maps A, B as a pair (see the forward), C apart
  __shared__ float pm2[kSP * kSP];
  __shared__ float2 hz01[kSP * kST];
  __shared__ float hz2[kSP * kST];
  float w[11];
  gauss11(w);
  const int t = (int)threadIdx.x;
  const size_t P3 = (size_t)C * H * W;
  const int ntile = C * tiles_x * tiles_y;                            // tiles walked with the next one's patch in flight (see the forward)
  constexpr int kStage = (kSP * kSP + 255) / 256;
  float rm[3][kStage];
  const int rt = t / kSP, qt = t - rt * kSP;
  const int row_lo = rb > 0 ? rb : 0, row_hi = re < H ? re : H;       // the maps exist on these rows
  auto fetch = [&](int bb) {                                          // (staging as in the forward: no division, no branch)
    const int c_ = bb / (tiles_x * tiles_y), tb_ = bb - c_ * tiles_x * tiles_y;
    const int ty_ = ty0 + tb_ / tiles_x, tx_ = tb_ - (tb_ / tiles_x) * tiles_x;
    const int x0 = tx_ * kST - kSR, y0 = ty_ * kST - kSR;
    const float* __restrict__ g0 = gmaps + (size_t)c_ * H * W;
    const float* __restrict__ g1 = g0 + P3;
    const float* __restrict__ g2 = g1 + P3;
    int r = rt, q = qt;
#pragma unroll
    for (int k = 0; k < kStage; ++k) {
      const int gy = y0 + r, gx = x0 + q;
      const bool in = (k < kStage - 1 || t + 256 * k < kSP * kSP) && gy >= row_lo && gy < row_hi && (unsigned)gx < (unsigned)W;   // the adjoint of a zero-padded blur is the same blur
      const int o = in ? gy * W + gx : row_lo * W;
      const float a0 = g0[o], a1 = g1[o], a2 = g2[o];
      rm[0][k] = in ? a0 : 0.f; rm[1][k] = in ? a1 : 0.f; rm[2][k] = in ? a2 : 0.f;
      q += 256 - 6 * kSP; r += 6;
      if (q >= kSP) { q -= kSP; r += 1; }
    }
  };
  if ((int)blockIdx.x < ntile) fetch((int)blockIdx.x);
  const float scale = upstream[0] * ssim_coef / (float)((size_t)C * H * W);
  const float l1s = upstream[0] * l1_coef;
  for (int b = (int)blockIdx.x; b < ntile; b += (int)gridDim.x) {
  const int c = b / (tiles_x * tiles_y), tb = b - c * tiles_x * tiles_y;
  const int ty = ty0 + tb / tiles_x, tx = tb - (tb / tiles_x) * tiles_x;
  const size_t plane = (size_t)c * H * W;
#pragma unroll
  for (int k = 0; k < kStage; ++k) {
    const int i = t + 256 * k;
    if (i < kSP * kSP) { pm01[i] = make_float2(rm[0][k], rm[1][k]); pm2[i] = rm[2][k]; }
  }
  __syncthreads();
  if (b + (int)gridDim.x < ntile) fetch(b + (int)gridDim.x);
  for (int i = t; i < kSP * (kST / 4); i += 256) {                    // rows, four outputs per item (see the forward)
    const int r = i / (kST / 4), q = 4 * (i - r * (kST / 4));
    f32x2s v01[14];
    float v2_[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      const float2 pp = pm01[r * kSP + q + k];
      v01[k] = f32x2s{pp.x, pp.y}; v2_[k] = pm2[r * kSP + q + k];
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      f32x2s s01 = {0.f, 0.f};
      float s2 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const f32x2s wk = {w[k], w[k]};
        s01 = __builtin_elementwise_fma(wk, v01[o + k], s01);
        s2 = fmaf(w[k], v2_[o + k], s2);
      }
      const int j = r * kST + q + o;
      hz01[j] = make_float2(s01.x, s01.y); hz2[j] = s2;
    }
  }
  __syncthreads();
  // dL/dimg1 = upstream * ( ssim_coef * d(mean SSIM)/dimg1 + l1_coef * sign(img1 - img2) )   (plain SSIM: 1, 0)
  {                                                                   // columns, four rows per item
    const int q = t & (kST - 1), r0 = 4 * (t >> 5);
    f32x2s h01[14];
    float h2_[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      const float2 pp = hz01[(r0 + k) * kST + q];
      h01[k] = f32x2s{pp.x, pp.y}; h2_[k] = hz2[(r0 + k) * kST + q];
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int gy = ty * kST + r0 + o, gx = tx * kST + q;
      if (gy >= H || gx >= W || gy < rb - kSR || gy >= re + kSR) continue;
      f32x2s s01 = {0.f, 0.f};
      float s2 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const f32x2s wk = {w[k], w[k]};
        s01 = __builtin_elementwise_fma(wk, h01[o + k], s01);
        s2 = fmaf(w[k], h2_[o + k], s2);
      }
      const float s0 = s01.x, s1 = s01.y;
      // (the tile's own pixels are read here, not with the next tile's patch: eight more registers in flight across the
      //  blurs cost the fourth wavefront per SIMD, 26 -> 29 us)
      const int o2 = gy * W + gx;                                     // uniform bases, 32-bit offsets
      const float x = (img1 + plane)[o2], y = (img2 + plane)[o2];
      const float lw = (gy >= rb && gy < re) ? (l1_weight ? l1s * (l1_weight + plane)[o2] : l1s) : 0.f;
      (g_img1 + plane)[o2] = scale * (s0 + 2.f * x * s1 + y * s2) + ((x > y) ? lw : (x < y) ? -lw : 0.f);
    }
  }
  __syncthreads();                                                    // hz / the patch are rewritten by the next tile
  }
}

// ---- masked L1 terms of get_loss (src/vtgaussian_slam.py:519-608), value and gradient images in one pass -------------
// mode 0 (tracking): mask = gt_depth > 0 & finite depth & finite uncertainty & silhouette > sil_thres;
//                    partial[b] = {sum_mask |gt_im - im| (3 channels), sum_mask |gt_depth - depth|, count}
// mode 1 (mapping):  mask = gt_depth > 0 & finite depth & finite uncertainty;   colour L1 over ALL pixels (it is a mean)
// mode 2 (tracking with neither use_sil_for_loss nor ignore_outlier_depth_loss, src/vtgaussian_slam.py:601-602): the depth
//                    term as in mode 0, the colour SUM over all pixels
// Gradient images hold d(sum)/d(im) and d(sum)/d(depth_sil[0]) (channels 1, 2 of depth_sil only enter detached masks).
__global__ __launch_bounds__(256) void masked_l1_kernel(const float* __restrict__ im, const float* __restrict__ ds,
                                                        const float* __restrict__ gt_im, const float* __restrict__ gt_depth,
                                                        int P, float sil_thres, int mode, float* __restrict__ partial,
                                                        float* __restrict__ g_im, float* __restrict__ g_ds,
                                                        const float* __restrict__ extra_mask = nullptr,
                                                        const float* __restrict__ color_weight = nullptr,
                                                        int p0 = 0, int p1 = -1) {
  // [p0, p1): the pixels this call sums (a band of rows of the tile-row partition); P stays the plane stride
  __shared__ float red[4][3];
  float s_im = 0.f, s_d = 0.f, cnt = 0.f;
  if (p1 < 0) p1 = P;
  // one pixel: the masks, the three sums, and (optionally) the unscaled gradient images
  auto pixel = [&](float depth, float sil, float dsq, float gd, float em, const float (&x)[3], const float (&y)[3],
                   const float (&cwv)[3], float& gds, float (&gim)[3]) {
    const float unc = dsq - depth * depth;
    bool m = gd > 0.f && depth == depth && unc == unc;
    if (mode != 1) m = m && sil > sil_thres;
    m = m && em != 0.f;                                      // visibility / far-depth / outlier masks of the other datasets
    const bool mc = (mode == 0) ? m : true;
    const float d = gd - depth;
    s_d += m ? fabsf(d) : 0.f;
    cnt += m ? 1.f : 0.f;
    gds = m ? (d > 0.f ? -1.f : (d < 0.f ? 1.f : 0.f)) : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float e = y[c] - x[c];
      s_im += mc ? fabsf(e) * cwv[c] : 0.f;                  // cw -- mapping: 10 additional_mask + 0.8
      gim[c] = mc ? (e > 0.f ? -cwv[c] : (e < 0.f ? cwv[c] : 0.f)) : 0.f;
    }
  };
  // Round 5: FOUR pixels per thread as float4 when the planes allow it (plane stride and band start multiples of 4, 16-byte
  // bases: every reference frame size): a thread of the one-pixel form walked ~3 pixels one after the other with ten scalar
  // loads each -- 12 us for 33 MB.
  const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0u; };
  const bool vec = (P & 3) == 0 && (p0 & 3) == 0 && al16(im) && al16(ds) && al16(gt_im) && al16(gt_depth) &&
                   (!extra_mask || al16(extra_mask)) && (!color_weight || al16(color_weight)) && (!g_im || al16(g_im)) &&
                   (!g_ds || al16(g_ds));
  const int pv = vec ? p0 + ((p1 - p0) & ~3) : p0;           // [p0, pv) in groups of four, [pv, p1) one by one
  for (int i = p0 + 4 * (int)(blockIdx.x * 256u + threadIdx.x); i < pv; i += 4 * (int)(gridDim.x * 256u)) {
    const float4 dz = *reinterpret_cast<const float4*>(ds + i), sl = *reinterpret_cast<const float4*>(ds + P + i),
                 dq = *reinterpret_cast<const float4*>(ds + 2 * (size_t)P + i), gd = *reinterpret_cast<const float4*>(gt_depth + i);
    const float4 em = extra_mask ? *reinterpret_cast<const float4*>(extra_mask + i) : make_float4(1.f, 1.f, 1.f, 1.f);
    float4 xs[3], ys[3], cws[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      xs[c] = *reinterpret_cast<const float4*>(im + (size_t)c * P + i);
      ys[c] = *reinterpret_cast<const float4*>(gt_im + (size_t)c * P + i);
      cws[c] = color_weight ? *reinterpret_cast<const float4*>(color_weight + (size_t)c * P + i) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
    float gd4[4], gi4[3][4];
#define VTGS_L1_PIXEL(k, F)                                                                              \
    {                                                                                                     \
      const float x[3] = {xs[0].F, xs[1].F, xs[2].F}, y[3] = {ys[0].F, ys[1].F, ys[2].F}, cwv[3] = {cws[0].F, cws[1].F, cws[2].F}; \
      float gim[3];                                                                                       \
      pixel(dz.F, sl.F, dq.F, gd.F, em.F, x, y, cwv, gd4[k], gim);                                        \
      gi4[0][k] = gim[0]; gi4[1][k] = gim[1]; gi4[2][k] = gim[2];                                         \
    }
    VTGS_L1_PIXEL(0, x) VTGS_L1_PIXEL(1, y) VTGS_L1_PIXEL(2, z) VTGS_L1_PIXEL(3, w)
#undef VTGS_L1_PIXEL
    if (g_ds) {
      *reinterpret_cast<float4*>(g_ds + i) = make_float4(gd4[0], gd4[1], gd4[2], gd4[3]);
      *reinterpret_cast<float4*>(g_ds + P + i) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(g_ds + 2 * (size_t)P + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (g_im) {
#pragma unroll
      for (int c = 0; c < 3; ++c) *reinterpret_cast<float4*>(g_im + (size_t)c * P + i) = make_float4(gi4[c][0], gi4[c][1], gi4[c][2], gi4[c][3]);
    }
  }
  for (int i = pv + (int)(blockIdx.x * 256u + threadIdx.x); i < p1; i += (int)(gridDim.x * 256u)) {
    const float x[3] = {im[i], im[(size_t)P + i], im[2 * (size_t)P + i]};
    const float y[3] = {gt_im[i], gt_im[(size_t)P + i], gt_im[2 * (size_t)P + i]};
    const float cwv[3] = {color_weight ? color_weight[i] : 1.f, color_weight ? color_weight[(size_t)P + i] : 1.f,
                          color_weight ? color_weight[2 * (size_t)P + i] : 1.f};
    float gds, gim[3];
    pixel(ds[i], ds[P + i], ds[2 * (size_t)P + i], gt_depth[i], extra_mask ? extra_mask[i] : 1.f, x, y, cwv, gds, gim);
    if (g_ds) { g_ds[i] = gds; g_ds[P + i] = 0.f; g_ds[2 * (size_t)P + i] = 0.f; }
    if (g_im) { g_im[i] = gim[0]; g_im[(size_t)P + i] = gim[1]; g_im[2 * (size_t)P + i] = gim[2]; }
  }
  s_im = wave_sum(s_im); s_d = wave_sum(s_d); cnt = wave_sum(cnt);
  if (lane_id() == 0) { red[threadIdx.x >> 6][0] = s_im; red[threadIdx.x >> 6][1] = s_d; red[threadIdx.x >> 6][2] = cnt; }
  __syncthreads();
  if (threadIdx.x < 3) partial[blockIdx.x * 3 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}


// ---- silhouette-threshold sweep (src/vtgaussian_slam.py:472-510): masked squared colour error per candidate ----------
struct SweepThresholds { float c[8]; int n; };

__global__ __launch_bounds__(256) void silhouette_sweep_kernel(const float* __restrict__ im, const float* __restrict__ sil,
                                                               const float* __restrict__ gt_im, const float* __restrict__ gt_depth,
                                                               int P, SweepThresholds th, float* __restrict__ partial,
                                                               int p0 = 0, int p1 = -1) {
  __shared__ float red[4][16];
  float sum[8], cnt[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { sum[k] = 0.f; cnt[k] = 0.f; }
  if (p1 < 0) p1 = P;
  for (int i = p0 + (int)(blockIdx.x * 256u + threadIdx.x); i < p1; i += (int)(gridDim.x * 256u)) {
    const float e0 = gt_im[i] - im[i], e1 = gt_im[(size_t)P + i] - im[(size_t)P + i],
                e2 = gt_im[2 * (size_t)P + i] - im[2 * (size_t)P + i];
    const float sq = e0 * e0 + e1 * e1 + e2 * e2;
    const float s = sil[i];
    const bool valid = gt_depth[i] > 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool m = valid && k < th.n && s > th.c[k];
      sum[k] += m ? sq : 0.f;
      cnt[k] += m ? 1.f : 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float a = wave_sum(sum[k]), b = wave_sum(cnt[k]);
    if (lane_id() == 0) { red[threadIdx.x >> 6][2 * k] = a; red[threadIdx.x >> 6][2 * k + 1] = b; }
  }
  __syncthreads();
  if ((int)threadIdx.x < 2 * th.n)
    partial[(size_t)blockIdx.x * 2 * th.n + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ---- Adam over up to 8 tensors in one launch (src/vtgaussian_slam.py:180-187) ----------------------------------------------
struct AdamLaunch {
  VtgsAdamGroup g[VTGS_ADAM_MAX_GROUPS];
  float step_size_scale;   // 1 / (1 - b1^step)
  float sqrt_bias2;        // sqrt(1 - b2^step)
  float beta1, beta2;
};

// one element of the update; the multiply-adds are spelled out (and contraction is off) so that the float4 form, the scalar
// tail and the row-list kernel produce the same bits whatever the compiler would fuse in each of them
__device__ __forceinline__ void adam_element(const AdamLaunch& a, const VtgsAdamGroup& g, float gr, float& m, float& v, float& p) {
#pragma clang fp contract(off)
  m = fmaf(1.f - a.beta1, gr - m, m);                                 // lerp, like torch
  v = fmaf(a.beta2, v, ((1.f - a.beta2) * gr) * gr);
  const float denom = sqrtf(v) / a.sqrt_bias2 + g.eps;
  p = fmaf(-(g.lr * a.step_size_scale), m / denom, p);
}
// four elements per thread as float4 when the group's four arrays are 16-byte aligned (every torch allocation and the
// padded segments of the fused nodes' gradient blocks are): the one-float form ran at ~3 TB/s (27 us for the five trainable
// floats of 1 M Gaussians); same arithmetic per element
__global__ __launch_bounds__(256) void adam_step_kernel(AdamLaunch a) {
  const VtgsAdamGroup& g = a.g[blockIdx.y];
  const size_t i = ((size_t)blockIdx.x * 256u + threadIdx.x) * 4u;
  if (i >= g.count) return;
  const bool vec = ((reinterpret_cast<uintptr_t>(g.param) | reinterpret_cast<uintptr_t>(g.grad) | reinterpret_cast<uintptr_t>(g.exp_avg) |
                     reinterpret_cast<uintptr_t>(g.exp_avg_sq)) & 15u) == 0u;
  if (vec && i + 4u <= g.count) {
    const float4 gr = *reinterpret_cast<const float4*>(g.grad + i);
    float4 m = *reinterpret_cast<const float4*>(g.exp_avg + i), v = *reinterpret_cast<const float4*>(g.exp_avg_sq + i);
    float4 p = *reinterpret_cast<const float4*>(g.param + i);
    adam_element(a, g, gr.x, m.x, v.x, p.x); adam_element(a, g, gr.y, m.y, v.y, p.y);
    adam_element(a, g, gr.z, m.z, v.z, p.z); adam_element(a, g, gr.w, m.w, v.w, p.w);
    *reinterpret_cast<float4*>(g.exp_avg + i) = m; *reinterpret_cast<float4*>(g.exp_avg_sq + i) = v;
    *reinterpret_cast<float4*>(g.param + i) = p;
    return;
  }
  for (size_t j = i; j < g.count && j < i + 4u; ++j) {
    float m = g.exp_avg[j], v = g.exp_avg_sq[j], p = g.param[j];
    adam_element(a, g, g.grad[j], m, v, p);
    g.exp_avg[j] = m; g.exp_avg_sq[j] = v; g.param[j] = p;
  }
}

struct AdamRows { const int32_t* rows; int32_t n_rows; uint32_t width[VTGS_ADAM_MAX_GROUPS]; };
__global__ __launch_bounds__(256) void adam_step_rows_kernel(AdamLaunch a, AdamRows r) {
  const VtgsAdamGroup& g = a.g[blockIdx.y];
  const uint32_t w = r.width[blockIdx.y];
  const size_t j = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (w == 0u || j >= (size_t)r.n_rows * w) return;
  const size_t i = (size_t)r.rows[j / w] * w + (j % w);
  float m = g.exp_avg[i], v = g.exp_avg_sq[i], p = g.param[i];
  adam_element(a, g, g.grad[i], m, v, p);
  g.exp_avg[i] = m; g.exp_avg_sq[i] = v; g.param[i] = p;
}

// seen = radius > 0 and the running maximum of the screen-space radius (src/vtgaussian_slam.py:681-689): four Gaussians per thread
__device__ __forceinline__ void seen_and_max_radius_block(uint32_t block, int n, const int32_t* __restrict__ radii, float* __restrict__ mx,
                                                          uint8_t* __restrict__ seen) {
  const int i0 = (int)(block * 1024u + threadIdx.x * 4u);
  if (i0 + 3 < n) {
    const int4 r = *reinterpret_cast<const int4*>(radii + i0);
    float4 m = *reinterpret_cast<const float4*>(mx + i0);
    m.x = fmaxf(m.x, (float)r.x); m.y = fmaxf(m.y, (float)r.y); m.z = fmaxf(m.z, (float)r.z); m.w = fmaxf(m.w, (float)r.w);
    *reinterpret_cast<float4*>(mx + i0) = m;
    *reinterpret_cast<uint32_t*>(seen + i0) = (r.x > 0 ? 1u : 0u) | (r.y > 0 ? 0x100u : 0u) | (r.z > 0 ? 0x10000u : 0u) | (r.w > 0 ? 0x1000000u : 0u);
  } else {
    for (int i = i0; i < min(n, i0 + 4); ++i) { mx[i] = fmaxf(mx[i], (float)radii[i]); seen[i] = radii[i] > 0 ? 1 : 0; }
  }
}
__global__ __launch_bounds__(256) void seen_and_max_radius_kernel(int n, const int32_t* __restrict__ radii, float* __restrict__ mx,
                                                                  uint8_t* __restrict__ seen) {
  seen_and_max_radius_block(blockIdx.x, n, radii, mx, seen);
}

// ---- whole loss of get_loss in a handful of launches (src/vtgaussian_slam.py:519-608, 678-679) ----------------------------
// value:    masked_l1_kernel (no gradient images) [+ ssim_forward_kernel] + loss_finalize_kernel
//           out = {loss, mask count, sum |gt_im - im|, sum |gt_depth - depth|, mean SSIM}
// gradient: loss_backward_kernel (depth plane, and the colour planes of the tracking loss)
//           [+ ssim_backward_kernel with the L1 term folded in for the mapping loss]; the upstream gradient is read from
//           device memory, so nothing waits for the host and no element-wise multiply follows.
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ l1_partial, uint32_t l1_rows,
                                                            const float* __restrict__ ssim_partial, uint32_t ssim_rows,
                                                            int mode, float w_im, float w_depth, float numel_im,
                                                            float* __restrict__ out, float l1_coef = 0.8f,
                                                            int raw = 0, int n_seen = 0, const int32_t* __restrict__ radii = nullptr,
                                                            float* __restrict__ max_radius = nullptr, uint8_t* __restrict__ seen = nullptr) {
  // Round 6: get_loss's bookkeeping (seen = radius > 0, the running maximum of the radius, src/vtgaussian_slam.py:681-689) rides in
  // this launch -- workgroups 1 .. ceil(n / 1024) -- instead of in one of its own: both depend on the render alone, and a
  // launch of a few microseconds of work costs ~5 us on this runtime (gpurun_out/r6/slamlate_b_dens.txt).
  if (blockIdx.x > 0) { seen_and_max_radius_block(blockIdx.x - 1u, n_seen, radii, max_radius, seen); return; }
  __shared__ float red[4][4];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (uint32_t r = threadIdx.x; r < l1_rows; r += 256u) { a0 += l1_partial[3 * r]; a1 += l1_partial[3 * r + 1]; a2 += l1_partial[3 * r + 2]; }
  for (uint32_t r = threadIdx.x; r < ssim_rows; r += 256u) a3 += ssim_partial[r];
  a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
  if (lane_id() == 0) { red[threadIdx.x >> 6][0] = a0; red[threadIdx.x >> 6][1] = a1; red[threadIdx.x >> 6][2] = a2; red[threadIdx.x >> 6][3] = a3; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float s_im = red[0][0] + red[1][0] + red[2][0] + red[3][0], s_d = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    const float cnt = red[0][2] + red[1][2] + red[2][2] + red[3][2];
    if (raw) {                                                // a band's additive sums (vtgs_slam_loss_band_sums)
      out[0] = s_im; out[1] = s_d; out[2] = cnt; out[3] = red[0][3] + red[1][3] + red[2][3] + red[3][3];
      out[4] = 0.f; out[5] = 0.f; out[6] = 0.f; out[7] = 0.f;
      return;
    }
    const float ssim = (red[0][3] + red[1][3] + red[2][3] + red[3][3]) / numel_im;
    // the two weighted terms the reference's get_loss reports beside their sum (weighted_losses['im'], ['depth'])
    const float t_im = (mode != 1) ? w_im * s_im : w_im * (l1_coef * s_im / numel_im + 0.2f * (1.f - ssim));
    const float t_d = (mode != 1) ? w_depth * s_d : w_depth * s_d / cnt;                   // tracking: masked SUMS, mapping: means
    out[0] = t_im + t_d; out[1] = cnt; out[2] = s_im; out[3] = s_d; out[4] = ssim; out[5] = t_im; out[6] = t_d; out[7] = 0.f;
  }
}

// mode 0 (tracking): g_im = up w_im sign(im - gt) on the mask, g_ds[0] = up w_depth sign(depth - gt) on the mask.
// mode 1 (mapping):  g_ds[0] = up w_depth / count * sign(depth - gt) on the mask; g_im comes from ssim_backward_kernel.
__global__ __launch_bounds__(256) void loss_backward_kernel(const float* __restrict__ im, const float* __restrict__ ds,
                                                            const float* __restrict__ gt_im, const float* __restrict__ gt_depth,
                                                            int P, float sil_thres, int mode, float w_im, float w_depth,
                                                            const float* __restrict__ upstream, const float* __restrict__ fwd_out,
                                                            float* __restrict__ g_im, float* __restrict__ g_ds,
                                                            const float* __restrict__ extra_mask = nullptr,
                                                            int p0 = 0, int p1 = -1) {
  if (p1 < 0) p1 = P;
  const float up = upstream[0];
  const float cd = (mode != 1) ? up * w_depth : up * w_depth / fwd_out[1];
  const float ci = up * w_im;
  auto mask_of = [&](float depth, float sil, float dsq, float gd, float em) {
    const float unc = dsq - depth * depth;
    bool m = gd > 0.f && depth == depth && unc == unc;
    if (mode != 1) m = m && sil > sil_thres;
    return m && em != 0.f;
  };
  auto gdepth = [&](bool m, float gd, float depth) { const float d = gd - depth; return m ? (d > 0.f ? -cd : (d < 0.f ? cd : 0.f)) : 0.f; };
  auto gcol = [&](bool mc, float y, float x) { const float e = y - x; return mc ? (e > 0.f ? -ci : (e < 0.f ? ci : 0.f)) : 0.f; };
  // four pixels per thread as float4 when the planes allow it (see masked_l1_kernel)
  const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0u; };
  const bool vec = (P & 3) == 0 && (p0 & 3) == 0 && al16(ds) && al16(gt_depth) && al16(g_ds) && (!extra_mask || al16(extra_mask)) &&
                   (mode == 1 || (al16(im) && al16(gt_im) && al16(g_im)));
  const int pv = vec ? p0 + ((p1 - p0) & ~3) : p0;
  for (int i = p0 + 4 * (int)(blockIdx.x * 256u + threadIdx.x); i < pv; i += 4 * (int)(gridDim.x * 256u)) {
    const float4 dz = *reinterpret_cast<const float4*>(ds + i), sl = *reinterpret_cast<const float4*>(ds + P + i),
                 dq = *reinterpret_cast<const float4*>(ds + 2 * (size_t)P + i), gd = *reinterpret_cast<const float4*>(gt_depth + i);
    const float4 em = extra_mask ? *reinterpret_cast<const float4*>(extra_mask + i) : make_float4(1.f, 1.f, 1.f, 1.f);
    const bool m0 = mask_of(dz.x, sl.x, dq.x, gd.x, em.x), m1 = mask_of(dz.y, sl.y, dq.y, gd.y, em.y),
               m2 = mask_of(dz.z, sl.z, dq.z, gd.z, em.z), m3 = mask_of(dz.w, sl.w, dq.w, gd.w, em.w);
    *reinterpret_cast<float4*>(g_ds + i) = make_float4(gdepth(m0, gd.x, dz.x), gdepth(m1, gd.y, dz.y), gdepth(m2, gd.z, dz.z), gdepth(m3, gd.w, dz.w));
    *reinterpret_cast<float4*>(g_ds + P + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(g_ds + 2 * (size_t)P + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mode != 1) {
      const bool c0 = mode == 0 ? m0 : true, c1 = mode == 0 ? m1 : true, c2 = mode == 0 ? m2 : true, c3 = mode == 0 ? m3 : true;   // mode 2: all pixels
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float4 x = *reinterpret_cast<const float4*>(im + (size_t)c * P + i), y = *reinterpret_cast<const float4*>(gt_im + (size_t)c * P + i);
        *reinterpret_cast<float4*>(g_im + (size_t)c * P + i) = make_float4(gcol(c0, y.x, x.x), gcol(c1, y.y, x.y), gcol(c2, y.z, x.z), gcol(c3, y.w, x.w));
      }
    }
  }
  for (int i = pv + (int)(blockIdx.x * 256u + threadIdx.x); i < p1; i += (int)(gridDim.x * 256u)) {
    const float depth = ds[i];
    const bool m = mask_of(depth, ds[P + i], ds[2 * (size_t)P + i], gt_depth[i], extra_mask ? extra_mask[i] : 1.f);
    const bool mc = (mode == 0) ? m : true;                  // mode 2: the colour sum runs over all pixels
    g_ds[i] = gdepth(m, gt_depth[i], depth);
    g_ds[P + i] = 0.f; g_ds[2 * (size_t)P + i] = 0.f;
    if (mode != 1) {
#pragma unroll
      for (int c = 0; c < 3; ++c) g_im[(size_t)c * P + i] = gcol(mc, gt_im[(size_t)c * P + i], im[(size_t)c * P + i]);
    }
  }
}

// One band's share of the loss from its own sums and the sums over all bands ({sum_im, sum_depth, count, sum SSIM}): the
// shares of all bands add up to the full-frame loss of loss_finalize_kernel.  Only the mapping depth term is not additive
// (a masked MEAN: the band's sum over the GLOBAL count); the "1" of (1 - SSIM) goes to the first band.
__global__ void band_share_kernel(const float* __restrict__ own, const float* __restrict__ all, int mode, float w_im,
                                  float w_depth, float numel_im, float l1_coef, int first, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float s_im = own[0], s_d = own[1], cnt = all[2];
  const float t_im = (mode != 1) ? w_im * s_im : w_im * (l1_coef * s_im / numel_im + 0.2f * ((first ? 1.f : 0.f) - own[3] / numel_im));
  const float t_d = (mode != 1) ? w_depth * s_d : w_depth * s_d / cnt;
  out[0] = t_im + t_d; out[1] = cnt; out[2] = s_im; out[3] = s_d; out[4] = all[3] / numel_im; out[5] = t_im; out[6] = t_d; out[7] = 0.f;
}

}  // namespace vtgs

using namespace vtgs;

extern "C" {

uint32_t vtgs_masked_l1_partial_rows(int32_t pixels) {
  if (pixels <= 0) return 0;
  const uint32_t b = (uint32_t)((pixels + 255) / 256);
  return b < 1024u ? b : 1024u;
}

int vtgs_masked_l1(const float* im, const float* depth_sil, const float* gt_im, const float* gt_depth, int32_t pixels,
                   float sil_thres, int32_t mode, float* partial_sums, float* g_im, float* g_depth_sil, void* stream) {
  if (!im || !depth_sil || !gt_im || !gt_depth || !partial_sums || !g_im || !g_depth_sil || pixels <= 0 || (mode < 0 || mode > 2))
    return VTGS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(masked_l1_kernel, dim3(vtgs_masked_l1_partial_rows(pixels)), dim3(256), 0, (hipStream_t)stream, im,
                     depth_sil, gt_im, gt_depth, pixels, sil_thres, mode, partial_sums, g_im, g_depth_sil);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_silhouette_sweep(const float* im, const float* silhouette, const float* gt_im, const float* gt_depth,
                          int32_t pixels, const float* thresholds, int32_t n_thresholds, float* partial_sums, void* stream) {
  if (!im || !silhouette || !gt_im || !gt_depth || !thresholds || !partial_sums || pixels <= 0 || n_thresholds <= 0 ||
      n_thresholds > 8)
    return VTGS_ERR_INVALID_ARGUMENT;
  SweepThresholds th;
  th.n = n_thresholds;
  for (int k = 0; k < 8; ++k) th.c[k] = k < n_thresholds ? thresholds[k] : 0.f;
  hipLaunchKernelGGL(silhouette_sweep_kernel, dim3(vtgs_masked_l1_partial_rows(pixels)), dim3(256), 0, (hipStream_t)stream,
                     im, silhouette, gt_im, gt_depth, pixels, th, partial_sums);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_adam_step(const VtgsAdamGroup* groups, int32_t n_groups, int32_t step, float beta1, float beta2, void* stream) {
  if (!groups || n_groups <= 0 || n_groups > VTGS_ADAM_MAX_GROUPS || step <= 0 || !(beta1 >= 0.f && beta1 < 1.f) ||
      !(beta2 >= 0.f && beta2 < 1.f))
    return VTGS_ERR_INVALID_ARGUMENT;
  AdamLaunch a;
  uint64_t longest = 0;
  for (int k = 0; k < VTGS_ADAM_MAX_GROUPS; ++k) {
    if (k < n_groups) {
      a.g[k] = groups[k];
      if (a.g[k].count && (!a.g[k].param || !a.g[k].grad || !a.g[k].exp_avg || !a.g[k].exp_avg_sq)) return VTGS_ERR_INVALID_ARGUMENT;
      longest = a.g[k].count > longest ? a.g[k].count : longest;
    } else {
      a.g[k] = VtgsAdamGroup{nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0.f};
    }
  }
  if (longest == 0) return VTGS_OK;
  if (longest > (uint64_t)0x7fffffff * 256u) return VTGS_ERR_INVALID_ARGUMENT;
  a.step_size_scale = (float)(1.0 / (1.0 - pow((double)beta1, (double)step)));   // lr / bias_correction1
  a.sqrt_bias2 = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.beta1 = beta1;
  a.beta2 = beta2;
  hipLaunchKernelGGL(adam_step_kernel, dim3((uint32_t)((longest + 1023) / 1024), (uint32_t)n_groups), dim3(256), 0,
                     (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_adam_step_rows(const VtgsAdamGroup* groups, int32_t n_groups, int32_t step, float beta1, float beta2,
                        const int32_t* rows, int32_t n_rows, int32_t n_total_rows, void* stream) {
  if (!groups || n_groups <= 0 || n_groups > VTGS_ADAM_MAX_GROUPS || step <= 0 || !(beta1 >= 0.f && beta1 < 1.f) ||
      !(beta2 >= 0.f && beta2 < 1.f) || n_rows < 0 || n_total_rows <= 0 || (n_rows > 0 && !rows))
    return VTGS_ERR_INVALID_ARGUMENT;
  AdamLaunch a;
  AdamRows r;
  r.rows = rows; r.n_rows = n_rows;
  uint64_t longest = 0;
  for (int k = 0; k < VTGS_ADAM_MAX_GROUPS; ++k) {
    r.width[k] = 0u;
    if (k < n_groups) {
      a.g[k] = groups[k];
      if (a.g[k].count % (uint64_t)n_total_rows) return VTGS_ERR_INVALID_ARGUMENT;
      if (a.g[k].count && (!a.g[k].param || !a.g[k].grad || !a.g[k].exp_avg || !a.g[k].exp_avg_sq)) return VTGS_ERR_INVALID_ARGUMENT;
      r.width[k] = (uint32_t)(a.g[k].count / (uint64_t)n_total_rows);
      const uint64_t work = (uint64_t)n_rows * r.width[k];
      longest = work > longest ? work : longest;
    } else {
      a.g[k] = VtgsAdamGroup{nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0.f};
    }
  }
  if (longest == 0) return VTGS_OK;
  if (longest > (uint64_t)0x7fffffff * 256u) return VTGS_ERR_INVALID_ARGUMENT;
  a.step_size_scale = (float)(1.0 / (1.0 - pow((double)beta1, (double)step)));
  a.sqrt_bias2 = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.beta1 = beta1;
  a.beta2 = beta2;
  hipLaunchKernelGGL(adam_step_rows_kernel, dim3((uint32_t)((longest + 255) / 256), (uint32_t)n_groups), dim3(256), 0,
                     (hipStream_t)stream, a, r);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_seen_and_max_radius(int32_t n, const int32_t* radii, float* max_2d_radius, uint8_t* seen, void* stream) {
  if (n < 0 || (n > 0 && (!radii || !max_2d_radius || !seen))) return VTGS_ERR_INVALID_ARGUMENT;
  if (n == 0) return VTGS_OK;
  hipLaunchKernelGGL(seen_and_max_radius_kernel, dim3((uint32_t)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, n, radii,
                     max_2d_radius, seen);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

// workgroups of an SSIM launch over `tiles` tiles: what the chip holds at once (256 CUs x 3 workgroups of 41 KB LDS), every
// workgroup walking the same number of tiles when it can (ceil(tiles / rounds))
static inline uint32_t ssim_grid(uint32_t tiles, uint32_t per_cu = 3u) {   // (the backward's 37 KB: 4 per CU)
  const uint32_t kResident = 256u * per_cu;
  if (tiles <= kResident) return tiles ? tiles : 1u;
  const uint32_t rounds = (tiles + kResident - 1u) / kResident;
  return (tiles + rounds - 1u) / rounds;
}

uint32_t vtgs_ssim_partial_rows(int32_t channels, int32_t height, int32_t width) {
  if (channels <= 0 || height <= 0 || width <= 0) return 0;
  return (uint32_t)(channels * ((height + kST - 1) / kST) * ((width + kST - 1) / kST));
}

int vtgs_ssim_forward(const float* img1, const float* img2, int32_t channels, int32_t height, int32_t width,
                      float* partial_sums, float* grad_maps, void* stream) {
  if (!img1 || !img2 || !partial_sums || channels <= 0 || height <= 0 || width <= 0) return VTGS_ERR_INVALID_ARGUMENT;
  const int tx = (width + kST - 1) / kST, ty = (height + kST - 1) / kST;
  hipLaunchKernelGGL(ssim_forward_kernel, dim3(ssim_grid((uint32_t)(channels * tx * ty))), dim3(256), 0, (hipStream_t)stream, img1, img2, channels,
                     height, width, tx, ty, partial_sums, grad_maps);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_ssim_backward(const float* img1, const float* img2, const float* grad_maps, const float* upstream,
                       int32_t channels, int32_t height, int32_t width, float* grad_img1, void* stream) {
  if (!img1 || !img2 || !grad_maps || !upstream || !grad_img1 || channels <= 0 || height <= 0 || width <= 0)
    return VTGS_ERR_INVALID_ARGUMENT;
  const int tx = (width + kST - 1) / kST, ty = (height + kST - 1) / kST;
  hipLaunchKernelGGL(ssim_backward_kernel, dim3(ssim_grid((uint32_t)(channels * tx * ty), 4u)), dim3(256), 0, (hipStream_t)stream, img1, img2, grad_maps,
                     upstream, channels, height, width, tx, ty, grad_img1, 1.f, 0.f);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

size_t vtgs_loss_scratch_floats(int32_t height, int32_t width) {
  if (height <= 0 || width <= 0) return 0;
  return (size_t)vtgs_masked_l1_partial_rows(height * width) * 3 + vtgs_ssim_partial_rows(3, height, width);
}

int vtgs_slam_loss_forward(int32_t mode, const float* im, const float* depth_sil, const float* gt_im, const float* gt_depth,
                           int32_t height, int32_t width, float sil_thres, float w_im, float w_depth, float* scratch,
                           float* ssim_grad_maps, float* out5, const float* extra_mask, const float* color_weight,
                           void* stream) {
  return vtgs_slam_loss_forward_seen(mode, im, depth_sil, gt_im, gt_depth, height, width, sil_thres, w_im, w_depth, scratch,
                                     ssim_grad_maps, out5, extra_mask, color_weight, 0, nullptr, nullptr, nullptr, stream);
}

int vtgs_slam_loss_forward_seen(int32_t mode, const float* im, const float* depth_sil, const float* gt_im, const float* gt_depth,
                                int32_t height, int32_t width, float sil_thres, float w_im, float w_depth, float* scratch,
                                float* ssim_grad_maps, float* out5, const float* extra_mask, const float* color_weight,
                                int32_t n, const int32_t* radii, float* max_2d_radius, uint8_t* seen, void* stream) {
  if ((mode < 0 || mode > 2) || !im || !depth_sil || !gt_im || !gt_depth || !scratch || !out5 || height <= 0 || width <= 0)
    return VTGS_ERR_INVALID_ARGUMENT;
  if (n < 0 || (n > 0 && (!radii || !max_2d_radius || !seen))) return VTGS_ERR_INVALID_ARGUMENT;
  if (n > 0 && (((uintptr_t)radii | (uintptr_t)max_2d_radius) & 15u)) return VTGS_ERR_INVALID_ARGUMENT;   // (read as int4 / float4)
  const int32_t P = height * width;
  const uint32_t l1_rows = vtgs_masked_l1_partial_rows(P);
  float* ssim_partial = scratch + (size_t)l1_rows * 3;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(masked_l1_kernel, dim3(l1_rows), dim3(256), 0, st, im, depth_sil, gt_im, gt_depth, P, sil_thres, mode,
                     scratch, (float*)nullptr, (float*)nullptr, extra_mask, mode == 1 ? color_weight : (const float*)nullptr);
  uint32_t ssim_rows = 0;
  if (mode == 1) {
    const int tx = (width + kST - 1) / kST, ty = (height + kST - 1) / kST;
    ssim_rows = (uint32_t)(3 * tx * ty);
    hipLaunchKernelGGL(ssim_forward_kernel, dim3(ssim_grid((uint32_t)(ssim_rows))), dim3(256), 0, st, im, gt_im, 3, height, width, tx, ty,
                       ssim_partial, ssim_grad_maps);
  }
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1u + (uint32_t)((n + 1023) / 1024)), dim3(256), 0, st, scratch, l1_rows, ssim_partial,
                     ssim_rows, mode, w_im, w_depth, (float)((size_t)3 * P), out5, (mode == 1 && color_weight) ? 1.0f : 0.8f, 0,
                     (int)n, radii, max_2d_radius, seen);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_slam_loss_backward(int32_t mode, const float* im, const float* depth_sil, const float* gt_im, const float* gt_depth,
                            int32_t height, int32_t width, float sil_thres, float w_im, float w_depth,
                            const float* ssim_grad_maps, const float* fwd_out5, const float* upstream, float* g_im,
                            float* g_depth_sil, const float* extra_mask, const float* color_weight, void* stream) {
  if ((mode < 0 || mode > 2) || !im || !depth_sil || !gt_im || !gt_depth || !fwd_out5 || !upstream || !g_im || !g_depth_sil ||
      height <= 0 || width <= 0 || (mode == 1 && !ssim_grad_maps))
    return VTGS_ERR_INVALID_ARGUMENT;
  const int32_t P = height * width;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(loss_backward_kernel, dim3(vtgs_masked_l1_partial_rows(P)), dim3(256), 0, st, im, depth_sil, gt_im,
                     gt_depth, P, sil_thres, mode, w_im, w_depth, upstream, fwd_out5, g_im, g_depth_sil, extra_mask);
  if (mode == 1) {
    const int tx = (width + kST - 1) / kST, ty = (height + kST - 1) / kST;
    hipLaunchKernelGGL(ssim_backward_kernel, dim3(ssim_grid((uint32_t)(3 * tx * ty), 4u)), dim3(256), 0, st, im, gt_im, ssim_grad_maps, upstream, 3,
                       height, width, tx, ty, g_im, -0.2f * w_im, (color_weight ? 1.0f : 0.8f) * w_im / (float)((size_t)3 * P),
                       color_weight);
  }
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

// ---- one band of the tile-row partition: additive sums, the band's share, its gradient images -----------
static bool band_ok(int32_t height, int32_t width, int32_t rb, int32_t re) {
  return height > 0 && width > 0 && rb >= 0 && re > rb && re <= height;
}

int vtgs_slam_loss_band_sums(int32_t mode, const float* im, const float* depth_sil, const float* gt_im, const float* gt_depth,
                             int32_t height, int32_t width, int32_t row_begin, int32_t row_end, float sil_thres,
                             float* scratch, float* ssim_grad_maps, float* sums8, const float* extra_mask,
                             const float* color_weight, void* stream) {
  if ((mode < 0 || mode > 2) || !im || !depth_sil || !gt_im || !gt_depth || !scratch || !sums8 ||
      !band_ok(height, width, row_begin, row_end))
    return VTGS_ERR_INVALID_ARGUMENT;
  const int32_t P = height * width, p0 = row_begin * width, p1 = row_end * width;
  const uint32_t l1_rows = vtgs_masked_l1_partial_rows(p1 - p0);
  float* ssim_partial = scratch + (size_t)vtgs_masked_l1_partial_rows(P) * 3;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(masked_l1_kernel, dim3(l1_rows), dim3(256), 0, st, im, depth_sil, gt_im, gt_depth, P, sil_thres, mode,
                     scratch, (float*)nullptr, (float*)nullptr, extra_mask, mode == 1 ? color_weight : (const float*)nullptr,
                     p0, p1);
  uint32_t ssim_rows = 0;
  if (mode == 1) {
    const int tx = (width + kST - 1) / kST, ty0 = row_begin / kST, ty = (row_end - 1) / kST - ty0 + 1;
    ssim_rows = (uint32_t)(3 * tx * ty);
    hipLaunchKernelGGL(ssim_forward_kernel, dim3(ssim_grid((uint32_t)(ssim_rows))), dim3(256), 0, st, im, gt_im, 3, height, width, tx, ty,
                       ssim_partial, ssim_grad_maps, ty0, row_begin, row_end);
  }
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, scratch, l1_rows, ssim_partial, ssim_rows, mode, 0.f,
                     0.f, (float)((size_t)3 * P), sums8, 0.8f, 1);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_slam_loss_band_share(int32_t mode, const float* own_sums8, const float* all_sums8, int32_t height, int32_t width,
                              float w_im, float w_depth, int32_t has_color_weight, int32_t first_band, float* out8,
                              void* stream) {
  if ((mode < 0 || mode > 2) || !own_sums8 || !all_sums8 || !out8 || height <= 0 || width <= 0) return VTGS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(band_share_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, own_sums8, all_sums8, mode, w_im, w_depth,
                     (float)((size_t)3 * height * width), (mode == 1 && has_color_weight) ? 1.0f : 0.8f, first_band, out8);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_slam_loss_band_backward(int32_t mode, const float* im, const float* depth_sil, const float* gt_im,
                                 const float* gt_depth, int32_t height, int32_t width, int32_t row_begin, int32_t row_end,
                                 float sil_thres, float w_im, float w_depth, const float* ssim_grad_maps,
                                 const float* share_out8, const float* upstream, float* g_im, float* g_depth_sil,
                                 const float* extra_mask, const float* color_weight, void* stream) {
  if ((mode < 0 || mode > 2) || !im || !depth_sil || !gt_im || !gt_depth || !share_out8 || !upstream || !g_im || !g_depth_sil ||
      !band_ok(height, width, row_begin, row_end) || (mode == 1 && !ssim_grad_maps))
    return VTGS_ERR_INVALID_ARGUMENT;
  const int32_t P = height * width, p0 = row_begin * width, p1 = row_end * width;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(loss_backward_kernel, dim3(vtgs_masked_l1_partial_rows(p1 - p0)), dim3(256), 0, st, im, depth_sil, gt_im,
                     gt_depth, P, sil_thres, mode, w_im, w_depth, upstream, share_out8, g_im, g_depth_sil, extra_mask, p0, p1);
  if (mode == 1) {
    const int lo = row_begin - kSR > 0 ? row_begin - kSR : 0, hi = row_end + kSR < height ? row_end + kSR : height;
    const int tx = (width + kST - 1) / kST, ty0 = lo / kST, ty = (hi - 1) / kST - ty0 + 1;
    hipLaunchKernelGGL(ssim_backward_kernel, dim3(ssim_grid((uint32_t)(3 * tx * ty), 4u)), dim3(256), 0, st, im, gt_im, ssim_grad_maps, upstream, 3,
                       height, width, tx, ty, g_im, -0.2f * w_im, (color_weight ? 1.0f : 0.8f) * w_im / (float)((size_t)3 * P),
                       color_weight, ty0, row_begin, row_end);
  }
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_silhouette_sweep_band(const float* im, const float* silhouette, const float* gt_im, const float* gt_depth,
                               int32_t pixels, int32_t pixel_begin, int32_t pixel_end, const float* thresholds,
                               int32_t n_thresholds, float* partial_sums, void* stream) {
  if (!im || !silhouette || !gt_im || !gt_depth || !thresholds || !partial_sums || pixels <= 0 || n_thresholds <= 0 ||
      n_thresholds > 8 || pixel_begin < 0 || pixel_end <= pixel_begin || pixel_end > pixels)
    return VTGS_ERR_INVALID_ARGUMENT;
  SweepThresholds th;
  th.n = n_thresholds;
  for (int k = 0; k < 8; ++k) th.c[k] = k < n_thresholds ? thresholds[k] : 0.f;
  hipLaunchKernelGGL(silhouette_sweep_kernel, dim3(vtgs_masked_l1_partial_rows(pixel_end - pixel_begin)), dim3(256), 0,
                     (hipStream_t)stream, im, silhouette, gt_im, gt_depth, pixels, th, partial_sums, pixel_begin, pixel_end);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

}  // extern "C"
