// vtgs_frame.hip -- the reference's pose transform and render-variable builders as one kernel each way.
//
// Replaces, for isotropic maps (log_scales [N,1], every reference config), the ~20 element-wise PyTorch launches of
//   utils/slam_helpers.py:323-385  transform_to_frame            means_cam = R(q/|q|) means + t
//   utils/slam_helpers.py:127-160  transformed_params2rendervar  opacities = sigmoid, scales = exp(tile(log_s)), rot = normalize
//   utils/slam_helpers.py:217-287  get_depth_and_silhouette      colours of the 2nd render = [z, 1, z^2], z in the first-frame camera
// and their autograd backward, including the N -> 12 reduction that carries dL/dmeans_cam back to the pose
// (dL/dt = sum g, dL/dR = sum g p^T; the 12 -> 7 step through the quaternion is done by the caller on 12 floats).
#include "../../include/vtgs.h"
#include "vtgs_internal.h"

namespace vtgs {

// `idx` (may be NULL): row i of the outputs is Gaussian idx[i] of the map -- the owned set of a rank of the tile-row partition
// (vtgs_prepare_frame_owned); the colours are then copied to compact rows too, so that the rasterizer sees n rows of everything.
__global__ __launch_bounds__(256) void prepare_frame_kernel(
    int n, const int32_t* __restrict__ idx, const float* __restrict__ means3D, const float* __restrict__ logit_op,
    const float* __restrict__ log_scales, const float* __restrict__ unnorm_rot, const float* __restrict__ rgb,
    const float* __restrict__ cam_q, const float* __restrict__ cam_t,
    const float* __restrict__ depth_w2c, float* __restrict__ means_cam, float* __restrict__ opac,
    float* __restrict__ scales, float* __restrict__ rot, float* __restrict__ dcol, float* __restrict__ rgb_out,
    int pose_stride = 1, float* __restrict__ pose7_out = nullptr) {
  // pose_stride / pose7_out (round 6, vtgs_prepare_frame_slot): the pose is column t of the reference's [1,4,T] / [1,3,T] camera
  // tensors -- element k of q sits at cam_q[k * T] -- read in place, and workgroup 0 leaves the seven floats contiguous for the
  // backward: the slot-gather launch of round 5 is gone
  const float qs[4] = {cam_q[0], cam_q[pose_stride], cam_q[2 * pose_stride], cam_q[3 * pose_stride]};
  const float ts[3] = {cam_t[0], cam_t[pose_stride], cam_t[2 * pose_stride]};
  const FramePose P = load_pose(qs, ts, depth_w2c);
  if (pose7_out && blockIdx.x == 0 && threadIdx.x < 7) pose7_out[threadIdx.x] = threadIdx.x < 4 ? qs[threadIdx.x] : ts[threadIdx.x - 4];
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  if (i >= n) return;
  const int row = idx ? idx[i] : i;
  const float x = means3D[3 * row], y = means3D[3 * row + 1], z = means3D[3 * row + 2];
  float cx, cy, cz, zz;
  pose_apply(P, x, y, z, cx, cy, cz, zz);
  means_cam[3 * i] = cx; means_cam[3 * i + 1] = cy; means_cam[3 * i + 2] = cz;
  if (opac) {                      // (NULL, kernel-uniform: the render applies the activations itself, VTGS_FORWARD_RAW_ACTIVATIONS)
    opac[i] = 1.f / (1.f + __expf(-logit_op[row]));
    const float s = __expf(log_scales[row]);
    scales[3 * i] = s; scales[3 * i + 1] = s; scales[3 * i + 2] = s;
    const float4 u = reinterpret_cast<const float4*>(unnorm_rot)[row];
    const float un = rsqrtf(fmaxf(u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w, 1e-24f));
    reinterpret_cast<float4*>(rot)[i] = make_float4(u.x * un, u.y * un, u.z * un, u.w * un);
  }
  dcol[3 * i] = zz; dcol[3 * i + 1] = 1.f; dcol[3 * i + 2] = zz * zz;
  if (rgb_out) { rgb_out[3 * i] = rgb[3 * row]; rgb_out[3 * i + 1] = rgb[3 * row + 1]; rgb_out[3 * i + 2] = rgb[3 * row + 2]; }
}

// The pose-dependent half alone (round 6, VTGS_FORWARD_RAW_ACTIVATIONS: the render kernels apply the activations themselves):
// camera-frame means and depth colours, 36 bytes per Gaussian.  One Gaussian per thread is bound by its own chain (load ->
// transform -> store: 3 KB in flight per workgroup, 13.9 us for 1.12 M Gaussians = 2.9 TB/s -- the same through an LDS
// transposition with 16-byte accesses, 13.7 us: not the access pattern).  Here a thread owns FOUR consecutive Gaussians:
// twelve floats = three 16-byte loads in flight, six 16-byte stores.
__global__ __launch_bounds__(256) void prepare_frame_pose_kernel(
    int n, const float* __restrict__ means3D, const float* __restrict__ cam_q, const float* __restrict__ cam_t,
    const float* __restrict__ depth_w2c, float* __restrict__ means_cam, float* __restrict__ dcol, int pose_stride,
    float* __restrict__ pose7_out, uint4* __restrict__ clear, uint32_t clear_words16) {
  // `clear`: the head of the forward's workspace (counters, per-tile list lengths), zeroed here instead of by a fill command of
  // its own in front of project_and_bin -- every command costs the queue ~4.6 us, whatever it does
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < clear_words16; i += gridDim.x * 256u) clear[i] = make_uint4(0u, 0u, 0u, 0u);
  const float qs[4] = {cam_q[0], cam_q[pose_stride], cam_q[2 * pose_stride], cam_q[3 * pose_stride]};
  const float ts[3] = {cam_t[0], cam_t[pose_stride], cam_t[2 * pose_stride]};
  const FramePose P = load_pose(qs, ts, depth_w2c);
  if (pose7_out && blockIdx.x == 0 && threadIdx.x < 7) pose7_out[threadIdx.x] = threadIdx.x < 4 ? qs[threadIdx.x] : ts[threadIdx.x - 4];
  const int g0 = 4 * (int)(blockIdx.x * 256u + threadIdx.x);      // the thread's first Gaussian
  if (g0 >= n) return;
  if (g0 + 4 <= n) {
    const float4* __restrict__ in = reinterpret_cast<const float4*>(means3D + 3 * (size_t)g0);
    const float4 a = in[0], b = in[1], c = in[2];
    const float v[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
    float m[12], d[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float zz;
      pose_apply(P, v[3 * k], v[3 * k + 1], v[3 * k + 2], m[3 * k], m[3 * k + 1], m[3 * k + 2], zz);
      d[3 * k] = zz; d[3 * k + 1] = 1.f; d[3 * k + 2] = zz * zz;
    }
    float4* __restrict__ om = reinterpret_cast<float4*>(means_cam + 3 * (size_t)g0);
    float4* __restrict__ od = reinterpret_cast<float4*>(dcol + 3 * (size_t)g0);
    om[0] = make_float4(m[0], m[1], m[2], m[3]); om[1] = make_float4(m[4], m[5], m[6], m[7]); om[2] = make_float4(m[8], m[9], m[10], m[11]);
    od[0] = make_float4(d[0], d[1], d[2], d[3]); od[1] = make_float4(d[4], d[5], d[6], d[7]); od[2] = make_float4(d[8], d[9], d[10], d[11]);
  } else {
    for (int g = g0; g < n; ++g) {                                 // the map's last (up to three) Gaussians
      float cx, cy, cz, zz;
      pose_apply(P, means3D[3 * g], means3D[3 * g + 1], means3D[3 * g + 2], cx, cy, cz, zz);
      means_cam[3 * g] = cx; means_cam[3 * g + 1] = cy; means_cam[3 * g + 2] = cz;
      dcol[3 * g] = zz; dcol[3 * g + 1] = 1.f; dcol[3 * g + 2] = zz * zz;
    }
  }
}

// Owned sets of the tile-row partition: which Gaussians of the map can meet this rank's rows?  The test is the one
// project_and_bin makes on a band (outside_tile_rows, on the camera-frame mean and the exponentiated scale), here with a margin:
//   mask_out != NULL   mask_out[i] = 1 when Gaussian i could meet the rows with its radius widened by margin_px and its
//                      scale multiplied by growth (the list is built from this, for the poses / scales of the coming iterations)
//   escapes  != NULL   counts (atomically, accumulating) the Gaussians with owned[i] == 0 that could meet the rows NOW
//                      (the caller passes margin 1 px, growth 1): while the count stays 0 every render of the compact list was
//                      exactly the render of the whole map on those rows.
__global__ __launch_bounds__(256) void band_owner_kernel(
    CamScalars cs, const float* __restrict__ Vp, const float* __restrict__ PVp, int n, const float* __restrict__ means3D,
    const float* __restrict__ scales, int scales_are_log, const float* __restrict__ cam_q, const float* __restrict__ cam_t,
    float margin_px, float growth, const uint8_t* __restrict__ owned, uint8_t* __restrict__ mask_out,
    uint32_t* __restrict__ escapes, int32_t* __restrict__ centre_rows) {
  const CamParams cam = load_cam(cs, Vp, PVp);
  float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, t[3] = {0.f, 0.f, 0.f};
  if (cam_q) {                                     // (NULL: the means are already in the camera frame, as the plain operator gets them)
    const float nq = rsqrtf(cam_q[0] * cam_q[0] + cam_q[1] * cam_q[1] + cam_q[2] * cam_q[2] + cam_q[3] * cam_q[3]);
    const float qq[4] = {cam_q[0] * nq, cam_q[1] * nq, cam_q[2] * nq, cam_q[3] * nq};
    quat_to_R(qq, R);
    t[0] = cam_t[0]; t[1] = cam_t[1]; t[2] = cam_t[2];
  }
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  bool touch = false;
  if (i < n) {
    const float x = means3D[3 * i], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
    float mean[3] = {x, y, z};
    if (cam_q) {
      mean[0] = R[0] * x + R[1] * y + R[2] * z + t[0];
      mean[1] = R[3] * x + R[4] * y + R[5] * z + t[1];
      mean[2] = R[6] * x + R[7] * y + R[8] * z + t[2];
    }
    float sc[3];
    if (scales_are_log) { sc[0] = sc[1] = sc[2] = __expf(scales[i]); }
    else { sc[0] = scales[3 * i]; sc[1] = scales[3 * i + 1]; sc[2] = scales[3 * i + 2]; }
    touch = !outside_tile_rows_ext(cam, mean, sc, cam.row8_begin / 2, (cam.row8_end + 1) / 2, margin_px, growth);
    if (mask_out) mask_out[i] = touch ? 1 : 0;
    if (centre_rows) centre_rows[i] = centre_tile_row(cam, mean);
  }
  if (escapes) {
    const unsigned long long b = __ballot(touch && i < n && !owned[i]);
    if (b && lane_id() == 0) atomicAdd(escapes, (uint32_t)__popcll(b));
  }
}

// flags: bit 0 = gradients to the Gaussian geometry (means3D, unnorm_rotations), bit 1 = to the pose,
//        bit 2 = to the appearance (logit_opacities, log_scales; the colour gradient needs no kernel work)
__global__ __launch_bounds__(256) void prepare_frame_backward_kernel(
    int n, int flags, const float* __restrict__ means3D, const float* __restrict__ logit_op,
    const float* __restrict__ log_scales, const float* __restrict__ unnorm_rot, const float* __restrict__ cam_q,
    const float* __restrict__ cam_t, const float* __restrict__ depth_w2c,
    const float* __restrict__ gm_a, const float* __restrict__ gm_b, const float* __restrict__ g_dcol,
    const float* __restrict__ gop_a, const float* __restrict__ gop_b, const float* __restrict__ gsc_a,
    const float* __restrict__ gsc_b, const float* __restrict__ grot_a, const float* __restrict__ grot_b,
    float* __restrict__ g_means3D, float* __restrict__ g_logit, float* __restrict__ g_log_scales,
    float* __restrict__ g_unnorm_rot, float* __restrict__ pose_partials) {
  __shared__ float red[4][12];
  const FramePose P = load_pose(cam_q, cam_t, depth_w2c);
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
  if (i < n) {
    const float x = means3D[3 * i], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
    float cx, cy, cz, zz;
    pose_apply(P, x, y, z, cx, cy, cz, zz);
    const float dz = g_dcol[3 * i] + 2.f * zz * g_dcol[3 * i + 2];
    const bool hb = gm_b != nullptr;          // second set of rasterizer gradients (absent after a dual backward)
    const float g0 = gm_a[3 * i] + (hb ? gm_b[3 * i] : 0.f) + dz * P.zr[0];
    const float g1 = gm_a[3 * i + 1] + (hb ? gm_b[3 * i + 1] : 0.f) + dz * P.zr[1];
    const float g2 = gm_a[3 * i + 2] + (hb ? gm_b[3 * i + 2] : 0.f) + dz * P.zr[2];
    if (flags & 1) {
      g_means3D[3 * i] = P.R[0] * g0 + P.R[3] * g1 + P.R[6] * g2;
      g_means3D[3 * i + 1] = P.R[1] * g0 + P.R[4] * g1 + P.R[7] * g2;
      g_means3D[3 * i + 2] = P.R[2] * g0 + P.R[5] * g1 + P.R[8] * g2;
      const float4 u = reinterpret_cast<const float4*>(unnorm_rot)[i];
      const float4 ga = reinterpret_cast<const float4*>(grot_a)[i];
      const float4 gb = hb ? reinterpret_cast<const float4*>(grot_b)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float gr[4] = {ga.x + gb.x, ga.y + gb.y, ga.z + gb.z, ga.w + gb.w};
      const float un = rsqrtf(fmaxf(u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w, 1e-24f));
      const float r[4] = {u.x * un, u.y * un, u.z * un, u.w * un};
      const float dot = r[0] * gr[0] + r[1] * gr[1] + r[2] * gr[2] + r[3] * gr[3];
      reinterpret_cast<float4*>(g_unnorm_rot)[i] = make_float4((gr[0] - r[0] * dot) * un, (gr[1] - r[1] * dot) * un,
                                                                (gr[2] - r[2] * dot) * un, (gr[3] - r[3] * dot) * un);
    }
    if (flags & 4) {
      const float o = 1.f / (1.f + __expf(-logit_op[i]));
      g_logit[i] = (gop_a[i] + (hb ? gop_b[i] : 0.f)) * o * (1.f - o);
      const float s = __expf(log_scales[i]);
      g_log_scales[i] = s * (gsc_a[3 * i] + gsc_a[3 * i + 1] + gsc_a[3 * i + 2] +
                             (hb ? gsc_b[3 * i] + gsc_b[3 * i + 1] + gsc_b[3 * i + 2] : 0.f));
    }
    if (flags & 2) {
      acc[0] = g0; acc[1] = g1; acc[2] = g2;                                   // dL/dt
      acc[3] = g0 * x; acc[4] = g0 * y; acc[5] = g0 * z;                       // dL/dR row 0
      acc[6] = g1 * x; acc[7] = g1 * y; acc[8] = g1 * z;
      acc[9] = g2 * x; acc[10] = g2 * y; acc[11] = g2 * z;
    }
  }
  if (flags & 2) {                                   // fixed-order block reduction: bitwise reproducible
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = wave_sum(acc[k]);
    const int wv = (int)(threadIdx.x >> 6), l = lane_id();
    if (l == 0)
      for (int k = 0; k < 12; ++k) red[wv][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 12) pose_partials[(size_t)blockIdx.x * 12 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  }
}

// One workgroup: sums the per-workgroup partial rows (fixed order) and takes the 12 -> 7 step:
// dL/dt = sum[0..2];  dL/dq = J_norm^T ( dR/dq_hat : dL/dR )  with q_hat = q/|q|  (utils/slam_external.py:25-42).
__global__ __launch_bounds__(256) void pose_gradient_kernel(const float* __restrict__ partials, uint32_t rows,
                                                            const float* __restrict__ cam_q, float* __restrict__ g_q,
                                                            float* __restrict__ g_t, int frames = 0, int slot = 0,
                                                            float* __restrict__ g_rots = nullptr, float* __restrict__ g_trans = nullptr) {
  // frames > 0 (round 6, vtgs_pose_gradient_slot): the seven floats go straight into the FULL-SIZE gradients of the reference's
  // [1,4,T] / [1,3,T] camera tensors -- zero except column `slot` -- in this launch: the slot-scatter launch of round 5 is gone
  __shared__ float red[4][12];
  __shared__ float s_g7[7];
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
  // (round 6: a row is three 16-byte loads, four rows requested together -- twelve dword loads per row, one row after the other,
  //  took 8 us for the 4,300 rows of a 1.1 M-Gaussian map on ONE workgroup; the order of every sum is unchanged)
  const float4* __restrict__ p4 = reinterpret_cast<const float4*>(partials);
  auto add_row = [&](const float4& a, const float4& b, const float4& c) {
    acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w; acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
    acc[8] += c.x; acc[9] += c.y; acc[10] += c.z; acc[11] += c.w;
  };
  uint32_t r = threadIdx.x;
  for (; r + 768u < rows; r += 1024u) {
    float4 v[12];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const size_t o = (size_t)(r + 256u * j) * 3; v[3 * j] = p4[o]; v[3 * j + 1] = p4[o + 1]; v[3 * j + 2] = p4[o + 2]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) add_row(v[3 * j], v[3 * j + 1], v[3 * j + 2]);
  }
  for (; r < rows; r += 256u) { const size_t o = (size_t)r * 3; add_row(p4[o], p4[o + 1], p4[o + 2]); }
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = wave_sum(acc[k]);
  if (lane_id() == 0)
    for (int k = 0; k < 12; ++k) red[threadIdx.x >> 6][k] = acc[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    float s[12];
    for (int k = 0; k < 12; ++k) s[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    const float* dR = s + 3;                                     // row-major dL/dR
    const float n2 = cam_q[0] * cam_q[0] + cam_q[1] * cam_q[1] + cam_q[2] * cam_q[2] + cam_q[3] * cam_q[3];
    const float inv = rsqrtf(n2);
    const float qr = cam_q[0] * inv, qx = cam_q[1] * inv, qy = cam_q[2] * inv, qz = cam_q[3] * inv;
    float gh[4];                                                 // dL/dq_hat (same expressions as splat_backward's rot)
    gh[0] = 2.f * (-qz * dR[1] + qy * dR[2] + qz * dR[3] - qx * dR[5] - qy * dR[6] + qx * dR[7]);
    gh[1] = 2.f * (qy * dR[1] + qz * dR[2] + qy * dR[3] - 2.f * qx * dR[4] - qr * dR[5] + qz * dR[6] + qr * dR[7] - 2.f * qx * dR[8]);
    gh[2] = 2.f * (-2.f * qy * dR[0] + qx * dR[1] + qr * dR[2] + qx * dR[3] + qz * dR[5] - qr * dR[6] + qz * dR[7] - 2.f * qy * dR[8]);
    gh[3] = 2.f * (-2.f * qz * dR[0] - qr * dR[1] + qx * dR[2] + qr * dR[3] - 2.f * qz * dR[4] + qy * dR[5] + qx * dR[6] + qy * dR[7]);
    const float qh[4] = {qr, qx, qy, qz};
    const float dot = qh[0] * gh[0] + qh[1] * gh[1] + qh[2] * gh[2] + qh[3] * gh[3];
    for (int k = 0; k < 4; ++k) {
      const float gk = (gh[k] - qh[k] * dot) * inv;                       // through q_hat = q / |q|
      if (g_q) g_q[k] = gk;
      s_g7[k] = gk;
    }
    if (g_t) { g_t[0] = s[0]; g_t[1] = s[1]; g_t[2] = s[2]; }
    s_g7[4] = s[0]; s_g7[5] = s[1]; s_g7[6] = s[2];
  }
  if (frames > 0) {
    __syncthreads();
    for (int i = (int)threadIdx.x; i < 7 * frames; i += 256) {
      if (i < 4 * frames) { const int r = i / frames, c = i - r * frames; g_rots[i] = (c == slot) ? s_g7[r] : 0.f; }
      else { const int k = i - 4 * frames, r = k / frames, c = k - r * frames; g_trans[k] = (c == slot) ? s_g7[4 + r] : 0.f; }
    }
  }
}

// Pose-gradient reduction of the tile-row partition: with the camera at the identity, dL/dt = sum g and the
// rotation part is sum p x g (plus sum g_z kept for the depth scale) -- the 7 floats every rank all-reduces.  Two
// launches, fixed summation order.
__global__ __launch_bounds__(256) void pose7_partial_kernel(int n, const float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ partials) {
  __shared__ float red[4][7];
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    const float px = p[3 * i], py = p[3 * i + 1], pz = p[3 * i + 2];
    const float gx = g[3 * i], gy = g[3 * i + 1], gz = g[3 * i + 2];
    acc[0] = gx; acc[1] = gy; acc[2] = gz;
    acc[3] = py * gz - pz * gy; acc[4] = pz * gx - px * gz; acc[5] = px * gy - py * gx;
    acc[6] = gz;
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) acc[k] = wave_sum(acc[k]);
  if (lane_id() == 0)
    for (int k = 0; k < 7; ++k) red[threadIdx.x >> 6][k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 7) partials[(size_t)blockIdx.x * 7 + threadIdx.x] =
      red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void pose7_final_kernel(const float* __restrict__ partials, uint32_t rows, float* __restrict__ out) {
  __shared__ float red[4][7];
  float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (uint32_t r = threadIdx.x; r < rows; r += 256u)
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[k] += partials[(size_t)r * 7 + k];
#pragma unroll
  for (int k = 0; k < 7; ++k) acc[k] = wave_sum(acc[k]);
  if (lane_id() == 0)
    for (int k = 0; k < 7; ++k) red[threadIdx.x >> 6][k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 7) out[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// The camera parameters of the reference are [1, 4, T] / [1, 3, T] tensors holding every frame's pose
// (src/vtgaussian_slam.py:160-167); an iteration uses column t.  Taken with tensor indexing that is two strided copies forward
// and, per tensor, two zero-fills and two slice copies backward -- ten launches of ~5 us around a 7-float gradient.  One
// launch each way instead: pose7 = (q[4], t[3]) of frame t; the backward writes full-size gradients, zero except column t.
__global__ __launch_bounds__(64) void pose_slot_gather_kernel(const float* __restrict__ rots, const float* __restrict__ trans, int T,
                                                             int t, float* __restrict__ pose7) {
  const int i = (int)threadIdx.x;
  if (i < 4) pose7[i] = rots[(size_t)i * T + t];
  else if (i < 7) pose7[i] = trans[(size_t)(i - 4) * T + t];
}
__global__ __launch_bounds__(256) void pose_slot_scatter_kernel(const float* __restrict__ g_q, const float* __restrict__ g_t, int T, int t,
                                                               float* __restrict__ g_rots, float* __restrict__ g_trans) {
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  if (i < 4 * T) { const int r = i / T, c = i - r * T; g_rots[i] = (c == t && g_q) ? g_q[r] : 0.f; }
  else if (i < 7 * T) { const int k = i - 4 * T, r = k / T, c = k - r * T; g_trans[k] = (c == t && g_t) ? g_t[r] : 0.f; }
}

}  // namespace vtgs

using namespace vtgs;

extern "C" {

int vtgs_pose_slot_gather(const float* cam_unnorm_rots, const float* cam_trans, int32_t frames, int32_t t, float* pose7, void* stream) {
  if (!cam_unnorm_rots || !cam_trans || !pose7 || frames <= 0 || t < 0 || t >= frames) return VTGS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(pose_slot_gather_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cam_unnorm_rots, cam_trans, frames, t, pose7);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_pose_slot_scatter(const float* g_q, const float* g_t, int32_t frames, int32_t t, float* g_cam_unnorm_rots, float* g_cam_trans,
                           void* stream) {
  if (!g_cam_unnorm_rots || !g_cam_trans || frames <= 0 || t < 0 || t >= frames) return VTGS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(pose_slot_scatter_kernel, dim3((7 * frames + 255) / 256), dim3(256), 0, (hipStream_t)stream, g_q, g_t, frames, t,
                     g_cam_unnorm_rots, g_cam_trans);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_pose_gradient(const float* pose_partials, uint32_t rows, const float* cam_q, float* g_cam_q, float* g_cam_t,
                       void* stream) {
  if ((rows > 0 && !pose_partials) || !cam_q || !g_cam_q || !g_cam_t) return VTGS_ERR_INVALID_ARGUMENT;   // (rows = 0: an empty map)
  if ((uintptr_t)pose_partials & 15u) return VTGS_ERR_INVALID_ARGUMENT;                                  // (rows are read as three 16-byte words)
  hipLaunchKernelGGL(pose_gradient_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pose_partials, rows, cam_q, g_cam_q,
                     g_cam_t);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_pose_gradient_slot(const float* pose_partials, uint32_t rows, const float* cam_q, int32_t frames, int32_t t,
                            float* g_cam_unnorm_rots, float* g_cam_trans, void* stream) {
  if ((rows > 0 && !pose_partials) || !cam_q || !g_cam_unnorm_rots || !g_cam_trans || frames <= 0 || t < 0 || t >= frames)
    return VTGS_ERR_INVALID_ARGUMENT;
  if ((uintptr_t)pose_partials & 15u) return VTGS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(pose_gradient_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pose_partials, rows, cam_q, (float*)nullptr,
                     (float*)nullptr, (int)frames, (int)t, g_cam_unnorm_rots, g_cam_trans);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

uint32_t vtgs_pose_partial_rows(int32_t n) { return n > 0 ? (uint32_t)((n + 255) / 256) : 0u; }

int vtgs_prepare_frame_slot(int32_t n, const float* means3D, const float* logit_opacities, const float* log_scales,
                            const float* unnorm_rotations, const float* cam_unnorm_rots, const float* cam_trans, int32_t frames,
                            int32_t t, const float* depth_w2c, float* out_means_cam, float* out_opacities, float* out_scales,
                            float* out_rotations, float* out_depth_colors, float* out_pose7, void* clear, size_t clear_bytes,
                            void* stream) {
  if (n < 0 || !cam_unnorm_rots || !cam_trans || !depth_w2c || !out_pose7 || frames <= 0 || t < 0 || t >= frames)
    return VTGS_ERR_INVALID_ARGUMENT;
  if ((clear_bytes && !clear) || ((uintptr_t)clear & 15u) || (clear_bytes & 15u) || clear_bytes > (1ull << 32))
    return VTGS_ERR_INVALID_ARGUMENT;
  if (n == 0) {                                                // an empty map: the pose still has to reach the caller's seven floats
    if (clear_bytes && hipMemsetAsync(clear, 0, clear_bytes, (hipStream_t)stream) != hipSuccess) return VTGS_ERR_HIP;
    return vtgs_pose_slot_gather(cam_unnorm_rots, cam_trans, frames, t, out_pose7, stream);
  }
  const bool lite = !out_opacities && !out_scales && !out_rotations;      // means_cam + depth colours only
  if (!means3D || !out_means_cam || !out_depth_colors) return VTGS_ERR_INVALID_ARGUMENT;
  if (!lite && (!logit_opacities || !log_scales || !unnorm_rotations || !out_opacities || !out_scales || !out_rotations))
    return VTGS_ERR_INVALID_ARGUMENT;
  if (lite && !(((uintptr_t)means3D | (uintptr_t)out_means_cam | (uintptr_t)out_depth_colors) & 15u)) {   // (16-byte accesses)
    hipLaunchKernelGGL(prepare_frame_pose_kernel, dim3((n + 1023) / 1024), dim3(256), 0, (hipStream_t)stream, n, means3D,
                       cam_unnorm_rots + t, cam_trans + t, depth_w2c, out_means_cam, out_depth_colors, (int)frames, out_pose7,
                       (uint4*)clear, (uint32_t)(clear_bytes / 16));
    return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
  }
  if (clear_bytes && hipMemsetAsync(clear, 0, clear_bytes, (hipStream_t)stream) != hipSuccess) return VTGS_ERR_HIP;
  hipLaunchKernelGGL(prepare_frame_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, (const int32_t*)nullptr,
                     means3D, logit_opacities, log_scales, unnorm_rotations, (const float*)nullptr, cam_unnorm_rots + t, cam_trans + t,
                     depth_w2c, out_means_cam, out_opacities, out_scales, out_rotations, out_depth_colors, (float*)nullptr,
                     (int)frames, out_pose7);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_prepare_frame(int32_t n, const float* means3D, const float* logit_opacities, const float* log_scales,
                       const float* unnorm_rotations, const float* cam_q, const float* cam_t, const float* depth_w2c,
                       float* out_means_cam, float* out_opacities, float* out_scales, float* out_rotations,
                       float* out_depth_colors, void* stream) {
  if (n < 0 || !cam_q || !cam_t || !depth_w2c) return VTGS_ERR_INVALID_ARGUMENT;
  if (n == 0) return VTGS_OK;
  if (!means3D || !logit_opacities || !log_scales || !unnorm_rotations || !out_means_cam || !out_opacities || !out_scales ||
      !out_rotations || !out_depth_colors)
    return VTGS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(prepare_frame_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, (const int32_t*)nullptr,
                     means3D, logit_opacities, log_scales, unnorm_rotations, (const float*)nullptr, cam_q, cam_t, depth_w2c,
                     out_means_cam, out_opacities, out_scales, out_rotations, out_depth_colors, (float*)nullptr);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_prepare_frame_owned(int32_t n_owned, const int32_t* owned_idx, const float* means3D, const float* logit_opacities,
                             const float* log_scales, const float* unnorm_rotations, const float* rgb_colors,
                             const float* cam_q, const float* cam_t, const float* depth_w2c, float* out_means_cam,
                             float* out_opacities, float* out_scales, float* out_rotations, float* out_depth_colors,
                             float* out_rgb_colors, void* stream) {
  if (n_owned < 0 || !cam_q || !cam_t || !depth_w2c) return VTGS_ERR_INVALID_ARGUMENT;
  if (n_owned == 0) return VTGS_OK;
  if (!owned_idx || !means3D || !logit_opacities || !log_scales || !unnorm_rotations || !rgb_colors || !out_means_cam ||
      !out_opacities || !out_scales || !out_rotations || !out_depth_colors || !out_rgb_colors)
    return VTGS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(prepare_frame_kernel, dim3((n_owned + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_owned, owned_idx,
                     means3D, logit_opacities, log_scales, unnorm_rotations, rgb_colors, cam_q, cam_t, depth_w2c,
                     out_means_cam, out_opacities, out_scales, out_rotations, out_depth_colors, out_rgb_colors);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_prepare_frame_backward(int32_t n, uint32_t flags, const float* means3D, const float* logit_opacities,
                                const float* log_scales, const float* unnorm_rotations, const float* cam_q,
                                const float* cam_t, const float* depth_w2c, const float* g_means_a, const float* g_means_b,
                                const float* g_depth_colors, const float* g_opac_a, const float* g_opac_b,
                                const float* g_scales_a, const float* g_scales_b, const float* g_rot_a, const float* g_rot_b,
                                float* g_means3D, float* g_logit_opacities, float* g_log_scales, float* g_unnorm_rotations,
                                float* pose_partials, void* stream) {
  if (n < 0 || !cam_q || !cam_t || !depth_w2c || (flags & ~7u)) return VTGS_ERR_INVALID_ARGUMENT;
  if (n == 0 || flags == 0) return VTGS_OK;
  const bool any_b = g_means_b || g_opac_b || g_scales_b || g_rot_b;      // the *_b set is all-or-none
  if (!means3D || !logit_opacities || !log_scales || !unnorm_rotations || !g_means_a || !g_depth_colors ||
      !g_opac_a || !g_scales_a || !g_rot_a || (any_b && (!g_means_b || !g_opac_b || !g_scales_b || !g_rot_b)))
    return VTGS_ERR_INVALID_ARGUMENT;
  if ((flags & 1u) && (!g_means3D || !g_unnorm_rotations)) return VTGS_ERR_INVALID_ARGUMENT;
  if ((flags & 4u) && (!g_logit_opacities || !g_log_scales)) return VTGS_ERR_INVALID_ARGUMENT;
  if ((flags & 2u) && !pose_partials) return VTGS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(prepare_frame_backward_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, (int)flags,
                     means3D, logit_opacities, log_scales, unnorm_rotations, cam_q, cam_t, depth_w2c, g_means_a, g_means_b,
                     g_depth_colors, g_opac_a, g_opac_b, g_scales_a, g_scales_b, g_rot_a, g_rot_b, g_means3D,
                     g_logit_opacities, g_log_scales, g_unnorm_rotations, pose_partials);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_pose7_reduce(int32_t n, const float* points, const float* g_points, float* partials, float* out7, void* stream) {
  if (n < 0 || !out7 || (n > 0 && (!points || !g_points || !partials))) return VTGS_ERR_INVALID_ARGUMENT;
  const uint32_t rows = vtgs_pose_partial_rows(n);
  hipStream_t st = (hipStream_t)stream;
  if (rows) hipLaunchKernelGGL(pose7_partial_kernel, dim3(rows), dim3(256), 0, st, n, points, g_points, partials);
  hipLaunchKernelGGL(pose7_final_kernel, dim3(1), dim3(256), 0, st, partials, rows, out7);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

}  // extern "C"
