// vtgs_xcheck.hip -- the launchers of libvtgs_xcheck.so, the TEST-ONLY library of cross-check composites.
//
// libvtgs.so ships the default kernels only.  The independent implementations the GPU tests check them against -- the scalar
// ("readlane") forward and backward, the quad-form matrix-core kernels, the lane = pixel forward, the quadrant-queue backward
// -- are compiled from the same sources with -DVTGS_XCHECK_BUILD=1 into this library (vtgaussian-slam_amd/build.py), which
// libvtgs.so opens next to itself the first time an implementation switch (vtgs_set_option: VTGS_FWD_IMPL / VTGS_BWD_IMPL) asks
// for one of them.  Without it such a request returns VTGS_ERR_INVALID_ARGUMENT; nothing of the product path is in here.
#define VTGS_XCHECK_BUILD 1
#include "vtgs_internal.h"

namespace vtgs {
__global__ void composite_forward(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*,
                                  const GeomRec*, const float*, float*, float*, float*, const Counters*);
template <int WAVES, bool DUAL>
__global__ void composite_forward_mx(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*,
                                     const GeomRec*, const float*, float*, float*, float*, const Counters*, const float*, float*);
template <int WAVES, bool DUAL>
__global__ void composite_forward_px(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*,
                                     const GeomRec*, const float*, float*, float*, float*, const Counters*, const float*, float*);
__global__ void composite_backward(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*,
                                   const uint32_t*, const GeomRec*, const float*, const float*, const float*,
                                   const float*, float*, const Counters*);
template <int WAVES, bool DUAL, bool PX, bool B1>
__global__ void composite_backward_mx(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*,
                                      const uint32_t*, const GeomRec*, const float*, const float*, const float*,
                                      const float*, float*, const Counters*, const float*, const float*, const float*, uint32_t*);
__global__ void composite_backward_q(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*,
                                     const uint32_t*, const uint8_t*, const GeomRec*, const float*, const float*, const float*,
                                     const float*, float*, const Counters*, uint32_t*);
}  // namespace vtgs

using namespace vtgs;

extern "C" {

// must equal the number libvtgs.so was built with: the argument records (CamScalars, GeomRec, Counters) are shared headers
uint32_t vtgs_xcheck_abi_version(void) { return VTGS_ABI_VERSION; }

// impl: 0 scalar, 1 quad form, 2 lane = pixel.  dual != 0: colors_b / out_color_b valid, no depth image (impl 0 has no dual form).
int vtgs_xcheck_forward(int impl, int dual, const CamScalars* cs, const float* bg, uint32_t nblk, const uint32_t* tile_cnt,
                        uint32_t tile_cap, const uint32_t* sorted_gid, const GeomRec* geom, const float* colors, float* out_color,
                        float* out_depth, float* final_T, const Counters* ctr, const float* colors_b, float* out_color_b,
                        void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (dual && impl != 1)
    hipLaunchKernelGGL((composite_forward_px<4, true>), dim3(nblk), dim3(256), 0, st, *cs, bg, nblk, tile_cnt, tile_cap, sorted_gid,
                       geom, colors, out_color, (float*)nullptr, final_T, ctr, colors_b, out_color_b);
  else if (dual)
    hipLaunchKernelGGL((composite_forward_mx<4, true>), dim3(nblk), dim3(256), 0, st, *cs, bg, nblk, tile_cnt, tile_cap, sorted_gid,
                       geom, colors, out_color, (float*)nullptr, final_T, ctr, colors_b, out_color_b);
  else if (impl == 2)
    hipLaunchKernelGGL((composite_forward_px<4, false>), dim3(nblk), dim3(256), 0, st, *cs, bg, nblk, tile_cnt, tile_cap, sorted_gid,
                       geom, colors, out_color, out_depth, final_T, ctr, (const float*)nullptr, (float*)nullptr);
  else if (impl == 1)
    hipLaunchKernelGGL((composite_forward_mx<4, false>), dim3(nblk), dim3(256), 0, st, *cs, bg, nblk, tile_cnt, tile_cap, sorted_gid,
                       geom, colors, out_color, out_depth, final_T, ctr, (const float*)nullptr, (float*)nullptr);
  else
    hipLaunchKernelGGL(composite_forward, dim3(nblk), dim3(256), 0, st, *cs, bg, nblk, tile_cnt, tile_cap, sorted_gid, geom, colors,
                       out_color, out_depth, final_T, ctr);
  return hipGetLastError() == hipSuccess ? 0 : 4;
}

// impl: 0 scalar (single render), 1 quad form, 3 quadrant queues (single render).
int vtgs_xcheck_backward(int impl, int dual, const CamScalars* cs, const float* bg, uint32_t nblk, const uint32_t* tile_cnt,
                         uint32_t tile_cap, const uint32_t* sorted_gid, const uint32_t* sorted_inst, const uint8_t* qmask,
                         const GeomRec* geom, const float* colors, const float* out_color, const float* grad_color,
                         const float* state, float* grad_inst, const Counters* ctr, const float* colors_b,
                         const float* out_color_b, const float* grad_color_b, uint32_t* dbg, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (dual)
    hipLaunchKernelGGL((composite_backward_mx<4, true, false, false>), dim3(nblk), dim3(256), 0, st, *cs, bg, nblk, tile_cnt, tile_cap,
                       sorted_gid, sorted_inst, geom, colors, out_color, grad_color, state, grad_inst, ctr, colors_b, out_color_b,
                       grad_color_b, dbg);
  else if (impl == 3)
    hipLaunchKernelGGL(composite_backward_q, dim3(nblk), dim3(256), 0, st, *cs, bg, nblk, tile_cnt, tile_cap, sorted_gid, sorted_inst,
                       qmask, geom, colors, out_color, grad_color, state, grad_inst, ctr, dbg);
  else if (impl == 1)
    hipLaunchKernelGGL((composite_backward_mx<4, false, false, false>), dim3(nblk), dim3(256), 0, st, *cs, bg, nblk, tile_cnt, tile_cap,
                       sorted_gid, sorted_inst, geom, colors, out_color, grad_color, state, grad_inst, ctr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, dbg);
  else
    hipLaunchKernelGGL(composite_backward, dim3(nblk), dim3(256), 0, st, *cs, bg, nblk, tile_cnt, tile_cap, sorted_gid, sorted_inst,
                       geom, colors, out_color, grad_color, state, grad_inst, ctr);
  return hipGetLastError() == hipSuccess ? 0 : 4;
}

}  // extern "C"
