// Device-side helpers shared by the HIP kernels: wave64 / block reductions and the
// launch-status plumbing of the C ABI.  gfx950 only (wavefront = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fm_math.h"

namespace fm {

constexpr int kWave = 64;

// Sum over the 64 lanes of a wavefront (butterfly; every lane ends with the total).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// Wave64 sum with DPP lane moves only (no LDS traffic, unlike ds_bpermute shuffles):
// quad butterflies, the two row mirrors, then the cross-row broadcasts.  The total is
// valid in lane 63 (and every lane of the last 16-lane row).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v += dpp_mov<0xB1, 0xf>(v);   // quad_perm:[1,0,3,2]
  v += dpp_mov<0x4E, 0xf>(v);   // quad_perm:[2,3,0,1]
  v += dpp_mov<0x141, 0xf>(v);  // row_half_mirror
  v += dpp_mov<0x140, 0xf>(v);  // row_mirror        -> every lane holds its row's sum
  v += dpp_mov<0x142, 0xa>(v);  // row_bcast:15      -> rows 1,3 += rows 0,2
  v += dpp_mov<0x143, 0xc>(v);  // row_bcast:31      -> rows 2,3 += rows 0+1
  return v;
}

// Block-wide reduction of NV per-thread fp32 partials followed by ONE fp64 atomic per
// value per block (guide §6 G12).  `lds` must hold (blockDim.x/64) * NV doubles.
// Per-thread partials and the in-wave tree are fp32 (a 6-level tree adds less rounding
// than the ≥32-term per-thread sums below it); everything above a wave (cross-wave,
// cross-block) is fp64, so the heavily cancelling pose/intrinsics gradient sums over
// ~1e6 pixels do not lose accuracy as the image grows.
template <int NV>
__device__ __forceinline__ void block_accumulate(const float (&v)[NV], double* lds, double* dst) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float s = wave_sum_lane63(v[i]);
    if (lane == kWave - 1) lds[wave * NV + i] = (double)s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double tot = 0.0;
    for (int w = 0; w < nwaves; ++w) tot += lds[w * NV + threadIdx.x];
    if (tot != 0.0) atomicAdd(dst + threadIdx.x, tot);
  }
  __syncthreads();
}

}  // namespace fm

#define FM_OK 0
#define FM_ERR_ARG 1
#define FM_ERR_LAUNCH 2

#define FM_CHECK_ARG(cond) \
  do {                     \
    if (!(cond)) return FM_ERR_ARG; \
  } while (0)

#define FM_LAUNCH_STATUS()                                   \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return FM_ERR_LAUNCH; \
    return FM_OK;                                            \
  } while (0)
