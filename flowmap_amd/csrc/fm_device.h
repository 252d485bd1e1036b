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

// Block-wide reduction of NV per-thread fp32 partials followed by ONE fp64 atomic per
// value per block (guide §6 G12).  `lds` must hold (blockDim.x/64) * NV doubles.
// Per-thread partials are fp32; everything above a thread (wave butterfly, cross-wave,
// cross-block) is fp64, so the heavily cancelling pose/intrinsics gradient sums over
// ~1e6 pixels keep ~1e-7 relative accuracy and are reproducible to fp32 rounding.
template <int NV>
__device__ __forceinline__ void block_accumulate(const float (&v)[NV], double* lds, double* dst) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const double s = wave_sum((double)v[i]);
    if (lane == 0) lds[wave * NV + i] = s;
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
