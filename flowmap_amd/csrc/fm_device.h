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

// SIXTEEN values per lane -> their wave totals, transposing as it reduces: lane L ends with the total of value L >> 2.  gfx950's
// v_permlane32_swap / v_permlane16_swap exchange register halves between lane groups, so after one swap and one add a register
// holds value i in one half of the lanes and value i + 8 in the other, each already summed over the lane pair it came from:
// 16 registers -> 8 -> 4, then two select-and-DPP-add rounds inside the rows and the quad's two steps — 35 instructions where
// sixteen separate butterflies take 96.  (Inline asm: this compiler lowers the SECOND result of __builtin_amdgcn_permlane32_swap
// to the first; tools/probes/transpose_reduce_probe.hip.)
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_transpose_sum16(const float (&v)[16]) {
  // the swaps of a round touch disjoint registers: one asm block per round, one leading s_nop for the VALU-write -> lane-swap hazard
  float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4], a5 = v[5], a6 = v[6], a7 = v[7];
  float b0 = v[8], b1 = v[9], b2 = v[10], b3 = v[11], b4 = v[12], b5 = v[13], b6 = v[14], b7 = v[15];
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %8\n\tv_permlane32_swap_b32 %1, %9\n\tv_permlane32_swap_b32 %2, %10\n\tv_permlane32_swap_b32 %3, %11"
               "\n\tv_permlane32_swap_b32 %4, %12\n\tv_permlane32_swap_b32 %5, %13\n\tv_permlane32_swap_b32 %6, %14\n\tv_permlane32_swap_b32 %7, %15"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5),
                 "+v"(b6), "+v"(b7));
  float r0 = a0 + b0, r1 = a1 + b1, r2 = a2 + b2, r3 = a3 + b3, r4 = a4 + b4, r5 = a5 + b5, r6 = a6 + b6, r7 = a7 + b7;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7"
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
  const float s[4] = {r0 + r4, r1 + r5, r2 + r6, r3 + r7};
  const int lane = threadIdx.x & (kWave - 1);
  const bool bit3 = (lane & 8) != 0, bit2 = (lane & 4) != 0;
  // xor 8 inside a row: lanes with bit 3 clear keep s[j], the others s[j+2]; what a lane does not keep goes to its partner
  const float u0 = (bit3 ? s[2] : s[0]) + dpp_move<0x128>(bit3 ? s[0] : s[2]);  // row_ror:8
  const float u1 = (bit3 ? s[3] : s[1]) + dpp_move<0x128>(bit3 ? s[1] : s[3]);
  // the two halves of an 8-lane group (mirror inside the half row pairs lane l with 7 - l: the other quad)
  float w = (bit2 ? u1 : u0) + dpp_move<0x141>(bit2 ? u0 : u1);  // row_half_mirror
  w += dpp_move<0xB1>(w);                                        // quad_perm:[1,0,3,2]
  w += dpp_move<0x4E>(w);                                        // quad_perm:[2,3,0,1]
  return w;  // lane L: the wave total of v[L >> 2]
}

// Block-wide reduction of NV per-thread fp32 partials followed by ONE fp64 atomic per
// value per block (guide §6 G12).  `lds` must hold (blockDim.x/64) * NV doubles.
// Per-thread partials and the in-wave tree are fp32 (a 6-level tree adds less rounding
// than the ≥32-term per-thread sums below it); everything above a wave (cross-wave,
// cross-block) is fp64, so the heavily cancelling pose/intrinsics gradient sums over
// ~1e6 pixels do not lose accuracy as the image grows.  The in-wave tree is the transposing one, sixteen values at a time.
template <int NV>
__device__ __forceinline__ void block_accumulate(const float (&v)[NV], double* lds, double* dst) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
#pragma unroll
  for (int base = 0; base < NV; base += 16) {
    float g[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) g[i] = base + i < NV ? v[base + i] : 0.f;
    const float s = wave_transpose_sum16(g);  // lane L: the total of g[L >> 2]
    if ((lane & 3) == 0 && base + (lane >> 2) < NV) lds[wave * NV + base + (lane >> 2)] = (double)s;
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
