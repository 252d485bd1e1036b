// Probe: does wave_transpose_sum16 (csrc/fm_track.hip) give lane L the wave total of value L >> 2?   hipcc --offload-arch=gfx950 -O3 -o /tmp/trp tools/probes/transpose_reduce_probe.hip && /tmp/trp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
constexpr int kWave = 64;
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float swap32_sum(float a, float b) {  // lanes < 32: a summed over (l, l+32); lanes >= 32: b likewise
  // (inline asm: this compiler lowers the SECOND result of __builtin_amdgcn_permlane32_swap / 16_swap to the first — it emitted
  // v_add v, vdst, vdst — tools/probes/transpose_reduce_probe.hip; the s_nop covers the VALU-write -> lane-swap hazard)
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float swap16_sum(float a, float b) {  // even rows: a summed over (row, row+1); odd rows: b likewise
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__global__ void probe(const float* in, float* out, float* stage) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = in[i * 64 + threadIdx.x];
  float r[8], s[4];
  for (int i = 0; i < 8; ++i) r[i] = swap32_sum(v[i], v[i + 8]);
  for (int i = 0; i < 4; ++i) s[i] = swap16_sum(r[i], r[i + 4]);
  const int lane = threadIdx.x & 63;
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
  const float u0 = (b3 ? s[2] : s[0]) + dpp_move<0x128>(b3 ? s[0] : s[2]);
  const float u1 = (b3 ? s[3] : s[1]) + dpp_move<0x128>(b3 ? s[1] : s[3]);
  float w = (b2 ? u1 : u0) + dpp_move<0x141>(b2 ? u0 : u1);
  w += dpp_move<0xB1>(w);
  w += dpp_move<0x4E>(w);
  out[threadIdx.x] = w;
  stage[threadIdx.x] = r[0];
  stage[64 + threadIdx.x] = s[0];
  stage[128 + threadIdx.x] = u0;
}
int main() {
  float h_in[16 * 64], h_out[64], h_stage[192];
  double want[16] = {};
  for (int i = 0; i < 16; ++i)
    for (int l = 0; l < 64; ++l) {
      h_in[i * 64 + l] = (float)((i + 1) * 1000 + l);  // value i: distinct per lane
      want[i] += h_in[i * 64 + l];
    }
  float *d_in, *d_out, *d_stage;
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out)); hipMalloc(&d_stage, sizeof(h_stage));
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d_in, d_out, d_stage);
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  hipMemcpy(h_stage, d_stage, sizeof(h_stage), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const bool ok = std::fabs(h_out[l] - want[l >> 2]) < 1e-3 * want[l >> 2];
    if (!ok) ++bad;
  }
  printf("lane: got (want value lane>>2)\n");
  for (int l = 0; l < 64; l += 4) printf("%2d: %.0f (%.0f)\n", l, h_out[l], want[l >> 2]);
  printf("r0 (after swap32 of v0,v8) lanes 0,31,32,63: %.0f %.0f %.0f %.0f   [v0: lane+1000, v8: lane+9000]\n", h_stage[0], h_stage[31], h_stage[32], h_stage[63]);
  printf("s0 lanes 0,16,32,48: %.0f %.0f %.0f %.0f\n", h_stage[64], h_stage[64 + 16], h_stage[64 + 32], h_stage[64 + 48]);
  printf("u0 lanes 0,8,16,24: %.0f %.0f %.0f %.0f\n", h_stage[128], h_stage[128 + 8], h_stage[128 + 16], h_stage[128 + 24]);
  printf("%s (%d lanes wrong)\n", bad ? "MISMATCH" : "OK", bad);
  return bad != 0;
}
