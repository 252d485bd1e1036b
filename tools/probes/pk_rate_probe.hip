// Does v_pk_fma_f32 deliver two fp32 lanes per issue slot on this part?  Eight independent multiply-add chains per lane, scalar
// (v_fma_f32) against packed (v_pk_fma_f32), enough waves to fill every SIMD; prints the time per instruction per SIMD in cycles at
// the clock the launch ran at (GRBM counters are not needed: both variants run back to back, the ratio is what matters).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/pk_rate_probe.hip -o /tmp/pk_rate_probe && /tmp/pk_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int PACKED>
__global__ void __launch_bounds__(256) chains(float* out, int iters, float a, float b) {
  if (PACKED) {
    v2f x[8];
    for (int i = 0; i < 8; ++i) x[i] = v2f{(float)threadIdx.x + i, (float)i};
    const v2f av = {a, a}, bv = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
}

template <int PACKED>
static double run(float* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(chains<PACKED>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f, 0.001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(chains<PACKED>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f, 0.001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  const int blocks = 256 * 8, iters = 20000;  // 8 blocks of 4 waves per CU: 8 waves per SIMD
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  const double ms_s = run<0>(out, blocks, iters), ms_p = run<1>(out, blocks, iters);
  const double insts_per_simd = (double)blocks * 4 / 1024 * iters * 8;  // wave instructions each SIMD issues
  printf("{\"scalar_fma_ms\": %.3f, \"packed_fma_ms\": %.3f, \"ns_per_wave_instruction_scalar\": %.3f, \"ns_per_wave_instruction_packed\": %.3f, \"packed_over_scalar\": %.3f, "
         "\"fp32_tflops_scalar\": %.1f, \"fp32_tflops_packed\": %.1f}\n",
         ms_s, ms_p, ms_s * 1e6 / insts_per_simd, ms_p * 1e6 / insts_per_simd, ms_p / ms_s,
         (double)blocks * 256 * iters * 8 * 2 / (ms_s * 1e-3) / 1e12, (double)blocks * 256 * iters * 8 * 4 / (ms_p * 1e-3) / 1e12);
  return 0;
}
