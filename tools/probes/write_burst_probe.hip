// Does the 7-reads : 1-write mix of the fused flow kernel stream faster when a block WRITES IN A BURST (its outputs staged in LDS and
// stored together at the end of the block) than when every iteration stores its 16 bytes right after its seven loads?  Same block ->
// data mapping as the kernel (a block owns ITERS consecutive groups of 256 float4), trivial arithmetic, non-temporal accesses.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/write_burst_probe.hip -o /tmp/write_burst_probe && /tmp/write_burst_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int ITERS, bool BURST, bool WRITE>
__global__ void __launch_bounds__(256) mix(const v4f* __restrict__ in, v4f* __restrict__ out, long quads) {
  __shared__ v4f stage[BURST ? ITERS * 256 : 1];
  const long base = (long)blockIdx.x * 256 * ITERS;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const long i = base + it * 256 + threadIdx.x;
    if (i >= quads) break;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 7; ++r) acc += __builtin_nontemporal_load(in + r * quads + i);
    if (!WRITE) {
      if (acc.x == 123.456f) out[i] = acc;
    } else if (BURST) {
      stage[it * 256 + threadIdx.x] = acc;
    } else {
      __builtin_nontemporal_store(acc, out + i);
    }
  }
  if (WRITE && BURST) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const long i = base + it * 256 + threadIdx.x;
      if (i < quads) __builtin_nontemporal_store(stage[it * 256 + threadIdx.x], out + i);
    }
  }
}

template <int ITERS, bool BURST, bool WRITE>
static float run(const v4f* in, v4f* out, long quads) {
  const int blocks = (int)((quads + 256L * ITERS - 1) / (256L * ITERS));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix<ITERS, BURST, WRITE>), dim3(blocks), dim3(256), 0, 0, in, out, quads);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}

int main() {
  const long quads = 150L * 230400;  // 150 frames of 720 x 1280 / 4
  v4f *in, *out;
  hipMalloc(&in, sizeof(v4f) * quads * 7);
  hipMalloc(&out, sizeof(v4f) * quads);
  hipMemset(in, 0, sizeof(v4f) * quads * 7);
  const double gb_r = 7.0 * quads * 16 / 1e9, gb_w = quads * 16 / 1e9;
  auto line = [&](const char* name, float ms, bool w) { printf("%-28s %.4f ms  %.0f GB/s\n", name, ms, (gb_r + (w ? gb_w : 0.0)) / (ms * 1e-3)); };
  line("reads only, iters 4", run<4, false, false>(in, out, quads), false);
  line("store each iteration, 4", run<4, false, true>(in, out, quads), true);
  line("burst at block end, 4", run<4, true, true>(in, out, quads), true);
  line("store each iteration, 8", run<8, false, true>(in, out, quads), true);
  line("burst at block end, 8", run<8, true, true>(in, out, quads), true);
  line("burst at block end, 16", run<16, true, true>(in, out, quads), true);
  line("store each iteration, 4", run<4, false, true>(in, out, quads), true);
  line("burst at block end, 4", run<4, true, true>(in, out, quads), true);
  return 0;
}
