// HBM streaming ceilings for read:write mixes (measurement aid, not part of the library).
// R read streams + W write streams of `quads` float4 each, consecutive regions of one buffer.
#include <hip/hip_runtime.h>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int R, int W, bool NT>
__global__ void __launch_bounds__(256) probe(const v4f* __restrict__ in, v4f* __restrict__ out, long quads) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += stride) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < R; ++r) acc += NT ? __builtin_nontemporal_load(in + r * quads + i) : in[r * quads + i];
#pragma unroll
    for (int w = 0; w < W; ++w) {
      if (NT) __builtin_nontemporal_store(acc, out + w * quads + i);
      else out[w * quads + i] = acc;
    }
    if (W == 0 && acc.x == 123.456f) out[i] = acc;  // keep the loads alive
  }
}

// The fused flow kernel's mix (7 x 16 B in, 1 x 16 B out per thread) with the six flow / mask
// vectors of a wave stored as ONE contiguous 6 KB chunk: [chunk][6][64 lanes] float4.
template <bool NT>
__global__ void __launch_bounds__(256) probe_chunked(const v4f* __restrict__ in, v4f* __restrict__ out, long quads) {
  const long stride = (long)gridDim.x * blockDim.x;
  const v4f* packed = in + quads;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += stride) {
    const long chunk = i >> 6, lane = i & 63;
    v4f acc = NT ? __builtin_nontemporal_load(in + i) : in[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const v4f* q = packed + (chunk * 6 + k) * 64 + lane;
      acc += NT ? __builtin_nontemporal_load(q) : *q;
    }
    if (NT) __builtin_nontemporal_store(acc, out + i);
    else out[i] = acc;
  }
}

#define CASE(r, w)                                                                                               \
  if (reads == r && writes == w) {                                                                               \
    if (nt) hipLaunchKernelGGL((probe<r, w, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, quads); \
    else hipLaunchKernelGGL((probe<r, w, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, quads);   \
    return 0;                                                                                                    \
  }

extern "C" int bw_probe(const void* in_, void* out_, long quads, int reads, int writes, int nt, int blocks, void* stream) {
  const v4f* in = (const v4f*)in_;
  v4f* out = (v4f*)out_;
  if (reads == -6) {
    if (nt) hipLaunchKernelGGL((probe_chunked<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, quads);
    else hipLaunchKernelGGL((probe_chunked<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, quads);
    return 0;
  }
  CASE(1, 0) CASE(4, 0) CASE(7, 0) CASE(1, 1) CASE(4, 3) CASE(7, 1) CASE(0, 1) CASE(2, 1) CASE(3, 1)
  return 1;
}
