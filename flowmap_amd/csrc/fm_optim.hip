// Adam step for the explicit-depth parameters.
//
// With the loss at the HBM roofline the optimiser is the larger half of an overfit step:
// depth (F·N) + correspondence-weight logits ((F−1)·N) are 275 M fp32 parameters at C1, and
// Adam moves 28 B per parameter (read p, g, m, v; write p, m, v) = 7.7 GB — 1.75x the bytes
// of the fused loss kernel.  One streaming pass, 16-byte non-temporal accesses, nothing kept.
//
// Elements whose gradient AND both moments are exactly zero are fixed points of the update
// (m = v = 0 -> step 0/(0+eps) = 0), so their three stores are skipped — bit-identical, and
// it is the common case for the correspondence weights: only the P Procrustes sample points
// of each pair (1000 of 921 600 at C1) ever receive a gradient (extrinsics_procrustes.py:45-52).
#include "fm_device.h"
#include "fm_math.h"

namespace fm {

typedef float v4f __attribute__((ext_vector_type(4)));

// A/B knobs (tools/adam_microbench.py --lib): FM_ADAM_NO_NT_LOAD / FM_ADAM_NO_NT_STORE drop the
// non-temporal hints, FM_ADAM_BLOCKS_PER_CU sets the grid (grid-stride loop beyond it).
#ifdef FM_ADAM_NO_NT_LOAD
#define FM_ALOAD(p) (*(p))
#else
#define FM_ALOAD(p) __builtin_nontemporal_load(p)
#endif
#ifdef FM_ADAM_NO_NT_STORE
#define FM_ASTORE(v, p) (*(p) = (v))
#else
#define FM_ASTORE(v, p) __builtin_nontemporal_store(v, p)
#endif
#ifndef FM_ADAM_BLOCKS_PER_CU
#define FM_ADAM_BLOCKS_PER_CU 4
#endif

__device__ __forceinline__ void adam_apply(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ exp_avg,
                                           float* __restrict__ exp_avg_sq, long count, const AdamCoef& c, int vec_ok) {
  const long stride = (long)gridDim.x * blockDim.x;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long done = 0;
  if (vec_ok) {
    const long quads = count / 4;
    v4f* p4 = reinterpret_cast<v4f*>(param);
    const v4f* g4 = reinterpret_cast<const v4f*>(grad);
    v4f* m4 = reinterpret_cast<v4f*>(exp_avg);
    v4f* v4 = reinterpret_cast<v4f*>(exp_avg_sq);
    for (long i = tid; i < quads; i += stride) {
      v4f p = FM_ALOAD(p4 + i);
      const v4f g = FM_ALOAD(g4 + i);
      v4f m = FM_ALOAD(m4 + i);
      v4f v = FM_ALOAD(v4 + i);
      bool idle = c.weight_decay == 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        idle = idle && g[e] == 0.f && m[e] == 0.f && v[e] == 0.f;
        float pe = p[e], me = m[e], ve = v[e];
        adam_update(c, pe, g[e], me, ve);
        p[e] = pe; m[e] = me; v[e] = ve;
      }
      if (idle) continue;
      FM_ASTORE(p, p4 + i);
      FM_ASTORE(m, m4 + i);
      FM_ASTORE(v, v4 + i);
    }
    done = quads * 4;
  }
  for (long i = done + tid; i < count; i += stride) {
    float p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
    adam_update(c, p, grad[i], m, v);
    param[i] = p;
    exp_avg[i] = m;
    exp_avg_sq[i] = v;
  }
}

__global__ void __launch_bounds__(256) adam_kernel(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long count,
                                                   AdamCoef c, int vec_ok) {
  adam_apply(param, grad, exp_avg, exp_avg_sq, count, c, vec_ok);
}

// hipGraph-capturable variant: the step number lives in device memory (a captured launch cannot
// carry a host value that changes from replay to replay), the coefficients are derived per thread.
__global__ void __launch_bounds__(256) adam_capturable_kernel(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                                              long count, const float* step, double lr, double beta1, double beta2,
                                                              double eps, double weight_decay, int vec_ok) {
  const AdamCoef c = adam_coefficients((double)step[0], lr, beta1, beta2, eps, weight_decay);
  adam_apply(param, grad, exp_avg, exp_avg_sq, count, c, vec_ok);
}

// Adam on a LIST of elements (the pixels of dL/ddepth that the sparse losses touch, when the fused flow loss has
// already updated every other pixel in its own pass — fm_flow_loss_fused_adam): one thread per listed element.
__global__ void __launch_bounds__(256) adam_elements_kernel(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                                            const int64_t* elements, long count, AdamCoef c) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int64_t e = elements[i];
  float p = param[e], m = exp_avg[e], v = exp_avg_sq[e];
  adam_update(c, p, grad[e], m, v);
  param[e] = p;
  exp_avg[e] = m;
  exp_avg_sq[e] = v;
}

// Zero fill with a bounded footprint: `blocks` workgroups stream 16-byte non-temporal stores over the
// buffer, so the fill can share the GPU with latency-bound kernels on another stream instead of
// flooding every CU with its own workgroups.
__global__ void __launch_bounds__(256) fill_zero_kernel(float* x, long count) {
  const long quads = count >> 2;
  v4f* q = reinterpret_cast<v4f*>(x);
  const v4f z = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (long)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(z, q + i);
  for (long i = (quads << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) x[i] = 0.f;
}

}  // namespace fm

using namespace fm;

extern "C" {

int fm_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long count, long step, double lr, double beta1,
                 double beta2, double eps, double weight_decay, void* stream) {
  FM_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && count >= 0 && step >= 1);
  FM_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0);
  if (count == 0) return FM_OK;
  const AdamCoef c = adam_coefficients((double)step, lr, beta1, beta2, eps, weight_decay);
  auto aligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const int vec_ok = aligned(param) && aligned(grad) && aligned(exp_avg) && aligned(exp_avg_sq);
  long blocks = (count / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256L * FM_ADAM_BLOCKS_PER_CU) blocks = 256L * FM_ADAM_BLOCKS_PER_CU;  // grid-stride beyond that
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, count, c,
                     vec_ok);
  FM_LAUNCH_STATUS();
}

int fm_adam_step_capturable(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long count, const float* step,
                            double lr, double beta1, double beta2, double eps, double weight_decay, void* stream) {
  FM_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && step && count >= 0);
  FM_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0);
  if (count == 0) return FM_OK;
  auto aligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const int vec_ok = aligned(param) && aligned(grad) && aligned(exp_avg) && aligned(exp_avg_sq);
  long blocks = (count / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256L * FM_ADAM_BLOCKS_PER_CU) blocks = 256L * FM_ADAM_BLOCKS_PER_CU;
  hipLaunchKernelGGL(adam_capturable_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, count, step, lr, beta1, beta2, eps, weight_decay, vec_ok);
  FM_LAUNCH_STATUS();
}

int fm_adam_step_elements(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const int64_t* elements, long count, long step,
                          double lr, double beta1, double beta2, double eps, double weight_decay, void* stream) {
  FM_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && count >= 0 && step >= 1 && (elements || count == 0));
  FM_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0);
  if (count == 0) return FM_OK;
  const AdamCoef c = adam_coefficients((double)step, lr, beta1, beta2, eps, weight_decay);
  hipLaunchKernelGGL(adam_elements_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, elements, count, c);
  FM_LAUNCH_STATUS();
}

int fm_fill_zero(float* x, long count, int blocks, void* stream) {
  FM_CHECK_ARG(x && count >= 0 && blocks >= 1 && (reinterpret_cast<uintptr_t>(x) & 15) == 0);
  if (count == 0) return FM_OK;
  long need = (count / 4 + 255) / 256;
  if (need < 1) need = 1;
  if (need > blocks) need = blocks;
  hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)need), dim3(256), 0, (hipStream_t)stream, x, count);
  FM_LAUNCH_STATUS();
}

}  // extern "C"
