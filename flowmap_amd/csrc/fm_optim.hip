// Adam step for the explicit-depth parameters.
//
// With the loss at the HBM roofline the optimiser is the larger half of an overfit step:
// depth (F·N) + correspondence-weight logits ((F−1)·N) are 275 M fp32 parameters at C1, and
// Adam moves 28 B per parameter (read p, g, m, v; write p, m, v) = 7.7 GB — 1.75x the bytes
// of the fused loss kernel.  One streaming pass, 16-byte non-temporal accesses, nothing kept.
#include "fm_device.h"
#include "fm_math.h"

namespace fm {

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ exp_avg,
                                                   float* __restrict__ exp_avg_sq, long count, AdamCoef c, int vec_ok) {
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
      v4f p = __builtin_nontemporal_load(p4 + i);
      const v4f g = __builtin_nontemporal_load(g4 + i);
      v4f m = __builtin_nontemporal_load(m4 + i);
      v4f v = __builtin_nontemporal_load(v4 + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = p[e], me = m[e], ve = v[e];
        adam_update(c, pe, g[e], me, ve);
        p[e] = pe; m[e] = me; v[e] = ve;
      }
      __builtin_nontemporal_store(p, p4 + i);
      __builtin_nontemporal_store(m, m4 + i);
      __builtin_nontemporal_store(v, v4 + i);
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

}  // namespace fm

using namespace fm;

extern "C" {

int fm_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long count, long step, double lr, double beta1,
                 double beta2, double eps, double weight_decay, void* stream) {
  FM_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && count >= 0 && step >= 1);
  FM_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0);
  if (count == 0) return FM_OK;
  // bias corrections in double on the host, exactly as torch.optim.adam._single_tensor_adam
  const double bc1 = 1.0 - std::pow(beta1, (double)step);
  const double bc2 = 1.0 - std::pow(beta2, (double)step);
  AdamCoef c;
  c.one_minus_b1 = (float)(1.0 - beta1);
  c.b2 = (float)beta2;
  c.one_minus_b2 = (float)(1.0 - beta2);
  c.step_size = (float)(lr / bc1);
  c.bc2_sqrt = (float)std::sqrt(bc2);
  c.eps = (float)eps;
  c.weight_decay = (float)weight_decay;
  auto aligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const int vec_ok = aligned(param) && aligned(grad) && aligned(exp_avg) && aligned(exp_avg_sq);
  long blocks = (count / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256L * 16) blocks = 256L * 16;  // 16 blocks per CU, grid-stride beyond that
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, count, c,
                     vec_ok);
  FM_LAUNCH_STATUS();
}

}  // extern "C"
