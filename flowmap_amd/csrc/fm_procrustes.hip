// Weighted Procrustes pose fit between adjacent frames, pose chain, and their backward.
//
// Replaces align_surfaces (flowmap/model/projection.py:213-252), align_rigid
// (flowmap/model/procrustes.py:7-51) and get_extrinsics (projection.py:187-210) plus
// what autograd replays for them (index / grid_sampler_2d_backward / svd_backward /
// a Python loop of F-1 matmuls).
//
// Forward, per frame pair i (later = i+1, earlier = i), point j with flat pixel idx_j:
//   p_j = xyz_{i+1}[idx_j]                                   (projection.py:226-227)
//   q_j = bilinear(xyz_i, xy[idx_j] + bwd_flow_i[idx_j])     (:231-242, border padding)
//   w_j = weights_i[idx_j]                                    (:245-249)
//   one pass: raw moments about a per-pair reference point; from them, exactly, the centroids
//   with weights/(Σw+1e-8) (procrustes.py:23-25) and M = Σ w (q−q̄)(p−p̄)ᵀ with RAW weights
//   (procrustes.py:28-32) — moments_add / moments_finish, fm_math.h
//   solve : R = Ũ Ṽᵀ (in-register Jacobi, fp64), t = q̄ − R p̄     (procrustes.py:35-42)
// xyz comes either from an explicit surfaces tensor (function-level API) or is
// recomputed on the fly from depth and K⁻¹ (fused path: surfaces never exist in HBM).
//
// Backward: dL/dT -> (polar differential) dL/dM, dL/dp̄, dL/dq̄ -> per-point gradients,
// scattered with fp32 atomics into dL/ddepth (or dL/dsurfaces) and dL/dweights; the
// K⁻¹ gradient is block-reduced into per-frame fp64 accumulators.
#include "../../include/flowmap_hip.h"
#include "fm_device.h"
#include "fm_pose.h"

namespace fm {

// -DFM_PHASE_CLOCKS (tools/build_variants.sh, tools/phase_clocks.py): thread 0 of the first blocks of the one-block-per-pair /
// one-block-per-frame kernels stamps the shader clock at its phase boundaries.  Not compiled into the product library.
#ifdef FM_PHASE_CLOCKS
constexpr int kPhaseSlots = 12, kPhaseBlocks = 256;
__device__ long long fm_phase_clock_buffer[kPhaseBlocks][kPhaseSlots];
#define FM_PHASE(slot)                                                                                       \
  do {                                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < kPhaseBlocks) fm_phase_clock_buffer[blockIdx.x][slot] = wall_clock64(); \
  } while (0)
#else
#define FM_PHASE(slot) \
  do {                 \
  } while (0)
#endif


struct ProcParams {
  const float* depth;      // (B,F,H,W)            [SRC_DEPTH]
  const float* kinv;       // (B,F,3,3)            [SRC_DEPTH]
  const float* surfaces;   // (B,F,H,W,3)          [SRC_SURF]
  const float* bwd_flow;   // (B,F-1,H,W,2)
  const float* weights;    // (B,F-1,H,W)
  const int64_t* indices;  // (P) flat pixel indices, or null for arange(H*W)
  double* stats;           // (B*(F-1), kStatStride)
  const double* pair_grad; // (B*(F-1), kPairGradStride)       [scatter]
  float* grad_depth;       // (B,F,H,W)    atomically accumulated [scatter, SRC_DEPTH]
  float* grad_surfaces;    // (B,F,H,W,3)  atomically accumulated [scatter, SRC_SURF]
  float* grad_weights;     // (B,F-1,H,W)  atomically accumulated [scatter]
  double* kinv_acc;        // (B*F, 9) fp64 accumulators          [scatter, SRC_DEPTH]
  float* point_grads;      // (B*(F-1)*P, 2, 3) dL/dq, dL/dp per correspondence instead of the depth atomics [planned scatter]
  float* point_weight_grads;  // (B*(F-1)*P) dL/dweight per correspondence instead of the store into grad_weights [planned scatter]
  int frames, height, width;
  long points;
  float weight_sens;       // != 0: `weights` holds logits, w = sigmoid(weight_sens·logit)
  int batch_repeat;        // R >= 1: depth/surfaces/flows/weights (and their gradients) have B/R batch
                           //         entries, shared by R consecutive (kinv, pose) batch entries
  // element strides between frames / batch entries of the INPUT image stacks depth, surfaces, bwd_flow, weights (fm_layout;
  // set by proc_layouts: dense unless the caller described a view).  Gradient buffers and workspaces are always dense.
  long fs[4], bs[4];
};

enum { SRC_DEPTH = 0, SRC_SURF = 1 };

// Per-pair view of the inputs for corr_load (fm_math.h).
template <int SRC>
__device__ __forceinline__ CorrSrc pair_source(const ProcParams& p, size_t pair, int b, int i) {
  const size_t n = (size_t)p.height * p.width;
  const int bd = b / p.batch_repeat;  // batch entry of the image data
  const size_t fe = (size_t)bd * p.frames + i, fl = fe + 1;
  pair = (size_t)bd * (p.frames - 1) + i;
  CorrSrc s;
  (void)fe, (void)fl, (void)n;
  s.depth_e = SRC == SRC_DEPTH ? p.depth + (size_t)bd * p.bs[0] + (size_t)i * p.fs[0] : nullptr;
  s.depth_l = SRC == SRC_DEPTH ? p.depth + (size_t)bd * p.bs[0] + (size_t)(i + 1) * p.fs[0] : nullptr;
  s.surf_e = SRC == SRC_SURF ? p.surfaces + (size_t)bd * p.bs[1] + (size_t)i * p.fs[1] : nullptr;
  s.surf_l = SRC == SRC_SURF ? p.surfaces + (size_t)bd * p.bs[1] + (size_t)(i + 1) * p.fs[1] : nullptr;
  s.bwd_flow = p.bwd_flow + (size_t)bd * p.bs[2] + (size_t)i * p.fs[2];
  s.weights = p.weights + (size_t)bd * p.bs[3] + (size_t)i * p.fs[3];
  s.weight_sens = p.weight_sens;
  s.height = p.height;
  s.width = p.width;
  return s;
}

// The per-pair reference point of the one-pass moments (fm_math.h): the later-frame point of the
// middle sample.
template <int SRC>
__device__ __forceinline__ void pair_shift(const ProcParams& p, const CorrSrc& src, const Mat3& kinv_l, float s[3]) {
  const long mid = p.points / 2;
  later_point(src, kinv_l, p.indices ? (int)p.indices[mid] : (int)mid, s);
}

// get_extrinsics for one batch element by ONE wave of a block whose 256 threads all call this (the barriers are
// block-wide): chunk products, Hillis-Steele scan over 64 chunks in LDS, re-walk (as pose_chain_fwd_kernel).
// LDS layout [matrix entry][lane]: consecutive lanes touch consecutive doubles (with [lane][entry] every lane of a
// wave hit the same bank: the 128-byte stride of a 4x4 double matrix).
__device__ __forceinline__ void pose_chain_by_wave0(const float* r, int steps, float* e, double (*buf)[16][64]) {
  const int t = threadIdx.x;
  const bool on = t < 64;
  const int chunk = (steps + 63) / 64;
  const int lo = t * chunk, hi = min(steps, lo + chunk);
  double prod[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (on) {
    for (int s = lo; s < hi; ++s) {
      double m[16], nxt[16];
      for (int k = 0; k < 16; ++k) m[k] = r[(size_t)s * 16 + k];
      mat4_mul(prod, m, nxt);
      for (int k = 0; k < 16; ++k) prod[k] = nxt[k];
    }
    for (int k = 0; k < 16; ++k) buf[0][k][t] = prod[k];
  }
  __syncthreads();
  int cur = 0;
  for (int off = 1; off < 64; off <<= 1) {
    if (on) {
      double mine[16], out[16];
      for (int k = 0; k < 16; ++k) mine[k] = buf[cur][k][t];
      if (t >= off) {
        double left[16];
        for (int k = 0; k < 16; ++k) left[k] = buf[cur][k][t - off];
        mat4_mul(left, mine, out);
      } else {
        for (int k = 0; k < 16; ++k) out[k] = mine[k];
      }
      for (int k = 0; k < 16; ++k) buf[cur ^ 1][k][t] = out[k];
    }
    __syncthreads();
    cur ^= 1;
  }
  if (on) {
    double run[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (t > 0) {
      for (int k = 0; k < 16; ++k) run[k] = buf[cur][k][t - 1];
    } else {
      for (int k = 0; k < 16; ++k) e[k] = (float)run[k];  // E_0 = I
    }
    for (int s = lo; s < hi; ++s) {
      double m[16], nxt[16];
      for (int k = 0; k < 16; ++k) m[k] = r[(size_t)s * 16 + k];
      mat4_mul(run, m, nxt);
      for (int k = 0; k < 16; ++k) {
        run[k] = nxt[k];
        e[(size_t)(s + 1) * 16 + k] = (float)nxt[k];
      }
    }
  }
  __syncthreads();
}

// The same chain by ONE wave in registers (round 5): the caller lets only threads 0..63 in.  The Hillis-Steele rounds exchange the 16 doubles
// with wave shuffles instead of going through LDS between block-wide barriers — in the one-block-per-pair fit the barriers of a 1024-thread
// block, eight of them, were most of the chain's 8-10 us (profiles/r04_fit_phase_clocks_*.txt: 7.6 us for 19 poses, 10 for 149).  Same
// products in the same order as pose_chain_by_wave0: chunk products, inclusive scan over the 64 chunks, re-walk — bit-identical results.
__device__ __forceinline__ void pose_chain_one_wave(const float* r, int steps, float* e) {
  const int t = threadIdx.x;  // the lane
  const int chunk = (steps + 63) / 64;
  const int lo = t * chunk, hi = min(steps, lo + chunk);
  double prod[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int s = lo; s < hi; ++s) {
    double m[16], nxt[16];
    for (int k = 0; k < 16; ++k) m[k] = r[(size_t)s * 16 + k];
    mat4_mul(prod, m, nxt);
    for (int k = 0; k < 16; ++k) prod[k] = nxt[k];
  }
  for (int off = 1; off < 64; off <<= 1) {
    double left[16], out[16];
    for (int k = 0; k < 16; ++k) left[k] = __shfl_up(prod[k], off, 64);
    mat4_mul(left, prod, out);
    if (t >= off)
      for (int k = 0; k < 16; ++k) prod[k] = out[k];
  }
  double run[16];
  for (int k = 0; k < 16; ++k) run[k] = __shfl_up(prod[k], 1, 64);  // the product of every chunk before this lane's
  if (t == 0) {
    for (int k = 0; k < 16; ++k) {
      run[k] = (k % 5 == 0) ? 1.0 : 0.0;
      e[k] = (float)run[k];  // E_0 = I
    }
  }
  for (int s = lo; s < hi; ++s) {
    double m[16], nxt[16];
    for (int k = 0; k < 16; ++k) m[k] = r[(size_t)s * 16 + k];
    mat4_mul(run, m, nxt);
    for (int k = 0; k < 16; ++k) {
      run[k] = nxt[k];
      e[(size_t)(s + 1) * 16 + k] = (float)nxt[k];
    }
  }
}

// A correspondence's gather chain is index -> (flow, weight, later depth) -> four taps of the earlier depth: three dependent
// round trips to HBM (~2 us each on cold lines).  The chain is split into stages so that BOTH pairs' chains are in flight
// together and the first two stages are issued before the block waits for the pose-solve backward.
struct CorrStage1 {  // what the index leads to
  int idx;
  float fx, fy, w_raw, z_p;
};
struct CorrStage2 {  // the taps and their depths
  Taps taps;
  float z[4];
};

__device__ __forceinline__ CorrStage1 corr_stage1(const CorrSrc& s, int idx) {
  CorrStage1 a;
  a.idx = idx;
  a.fx = s.bwd_flow[2 * (size_t)idx];
  a.fy = s.bwd_flow[2 * (size_t)idx + 1];
  a.w_raw = s.weights[idx];
  a.z_p = s.depth_l[idx];
  return a;
}

__device__ __forceinline__ CorrStage2 corr_stage2(const CorrSrc& s, const CorrStage1& a) {
  CorrStage2 b;
  const PixelRef px = pixel_ref(a.idx, s.height, s.width);
  b.taps = bilinear_taps(px.u + a.fx, px.v + a.fy, s.height, s.width);
#pragma unroll
  for (int k = 0; k < 4; ++k) b.z[k] = b.taps.in[k] ? s.depth_e[tap_row(b.taps, k) * s.width + tap_col(b.taps, k)] : 0.f;
  return b;
}

// The Corr that corr_load (fm_math.h) builds, from the staged loads: the same arithmetic in the same order.
__device__ __forceinline__ Corr corr_assemble(const CorrSrc& s, const Mat3& kinv_e, const Mat3& kinv_l, const CorrStage1& a, const CorrStage2& b) {
  Corr c;
  c.idx = a.idx;
  c.w = a.w_raw;
  if (s.weight_sens != 0.f) c.w = fm_sigmoid<false>(s.weight_sens * c.w);
  c.taps = b.taps;
  const PixelRef px = pixel_ref(a.idx, s.height, s.width);
  c.z_p = a.z_p;
  ray_dir(kinv_l, px.u, px.v, c.ray_p);
  c.p[0] = c.ray_p[0] * c.z_p;
  c.p[1] = c.ray_p[1] * c.z_p;
  c.p[2] = c.ray_p[2] * c.z_p;
  c.q[0] = c.q[1] = c.q[2] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!c.taps.in[k]) continue;
    float ray[3];
    ray_dir(kinv_e, pixel_center(tap_col(c.taps, k), s.width), pixel_center(tap_row(c.taps, k), s.height), ray);
    c.q[0] += (ray[0] * b.z[k]) * c.taps.w[k];
    c.q[1] += (ray[1] * b.z[k]) * c.taps.w[k];
    c.q[2] += (ray[2] * b.z[k]) * c.taps.w[k];
  }
  return c;
}

// A correspondence read through its static tap record (fm_procrustes_fit_chain's tap_records): the scattered loads, then the arithmetic of
// corr_load in the same order.
typedef float corr_v4 __attribute__((ext_vector_type(4)));
struct TapLoad {
  int idx, o0, o1, o2, o3;
  float w, z_p, z0, z1, z2, z3, w0, w1, w2, w3;
};
__device__ __forceinline__ TapLoad tap_load(const CorrSrc& s, int idx, corr_v4 ro, corr_v4 rw) {
  TapLoad t;
  t.idx = idx;
  t.o0 = __float_as_int(ro.x), t.o1 = __float_as_int(ro.y), t.o2 = __float_as_int(ro.z), t.o3 = __float_as_int(ro.w);
  t.w0 = rw.x, t.w1 = rw.y, t.w2 = rw.z, t.w3 = rw.w;
  t.w = s.weights[idx];
  t.z_p = s.depth_l[idx];
  t.z0 = t.o0 >= 0 ? s.depth_e[t.o0] : 0.f;
  t.z1 = t.o1 >= 0 ? s.depth_e[t.o1] : 0.f;
  t.z2 = t.o2 >= 0 ? s.depth_e[t.o2] : 0.f;
  t.z3 = t.o3 >= 0 ? s.depth_e[t.o3] : 0.f;
  return t;
}
__device__ __forceinline__ void tap_add(const CorrSrc& s, const Mat3& kinv_e, int off, float z, float w, float& q0, float& q1, float& q2) {
  if (off < 0) return;
  const int tr = off / s.width, tc = off - tr * s.width;
  float ray[3];
  ray_dir(kinv_e, pixel_center(tc, s.width), pixel_center(tr, s.height), ray);
  q0 += (ray[0] * z) * w;
  q1 += (ray[1] * z) * w;
  q2 += (ray[2] * z) * w;
}
// -> q, p, w of the correspondence (scalars on purpose: a Corr assembled field by field went through scratch memory)
struct CorrCore {
  float q0, q1, q2, p0, p1, p2, w;
  int idx;
};
__device__ __forceinline__ CorrCore tap_finish(const CorrSrc& s, const Mat3& kinv_e, const Mat3& kinv_l, const TapLoad& t) {
  CorrCore c;
  c.idx = t.idx;
  c.w = t.w;
  if (s.weight_sens != 0.f) c.w = fm_sigmoid<false>(s.weight_sens * c.w);
  const PixelRef px = pixel_ref(t.idx, s.height, s.width);
  float ray_p[3];
  ray_dir(kinv_l, px.u, px.v, ray_p);
  c.p0 = ray_p[0] * t.z_p;
  c.p1 = ray_p[1] * t.z_p;
  c.p2 = ray_p[2] * t.z_p;
  c.q0 = c.q1 = c.q2 = 0.f;
  tap_add(s, kinv_e, t.o0, t.z0, t.w0, c.q0, c.q1, c.q2);
  tap_add(s, kinv_e, t.o1, t.z1, t.w1, c.q0, c.q1, c.q2);
  tap_add(s, kinv_e, t.o2, t.z2, t.w2, c.q0, c.q1, c.q2);
  tap_add(s, kinv_e, t.o3, t.z3, t.w3, c.q0, c.q1, c.q2);
  return c;
}
// moments_add (fm_math.h) on the scalars
__device__ __forceinline__ void moments_add_core(const CorrCore& c, const float s[3], float (&acc)[kMomentCount]) {
  const float p0 = c.p0 - s[0], p1 = c.p1 - s[1], p2 = c.p2 - s[2];
  acc[0] += c.w;
  acc[1] = fmaf(c.w, p0, acc[1]);
  acc[2] = fmaf(c.w, p1, acc[2]);
  acc[3] = fmaf(c.w, p2, acc[3]);
  const float qv[3] = {c.q0, c.q1, c.q2};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float wq = c.w * (qv[a] - s[a]);
    acc[4 + a] += wq;
    acc[7 + a * 3 + 0] = fmaf(wq, p0, acc[7 + a * 3 + 0]);
    acc[7 + a * 3 + 1] = fmaf(wq, p1, acc[7 + a * 3 + 1]);
    acc[7 + a * 3 + 2] = fmaf(wq, p2, acc[7 + a * 3 + 2]);
  }
}
__device__ __forceinline__ void corr_record_core(float* corr_out, size_t slot, const CorrCore& c) {
  if (corr_out == nullptr) return;
  corr_v4* o = reinterpret_cast<corr_v4*>(corr_out) + slot * 2;
  corr_v4 lo4, hi4;
  lo4.x = c.q0, lo4.y = c.q1, lo4.z = c.q2, lo4.w = c.p0;
  hi4.x = c.p1, hi4.y = c.p2, hi4.z = c.w, hi4.w = __int_as_float(c.idx);
  o[0] = lo4;
  o[1] = hi4;
}

// (corr_record_core above is what the forward fit leaves behind per correspondence for the one-launch backward,
// fm_procrustes_bwd_planned: 32 bytes — q (3), p (3), w, the pixel index — written and later read as two 16-byte vectors, consecutive
// correspondences by consecutive lanes.  The backward then has no gather chain at all: the forward has just paid for it.  Records
// are written on the static-tap path only: a planned backward implies constant flows and indices, hence tap records.)

// What fm_procrustes_fit_chain adds to the moments kernel: the LAST block of a pair (a counter per pair) turns the
// pair's sums into its pose, clears the sums and the counter for the next step (the workspace is persistent and
// self-cleaning: no memset launch), and the last pair to finish (one more counter) chains the poses into the
// extrinsics — moments, finish + solve and get_extrinsics in one launch instead of a memset and three kernels.
struct FitChain {
  int* counters;  // (pairs + 1) ints, zero between launches; null = plain moments kernel
  float* t_bwd;
  float* t_fwd;
  double* aux;
  float* ext;     // (B, F, 4, 4)
  int batch;
  float* corr_out;  // (pairs·P, 8) or null: q (3), p (3), w, index bits of every correspondence, for fm_procrustes_bwd_planned
  const float* tap_records;  // (pairs·P, 8) or null: STATIC per correspondence (constant flows and indices): the four taps' pixel
};                           // offsets in the earlier frame (int bits, -1 = outside) and their bilinear weights (fm_procrustes_scatter_plan)

// grid: (chunks, B*(F-1)): raw moments of every correspondence into stats[0..15] (fp64 atomics).
template <int SRC>
__global__ void __launch_bounds__(256) procrustes_moments_kernel(ProcParams p, int iters, FitChain fc) {
  __shared__ double red[4 * kMomentCount];
  const size_t pair = blockIdx.y;
  const int b = (int)(pair / (p.frames - 1));
  const int i = (int)(pair % (p.frames - 1));
  Mat3 kinv_e, kinv_l;
  if (SRC == SRC_DEPTH) {
    load_mat3(p.kinv + ((size_t)b * p.frames + i) * 9, kinv_e);
    load_mat3(p.kinv + ((size_t)b * p.frames + i + 1) * 9, kinv_l);
  }
  const CorrSrc src = pair_source<SRC>(p, pair, b, i);
  float shift[3];
  pair_shift<SRC>(p, src, kinv_l, shift);
  float acc[kMomentCount];
#pragma unroll
  for (int k = 0; k < kMomentCount; ++k) acc[k] = 0.f;
  const long base = (long)blockIdx.x * blockDim.x * iters;
  for (int it = 0; it < iters; ++it) {
    const long j = base + (long)it * blockDim.x + threadIdx.x;
    if (j >= p.points) break;
    moments_add(corr_load(src, kinv_e, kinv_l, p.indices ? (int)p.indices[j] : (int)j), shift, acc);
  }
  block_accumulate<kMomentCount>(acc, red, p.stats + pair * kStatStride);
  if (fc.counters == nullptr) return;
  // ---- fused tail (block-uniform branches) ----
  __shared__ int last_of_pair, last_of_all;
  __shared__ double chain_buf[2][16][64];
  __shared__ double pair_stats[kStatStride];
  if (threadIdx.x == 0) {
    __threadfence();  // this block's sums are visible before the counter says so
    last_of_pair = atomicAdd(fc.counters + pair, 1) == (int)gridDim.x - 1;
    last_of_all = 0;
  }
  __syncthreads();
  if (last_of_pair) {  // block-uniform
    // the pair's 16 sums in ONE round trip to L2 (16 lanes, one load each): read by one thread one after the other, each
    // agent-scope load waited for the previous one — 28 of this kernel's 60 us at 149 pairs
    if (threadIdx.x < kStatStride) {
      __threadfence();  // see the other blocks' sums
      double* st = p.stats + pair * kStatStride;
      pair_stats[threadIdx.x] = __hip_atomic_load(st + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      st[threadIdx.x] = 0.0;  // clean for the next launch
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double local[kStatStride];
      for (int k = 0; k < kStatStride; ++k) local[k] = pair_stats[k];
      fc.counters[pair] = 0;
      moments_finish(local, shift);
      pose_solve_one(local, fc.t_bwd + pair * 16, fc.t_fwd ? fc.t_fwd + pair * 16 : nullptr, fc.aux + pair * kAuxStride);
      __threadfence();  // the pose is visible before the global counter says so
      const int pairs = (int)gridDim.y;
      last_of_all = atomicAdd(fc.counters + pairs, 1) == pairs - 1;
      if (last_of_all) fc.counters[pairs] = 0;
    }
    __syncthreads();
  }
  if (!last_of_all || fc.ext == nullptr) return;
  __threadfence();  // see every pair's pose
  for (int bb = 0; bb < fc.batch; ++bb)
    pose_chain_by_wave0(fc.t_bwd + (size_t)bb * (p.frames - 1) * 16, p.frames - 1, fc.ext + (size_t)bb * p.frames * 16, chain_buf);
}

// The same work with ONE block of 1024 threads per pair, for index sets of a few thousand points (the reference's P = 1000):
// the pair's sums never leave the block — wave sums (DPP) -> LDS -> 16 threads add the waves in fp64 -> thread 0 finishes,
// solves and stores the pose — so there are no fp64 atomics, no per-pair counter and no agent-scope fence before the solve.
// At 149 pairs the four-blocks-per-pair form above spent 28 of its 60 us in those 596 + 149 fences (each one writes the
// XCD's L2 back): 63 -> see DESIGN.md §3.2.  What is left is one fence per pair before the global counter that elects the
// block which chains the poses.  `counter`: one int, zero between launches.
template <int SRC>
__global__ void __launch_bounds__(1024) procrustes_fit_pair_kernel(ProcParams p, FitChain fc, int* counter) {
  __shared__ double red[16 * kMomentCount];
  __shared__ double pair_stats[kStatStride];
  __shared__ int last_of_all;
  FM_PHASE(0);  // (tools/phase_clocks_fit.py: entry)
  const size_t pair = blockIdx.x;
  const int b = (int)(pair / (p.frames - 1));
  const int i = (int)(pair % (p.frames - 1));
  Mat3 kinv_e, kinv_l;
  if (SRC == SRC_DEPTH) {
    load_mat3(p.kinv + ((size_t)b * p.frames + i) * 9, kinv_e);
    load_mat3(p.kinv + ((size_t)b * p.frames + i + 1) * 9, kinv_l);
  }
  const CorrSrc src = pair_source<SRC>(p, pair, b, i);
  float shift[3];
  float acc[kMomentCount];
#pragma unroll
  for (int k = 0; k < kMomentCount; ++k) acc[k] = 0.f;
  if (SRC == SRC_DEPTH && p.indices != nullptr && fc.tap_records != nullptr) {
    // Constant flows and indices (the overfit loop, from its second step on): where a correspondence's four taps lie and what they
    // weigh never changes, so it comes from a static record — one coalesced read — and every scattered access of the
    // correspondence (later depth, weight, four tap depths) is issued in ONE dependent round instead of two (index -> flow -> taps)
    // Order of issue: [record, index of this thread's first correspondence, index of the middle sample] -> [its weight, later depth and
    // four tap depths, the middle sample's depth]: two dependent rounds in all (the reference point's own chain rides along)
    const corr_v4* recs = reinterpret_cast<const corr_v4*>(fc.tap_records) + pair * (size_t)p.points * 2;
    const long first = threadIdx.x;
    const bool on = first < p.points;
    const long jf = on ? first : 0;
    const corr_v4 ro0 = recs[jf * 2], rw0 = recs[jf * 2 + 1];
    const int idx0 = (int)p.indices[jf];
    const int idx_mid = (int)p.indices[p.points / 2];
    TapLoad t0 = tap_load(src, idx0, ro0, rw0);
    const float z_mid = src.depth_l[idx_mid];
    {  // later_point (fm_math.h) of the middle sample
      const int row = idx_mid / src.width, col = idx_mid - row * src.width;
      float ray[3];
      ray_dir(kinv_l, pixel_center(col, src.width), pixel_center(row, src.height), ray);
      shift[0] = ray[0] * z_mid; shift[1] = ray[1] * z_mid; shift[2] = ray[2] * z_mid;
      for (int a3 = 0; a3 < 3; ++a3)
        if (!(fabsf(shift[a3]) <= 3.0e38f)) shift[a3] = 0.f;
    }
    if (on) {
      const CorrCore c = tap_finish(src, kinv_e, kinv_l, t0);
      moments_add_core(c, shift, acc);
      corr_record_core(fc.corr_out, pair * (size_t)p.points + (size_t)first, c);
    }
    for (long j = first + blockDim.x; j < p.points; j += blockDim.x) {
      const TapLoad t = tap_load(src, (int)p.indices[j], recs[j * 2], recs[j * 2 + 1]);
      const CorrCore c = tap_finish(src, kinv_e, kinv_l, t);
      moments_add_core(c, shift, acc);
      corr_record_core(fc.corr_out, pair * (size_t)p.points + (size_t)j, c);
    }
  } else if (SRC == SRC_DEPTH && p.indices != nullptr) {
    // The reference point's chain (index -> depth) and the thread's first correspondence's chain (index -> flow, weight, depth ->
    // taps) are issued stage by stage TOGETHER: three dependent round trips to HBM instead of five
    const long first = threadIdx.x;
    const bool on = first < p.points;
    const int idx_mid = (int)p.indices[p.points / 2];
    const int idx_j = on ? (int)p.indices[first] : idx_mid;
    const CorrStage1 a = corr_stage1(src, idx_j);
    float z_mid = src.depth_l[idx_mid];
    const CorrStage2 bb = corr_stage2(src, a);
    {  // later_point (fm_math.h) of the middle sample
      const int row = idx_mid / src.width, col = idx_mid - row * src.width;
      float ray[3];
      ray_dir(kinv_l, pixel_center(col, src.width), pixel_center(row, src.height), ray);
      shift[0] = ray[0] * z_mid; shift[1] = ray[1] * z_mid; shift[2] = ray[2] * z_mid;
      for (int a3 = 0; a3 < 3; ++a3)
        if (!(fabsf(shift[a3]) <= 3.0e38f)) shift[a3] = 0.f;
    }
    if (on) moments_add(corr_assemble(src, kinv_e, kinv_l, a, bb), shift, acc);
    for (long j = first + blockDim.x; j < p.points; j += blockDim.x) moments_add(corr_load(src, kinv_e, kinv_l, (int)p.indices[j]), shift, acc);
  } else {
    pair_shift<SRC>(p, src, kinv_l, shift);
    for (long j = threadIdx.x; j < p.points; j += blockDim.x)
      moments_add(corr_load(src, kinv_e, kinv_l, p.indices ? (int)p.indices[j] : (int)j), shift, acc);
  }
#ifdef FM_PHASE_CLOCKS
  asm volatile("" :: "v"(acc[0]), "v"(acc[kMomentCount - 1]));  // thread 0's gathers have returned and its moments are formed
#endif
  FM_PHASE(1);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < kMomentCount; ++k) {
    const float sum = wave_sum_lane63(acc[k]);
    if (lane == kWave - 1) red[wave * kMomentCount + k] = (double)sum;
  }
  if (threadIdx.x == 0) last_of_all = 0;
  __syncthreads();
  if (threadIdx.x < kStatStride) {
    double tot = 0.0;
    if (threadIdx.x < kMomentCount)
      for (int w = 0; w < nwaves; ++w) tot += red[w * kMomentCount + threadIdx.x];
    pair_stats[threadIdx.x] = tot;
  }
  __syncthreads();
  FM_PHASE(2);  // every wave's sums are in LDS and added
  if (threadIdx.x == 0) {
    double local[kStatStride];
    for (int k = 0; k < kStatStride; ++k) local[k] = pair_stats[k];
    moments_finish(local, shift);
    pose_solve_one(local, fc.t_bwd + pair * 16, fc.t_fwd ? fc.t_fwd + pair * 16 : nullptr, fc.aux + pair * kAuxStride);
    FM_PHASE(3);  // finished moments, solved, pose stored
    if (fc.ext != nullptr) {
      __threadfence();  // the pose is visible before the counter says so
      last_of_all = atomicAdd(counter, 1) == (int)gridDim.x - 1;
      if (last_of_all) *counter = 0;
    }
  }
  __syncthreads();
  FM_PHASE(4);  // fence + counter
  if (!last_of_all || threadIdx.x >= kWave) return;  // (one wave of the last block chains the poses: pose_chain_one_wave)
  __threadfence();  // see every pair's pose
  for (int bb = 0; bb < fc.batch; ++bb)
    pose_chain_one_wave(fc.t_bwd + (size_t)bb * (p.frames - 1) * 16, p.frames - 1, fc.ext + (size_t)bb * p.frames * 16);
  FM_PHASE(5);  // (the last block only: the poses are chained)
}

// One thread per pair: raw moments -> (Σw, Σw·p, Σw·q, M) in place.
template <int SRC>
__global__ void procrustes_moments_finish_kernel(ProcParams p, int pairs) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= pairs) return;
  const int b = pair / (p.frames - 1), i = pair % (p.frames - 1);
  Mat3 kinv_l{};
  if (SRC == SRC_DEPTH) load_mat3(p.kinv + ((size_t)b * p.frames + i + 1) * 9, kinv_l);
  const CorrSrc src = pair_source<SRC>(p, pair, b, i);
  float shift[3];
  pair_shift<SRC>(p, src, kinv_l, shift);
  moments_finish(p.stats + (size_t)pair * kStatStride, shift);
}

// The same, followed by the solve for that pair (fm_procrustes_fit: one launch instead of two).
template <int SRC>
__global__ void procrustes_finish_solve_kernel(ProcParams p, int pairs, float* t_bwd, float* t_fwd, double* aux) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= pairs) return;
  const int b = pair / (p.frames - 1), i = pair % (p.frames - 1);
  Mat3 kinv_l{};
  if (SRC == SRC_DEPTH) load_mat3(p.kinv + ((size_t)b * p.frames + i + 1) * 9, kinv_l);
  const CorrSrc src = pair_source<SRC>(p, pair, b, i);
  float shift[3];
  pair_shift<SRC>(p, src, kinv_l, shift);
  double* st = p.stats + (size_t)pair * kStatStride;
  moments_finish(st, shift);
  pose_solve_one(st, t_bwd + (size_t)pair * 16, t_fwd ? t_fwd + (size_t)pair * 16 : nullptr, aux + (size_t)pair * kAuxStride);
}

// Backward for batch_repeat > 1 (the softmin candidate sweep: R = 60 (K, pose) entries share one
// image pair).  The generic kernel would run one thread per (candidate, point) and have 60 threads
// add atomically to the same depth / weight pixels.  Here a thread owns one point and walks a GROUP
// of candidates, summing their gradients in registers (the taps and the sampled values are the same
// for every candidate; only the rays differ): R/GROUP-fold fewer atomics and contention.
// grid: (chunks of 256 points, image pairs x candidate groups).  One point per thread.
constexpr int kRepeatGroup = 6;

__global__ void __launch_bounds__(256) procrustes_scatter_repeat_kernel(ProcParams p, const double* aux) {
  __shared__ double red[4 * 18];
  const int rep = p.batch_repeat;
  const int groups = (rep + kRepeatGroup - 1) / kRepeatGroup;
  const size_t dpair = blockIdx.y / groups;  // image-data pair
  const int gid = (int)(blockIdx.y % groups);
  const int bd = (int)(dpair / (p.frames - 1)), i = (int)(dpair % (p.frames - 1));
  const int n = p.height * p.width;
  const size_t fe = (size_t)bd * p.frames + i, fl = fe + 1;  // image-data frames
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = j < p.points;
  const int idx = active ? (p.indices ? (int)p.indices[j] : (int)j) : 0;

  float s_gw = 0.f, s_later = 0.f, s_tap[4] = {0.f, 0.f, 0.f, 0.f};
  // Round 6: what a correspondence READS — its flow, weight, later depth, the four taps and their depths — is the same for every candidate (the
  // candidates differ in K⁻¹ and the pose only): the dependent gather chain index -> flow -> taps runs ONCE per thread instead of once per
  // candidate, and the candidates' arithmetic runs as one unrolled stretch without a barrier in it (their per-pair constants — 23 doubles and
  // two 3x3 matrices each, block-uniform — are then loaded ahead of use); the block reductions of the per-candidate intrinsics sums follow.
  // 50.4 -> 41.8 us per launch at 60 candidates x 8192 points (profiles/r06_c1_softmin_rocprofv3_summary.csv).  Same arithmetic per candidate.
  const CorrSrc src0 = pair_source<SRC_DEPTH>(p, (size_t)(bd * rep) * (p.frames - 1) + i, bd * rep, i);
  CorrStage1 st1 = {};
  CorrStage2 st2 = {};
  if (active) {
    st1 = corr_stage1(src0, idx);
    st2 = corr_stage2(src0, st1);
  }
  const Taps taps = st2.taps;
  float accs[kRepeatGroup][18];  // per candidate of the group: [0..8] dKinv later frame, [9..17] dKinv earlier frame
#pragma unroll
  for (int t = 0; t < kRepeatGroup; ++t) {
    const int r = gid * kRepeatGroup + t;
#pragma unroll
    for (int k = 0; k < 18; ++k) accs[t][k] = 0.f;
    if (r >= rep) continue;  // (block-uniform)
    const int b = bd * rep + r;
    const size_t pair = (size_t)b * (p.frames - 1) + i;  // (kinv, pose) pair
    Mat3 kinv_e, kinv_l;
    load_mat3(p.kinv + ((size_t)b * p.frames + i) * 9, kinv_e);
    load_mat3(p.kinv + ((size_t)b * p.frames + i + 1) * 9, kinv_l);
    const double* pg = p.pair_grad + pair * kPairGradStride;
    const double* ax = aux + pair * kAuxStride;
    PairGrad g;
    for (int k = 0; k < 9; ++k) g.gM[k] = (float)pg[k];
    for (int a = 0; a < 3; ++a) {
      g.gqbar[a] = (float)pg[9 + a];
      g.gpbar[a] = (float)pg[12 + a];
      g.pbar[a] = (float)ax[21 + a];
      g.qbar[a] = (float)ax[24 + a];
    }
    g.dbar = (float)pg[15];
    g.inv_wsum = (float)pg[16];
    if (!active) continue;
    float* acc = accs[t];
    const Corr c = corr_assemble(src0, kinv_e, kinv_l, st1, st2);
    float gq[3], gp[3], gw;
    corr_backward(c, g, gq, gp, gw);
    if (p.weight_sens != 0.f) gw *= p.weight_sens * c.w * (1.f - c.w);  // d sigmoid(s·x)/dx
    s_gw += gw;
    s_later += gp[0] * c.ray_p[0] + gp[1] * c.ray_p[1] + gp[2] * c.ray_p[2];
    const int row = idx / p.width, col = idx - row * p.width;
    const float u = pixel_center(col, p.width), v = pixel_center(row, p.height);
    const float zh[3] = {c.z_p * u, c.z_p * v, c.z_p};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int d = 0; d < 3; ++d) acc[a * 3 + d] += gp[a] * zh[d];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!c.taps.in[k]) continue;
      const int tc = tap_col(c.taps, k), tr = tap_row(c.taps, k);
      const float ut = pixel_center(tc, p.width), vt = pixel_center(tr, p.height);
      const float z = st2.z[k];
      float ray[3];
      ray_dir(kinv_e, ut, vt, ray);
      const float wt = c.taps.w[k];
      s_tap[k] += wt * (gq[0] * ray[0] + gq[1] * ray[1] + gq[2] * ray[2]);
      const float zt[3] = {z * ut * wt, z * vt * wt, z * wt};
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[9 + a * 3 + d] += gq[a] * zt[d];
    }
  }
  if (p.kinv_acc) {
#pragma unroll
    for (int t = 0; t < kRepeatGroup; ++t) {
      const int r = gid * kRepeatGroup + t;
      if (r >= rep) break;  // (block-uniform)
      float ordered[18];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        ordered[k] = accs[t][9 + k];
        ordered[9 + k] = accs[t][k];
      }
      block_accumulate<18>(ordered, red, p.kinv_acc + ((size_t)(bd * rep + r) * p.frames + i) * 9);
    }
  }
  if (!active) return;
  if (p.grad_weights) atomicAdd(p.grad_weights + dpair * (size_t)n + idx, s_gw);
  if (p.grad_depth) {
    atomicAdd(p.grad_depth + fl * n + idx, s_later);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (taps.in[k]) atomicAdd(p.grad_depth + fe * n + tap_row(taps, k) * p.width + tap_col(taps, k), s_tap[k]);
  }
}

// ---------------------------------------------------------------------------------
// Dense mode (all H·W pixels are correspondences: `num_points: null`, the reference's
// explicit-depth configuration, config/experiment/ablation_explicit_depth.yaml:11-12), in pixel
// space (fm_math.h "Dense Procrustes"): per correspondence ≈110 instructions forward, no ray, no
// camera-space point.  Three tiled kernels, NO atomics on dL/ddepth or dL/dweights:
//   moments      block = 16x64 tile of LATER-frame pixels + an LDS window of the EARLIER frame's
//                depth around where the tile's samples land (tile displaced by the backward flow at
//                its centre, +-16 rows / +-24 columns; samples outside fall back to global memory)
//   bwd_later    same blocks: dL/dweights (stored), the later pixel's own dL/ddepth (plain +=),
//                the intrinsics sums of both roles
//   bwd_taps     block = 16x64 tile of EARLIER-frame pixels; walks the static list of the later
//                pixels whose bilinear taps land in the tile (flows are constants of the
//                optimisation: the list is built once, fm_procrustes_dense_plan_*), accumulates
//                their tap gradients in an LDS tile and adds the tile to dL/ddepth with plain,
//                coalesced read-modify-writes.  Needs only (flow, weight, z) of the later pixel —
//                dL/dq does not depend on the sampled point.
// ---------------------------------------------------------------------------------
constexpr int kTileH = kDenseTileH, kTileW = kDenseTileW;    // 32 x 64 later-frame pixels per workgroup, 8 per thread (64 x 64: no faster, 3 blocks/CU)
#ifndef FM_DENSE_WIN_ROWS  // margins of the earlier-frame window around the flow-displaced tile (taps outside fall back to global loads)
#define FM_DENSE_WIN_ROWS 16
#define FM_DENSE_WIN_COLS 32
#endif
#ifndef FM_DENSE_UNROLL_MOMENTS
#define FM_DENSE_UNROLL_MOMENTS 2
#endif
#ifndef FM_DENSE_UNROLL_LATER
#define FM_DENSE_UNROLL_LATER 2
#endif
#ifndef FM_DENSE_LATER_BLOCKS
#define FM_DENSE_LATER_BLOCKS 1
#endif
#ifndef FM_DENSE_TAPS_BLOCKS
#define FM_DENSE_TAPS_BLOCKS 1
#endif
constexpr int kWinH = kTileH + FM_DENSE_WIN_ROWS, kWinW = kTileW + FM_DENSE_WIN_COLS;  // +-8 rows, +-16 columns: 18 KB, so the VGPRs and not the LDS set the occupancy (moments 0.88 -> 0.62 ms; +-16 / +-32 was 32 KB)
constexpr int kRowsPerThread = kTileH / (256 / kTileW);
static_assert(256 % kTileW == 0 && kWinW % 4 == 0 && kTileH % (256 / kTileW) == 0, "thread mapping: one column, every (256 / kTileW)-th row");

typedef float v4f __attribute__((ext_vector_type(4)));

// 1-D grid -> (pair, tile).  Workgroups are handed to the 8 XCDs round-robin by linear id and each
// XCD has its own L2, so consecutive ids are sent to the SAME XCD's contiguous run of tiles:
// neighbouring tiles (whose earlier-frame windows overlap) then share an L2.  32-bit arithmetic
// (the host checks pairs·tiles < 2^31): a 64-bit integer division costs hundreds of instructions.
constexpr int kXcds = 8;
struct DenseBlock {
  int pair, tile_x, tile_y;
  bool valid;
};

__device__ __forceinline__ DenseBlock dense_block(int height, int width, unsigned total) {
  const unsigned tiles_x = (width + kTileW - 1) / kTileW, tiles_y = (height + kTileH - 1) / kTileH;
  const unsigned per_xcd = (total + kXcds - 1) / kXcds;
  const unsigned logical = (blockIdx.x % kXcds) * per_xcd + blockIdx.x / kXcds;
  DenseBlock d;
  d.valid = blockIdx.x / kXcds < per_xcd && logical < total;
  const unsigned per_pair = tiles_x * tiles_y;
  const unsigned pair = logical / per_pair, tile = logical - pair * per_pair;
  d.pair = (int)pair;
  d.tile_y = (int)(tile / tiles_x);
  d.tile_x = (int)(tile - (unsigned)d.tile_y * tiles_x);
  return d;
}

// Everything a dense block knows about its pair and tile.
struct DenseCtx {
  const float* depth_e;   // earlier frame (H,W)
  const float* depth_l;   // later frame
  const float* bwd_flow;  // (H,W,2) of the pair
  const float* weights;   // (H,W) of the pair (logits when sens != 0)
  float sens;
  int height, width;
  float fw, fh, rcp_w, rcp_h;
  int tx0, ty0;           // tile origin
  int wx0, wy0;           // window origin in the earlier frame (wx0 % 4 == 0)
  size_t pair, fe;        // pair index, earlier frame index (batch folded in)
};

template <int WH = kWinH, int WW = kWinW>
__device__ __forceinline__ DenseCtx dense_ctx(const ProcParams& p, const DenseBlock& blk, bool window) {
  DenseCtx c;
  const int pairs_per_batch = p.frames - 1;
  const int b = blk.pair / pairs_per_batch, i = blk.pair - b * pairs_per_batch;
  const size_t n = (size_t)p.height * p.width;
  c.pair = (size_t)blk.pair;
  c.fe = (size_t)b * p.frames + i;
  c.depth_e = p.depth + c.fe * n;
  c.depth_l = c.depth_e + n;
  c.bwd_flow = p.bwd_flow + c.pair * n * 2;
  c.weights = p.weights + c.pair * n;
  c.sens = p.weight_sens;
  c.height = p.height;
  c.width = p.width;
  c.fw = (float)p.width;
  c.fh = (float)p.height;
  c.rcp_w = 1.0f / c.fw;
  c.rcp_h = 1.0f / c.fh;
  c.tx0 = blk.tile_x * kTileW;
  c.ty0 = blk.tile_y * kTileH;
  c.wx0 = c.wy0 = 0;
  if (window) {  // the window follows the backward flow at the tile's centre
    const int cx = min(c.tx0 + kTileW / 2, p.width - 1), cy = min(c.ty0 + kTileH / 2, p.height - 1);
    const float2 fl = reinterpret_cast<const float2*>(c.bwd_flow)[(size_t)cy * p.width + cx];
    const float ox = fminf(fmaxf(rintf(fl.x * c.fw), -1.0e6f), 1.0e6f);
    const float oy = fminf(fmaxf(rintf(fl.y * c.fh), -1.0e6f), 1.0e6f);
    c.wx0 = (c.tx0 - (WW - kTileW) / 2 + (ox == ox ? (int)ox : 0)) & ~3;
    c.wy0 = c.ty0 - (WH - kTileH) / 2 + (oy == oy ? (int)oy : 0);
  }
  return c;
}

// Window of the earlier frame's depth plus the pixel-centre coordinates of its columns / rows.
template <int WH, int WW>
struct DenseWindowT {
  static constexpr int kH = WH, kW = WW;
  float z[WH * WW];
  float u[WW];
  float v[WH];
};
typedef DenseWindowT<kWinH, kWinW> DenseWindow;

template <int WH, int WW>
__device__ __forceinline__ void stage_depth_window(const DenseCtx& c, DenseWindowT<WH, WW>& win) {
  constexpr int kWinH = WH, kWinW = WW;  // (shadow the defaults: this window's own size)
  static_assert(WW % 4 == 0 && WW + WH <= 256, "16-byte staging; one thread per coordinate");
  if ((c.width & 3) == 0) {  // 16-byte loads: the window's columns start at a multiple of 4
    for (int i = threadIdx.x; i < kWinH * (kWinW / 4); i += 256) {
      const int r = i / (kWinW / 4), q = i - r * (kWinW / 4);
      const int gy = c.wy0 + r, gx = c.wx0 + q * 4;
      v4f val = {0.f, 0.f, 0.f, 0.f};
      if (gy >= 0 && gy < c.height && gx >= 0 && gx < c.width) val = *reinterpret_cast<const v4f*>(c.depth_e + (size_t)gy * c.width + gx);
      *reinterpret_cast<v4f*>(win.z + r * kWinW + q * 4) = val;
    }
  } else {
    for (int i = threadIdx.x; i < kWinH * kWinW; i += 256) {
      const int r = i / kWinW, q = i - r * kWinW;
      const int gy = c.wy0 + r, gx = c.wx0 + q;
      win.z[i] = (gy >= 0 && gy < c.height && gx >= 0 && gx < c.width) ? c.depth_e[(size_t)gy * c.width + gx] : 0.f;
    }
  }
  if (threadIdx.x < kWinW) win.u[threadIdx.x] = center_fast(c.wx0 + (int)threadIdx.x, c.fw, c.rcp_w);
  else if (threadIdx.x < kWinW + kWinH) win.v[threadIdx.x - kWinW] = center_fast(c.wy0 + (int)threadIdx.x - kWinW, c.fh, c.rcp_h);
}

// The raw inputs of one later pixel (issued one loop iteration ahead of their use).
struct DenseRaw {
  float2 fl;
  float wt, z;
};
__device__ __forceinline__ DenseRaw dense_load(const DenseCtx& c, int idx) {
  DenseRaw r;
  r.fl = reinterpret_cast<const float2*>(c.bwd_flow)[idx];
  r.wt = c.weights[idx];
  r.z = c.depth_l[idx];
  return r;
}

// One dense correspondence before any intrinsics: g, h, w and its taps.  The four tap depths and their
// coordinates come from the LDS window when the 2x2 footprint lies inside it, else from global memory.
struct DensePixel {
  float g[3], h[3], w;
  Taps taps;
  float u0, u1, v0, v1;  // pixel-centre coordinates of the taps' columns (west, east) and rows (north, south)
  int cell;              // the north-west tap's cell in the window, -1 when the footprint is not inside it
};

template <int WH, int WW>
__device__ __forceinline__ DensePixel dense_pixel(const DenseCtx& c, const DenseWindowT<WH, WW>& win, const DenseRaw& in, float u, float v) {
  constexpr int kWinH = WH, kWinW = WW;
  DensePixel o;
  o.w = c.sens != 0.f ? fm_sigmoid<true>(c.sens * in.wt) : in.wt;
  o.g[0] = in.z * u;
  o.g[1] = in.z * v;
  o.g[2] = in.z;
  o.taps = dense_taps(u + in.fl.x, v + in.fl.y, c.height, c.width);
  const int wr = o.taps.y0 - c.wy0, wc = o.taps.x0 - c.wx0;
  float zt[4], u0, u1, v0, v1;
  if ((unsigned)wr < (unsigned)(kWinH - 1) && (unsigned)wc < (unsigned)(kWinW - 1)) {
    const float* zw = win.z + wr * kWinW + wc;
    zt[0] = zw[0], zt[1] = zw[1], zt[2] = zw[kWinW], zt[3] = zw[kWinW + 1];
    u0 = win.u[wc], u1 = win.u[wc + 1], v0 = win.v[wr], v1 = win.v[wr + 1];
    o.cell = wr * kWinW + wc;
  } else {
    o.cell = -1;
    const int x1 = min(o.taps.x0 + 1, c.width - 1), y1 = min(o.taps.y0 + 1, c.height - 1);  // clamped reads; masked by taps.in
    const float* d0 = c.depth_e + (size_t)o.taps.y0 * c.width;
    const float* d1 = c.depth_e + (size_t)y1 * c.width;
    zt[0] = d0[o.taps.x0], zt[1] = d0[x1], zt[2] = d1[o.taps.x0], zt[3] = d1[x1];
    u0 = center_fast(o.taps.x0, c.fw, c.rcp_w), u1 = center_fast(o.taps.x0 + 1, c.fw, c.rcp_w);
    v0 = center_fast(o.taps.y0, c.fh, c.rcp_h), v1 = center_fast(o.taps.y0 + 1, c.fh, c.rcp_h);
  }
  dense_h(o.taps, zt, u0, u1, v0, v1, o.h);
  o.u0 = u0, o.u1 = u1, o.v0 = v0, o.v1 = v1;
  return o;
}

// Four horizontally adjacent later pixels per thread: 16-byte loads (32 bytes of flow), so a wave's load instruction moves 1 KB
// (4 rows x 256 B) instead of 256 B — the tiled kernels are bound by the memory pipeline, not by arithmetic (measured: neither
// halving the occupancy nor adding the tap sums' 70 instructions per pixel moves their duration; quartering the number of
// memory instructions does).  lane & 15 -> the group of four columns, lane >> 4 and the wave -> the row: 16 rows per pass.
constexpr int kQuadRows = 256 / (kTileW / 4);          // rows of the tile one pass of the block covers
constexpr int kQuadPasses = kTileH / kQuadRows;
static_assert(kTileW % 4 == 0 && kTileH % kQuadRows == 0, "quad mapping");
struct DenseRaw4 {
  float fx[4], fy[4], wt[4], z[4];
};
// idx = row·width + col0, col0 % 4 == 0.  `vec`: width % 4 == 0 (then all four pixels exist and the addresses are 16-byte aligned
// on 16-byte aligned stacks); else element-wise with a range test.
__device__ __forceinline__ DenseRaw4 dense_load4(const DenseCtx& c, int idx, int col0, bool vec) {
  DenseRaw4 r;
  if (vec) {
    const v4f a = *reinterpret_cast<const v4f*>(c.bwd_flow + (size_t)idx * 2), b = *reinterpret_cast<const v4f*>(c.bwd_flow + (size_t)idx * 2 + 4);
    const v4f w = *reinterpret_cast<const v4f*>(c.weights + idx), z = *reinterpret_cast<const v4f*>(c.depth_l + idx);
    r.fx[0] = a.x, r.fy[0] = a.y, r.fx[1] = a.z, r.fy[1] = a.w, r.fx[2] = b.x, r.fy[2] = b.y, r.fx[3] = b.z, r.fy[3] = b.w;
    r.wt[0] = w.x, r.wt[1] = w.y, r.wt[2] = w.z, r.wt[3] = w.w;
    r.z[0] = z.x, r.z[1] = z.y, r.z[2] = z.z, r.z[3] = z.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool in = col0 + j < c.width;
      const float2 fl = in ? reinterpret_cast<const float2*>(c.bwd_flow)[idx + j] : make_float2(0.f, 0.f);
      r.fx[j] = fl.x, r.fy[j] = fl.y;
      r.wt[j] = in ? c.weights[idx + j] : 0.f;
      r.z[j] = in ? c.depth_l[idx + j] : 0.f;
    }
  }
  return r;
}
__device__ __forceinline__ DenseRaw dense_raw_of(const DenseRaw4& r, int j) {
  DenseRaw o;
  o.fl = make_float2(r.fx[j], r.fy[j]);
  o.wt = r.wt[j];
  o.z = r.z[j];
  return o;
}

// grid: 1-D, >= tiles·pairs blocks (dense_block).  Raw pixel-space moments of the tile's pixels.
#ifndef FM_DENSE_MOMENTS_BLOCKS
#define FM_DENSE_MOMENTS_BLOCKS 1
#endif
// (The two matrix-pipe formulations of these sums — v_mfma_f32_4x4x1 per pixel, and per 16-pixel group through LDS — were measured in round 4
// and rejected, 0.903 / 0.664 ms against 0.608: docs/history/patches/r04_dense_procrustes_mfma_and_quad.patch, profiles/r04_dense_microbench_mfma.txt.)
__global__ void __launch_bounds__(256, FM_DENSE_MOMENTS_BLOCKS) procrustes_moments_dense_kernel(ProcParams p, unsigned total) {
  __shared__ double red[4 * kMomentCount];
  __shared__ DenseWindow win;
  const DenseBlock blk = dense_block(p.height, p.width, total);
  if (!blk.valid) return;
  const DenseCtx c = dense_ctx(p, blk, true);
  float gs[3];
  dense_shift(c.depth_l, p.height, p.width, gs);
  stage_depth_window(c, win);
  float acc[kMomentCount];
#pragma unroll
  for (int k = 0; k < kMomentCount; ++k) acc[k] = 0.f;
  const bool vec = (p.width & 3) == 0;
  const int col0 = c.tx0 + 4 * (threadIdx.x & (kTileW / 4 - 1));
  int row = c.ty0 + threadIdx.x / (kTileW / 4);
  const bool live = col0 < p.width;
  float u[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) u[j] = center_fast(col0 + j, c.fw, c.rcp_w);
  DenseRaw4 next = {};
  if (live && row < p.height) next = dense_load4(c, row * p.width + col0, col0, vec);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kQuadPasses; ++k, row += kQuadRows) {
    if (!live || row >= p.height) break;
    const DenseRaw4 cur = next;
    if (k + 1 < kQuadPasses && row + kQuadRows < p.height) next = dense_load4(c, (row + kQuadRows) * p.width + col0, col0, vec);
    const float v = center_fast(row, c.fh, c.rcp_h);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (col0 + j >= p.width) break;
      const DensePixel px = dense_pixel(c, win, dense_raw_of(cur, j), u[j], v);
      dense_moments_add(px.g, px.h, px.w, gs, acc);
    }
  }
  // (Two pixels per pass in packed fp32 — v_pk_fma_f32 on pairs of the quad — issued 14 % fewer instructions and ran 7 % SLOWER: a
  // packed instruction occupies the SIMD about twice as long as a plain one on this part, tools/probes/pk_rate_probe.hip.)
  block_accumulate<kMomentCount>(acc, red, p.stats + c.pair * kStatStride);
}

// One thread per pair: pixel-space raw moments -> the statistics of align_rigid -> the pose.
__global__ void procrustes_finish_solve_dense_kernel(ProcParams p, int pairs, float* t_bwd, float* t_fwd, double* aux) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= pairs) return;
  const int b = pair / (p.frames - 1), i = pair % (p.frames - 1);
  const CorrSrc src = pair_source<SRC_DEPTH>(p, pair, b, i);
  float gs[3];
  dense_shift(src.depth_l, p.height, p.width, gs);
  double* st = p.stats + (size_t)pair * kStatStride;
  dense_moments_finish(st, gs, p.kinv + ((size_t)b * p.frames + i) * 9, p.kinv + ((size_t)b * p.frames + i + 1) * 9);
  if (t_bwd) pose_solve_one(st, t_bwd + (size_t)pair * 16, t_fwd ? t_fwd + (size_t)pair * 16 : nullptr, aux + (size_t)pair * kAuxStride);
}

// Per-pair constants of the dense backward, once per pair (fp64): consts (pairs, kDenseConstStride)
// = DenseBwd (21 floats' worth), K_e (9), K_l (9).
constexpr int kDenseConstStride = 40;
constexpr int kFusedScaleSlot = 39;  // of a pair's row of consts: e with T < 2^e, the fused backward's fixed-point scale (>= 1e9: none usable)
// (dL/dK⁻¹ of the pair's two frames is linear in the statistics of the forward pass: fm_pose_solve_bwd_kinv has written it.)
__global__ void __launch_bounds__(64) procrustes_dense_consts_kernel(ProcParams p, const double* aux, int pairs, double* consts) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= pairs) return;
  const int b = pair / (p.frames - 1), i = pair % (p.frames - 1);
  DenseBwd c;
  double* o = consts + (size_t)pair * kDenseConstStride;
  dense_bwd_consts(p.pair_grad + (size_t)pair * kPairGradStride, aux + (size_t)pair * kAuxStride, p.kinv + ((size_t)b * p.frames + i) * 9,
                   p.kinv + ((size_t)b * p.frames + i + 1) * 9, c, o + 21, o + 30);
  const float* f = c.bm;
  for (int k = 0; k < 21; ++k) o[k] = (double)f[k];  // bm, a0, b0, gbar, hbar are contiguous
  // the fused backward's fixed-point scale (see there): the typical tap magnitude of the pair, T < 2^e
  double t = 0.0;
  for (int a = 0; a < 3; ++a)
    t += fabs((double)c.bm[a * 3 + 0] * c.gbar[0]) + fabs((double)c.bm[a * 3 + 1] * c.gbar[1]) + fabs((double)c.bm[a * 3 + 2] * c.gbar[2]) + fabs((double)c.b0[a]);
  const double wsum = aux[(size_t)pair * kAuxStride + 27];
  const double wmean = fabs(wsum) / ((double)p.height * p.width);
  t *= wmean > 1e-30 ? wmean : 1.0;
  int e = 0;
  const bool ok = t > 1e-300 && t < 1e300;
  if (ok) (void)frexp(t, &e);
  o[kFusedScaleSlot] = (ok && e > -90 && e < 90) ? (double)e : 2.0e9;  // (outside: every tap goes to memory directly)
}

__device__ __forceinline__ DenseBwd dense_load_consts(const double* consts, size_t pair) {
  DenseBwd c;
  float* f = c.bm;
  const double* o = consts + pair * kDenseConstStride;
#pragma unroll
  for (int k = 0; k < 21; ++k) f[k] = (float)o[k];  // wave-uniform: scalar loads
  return c;
}

// Dense backward, later role.  dL/dweights STORED (every element of every pair exactly once),
// dL/ddepth of the later pixel added in place (this launch is the only writer of that pixel).
__global__ void __launch_bounds__(256, FM_DENSE_LATER_BLOCKS) procrustes_dense_bwd_later_kernel(ProcParams p, const double* consts, unsigned total) {
  __shared__ DenseWindow win;
  const DenseBlock blk = dense_block(p.height, p.width, total);
  if (!blk.valid) return;
  const DenseCtx c = dense_ctx(p, blk, true);
  const size_t n = (size_t)p.height * p.width;
  const DenseBwd cst = dense_load_consts(consts, c.pair);
  stage_depth_window(c, win);
  const int col = c.tx0 + (threadIdx.x & (kTileW - 1));
  int row = c.ty0 + threadIdx.x / kTileW;
  const bool live = col < p.width;
  const float u = center_fast(col, c.fw, c.rcp_w);
  float* gw_out = p.grad_weights ? p.grad_weights + c.pair * n : nullptr;
  float* gd_out = p.grad_depth ? p.grad_depth + (c.fe + 1) * n : nullptr;
  DenseRaw next = {};
  float gd_next = 0.f;
  if (live && row < p.height) {
    next = dense_load(c, row * p.width + col);
    if (gd_out) gd_next = gd_out[row * p.width + col];
  }
  __syncthreads();
#pragma unroll FM_DENSE_UNROLL_LATER
  for (int k = 0; k < kRowsPerThread; ++k, row += 256 / kTileW) {
    if (!live || row >= p.height) break;
    const int idx = row * p.width + col;
    const DenseRaw cur = next;
    const float gd_cur = gd_next;
    if (k + 1 < kRowsPerThread && row + 256 / kTileW < p.height) {
      next = dense_load(c, idx + (256 / kTileW) * p.width);
      if (gd_out) gd_next = gd_out[idx + (256 / kTileW) * p.width];
    }
    const float v = center_fast(row, c.fh, c.rcp_h);
    const DensePixel px = dense_pixel(c, win, cur, u, v);
    float tv[3], gc[3], sv[3], gw;
    dense_bwd_t(cst, px.g, tv, gc);
    dense_bwd_s(cst, px.h, tv, gc, sv, gw);
    if (c.sens != 0.f) gw *= c.sens * px.w * (1.f - px.w);  // d sigmoid(s·x)/dx
    if (gw_out) gw_out[idx] = gw;
    if (gd_out) gd_out[idx] = gd_cur + px.w * fmaf(sv[0], u, fmaf(sv[1], v, sv[2]));
  }
}

// Dense backward, BOTH roles in one pass over the later frame's pixels (the default; the planned pair of kernels above and
// below is kept for callers that want dL/ddepth bit-reproducible).  One read of (flow, weight, z) and one evaluation of the
// correspondence serve dL/dweights, the later pixel's own dL/ddepth and the four tap gradients of the earlier frame, which
// the two-kernel form pays for twice (its tap kernel re-derives β and the taps of 1.05 list entries per pixel).  The tap
// gradients are summed in an LDS image of the earlier-frame WINDOW the block has staged anyway — 64-bit integer atomics in
// fixed point — and the image is flushed to dL/ddepth with hardware float atomics (global_atomic_add_f32; neighbouring
// blocks' windows overlap).  The later pixel's own gradient is an atomic add too: in this launch frame f receives the later
// role of pair f-1 and the tap role of pair f concurrently.
//
// The fixed-point scale is per PAIR and needs no pass over the pixels: the consts kernel derives a typical tap magnitude
// T = w̄·Σ_a (Σ_d |Bm_ad|·|ḡ_d| + |b0_a|) from the pair's constants (ḡ: the weighted centroid, w̄: the mean weight) and
// values are scaled so that T ↦ 2^32 (kFusedMagic below).  A correspondence whose |β| exceeds 2^7·T — an outlier depth, a
// weight far above the mean — does not fit the 2048-contributions-per-cell headroom of the sums and sends its taps to memory
// directly as float atomics, exactly like a tap outside the window: any input is summed correctly, only the typical ones
// take the LDS path.
#ifndef FM_DENSE_FUSED_WIN_ROWS
#define FM_DENSE_FUSED_WIN_ROWS 8
#define FM_DENSE_FUSED_WIN_COLS 16
#endif
#ifndef FM_DENSE_FUSED_BLOCKS
#define FM_DENSE_FUSED_BLOCKS 4
#endif
#ifndef FM_DENSE_FUSED_UNROLL
#define FM_DENSE_FUSED_UNROLL 4
#endif
// (Four adjacent later pixels per thread with 16-byte accesses — what helps the moments kernel — made this pass slower, 2.26 against 1.17 ms:
// docs/history/patches/r04_dense_procrustes_mfma_and_quad.patch, profiles/r04_dense_microbench_quad.txt.)
#ifndef FM_DENSE_FUSED_SKIP  // timing experiments only (tools/dense_microbench.py): 1 = plain store for the later pixel (racy), 2 = no flush, 4 = no taps,
                            // 8 = the window flushed with plain read-modify-writes (racy in ONE launch: what a four-colour tile order would be allowed to do)
#define FM_DENSE_FUSED_SKIP 0
#endif
#ifndef FM_DENSE_FUSED_PAD   // timing experiments only: extra LDS per block (occupancy)
#define FM_DENSE_FUSED_PAD 0
#endif
// Fixed point without a conversion sequence: for |x| < 2^51 the double x + 1.5·2^52 has the bit pattern M + round(x), M = 0x4338'0000'0000'0000,
// and M's low 51 bits are zero — so 64-bit INTEGER sums of those patterns carry Σ round(x) in their low 51 bits (two's complement,
// exact modulo 2^51) whatever the number of terms.  T ↦ 2^32 and a block adds at most 2048 values to a cell, each below 2^39 = 2^7·T:
// |Σ| < 2^50.  Resolution 2^-32·T per contribution; the sums do not depend on the order of arrival.
constexpr double kFusedMagic = 6755399441055744.0;  // 1.5·2^52
constexpr int kFusedUnitBits = 32;
constexpr float kFusedLimit = 549755813888.f;       // 2^39
constexpr int kFusedWinH = kTileH + FM_DENSE_FUSED_WIN_ROWS, kFusedWinW = kTileW + FM_DENSE_FUSED_WIN_COLS;  // 40 x 80: 12.8 KB of depth + 25.6 KB of sums

__global__ void __launch_bounds__(256, FM_DENSE_FUSED_BLOCKS) procrustes_dense_bwd_fused_kernel(ProcParams p, const double* consts, unsigned total) {
  typedef DenseWindowT<kFusedWinH, kFusedWinW> Window;
  __shared__ Window win;
  __shared__ unsigned long long iacc[kFusedWinH * kFusedWinW];
#if FM_DENSE_FUSED_PAD
  __shared__ int pad[FM_DENSE_FUSED_PAD / 4];
  if (total == 0xffffffffu) pad[threadIdx.x] = 1;
#endif
  const DenseBlock blk = dense_block(p.height, p.width, total);
  if (!blk.valid) return;
  const DenseCtx c = dense_ctx<kFusedWinH, kFusedWinW>(p, blk, true);
  const size_t n = (size_t)p.height * p.width;
  const DenseBwd cst = dense_load_consts(consts, c.pair);
  const double scale_e = consts[c.pair * kDenseConstStride + kFusedScaleSlot];
  const bool usable = scale_e < 1.0e9;
  const int ex = usable ? (int)scale_e : 0;
  const float scale = usable ? ldexpf(1.0f, kFusedUnitBits - ex) : __builtin_nanf("");  // T·scale = 2^32 (NaN: every comparison below fails, every tap goes to memory)
  const double unscale = ldexp(1.0, ex - kFusedUnitBits);
  stage_depth_window(c, win);
  for (int i = threadIdx.x; i < kFusedWinH * kFusedWinW; i += 256) iacc[i] = 0ull;
  float* gw_out = p.grad_weights ? p.grad_weights + c.pair * n : nullptr;
  float* gd_l = p.grad_depth ? p.grad_depth + (c.fe + 1) * n : nullptr;
  float* gd_e = p.grad_depth ? p.grad_depth + c.fe * n : nullptr;
  // One later pixel's share of the backward: dL/dweights (returned), its own dL/ddepth (one float atomic: frame f receives the later
  // role of pair f-1 and the tap role of pair f in the same launch) and its four tap gradients (LDS image, or memory when they do not fit).
  auto one_pixel = [&](const DenseRaw& cur, float u, float v, int idx) -> float {
    const DensePixel px = dense_pixel(c, win, cur, u, v);
    float tv[3], gc[3], sv[3], gw;
    dense_bwd_t(cst, px.g, tv, gc);
    dense_bwd_s(cst, px.h, tv, gc, sv, gw);
    if (c.sens != 0.f) gw *= c.sens * px.w * (1.f - px.w);  // d sigmoid(s·x)/dx
    if (!gd_l) return gw;
    if (FM_DENSE_FUSED_SKIP & 1) gd_l[idx] += px.w * fmaf(sv[0], u, fmaf(sv[1], v, sv[2]));
    else atomicAdd(gd_l + idx, px.w * fmaf(sv[0], u, fmaf(sv[1], v, sv[2])));
    if (FM_DENSE_FUSED_SKIP & 4) return gw;
    // β = w·t = K⁻ᵀ_e·dL/dq; a tap's value is w_k·(β·[u_k, v_k, 1]) with w_k <= 1 and u_k, v_k in (0, 1): below |β0| + |β1| + |β2|
    const float b0 = px.w * tv[0], b1 = px.w * tv[1], b2 = px.w * tv[2];
    const float bsum = (fabsf(b0) + fabsf(b1)) + fabsf(b2);
    if (px.cell >= 0 && bsum * scale < kFusedLimit) {  // (false for a non-finite β and when the pair has no usable scale: scale = NaN)
      const float sb0 = b0 * scale, sb1 = b1 * scale, sb2 = b2 * scale;
      unsigned long long* cell = iacc + px.cell;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (!px.taps.in[t]) continue;
        const float sval = px.taps.w[t] * fmaf(sb0, (t & 1) ? px.u1 : px.u0, fmaf(sb1, (t >> 1) ? px.v1 : px.v0, sb2));
        // round(sval) as a 64-bit integer in two instructions: the low 51 bits of the double (sval + 1.5·2^52)
        atomicAdd(cell + (t >> 1) * kFusedWinW + (t & 1), (unsigned long long)__double_as_longlong((double)sval + kFusedMagic));
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (!px.taps.in[t]) continue;
        const float f0 = fabsf(b0) <= 3.0e38f ? b0 : 0.f, f1 = fabsf(b1) <= 3.0e38f ? b1 : 0.f, f2 = fabsf(b2) <= 3.0e38f ? b2 : 0.f;  // as the planned kernel: a non-finite gradient contributes nothing here
        const float val = px.taps.w[t] * fmaf(f0, (t & 1) ? px.u1 : px.u0, fmaf(f1, (t >> 1) ? px.v1 : px.v0, f2));
        if (val != 0.f) atomicAdd(gd_e + (size_t)(px.taps.y0 + (t >> 1)) * p.width + (px.taps.x0 + (t & 1)), val);
      }
    }
    return gw;
  };
  const int col = c.tx0 + (threadIdx.x & (kTileW - 1));
  int row = c.ty0 + threadIdx.x / kTileW;
  const bool live = col < p.width;
  const float u = center_fast(col, c.fw, c.rcp_w);
  DenseRaw next = {};
  if (live && row < p.height) next = dense_load(c, row * p.width + col);
  __syncthreads();
#pragma unroll FM_DENSE_FUSED_UNROLL
  for (int k = 0; k < kRowsPerThread; ++k, row += 256 / kTileW) {
    if (!live || row >= p.height) break;
    const int idx = row * p.width + col;
    const DenseRaw cur = next;
    if (k + 1 < kRowsPerThread && row + 256 / kTileW < p.height) next = dense_load(c, idx + (256 / kTileW) * p.width);
    const float v = center_fast(row, c.fh, c.rcp_h);
    const float gw = one_pixel(cur, u, v, idx);
    if (gw_out) gw_out[idx] = gw;
  }
  if (!gd_e || (FM_DENSE_FUSED_SKIP & 2)) return;
  __syncthreads();
  // the window's sums -> dL/ddepth of the earlier frame (only cells inside the image ever receive a tap)
  for (int i = threadIdx.x; i < kFusedWinH * kFusedWinW; i += 256) {
    const long long v = ((long long)(iacc[i] << 13)) >> 13;  // the low 51 bits, sign-extended
    if (v == 0) continue;
    const int r = i / kFusedWinW, q = i - r * kFusedWinW;
    float* cell = gd_e + (size_t)(c.wy0 + r) * p.width + (c.wx0 + q);
    if (FM_DENSE_FUSED_SKIP & 8) *cell += (float)((double)v * unscale);
    else atomicAdd(cell, (float)((double)v * unscale));
  }
}

// Later pixels are listed per earlier-frame tile as (row << 16 | col).
__device__ __forceinline__ uint32_t pack_pixel(int row, int col) { return ((uint32_t)row << 16) | (uint32_t)col; }

// Plan, pass 1 (list == null): counts[pair·tiles + tile] += 1 per (later pixel, tile its taps land in);
// pass 2: list[first[...] + cursor++] = packed later pixel.  grid: (pixel chunks, pairs).
__global__ void __launch_bounds__(256) procrustes_dense_plan_kernel(const float* bwd_flow, int height, int width, int* counts,
                                                                     const int64_t* first, uint32_t* list) {
  const size_t pair = blockIdx.y;
  const int n = height * width;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int row = j / width, col = j - row * width;
  const float fw = (float)width, fh = (float)height;
  const float2 fl = reinterpret_cast<const float2*>(bwd_flow + pair * (size_t)n * 2)[j];
  const Taps t = dense_taps(center_fast(col, fw, 1.0f / fw) + fl.x, center_fast(row, fh, 1.0f / fh) + fl.y, height, width);
  const int tiles_x = (width + kTileW - 1) / kTileW, tiles_y = (height + kTileH - 1) / kTileH;
  const int txa = t.x0 / kTileW, tya = t.y0 / kTileH;
  const int txb = t.in[1] ? (t.x0 + 1) / kTileW : txa, tyb = t.in[2] ? (t.y0 + 1) / kTileH : tya;
  int* cnt = counts + pair * ((size_t)tiles_x * tiles_y);
  const int64_t* fst = first ? first + pair * ((size_t)tiles_x * tiles_y) : nullptr;
  auto note = [&](int tile) {
    const int pos = atomicAdd(cnt + tile, 1);
    if (list) list[fst[tile] + pos] = pack_pixel(row, col);
  };
  note(tya * tiles_x + txa);
  if (txb != txa) note(tya * tiles_x + txb);
  if (tyb != tya) {
    note(tyb * tiles_x + txa);
    if (txb != txa) note(tyb * tiles_x + txb);
  }
}

// Plan, pass 3: every tile's list in ascending (row, col) order, so that a wave of the tap kernel reads
// (nearly) consecutive pixels.  The order the atomic cursors of pass 2 produce is arbitrary — adjacent
// lanes would gather from unrelated cache lines (measured: 2.9 ms per step instead of 0.9).  One block per
// list, bitonic sort in LDS, in chunks of kSortChunk entries (a longer list ends up sorted chunk-wise:
// the order only matters for coalescing, not for the result).
constexpr int kSortChunk = 4096;
__global__ void __launch_bounds__(256) procrustes_dense_plan_sort_kernel(const int64_t* first, uint32_t* list) {
  __shared__ uint32_t buf[kSortChunk];
  const int64_t lo = first[blockIdx.x], hi = first[blockIdx.x + 1];
  for (int64_t base = lo; base < hi; base += kSortChunk) {
    const int count = (int)min((int64_t)kSortChunk, hi - base);
    for (int i = threadIdx.x; i < kSortChunk; i += 256) buf[i] = i < count ? list[base + i] : 0xffffffffu;
    __syncthreads();
    for (int size = 2; size <= kSortChunk; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int i = threadIdx.x; i < kSortChunk / 2; i += 256) {
          const int lo_i = 2 * i - (i & (stride - 1));  // index with bit `stride` clear
          const int hi_i = lo_i + stride;
          const bool up = (lo_i & size) == 0;
          const uint32_t a = buf[lo_i], b = buf[hi_i];
          if ((a > b) == up) {
            buf[lo_i] = b;
            buf[hi_i] = a;
          }
        }
        __syncthreads();
      }
    for (int i = threadIdx.x; i < count; i += 256) list[base + i] = buf[i];
    __syncthreads();
  }
}

// Dense backward, earlier role: block = (pair, tile of the EARLIER frame).
//
// The tap gradients of the listed correspondences are summed per tile cell in LDS with 64-bit INTEGER
// atomics on a per-batch fixed-point scale: ds_add_f32 runs at about half a lane per clock on this
// part (measured, tools/dense_microbench.py: 227 M float LDS atomics 0.71 ms, the same adds as
// ds_add_u64 0.02 ms), and integer sums do not depend on the order of arrival, so dL/ddepth is
// bit-reproducible.  Per batch of kTapBatch·256 list entries: (1) load, form β = w·t and the bound
// |β0|+|β1|+|β2| >= |any tap value|, (2) block maximum -> scale 2^e with |value|·2^e <= 2^30, (3) taps,
// values converted to 49-bit fixed point and added as 64-bit words, (4) every thread folds the cells it
// owns into its fp32 running sums and clears them.
#ifndef FM_DENSE_TAP_BATCH
#define FM_DENSE_TAP_BATCH 6  // (8: 142 VGPRs, 3 waves per SIMD, 2.18 ms for later + taps; 6: 126 VGPRs, 4 waves, 1.98 ms; 4: 1.98; 2: 2.05)
#endif
constexpr int kTapBatch = FM_DENSE_TAP_BATCH;

__global__ void __launch_bounds__(256, FM_DENSE_TAPS_BLOCKS) procrustes_dense_bwd_taps_kernel(ProcParams p, const double* consts, const int64_t* first,
                                                                         const uint32_t* list, unsigned total) {
  __shared__ unsigned long long iacc[kTileH * kTileW];
  __shared__ float tile_u[kTileW], tile_v[kTileH];
  __shared__ float wave_max[4];
  const DenseBlock blk = dense_block(p.height, p.width, total);
  if (!blk.valid) return;
  const DenseCtx c = dense_ctx(p, blk, false);
  const size_t n = (size_t)p.height * p.width;
  const DenseBwd cst = dense_load_consts(consts, c.pair);
  const int tiles_x = (p.width + kTileW - 1) / kTileW, tiles_y = (p.height + kTileH - 1) / kTileH;
  for (int i = threadIdx.x; i < kTileH * kTileW; i += 256) iacc[i] = 0ull;
  if (threadIdx.x < kTileW) tile_u[threadIdx.x] = center_fast(c.tx0 + (int)threadIdx.x, c.fw, c.rcp_w);
  else if (threadIdx.x < kTileW + kTileH) tile_v[threadIdx.x - kTileW] = center_fast(c.ty0 + (int)threadIdx.x - kTileW, c.fh, c.rcp_h);
  const size_t slot = c.pair * ((size_t)tiles_x * tiles_y) + (size_t)blk.tile_y * tiles_x + blk.tile_x;
  const int64_t lo = first[slot], hi = first[slot + 1];
  // this thread's cells of the tile: the gradient already there is fetched early, added to at the end
  float* gd = p.grad_depth + c.fe * n;
  const int lcol = threadIdx.x & (kTileW - 1), lrow0 = threadIdx.x / kTileW;
  const int col = c.tx0 + lcol;
  float sum[kRowsPerThread];
#pragma unroll
  for (int k = 0; k < kRowsPerThread; ++k) {
    const int row = c.ty0 + lrow0 + k * (256 / kTileW);
    sum[k] = (col < p.width && row < p.height) ? gd[(size_t)row * p.width + col] : 0.f;
  }
  __syncthreads();
  for (int64_t base = lo; base < hi; base += kTapBatch * 256) {
    uint32_t pk[kTapBatch];
    DenseRaw raw[kTapBatch];
#pragma unroll
    for (int j = 0; j < kTapBatch; ++j) {
      const int64_t e = base + j * 256 + threadIdx.x;
      pk[j] = list[e < hi ? e : lo];  // clamped: the entry is ignored below
    }
#pragma unroll
    for (int j = 0; j < kTapBatch; ++j) raw[j] = dense_load(c, (int)(pk[j] >> 16) * p.width + (int)(pk[j] & 0xffffu));
    // (1) β per entry and the bound of the batch
    float beta[kTapBatch][3], uu[kTapBatch], vv[kTapBatch];
    float bound = 0.f;
#pragma unroll
    for (int j = 0; j < kTapBatch; ++j) {
      const bool used = base + j * 256 + threadIdx.x < hi;
      uu[j] = center_fast((int)(pk[j] & 0xffffu), c.fw, c.rcp_w);
      vv[j] = center_fast((int)(pk[j] >> 16), c.fh, c.rcp_h);
      const float w = c.sens != 0.f ? fm_sigmoid<true>(c.sens * raw[j].wt) : raw[j].wt;
      const float g[3] = {raw[j].z * uu[j], raw[j].z * vv[j], raw[j].z};
      float tv[3], gc[3];
      dense_bwd_t(cst, g, tv, gc);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float bv = w * tv[a];  // K⁻ᵀ_e·dL/dq
        beta[j][a] = (used && fabsf(bv) <= 3.0e38f) ? bv : 0.f;  // a non-finite gradient contributes nothing here (and would poison the scale)
      }
      bound = fmaxf(bound, fabsf(beta[j][0]) + fabsf(beta[j][1]) + fabsf(beta[j][2]));
    }
    // (2) block maximum -> power-of-two scale
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) bound = fmaxf(bound, __shfl_xor(bound, off, kWave));
    if ((threadIdx.x & (kWave - 1)) == 0) wave_max[threadIdx.x >> 6] = bound;
    __syncthreads();
    bound = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
    int ex = 0;
    (void)frexpf(bound, &ex);                       // bound < 2^ex
    const float scale = ldexpf(1.0f, 25 - ex);      // |value|·scale < 2^25; 24 more fractional bits below
    const double unscale = ldexp(1.0, ex - 49);
    // (3) taps
    if (bound > 0.f) {
#pragma unroll
      for (int j = 0; j < kTapBatch; ++j) {
        const Taps tp = dense_taps(uu[j] + raw[j].fl.x, vv[j] + raw[j].fl.y, p.height, p.width);  // exactly as the plan and the other dense kernels
        const int rr = tp.y0 - c.ty0, cc = tp.x0 - c.tx0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r1 = rr + (k >> 1), c1 = cc + (k & 1);
          if (!tp.in[k] || (unsigned)r1 >= (unsigned)kTileH || (unsigned)c1 >= (unsigned)kTileW) continue;
          const float val = tp.w[k] * fmaf(beta[j][0], tile_u[c1], fmaf(beta[j][1], tile_v[r1], beta[j][2]));
          // 49-bit fixed point in two exact steps: integer part, then the remainder scaled by 2^24 — every
          // fp32 value within 2^-25 of the batch maximum is represented exactly, smaller ones to 2^-49 of it;
          // |q| < 2^49 and a batch has at most 8192 tap contributions, so a cell cannot overflow 64 bits
          const float sv = val * scale, hi = rintf(sv);
          const long long q = ((long long)(int)hi << 24) + (long long)__float2int_rn((sv - hi) * 16777216.f);
          if (q != 0) atomicAdd(iacc + r1 * kTileW + c1, (unsigned long long)q);
        }
      }
    }
    __syncthreads();
    // (4) fold this thread's cells
#pragma unroll
    for (int k = 0; k < kRowsPerThread; ++k) {
      const int cell = (lrow0 + k * (256 / kTileW)) * kTileW + lcol;
      const long long v = (long long)iacc[cell];
      if (v != 0) {
        sum[k] += (float)((double)v * unscale);
        iacc[cell] = 0ull;
      }
    }
    // (the next batch's atomics come after its own barrier in step 2)
  }
#pragma unroll
  for (int k = 0; k < kRowsPerThread; ++k) {
    const int row = c.ty0 + lrow0 + k * (256 / kTileW);
    if (col < p.width && row < p.height) gd[(size_t)row * p.width + col] = sum[k];
  }
}

// Static pattern of the sparse depth-sourced scatter (indices and flows are constants of the
// optimisation): correspondence j of pair `pair` touches its four taps in the earlier frame and its
// own pixel in the later frame.  keys[(pair·P + j)·5 + slot] = frame·H·W + pixel (slot 0..3 taps with
// their bilinear weights, slot 4 the later pixel with weight 1), -1 for a tap outside the image.
__global__ void __launch_bounds__(256) procrustes_scatter_plan_kernel(const float* bwd_flow, const int64_t* indices, long points, int frames,
                                                                      int height, int width, int64_t* keys, float* weights, long flow_fs, long flow_bs) {
  const size_t pair = blockIdx.y;
  const int b = (int)(pair / (frames - 1)), i = (int)(pair % (frames - 1));
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= points) return;
  const int64_t n = (int64_t)height * width;
  const int idx = indices ? (int)indices[j] : (int)j;
  const PixelRef px = pixel_ref(idx, height, width);
  const float* fl = bwd_flow + (size_t)b * flow_bs + (size_t)i * flow_fs + (size_t)idx * 2;
  const Taps t = bilinear_taps(px.u + fl[0], px.v + fl[1], height, width);  // as corr_load_with
  const int64_t fe = (int64_t)b * frames + i;
  const size_t o = (pair * (size_t)points + (size_t)j) * 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    keys[o + k] = t.in[k] ? fe * n + (int64_t)tap_row(t, k) * width + tap_col(t, k) : (int64_t)-1;
    weights[o + k] = t.in[k] ? t.w[k] : 0.f;
  }
  keys[o + 4] = (fe + 1) * n + idx;
  weights[o + 4] = 1.f;
}

// out[group·stride + indices[j]] = values[group·points + j]: the per-correspondence weight gradients
// dropped into the (zeroed) dense dL/dweights.
__global__ void __launch_bounds__(256) sparse_store_kernel(const float* values, const int64_t* indices, long points, long stride, float* out) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= points) return;
  out[(size_t)blockIdx.y * stride + (size_t)indices[j]] = values[(size_t)blockIdx.y * points + j];
}

// ---------------------------------------------------------------------------------
// Pose solve: one thread per pair.
//   t_bwd[pair] = [R | t]  maps later-camera -> earlier-camera ("inverse relative
//                 transformation", projection.py:190-197)
//   t_fwd[pair] = its rigid inverse [Rᵀ | −Rᵀt]
// ---------------------------------------------------------------------------------
__global__ void pose_solve_kernel(const double* stats, int pairs, float* t_bwd, float* t_fwd, double* aux) {
  const int pr = blockIdx.x * blockDim.x + threadIdx.x;
  if (pr >= pairs) return;
  pose_solve_one(stats + (size_t)pr * kStatStride, t_bwd + (size_t)pr * 16, t_fwd ? t_fwd + (size_t)pr * 16 : nullptr,
                 aux + (size_t)pr * kAuxStride);
}

// Backward of the solve.  Inputs: dL/dT_bwd and (optionally) dL/dT_fwd, 4x4 row-major,
// bottom rows ignored.  Output per pair: gM, total centroid gradients, and the scalars
// the per-point pass needs.
__global__ void pose_solve_bwd_kernel(const float* g_t_bwd, const float* g_t_fwd, const float* t_bwd, const double* aux,
                                      int pairs, double* pair_grad, double* clear, long clear_count) {
  const int pr = blockIdx.x * blockDim.x + threadIdx.x;
  // the accumulator the per-point pass (next launch on this stream) adds into
  for (long i = pr; i < clear_count; i += (long)gridDim.x * blockDim.x) clear[i] = 0.0;
  if (pr >= pairs) return;
  pose_solve_bwd_one(g_t_bwd ? g_t_bwd + (size_t)pr * 16 : nullptr, g_t_fwd ? g_t_fwd + (size_t)pr * 16 : nullptr,
                     t_bwd + (size_t)pr * 16, aux + (size_t)pr * kAuxStride, pair_grad + (size_t)pr * kPairGradStride);
}

// pose_solve_bwd_one for every pair AND dL/dK⁻¹ of every frame, which is linear in the statistics the forward pass kept in
// `aux` (pair_kinv_grads): one thread per (batch entry, frame) writes its pair's row of pair_grad and
//   kinv_acc[b, f] = earlier role of pair (b, f)  +  later role of pair (b, f−1)
// (the neighbour's pair gradient is recomputed, 3 us of fp64, so that no value has two writers: no atomics, no zero fill,
// and the per-point pass carries no sums for K at all).
__global__ void __launch_bounds__(64) pose_solve_bwd_kinv_kernel(const float* g_t_bwd, const float* g_t_fwd, const float* t_bwd, const double* aux,
                                                                const float* kinv, int batch, int frames, double* pair_grad, double* kinv_acc) {
  const int bf = blockIdx.x * blockDim.x + threadIdx.x;
  if (bf >= batch * frames) return;
  const int b = bf / frames, f = bf % frames;
  double kd[9], kf[9], acc[9];
  for (int k = 0; k < 9; ++k) {
    kd[k] = kinv[(size_t)bf * 9 + k];
    acc[k] = 0.0;
  }
  inv3d(kd, kf);  // K of this frame
  if (f < frames - 1) {  // this frame is the EARLIER frame of pair (b, f)
    const size_t pr = (size_t)b * (frames - 1) + f;
    double* pg = pair_grad + pr * kPairGradStride;
    pose_solve_bwd_one(g_t_bwd ? g_t_bwd + pr * 16 : nullptr, g_t_fwd ? g_t_fwd + pr * 16 : nullptr, t_bwd + pr * 16, aux + pr * kAuxStride, pg);
    double ge[9];
    pair_kinv_grads(pg, aux + pr * kAuxStride, kf, nullptr, ge, nullptr);
    for (int k = 0; k < 9; ++k) acc[k] += ge[k];
  }
  if (f > 0) {  // ... and the LATER frame of pair (b, f−1)
    const size_t pr = (size_t)b * (frames - 1) + f - 1;
    double pg[kPairGradStride], gl[9];
    pose_solve_bwd_one(g_t_bwd ? g_t_bwd + pr * 16 : nullptr, g_t_fwd ? g_t_fwd + pr * 16 : nullptr, t_bwd + pr * 16, aux + pr * kAuxStride, pg);
    pair_kinv_grads(pg, aux + pr * kAuxStride, nullptr, kf, nullptr, gl);
    for (int k = 0; k < 9; ++k) acc[k] += gl[k];
  }
  for (int k = 0; k < 9; ++k) kinv_acc[(size_t)bf * 9 + k] = acc[k];
}

// Per-point backward + scatter.  grid: (chunks, B*(F-1)).
template <int SRC>
__global__ void __launch_bounds__(256) procrustes_scatter_kernel(ProcParams p, const double* aux, int iters) {
  __shared__ double red[4 * 18];
  const size_t pair = blockIdx.y;
  const int b = (int)(pair / (p.frames - 1));
  const int i = (int)(pair % (p.frames - 1));
  const int n = p.height * p.width;
  Mat3 kinv_e, kinv_l;
  if (SRC == SRC_DEPTH) {
    load_mat3(p.kinv + ((size_t)b * p.frames + i) * 9, kinv_e);
    load_mat3(p.kinv + ((size_t)b * p.frames + i + 1) * 9, kinv_l);
  }
  const CorrSrc src = pair_source<SRC>(p, pair, b, i);
  const double* pg = p.pair_grad + pair * kPairGradStride;
  const double* ax = aux + pair * kAuxStride;
  PairGrad g;
  for (int k = 0; k < 9; ++k) g.gM[k] = (float)pg[k];
  for (int a = 0; a < 3; ++a) {
    g.gqbar[a] = (float)pg[9 + a];
    g.gpbar[a] = (float)pg[12 + a];
    g.pbar[a] = (float)ax[21 + a];
    g.qbar[a] = (float)ax[24 + a];
  }
  g.dbar = (float)pg[15];
  g.inv_wsum = (float)pg[16];
  const int bd = b / p.batch_repeat;
  const size_t fe = (size_t)bd * p.frames + i, fl = fe + 1;        // image-data frames
  const size_t fk = (size_t)b * p.frames + i;                      // (kinv, pose) frame
  const size_t dpair = (size_t)bd * (p.frames - 1) + i;            // image-data pair

  float acc[18];  // [0..8] dKinv later frame, [9..17] dKinv earlier frame
#pragma unroll
  for (int k = 0; k < 18; ++k) acc[k] = 0.f;

  const long base = (long)blockIdx.x * blockDim.x * iters;
  for (int it = 0; it < iters; ++it) {
    const long j = base + (long)it * blockDim.x + threadIdx.x;
    if (j >= p.points) break;
    const Corr c = corr_load(src, kinv_e, kinv_l, p.indices ? (int)p.indices[j] : (int)j);
    float gq[3], gp[3], gw;
    corr_backward(c, g, gq, gp, gw);
    if (p.weight_sens != 0.f) gw *= p.weight_sens * c.w * (1.f - c.w);  // d sigmoid(s·x)/dx
    const bool planned = SRC == SRC_DEPTH && p.point_grads != nullptr;  // distinct indices: plain stores, depth by fm_depth_gather
    if (planned && p.point_weight_grads) {
      p.point_weight_grads[pair * (size_t)p.points + (size_t)j] = gw;  // placed by fm_sparse_store once the buffer is zeroed
    } else if (p.grad_weights) {
      if (planned) p.grad_weights[dpair * (size_t)n + c.idx] = gw;
      else atomicAdd(p.grad_weights + dpair * (size_t)n + c.idx, gw);
    }
    if (planned) {
      float* o = p.point_grads + (pair * (size_t)p.points + (size_t)j) * 6;
      o[0] = gq[0], o[1] = gq[1], o[2] = gq[2], o[3] = gp[0], o[4] = gp[1], o[5] = gp[2];
    }
    // (with kinv_acc == NULL — dL/dK⁻¹ comes from fm_pose_solve_bwd_kinv — and a planned scatter there is nothing left to do per tap)
    if (SRC == SRC_DEPTH && (p.kinv_acc != nullptr || (p.grad_depth && !planned))) {
      const int row = c.idx / p.width, col = c.idx - row * p.width;
      const float u = pixel_center(col, p.width), v = pixel_center(row, p.height);
      if (p.grad_depth && !planned) atomicAdd(p.grad_depth + fl * n + c.idx, gp[0] * c.ray_p[0] + gp[1] * c.ray_p[1] + gp[2] * c.ray_p[2]);
      const float zh[3] = {c.z_p * u, c.z_p * v, c.z_p};
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[a * 3 + d] += gp[a] * zh[d];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!c.taps.in[k]) continue;
        const int tc = tap_col(c.taps, k), tr = tap_row(c.taps, k);
        const float ut = pixel_center(tc, p.width), vt = pixel_center(tr, p.height);
        const float z = p.depth[fe * n + tr * p.width + tc];
        float ray[3];
        ray_dir(kinv_e, ut, vt, ray);
        const float wt = c.taps.w[k];
        if (p.grad_depth && !planned) atomicAdd(p.grad_depth + fe * n + tr * p.width + tc, wt * (gq[0] * ray[0] + gq[1] * ray[1] + gq[2] * ray[2]));
        const float zt[3] = {z * ut * wt, z * vt * wt, z * wt};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int d = 0; d < 3; ++d) acc[9 + a * 3 + d] += gq[a] * zt[d];
      }
    } else if (SRC != SRC_DEPTH) {
      if (p.grad_surfaces) {
        float* gl = p.grad_surfaces + (fl * n + c.idx) * 3;
        atomicAdd(gl + 0, gp[0]);
        atomicAdd(gl + 1, gp[1]);
        atomicAdd(gl + 2, gp[2]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (!c.taps.in[k]) continue;
          float* ge = p.grad_surfaces + (fe * n + tap_row(c.taps, k) * p.width + tap_col(c.taps, k)) * 3;
          atomicAdd(ge + 0, gq[0] * c.taps.w[k]);
          atomicAdd(ge + 1, gq[1] * c.taps.w[k]);
          atomicAdd(ge + 2, gq[2] * c.taps.w[k]);
        }
      }
    }
  }
  if (SRC == SRC_DEPTH && p.kinv_acc) {
    // later-frame and earlier-frame accumulators are adjacent rows of kinv_acc:
    // (fe)*9 = earlier, (fl)*9 = later  ->  reorder so one call covers 18 contiguous values
    float ordered[18];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      ordered[k] = acc[9 + k];
      ordered[9 + k] = acc[k];
    }
    block_accumulate<18>(ordered, red, p.kinv_acc + fk * 9);
  }
}

// ---------------------------------------------------------------------------------
// The whole backward of a PLANNED sparse fit (depth source, constant distinct indices, constant flows) in one launch:
// pose-solve backward, per-correspondence gradients, the planned gather into dL/ddepth and dL/dK — round 2 ran them as
// fm_pose_solve_bwd_kinv -> fm_procrustes_scatter -> fm_depth_gather_kgrad, three dependent launches of 10 + 32 + 24 us for
// ~1 us of arithmetic each.  One block per FRAME (b, f): everything that lands in dL/ddepth[b, f] comes from the two pairs the
// frame belongs to — the tap gradients dL/dq of pair f (f is its earlier frame) and the pixel gradients dL/dp of pair f−1
// (f is its later frame) — so the block evaluates both pairs' correspondences (each pair is evaluated by its two frames'
// blocks: 2 x 1 us of arithmetic for two launches less), keeps the 2·P gradient vectors in LDS and walks ITS slice of
// the gather plan (the plan is sorted by frame·H·W + pixel: `frame_first` marks where each frame's pixels begin).  No value
// has two writers: dL/dweights of pair f is stored by block f, dL/dK[f] by block f.
//   thread 0:  pose_solve_bwd_one(pair f)   -> LDS, earlier-role dL/dK⁻¹      (fp64, ~5 us of latency)
//   thread 64: pose_solve_bwd_one(pair f−1) -> LDS, later-role  dL/dK⁻¹       (another wave: the two run side by side)
// while the other waves already chase their correspondences' gather chains (index -> flow -> taps).
// ---------------------------------------------------------------------------------
struct FitBwdPlan {
  const float* g_t_bwd;        // (B·(F−1),4,4) or null
  const float* g_t_fwd;        // (B·(F−1),4,4) or null
  const float* t_bwd;          // (B·(F−1),4,4)
  const double* aux;           // (B·(F−1), kAuxStride)
  const int64_t* pixels;       // plan: touched pixels, ascending frame·H·W + pixel
  const int32_t* first;        //       entries of pixel m: [first[m], first[m+1])
  const int32_t* vectors;      //       2·(pair·P + j) + (later role)
  const float* tap_weights;    //       bilinear weight of the entry (1 for the later role)
  const int32_t* frame_first;  // (B·F + 1) first plan pixel of every frame
  float* g_k;                  // (B·F,3,3) or null
  int accumulate_k;            // g_k += instead of =
  int batch;
};

constexpr int kFitBwdThreads = 512;
constexpr int kFitBwdGather = 10;  // plan pixels a thread keeps in flight at a time (10 x 512 covers the ~4 900 pixels a frame's P = 1000 correspondences touch)

__device__ __forceinline__ PairGrad pair_grad_from(const double* pg, const double* ax) {
  PairGrad g;
  for (int k = 0; k < 9; ++k) g.gM[k] = (float)pg[k];
  for (int a = 0; a < 3; ++a) {
    g.gqbar[a] = (float)pg[9 + a];
    g.gpbar[a] = (float)pg[12 + a];
    g.pbar[a] = (float)ax[21 + a];
    g.qbar[a] = (float)ax[24 + a];
  }
  g.dbar = (float)pg[15];
  g.inv_wsum = (float)pg[16];
  return g;
}

// SLOTS: correspondences per thread and pair (P <= SLOTS · kFitBwdThreads): compile-time, so that every slot's record
// lives in registers.  512 threads = 8 waves = 2 per SIMD: the fp64 pose-solve backward of the two solver threads gets
// the 256-VGPR budget it wants (with 1024 threads — 128 VGPRs — it spilled 112 registers to scratch).
// `corr` (pairs·P, 8): the records the forward fit left (corr_record): the backward reads them as coalesced 16-byte vectors;
// the only scattered accesses left are the read-modify-writes of dL/ddepth and the stores of dL/dweights.  (Measured with
// phase clocks, tools/phase_clocks.py: re-gathering the correspondences — index -> flow -> taps, ~3 000 cold 64-byte lines per
// block and dependent round — cost 18 of the block's 52 us at C1, and a cold dependent round costs 5 us even on an idle GPU.)
template <int SLOTS>
__global__ void __launch_bounds__(kFitBwdThreads) procrustes_bwd_frame_kernel(ProcParams p, FitBwdPlan pl, const float* __restrict__ corr) {
  extern __shared__ float vec_lds[];  // [role 0: dL/dq of pair f | role 1: dL/dp of pair f−1][P][3]
  __shared__ double pg_lds[2][kPairGradStride];
  __shared__ double kacc_lds[2][9];
  const int bf = blockIdx.x;
  const int b = bf / p.frames, f = bf % p.frames;
  const int n = p.height * p.width;
  const bool has_e = f < p.frames - 1;  // this frame is the EARLIER frame of pair (b, f)
  const bool has_l = f > 0;             // ... and the LATER frame of pair (b, f−1)
  const size_t pair_e = (size_t)b * (p.frames - 1) + f, pair_l = pair_e - 1;
  const int t = threadIdx.x;
  const int P = (int)p.points;
  FM_PHASE(0);

  Mat3 kinv_f;  // K⁻¹ of this frame
  load_mat3(p.kinv + (size_t)bf * 9, kinv_f);
  const int lo = pl.frame_first ? pl.frame_first[bf] : 0, hi = pl.frame_first ? pl.frame_first[bf + 1] : 0;

  // ---- the records of both pairs (issued before anything waits; consumed after the pose-solve backward) ----
  corr_v4 r0a[SLOTS], r0b[SLOTS], r1a[SLOTS], r1b[SLOTS];
  const corr_v4* rec_e = reinterpret_cast<const corr_v4*>(corr) + (has_e ? pair_e : pair_l) * (size_t)P * 2;
  const corr_v4* rec_l = reinterpret_cast<const corr_v4*>(corr) + (has_l ? pair_l : pair_e) * (size_t)P * 2;
#pragma unroll
  for (int sl = 0; sl < SLOTS; ++sl) {
    const int j = sl * kFitBwdThreads + t;
    const int jj = j < P ? j : 0;
    r0a[sl] = rec_e[jj * 2], r0b[sl] = rec_e[jj * 2 + 1];
    r1a[sl] = rec_l[jj * 2], r1b[sl] = rec_l[jj * 2 + 1];
  }
  // ... and the first batch of this frame's plan slice — keys and entry ranges, then (as soon as those have landed) the old
  // dL/ddepth values and the entries themselves: nothing of this depends on the gradients, so both dependent rounds of cold
  // accesses run under the pose-solve backward instead of after it
  int px[kFitBwdGather], e0[kFitBwdGather], e1[kFitBwdGather];
  const int64_t frame_base = (int64_t)bf * n;
  float* gd = p.grad_depth ? p.grad_depth + frame_base : nullptr;
#pragma unroll
  for (int i = 0; i < kFitBwdGather; ++i) {
    const int m = lo + t + i * kFitBwdThreads;
    const bool on = gd != nullptr && m < hi;
    px[i] = on ? (int)(pl.pixels[m] - frame_base) : -1;
    e0[i] = on ? pl.first[m] : 0;
    e1[i] = on ? pl.first[m + 1] : 0;
  }
  float old[kFitBwdGather], wt[kFitBwdGather];
  int vi[kFitBwdGather];
#pragma unroll
  for (int i = 0; i < kFitBwdGather; ++i) {
    const bool on = px[i] >= 0;
    old[i] = on ? gd[px[i]] : 0.f;
    vi[i] = on && e0[i] < e1[i] ? pl.vectors[e0[i]] : 0;
    wt[i] = on && e0[i] < e1[i] ? pl.tap_weights[e0[i]] : 0.f;
  }

  // ---- pose-solve backward of the two pairs, side by side in two waves (fp64, one thread each) ----
  FM_PHASE(1);
  if (t == 0 || t == kWave) {
    const int role = t == 0 ? 0 : 1;
    const bool on = role == 0 ? has_e : has_l;
    const size_t pr = role == 0 ? pair_e : pair_l;
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (on) {
      double* pg = pg_lds[role];
      pose_solve_bwd_one(pl.g_t_bwd ? pl.g_t_bwd + pr * 16 : nullptr, pl.g_t_fwd ? pl.g_t_fwd + pr * 16 : nullptr, pl.t_bwd + pr * 16,
                         pl.aux + pr * kAuxStride, pg);
      FM_PHASE(2);
      if (pl.g_k) {
        double kd[9], kf[9];
        for (int k = 0; k < 9; ++k) kd[k] = kinv_f.m[k];
        inv3d(kd, kf);  // K of this frame
        if (role == 0) pair_kinv_grads(pg, pl.aux + pr * kAuxStride, kf, nullptr, acc, nullptr);
        else pair_kinv_grads(pg, pl.aux + pr * kAuxStride, nullptr, kf, nullptr, acc);
      }
    }
    for (int k = 0; k < 9; ++k) kacc_lds[role][k] = acc[k];
  }
  FM_PHASE(3);
  __syncthreads();
  FM_PHASE(4);
  if (t == 2 * kWave && pl.g_k) {  // dK = [g_k] − K⁻ᵀ·dK⁻¹·K⁻ᵀ (kinv_grad_to_k), both roles summed in fp64; a third wave, off the others' path
    double tot[9], gk[9];
    for (int k = 0; k < 9; ++k) tot[k] = kacc_lds[0][k] + kacc_lds[1][k];
    kinv_grad_to_k(tot, p.kinv + (size_t)bf * 9, gk);
    for (int k = 0; k < 9; ++k) {
      float* o = pl.g_k + (size_t)bf * 9 + k;
      *o = (pl.accumulate_k ? *o : 0.f) + (float)gk[k];
    }
  }

  // ---- per-correspondence gradients of the two pairs -> LDS (and dL/dweights of pair f) ----
  PairGrad g0 = {}, g1 = {};
  if (has_e) g0 = pair_grad_from(pg_lds[0], pl.aux + pair_e * kAuxStride);
  if (has_l) g1 = pair_grad_from(pg_lds[1], pl.aux + pair_l * kAuxStride);
#pragma unroll
  for (int sl = 0; sl < SLOTS; ++sl) {
    const int j = sl * kFitBwdThreads + t;
    if (j >= P) continue;
    float gq[3], gp[3], gw;
    Corr c;  // (corr_backward reads p, q, w only)
    if (has_e) {
      c.q[0] = r0a[sl].x, c.q[1] = r0a[sl].y, c.q[2] = r0a[sl].z, c.p[0] = r0a[sl].w, c.p[1] = r0b[sl].x, c.p[2] = r0b[sl].y, c.w = r0b[sl].z;
      corr_backward(c, g0, gq, gp, gw);
      if (p.weight_sens != 0.f) gw *= p.weight_sens * c.w * (1.f - c.w);  // d sigmoid(s·x)/dx
      if (p.grad_weights) p.grad_weights[pair_e * (size_t)n + __float_as_int(r0b[sl].w)] = gw;  // distinct indices: a plain store per slot
      float* out = vec_lds + (size_t)j * 3;
      out[0] = gq[0], out[1] = gq[1], out[2] = gq[2];
    }
    if (has_l) {
      c.q[0] = r1a[sl].x, c.q[1] = r1a[sl].y, c.q[2] = r1a[sl].z, c.p[0] = r1a[sl].w, c.p[1] = r1b[sl].x, c.p[2] = r1b[sl].y, c.w = r1b[sl].z;
      corr_backward(c, g1, gq, gp, gw);
      float* out = vec_lds + ((size_t)P + j) * 3;
      out[0] = gp[0], out[1] = gp[1], out[2] = gp[2];
    }
  }
  FM_PHASE(5);
  __syncthreads();
  FM_PHASE(6);
  if (p.grad_depth == nullptr) return;

  // ---- this frame's slice of the planned gather: one plain read-modify-write per touched pixel.  A thread keeps
  // kFitBwdGather pixels in flight; the first batch's loads were issued at the top ----
  const int base_e = (int)(pair_e * (size_t)P), base_l = (int)(pair_l * (size_t)P);
  for (int m0 = lo + t; m0 < hi; m0 += kFitBwdThreads * kFitBwdGather) {
    if (m0 != lo + t) {
#pragma unroll
      for (int i = 0; i < kFitBwdGather; ++i) {
        const int m = m0 + i * kFitBwdThreads;
        const bool on = m < hi;
        px[i] = on ? (int)(pl.pixels[m] - frame_base) : -1;
        e0[i] = on ? pl.first[m] : 0;
        e1[i] = on ? pl.first[m + 1] : 0;
      }
#pragma unroll
      for (int i = 0; i < kFitBwdGather; ++i) {
        const bool on = px[i] >= 0;
        old[i] = on ? gd[px[i]] : 0.f;
        vi[i] = on && e0[i] < e1[i] ? pl.vectors[e0[i]] : 0;
        wt[i] = on && e0[i] < e1[i] ? pl.tap_weights[e0[i]] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < kFitBwdGather; ++i) {
      if (px[i] < 0) continue;
      const int row = px[i] / p.width, col = px[i] - row * p.width;
      float ray[3];
      ray_dir(kinv_f, pixel_center(col, p.width), pixel_center(row, p.height), ray);
      float sum = 0.f;
      int v = vi[i];
      float w = wt[i];
      for (int e = e0[i]; e < e1[i]; ++e) {
        if (e > e0[i]) {  // (a pixel several correspondences touch: rare)
          v = pl.vectors[e];
          w = pl.tap_weights[e];
        }
        const int later = v & 1, j = (v >> 1) - (later ? base_l : base_e);
        const float* g3 = vec_lds + ((size_t)later * P + j) * 3;
        sum += w * (g3[0] * ray[0] + g3[1] * ray[1] + g3[2] * ray[2]);
      }
      gd[px[i]] = old[i] + sum;
    }
    if (m0 == lo + t) FM_PHASE(7);
  }
  FM_PHASE(8);
}

// ---------------------------------------------------------------------------------
// Pose chain (get_extrinsics): E_0 = I, E_k = E_{k-1} · T_{k-1}.  One thread per batch
// element walks the chain in fp64 (F ≤ a few thousand; latency ≈ tens of µs, replacing
// F-1 dependent matmul launches).  Backward is the reverse scan.
// ---------------------------------------------------------------------------------
// Forward: one block per batch element.  Each thread multiplies a chunk of consecutive
// relative poses, the chunk products are prefix-multiplied with a Hillis-Steele scan in
// LDS (8 dependent 4x4 products instead of F-1), then every thread re-walks its chunk
// from its exclusive prefix and writes the extrinsics.
constexpr int kChainThreads = 256;

__device__ __forceinline__ void load16(const float* p, double* o) {
#pragma unroll
  for (int k = 0; k < 16; ++k) o[k] = p[k];
}

__global__ void __launch_bounds__(kChainThreads) pose_chain_fwd_kernel(const float* rel, int batch, int steps, float* ext) {
  __shared__ double buf[2][16][kChainThreads];  // [entry][thread]: bank-conflict-free
  const int b = blockIdx.x, t = threadIdx.x;
  const float* r = rel + (size_t)b * steps * 16;
  float* e = ext + (size_t)b * (steps + 1) * 16;
  const int chunk = (steps + kChainThreads - 1) / kChainThreads;
  const int lo = t * chunk, hi = min(steps, lo + chunk);
  double prod[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int s = lo; s < hi; ++s) {
    double m[16], nxt[16];
    load16(r + (size_t)s * 16, m);
    mat4_mul(prod, m, nxt);
#pragma unroll
    for (int k = 0; k < 16; ++k) prod[k] = nxt[k];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) buf[0][k][t] = prod[k];
  __syncthreads();
  int cur = 0;
  for (int off = 1; off < kChainThreads; off <<= 1) {  // inclusive scan: buf[t] = C_0 · … · C_t
    double mine[16], out[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) mine[k] = buf[cur][k][t];
    if (t >= off) {
      double left[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) left[k] = buf[cur][k][t - off];
      mat4_mul(left, mine, out);
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) out[k] = mine[k];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) buf[cur ^ 1][k][t] = out[k];
    __syncthreads();
    cur ^= 1;
  }
  double run[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (t > 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) run[k] = buf[cur][k][t - 1];  // exclusive prefix
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) e[k] = (float)run[k];  // E_0 = I
  }
  for (int s = lo; s < hi; ++s) {
    double m[16], nxt[16];
    load16(r + (size_t)s * 16, m);
    mat4_mul(run, m, nxt);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      run[k] = nxt[k];
      e[(size_t)(s + 1) * 16 + k] = (float)nxt[k];
    }
  }
}

// Backward.  The reverse recurrence  carry_s = G_s + carry_{s+1}·T_sᵀ  unrolls, with
// T_s·…·T_{j-1} = E_s⁻¹·E_j, to  carry_s = (Σ_{j≥s} G_j·E_jᵀ)·E_s⁻ᵀ : an ADDITIVE suffix
// sum of A_j = G_j·E_jᵀ (parallel scan in LDS) followed by independent 4x4 products;
// dL/dT_s = E_sᵀ·carry_{s+1}.
__global__ void __launch_bounds__(kChainThreads) pose_chain_bwd_kernel(const float* rel, const float* ext, const float* g_ext,
                                                                     int batch, int steps, float* g_rel) {
  __shared__ double buf[2][16][kChainThreads];  // [entry][thread]: bank-conflict-free
  (void)rel;
  const int b = blockIdx.x, t = threadIdx.x;
  const float* e = ext + (size_t)b * (steps + 1) * 16;
  const float* ge = g_ext + (size_t)b * (steps + 1) * 16;
  float* o = g_rel + (size_t)b * steps * 16;
  const int frames = steps + 1;
  const int chunk = (frames + kChainThreads - 1) / kChainThreads;
  const int lo = t * chunk, hi = min(frames, lo + chunk);
  // chunk sums of A_j
  double sum[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) sum[k] = 0.0;
  for (int j = lo; j < hi; ++j) {
    double g[16], ej[16], a[16];
    load16(ge + (size_t)j * 16, g);
    load16(e + (size_t)j * 16, ej);
    mat4_mul_nt(g, ej, a);
#pragma unroll
    for (int k = 0; k < 16; ++k) sum[k] += a[k];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) buf[0][k][t] = sum[k];
  __syncthreads();
  int cur = 0;
  for (int off = 1; off < kChainThreads; off <<= 1) {  // inclusive SUFFIX sum over chunks
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = buf[cur][k][t] + (t + off < kChainThreads ? buf[cur][k][t + off] : 0.0);
#pragma unroll
    for (int k = 0; k < 16; ++k) buf[cur ^ 1][k][t] = v[k];
    __syncthreads();
    cur ^= 1;
  }
  // suffix[j] for j in this chunk, walking from the chunk's end: start with the sum of all later chunks
  double suf[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) suf[k] = (t + 1 < kChainThreads) ? buf[cur][k][t + 1] : 0.0;
  for (int j = hi - 1; j >= lo; --j) {
    double g[16], ej[16], a[16];
    load16(ge + (size_t)j * 16, g);
    load16(e + (size_t)j * 16, ej);
    mat4_mul_nt(g, ej, a);
#pragma unroll
    for (int k = 0; k < 16; ++k) suf[k] += a[k];  // = Σ_{i≥j} A_i
    if (j >= 1) {                                  // dL/dT_{j-1} = E_{j-1}ᵀ · (suf_j · E_j⁻ᵀ)
      double inv[16], carry[16], prev[16], gt[16];
      inv4(ej, inv);
      mat4_mul_nt(suf, inv, carry);
      load16(e + (size_t)(j - 1) * 16, prev);
      mat4_mul_tn(prev, carry, gt);
#pragma unroll
      for (int k = 0; k < 16; ++k) o[(size_t)(j - 1) * 16 + k] = (float)gt[k];
    }
  }
}

// ---------------------------------------------------------------------------------
// Relative poses from camera-to-world extrinsics with a GENERAL 4x4 inverse, exactly
// as the reference writes them (projection.py:154,176):
//   fwd[i] = inv(E_{i+1}) · E_i       bwd[i] = inv(E_i) · E_{i+1}
// and their backward  dE = −A⁻ᵀ·G·A⁻ᵀ  /  Aᵀ-products.  One thread per pair.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) relative_pose_fwd_kernel(const float* ext, int batch, int frames, float* fwd, float* bwd) {
  const int pr = blockIdx.x * blockDim.x + threadIdx.x;
  const int pairs = batch * (frames - 1);
  if (pr >= pairs) return;
  const int b = pr / (frames - 1), i = pr % (frames - 1);
  relative_pose_fwd_one(ext + ((size_t)b * frames + i) * 16, fwd + (size_t)pr * 16, bwd + (size_t)pr * 16);
}

// g_ext must be zero-initialised; adjacent pairs touch the same frame -> atomics.
__global__ void __launch_bounds__(64) relative_pose_bwd_kernel(const float* ext, const float* g_fwd, const float* g_bwd, int batch, int frames,
                                         float* g_ext) {
  const int pr = blockIdx.x * blockDim.x + threadIdx.x;
  const int pairs = batch * (frames - 1);
  if (pr >= pairs) return;
  const int b = pr / (frames - 1), i = pr % (frames - 1);
  double ge0[16], ge1[16];
  relative_pose_bwd_one(ext + ((size_t)b * frames + i) * 16, g_fwd ? g_fwd + (size_t)pr * 16 : nullptr,
                        g_bwd ? g_bwd + (size_t)pr * 16 : nullptr, ge0, ge1);
  float* o0 = g_ext + ((size_t)b * frames + i) * 16;
  for (int k = 0; k < 16; ++k) {
    atomicAdd(o0 + k, (float)ge0[k]);
    atomicAdd(o0 + 16 + k, (float)ge1[k]);
  }
}

// All-pairs relative poses inside a track segment: rel (B,f,f,4,4), one thread each.
__global__ void __launch_bounds__(64) allpairs_pose_fwd_kernel(const float* ext, int batch, int f, float* rel) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch * f * f) return;
  const int b = i / (f * f), r = i % (f * f);
  allpairs_pose_fwd_one(ext + (size_t)b * f * 16, r / f, r % f, rel + (size_t)i * 16);
}

__global__ void __launch_bounds__(64) allpairs_pose_bwd_kernel(const float* ext, const float* g_rel, int batch, int f, float* g_ext) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch * f) return;
  const int b = i / f;
  allpairs_pose_bwd_one(ext + (size_t)b * f * 16, g_rel + (size_t)b * f * f * 16, f, i % f, g_ext + (size_t)i * 16);
}

// K⁻¹ for every frame, and the map of a K⁻¹ gradient (fp64 accumulators) back to K:
//   dK += −K⁻ᵀ · dKinv · K⁻ᵀ
// IntrinsicsRegressed.forward (intrinsics_regressed.py:34-41): each focal length spread over
// `repeat` frames as K and K^-1 (the inverse every consumer of the step asks for).
__global__ void focal_intrinsics_fwd_kernel(const float* focal, long count, long repeat, int height, int width, float* k, float* kinv) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count * repeat) return;
  float m[9], inv[9];
  focal_to_k(focal[j / repeat], height, width, m);
  inv3(m, inv);
#pragma unroll
  for (int e = 0; e < 9; ++e) k[j * 9 + e] = m[e];
  if (kinv) {
#pragma unroll
    for (int e = 0; e < 9; ++e) kinv[j * 9 + e] = inv[e];
  }
}

// one block per focal length: dL/df = sqrt(hw) * sum over its frames of (gK00/w + gK11/h)
__global__ void __launch_bounds__(256) focal_intrinsics_bwd_kernel(const float* g_k, long repeat, int height, int width, float* g_focal) {
  __shared__ double red[4];
  const long i = blockIdx.x;
  double acc = 0.0;
  for (long r = threadIdx.x; r < repeat; r += blockDim.x) acc += focal_grad_term(g_k + (i * repeat + r) * 9, height, width);
  acc = wave_sum(acc);
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) g_focal[i] = (float)((red[0] + red[1] + red[2] + red[3]) * (double)(float)sqrt((double)height * (double)width));
}

__global__ void inv3_kernel(const float* k, int count, float* kinv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) inv3(k + (size_t)i * 9, kinv + (size_t)i * 9);
}

__global__ void kinv_grad_to_k_kernel(const double* kinv_acc, const float* kinv, int count, float* g_k, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double gk[9];
  kinv_grad_to_k(kinv_acc + (size_t)i * 9, kinv + (size_t)i * 9, gk);
  for (int e = 0; e < 9; ++e) {
    float* o = g_k + (size_t)i * 9 + e;
    *o = (accumulate ? *o : 0.f) + (float)gk[e];
  }
}

}  // namespace fm

using namespace fm;

// The tiled dense kernels apply to depth-sourced surfaces with every pixel a correspondence.
static inline long dense_blocks(int height, int width, int pairs) {
  return (long)((width + kTileW - 1) / kTileW) * ((height + kTileH - 1) / kTileH) * pairs;
}
static inline unsigned dense_grid(long total) { return (unsigned)(((total + kXcds - 1) / kXcds) * kXcds); }
static inline bool dense_tiled(const float* depth, const float* surfaces, const int64_t* indices, long points, int batch_repeat,
                               int height, int width, int pairs) {
  return depth && !surfaces && !indices && batch_repeat == 1 && points == (long)height * width && height <= 65535 && width <= 65535 &&
         dense_blocks(height, width, pairs) < (1L << 31) - kXcds;
}

// Fill p.fs / p.bs from the caller's fm_layout[4] = {depth, surfaces, bwd_flow, weights} (NULL or {0,0} entries = dense).
// `image_batch` = batch entries of the image data (B / batch_repeat).
static inline bool proc_layouts(ProcParams& p, const fm_layout* layouts, int frames, int height, int width) {
  const long n = (long)height * width;
  const long per_frame[4] = {n, 3 * n, 2 * n, n};
  const long frames_of[4] = {frames, frames, frames - 1, frames - 1};
  bool any = false;
  for (int i = 0; i < 4; ++i) {
    const bool given = layouts && (layouts[i].frame_stride != 0 || layouts[i].batch_stride != 0);
    any = any || given;
    p.fs[i] = given ? layouts[i].frame_stride : per_frame[i];
    p.bs[i] = given ? layouts[i].batch_stride : per_frame[i] * frames_of[i];
    if (p.fs[i] < per_frame[i]) return false;
  }
  (void)any;
  return true;
}
static inline bool proc_is_view(const fm_layout* layouts) {
  if (!layouts) return false;
  for (int i = 0; i < 4; ++i)
    if (layouts[i].frame_stride != 0 || layouts[i].batch_stride != 0) return true;
  return false;
}

static inline int choose_iters(long points) {
  // P ~ 1000: the work per pair is a latency chain of gathers, so spread it over as many
  // blocks as possible (4 blocks x 149 pairs instead of 149 blocks: 105 -> ~35 us for the
  // scatter at C1); P = H*W: 8 points per thread keeps the per-block reduction cheap.
  return points <= 65536 ? 1 : 8;
}

extern "C" {

static int procrustes_stats_launch(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow,
                                   const float* weights, float weight_sensitivity, const int64_t* indices, long points, int batch,
                                   int batch_repeat, int frames, int height, int width, double* stats, float* t_bwd, float* t_fwd,
                                   double* aux, hipStream_t st, const fm_layout* layouts = nullptr) {
  const int pairs = batch * (frames - 1);
  ProcParams p{};
  p.depth = depth; p.kinv = kinv; p.surfaces = surfaces; p.bwd_flow = bwd_flow; p.weights = weights; p.indices = indices;
  p.stats = stats; p.frames = frames; p.height = height; p.width = width; p.points = points;
  p.weight_sens = weight_sensitivity;
  p.batch_repeat = batch_repeat;
  FM_CHECK_ARG(proc_layouts(p, layouts, frames, height, width));
  const int iters = choose_iters(points);
  dim3 grid((unsigned)((points + 256L * iters - 1) / (256L * iters)), (unsigned)pairs);
  const dim3 fgrid((unsigned)((pairs + 63) / 64));
  const bool dense = dense_tiled(depth, surfaces, indices, points, batch_repeat, height, width, pairs);
  FM_CHECK_ARG(!(dense && proc_is_view(layouts)));  // the tiled dense kernels read dense stacks
  if (hipMemsetAsync(stats, 0, sizeof(double) * (size_t)pairs * kStatStride, st) != hipSuccess) return FM_ERR_LAUNCH;
  if (dense) {
    const long total = dense_blocks(height, width, pairs);
    hipLaunchKernelGGL(procrustes_moments_dense_kernel, dim3(dense_grid(total)), dim3(256), 0, st, p, (unsigned)total);
  } else if (surfaces) {
    hipLaunchKernelGGL((procrustes_moments_kernel<SRC_SURF>), grid, dim3(256), 0, st, p, iters, FitChain{});
  } else {
    hipLaunchKernelGGL((procrustes_moments_kernel<SRC_DEPTH>), grid, dim3(256), 0, st, p, iters, FitChain{});
  }
  if (dense) {  // pixel-space sums: intrinsics applied here, then the solve (t_bwd may be null: statistics only)
    hipLaunchKernelGGL(procrustes_finish_solve_dense_kernel, fgrid, dim3(64), 0, st, p, pairs, t_bwd, t_fwd, aux);
  } else if (t_bwd) {  // finish + solve per pair in one launch
    if (surfaces) hipLaunchKernelGGL((procrustes_finish_solve_kernel<SRC_SURF>), fgrid, dim3(64), 0, st, p, pairs, t_bwd, t_fwd, aux);
    else hipLaunchKernelGGL((procrustes_finish_solve_kernel<SRC_DEPTH>), fgrid, dim3(64), 0, st, p, pairs, t_bwd, t_fwd, aux);
  } else if (surfaces) {
    hipLaunchKernelGGL((procrustes_moments_finish_kernel<SRC_SURF>), fgrid, dim3(64), 0, st, p, pairs);
  } else {
    hipLaunchKernelGGL((procrustes_moments_finish_kernel<SRC_DEPTH>), fgrid, dim3(64), 0, st, p, pairs);
  }
  FM_LAUNCH_STATUS();
}

int fm_procrustes_stats(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow,
                        const float* weights, float weight_sensitivity, const int64_t* indices, long points, int batch,
                        int batch_repeat, int frames, int height, int width, double* stats, void* stream) {
  FM_CHECK_ARG((depth && kinv) || surfaces);
  FM_CHECK_ARG(bwd_flow && weights && stats && points >= 1 && batch >= 1 && frames >= 2);
  FM_CHECK_ARG(batch_repeat >= 1 && batch % batch_repeat == 0);
  FM_CHECK_ARG((long)height * width < (1L << 30) && (long)batch * (frames - 1) <= 65535);
  return procrustes_stats_launch(depth, kinv, surfaces, bwd_flow, weights, weight_sensitivity, indices, points, batch, batch_repeat,
                                 frames, height, width, stats, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

int fm_procrustes_fit(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow, const float* weights,
                      float weight_sensitivity, const int64_t* indices, long points, int batch, int batch_repeat, int frames,
                      int height, int width, double* stats, float* t_bwd, float* t_fwd, double* aux, void* stream) {
  FM_CHECK_ARG((depth && kinv) || surfaces);
  FM_CHECK_ARG(bwd_flow && weights && stats && t_bwd && aux && points >= 1 && batch >= 1 && frames >= 2);
  FM_CHECK_ARG(batch_repeat >= 1 && batch % batch_repeat == 0);
  FM_CHECK_ARG((long)height * width < (1L << 30) && (long)batch * (frames - 1) <= 65535);
  return procrustes_stats_launch(depth, kinv, surfaces, bwd_flow, weights, weight_sensitivity, indices, points, batch, batch_repeat,
                                 frames, height, width, stats, t_bwd, t_fwd, aux, (hipStream_t)stream);
}

int fm_procrustes_fit_views(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow, const float* weights,
                            float weight_sensitivity, const int64_t* indices, long points, int batch, int frames, int height, int width,
                            double* stats, float* t_bwd, float* t_fwd, double* aux, const fm_layout* layouts, void* stream) {
  FM_CHECK_ARG((depth && kinv) || surfaces);
  FM_CHECK_ARG(bwd_flow && weights && stats && t_bwd && aux && points >= 1 && batch >= 1 && frames >= 2);
  FM_CHECK_ARG((long)height * width < (1L << 30) && (long)batch * (frames - 1) <= 65535);
  return procrustes_stats_launch(depth, kinv, surfaces, bwd_flow, weights, weight_sensitivity, indices, points, batch, 1, frames, height, width,
                                 stats, t_bwd, t_fwd, aux, (hipStream_t)stream, layouts);
}

static int fit_chain_launch(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow, const float* weights,
                            float weight_sensitivity, const int64_t* indices, long points, int batch, int frames, int height, int width,
                            double* work, float* t_bwd, float* t_fwd, double* aux, float* ext, float* corr_out, const float* tap_records,
                            const fm_layout* layouts, void* stream) {
  FM_CHECK_ARG((depth && kinv) || surfaces);
  FM_CHECK_ARG(!tap_records || (depth && indices && points <= 4096 && (reinterpret_cast<uintptr_t>(tap_records) & 15) == 0));
  FM_CHECK_ARG(bwd_flow && weights && work && t_bwd && aux && points >= 1 && batch >= 1 && frames >= 2);
  FM_CHECK_ARG((long)height * width < (1L << 30) && (long)batch * (frames - 1) <= 65535);
  FM_CHECK_ARG(!corr_out || (tap_records && points <= 4096 && (reinterpret_cast<uintptr_t>(corr_out) & 15) == 0));
  const int pairs = batch * (frames - 1);
  ProcParams p{};
  p.depth = depth; p.kinv = kinv; p.surfaces = surfaces; p.bwd_flow = bwd_flow; p.weights = weights; p.indices = indices;
  p.stats = work; p.frames = frames; p.height = height; p.width = width; p.points = points;
  p.weight_sens = weight_sensitivity;
  p.batch_repeat = 1;
  FM_CHECK_ARG(proc_layouts(p, layouts, frames, height, width));
  FitChain fc{reinterpret_cast<int*>(work + (size_t)pairs * kStatStride), t_bwd, t_fwd, aux, ext, batch, corr_out, tap_records};
  if (points <= 4096) {  // one block per pair: the sums stay in the block (procrustes_fit_pair_kernel)
    int* counter = fc.counters + pairs;  // the last of the workspace's ints
    if (surfaces) hipLaunchKernelGGL((procrustes_fit_pair_kernel<SRC_SURF>), dim3(pairs), dim3(1024), 0, (hipStream_t)stream, p, fc, counter);
    else hipLaunchKernelGGL((procrustes_fit_pair_kernel<SRC_DEPTH>), dim3(pairs), dim3(1024), 0, (hipStream_t)stream, p, fc, counter);
    FM_LAUNCH_STATUS();
  }
  const int iters = choose_iters(points);
  dim3 grid((unsigned)((points + 256L * iters - 1) / (256L * iters)), (unsigned)pairs);
  if (surfaces) hipLaunchKernelGGL((procrustes_moments_kernel<SRC_SURF>), grid, dim3(256), 0, (hipStream_t)stream, p, iters, fc);
  else hipLaunchKernelGGL((procrustes_moments_kernel<SRC_DEPTH>), grid, dim3(256), 0, (hipStream_t)stream, p, iters, fc);
  FM_LAUNCH_STATUS();
}

int fm_procrustes_fit_chain(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow, const float* weights,
                            float weight_sensitivity, const int64_t* indices, long points, int batch, int frames, int height, int width,
                            double* work, float* t_bwd, float* t_fwd, double* aux, float* ext, float* corr_out, const float* tap_records, void* stream) {
  return fit_chain_launch(depth, kinv, surfaces, bwd_flow, weights, weight_sensitivity, indices, points, batch, frames, height, width, work, t_bwd,
                          t_fwd, aux, ext, corr_out, tap_records, nullptr, stream);
}

int fm_procrustes_fit_chain_views(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow, const float* weights,
                                  float weight_sensitivity, const int64_t* indices, long points, int batch, int frames, int height, int width,
                                  double* work, float* t_bwd, float* t_fwd, double* aux, float* ext, float* corr_out, const float* tap_records,
                                  const fm_layout* layouts, void* stream) {
  return fit_chain_launch(depth, kinv, surfaces, bwd_flow, weights, weight_sensitivity, indices, points, batch, frames, height, width, work, t_bwd,
                          t_fwd, aux, ext, corr_out, tap_records, layouts, stream);
}

int fm_pose_solve(const double* stats, int pairs, float* t_bwd, float* t_fwd, double* aux, void* stream) {
  FM_CHECK_ARG(stats && t_bwd && aux && pairs >= 1);
  hipLaunchKernelGGL(pose_solve_kernel, dim3((pairs + 63) / 64), dim3(64), 0, (hipStream_t)stream, stats, pairs, t_bwd, t_fwd, aux);
  FM_LAUNCH_STATUS();
}

int fm_pose_solve_bwd(const float* g_t_bwd, const float* g_t_fwd, const float* t_bwd, const double* aux, int pairs,
                      double* pair_grad, double* clear, long clear_count, void* stream) {
  FM_CHECK_ARG(t_bwd && aux && pair_grad && pairs >= 1 && clear_count >= 0 && (clear || clear_count == 0));
  hipLaunchKernelGGL(pose_solve_bwd_kernel, dim3((pairs + 63) / 64), dim3(64), 0, (hipStream_t)stream, g_t_bwd, g_t_fwd, t_bwd,
                     aux, pairs, pair_grad, clear, clear_count);
  FM_LAUNCH_STATUS();
}

int fm_pose_solve_bwd_kinv(const float* g_t_bwd, const float* g_t_fwd, const float* t_bwd, const double* aux, const float* kinv, int batch,
                           int frames, double* pair_grad, double* kinv_acc, void* stream) {
  FM_CHECK_ARG(t_bwd && aux && kinv && pair_grad && kinv_acc && batch >= 1 && frames >= 2);
  const int n = batch * frames;
  hipLaunchKernelGGL(pose_solve_bwd_kinv_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, g_t_bwd, g_t_fwd, t_bwd, aux, kinv, batch,
                     frames, pair_grad, kinv_acc);
  FM_LAUNCH_STATUS();
}

static int scatter_launch(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow,
                          const float* weights, float weight_sensitivity, const int64_t* indices, long points, int batch,
                          int batch_repeat, int frames, int height, int width, const double* aux, const double* pair_grad,
                          float* grad_depth, float* grad_surfaces, float* grad_weights, double* kinv_acc, float* point_grads,
                          float* point_weight_grads, const fm_layout* layouts, void* stream) {
  FM_CHECK_ARG((depth && kinv) || surfaces);
  FM_CHECK_ARG(bwd_flow && weights && aux && pair_grad && points >= 1);
  FM_CHECK_ARG(batch_repeat >= 1 && batch % batch_repeat == 0);
  FM_CHECK_ARG(!point_grads || (depth && !surfaces && indices && batch_repeat == 1));
  FM_CHECK_ARG(!point_weight_grads || point_grads);
  hipStream_t st = (hipStream_t)stream;
  const int pairs = batch * (frames - 1);
  ProcParams p{};
  p.depth = depth; p.kinv = kinv; p.surfaces = surfaces; p.bwd_flow = bwd_flow; p.weights = weights; p.indices = indices;
  p.pair_grad = pair_grad; p.grad_depth = grad_depth; p.grad_surfaces = grad_surfaces; p.grad_weights = grad_weights;
  p.kinv_acc = kinv_acc; p.frames = frames; p.height = height; p.width = width; p.points = points;
  p.point_grads = point_grads;
  p.point_weight_grads = point_weight_grads;
  p.weight_sens = weight_sensitivity;
  p.batch_repeat = batch_repeat;
  FM_CHECK_ARG(proc_layouts(p, layouts, frames, height, width));
  FM_CHECK_ARG(!(proc_is_view(layouts) && batch_repeat > 1));
  const int iters = choose_iters(points);
  dim3 grid((unsigned)((points + 256L * iters - 1) / (256L * iters)), (unsigned)pairs);
  if (batch_repeat > 1 && !surfaces) {
    const int image_pairs = pairs / batch_repeat, groups = (batch_repeat + kRepeatGroup - 1) / kRepeatGroup;
    const dim3 rgrid((unsigned)((points + 255) / 256), (unsigned)(image_pairs * groups));
    hipLaunchKernelGGL(procrustes_scatter_repeat_kernel, rgrid, dim3(256), 0, st, p, aux);
  } else if (surfaces) hipLaunchKernelGGL((procrustes_scatter_kernel<SRC_SURF>), grid, dim3(256), 0, st, p, aux, iters);
  else hipLaunchKernelGGL((procrustes_scatter_kernel<SRC_DEPTH>), grid, dim3(256), 0, st, p, aux, iters);
  FM_LAUNCH_STATUS();
}

int fm_procrustes_scatter(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow,
                          const float* weights, float weight_sensitivity, const int64_t* indices, long points, int batch,
                          int batch_repeat, int frames, int height, int width, const double* aux, const double* pair_grad,
                          float* grad_depth, float* grad_surfaces, float* grad_weights, double* kinv_acc, float* point_grads,
                          float* point_weight_grads, void* stream) {
  return scatter_launch(depth, kinv, surfaces, bwd_flow, weights, weight_sensitivity, indices, points, batch, batch_repeat, frames, height, width, aux,
                        pair_grad, grad_depth, grad_surfaces, grad_weights, kinv_acc, point_grads, point_weight_grads, nullptr, stream);
}

int fm_procrustes_scatter_views(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow,
                                const float* weights, float weight_sensitivity, const int64_t* indices, long points, int batch, int frames,
                                int height, int width, const double* aux, const double* pair_grad, float* grad_depth, float* grad_surfaces,
                                float* grad_weights, double* kinv_acc, float* point_grads, const fm_layout* layouts, void* stream) {
  return scatter_launch(depth, kinv, surfaces, bwd_flow, weights, weight_sensitivity, indices, points, batch, 1, frames, height, width, aux, pair_grad,
                        grad_depth, grad_surfaces, grad_weights, kinv_acc, point_grads, nullptr, layouts, stream);
}

int fm_procrustes_bwd_planned(const float* corr, const float* kinv, float weight_sensitivity, long points, int batch, int frames, int height, int width,
                              const double* aux, const float* t_bwd, const float* g_t_bwd, const float* g_t_fwd, const int64_t* plan_pixels,
                              const int32_t* plan_first, const int32_t* plan_vectors, const float* plan_weights, const int32_t* frame_first,
                              float* grad_depth, float* grad_weights, float* g_k, int accumulate_k, void* stream) {
  FM_CHECK_ARG(corr && kinv && aux && t_bwd && (reinterpret_cast<uintptr_t>(corr) & 15) == 0);
  FM_CHECK_ARG(points >= 1 && points <= FM_FIT_BWD_MAX_POINTS && batch >= 1 && frames >= 2 && height >= 1 && width >= 1);
  FM_CHECK_ARG(!grad_depth || (plan_pixels && plan_first && plan_vectors && plan_weights && frame_first));
  FM_CHECK_ARG((long)batch * (frames - 1) * points < (1L << 30));
  ProcParams p{};
  p.kinv = kinv;
  p.grad_depth = grad_depth; p.grad_weights = grad_weights;
  p.frames = frames; p.height = height; p.width = width; p.points = points;
  p.weight_sens = weight_sensitivity;
  p.batch_repeat = 1;
  proc_layouts(p, nullptr, frames, height, width);
  FitBwdPlan pl{g_t_bwd, g_t_fwd, t_bwd, aux, plan_pixels, plan_first, plan_vectors, plan_weights, frame_first, g_k, accumulate_k, batch};
  const size_t lds = sizeof(float) * 2 * 3 * (size_t)points;
  const dim3 grid((unsigned)(batch * frames)), block(kFitBwdThreads);
  const int slots = (int)((points + kFitBwdThreads - 1) / kFitBwdThreads);
  static_assert(FM_FIT_BWD_MAX_POINTS <= 4 * kFitBwdThreads, "slots per thread");
  if (slots <= 1) hipLaunchKernelGGL(procrustes_bwd_frame_kernel<1>, grid, block, lds, (hipStream_t)stream, p, pl, corr);
  else if (slots == 2) hipLaunchKernelGGL(procrustes_bwd_frame_kernel<2>, grid, block, lds, (hipStream_t)stream, p, pl, corr);
  else if (slots == 3) hipLaunchKernelGGL(procrustes_bwd_frame_kernel<3>, grid, block, lds, (hipStream_t)stream, p, pl, corr);
  else hipLaunchKernelGGL(procrustes_bwd_frame_kernel<4>, grid, block, lds, (hipStream_t)stream, p, pl, corr);
  FM_LAUNCH_STATUS();
}

#ifdef FM_PHASE_CLOCKS
int fm_debug_phase_clocks(long long* host_out, int blocks) {  // (blocks, kPhaseSlots) wall_clock64 stamps (100 MHz) of the last launch
  if (blocks > kPhaseBlocks) blocks = kPhaseBlocks;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(fm_phase_clock_buffer), sizeof(long long) * blocks * kPhaseSlots) == hipSuccess ? kPhaseSlots : -1;
}
#endif

int fm_procrustes_dense_tiles(int height, int width, int* tiles) {
  FM_CHECK_ARG(tiles && height >= 1 && width >= 1);
  *tiles = ((width + kTileW - 1) / kTileW) * ((height + kTileH - 1) / kTileH);
  return FM_OK;
}

int fm_procrustes_dense_plan(const float* bwd_flow, int batch, int frames, int height, int width, int* counts, const int64_t* first,
                             uint32_t* list, void* stream) {
  FM_CHECK_ARG(bwd_flow && counts && batch >= 1 && frames >= 2 && height >= 1 && width >= 1 && height <= 65535 && width <= 65535);
  FM_CHECK_ARG((list == nullptr) == (first == nullptr) && (long)batch * (frames - 1) <= 65535);
  const long n = (long)height * width;
  hipLaunchKernelGGL(procrustes_dense_plan_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)(batch * (frames - 1))), dim3(256), 0,
                     (hipStream_t)stream, bwd_flow, height, width, counts, first, list);
  if (list) {  // pass 2 is followed by the per-tile sort
    int tiles = 0;
    fm_procrustes_dense_tiles(height, width, &tiles);
    hipLaunchKernelGGL(procrustes_dense_plan_sort_kernel, dim3((unsigned)(batch * (frames - 1) * tiles)), dim3(256), 0, (hipStream_t)stream, first, list);
  }
  FM_LAUNCH_STATUS();
}

int fm_procrustes_scatter_dense(const float* depth, const float* kinv, const float* bwd_flow, const float* weights,
                                float weight_sensitivity, int batch, int frames, int height, int width, const double* aux,
                                const double* pair_grad, float* grad_depth, float* grad_weights, const int64_t* first, const uint32_t* list,
                                double* consts, void* stream) {
  FM_CHECK_ARG(depth && kinv && bwd_flow && weights && aux && pair_grad && consts && batch >= 1 && frames >= 2);
  FM_CHECK_ARG(height >= 1 && width >= 1 && height <= 65535 && width <= 65535 && (long)height * width < (1L << 30));
  FM_CHECK_ARG((first == nullptr) == (list == nullptr));  // both: the planned pair of kernels; neither: one fused pass with atomics
  const int pairs = batch * (frames - 1);
  FM_CHECK_ARG(dense_blocks(height, width, pairs) < (1L << 31) - kXcds);
  hipStream_t st = (hipStream_t)stream;
  ProcParams p{};
  p.depth = depth; p.kinv = kinv; p.bwd_flow = bwd_flow; p.weights = weights; p.pair_grad = pair_grad; p.grad_depth = grad_depth;
  p.grad_weights = grad_weights; p.frames = frames; p.height = height; p.width = width;
  p.points = (long)height * width; p.weight_sens = weight_sensitivity; p.batch_repeat = 1;
  proc_layouts(p, nullptr, frames, height, width);
  const unsigned total = (unsigned)dense_blocks(height, width, pairs);
  hipLaunchKernelGGL(procrustes_dense_consts_kernel, dim3((pairs + 63) / 64), dim3(64), 0, st, p, aux, pairs, consts);
  if (!first) {
    if (grad_depth || grad_weights)
      hipLaunchKernelGGL(procrustes_dense_bwd_fused_kernel, dim3(dense_grid(total)), dim3(256), 0, st, p, consts, total);
    FM_LAUNCH_STATUS();
  }
  if (grad_depth || grad_weights)
    hipLaunchKernelGGL(procrustes_dense_bwd_later_kernel, dim3(dense_grid(total)), dim3(256), 0, st, p, consts, total);
  if (grad_depth) hipLaunchKernelGGL(procrustes_dense_bwd_taps_kernel, dim3(dense_grid(total)), dim3(256), 0, st, p, consts, first, list, total);
  FM_LAUNCH_STATUS();
}

static int scatter_plan_launch(const float* bwd_flow, const int64_t* indices, long points, int batch, int frames, int height, int width,
                               int64_t* keys, float* weights, const fm_layout* flow_layout, void* stream) {
  FM_CHECK_ARG(bwd_flow && keys && weights && points >= 1 && batch >= 1 && frames >= 2 && height >= 1 && width >= 1);
  FM_CHECK_ARG((long)height * width < (1L << 30) && (long)batch * (frames - 1) <= 65535);
  const long n2 = 2L * height * width;
  const bool given = flow_layout && (flow_layout->frame_stride != 0 || flow_layout->batch_stride != 0);
  const long fs = given ? flow_layout->frame_stride : n2, bs = given ? flow_layout->batch_stride : n2 * (frames - 1);
  FM_CHECK_ARG(fs >= n2);
  hipLaunchKernelGGL(procrustes_scatter_plan_kernel, dim3((unsigned)((points + 255) / 256), (unsigned)(batch * (frames - 1))), dim3(256), 0,
                     (hipStream_t)stream, bwd_flow, indices, points, frames, height, width, keys, weights, fs, bs);
  FM_LAUNCH_STATUS();
}

int fm_procrustes_scatter_plan(const float* bwd_flow, const int64_t* indices, long points, int batch, int frames, int height, int width,
                               int64_t* keys, float* weights, void* stream) {
  return scatter_plan_launch(bwd_flow, indices, points, batch, frames, height, width, keys, weights, nullptr, stream);
}

int fm_procrustes_scatter_plan_views(const float* bwd_flow, const int64_t* indices, long points, int batch, int frames, int height, int width,
                                     int64_t* keys, float* weights, const fm_layout* flow_layout, void* stream) {
  return scatter_plan_launch(bwd_flow, indices, points, batch, frames, height, width, keys, weights, flow_layout, stream);
}

int fm_sparse_store(const float* values, const int64_t* indices, long points, int groups, long stride, float* out, void* stream) {
  FM_CHECK_ARG(values && indices && out && points >= 1 && groups >= 1 && groups <= 65535 && stride >= 1);
  hipLaunchKernelGGL(sparse_store_kernel, dim3((unsigned)((points + 255) / 256), (unsigned)groups), dim3(256), 0, (hipStream_t)stream, values,
                     indices, points, stride, out);
  FM_LAUNCH_STATUS();
}

int fm_pose_chain_fwd(const float* rel, int batch, int steps, float* ext, void* stream) {
  FM_CHECK_ARG(rel && ext && batch >= 1 && steps >= 0);
  hipLaunchKernelGGL(pose_chain_fwd_kernel, dim3(batch), dim3(kChainThreads), 0, (hipStream_t)stream, rel, batch, steps, ext);
  FM_LAUNCH_STATUS();
}

int fm_pose_chain_bwd(const float* rel, const float* ext, const float* g_ext, int batch, int steps, float* g_rel, void* stream) {
  FM_CHECK_ARG(rel && ext && g_ext && g_rel && batch >= 1 && steps >= 0);
  hipLaunchKernelGGL(pose_chain_bwd_kernel, dim3(batch), dim3(kChainThreads), 0, (hipStream_t)stream, rel, ext, g_ext, batch,
                     steps, g_rel);
  FM_LAUNCH_STATUS();
}

int fm_relative_pose_fwd(const float* ext, int batch, int frames, float* fwd, float* bwd, void* stream) {
  FM_CHECK_ARG(ext && fwd && bwd && batch >= 1 && frames >= 2);
  const int pairs = batch * (frames - 1);
  hipLaunchKernelGGL(relative_pose_fwd_kernel, dim3((pairs + 63) / 64), dim3(64), 0, (hipStream_t)stream, ext, batch, frames, fwd, bwd);
  FM_LAUNCH_STATUS();
}

int fm_relative_pose_bwd(const float* ext, const float* g_fwd, const float* g_bwd, int batch, int frames, float* g_ext,
                         void* stream) {
  FM_CHECK_ARG(ext && g_ext && batch >= 1 && frames >= 2);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(g_ext, 0, sizeof(float) * (size_t)batch * frames * 16, st) != hipSuccess) return FM_ERR_LAUNCH;
  const int pairs = batch * (frames - 1);
  hipLaunchKernelGGL(relative_pose_bwd_kernel, dim3((pairs + 63) / 64), dim3(64), 0, st, ext, g_fwd, g_bwd, batch, frames, g_ext);
  FM_LAUNCH_STATUS();
}

int fm_allpairs_pose_fwd(const float* ext, int batch, int frames, float* rel, void* stream) {
  FM_CHECK_ARG(ext && rel && batch >= 1 && frames >= 1);
  const int n = batch * frames * frames;
  hipLaunchKernelGGL(allpairs_pose_fwd_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, ext, batch, frames, rel);
  FM_LAUNCH_STATUS();
}

int fm_allpairs_pose_bwd(const float* ext, const float* g_rel, int batch, int frames, float* g_ext, void* stream) {
  FM_CHECK_ARG(ext && g_rel && g_ext && batch >= 1 && frames >= 1);
  const int n = batch * frames;
  hipLaunchKernelGGL(allpairs_pose_bwd_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, ext, g_rel, batch, frames, g_ext);
  FM_LAUNCH_STATUS();
}

int fm_focal_intrinsics_fwd(const float* focal, long count, long repeat, int height, int width, float* k, float* kinv, void* stream) {
  FM_CHECK_ARG(focal && k && count >= 1 && repeat >= 1 && height >= 1 && width >= 1);
  const long n = count * repeat;
  hipLaunchKernelGGL(focal_intrinsics_fwd_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, focal, count, repeat,
                     height, width, k, kinv);
  FM_LAUNCH_STATUS();
}

int fm_focal_intrinsics_bwd(const float* grad_k, long count, long repeat, int height, int width, float* grad_focal, void* stream) {
  FM_CHECK_ARG(grad_k && grad_focal && count >= 1 && count <= 0x7fffffffL && repeat >= 1 && height >= 1 && width >= 1);
  hipLaunchKernelGGL(focal_intrinsics_bwd_kernel, dim3((unsigned)count), dim3(256), 0, (hipStream_t)stream, grad_k, repeat, height, width,
                     grad_focal);
  FM_LAUNCH_STATUS();
}

int fm_intrinsics_inverse(const float* k, int count, float* kinv, void* stream) {
  FM_CHECK_ARG(k && kinv && count >= 1);
  hipLaunchKernelGGL(inv3_kernel, dim3((count + 63) / 64), dim3(64), 0, (hipStream_t)stream, k, count, kinv);
  FM_LAUNCH_STATUS();
}

int fm_intrinsics_inverse_bwd(const double* kinv_acc, const float* kinv, int count, float* g_k, int accumulate, void* stream) {
  FM_CHECK_ARG(kinv_acc && kinv && g_k && count >= 1);
  hipLaunchKernelGGL(kinv_grad_to_k_kernel, dim3((count + 63) / 64), dim3(64), 0, (hipStream_t)stream, kinv_acc, kinv, count, g_k,
                     accumulate);
  FM_LAUNCH_STATUS();
}

}  // extern "C"
