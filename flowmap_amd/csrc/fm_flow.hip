// Fused flow-consistency loss + analytic gradients (the roofline kernel).
//
// Replaces, in ONE pass over HBM, the reference chain
//   unproject (projection.py:76-90) -> compute_forward_flow / compute_backward_flow
//   (projection.py:143-184) -> reproject_points / project_camera_space (:116-134,:49-58)
//   -> Mapping.forward (loss/mapping/mapping.py:35-43, mapping_huber.py:19-34)
//   -> masked sums (loss/loss_flow.py:55-70)
// and everything autograd would replay for it.
//
// Work decomposition: the grid is organised per SOURCE frame.
// Frame f is the source of the forward term of pair f (into camera f+1) and of the
// backward term of pair f-1 (into camera f-1), so one thread per (frame, pixel) reads
// depth[f] once, reads flow/mask of both directions once, and writes dL/ddepth[f] once
// with no atomics on the big tensor.  Algorithmic traffic: N·(8F + 24(F−1)) bytes.
//
// Pose / intrinsics gradients are linear in per-pixel quantities and are reduced
// per (frame, direction) into 13 numbers (fm_math.h "One flow residual"):
//   [0]      Σ ρ·m                        (loss numerator, unscaled)
//   [1..3]   σ = Σ ω                      ω = (q·w_u, q·w_v, q·(w·kd p)),  q = 1/(Z'+eps)
//   [4..12]  Ω = Σ ω ⊗ (z·[u,v,1])
// from which flow_finalize_frame derives dL/dR, dL/dt, dL/dK⁻¹_src and dL/dK_dst.
// Transposing wave reduction (fm_device.h) -> LDS (fp64) -> one fp64 atomic per value per block.
#include "../../include/flowmap_hip.h"
#include "fm_device.h"
#include "fm_pose.h"

namespace fm {

struct FlowParams {
  const float* depth;     // (B,F,H,W)
  const float* k;         // (B,F,3,3)
  const float* kinv;      // (B,F,3,3)
  const float* t_fwd;     // (B,F-1,4,4) camera f -> camera f+1
  const float* t_bwd;     // (B,F-1,4,4) camera f+1 -> camera f
  const float* flow_fwd;  // (B,F-1,H,W,2)
  const float* flow_bwd;  // (B,F-1,H,W,2)
  const float* mask_fwd;  // (B,F-1,H,W)
  const float* mask_bwd;  // (B,F-1,H,W)
  const float* packed;    // (B·F, chunks, 6, 64, 4) re-laid-out flows + masks (fm_flow_pack_inputs) or null
  const float* scale;     // device scalar multiplied into every per-residual gradient
  float* grad_depth;      // (B,F,H,W) or null
  double* acc;            // (B*F, 2, kFlowAccStride)
  int frames, height, width;
  int kind;
  float delta, ax, ay;
  int iters;              // items per thread
  // fm_flow_loss_fused_adam: the Adam update of `depth` applied by this very pass (depth, exp_avg, exp_avg_sq rewritten
  // in place) for every pixel whose bit in `touched` is clear; the others keep their values and get dL/ddepth written
  float* depth_rw;        // = depth
  float* exp_avg;         // (B,F,H,W)
  float* exp_avg_sq;      // (B,F,H,W)
  const uint8_t* touched; // (B·F·H·W/4): bit e of byte q = pixel 4q+e receives gradient from (or is read by) another operator
  AdamCoef adam;
  // fm_flow_loss_fused_taps: the STATIC tap set of the tracking loss (fm_flow_taps, include/flowmap_hip.h), M pixels ranked in (frame,
  // pixel) order — tap_chunk_base: the rank of the first tap at or after quad 64·c of frame bf (one more entry at the end: M);
  // tap_pixel: each tap's pixel index inside its frame.  A workgroup owns a run of consecutive quads of one frame, hence a run of
  // consecutive taps: it stages tap_scale[0]·tap_grad of its taps in an LDS image of its pixels before its main loop (added into
  // dL/ddepth with one LDS read per quad), collects the depth it leaves behind in a second LDS image and writes its taps' values to
  // tap_depth afterwards — coalesced on the tap side, branch-free in the loop, no cold line of the images touched by the tracking loss.
  const int32_t* tap_chunk_base;
  const int32_t* tap_pixel;
  const float* tap_grad;
  const float* tap_scale;
  float* tap_depth;
  int* tap_stale;  // raised when a tap depth differs from the value tap_depth held on entry (the image the tracking loss sampled from)
  // element strides between frames / batch entries of the caller's image stacks (fm_layout; dense when the caller gave none):
  // depth, flow_fwd, flow_bwd, mask_fwd, mask_bwd — a frame window x[:, s:s+f] of a larger tensor is read in place
  long fs[5], bs[5];
};

// 16-byte streaming accesses.  Every input element is read exactly once and the gradient is
// written exactly once, so the accesses carry the non-temporal hint (measured on C1:
// 0.90 -> 0.83 ms; loads-only 0.86, stores-only 0.85).  -DFM_FLOW_NO_NT turns it off.
#if !defined(FM_FLOW_NO_NT) && !defined(FM_FLOW_NT_LOAD) && !defined(FM_FLOW_NT_STORE)
#define FM_FLOW_NT 1
#endif
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#if defined(FM_FLOW_NT) || defined(FM_FLOW_NT_LOAD)
#define FM_LOAD(p) __builtin_nontemporal_load(p)
#else
#define FM_LOAD(p) (*(p))
#endif
#if defined(FM_FLOW_NT) || defined(FM_FLOW_NT_STORE)
#define FM_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define FM_STORE(v, p) (*(p) = (v))
#endif
__device__ __forceinline__ v4f ld4(const float* base, int i) { return FM_LOAD(reinterpret_cast<const v4f*>(base) + i); }
__device__ __forceinline__ v2f ld2(const float* base, int i) { return FM_LOAD(reinterpret_cast<const v2f*>(base) + i); }

// One quad's raw inputs, kept as the 16-byte vectors they were loaded as.
struct QuadIn {
  v4f z, fa, fc, fm, ba, bc, bm;
};

// Reference layout: five separate arrays (depth, 2 flows, 2 masks) = 7 streams of 16 B per lane.
__device__ __forceinline__ void load_quad(QuadIn& q, const float* depth, const float* ff, const float* mf, const float* fb,
                                          const float* mb, int item, bool has_fwd, bool has_bwd) {
  q.z = ld4(depth, item);
  if (has_fwd) {
    q.fa = ld4(ff, item * 2);
    q.fc = ld4(ff, item * 2 + 1);
    q.fm = ld4(mf, item);
  }
  if (has_bwd) {
    q.ba = ld4(fb, item * 2);
    q.bc = ld4(fb, item * 2 + 1);
    q.bm = ld4(mb, item);
  }
}

// Packed layout (fm_flow_pack_inputs): the six flow / mask vectors of the 64 quads a wave
// handles are ONE contiguous 6 KB chunk, [chunk][vector 0..5][lane], so a wave's six loads hit
// consecutive kilobytes of one stream instead of six arrays hundreds of MB apart.
constexpr int kPackLanes = 64;
constexpr int kPackVecs = 6;

__device__ __forceinline__ void load_quad_packed(QuadIn& q, const float* depth, const float* packed, int item, bool has_fwd,
                                                 bool has_bwd) {
  q.z = ld4(depth, item);
  const int base = (item / kPackLanes) * (kPackVecs * kPackLanes) + (item % kPackLanes);
  if (has_fwd) {
    q.fa = ld4(packed, base);
    q.fc = ld4(packed, base + kPackLanes);
    q.fm = ld4(packed, base + 2 * kPackLanes);
  }
  if (has_bwd) {
    q.ba = ld4(packed, base + 3 * kPackLanes);
    q.bc = ld4(packed, base + 4 * kPackLanes);
    q.bm = ld4(packed, base + 5 * kPackLanes);
  }
}

// ---------------------------------------------------------------------------------
// The two residual terms of a pixel — towards the next frame and towards the previous one — as ONE pass of packed fp32
// arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  The two terms run the same ~60 operations on different constants,
// flows and masks, so component x carries the forward term and component y the backward one; no run-time direction tests, 32 fewer
// instructions per pixel.  Each component performs exactly the operations of flow_term_fast (fm_math.h) in the same order: the
// results are bit-identical to the scalar form the host double runs.  (What it does NOT buy on this part is issue time: a packed
// instruction occupies the SIMD about twice as long as a plain one — tools/probes/pk_rate_probe.hip — and the kernel waits for HBM
// in either form; -DFM_FLOW_SCALAR_TERMS keeps the scalar form for A/B runs, tools/ab_flow.sh.  DESIGN.md §3.1.)
// ---------------------------------------------------------------------------------
struct DirPair {
  v2f au, a1, a2, ta, bu, b1, b2, tb, cu, c1, c2, tc;
};
__device__ __forceinline__ v2f pk(float x, float y) {
  v2f r;
  r.x = x, r.y = y;
  return r;
}
__device__ __forceinline__ v2f pk1(float x) { return pk(x, x); }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// kf / kb: 1 for a direction the frame has, 0 for one it has not (its constants are zeroed, whatever they were computed from)
__device__ __forceinline__ DirPair make_pair(const DirConst& f, const DirConst& b, float kf, float kb) {
  DirPair d;
  auto two = [&](float x, float y) { return pk(kf != 0.f ? x : 0.f, kb != 0.f ? y : 0.f); };
  d.au = two(f.au, b.au), d.a1 = two(f.a1, b.a1), d.a2 = two(f.a2, b.a2), d.ta = two(f.ta, b.ta);
  d.bu = two(f.bu, b.bu), d.b1 = two(f.b1, b.b1), d.b2 = two(f.b2, b.b2), d.tb = two(f.tb, b.tb);
  d.cu = two(f.cu, b.cu), d.c1 = two(f.c1, b.c1), d.c2 = two(f.c2, b.c2), d.tc = two(f.tc, b.tc);
  return d;
}

// arow / brow / crow = a1·v + a2 etc. (constant along an image row).  A direction the frame does not have (first / last frame
// of the video) runs on zero constants, zero flows and a zero mask: every sum it adds is exactly zero.
template <int KIND, bool GRAD>
__device__ __forceinline__ void flow_term_pair(const DirPair& d, v2f arow, v2f brow, v2f crow, float z, float u, float zu, float zv, float u_ax,
                                               float v_ay, v2f flow_x, v2f flow_y, v2f m, float scale, float delta, float inv_delta,
                                               float ax, float ay, v2f (&acc)[kFlowAcc], float& gz) {
  const v2f u2 = pk1(u), z2 = pk1(z);
  const v2f a = pk_fma(d.au, u2, arow);
  const v2f b = pk_fma(d.bu, u2, brow);
  const v2f c = pk_fma(d.cu, u2, crow);
  const v2f xu = pk_fma(z2, a, d.ta);
  const v2f xv = pk_fma(z2, b, d.tb);
  const v2f x2 = pk_fma(z2, c, d.tc);
  const v2f den = x2 + pk1(kProjEps);
  v2f q = pk(fm_rcp(den.x), fm_rcp(den.y));
  const bool ok_x = fabsf(q.x) <= 3.0e38f, ok_y = fabsf(q.y) <= 3.0e38f;
  q = pk(ok_x ? q.x : 0.f, ok_y ? q.y : 0.f);
  m = pk(ok_x ? m.x : 0.f, ok_y ? m.y : 0.f);
  const v2f pu = xu * q;
  const v2f pv = xv * q;
  const v2f rx = pu - pk_fma(flow_x, pk1(ax), pk1(u_ax));
  const v2f ry = pv - pk_fma(flow_y, pk1(ay), pk1(v_ay));
  const v2f ss = pk_fma(rx, rx, ry * ry);
  v2f rho, coef;  // ρ and dρ/dr = coef·r
  if (KIND == kL2) {
    rho = pk1(0.5f) * ss;
    coef = pk1(1.f);
  } else {
    const v2f inv_n = pk(ss.x > 0.f ? fm_rsq(ss.x) : 0.f, ss.y > 0.f ? fm_rsq(ss.y) : 0.f);
    const v2f n = ss * inv_n;
    if (KIND == kL1) {
      rho = n;
      coef = inv_n;
    } else {
      const bool quad_x = n.x < delta, quad_y = n.y < delta;
      const v2f inner = (pk1(0.5f) * ss) * pk1(inv_delta), outer = n - pk1(0.5f * delta);
      rho = pk(quad_x ? inner.x : outer.x, quad_y ? inner.y : outer.y);
      coef = pk(quad_x ? inv_delta : inv_n.x, quad_y ? inv_delta : inv_n.y);
    }
  }
  acc[0] = pk_fma(rho, m, acc[0]);
  if (GRAD) {
    const v2f gc = (pk1(scale) * m) * coef;
    const v2f wu = gc * rx, wv = gc * ry;  // dL/d(kd·p)
    const v2f o0 = q * wu, o1 = q * wv, o2 = q * pk_fma(wu, pu, wv * pv);
    const v2f zu2 = pk1(zu), zv2 = pk1(zv);
    acc[1] += o0;
    acc[2] += o1;
    acc[3] += o2;
    acc[4] = pk_fma(o0, zu2, acc[4]);
    acc[5] = pk_fma(o0, zv2, acc[5]);
    acc[6] = pk_fma(o0, z2, acc[6]);
    acc[7] = pk_fma(o1, zu2, acc[7]);
    acc[8] = pk_fma(o1, zv2, acc[8]);
    acc[9] = pk_fma(o1, z2, acc[9]);
    acc[10] = pk_fma(o2, zu2, acc[10]);
    acc[11] = pk_fma(o2, zv2, acc[11]);
    acc[12] = pk_fma(o2, z2, acc[12]);
    const v2f g = pk_fma(o0, a, pk_fma(o1, b, -o2 * c));  // dL/dz = ω·(a, b, −c) of each direction
    gz += g.x;
    gz += g.y;
  }
}

#ifndef FM_FLOW_WAVES
// Waves per SIMD the register allocator must leave room for.  3 (<=168 VGPRs; the packed pair of terms uses 150, the scalar
// form 106, no scratch).  Round 2's scalar kernel: 3 beat 4 by 3-5 % and 2 by 4 %; round 3 (tools/flow_microbench.py, tools/ab_flow.sh):
// 2, 3 and (scalar form) 4 within the run-to-run spread — the kernel waits for HBM either way.
#define FM_FLOW_WAVES 3
#endif

template <int VEC, int KIND, bool GRAD, bool PACKED, bool ADAM = false, bool TAPS = false>
__global__ void __launch_bounds__(256, FM_FLOW_WAVES) flow_fused_kernel(FlowParams p) {
  static_assert(!ADAM || (VEC == 4 && GRAD), "the in-pass Adam update runs on the 16-byte gradient path");
  static_assert(!TAPS || (VEC == 4 && GRAD), "the tap exchange runs on the 16-byte gradient path");
  extern __shared__ double lds[];  // reduction scratch (fp64), then the [width] u-table, then (TAPS) the two tap images of the block's pixels
  double* red = lds;
  float* u_tab = reinterpret_cast<float*>(lds + (256 / 64) * kFlowAcc);
  v4f* tap_g4 = reinterpret_cast<v4f*>(u_tab + ((p.width + 3) & ~3));  // [256·iters] quads: scale·(tracking gradient) at the block's pixels, 0 off the taps
  v4f* tap_z4 = tap_g4 + 256 * p.iters;                                 // [256·iters] quads: the depth this pass leaves at the block's pixels

  const int bf = blockIdx.y;  // batch*frames + frame
  const int f = bf % p.frames;
  const int b = bf / p.frames;
  const bool has_fwd = f < p.frames - 1;
  const bool has_bwd = f > 0;
  const int n = p.height * p.width;
  const int items = n / VEC;
  const int items_per_row = p.width / VEC;

  for (int c = threadIdx.x; c < p.width; c += blockDim.x) u_tab[c] = pixel_center(c, p.width);
  int tap_first = 0, tap_end = 0;  // the block's run of taps
  const int block_quad0 = blockIdx.x * (blockDim.x * p.iters);
  if constexpr (TAPS) {
    const size_t chunk_count = ((size_t)items + kPackLanes - 1) / kPackLanes;
    const size_t c0 = (size_t)bf * chunk_count + (size_t)(block_quad0 >> 6);
    const size_t c1 = (size_t)bf * chunk_count + min((size_t)(block_quad0 >> 6) + (size_t)(blockDim.x * p.iters >> 6), chunk_count);
    tap_first = p.tap_chunk_base[c0];
    tap_end = p.tap_chunk_base[c1];
    if (p.tap_grad) {
      const v4f zero = {0.f, 0.f, 0.f, 0.f};
      for (int it = 0; it < p.iters; ++it) tap_g4[it * blockDim.x + threadIdx.x] = zero;
    }
  }
  __syncthreads();

  const size_t pair_f = (size_t)b * (p.frames - 1) + f;  // pair whose earlier frame is f
  const size_t pair_b = pair_f - 1;                       // pair whose later frame is f
  DirConst df, db;
  {
    // (both directions unconditionally, on in-range stand-ins where the frame has no such neighbour — conditional initialisation
    // of the two structs costs a scratch allocation; the stand-in's constants are zeroed in make_pair)
    Mat3 kinv, kd;
    Pose t;
    load_mat3(p.kinv + (size_t)bf * 9, kinv);
    load_mat3(p.k + (size_t)(has_fwd ? bf + 1 : bf) * 9, kd);
    load_pose44(p.t_fwd + (has_fwd ? pair_f : pair_b) * 16, t);
    make_dir(t, kinv, kd, p.ax, p.ay, df);
    load_mat3(p.k + (size_t)(has_bwd ? bf - 1 : bf) * 9, kd);
    load_pose44(p.t_bwd + (has_bwd ? pair_b : pair_f) * 16, t);
    make_dir(t, kinv, kd, p.ax, p.ay, db);
  }
  const DirPair dp = make_pair(df, db, has_fwd ? 1.f : 0.f, has_bwd ? 1.f : 0.f);  // (a direction this frame does not have: zeros)
  const float scale = GRAD ? p.scale[0] : 0.f;
  const float inv_delta = KIND == kHuber ? 1.0f / p.delta : 0.f;

  const float* depth = p.depth + (size_t)b * p.bs[0] + (size_t)f * p.fs[0];
  const float* ff = p.flow_fwd + (size_t)b * p.bs[1] + (size_t)f * p.fs[1];  // pair f of this batch entry
  const float* mf = p.mask_fwd + (size_t)b * p.bs[3] + (size_t)f * p.fs[3];
  const float* fb = p.flow_bwd + (size_t)b * p.bs[2] + (size_t)(f - 1) * p.fs[2];  // pair f−1 (never dereferenced for f = 0)
  const float* mb = p.mask_bwd + (size_t)b * p.bs[4] + (size_t)(f - 1) * p.fs[4];
  const size_t chunks = ((size_t)items + kPackLanes - 1) / kPackLanes;
  const float* packed = PACKED ? p.packed + (size_t)bf * chunks * (kPackVecs * kPackLanes * 4) : nullptr;
  float* gd = GRAD && p.grad_depth ? p.grad_depth + (size_t)bf * n : nullptr;

  v2f acc[kFlowAcc];  // x: towards the next frame, y: towards the previous one
#pragma unroll
  for (int i = 0; i < kFlowAcc; ++i) acc[i] = pk1(0.f);
#ifdef FM_FLOW_SCALAR_TERMS
  float sacc_f[kFlowAcc], sacc_b[kFlowAcc];
#pragma unroll
  for (int i = 0; i < kFlowAcc; ++i) sacc_f[i] = sacc_b[i] = 0.f;
#endif

  const int base = blockIdx.x * (blockDim.x * p.iters);

  // Everything after the loads of one item: coordinates, both residual terms per pixel, store.
  auto compute = [&](const float (&z)[VEC], const float (&fxf)[VEC], const float (&fyf)[VEC], const float (&mmf)[VEC],
                     const float (&fxb)[VEC], const float (&fyb)[VEC], const float (&mmb)[VEC], int item) {
    float gz[VEC];
    const int row = item / items_per_row;
    const int col0 = (item - row * items_per_row) * VEC;
    const float v = pixel_center(row, p.height);
    const float v_ay = v * p.ay;
    // a1·v + a2 (and b, c alike) is constant along the image row the quad lies in
#ifdef FM_FLOW_SCALAR_TERMS  // (A/B: the two directions one after the other in plain fp32, behind wave-uniform run-time tests)
    const float rf0 = fmaf(df.a1, v, df.a2), rf1 = fmaf(df.b1, v, df.b2), rf2 = fmaf(df.c1, v, df.c2);
    const float rb0 = fmaf(db.a1, v, db.a2), rb1 = fmaf(db.b1, v, db.b2), rb2 = fmaf(db.c1, v, db.c2);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float u = u_tab[col0 + e];
      const float zu = z[e] * u, zv = z[e] * v, u_ax = u * p.ax;
      gz[e] = 0.f;
      if (has_fwd)
        flow_term_fast<KIND, GRAD>(df, rf0, rf1, rf2, z[e], u, zu, zv, u_ax, v_ay, fxf[e], fyf[e], mmf[e], scale, p.delta, inv_delta, p.ax, p.ay, sacc_f, gz[e]);
      if (has_bwd)
        flow_term_fast<KIND, GRAD>(db, rb0, rb1, rb2, z[e], u, zu, zv, u_ax, v_ay, fxb[e], fyb[e], mmb[e], scale, p.delta, inv_delta, p.ax, p.ay, sacc_b, gz[e]);
    }
#else
    const v2f v2 = pk1(v);
    const v2f arow = pk_fma(dp.a1, v2, dp.a2), brow = pk_fma(dp.b1, v2, dp.b2), crow = pk_fma(dp.c1, v2, dp.c2);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float u = u_tab[col0 + e];
      const float zu = z[e] * u, zv = z[e] * v, u_ax = u * p.ax;
      gz[e] = 0.f;
      flow_term_pair<KIND, GRAD>(dp, arow, brow, crow, z[e], u, zu, zv, u_ax, v_ay, pk(fxf[e], fxb[e]), pk(fyf[e], fyb[e]), pk(mmf[e], mmb[e]),
                                 scale, p.delta, inv_delta, p.ax, p.ay, acc, gz[e]);
    }
#endif
    const int local_quad = item - block_quad0;
    if (TAPS) {
      if (p.tap_grad) {  // (wave-uniform) the tracking loss's share of dL/ddepth at this quad's pixels: zero off the taps
        const v4f tg = tap_g4[local_quad];
#pragma unroll
        for (int e = 0; e < VEC; ++e) gz[e] += tg[e];
      }
    }
    if (ADAM) {
      // depth, exp_avg, exp_avg_sq of this quad rewritten in place (model_wrapper_overfit.py:104-105: torch.optim.Adam);
      // pixels another operator still reads from / adds gradient to keep their values and get dL/ddepth stored instead
      const size_t q = (size_t)bf * items + item;
      const unsigned bits = p.touched[q];
      v4f m4 = FM_LOAD(reinterpret_cast<const v4f*>(p.exp_avg) + q), v4 = FM_LOAD(reinterpret_cast<const v4f*>(p.exp_avg_sq) + q);
      v4f z4;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float pe = z[e], me = m4[e], ve = v4[e];
        adam_update(p.adam, pe, gz[e], me, ve);
        const bool keep = (bits >> e) & 1u;
        z4[e] = keep ? z[e] : pe;
        m4[e] = keep ? m4[e] : me;
        v4[e] = keep ? v4[e] : ve;
        if (keep) gd[(size_t)item * VEC + e] = gz[e];
      }
      FM_STORE(z4, reinterpret_cast<v4f*>(p.depth_rw) + q);
      FM_STORE(m4, reinterpret_cast<v4f*>(p.exp_avg) + q);
      FM_STORE(v4, reinterpret_cast<v4f*>(p.exp_avg_sq) + q);
      if (TAPS) tap_z4[local_quad] = z4;  // (a pixel another operator keeps is refreshed by its own update: the caller knows which)
    } else if (GRAD && gd) {
      if (TAPS) {
        v4f zq;
        zq.x = z[0]; zq.y = z[1 % VEC]; zq.z = z[2 % VEC]; zq.w = z[3 % VEC];
        tap_z4[local_quad] = zq;
      }
      if (VEC == 4) {
        v4f o;
        o.x = gz[0]; o.y = gz[1]; o.z = gz[2]; o.w = gz[3];
        FM_STORE(o, reinterpret_cast<v4f*>(gd) + item);
      } else if (VEC == 2) {
        v2f o;
        o.x = gz[0]; o.y = gz[VEC - 1];
        FM_STORE(o, reinterpret_cast<v2f*>(gd) + item);
      } else {
        gd[item] = gz[0];
      }
    }
  };

  auto compute_quad = [&](const QuadIn& q, int item) {
    float z[VEC], fxf[VEC], fyf[VEC], mmf[VEC], fxb[VEC], fyb[VEC], mmb[VEC];
    if (VEC == 4) {
      z[0] = q.z.x; z[1 % VEC] = q.z.y; z[2 % VEC] = q.z.z; z[3 % VEC] = q.z.w;
      fxf[0] = q.fa.x; fyf[0] = q.fa.y; fxf[1 % VEC] = q.fa.z; fyf[1 % VEC] = q.fa.w;
      fxf[2 % VEC] = q.fc.x; fyf[2 % VEC] = q.fc.y; fxf[3 % VEC] = q.fc.z; fyf[3 % VEC] = q.fc.w;
      mmf[0] = q.fm.x; mmf[1 % VEC] = q.fm.y; mmf[2 % VEC] = q.fm.z; mmf[3 % VEC] = q.fm.w;
      mmb[0] = q.bm.x; mmb[1 % VEC] = q.bm.y; mmb[2 % VEC] = q.bm.z; mmb[3 % VEC] = q.bm.w;
      fxb[0] = q.ba.x; fyb[0] = q.ba.y; fxb[1 % VEC] = q.ba.z; fyb[1 % VEC] = q.ba.w;
      fxb[2 % VEC] = q.bc.x; fyb[2 % VEC] = q.bc.y; fxb[3 % VEC] = q.bc.z; fyb[3 % VEC] = q.bc.w;
    }
    compute(z, fxf, fyf, mmf, fxb, fyb, mmb, item);
  };

  // (A depth-2 software pipeline of the loads — two QuadIn register sets, loop unrolled by
  // two, 168 VGPRs — measured 0.856 vs 0.845 ms: no gain, removed.)
  // The tap exchange's set-up costs a workgroup two dependent round trips (its taps' pixels and gradients, then the LDS image): they are
  // requested BEHIND the first quad's own seven loads, so that they travel together, and the image is filled while those loads are still
  // in flight (filled before them, as the first version did, the set-up delayed every workgroup's first load: +70 us on the 0.8 ms pass).
  QuadIn q_first = {};
  int my_tap = -1;  // this thread's tap of the workgroup's run (the taps past 256 per workgroup, if any, are walked in loops)
  int my_tap_pixel = 0;
  if constexpr (TAPS) {
    float my_tap_grad = 0.f;
    if (tap_first + (int)threadIdx.x < tap_end) {  // (requested first: the quad's loads below are waited for one by one — the memory counter is in order)
      my_tap = tap_first + (int)threadIdx.x;
      my_tap_pixel = p.tap_pixel[my_tap];
      if (p.tap_grad) my_tap_grad = p.tap_grad[my_tap];
    }
    const int item0 = base + (int)threadIdx.x;
    if (item0 < items) {
      if constexpr (PACKED) load_quad_packed(q_first, depth, packed, item0, has_fwd, has_bwd);
      else load_quad(q_first, depth, ff, mf, fb, mb, item0, has_fwd, has_bwd);
    }
    my_tap_pixel -= 4 * block_quad0;
    if (p.tap_grad) {
      const float ts = p.tap_scale[0];
      float* tap_g = reinterpret_cast<float*>(tap_g4);
      if (my_tap >= 0) tap_g[my_tap_pixel] = ts * my_tap_grad;
      for (int t = tap_first + (int)blockDim.x + (int)threadIdx.x; t < tap_end; t += blockDim.x) tap_g[p.tap_pixel[t] - 4 * block_quad0] = ts * p.tap_grad[t];
      __syncthreads();
    }
  }
  for (int it = 0; it < p.iters; ++it) {
    const int item = base + it * blockDim.x + threadIdx.x;
    if (item >= items) break;
    if (VEC == 4) {
      QuadIn q = {};
      if (TAPS && it == 0) q = q_first;
      else if constexpr (PACKED) load_quad_packed(q, depth, packed, item, has_fwd, has_bwd);
      else load_quad(q, depth, ff, mf, fb, mb, item, has_fwd, has_bwd);
      compute_quad(q, item);
      continue;
    }
    float z[VEC], fxf[VEC] = {}, fyf[VEC] = {}, mmf[VEC] = {}, fxb[VEC] = {}, fyb[VEC] = {}, mmb[VEC] = {};  // (an absent direction: zeros)
    if (VEC == 2) {
      const v2f zq = ld2(depth, item);
      z[0] = zq.x; z[VEC - 1] = zq.y;
      if (has_fwd) {
        const v4f a = ld4(ff, item);
        const v2f mq = ld2(mf, item);
        fxf[0] = a.x; fyf[0] = a.y; fxf[VEC - 1] = a.z; fyf[VEC - 1] = a.w;
        mmf[0] = mq.x; mmf[VEC - 1] = mq.y;
      }
      if (has_bwd) {
        const v4f a = ld4(fb, item);
        const v2f mq = ld2(mb, item);
        fxb[0] = a.x; fyb[0] = a.y; fxb[VEC - 1] = a.z; fyb[VEC - 1] = a.w;
        mmb[0] = mq.x; mmb[VEC - 1] = mq.y;
      }
    } else {
      z[0] = depth[item];
      if (has_fwd) {
        const float2 a = reinterpret_cast<const float2*>(ff)[item];
        fxf[0] = a.x; fyf[0] = a.y; mmf[0] = mf[item];
      }
      if (has_bwd) {
        const float2 a = reinterpret_cast<const float2*>(fb)[item];
        fxb[0] = a.x; fyb[0] = a.y; mmb[0] = mb[item];
      }
    }
    compute(z, fxf, fyf, mmf, fxb, fyb, mmb, item);
  }

  if constexpr (TAPS) {
    if (p.tap_depth) {  // the block's taps: the depth this pass leaves at each, from the LDS image (coalesced on the tap side)
      __syncthreads();
      const float* tap_z = reinterpret_cast<const float*>(tap_z4);
      if (my_tap >= 0) {
        const float znew = tap_z[my_tap_pixel];
        if (!ADAM && p.tap_stale && __float_as_uint(p.tap_depth[my_tap]) != __float_as_uint(znew)) *p.tap_stale = 1;
        p.tap_depth[my_tap] = znew;
      }
      for (int t = tap_first + (int)blockDim.x + (int)threadIdx.x; t < tap_end; t += blockDim.x) {
        const float znew = tap_z[p.tap_pixel[t] - 4 * block_quad0];
        if (!ADAM && p.tap_stale && __float_as_uint(p.tap_depth[t]) != __float_as_uint(znew)) *p.tap_stale = 1;
        p.tap_depth[t] = znew;
      }
    }
  }
  double* dst = p.acc + (size_t)bf * 2 * kFlowAccStride;
  float acc_f[kFlowAcc], acc_b[kFlowAcc];
#pragma unroll
  for (int i = 0; i < kFlowAcc; ++i) acc_f[i] = acc[i].x, acc_b[i] = acc[i].y;
#ifdef FM_FLOW_SCALAR_TERMS
#pragma unroll
  for (int i = 0; i < kFlowAcc; ++i) acc_f[i] = sacc_f[i], acc_b[i] = sacc_b[i];
#endif
  if (has_fwd) block_accumulate<kFlowAcc>(acc_f, red, dst);
  if (has_bwd) block_accumulate<kFlowAcc>(acc_b, red, dst + kFlowAccStride);
}

// ---------------------------------------------------------------------------------
// Finalize: turn the per-(frame, direction) sums into the loss and the small gradients.
//   loss[0]      = weight · Σρm / V_eff                     (loss.py:47, loss_flow.py:70)
//   g_t_fwd/bwd  = dL/dT of the relative poses (B,F-1,4,4), bottom row 0
//   g_k          = dL/dK (B,F,3,3) through the destination role (rows 0,1) AND through
//                  K⁻¹ of the source role:  dK = −K⁻ᵀ · dKinv · K⁻ᵀ
// One thread per frame; loss summed by thread 0 of block 0 (B·F ≤ a few thousand).
// ---------------------------------------------------------------------------------
struct FlowFinalizeParams {
  double* acc;
  const float* k;
  const float* kinv;
  const float* t_fwd;
  const float* t_bwd;
  const float* norm;  // device: [0] = weight / V_eff (loss normaliser)
  float* loss;        // (1)
  float* g_t_fwd;     // (B,F-1,4,4)
  float* g_t_bwd;     // (B,F-1,4,4)
  float* g_k;         // (B,F,3,3)
  int batch, frames;
  float ax, ay;
};

// ONE block (frames read each other's sums — a frame's dL/dK has a destination-role part from its neighbours —
// so the clearing of `acc` must wait for every frame: a block-wide barrier).  Four neighbouring lanes share a frame, one
// per role (flow_finalize_role, fm_pose.h): the dependent fp64 chain per lane is one flow_dir_grads instead of four, and
// the four parts of dL/dK meet in a quad reduction (two DPP-free shuffles of 9 doubles).
constexpr int kFinalizeThreads = 1024;

__global__ void __launch_bounds__(kFinalizeThreads) flow_finalize_kernel(FlowFinalizeParams p) {
  const int total = p.batch * p.frames;
  const int role = threadIdx.x & 3;
  for (int bf0 = 0; bf0 < total; bf0 += kFinalizeThreads / 4) {  // (wave-uniform trip count)
    const int bf = bf0 + (int)(threadIdx.x >> 2);
    double gk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (bf < total) flow_finalize_role(p.acc, p.k, p.kinv, p.t_fwd, p.t_bwd, p.frames, bf, role, p.ax, p.ay, p.g_t_fwd, p.g_t_bwd, gk);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      gk[i] += __shfl_xor(gk[i], 1, kWave);  // (0 + 1), (2 + 3)
      gk[i] += __shfl_xor(gk[i], 2, kWave);  // (0 + 1) + (2 + 3)
    }
    if (bf < total && role == 0)
      for (int i = 0; i < 9; ++i) p.g_k[(size_t)bf * 9 + i] = (float)gk[i];
  }
  // loss numerator: sum over all (frame, direction) in fp64 by one wave
  double s = 0.0;
  if (threadIdx.x < kWave)
    for (int i = threadIdx.x; i < total * 2; i += kWave) s += p.acc[(size_t)i * kFlowAccStride];
  if (threadIdx.x < kWave) s = wave_sum(s);
  if (threadIdx.x == 0) p.loss[0] = (float)(s * (double)p.norm[0]);
  __syncthreads();
  for (int i = threadIdx.x; i < total * 2 * kFlowAccStride; i += blockDim.x) p.acc[i] = 0.0;  // clean for the next launch
}

// ---------------------------------------------------------------------------------
// Σ mask (loss_flow.py:56,66).  Masks never change during an optimisation, so the host
// caches the result per Flows object; this runs once.
// out[0] += Σ a + Σ b  (fp64)
// ---------------------------------------------------------------------------------
// a, b: image stacks of `frames_per_batch` frames of n elements per batch entry, frames fs_* and batch entries bs_* elements apart.
// grid: (chunks, batch · frames_per_batch).
__global__ void __launch_bounds__(256) sum2_kernel(const float* a, const float* b, long n, int frames_per_batch, long fs_a, long bs_a, long fs_b,
                                                   long bs_b, double* out) {
  const int fr = blockIdx.y % frames_per_batch, be = blockIdx.y / frames_per_batch;
  const float* pa = a + (size_t)be * bs_a + (size_t)fr * fs_a;
  const float* pb = b ? b + (size_t)be * bs_b + (size_t)fr * fs_b : nullptr;
  double s = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float t = pa[i];
    if (pb) t += pb[i];
    s += (double)t;
  }
  s = wave_sum(s);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// norm[0] = weight / (V or 1) (loss normaliser);  norm[1] = V_eff
__global__ void flow_norm_kernel(const double* vsum, float weight, float* norm) {
  const double v = vsum[0];
  const double veff = v != 0.0 ? v : 1.0;  // `valid_sum or 1` (loss_flow.py:70)
  norm[0] = (float)((double)weight / veff);
  norm[1] = (float)veff;
}

// One-time re-layout of the optimisation's constants (flows and masks never change after
// FlowPredictor.compute_bidirectional_flow): for source frame f, quad q = 4 consecutive pixels,
//   vec 0,1 = forward flow of pair f   (x0 y0 x1 y1 | x2 y2 x3 y3)     vec 2 = forward mask
//   vec 3,4 = backward flow of pair f-1                                  vec 5 = backward mask
// stored [frame][q/64][vec][q%64] as float4; absent pairs and the padding lanes are zero.
struct PackLayouts {
  long fs[4], bs[4];  // flow_fwd, flow_bwd, mask_fwd, mask_bwd
};

__global__ void __launch_bounds__(256) pack_inputs_kernel(const float* flow_fwd, const float* flow_bwd, const float* mask_fwd,
                                                          const float* mask_bwd, int frames, int n, float* packed, PackLayouts lay) {
  const int quads = n / 4;
  const int chunks = (quads + kPackLanes - 1) / kPackLanes;
  const int bf = blockIdx.y, f = bf % frames, b = bf / frames;
  const bool has_fwd = f < frames - 1, has_bwd = f > 0;
  const v4f zero = {0.f, 0.f, 0.f, 0.f};
  v4f* out = reinterpret_cast<v4f*>(packed) + (size_t)bf * chunks * (kPackVecs * kPackLanes);
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < chunks * kPackLanes; q += gridDim.x * blockDim.x) {
    const bool live = q < quads;
    v4f* o = out + (size_t)(q / kPackLanes) * (kPackVecs * kPackLanes) + (q % kPackLanes);
    const v4f* ff = reinterpret_cast<const v4f*>(flow_fwd + (size_t)b * lay.bs[0] + (size_t)f * lay.fs[0]);
    const v4f* fb = reinterpret_cast<const v4f*>(flow_bwd + (size_t)b * lay.bs[1] + (size_t)(f - 1) * lay.fs[1]);
    const v4f* mf = reinterpret_cast<const v4f*>(mask_fwd + (size_t)b * lay.bs[2] + (size_t)f * lay.fs[2]);
    const v4f* mb = reinterpret_cast<const v4f*>(mask_bwd + (size_t)b * lay.bs[3] + (size_t)(f - 1) * lay.fs[3]);
    o[0 * kPackLanes] = live && has_fwd ? ff[2 * q] : zero;
    o[1 * kPackLanes] = live && has_fwd ? ff[2 * q + 1] : zero;
    o[2 * kPackLanes] = live && has_fwd ? mf[q] : zero;
    o[3 * kPackLanes] = live && has_bwd ? fb[2 * q] : zero;
    o[4 * kPackLanes] = live && has_bwd ? fb[2 * q + 1] : zero;
    o[5 * kPackLanes] = live && has_bwd ? mb[q] : zero;
  }
}

// In-place scale of gradient buffers by a device scalar, skipped entirely when the
// scalar is exactly 1 (the autograd root case): every block exits after one load.
__global__ void __launch_bounds__(256) scale_if_needed_kernel(float* x, long n, float* y, long ny, const float* s, int* not_one) {
  const float sv = s[0];
  if (sv == 1.0f) return;
  if (not_one && blockIdx.x == 0 && threadIdx.x == 0) *not_one = 1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n + ny; i += (long)gridDim.x * blockDim.x) {
    if (i < n) x[i] *= sv;
    else y[i - n] *= sv;
  }
}

}  // namespace fm

using namespace fm;

extern "C" {

struct FlowAdam {
  float* exp_avg;
  float* exp_avg_sq;
  const uint8_t* touched;
  AdamCoef coef;
};

static int flow_loss_launch(const float* depth, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd,
                            const float* flow_fwd, const float* flow_bwd, const float* mask_fwd, const float* mask_bwd,
                            const float* packed, const float* scale, int batch, int frames, int height, int width,
                            int mapping_kind, float delta, float aspect_x, float aspect_y, float* grad_depth, double* acc,
                            int items_per_thread, const FlowAdam* adam, const fm_layout* layouts, void* stream, const fm_flow_taps* taps = nullptr) {
  FM_CHECK_ARG(depth && k && kinv && acc);
  FM_CHECK_ARG(!(layouts && adam));  // the in-pass update rewrites the depth PARAMETER: dense by construction
  FM_CHECK_ARG(!taps || (!layouts && taps->chunk_base && taps->pixel && scale && grad_depth && width % 4 == 0 && (taps->grad == nullptr || taps->scale != nullptr)));
  FM_CHECK_ARG(packed || (flow_fwd && flow_bwd && mask_fwd && mask_bwd));
  FM_CHECK_ARG(batch >= 1 && frames >= 2 && height >= 1 && width >= 1);
  FM_CHECK_ARG(mapping_kind >= 0 && mapping_kind <= 2);
  FM_CHECK_ARG((long)height * width < (1L << 30));
  FM_CHECK_ARG((long)batch * frames <= 65535);
  hipStream_t st = (hipStream_t)stream;
  const bool grad = scale != nullptr;
  FlowParams p{depth, k, kinv, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, packed, scale, grad_depth, acc,
               frames, height, width, mapping_kind, delta, aspect_x, aspect_y, items_per_thread > 0 ? items_per_thread : 4};
  {  // element strides of the five image stacks (dense unless the caller described a view)
    const long n1 = (long)height * width;
    const long per_frame[5] = {n1, 2 * n1, 2 * n1, n1, n1};
    const long frames_of[5] = {frames, frames - 1, frames - 1, frames - 1, frames - 1};
    for (int i = 0; i < 5; ++i) {
      const bool given = layouts && (layouts[i].frame_stride != 0 || layouts[i].batch_stride != 0);
      p.fs[i] = given ? layouts[i].frame_stride : per_frame[i];
      p.bs[i] = given ? layouts[i].batch_stride : per_frame[i] * frames_of[i];
      FM_CHECK_ARG(p.fs[i] >= per_frame[i] && (batch == 1 || p.bs[i] >= p.fs[i] * (frames_of[i] - 1) + per_frame[i]));
    }
  }
  if (adam) {
    p.depth_rw = const_cast<float*>(depth);
    p.exp_avg = adam->exp_avg;
    p.exp_avg_sq = adam->exp_avg_sq;
    p.touched = adam->touched;
    p.adam = adam->coef;
  }
  if (taps) {
    p.tap_chunk_base = taps->chunk_base;
    p.tap_pixel = taps->pixel;
    p.tap_grad = taps->grad;
    p.tap_scale = taps->scale;
    p.tap_depth = taps->depth;
    p.tap_stale = taps->stale;
  }
  // (`acc` is zero on entry: fm_flow_loss_finalize clears what it has read, so a workspace kept across steps never
  // needs a memset launch)
  auto aligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool use_packed = packed != nullptr;
  bool strides16 = true;  // every frame of a view starts on a 16-byte boundary (the 16-byte path's loads)
  for (int i = 0; i < (use_packed ? 1 : 5); ++i) strides16 = strides16 && p.fs[i] % 4 == 0 && p.bs[i] % 4 == 0;
  FM_CHECK_ARG(!use_packed || (width % 4 == 0 && aligned(packed) && aligned(depth) && strides16 && (!grad_depth || aligned(grad_depth))));
  const bool vec4 = use_packed || ((width % 4 == 0) && strides16 && aligned(depth) && aligned(flow_fwd) && aligned(flow_bwd) && aligned(mask_fwd) &&
                                   aligned(mask_bwd) && (!grad_depth || aligned(grad_depth)));
#ifdef FM_FLOW_FORCE_VEC2
  const int vec = vec4 ? 2 : 1;
#else
  const int vec = vec4 ? 4 : 1;
#endif
  const long items = (long)height * width / vec;
  const int threads = 256;
  if (items_per_thread <= 0) {
    // Quads per thread.  At C1 (230 400 quads per frame): 2: 0.795, 3: 0.751, 4: 0.715-0.779, 5: 0.720,
    // 6: 0.793, 8-12: 0.835 ms.  A partially filled tail block per frame costs 2-5 % (1080p, 518 400
    // quads: 3 -> 675 full blocks 1.68 ms, 4 -> 506.25 blocks 1.73 ms, 6 -> 337.5 blocks 1.82 ms), so
    // take the first of 4, 3, 5 that tiles the frame exactly.
    p.iters = 4;
    // (with the tap exchange a workgroup pays a fixed set-up — its taps' LDS images, two barriers —: five quads per thread first.  C2,
    // 230 400 quads per frame: 0.78 ms against 0.80 with four, 0.83 with three, 0.95 with six; gpurun_out r04r / r04s)
    const int order_plain[3] = {4, 3, 5}, order_taps[3] = {5, 4, 3};
    for (int cand : (taps ? order_taps : order_plain))
      if (items % ((long)threads * cand) == 0) {
        p.iters = cand;
        break;
      }
  }
  const long per_block = (long)threads * p.iters;
  dim3 grid((unsigned)((items + per_block - 1) / per_block), (unsigned)(batch * frames));
  FM_CHECK_ARG(!taps || p.iters <= 6);  // (the two LDS images of a workgroup's pixels: 2 x 4 KB per quad per thread)
  size_t lds = sizeof(float) * (size_t)width + sizeof(double) * (threads / 64) * kFlowAcc;
  if (taps) lds = sizeof(float) * (size_t)((width + 3) & ~3) + sizeof(double) * (threads / 64) * kFlowAcc + 2 * sizeof(float) * 4 * (size_t)threads * p.iters;
#define FM_FLOW_LAUNCH(V, K, P)                                                                              \
  do {                                                                                                       \
    if (taps) {                                                                                              \
      if constexpr (V == 4) {                                                                                \
        if (adam) hipLaunchKernelGGL((flow_fused_kernel<4, K, true, P, true, true>), grid, dim3(threads), lds, st, p);  \
        else hipLaunchKernelGGL((flow_fused_kernel<4, K, true, P, false, true>), grid, dim3(threads), lds, st, p);      \
      }                                                                                                      \
    } else if (adam) {                                                                                       \
      if constexpr (V == 4) hipLaunchKernelGGL((flow_fused_kernel<4, K, true, P, true>), grid, dim3(threads), lds, st, p); \
    } else if (grad) hipLaunchKernelGGL((flow_fused_kernel<V, K, true, P>), grid, dim3(threads), lds, st, p);  \
    else hipLaunchKernelGGL((flow_fused_kernel<V, K, false, P>), grid, dim3(threads), lds, st, p);          \
  } while (0)
#define FM_FLOW_KIND(V, P)                                          \
  do {                                                              \
    if (mapping_kind == kHuber) FM_FLOW_LAUNCH(V, kHuber, P);       \
    else if (mapping_kind == kL1) FM_FLOW_LAUNCH(V, kL1, P);        \
    else FM_FLOW_LAUNCH(V, kL2, P);                                 \
  } while (0)
#define FM_FLOW_VEC(V)                 \
  do {                                 \
    if (use_packed) FM_FLOW_KIND(V, true); \
    else FM_FLOW_KIND(V, false);       \
  } while (0)
  FM_CHECK_ARG(!adam || (vec == 4 && grad && grad_depth));
  FM_CHECK_ARG(!taps || vec == 4);
  if (vec == 4) FM_FLOW_VEC(4);
#ifdef FM_FLOW_FORCE_VEC2
  else if (vec == 2) FM_FLOW_KIND(2, false);
#endif
  else FM_FLOW_KIND(1, false);
#undef FM_FLOW_VEC
#undef FM_FLOW_KIND
#undef FM_FLOW_LAUNCH
  FM_LAUNCH_STATUS();
}

int fm_flow_loss_fused(const float* depth, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd,
                       const float* flow_fwd, const float* flow_bwd, const float* mask_fwd, const float* mask_bwd,
                       const float* packed, const float* scale, int batch, int frames, int height, int width,
                       int mapping_kind, float delta, float aspect_x, float aspect_y, float* grad_depth, double* acc,
                       int items_per_thread, void* stream) {
  return flow_loss_launch(depth, k, kinv, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, packed, scale, batch, frames, height, width,
                          mapping_kind, delta, aspect_x, aspect_y, grad_depth, acc, items_per_thread, nullptr, nullptr, stream);
}

int fm_flow_loss_fused_views(const float* depth, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd,
                             const float* flow_fwd, const float* flow_bwd, const float* mask_fwd, const float* mask_bwd,
                             const float* packed, const float* scale, int batch, int frames, int height, int width,
                             int mapping_kind, float delta, float aspect_x, float aspect_y, float* grad_depth, double* acc,
                             int items_per_thread, const fm_layout* layouts, void* stream) {
  return flow_loss_launch(depth, k, kinv, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, packed, scale, batch, frames, height, width,
                          mapping_kind, delta, aspect_x, aspect_y, grad_depth, acc, items_per_thread, nullptr, layouts, stream);
}

int fm_flow_loss_fused_adam(float* depth, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd, const float* flow_fwd,
                            const float* flow_bwd, const float* mask_fwd, const float* mask_bwd, const float* packed, const float* scale,
                            int batch, int frames, int height, int width, int mapping_kind, float delta, float aspect_x, float aspect_y,
                            float* grad_depth, double* acc, int items_per_thread, float* exp_avg, float* exp_avg_sq, const uint8_t* touched,
                            long step, double lr, double beta1, double beta2, double eps, void* stream) {
  FM_CHECK_ARG(exp_avg && exp_avg_sq && touched && scale && grad_depth && step >= 1 && width % 4 == 0);
  FM_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0);
  auto aligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  FM_CHECK_ARG(aligned(exp_avg) && aligned(exp_avg_sq));
  const FlowAdam adam{exp_avg, exp_avg_sq, touched, adam_coefficients((double)step, lr, beta1, beta2, eps, 0.0)};
  return flow_loss_launch(depth, k, kinv, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, packed, scale, batch, frames, height, width,
                          mapping_kind, delta, aspect_x, aspect_y, grad_depth, acc, items_per_thread, &adam, nullptr, stream);
}

int fm_flow_loss_fused_taps(float* depth, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd, const float* flow_fwd,
                            const float* flow_bwd, const float* mask_fwd, const float* mask_bwd, const float* packed, const float* scale,
                            int batch, int frames, int height, int width, int mapping_kind, float delta, float aspect_x, float aspect_y,
                            float* grad_depth, double* acc, int items_per_thread, const fm_flow_taps* taps, float* exp_avg, float* exp_avg_sq,
                            const uint8_t* touched, long step, double lr, double beta1, double beta2, double eps, void* stream) {
  FM_CHECK_ARG(taps != nullptr);
  if (exp_avg == nullptr) {
    FM_CHECK_ARG(exp_avg_sq == nullptr && touched == nullptr);
    return flow_loss_launch(depth, k, kinv, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, packed, scale, batch, frames, height, width,
                            mapping_kind, delta, aspect_x, aspect_y, grad_depth, acc, items_per_thread, nullptr, nullptr, stream, taps);
  }
  FM_CHECK_ARG(exp_avg_sq && touched && scale && grad_depth && step >= 1 && width % 4 == 0);
  FM_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0);
  auto aligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  FM_CHECK_ARG(aligned(exp_avg) && aligned(exp_avg_sq));
  const FlowAdam adam{exp_avg, exp_avg_sq, touched, adam_coefficients((double)step, lr, beta1, beta2, eps, 0.0)};
  return flow_loss_launch(depth, k, kinv, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, packed, scale, batch, frames, height, width,
                          mapping_kind, delta, aspect_x, aspect_y, grad_depth, acc, items_per_thread, &adam, nullptr, stream, taps);
}

int fm_flow_loss_finalize(double* acc, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd,
                          const float* norm, int batch, int frames, float aspect_x, float aspect_y, float* loss, float* g_t_fwd,
                          float* g_t_bwd, float* g_k, void* stream) {
  FM_CHECK_ARG(acc && k && kinv && t_fwd && t_bwd && norm && loss && g_t_fwd && g_t_bwd && g_k);
  FlowFinalizeParams p{acc, k, kinv, t_fwd, t_bwd, norm, loss, g_t_fwd, g_t_bwd, g_k, batch, frames, aspect_x, aspect_y};
  hipLaunchKernelGGL(flow_finalize_kernel, dim3(1), dim3(kFinalizeThreads), 0, (hipStream_t)stream, p);
  FM_LAUNCH_STATUS();
}

static int valid_norm_launch(const float* mask_fwd, const float* mask_bwd, int batch, int frames_per_batch, long n, float weight, double* vsum,
                             float* norm, const fm_layout* layouts, void* stream) {
  FM_CHECK_ARG(mask_fwd && vsum && norm && batch >= 0 && frames_per_batch >= 0 && n >= 0 && (long)batch * frames_per_batch <= 65535);
  hipStream_t st = (hipStream_t)stream;
  long fs[2], bs[2];
  for (int i = 0; i < 2; ++i) {
    const bool given = layouts && (layouts[i].frame_stride != 0 || layouts[i].batch_stride != 0);
    fs[i] = given ? layouts[i].frame_stride : n;
    bs[i] = given ? layouts[i].batch_stride : n * frames_per_batch;
  }
  if (hipMemsetAsync(vsum, 0, sizeof(double), st) != hipSuccess) return FM_ERR_LAUNCH;
  if (n > 0 && batch * frames_per_batch > 0) {
    long blocks = (n + 256 * 16 - 1) / (256 * 16);
    const long cap = 4096 / ((long)batch * frames_per_batch) + 1;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(sum2_kernel, dim3((unsigned)blocks, (unsigned)(batch * frames_per_batch)), dim3(256), 0, st, mask_fwd, mask_bwd, n, frames_per_batch,
                       fs[0], bs[0], fs[1], bs[1], vsum);
  }
  hipLaunchKernelGGL(flow_norm_kernel, dim3(1), dim3(1), 0, st, vsum, weight, norm);
  FM_LAUNCH_STATUS();
}

int fm_flow_valid_norm(const float* mask_fwd, const float* mask_bwd, long count, float weight, double* vsum, float* norm,
                       void* stream) {
  FM_CHECK_ARG(mask_fwd && vsum && norm && count >= 0);
  // one dense run of `count` elements: cut into at most 1024 equal "frames" so that the launch has a second grid dimension
  long frames = 1;
  while (frames < 1024 && count % (frames * 2) == 0 && count / (frames * 2) >= 4096) frames *= 2;
  return valid_norm_launch(mask_fwd, mask_bwd, 1, (int)frames, count / frames, weight, vsum, norm, nullptr, stream);
}

int fm_flow_valid_norm_views(const float* mask_fwd, const float* mask_bwd, int batch, int pairs, long pixels, float weight, double* vsum,
                             float* norm, const fm_layout* layouts, void* stream) {
  return valid_norm_launch(mask_fwd, mask_bwd, batch, pairs, pixels, weight, vsum, norm, layouts, stream);
}

static int pack_launch(const float* flow_fwd, const float* flow_bwd, const float* mask_fwd, const float* mask_bwd, int batch,
                       int frames, int height, int width, float* packed, const fm_layout* layouts, void* stream) {
  FM_CHECK_ARG(flow_fwd && flow_bwd && mask_fwd && mask_bwd && packed);
  FM_CHECK_ARG(batch >= 1 && frames >= 2 && height >= 1 && width >= 1 && width % 4 == 0 && (long)batch * frames <= 65535);
  auto aligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  FM_CHECK_ARG(aligned(flow_fwd) && aligned(flow_bwd) && aligned(mask_fwd) && aligned(mask_bwd) && aligned(packed));
  const int n = height * width, quads = n / 4;
  PackLayouts lay;
  const long per_frame[4] = {2L * n, 2L * n, n, n};
  for (int i = 0; i < 4; ++i) {
    const bool given = layouts && (layouts[i].frame_stride != 0 || layouts[i].batch_stride != 0);
    lay.fs[i] = given ? layouts[i].frame_stride : per_frame[i];
    lay.bs[i] = given ? layouts[i].batch_stride : per_frame[i] * (frames - 1);
    FM_CHECK_ARG(lay.fs[i] >= per_frame[i] && lay.fs[i] % 4 == 0 && lay.bs[i] % 4 == 0);
  }
  int bx = (quads + 255) / 256;
  if (bx > 2048) bx = 2048;
  hipLaunchKernelGGL(pack_inputs_kernel, dim3((unsigned)bx, (unsigned)(batch * frames)), dim3(256), 0, (hipStream_t)stream, flow_fwd,
                     flow_bwd, mask_fwd, mask_bwd, frames, n, packed, lay);
  FM_LAUNCH_STATUS();
}

int fm_flow_pack_inputs(const float* flow_fwd, const float* flow_bwd, const float* mask_fwd, const float* mask_bwd, int batch,
                        int frames, int height, int width, float* packed, void* stream) {
  return pack_launch(flow_fwd, flow_bwd, mask_fwd, mask_bwd, batch, frames, height, width, packed, nullptr, stream);
}

int fm_flow_pack_inputs_views(const float* flow_fwd, const float* flow_bwd, const float* mask_fwd, const float* mask_bwd, int batch,
                              int frames, int height, int width, float* packed, const fm_layout* layouts, void* stream) {
  return pack_launch(flow_fwd, flow_bwd, mask_fwd, mask_bwd, batch, frames, height, width, packed, layouts, stream);
}

int fm_abi_version(void) { return FM_ABI_VERSION; }

int fm_scale_if_needed(float* x, long count, float* y, long count_y, const float* scalar, int* not_one, void* stream) {
  FM_CHECK_ARG(scalar && count >= 0 && count_y >= 0 && (x || count == 0) && (y || count_y == 0));
  if (count + count_y == 0 && !not_one) return FM_OK;
  if (count + count_y == 0) count_y = 0;
  long blocks = (count + count_y + 256 * 8 - 1) / (256 * 8);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(scale_if_needed_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, count, y, count_y, scalar, not_one);
  FM_LAUNCH_STATUS();
}

}  // extern "C"
