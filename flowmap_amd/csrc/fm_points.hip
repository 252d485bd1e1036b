// Function-level building blocks on EXPLICIT point sets — the reference's standalone
// call surface (visualiser, softmin intrinsics, exporters call these directly):
//   unproject               flowmap/model/projection.py:76-90
//   reproject_points        flowmap/model/projection.py:116-134 (+ project_camera_space :49-58)
//   grid_sample(bilinear, border, align_corners=False)  as used at projection.py:235-241,266-272
//   Mapping.forward         flowmap/loss/mapping/mapping.py:35-43 (+ huber / l1 / l2)
//   align_rigid             flowmap/model/procrustes.py:7-51 on explicit (p, q, w)
// All are bandwidth-bound streaming kernels: one thread per point, small per-group
// matrices in SGPRs, per-group gradient sums reduced wave -> LDS -> fp64 atomics.
#include "fm_device.h"
#include "fm_pose.h"

namespace fm {

// ------------------------------------------------------------------ unproject ------
// out[g,i,:] = (Kinv_g · [x,y,1]) · z[g,i];  xy is (G,N,2) or broadcast (N,2).
__global__ void __launch_bounds__(256) unproject_fwd_kernel(const float* xy, long xy_group_stride, const float* z,
                                                            const float* kinv, long n, float* out) {
  const int g = blockIdx.y;
  Mat3 ki;
  load_mat3(kinv + (size_t)g * 9, ki);
  const float2* c = reinterpret_cast<const float2*>(xy + (size_t)g * xy_group_stride);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float2 p = c[i];
    float ray[3];
    ray_dir(ki, p.x, p.y, ray);
    const float zz = z[(size_t)g * n + i];
    float* o = out + ((size_t)g * n + i) * 3;
    o[0] = ray[0] * zz;
    o[1] = ray[1] * zz;
    o[2] = ray[2] * zz;
  }
}

// g_z[g,i] = g_out · ray ;  kinv_acc[g] += Σ g_out ⊗ (z·[x,y,1])
__global__ void __launch_bounds__(256) unproject_bwd_kernel(const float* xy, long xy_group_stride, const float* z,
                                                            const float* kinv, const float* g_out, long n, float* g_z,
                                                            double* kinv_acc) {
  __shared__ double red[4 * 9];
  const int g = blockIdx.y;
  Mat3 ki;
  load_mat3(kinv + (size_t)g * 9, ki);
  const float2* c = reinterpret_cast<const float2*>(xy + (size_t)g * xy_group_stride);
  float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float2 p = c[i];
    float ray[3];
    ray_dir(ki, p.x, p.y, ray);
    const float zz = z[(size_t)g * n + i];
    const float* go = g_out + ((size_t)g * n + i) * 3;
    const float g0 = go[0], g1 = go[1], g2 = go[2];
    if (g_z) g_z[(size_t)g * n + i] = g0 * ray[0] + g1 * ray[1] + g2 * ray[2];
    const float zh[3] = {zz * p.x, zz * p.y, zz};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      acc[0 + d] += g0 * zh[d];
      acc[3 + d] += g1 * zh[d];
      acc[6 + d] += g2 * zh[d];
    }
  }
  if (kinv_acc) block_accumulate<9>(acc, red, kinv_acc + (size_t)g * 9);
}

// ------------------------------------------------------------------ reproject ------
// xy[g,i,:] = project_camera_space((T_g · [xyz;1])[:3], K_g)
__global__ void __launch_bounds__(256) reproject_fwd_kernel(const float* xyz, const float* t, const float* k, long n, float* xy) {
  const int g = blockIdx.y;
  Pose tr;
  Mat3 kk;
  load_pose44(t + (size_t)g * 16, tr);
  load_mat3(k + (size_t)g * 9, kk);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float* p = xyz + ((size_t)g * n + i) * 3;
    const float x[3] = {p[0], p[1], p[2]};
    float xc[3];
    apply_pose(tr, x, xc);
    const Projected pr = project_point(xc, kk);
    reinterpret_cast<float2*>(xy)[(size_t)g * n + i] = make_float2(pr.u, pr.v);
  }
}

// acc per group: [0..2] dL/dt, [3..11] dL/dR, [12..17] dL/dK rows 0,1
__global__ void __launch_bounds__(256) reproject_bwd_kernel(const float* xyz, const float* t, const float* k, const float* g_xy,
                                                            long n, float* g_xyz, double* acc_out) {
  __shared__ double red[4 * 18];
  const int g = blockIdx.y;
  Pose tr;
  Mat3 kk;
  load_pose44(t + (size_t)g * 16, tr);
  load_mat3(k + (size_t)g * 9, kk);
  float acc[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) acc[i] = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float* p = xyz + ((size_t)g * n + i) * 3;
    const float x[3] = {p[0], p[1], p[2]};
    float xc[3];
    apply_pose(tr, x, xc);
    const Projected pr = project_point(xc, kk);
    const float2 go = reinterpret_cast<const float2*>(g_xy)[(size_t)g * n + i];
    float gk[6] = {0, 0, 0, 0, 0, 0}, gxc[3];
    project_point_bwd(pr, kk, go.x, go.y, gk, gxc);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      acc[a] += gxc[a];
#pragma unroll
      for (int d = 0; d < 3; ++d) acc[3 + a * 3 + d] += gxc[a] * x[d];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[12 + a] += gk[a];
    if (g_xyz) {
      float gx[3];
      apply_rot_t(tr, gxc, gx);
      float* o = g_xyz + ((size_t)g * n + i) * 3;
      o[0] = gx[0];
      o[1] = gx[1];
      o[2] = gx[2];
    }
  }
  block_accumulate<18>(acc, red, acc_out + (size_t)g * 18);
}

// fp64 group sums -> g_t (G,4,4) and g_k (G,3,3)
__global__ void reproject_finalize_kernel(const double* acc, int groups, float* g_t, float* g_k) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= groups) return;
  const double* a = acc + (size_t)g * 18;
  if (g_t) {
    float* o = g_t + (size_t)g * 16;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) o[r * 4 + c] = (float)a[3 + r * 3 + c];
      o[r * 4 + 3] = (float)a[r];
    }
    o[12] = o[13] = o[14] = o[15] = 0.f;
  }
  if (g_k) {
    float* o = g_k + (size_t)g * 9;
    for (int i = 0; i < 6; ++i) o[i] = (float)a[12 + i];
    o[6] = o[7] = o[8] = 0.f;
  }
}

// ------------------------------------------------------------ bilinear sampling ------
// img (G,H,W,C) channels-last, xy (G,P,2) normalised (0,1) -> out (G,P,C)
__global__ void __launch_bounds__(256) bilinear_fwd_kernel(const float* img, const float* xy, int h, int w, int c, long p,
                                                           float* out) {
  const int g = blockIdx.y;
  const float* im = img + (size_t)g * h * w * c;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < p; i += (long)gridDim.x * blockDim.x) {
    const float2 q = reinterpret_cast<const float2*>(xy)[(size_t)g * p + i];
    const Taps t = bilinear_taps(q.x, q.y, h, w);
    float* o = out + ((size_t)g * p + i) * c;
    for (int ch = 0; ch < c; ++ch) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (t.in[k]) s += im[((size_t)tap_row(t, k) * w + tap_col(t, k)) * c + ch] * t.w[k];
      o[ch] = s;
    }
  }
}

// g_img (G,H,W,C) must be zero-initialised by the caller (atomic scatter).
__global__ void __launch_bounds__(256) bilinear_bwd_kernel(const float* g_out, const float* xy, int h, int w, int c, long p,
                                                           float* g_img) {
  const int g = blockIdx.y;
  float* gi = g_img + (size_t)g * h * w * c;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < p; i += (long)gridDim.x * blockDim.x) {
    const float2 q = reinterpret_cast<const float2*>(xy)[(size_t)g * p + i];
    const Taps t = bilinear_taps(q.x, q.y, h, w);
    const float* go = g_out + ((size_t)g * p + i) * c;
    for (int ch = 0; ch < c; ++ch) {
      const float gv = go[ch];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (t.in[k]) atomicAdd(gi + ((size_t)tap_row(t, k) * w + tap_col(t, k)) * c + ch, gv * t.w[k]);
    }
  }
}

// ---------------------------------------------------------------------- mapping ------
// out[i] = ρ(fix_aspect(a_i) − fix_aspect(b_i));  a, b (n,2) with b optionally broadcast
// handled by the host (it expands).  bwd: g_a = g_out · dρ/dr ⊙ aspect ; g_b = −g_a.
__global__ void __launch_bounds__(256) mapping_fwd_kernel(const float* a, const float* b, long n, int kind, float delta, float ax,
                                                          float ay, float* out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float2 pa = reinterpret_cast<const float2*>(a)[i];
    const float2 pb = reinterpret_cast<const float2*>(b)[i];
    float dx, dy;
    out[i] = robust_map(kind, delta, aspect_diff(pa.x, pb.x, ax), aspect_diff(pa.y, pb.y, ay), dx, dy);
  }
}

__global__ void __launch_bounds__(256) mapping_bwd_kernel(const float* a, const float* b, const float* g_out, long n, int kind,
                                                          float delta, float ax, float ay, float* g_a, float* g_b) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float2 pa = reinterpret_cast<const float2*>(a)[i];
    const float2 pb = reinterpret_cast<const float2*>(b)[i];
    float dx, dy;
    robust_map(kind, delta, aspect_diff(pa.x, pb.x, ax), aspect_diff(pa.y, pb.y, ay), dx, dy);
    const float go = g_out[i];
    const float gx = go * dx * ax, gy = go * dy * ay;
    if (g_a) reinterpret_cast<float2*>(g_a)[i] = make_float2(gx, gy);
    if (g_b) reinterpret_cast<float2*>(g_b)[i] = make_float2(-gx, -gy);
  }
}

// ------------------------------------------------- align_rigid on explicit points ------
// p, q (G,P,3); w (G,P).  Same two-pass statistics as fm_procrustes.hip.
template <int PASS>
__global__ void __launch_bounds__(256) rigid_stats_kernel(const float* p, const float* q, const float* w, long points,
                                                          double* stats) {
  __shared__ double red[4 * 9];
  const int g = blockIdx.y;
  double* st = stats + (size_t)g * kStatStride;
  float pbar[3] = {0, 0, 0}, qbar[3] = {0, 0, 0};
  if (PASS == 2) {
    const double inv = 1.0 / (st[0] + 1e-8);
    for (int a = 0; a < 3; ++a) {
      pbar[a] = (float)(st[1 + a] * inv);
      qbar[a] = (float)(st[4 + a] * inv);
    }
  }
  constexpr int NV = PASS == 1 ? 7 : 9;
  float acc[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = 0.f;
  for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < points; j += (long)gridDim.x * blockDim.x) {
    const float* pp = p + ((size_t)g * points + j) * 3;
    const float* qq = q + ((size_t)g * points + j) * 3;
    const float ww = w[(size_t)g * points + j];
    if (PASS == 1) {
      acc[0] += ww;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        acc[1 + a] += ww * pp[a];
        acc[4 + a] += ww * qq[a];
      }
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float wq = ww * (qq[a] - qbar[a]);
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[a * 3 + d] += wq * (pp[d] - pbar[d]);
      }
    }
  }
  block_accumulate<NV>(acc, red, st + (PASS == 1 ? 0 : 7));
}

__global__ void __launch_bounds__(256) rigid_bwd_kernel(const float* p, const float* q, const float* w, long points,
                                                        const double* aux, const double* pair_grad, float* g_p, float* g_q,
                                                        float* g_w) {
  const int g = blockIdx.y;
  const double* pg = pair_grad + (size_t)g * kPairGradStride;
  const double* ax = aux + (size_t)g * kAuxStride;
  PairGrad gr;
  for (int k = 0; k < 9; ++k) gr.gM[k] = (float)pg[k];
  for (int a = 0; a < 3; ++a) {
    gr.gqbar[a] = (float)pg[9 + a];
    gr.gpbar[a] = (float)pg[12 + a];
    gr.pbar[a] = (float)ax[21 + a];
    gr.qbar[a] = (float)ax[24 + a];
  }
  gr.dbar = (float)pg[15];
  gr.inv_wsum = (float)pg[16];
  for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < points; j += (long)gridDim.x * blockDim.x) {
    const size_t o = (size_t)g * points + j;
    Corr c;
    for (int a = 0; a < 3; ++a) {
      c.p[a] = p[o * 3 + a];
      c.q[a] = q[o * 3 + a];
    }
    c.w = w[o];
    float gq[3], gp[3], gw;
    corr_backward(c, gr, gq, gp, gw);
    if (g_p)
      for (int a = 0; a < 3; ++a) g_p[o * 3 + a] = gp[a];
    if (g_q)
      for (int a = 0; a < 3; ++a) g_q[o * 3 + a] = gq[a];
    if (g_w) g_w[o] = gw;
  }
}

}  // namespace fm

using namespace fm;

// World-space point cloud of export_to_colmap (flowmap/export/colmap.py:86-101): per frame
//   xyz = unproject(xy, depth, K);  world = (E · [xyz; 1])[:3];  rgb (3,H,W) -> (H·W,3)
// one streaming pass: 4 + 12 B in, 12 + 12 B out per pixel, frames concatenated.
__global__ void __launch_bounds__(256) world_points_kernel(const float* depth, const float* kinv, const float* ext, const float* colors,
                                                           int h, int w, float* out_xyz, float* out_rgb) {
  const int fr = blockIdx.y;
  const size_t n = (size_t)h * w;
  Mat3 ki;
  Pose e;
  load_mat3(kinv + (size_t)fr * 9, ki);
  load_pose44(ext + (size_t)fr * 16, e);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / w), col = (int)(i - (size_t)row * w);
    float ray[3], xyz[3], xw[3];
    ray_dir(ki, pixel_center(col, w), pixel_center(row, h), ray);
    const float z = depth[(size_t)fr * n + i];
    xyz[0] = ray[0] * z; xyz[1] = ray[1] * z; xyz[2] = ray[2] * z;
    apply_pose(e, xyz, xw);
    float* o = out_xyz + ((size_t)fr * n + i) * 3;
    o[0] = xw[0]; o[1] = xw[1]; o[2] = xw[2];
    if (colors) {
      const float* c = colors + (size_t)fr * 3 * n + i;
      float* oc = out_rgb + ((size_t)fr * n + i) * 3;
      oc[0] = c[0]; oc[1] = c[n]; oc[2] = c[2 * n];
    }
  }
}

// ---------------------------------------------------------------------------------
// IntrinsicsSoftmin candidate score (intrinsics_softmin.py:105-121), straight from the images:
// the sampled pixels' depth / weight / flow are gathered here (no gathered copies, no (60, P, 3)
// point clouds).  depth (B,2,H,W): frame 1 is the later frame; weights, flow: pair 0.
// ---------------------------------------------------------------------------------
struct SoftminArgs {
  const float* depth;      // (B,2,H,W)
  const float* weights;    // (B,H,W) weights or logits (weight_sens != 0)
  const float* bwd_flow;   // (B,H,W,2)
  const int64_t* indices;  // (P)
  const float* k;          // (N,3,3)
  const float* kinv;       // (N,3,3)
  const float* rel;        // (B*N,4,4) later -> earlier
  long points;
  int batch, candidates, height, width;
  float weight_sens;
};

struct SoftminPoint {
  float u, v, z, w, wraw, gx, gy;
  int idx;
};

__device__ __forceinline__ SoftminPoint softmin_point(const SoftminArgs& a, int b, long j) {
  SoftminPoint s;
  s.idx = (int)a.indices[j];
  const size_t n = (size_t)a.height * a.width;
  const int row = s.idx / a.width, col = s.idx - row * a.width;
  s.u = pixel_center(col, a.width);
  s.v = pixel_center(row, a.height);
  s.z = a.depth[((size_t)b * 2 + 1) * n + s.idx];
  s.wraw = a.weights[(size_t)b * n + s.idx];
  s.w = a.weight_sens != 0.f ? 1.0f / (1.0f + expf(-a.weight_sens * s.wraw)) : s.wraw;
  s.gx = a.bwd_flow[((size_t)b * n + s.idx) * 2];
  s.gy = a.bwd_flow[((size_t)b * n + s.idx) * 2 + 1];
  return s;
}

// grid: (point chunks, B*N).  err[b*N + n] += Σ_j e(n, j)   (fp64, zeroed by the entry point)
__global__ void __launch_bounds__(256) softmin_score_fwd_kernel(SoftminArgs a, double* err) {
  __shared__ double red[4];
  const int bn = blockIdx.y, b = bn / a.candidates, n = bn % a.candidates;
  Mat3 k, kinv;
  Pose t;
  load_mat3(a.k + (size_t)n * 9, k);
  load_mat3(a.kinv + (size_t)n * 9, kinv);
  load_pose44(a.rel + (size_t)bn * 16, t);
  float e[1] = {0.f};
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < a.points) {
    const SoftminPoint s = softmin_point(a, b, j);
    SoftminTerm o;
    e[0] = softmin_term(k, kinv, t, s.u, s.v, s.z, s.gx, s.gy, s.w, o);
  }
  block_accumulate<1>(e, red, err + bn);
}

// grid: (point chunks, B, candidate groups).  One thread per sampled pixel walks a group of kSoftminGroup candidates:
// dL/dz and dL/dw sum in registers and are ADDED once per group (float atomics into the caller's zero-filled
// images: 10 adds per pixel for 60 candidates), dL/dT per candidate is block-reduced.  (One group = all candidates
// was 32 blocks each doing 60 block reductions in a row: 100 us; the groups run side by side.)
constexpr int kSoftminGroup = 6;
__global__ void __launch_bounds__(256) softmin_score_bwd_kernel(SoftminArgs a, const float* g_err, float* g_depth, float* g_weights,
                                                                double* g_rel) {
  __shared__ double red[4 * 12];
  const int b = blockIdx.y;
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = j < a.points;
  SoftminPoint s = {};
  if (active) s = softmin_point(a, b, j);
  float gz = 0.f, gw = 0.f;
  const int n0 = blockIdx.z * kSoftminGroup, n1 = min(n0 + kSoftminGroup, a.candidates);
  for (int n = n0; n < n1; ++n) {
    const int bn = b * a.candidates + n;
    Mat3 k, kinv;
    Pose t;
    load_mat3(a.k + (size_t)n * 9, k);
    load_mat3(a.kinv + (size_t)n * 9, kinv);
    load_pose44(a.rel + (size_t)bn * 16, t);
    float gt[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) gt[i] = 0.f;
    if (active) {
      SoftminTerm o;
      softmin_term(k, kinv, t, s.u, s.v, s.z, s.gx, s.gy, s.w, o);
      softmin_term_bwd(k, t, o, s.w, g_err[bn], gz, gw, gt);
    }
    block_accumulate<12>(gt, red, g_rel + (size_t)bn * 12);
  }
  if (!active) return;
  const size_t npx = (size_t)a.height * a.width;
  if (g_depth) atomicAdd(g_depth + ((size_t)b * 2 + 1) * npx + s.idx, gz);
  if (g_weights) atomicAdd(g_weights + (size_t)b * npx + s.idx, a.weight_sens != 0.f ? gw * a.weight_sens * s.w * (1.f - s.w) : gw);
}

__global__ void softmin_rel_grad_kernel(const double* acc, int count, float* g_rel) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  for (int e = 0; e < 12; ++e) g_rel[(size_t)i * 16 + e] = (float)acc[(size_t)i * 12 + e];
  for (int e = 12; e < 16; ++e) g_rel[(size_t)i * 16 + e] = 0.f;
}

// The tail of IntrinsicsSoftmin.forward (intrinsics_softmin.py:123-141), one wave per batch entry:
//   soft = softmin((err − min err)·10) over the candidates   (fp32, like the reference's F.softmin)
//   K    = Σ_n soft[n]·candidate_k[n], written for every frame, with K^-1 beside it.
__global__ void __launch_bounds__(kWave) softmin_blend_fwd_kernel(const double* err, const float* cand_k, int candidates, int frames,
                                                                  float* soft, float* k, float* kinv) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const double* e = err + (size_t)b * candidates;
  float lo = 3.0e38f;
  for (int n = lane; n < candidates; n += kWave) lo = fminf(lo, (float)e[n]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) lo = fminf(lo, __shfl_xor(lo, off, kWave));
  float total = 0.f;
  for (int n = lane; n < candidates; n += kWave) total += expf(-(((float)e[n] - lo) * 10.f));
  total = wave_sum(total);
  float m[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) m[i] = 0.f;
  for (int n = lane; n < candidates; n += kWave) {
    const float sn = expf(-(((float)e[n] - lo) * 10.f)) / total;
    soft[(size_t)b * candidates + n] = sn;
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] += cand_k[(size_t)n * 9 + i] * sn;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) m[i] = wave_sum(m[i]);
  float inv[9];
  inv3(m, inv);
  for (int f = lane; f < frames; f += kWave) {
    float* ko = k + ((size_t)b * frames + f) * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) ko[i] = m[i];
    if (kinv) {
      float* io = kinv + ((size_t)b * frames + f) * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) io[i] = inv[i];
    }
  }
}

// Backward: G = Σ_f dL/dK[f];  gs[n] = <candidate_k[n], G>;  dL/derr[n] = −10·soft[n]·(gs[n] − Σ soft·gs).
// (Autograd also routes Σ_n of that — analytically 0, rounding noise in fp32 — to the arg-min entry.)
__global__ void __launch_bounds__(kWave) softmin_blend_bwd_kernel(const float* g_k, const float* soft, const float* cand_k, int candidates,
                                                                  int frames, float* g_err) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float g[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) g[i] = 0.f;
  for (int f = lane; f < frames; f += kWave)
#pragma unroll
    for (int i = 0; i < 9; ++i) g[i] += g_k[((size_t)b * frames + f) * 9 + i];
#pragma unroll
  for (int i = 0; i < 9; ++i) g[i] = wave_sum(g[i]);
  const float* sb = soft + (size_t)b * candidates;
  float dot = 0.f;
  for (int n = lane; n < candidates; n += kWave) {
    float gs = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) gs += cand_k[(size_t)n * 9 + i] * g[i];
    dot += sb[n] * gs;
  }
  dot = wave_sum(dot);
  for (int n = lane; n < candidates; n += kWave) {
    float gs = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) gs += cand_k[(size_t)n * 9 + i] * g[i];
    g_err[(size_t)b * candidates + n] = -10.f * sb[n] * (gs - dot);
  }
}

__global__ void __launch_bounds__(256) random_subset_kernel(unsigned long long seed, long n, long count, int64_t* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = (int64_t)permuted_index((uint64_t)i, (uint64_t)n, seed);
}

// hipGraph-capturable variant: the seed lives in device memory and is advanced by a follow-up
// launch, so every replay of a captured step draws a fresh subset.
__global__ void __launch_bounds__(256) random_subset_state_kernel(const unsigned long long* state, long n, long count, int64_t* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = (int64_t)permuted_index((uint64_t)i, (uint64_t)n, state[0]);
}
__global__ void random_state_advance_kernel(unsigned long long* state) {
  unsigned long long z = state[0] + 0x9e3779b97f4a7c15ULL;  // splitmix64
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  state[0] = z ^ (z >> 31);
}

static inline unsigned blocks_for(long n, int per_thread = 4, unsigned cap = 4096) {
  long b = (n + 256L * per_thread - 1) / (256L * per_thread);
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

extern "C" {

int fm_unproject_fwd(const float* xy, long xy_group_stride, const float* z, const float* kinv, int groups, long points,
                     float* out, void* stream) {
  FM_CHECK_ARG(xy && z && kinv && out && groups >= 1 && groups <= 65535 && points >= 1);
  hipLaunchKernelGGL(unproject_fwd_kernel, dim3(blocks_for(points), groups), dim3(256), 0, (hipStream_t)stream, xy,
                     xy_group_stride, z, kinv, points, out);
  FM_LAUNCH_STATUS();
}

int fm_unproject_bwd(const float* xy, long xy_group_stride, const float* z, const float* kinv, const float* g_out, int groups,
                     long points, float* g_z, double* kinv_acc, void* stream) {
  FM_CHECK_ARG(xy && z && kinv && g_out && groups >= 1 && groups <= 65535 && points >= 1);
  hipStream_t st = (hipStream_t)stream;
  if (kinv_acc && hipMemsetAsync(kinv_acc, 0, sizeof(double) * (size_t)groups * 9, st) != hipSuccess) return FM_ERR_LAUNCH;
  hipLaunchKernelGGL(unproject_bwd_kernel, dim3(blocks_for(points), groups), dim3(256), 0, st, xy, xy_group_stride, z, kinv,
                     g_out, points, g_z, kinv_acc);
  FM_LAUNCH_STATUS();
}

int fm_world_points(const float* depth, const float* kinv, const float* ext, const float* colors, int frames, int height, int width,
                    float* out_xyz, float* out_rgb, void* stream) {
  FM_CHECK_ARG(depth && kinv && ext && out_xyz && frames >= 1 && frames <= 65535 && height >= 1 && width >= 1);
  FM_CHECK_ARG(!colors || out_rgb);
  hipLaunchKernelGGL(world_points_kernel, dim3(blocks_for((long)height * width), frames), dim3(256), 0, (hipStream_t)stream, depth,
                     kinv, ext, colors, height, width, out_xyz, out_rgb);
  FM_LAUNCH_STATUS();
}

int fm_softmin_score_fwd(const float* depth, const float* weights, float weight_sensitivity, const float* bwd_flow,
                         const int64_t* indices, long points, const float* k, const float* kinv, const float* rel, int batch,
                         int candidates, int height, int width, double* err, void* stream) {
  FM_CHECK_ARG(depth && weights && bwd_flow && indices && k && kinv && rel && err);
  FM_CHECK_ARG(points >= 1 && batch >= 1 && candidates >= 1 && (long)batch * candidates <= 65535 && height >= 1 && width >= 1);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(err, 0, sizeof(double) * (size_t)batch * candidates, st) != hipSuccess) return FM_ERR_LAUNCH;
  const SoftminArgs a{depth, weights, bwd_flow, indices, k, kinv, rel, points, batch, candidates, height, width, weight_sensitivity};
  hipLaunchKernelGGL(softmin_score_fwd_kernel, dim3((unsigned)((points + 255) / 256), batch * candidates), dim3(256), 0, st, a, err);
  FM_LAUNCH_STATUS();
}

int fm_softmin_score_bwd(const float* depth, const float* weights, float weight_sensitivity, const float* bwd_flow,
                         const int64_t* indices, long points, const float* k, const float* kinv, const float* rel, int batch,
                         int candidates, int height, int width, const float* g_err, float* g_depth, float* g_weights,
                         double* g_rel_acc, float* g_rel, void* stream) {
  FM_CHECK_ARG(depth && weights && bwd_flow && indices && k && kinv && rel && g_err && g_rel_acc && g_rel);
  FM_CHECK_ARG(points >= 1 && batch >= 1 && candidates >= 1 && (long)batch * candidates <= 65535 && height >= 1 && width >= 1);
  hipStream_t st = (hipStream_t)stream;
  const int bn = batch * candidates;
  if (hipMemsetAsync(g_rel_acc, 0, sizeof(double) * (size_t)bn * 12, st) != hipSuccess) return FM_ERR_LAUNCH;
  const SoftminArgs a{depth, weights, bwd_flow, indices, k, kinv, rel, points, batch, candidates, height, width, weight_sensitivity};
  hipLaunchKernelGGL(softmin_score_bwd_kernel, dim3((unsigned)((points + 255) / 256), batch, (candidates + kSoftminGroup - 1) / kSoftminGroup),
                     dim3(256), 0, st, a, g_err, g_depth, g_weights, g_rel_acc);
  hipLaunchKernelGGL(softmin_rel_grad_kernel, dim3((bn + 63) / 64), dim3(64), 0, st, g_rel_acc, bn, g_rel);
  FM_LAUNCH_STATUS();
}

int fm_softmin_blend_fwd(const double* err, const float* candidate_k, int batch, int candidates, int frames, float* soft, float* k,
                         float* kinv, void* stream) {
  FM_CHECK_ARG(err && candidate_k && soft && k && batch >= 1 && candidates >= 1 && frames >= 1);
  hipLaunchKernelGGL(softmin_blend_fwd_kernel, dim3(batch), dim3(kWave), 0, (hipStream_t)stream, err, candidate_k, candidates, frames, soft,
                     k, kinv);
  FM_LAUNCH_STATUS();
}

int fm_softmin_blend_bwd(const float* grad_k, const float* soft, const float* candidate_k, int batch, int candidates, int frames,
                         float* grad_err, void* stream) {
  FM_CHECK_ARG(grad_k && soft && candidate_k && grad_err && batch >= 1 && candidates >= 1 && frames >= 1);
  hipLaunchKernelGGL(softmin_blend_bwd_kernel, dim3(batch), dim3(kWave), 0, (hipStream_t)stream, grad_k, soft, candidate_k, candidates,
                     frames, grad_err);
  FM_LAUNCH_STATUS();
}

int fm_random_subset(unsigned long long seed, long n, long count, int64_t* out, void* stream) {
  FM_CHECK_ARG(out && n >= 1 && count >= 1 && count <= n && n < (1L << 40));
  hipLaunchKernelGGL(random_subset_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, n, count, out);
  FM_LAUNCH_STATUS();
}

int fm_random_subset_stateful(unsigned long long* state, long n, long count, int64_t* out, void* stream) {
  FM_CHECK_ARG(state && out && n >= 1 && count >= 1 && count <= n && n < (1L << 40));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(random_subset_state_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, state, n, count, out);
  hipLaunchKernelGGL(random_state_advance_kernel, dim3(1), dim3(1), 0, st, state);
  FM_LAUNCH_STATUS();
}

int fm_reproject_fwd(const float* xyz, const float* t, const float* k, int groups, long points, float* xy, void* stream) {
  FM_CHECK_ARG(xyz && t && k && xy && groups >= 1 && groups <= 65535 && points >= 1);
  hipLaunchKernelGGL(reproject_fwd_kernel, dim3(blocks_for(points), groups), dim3(256), 0, (hipStream_t)stream, xyz, t, k, points, xy);
  FM_LAUNCH_STATUS();
}

int fm_reproject_bwd(const float* xyz, const float* t, const float* k, const float* g_xy, int groups, long points, float* g_xyz,
                     float* g_t, float* g_k, double* acc, void* stream) {
  FM_CHECK_ARG(xyz && t && k && g_xy && acc && groups >= 1 && groups <= 65535 && points >= 1);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(acc, 0, sizeof(double) * (size_t)groups * 18, st) != hipSuccess) return FM_ERR_LAUNCH;
  hipLaunchKernelGGL(reproject_bwd_kernel, dim3(blocks_for(points), groups), dim3(256), 0, st, xyz, t, k, g_xy, points, g_xyz, acc);
  hipLaunchKernelGGL(reproject_finalize_kernel, dim3((groups + 63) / 64), dim3(64), 0, st, acc, groups, g_t, g_k);
  FM_LAUNCH_STATUS();
}

int fm_bilinear_sample_fwd(const float* img, const float* xy, int groups, int height, int width, int channels, long points,
                           float* out, void* stream) {
  FM_CHECK_ARG(img && xy && out && groups >= 1 && groups <= 65535 && points >= 1 && channels >= 1);
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(blocks_for(points, 1), groups), dim3(256), 0, (hipStream_t)stream, img, xy, height,
                     width, channels, points, out);
  FM_LAUNCH_STATUS();
}

int fm_bilinear_sample_bwd(const float* g_out, const float* xy, int groups, int height, int width, int channels, long points,
                           float* g_img, void* stream) {
  FM_CHECK_ARG(g_out && xy && g_img && groups >= 1 && groups <= 65535 && points >= 1 && channels >= 1);
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(blocks_for(points, 1), groups), dim3(256), 0, (hipStream_t)stream, g_out, xy, height,
                     width, channels, points, g_img);
  FM_LAUNCH_STATUS();
}

int fm_mapping_fwd(const float* a, const float* b, long count, int kind, float delta, float aspect_x, float aspect_y, float* out,
                   void* stream) {
  FM_CHECK_ARG(a && b && out && count >= 0 && kind >= 0 && kind <= 2);
  if (count == 0) return FM_OK;
  hipLaunchKernelGGL(mapping_fwd_kernel, dim3(blocks_for(count)), dim3(256), 0, (hipStream_t)stream, a, b, count, kind, delta,
                     aspect_x, aspect_y, out);
  FM_LAUNCH_STATUS();
}

int fm_mapping_bwd(const float* a, const float* b, const float* g_out, long count, int kind, float delta, float aspect_x,
                   float aspect_y, float* g_a, float* g_b, void* stream) {
  FM_CHECK_ARG(a && b && g_out && count >= 0 && kind >= 0 && kind <= 2);
  if (count == 0) return FM_OK;
  hipLaunchKernelGGL(mapping_bwd_kernel, dim3(blocks_for(count)), dim3(256), 0, (hipStream_t)stream, a, b, g_out, count, kind,
                     delta, aspect_x, aspect_y, g_a, g_b);
  FM_LAUNCH_STATUS();
}

int fm_align_rigid_stats(const float* p, const float* q, const float* w, int groups, long points, double* stats, void* stream) {
  FM_CHECK_ARG(p && q && w && stats && groups >= 1 && groups <= 65535 && points >= 1);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(stats, 0, sizeof(double) * (size_t)groups * kStatStride, st) != hipSuccess) return FM_ERR_LAUNCH;
  dim3 grid(blocks_for(points, 4, 1024), groups);
  hipLaunchKernelGGL((rigid_stats_kernel<1>), grid, dim3(256), 0, st, p, q, w, points, stats);
  hipLaunchKernelGGL((rigid_stats_kernel<2>), grid, dim3(256), 0, st, p, q, w, points, stats);
  FM_LAUNCH_STATUS();
}

int fm_align_rigid_bwd(const float* p, const float* q, const float* w, int groups, long points, const double* aux,
                       const double* pair_grad, float* g_p, float* g_q, float* g_w, void* stream) {
  FM_CHECK_ARG(p && q && w && aux && pair_grad && groups >= 1 && groups <= 65535 && points >= 1);
  hipLaunchKernelGGL(rigid_bwd_kernel, dim3(blocks_for(points, 4, 1024), groups), dim3(256), 0, (hipStream_t)stream, p, q, w,
                     points, aux, pair_grad, g_p, g_q, g_w);
  FM_LAUNCH_STATUS();
}

}  // extern "C"
