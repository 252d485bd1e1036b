// Fused point-tracking loss.
//
// Replaces LossTracking.compute_unweighted_loss (flowmap/loss/loss_tracking.py:28-61)
// and compute_track_flow (flowmap/model/projection.py:255-298) for ALL segments in a
// handful of launches, without materialising the (f, f, P, 2) all-pairs tensors the
// reference allocates per segment (16.5 MB each, ~30 segments per step).
//
// Per segment (frames s .. s+f-1, P tracks), source frame fs, target frame ft, point p:
//   xyz   = bilinear(surfaces[s+fs], track_xy[fs,p])      (border, align_corners=False)
//   X'    = inv(E_ft) · E_fs · [xyz; 1]                     (projection.py:288)
//   uv    = project_camera_space(X', K_ft)
//   vis   = vis[fs,p] ∧ vis[ft,p] ∧ xy[fs,p] ∈ [0,1)² ∧ uv ∈ [0,1)²   (projection.py:290-296)
//   L     = weight · Σ ρ(uv, xy[ft,p]) · vis / max(Σ vis, 1)          (loss_tracking.py:55-61)
// surfaces are recomputed from depth and K⁻¹ (never stored).
//
// Launch plan (VALU/latency-bound, inputs ≈ 13 MB -> L2 resident):
//   track_points   one thread per (frame-in-segment, point): sample xyz, lift to world
//                  X_w = E_fs·xyz, fold visibility ∧ source-in-frame into one byte.
//   track_fwd      block per (segment, TARGET frame, point chunk), loop over sources:
//                  loss sum, visible count, Σ gX' ⊗ [X_w;1] (-> dL/d inv(E_ft)) and the
//                  K_ft gradient, all UNSCALED (count is not known yet).
//   finalize_fwd   loss, count, scale = weight / max(count, 1).
//   track_bwd      block per (segment, SOURCE frame, point chunk), loop over targets:
//                  gX_w = Σ_ft R_invᵀ gX' in registers, then one scatter per point into
//                  dL/ddepth (4 taps, atomics) + per-frame sums for dL/dE_fs, dL/dK⁻¹.
//   finalize_bwd   small-matrix chain rules -> dL/dE (F,4,4), dL/dK (F,3,3).
// The forward chain is evaluated twice (once per pass) instead of keeping per-(fs,ft,p)
// state or doing a 20-value block reduction per (fs,ft) pair.
#include "fm_device.h"
#include "fm_pose.h"

namespace fm {

constexpr int kTrackAccStride = 20;   // per frame: [0..11] Σ gX'⊗[X_w;1], [12..17] dK rows 0,1
constexpr int kTrackAcc2Stride = 24;  // per frame: [0..11] Σ gX_w⊗[xyz;1], [12..20] dKinv

struct TrackGeom {
  const float* xy;        // (total, 2) packed track positions
  const uint8_t* vis;     // (total)
  const int32_t* seg;     // (S, 4): start_frame, f, p, offset (in points)
  const int32_t* blocks;  // (NB, 2): segment, local frame
  int height, width;
};

// ---------------------------------------------------------------- track_points ------
__global__ void __launch_bounds__(256) track_points_kernel(TrackGeom g, const float* depth, const float* kinv, const float* ext,
                                                           float* ws, uint8_t* flag) {
  const int sg = g.blocks[blockIdx.x * 2], fl = g.blocks[blockIdx.x * 2 + 1];
  const int start = g.seg[sg * 4], p_count = g.seg[sg * 4 + 2], off = g.seg[sg * 4 + 3];
  const int p = blockIdx.y * blockDim.x + threadIdx.x;
  if (p >= p_count) return;
  const int frame = start + fl;
  const size_t idx = (size_t)off + (size_t)fl * p_count + p;
  const float2 q = reinterpret_cast<const float2*>(g.xy)[idx];
  Mat3 ki;
  Pose e;
  load_mat3(kinv + (size_t)frame * 9, ki);
  load_pose44(ext + (size_t)frame * 16, e);
  const Taps t = bilinear_taps(q.x, q.y, g.height, g.width);
  const float* d = depth + (size_t)frame * g.height * g.width;
  float xyz[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!t.in[k]) continue;
    const int tc = tap_col(t, k), tr = tap_row(t, k);
    float ray[3];
    ray_dir(ki, pixel_center(tc, g.width), pixel_center(tr, g.height), ray);
    const float z = d[tr * g.width + tc];
    xyz[0] += (ray[0] * z) * t.w[k];
    xyz[1] += (ray[1] * z) * t.w[k];
    xyz[2] += (ray[2] * z) * t.w[k];
  }
  float xw[3];
  apply_pose(e, xyz, xw);
  float* o = ws + idx * 6;
  o[0] = xyz[0]; o[1] = xyz[1]; o[2] = xyz[2];
  o[3] = xw[0];  o[4] = xw[1];  o[5] = xw[2];
  const bool inside = q.x >= 0.f && q.y >= 0.f && q.x < 1.f && q.y < 1.f;
  flag[idx] = (g.vis[idx] != 0 && inside) ? 1 : 0;
}

// One (source, target, point) residual.  Returns false when not visible.
struct TrackEval {
  Projected pr;
  float drx, dry, rho;
};

__device__ __forceinline__ bool track_eval(const Pose& einv_t, const Mat3& k_t, const float xw[3], float gt_x, float gt_y, int kind,
                                           float delta, float ax, float ay, TrackEval& o) {
  float xc[3];
  apply_pose(einv_t, xw, xc);
  o.pr = project_point(xc, k_t);
  if (!(o.pr.u >= 0.f && o.pr.v >= 0.f && o.pr.u < 1.f && o.pr.v < 1.f)) return false;
  o.rho = robust_map(kind, delta, aspect_diff(o.pr.u, gt_x, ax), aspect_diff(o.pr.v, gt_y, ay), o.drx, o.dry);
  return true;
}

// ------------------------------------------------------------------- track_fwd ------
__global__ void __launch_bounds__(256) track_fwd_kernel(TrackGeom g, const float* ws, const uint8_t* flag, const float* ext_inv,
                                                        const float* k, int kind, float delta, float ax, float ay, int frames,
                                                        double* acc) {
  __shared__ double red[4 * 20];
  const int sg = g.blocks[blockIdx.x * 2], ft = g.blocks[blockIdx.x * 2 + 1];
  const int start = g.seg[sg * 4], f = g.seg[sg * 4 + 1], p_count = g.seg[sg * 4 + 2], off = g.seg[sg * 4 + 3];
  const int p = blockIdx.y * blockDim.x + threadIdx.x;
  const int frame_t = start + ft;
  Pose einv;
  Mat3 kt;
  load_pose44(ext_inv + (size_t)frame_t * 16, einv);
  load_mat3(k + (size_t)frame_t * 9, kt);
  float a[20];
#pragma unroll
  for (int i = 0; i < 20; ++i) a[i] = 0.f;
  if (p < p_count) {
    const size_t it = (size_t)off + (size_t)ft * p_count + p;
    if (g.vis[it] != 0) {  // target role needs only the track's visibility (projection.py:290)
      const float2 gt = reinterpret_cast<const float2*>(g.xy)[it];
      {
        for (int fs = 0; fs < f; ++fs) {
          const size_t is = (size_t)off + (size_t)fs * p_count + p;
          if (flag[is] == 0) continue;
          const float* w6 = ws + is * 6;
          const float xw[3] = {w6[3], w6[4], w6[5]};
          TrackEval ev;
          if (!track_eval(einv, kt, xw, gt.x, gt.y, kind, delta, ax, ay, ev)) continue;
          a[18] += ev.rho;
          a[19] += 1.f;
          float gk[6] = {0, 0, 0, 0, 0, 0}, gxc[3];
          project_point_bwd(ev.pr, kt, ev.drx * ax, ev.dry * ay, gk, gxc);
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            a[r * 4 + 0] += gxc[r] * xw[0];
            a[r * 4 + 1] += gxc[r] * xw[1];
            a[r * 4 + 2] += gxc[r] * xw[2];
            a[r * 4 + 3] += gxc[r];
          }
#pragma unroll
          for (int r = 0; r < 6; ++r) a[12 + r] += gk[r];
        }
      }
    }
  }
  // per-frame sums [0..17]; global loss / count live after the last frame
  float per_frame[18], glob[2] = {a[18], a[19]};
#pragma unroll
  for (int i = 0; i < 18; ++i) per_frame[i] = a[i];
  block_accumulate<18>(per_frame, red, acc + (size_t)frame_t * kTrackAccStride);
  block_accumulate<2>(glob, red, acc + (size_t)frames * kTrackAccStride);
}

// loss[0] = weight·Σρ/max(count,1); scale[0] = weight/max(count,1); scale[1] = count
__global__ void track_finalize_fwd_kernel(const double* acc, int frames, float weight, float* loss, float* scale) {
  const double sum = acc[(size_t)frames * kTrackAccStride], cnt = acc[(size_t)frames * kTrackAccStride + 1];
  const double den = cnt != 0.0 ? cnt : 1.0;  // `valid_sum or 1` (loss_tracking.py:61)
  loss[0] = (float)((double)weight * sum / den);
  scale[0] = (float)((double)weight / den);
  scale[1] = (float)cnt;
}

// ------------------------------------------------------------------- track_bwd ------
__global__ void __launch_bounds__(256) track_bwd_kernel(TrackGeom g, const float* ws, const uint8_t* flag, const float* depth,
                                                        const float* kinv, const float* ext, const float* ext_inv, const float* k,
                                                        int kind, float delta, float ax, float ay, const float* scale,
                                                        const float* upstream, float* gws, double* acc2) {
  extern __shared__ double lds_d[];  // reduction scratch (fp64), then [f][21] target poses + intrinsics
  const int sg = g.blocks[blockIdx.x * 2], fs = g.blocks[blockIdx.x * 2 + 1];
  const int start = g.seg[sg * 4], f = g.seg[sg * 4 + 1], p_count = g.seg[sg * 4 + 2], off = g.seg[sg * 4 + 3];
  double* red = lds_d;
  float* tgt = reinterpret_cast<float*>(lds_d + 4 * 21);
  for (int i = threadIdx.x; i < f * 21; i += blockDim.x) {
    const int ft = i / 21, e = i % 21;
    const int frame = start + ft;
    tgt[i] = e < 12 ? ext_inv[(size_t)frame * 16 + (e / 4) * 4 + (e % 4)] : k[(size_t)frame * 9 + (e - 12)];
  }
  __syncthreads();
  const int p = blockIdx.y * blockDim.x + threadIdx.x;
  const int frame_s = start + fs;
  const float sc = scale[0] * (upstream ? upstream[0] : 1.f);
  float a[21];
#pragma unroll
  for (int i = 0; i < 21; ++i) a[i] = 0.f;
  if (p < p_count) {
    const size_t is = (size_t)off + (size_t)fs * p_count + p;
    if (flag[is] != 0) {
      const float* w6 = ws + is * 6;
      const float xyz[3] = {w6[0], w6[1], w6[2]};
      const float xw[3] = {w6[3], w6[4], w6[5]};
      float gxw[3] = {0.f, 0.f, 0.f};
      for (int ft = 0; ft < f; ++ft) {
        const size_t it = (size_t)off + (size_t)ft * p_count + p;
        if (g.vis[it] == 0) continue;
        const float* tp = tgt + ft * 21;
        Pose einv;
        Mat3 kt;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          einv.r[r * 3 + 0] = tp[r * 4 + 0];
          einv.r[r * 3 + 1] = tp[r * 4 + 1];
          einv.r[r * 3 + 2] = tp[r * 4 + 2];
          einv.t[r] = tp[r * 4 + 3];
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) kt.m[i] = tp[12 + i];
        const float2 gt = reinterpret_cast<const float2*>(g.xy)[it];
        TrackEval ev;
        if (!track_eval(einv, kt, xw, gt.x, gt.y, kind, delta, ax, ay, ev)) continue;
        float gk[6] = {0, 0, 0, 0, 0, 0}, gxc[3], gx[3];
        project_point_bwd(ev.pr, kt, ev.drx * ax, ev.dry * ay, gk, gxc);
        apply_rot_t(einv, gxc, gx);
        gxw[0] += gx[0];
        gxw[1] += gx[1];
        gxw[2] += gx[2];
      }
      gxw[0] *= sc;
      gxw[1] *= sc;
      gxw[2] *= sc;
      // X_w = E_fs·[xyz;1]
      Pose e;
      load_pose44(ext + (size_t)frame_s * 16, e);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        a[r * 4 + 0] = gxw[r] * xyz[0];
        a[r * 4 + 1] = gxw[r] * xyz[1];
        a[r * 4 + 2] = gxw[r] * xyz[2];
        a[r * 4 + 3] = gxw[r];
      }
      float gxyz[3];
      apply_rot_t(e, gxw, gxyz);
      // xyz = Σ taps w_k · z_k · Kinv·[u_k, v_k, 1]
      const float2 q = reinterpret_cast<const float2*>(g.xy)[is];
      const Taps t = bilinear_taps(q.x, q.y, g.height, g.width);
      const float* d = depth + (size_t)frame_s * g.height * g.width;
      gws[is * 3 + 0] = gxyz[0];
      gws[is * 3 + 1] = gxyz[1];
      gws[is * 3 + 2] = gxyz[2];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (!t.in[kk]) continue;
        const int tc = tap_col(t, kk), tr = tap_row(t, kk);
        const float ut = pixel_center(tc, g.width), vt = pixel_center(tr, g.height);
        const float wt = t.w[kk];
        const float z = d[tr * g.width + tc];
        const float zt[3] = {z * ut * wt, z * vt * wt, z * wt};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          a[12 + r * 3 + 0] += gxyz[r] * zt[0];
          a[12 + r * 3 + 1] += gxyz[r] * zt[1];
          a[12 + r * 3 + 2] += gxyz[r] * zt[2];
        }
      }
    }
  }
  block_accumulate<21>(a, red, acc2 + (size_t)frame_s * kTrackAcc2Stride);
}

// Scatter of the per-point surface gradients gws (total,3) into dL/ddepth through the
// bilinear taps: xyz = Σ_k w_k · z_k · Kinv·[u_k, v_k, 1].  Separate launch so the caller
// can aim it at whichever dense buffer will finally hold dL/ddepth.
__global__ void __launch_bounds__(256) track_scatter_kernel(TrackGeom g, const uint8_t* flag, const float* gws, const float* kinv,
                                                            float* grad_depth) {
  const int sg = g.blocks[blockIdx.x * 2], fs = g.blocks[blockIdx.x * 2 + 1];
  const int start = g.seg[sg * 4], p_count = g.seg[sg * 4 + 2], off = g.seg[sg * 4 + 3];
  const int p = blockIdx.y * blockDim.x + threadIdx.x;
  if (p >= p_count) return;
  const size_t is = (size_t)off + (size_t)fs * p_count + p;
  if (flag[is] == 0) return;
  const int frame_s = start + fs;
  Mat3 ki;
  load_mat3(kinv + (size_t)frame_s * 9, ki);
  const float gx = gws[is * 3], gy = gws[is * 3 + 1], gz = gws[is * 3 + 2];
  const float2 q = reinterpret_cast<const float2*>(g.xy)[is];
  const Taps t = bilinear_taps(q.x, q.y, g.height, g.width);
  float* gd = grad_depth + (size_t)frame_s * g.height * g.width;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    if (!t.in[kk]) continue;
    const int tc = tap_col(t, kk), tr = tap_row(t, kk);
    float ray[3];
    ray_dir(ki, pixel_center(tc, g.width), pixel_center(tr, g.height), ray);
    atomicAdd(gd + tr * g.width + tc, t.w[kk] * (gx * ray[0] + gy * ray[1] + gz * ray[2]));
  }
}

// dL/dE and dL/dK per frame from the two accumulators.
__global__ void track_finalize_bwd_kernel(const double* acc, const double* acc2, const float* scale, const float* upstream,
                                          const float* ext_inv, const float* kinv, int frames, float* g_ext, float* g_k) {
  const int fr = blockIdx.x * blockDim.x + threadIdx.x;
  if (fr >= frames) return;
  const double sc = (double)scale[0] * (upstream ? (double)upstream[0] : 1.0);
  const double* a = acc + (size_t)fr * kTrackAccStride;
  const double* b = acc2 + (size_t)fr * kTrackAcc2Stride;
  // target role: X' = inv(E)·X_w, G_inv = Σ gX'⊗[X_w;1] (top 3 rows) -> dE = −invᵀ·G_inv·invᵀ
  double ginv[16], inv[16], tmp[16], tmp2[16];
  for (int i = 0; i < 12; ++i) ginv[i] = a[i] * sc;
  for (int i = 12; i < 16; ++i) ginv[i] = 0.0;
  for (int i = 0; i < 16; ++i) inv[i] = ext_inv[(size_t)fr * 16 + i];
  mat4_mul_tn(inv, ginv, tmp);
  mat4_mul_nt(tmp, inv, tmp2);
  float* ge = g_ext + (size_t)fr * 16;
  for (int i = 0; i < 16; ++i) ge[i] = (float)(-tmp2[i] + (i < 12 ? b[i] : 0.0));
  // intrinsics: destination role (rows 0,1; unscaled in acc) + source role through K⁻¹
  double gk[9];
  kinv_grad_to_k(b + 12, kinv + (size_t)fr * 9, gk);
  for (int i = 0; i < 6; ++i) gk[i] += a[12 + i] * sc;
  for (int i = 0; i < 9; ++i) g_k[(size_t)fr * 9 + i] = (float)gk[i];
}

__global__ void inv4_kernel(const float* m, int count, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double a[16], o[16];
  for (int k = 0; k < 16; ++k) a[k] = m[(size_t)i * 16 + k];
  inv4(a, o);
  for (int k = 0; k < 16; ++k) out[(size_t)i * 16 + k] = (float)o[k];
}

}  // namespace fm

using namespace fm;

extern "C" {

int fm_extrinsics_inverse(const float* ext, int count, float* inv, void* stream) {
  FM_CHECK_ARG(ext && inv && count >= 1);
  hipLaunchKernelGGL(inv4_kernel, dim3((count + 63) / 64), dim3(64), 0, (hipStream_t)stream, ext, count, inv);
  FM_LAUNCH_STATUS();
}

int fm_track_points(const float* depth, const float* kinv, const float* ext, const float* xy, const uint8_t* vis,
                    const int32_t* seg, const int32_t* blocks, int nblocks, int pmax, int height, int width, float* ws,
                    uint8_t* flag, void* stream) {
  FM_CHECK_ARG(depth && kinv && ext && xy && vis && seg && blocks && ws && flag && nblocks >= 1 && pmax >= 1);
  TrackGeom g{xy, vis, seg, blocks, height, width};
  hipLaunchKernelGGL(track_points_kernel, dim3(nblocks, (pmax + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, depth, kinv, ext,
                     ws, flag);
  FM_LAUNCH_STATUS();
}

int fm_track_loss_fwd(const float* ws, const uint8_t* flag, const float* xy, const uint8_t* vis, const int32_t* seg,
                      const int32_t* blocks, int nblocks, int pmax, const float* ext_inv, const float* k, int frames, int height,
                      int width, int mapping_kind, float delta, float aspect_x, float aspect_y, float weight, double* acc,
                      float* loss, float* scale, void* stream) {
  FM_CHECK_ARG(ws && flag && xy && vis && seg && blocks && ext_inv && k && acc && loss && scale && nblocks >= 1 && pmax >= 1);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(acc, 0, sizeof(double) * ((size_t)frames * kTrackAccStride + 2), st) != hipSuccess) return FM_ERR_LAUNCH;
  TrackGeom g{xy, vis, seg, blocks, height, width};
  hipLaunchKernelGGL(track_fwd_kernel, dim3(nblocks, (pmax + 255) / 256), dim3(256), 0, st, g, ws, flag, ext_inv, k, mapping_kind,
                     delta, aspect_x, aspect_y, frames, acc);
  hipLaunchKernelGGL(track_finalize_fwd_kernel, dim3(1), dim3(1), 0, st, acc, frames, weight, loss, scale);
  FM_LAUNCH_STATUS();
}

int fm_track_loss_bwd(const float* ws, const uint8_t* flag, const float* xy, const uint8_t* vis, const int32_t* seg,
                      const int32_t* blocks, int nblocks, int pmax, int fmax, const float* depth, const float* kinv, const float* ext,
                      const float* ext_inv, const float* k, int frames, int height, int width, int mapping_kind, float delta,
                      float aspect_x, float aspect_y, const double* acc, const float* scale, const float* upstream,
                      float* gws, double* acc2, float* g_ext, float* g_k, void* stream) {
  FM_CHECK_ARG(ws && flag && xy && vis && seg && blocks && depth && kinv && ext && ext_inv && k && acc && scale && gws && acc2 && g_ext && g_k);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(acc2, 0, sizeof(double) * (size_t)frames * kTrackAcc2Stride, st) != hipSuccess) return FM_ERR_LAUNCH;
  TrackGeom g{xy, vis, seg, blocks, height, width};
  const size_t lds = sizeof(float) * (size_t)fmax * 21 + sizeof(double) * 4 * 21;
  hipLaunchKernelGGL(track_bwd_kernel, dim3(nblocks, (pmax + 255) / 256), dim3(256), lds, st, g, ws, flag, depth, kinv, ext, ext_inv,
                     k, mapping_kind, delta, aspect_x, aspect_y, scale, upstream, gws, acc2);
  hipLaunchKernelGGL(track_finalize_bwd_kernel, dim3((frames + 63) / 64), dim3(64), 0, st, acc, acc2, scale, upstream, ext_inv, kinv,
                     frames, g_ext, g_k);
  FM_LAUNCH_STATUS();
}

int fm_track_scatter(const float* gws, const uint8_t* flag, const float* xy, const uint8_t* vis, const int32_t* seg,
                     const int32_t* blocks, int nblocks, int pmax, const float* kinv, int height, int width, float* grad_depth,
                     void* stream) {
  FM_CHECK_ARG(gws && flag && xy && vis && seg && blocks && kinv && grad_depth && nblocks >= 1 && pmax >= 1);
  TrackGeom g{xy, vis, seg, blocks, height, width};
  hipLaunchKernelGGL(track_scatter_kernel, dim3(nblocks, (pmax + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, flag, gws, kinv,
                     grad_depth);
  FM_LAUNCH_STATUS();
}

}  // extern "C"
