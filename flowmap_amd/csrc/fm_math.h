// Per-element math of FlowMap's reprojection / flow-consistency hot path.
//
// Everything here is a plain inline function usable from HIP device code (hipcc) and
// from a host C++ compiler (g++, used only by tests/host_sim to check the analytic
// gradients on the CPU before spending GPU minutes).  No memory traffic, no launch
// logic: kernels in *.hip own indexing, coalescing and reductions.
//
// Reference semantics restated (file:line are relative to dcharatan/flowmap):
//   unproject              flowmap/model/projection.py:76-90
//   reproject_points       flowmap/model/projection.py:116-134
//   project_camera_space   flowmap/model/projection.py:49-58   (eps=1e-5, inf=1e8)
//   grid_sample(bilinear, border, align_corners=False)  projection.py:235-241,266-272
//   Mapping / Huber / L1 / L2   flowmap/loss/mapping/*.py
//   align_rigid            flowmap/model/procrustes.py:7-51
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define FM_HD __host__ __device__ __forceinline__
#else
#define FM_HD inline
#endif

namespace fm {

constexpr float kProjEps = 1e-5f;  // projection.py:52
constexpr float kProjInf = 1e8f;   // projection.py:53

enum MappingKind : int { kHuber = 0, kL1 = 1, kL2 = 2 };

struct Mat3 {  // row-major 3x3
  float m[9];
};

struct Pose {  // top three rows of a 4x4 [R | t]
  float r[9];
  float t[3];
};

FM_HD void load_mat3(const float* p, Mat3& o) {
  for (int i = 0; i < 9; ++i) o.m[i] = p[i];
}

// Top 3 rows of a row-major 4x4.
FM_HD void load_pose44(const float* p, Pose& o) {
  for (int r = 0; r < 3; ++r) {
    o.r[r * 3 + 0] = p[r * 4 + 0];
    o.r[r * 3 + 1] = p[r * 4 + 1];
    o.r[r * 3 + 2] = p[r * 4 + 2];
    o.t[r] = p[r * 4 + 3];
  }
}

// Pixel-centre coordinate, evaluated like sample_image_grid (projection.py:109):
// (int + 0.5) / length in fp32 with a true division.
FM_HD float pixel_center(int i, int n) { return ((float)i + 0.5f) / (float)n; }

// ray = Kinv · [u, v, 1]   (projection.py:84-87)
FM_HD void ray_dir(const Mat3& kinv, float u, float v, float ray[3]) {
  ray[0] = kinv.m[0] * u + kinv.m[1] * v + kinv.m[2];
  ray[1] = kinv.m[3] * u + kinv.m[4] * v + kinv.m[5];
  ray[2] = kinv.m[6] * u + kinv.m[7] * v + kinv.m[8];
}

// nan_to_num(posinf=1e8, neginf=-1e8) with its autograd mask (projection.py:56).
FM_HD float nan_to_num(float x, float& pass) {
  if (x != x) {
    pass = 0.f;
    return 0.f;
  }
  if (x > 3.0e38f) {
    pass = 0.f;
    return kProjInf;
  }
  if (x < -3.0e38f) {
    pass = 0.f;
    return -kProjInf;
  }
  pass = 1.f;
  return x;
}

// Result of projecting one camera-space point X' with destination intrinsics K.
struct Projected {
  float u, v;        // image position
  float p[3];        // X'/(Z'+eps) after nan_to_num
  float inv_s;       // 1/(Z'+eps)
  float pass[3];     // nan_to_num gradient masks
};

FM_HD Projected project_point(const float xc[3], const Mat3& k) {
  Projected o;
  const float s = xc[2] + kProjEps;
  o.inv_s = 1.0f / s;
  // The reference divides each component by s; x * (1/s) differs by <=1 ulp.
  for (int a = 0; a < 3; ++a) o.p[a] = nan_to_num(xc[a] * o.inv_s, o.pass[a]);
  o.u = k.m[0] * o.p[0] + k.m[1] * o.p[1] + k.m[2] * o.p[2];
  o.v = k.m[3] * o.p[0] + k.m[4] * o.p[1] + k.m[5] * o.p[2];
  return o;
}

// Back-propagate (gu, gv) = dL/d(u,v) through project_point.
//   gk[6] += outer((gu,gv), p)       (rows 0,1 of K)
//   gxc    = dL/dX'
// At the exact singularity Z' = -eps the reference's gradient is NaN (inf·0); we
// define it as 0 (SURVEY.md §7 "hard parts").
FM_HD void project_point_bwd(const Projected& f, const Mat3& k, float gu, float gv, float gk[6], float gxc[3]) {
  gk[0] += gu * f.p[0];
  gk[1] += gu * f.p[1];
  gk[2] += gu * f.p[2];
  gk[3] += gv * f.p[0];
  gk[4] += gv * f.p[1];
  gk[5] += gv * f.p[2];
  float gp[3];
  gp[0] = (k.m[0] * gu + k.m[3] * gv) * f.pass[0];
  gp[1] = (k.m[1] * gu + k.m[4] * gv) * f.pass[1];
  gp[2] = (k.m[2] * gu + k.m[5] * gv) * f.pass[2];
  const bool finite = (f.pass[0] + f.pass[1] + f.pass[2]) == 3.f;
  const float is = finite ? f.inv_s : 0.f;
  const float dot = gp[0] * f.p[0] + gp[1] * f.p[1] + gp[2] * f.p[2];
  gxc[0] = gp[0] * is;
  gxc[1] = gp[1] * is;
  gxc[2] = (gp[2] - dot) * is;
}

FM_HD void apply_pose(const Pose& t, const float x[3], float o[3]) {
  // einsum over the homogeneous point: r0·x0 + r1·x1 + r2·x2 + t·1 (projection.py:127-131)
  o[0] = t.r[0] * x[0] + t.r[1] * x[1] + t.r[2] * x[2] + t.t[0];
  o[1] = t.r[3] * x[0] + t.r[4] * x[1] + t.r[5] * x[2] + t.t[1];
  o[2] = t.r[6] * x[0] + t.r[7] * x[1] + t.r[8] * x[2] + t.t[2];
}

FM_HD void apply_rot_t(const Pose& t, const float g[3], float o[3]) {  // Rᵀ g
  o[0] = t.r[0] * g[0] + t.r[3] * g[1] + t.r[6] * g[2];
  o[1] = t.r[1] * g[0] + t.r[4] * g[1] + t.r[7] * g[2];
  o[2] = t.r[2] * g[0] + t.r[5] * g[1] + t.r[8] * g[2];
}

// fix_aspect_ratio(a) − fix_aspect_ratio(b) (mapping.py:41-43): two ROUNDED products, then
// the difference.  Must not be contracted into an FMA: with a == b the reference gets an
// exact 0 (and ‖·‖'s sub-gradient 0 there), a fused multiply-add leaves the rounding
// residue of one product and turns that into a unit-length L1 gradient.
FM_HD float aspect_diff(float a, float b, float s) {
#if defined(__clang__)
#pragma clang fp contract(off)
  const float pa = a * s;
  const float pb = b * s;
  return pa - pb;
#else
  volatile float pa = a * s, pb = b * s;
  return pa - pb;
#endif
}

// Robust mapping of an aspect-corrected residual (rx, ry): value and d(value)/d(rx,ry).
//   huber: F.huber_loss(n, 0, delta)/delta  (mapping_huber.py:23-34)
//   l1:    n                                 (mapping_l1.py:20)
//   l2:    0.5 (rx² + ry²)                   (mapping_l2.py:21)
// ‖·‖ has sub-gradient 0 at 0 (torch's norm backward masks n == 0).
FM_HD float robust_map(int kind, float delta, float rx, float ry, float& drx, float& dry) {
  const float ss = rx * rx + ry * ry;
  if (kind == kL2) {
    drx = rx;
    dry = ry;
    return 0.5f * ss;
  }
  const float n = sqrtf(ss);
  const float inv_n = n > 0.f ? 1.0f / n : 0.f;
  if (kind == kL1 || !(n < delta)) {
    drx = rx * inv_n;
    dry = ry * inv_n;
    return kind == kL1 ? n : (delta * (n - 0.5f * delta)) / delta;
  }
  const float inv_d = 1.0f / delta;
  drx = rx * inv_d;
  dry = ry * inv_d;
  return (0.5f * n * n) * inv_d;
}

// ---------------------------------------------------------------------------------
// Bilinear sampling with border padding, align_corners=False (ATen grid_sampler_2d).
// ---------------------------------------------------------------------------------
struct Taps {
  int x0, y0;          // north-west tap
  float w[4];          // nw, ne, sw, se
  bool in[4];          // tap inside the image
};

// coordinate in normalised (0,1) units -> taps; the reference feeds grid_sample with
// xy*2-1 (projection.py:237,268) and ATen un-normalises with ((g+1)*size-1)/2.
FM_HD Taps bilinear_taps(float x01, float y01, int h, int w) {
  Taps t;
  float gx = x01 * 2.f - 1.f;
  float gy = y01 * 2.f - 1.f;
  float ix = ((gx + 1.f) * (float)w - 1.f) / 2.f;
  float iy = ((gy + 1.f) * (float)h - 1.f) / 2.f;
  ix = fminf((float)(w - 1), fmaxf(ix, 0.f));
  iy = fminf((float)(h - 1), fmaxf(iy, 0.f));
  const float fx = floorf(ix), fy = floorf(iy);
  t.x0 = (int)fx;
  t.y0 = (int)fy;
  const float ex = fx + 1.f, ey = fy + 1.f;
  t.w[0] = (ex - ix) * (ey - iy);
  t.w[1] = (ix - fx) * (ey - iy);
  t.w[2] = (ex - ix) * (iy - fy);
  t.w[3] = (ix - fx) * (iy - fy);
  const bool xin0 = t.x0 >= 0 && t.x0 < w, xin1 = t.x0 + 1 >= 0 && t.x0 + 1 < w;
  const bool yin0 = t.y0 >= 0 && t.y0 < h, yin1 = t.y0 + 1 >= 0 && t.y0 + 1 < h;
  t.in[0] = xin0 && yin0;
  t.in[1] = xin1 && yin0;
  t.in[2] = xin0 && yin1;
  t.in[3] = xin1 && yin1;
  return t;
}

FM_HD int tap_col(const Taps& t, int k) { return t.x0 + (k & 1); }
FM_HD int tap_row(const Taps& t, int k) { return t.y0 + (k >> 1); }

// ---------------------------------------------------------------------------------
// Flow post-processing (flowmap/flow/flow_predictor.py:39-102), SURVEY.md §8f rank 3.
//
// consistency_mask_at: FlowPredictor.compute_consistency_mask (:60-80) at ONE source pixel:
//   target colour = grid_sample(target, (xy + flow)·2 − 1, bilinear, padding "zeros",
//   align_corners=False);  δ = max_c |source_c − target_c|;  mask = (1 − δ)⁸.
// src / tgt address one frame stored (3, H, W).
// ---------------------------------------------------------------------------------
FM_HD float consistency_mask_at(const float* src, const float* tgt, int h, int w, int row, int col, float flow_x, float flow_y) {
#pragma clang fp contract(off)
  const float gx = (pixel_center(col, w) + flow_x) * 2.f - 1.f;
  const float gy = (pixel_center(row, h) + flow_y) * 2.f - 1.f;
  const float ix = ((gx + 1.f) * (float)w - 1.f) / 2.f;
  const float iy = ((gy + 1.f) * (float)h - 1.f) / 2.f;
  const float fx = floorf(ix), fy = floorf(iy);
  const float we = ix - fx, ww = 1.f - we, ws = iy - fy, wn = 1.f - ws;
  // the float -> int conversions saturate, so far-away samples simply fail the range tests
  const float cx = fminf(fmaxf(fx, -2.f), (float)w + 1.f), cy = fminf(fmaxf(fy, -2.f), (float)h + 1.f);
  const int x0 = (int)cx, y0 = (int)cy;
  const bool ok = ix == ix && iy == iy;  // NaN coordinates sample nothing
  const bool xin0 = ok && x0 >= 0 && x0 < w, xin1 = ok && x0 + 1 >= 0 && x0 + 1 < w;
  const bool yin0 = ok && y0 >= 0 && y0 < h, yin1 = ok && y0 + 1 >= 0 && y0 + 1 < h;
  const size_t plane = (size_t)h * w;
  float delta = 0.f;
  for (int c = 0; c < 3; ++c) {
    const float* t = tgt + c * plane;
    const float nw = (xin0 && yin0) ? t[(size_t)y0 * w + x0] : 0.f;
    const float ne = (xin1 && yin0) ? t[(size_t)y0 * w + x0 + 1] : 0.f;
    const float sw = (xin0 && yin1) ? t[(size_t)(y0 + 1) * w + x0] : 0.f;
    const float se = (xin1 && yin1) ? t[(size_t)(y0 + 1) * w + x0 + 1] : 0.f;
    const float sampled = nw * (wn * ww) + ne * (wn * we) + sw * (ws * ww) + se * (ws * we);
    delta = fmaxf(delta, fabsf(src[c * plane + (size_t)row * w + col] - sampled));
  }
  const float b = 1.f - delta, b2 = b * b, b4 = b2 * b2;
  return b4 * b4;
}

// F.interpolate(mode="bilinear", align_corners=False) source taps for output index `o`
// (rescale_flow / rescale_mask, flow_predictor.py:39-57; ATen area_pixel_compute_source_index).
struct ResizeTap {
  int i0, i1;
  float l0, l1;
};

FM_HD ResizeTap resize_tap(int o, int in_size, int out_size) {
#pragma clang fp contract(off)
  ResizeTap t;
  if (in_size == out_size) {
    t.i0 = t.i1 = o;
    t.l0 = 1.f;
    t.l1 = 0.f;
    return t;
  }
  const float scale = (float)in_size / (float)out_size;
  float real = scale * ((float)o + 0.5f) - 0.5f;
  real = fminf(fmaxf(real, 0.f), (float)(in_size - 1));
  t.i0 = (int)real;
  t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
  t.l1 = fminf(fmaxf(real - (float)t.i0, 0.f), 1.f);
  t.l0 = 1.f - t.l1;
  return t;
}

// One output pixel of the fused post-processing: bilinear resize of the raw flow AND of the
// consistency mask, the mask being evaluated only at the four full-resolution taps the
// resize reads (a 4x down-scale touches a quarter of the full-resolution pixels).
FM_HD void flow_postprocess_at(const float* src, const float* tgt, const float* flow, int h, int w, int oh, int ow, int orow, int ocol,
                               float out_flow[2], float& out_mask) {
#pragma clang fp contract(off)
  const ResizeTap ty = resize_tap(orow, h, oh), tx = resize_tap(ocol, w, ow);
  const int rows[2] = {ty.i0, ty.i1}, cols[2] = {tx.i0, tx.i1};
  float f[2][2][2], m[2][2];
  for (int a = 0; a < 2; ++a)
    for (int c = 0; c < 2; ++c) {
      const float* fl = flow + ((size_t)rows[a] * w + cols[c]) * 2;
      f[a][c][0] = fl[0];
      f[a][c][1] = fl[1];
      m[a][c] = consistency_mask_at(src, tgt, h, w, rows[a], cols[c], fl[0], fl[1]);
    }
  for (int k = 0; k < 2; ++k)
    out_flow[k] = ty.l0 * (tx.l0 * f[0][0][k] + tx.l1 * f[0][1][k]) + ty.l1 * (tx.l0 * f[1][0][k] + tx.l1 * f[1][1][k]);
  out_mask = ty.l0 * (tx.l0 * m[0][0] + tx.l1 * m[0][1]) + ty.l1 * (tx.l0 * m[1][0] + tx.l1 * m[1][1]);
}

// ---------------------------------------------------------------------------------
// One flow residual of the FUSED flow loss (fm_flow.hip; tests/host_sim runs the same code).
//
// With m = R·K⁻¹_src (X' = z·m·[u,v,1] + t) and kd = rows 0,1 of the destination intrinsics
// PRE-SCALED by the aspect factors (fix_aspect_ratio, mapping.py:17-23; r = kd·p − aspect·(xy+flow)),
// the per (source frame, direction) constants are the twelve numbers of DirConst below
// (wave-uniform, SGPRs on the GPU): kd·X' and Z' are affine in the pixel, so neither the source
// ray, nor the camera-space point, nor p = X'/(Z'+eps) is formed, and dL/dz = ω·(a, b, −c).
// Per pixel the residual's gradient enters every pose / intrinsics gradient only through
//   ω = (q·w_u, q·w_v, q·(w_u·pu + w_v·pv)),   q = 1/(Z'+eps),  w = dL/d(kd·p),
// so ONE 3x3 sum Ω = Σ ω ⊗ z[u,v,1] and σ = Σ ω carry all of them (flow_finalize_frame):
//   dL/dX' = A·ω with A = [[kd00,kd10,0],[kd01,kd11,0],[kd02,kd12,-1]]
//   S = Σ dL/dX' ⊗ z h = A·Ω     (dL/dR = S·K⁻ᵀ, dL/dK⁻¹_src = Rᵀ·S),   dL/dt = A·σ
//   dL/dkd[a][b] = Σ w_a p_b = Σ_c m[b][c]·Ω[a][c] + t[b]·σ[a]          (p = q·X')
//   acc[0] += ρ·mask    acc[1..3] += ω    acc[4..12] += ω ⊗ (z·u, z·v, z)
//
// Division / square root: hardware reciprocal and reciprocal-sqrt (1 ulp) on the GPU,
// IEEE on the host.  The singular case Z'+1e-5 == 0 (reference: projection clamped to
// ±1e8, GRADIENT NaN — SURVEY.md A.4) is dropped: that pixel's mask is zeroed.  The
// function-level reproject kernels keep the exact clamp semantics.
// ---------------------------------------------------------------------------------
constexpr int kFlowAcc = 13;

FM_HD float fm_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}
FM_HD float fm_rsq(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rsqf(x);
#else
  return 1.0f / sqrtf(x);
#endif
}

// kd·X' is affine in the source pixel:  kd·(z·m·[u,v,1] + t) = z·(au·u + a1·v + a2) + ta  etc.
// (kd = aspect-scaled rows 0,1 of K_dst, m = R·K⁻¹_src), so neither the camera-space point nor
// p = X'/(Z'+eps) is ever formed:  pu = q·(z·a + ta),  dL/dz = ω·(a, b, −c).
struct DirConst {
  float au, a1, a2, ta;  // a(u,v) = kd_row0·m·[u,v,1],  ta = kd_row0·t
  float bu, b1, b2, tb;  // b(u,v) = kd_row1·m·[u,v,1],  tb = kd_row1·t
  float cu, c1, c2, tc;  // c(u,v) = m_row2·[u,v,1] (Z' = z·c + tc),  tc = t[2]
};

FM_HD void make_dir(const Pose& pose, const Mat3& kinv, const Mat3& kd, float ax, float ay, DirConst& d) {
  double m[9], k0[3], k1[3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      m[r * 3 + c] = (double)pose.r[r * 3 + 0] * kinv.m[0 * 3 + c] + (double)pose.r[r * 3 + 1] * kinv.m[1 * 3 + c] +
                     (double)pose.r[r * 3 + 2] * kinv.m[2 * 3 + c];
  for (int i = 0; i < 3; ++i) {
    k0[i] = (double)kd.m[i] * ax;
    k1[i] = (double)kd.m[3 + i] * ay;
  }
  // (field by field: indexing through a pointer into the struct sends it to scratch memory on the device)
  d.au = (float)(k0[0] * m[0] + k0[1] * m[3] + k0[2] * m[6]);
  d.a1 = (float)(k0[0] * m[1] + k0[1] * m[4] + k0[2] * m[7]);
  d.a2 = (float)(k0[0] * m[2] + k0[1] * m[5] + k0[2] * m[8]);
  d.bu = (float)(k1[0] * m[0] + k1[1] * m[3] + k1[2] * m[6]);
  d.b1 = (float)(k1[0] * m[1] + k1[1] * m[4] + k1[2] * m[7]);
  d.b2 = (float)(k1[0] * m[2] + k1[1] * m[5] + k1[2] * m[8]);
  d.cu = (float)m[6];
  d.c1 = (float)m[7];
  d.c2 = (float)m[8];
  d.ta = (float)(k0[0] * pose.t[0] + k0[1] * pose.t[1] + k0[2] * pose.t[2]);
  d.tb = (float)(k1[0] * pose.t[0] + k1[1] * pose.t[1] + k1[2] * pose.t[2]);
  d.tc = pose.t[2];
}

// arow / brow / crow = a1·v + a2 etc. (constant along an image row); u_ax, v_ay = aspect·(u, v).
template <int KIND, bool GRAD>
FM_HD void flow_term_fast(const DirConst& d, float arow, float brow, float crow, float z, float u, float zu, float zv, float u_ax,
                          float v_ay, float flow_x, float flow_y, float m, float scale, float delta, float inv_delta, float ax,
                          float ay, float (&acc)[kFlowAcc], float& gz) {
  const float a = fmaf(d.au, u, arow);
  const float b = fmaf(d.bu, u, brow);
  const float c = fmaf(d.cu, u, crow);
  const float xu = fmaf(z, a, d.ta);
  const float xv = fmaf(z, b, d.tb);
  const float x2 = fmaf(z, c, d.tc);
  float q = fm_rcp(x2 + kProjEps);
  const bool ok = fabsf(q) <= 3.0e38f;
  q = ok ? q : 0.f;
  m = ok ? m : 0.f;
  const float pu = xu * q;
  const float pv = xv * q;
  const float rx = pu - fmaf(flow_x, ax, u_ax);
  const float ry = pv - fmaf(flow_y, ay, v_ay);
  const float ss = fmaf(rx, rx, ry * ry);
  float rho, coef;  // ρ and dρ/dr = coef·r
  if (KIND == kL2) {
    rho = 0.5f * ss;
    coef = 1.f;
  } else {
    const float inv_n = ss > 0.f ? fm_rsq(ss) : 0.f;
    const float n = ss * inv_n;
    if (KIND == kL1) {
      rho = n;
      coef = inv_n;
    } else {
      const bool quad = n < delta;
      rho = quad ? 0.5f * ss * inv_delta : n - 0.5f * delta;
      coef = quad ? inv_delta : inv_n;
    }
  }
  acc[0] = fmaf(rho, m, acc[0]);
  if (GRAD) {
    const float gc = (scale * m) * coef;
    const float wu = gc * rx, wv = gc * ry;  // dL/d(kd·p)
    const float o0 = q * wu, o1 = q * wv, o2 = q * fmaf(wu, pu, wv * pv);
    acc[1] += o0;
    acc[2] += o1;
    acc[3] += o2;
    acc[4] = fmaf(o0, zu, acc[4]);
    acc[5] = fmaf(o0, zv, acc[5]);
    acc[6] = fmaf(o0, z, acc[6]);
    acc[7] = fmaf(o1, zu, acc[7]);
    acc[8] = fmaf(o1, zv, acc[8]);
    acc[9] = fmaf(o1, z, acc[9]);
    acc[10] = fmaf(o2, zu, acc[10]);
    acc[11] = fmaf(o2, zv, acc[11]);
    acc[12] = fmaf(o2, z, acc[12]);
    gz += fmaf(o0, a, fmaf(o1, b, -o2 * c));  // dL/dz = ω·(a, b, −c)  (the term on its own, then added: what the device's packed pair of directions forms)
  }
}

// ---------------------------------------------------------------------------------
// One (candidate, sampled pixel) term of IntrinsicsSoftmin's score (intrinsics_softmin.py:105-121):
//   X = z·K⁻¹[u,v,1] (unproject the later frame's pixel), X' = T·X (fitted pose, later -> earlier),
//   xy = project_camera_space(X', K) (exact ±1e8 / NaN->0 semantics), flow = xy − (u,v),
//   e = |w·(flow_x − gt_x)| + |w·(flow_y − gt_y)|.
// softmin_term_bwd: with g = dL/de, the gradients w.r.t. z, w and T (top 3 rows, gt[12]); K is a
// constant (the candidates are a buffer).
// ---------------------------------------------------------------------------------
struct SoftminTerm {
  float ray[3], x[3];
  Projected pr;
  float dx, dy;  // flow − gt
};

FM_HD float softmin_term(const Mat3& k, const Mat3& kinv, const Pose& t, float u, float v, float z, float gt_x, float gt_y, float w,
                         SoftminTerm& o) {
  ray_dir(kinv, u, v, o.ray);
  o.x[0] = o.ray[0] * z; o.x[1] = o.ray[1] * z; o.x[2] = o.ray[2] * z;
  float xc[3];
  apply_pose(t, o.x, xc);
  o.pr = project_point(xc, k);
  o.dx = (o.pr.u - u) - gt_x;
  o.dy = (o.pr.v - v) - gt_y;
  return fabsf(o.dx * w) + fabsf(o.dy * w);
}

FM_HD float fm_sign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

FM_HD void softmin_term_bwd(const Mat3& k, const Pose& t, const SoftminTerm& o, float w, float g, float& gz, float& gw, float (&gt)[12]) {
  const float sx = fm_sign(o.dx * w), sy = fm_sign(o.dy * w);  // d|x|/dx = sign(x), 0 at 0 (torch.abs)
  gw += g * (sx * o.dx + sy * o.dy);
  float gk[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gxc[3], gx[3];
  project_point_bwd(o.pr, k, g * sx * w, g * sy * w, gk, gxc);
  apply_rot_t(t, gxc, gx);
  gz += gx[0] * o.ray[0] + gx[1] * o.ray[1] + gx[2] * o.ray[2];
  for (int r = 0; r < 3; ++r) {
    gt[r * 4 + 0] += gxc[r] * o.x[0];
    gt[r * 4 + 1] += gxc[r] * o.x[1];
    gt[r * 4 + 2] += gxc[r] * o.x[2];
    gt[r * 4 + 3] += gxc[r];
  }
}

// ---------------------------------------------------------------------------------
// Random subset without replacement: element i of a keyed pseudo-random PERMUTATION of [0, n).
// IntrinsicsSoftmin draws `torch.randperm(h*w)[:P]` every step (intrinsics_softmin.py:90) — a
// full device sort of 921 600 keys at 720p for 8192 samples.  A 4-round Feistel network over
// 2·ceil(bits/2) bits with cycle walking is a bijection on [0, n); evaluating it at 0..P-1 gives P
// distinct pseudo-random indices in pseudo-random order with no sorting at all.
// ---------------------------------------------------------------------------------
FM_HD uint32_t fm_mix32(uint32_t x) {  // lowbias32 (Wellons): a good 32-bit avalanche
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

FM_HD uint64_t permuted_index(uint64_t i, uint64_t n, uint64_t seed) {
  int half = 1;
  while ((1ULL << (2 * half)) < n) ++half;  // domain 2^(2·half) >= n, < 4n
  const uint32_t mask = (uint32_t)((1ULL << half) - 1);
  uint32_t keys[4];
  uint64_t z = seed;
  for (int r = 0; r < 4; ++r) {  // splitmix64 round keys
    z += 0x9e3779b97f4a7c15ULL;
    uint64_t k = z;
    k = (k ^ (k >> 30)) * 0xbf58476d1ce4e5b9ULL;
    k = (k ^ (k >> 27)) * 0x94d049bb133111ebULL;
    keys[r] = (uint32_t)(k ^ (k >> 31));
  }
  uint64_t x = i;
  do {  // cycle walking: re-apply until the value falls inside [0, n)
    uint32_t left = (uint32_t)(x >> half) & mask, right = (uint32_t)x & mask;
    for (int r = 0; r < 4; ++r) {
      const uint32_t next = left ^ (fm_mix32(right ^ keys[r]) & mask);
      left = right;
      right = next;
    }
    x = ((uint64_t)left << half) | right;
  } while (x >= n);
  return x;
}

// ---------------------------------------------------------------------------------
// One Adam update (torch.optim.Adam as the reference configures it,
// model_wrapper_overfit.py:104-105: no amsgrad, not maximising; optional L2 weight decay):
//   g += wd·p;  m = lerp(m, g, 1-β1);  v = β2·v + (1-β2)·g²
//   p -= step_size · m / (√v / √(1-β2ᵗ) + ε),   step_size = lr / (1-β1ᵗ)
// ---------------------------------------------------------------------------------
struct AdamCoef {
  float one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps, weight_decay;
};

// bias corrections in double, exactly as torch.optim.adam._single_tensor_adam
FM_HD AdamCoef adam_coefficients(double step, double lr, double beta1, double beta2, double eps, double weight_decay) {
  const double bc1 = 1.0 - pow(beta1, step);
  const double bc2 = 1.0 - pow(beta2, step);
  AdamCoef c;
  c.one_minus_b1 = (float)(1.0 - beta1);
  c.b2 = (float)beta2;
  c.one_minus_b2 = (float)(1.0 - beta2);
  c.step_size = (float)(lr / bc1);
  c.bc2_sqrt = (float)sqrt(bc2);
  c.eps = (float)eps;
  c.weight_decay = (float)weight_decay;
  return c;
}

FM_HD void adam_update(const AdamCoef& c, float& p, float g, float& m, float& v) {
  if (c.weight_decay != 0.f) g = fmaf(c.weight_decay, p, g);
  m = fmaf(c.one_minus_b1, g - m, m);
  v = fmaf(c.one_minus_b2 * g, g, v * c.b2);
  const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
  p = p - c.step_size * (m / denom);
}

// ---------------------------------------------------------------------------------
// One Procrustes correspondence of frame pair (earlier e, later l), evaluated
// identically in the statistics passes and in the backward scatter.
//   p = xyz_l[idx]                                   (projection.py:226-227)
//   q = bilinear(xyz_e, xy[idx] + bwd_flow[idx])     (projection.py:231-242)
//   w = weights[idx]                                  (projection.py:245-249)
// xyz is read from an explicit surfaces image (surf_* != null) or recomputed from
// depth and K⁻¹.  All image pointers address ONE frame (H·W pixels).
// ---------------------------------------------------------------------------------
struct CorrSrc {
  const float* depth_e;   // (H,W)   earlier frame depth       [depth-sourced]
  const float* depth_l;   // (H,W)   later frame depth         [depth-sourced]
  const float* surf_e;    // (H,W,3) earlier frame surfaces    [surface-sourced]
  const float* surf_l;    // (H,W,3)
  const float* bwd_flow;  // (H,W,2)
  const float* weights;   // (H,W)  correspondence weights, or their logits when weight_sens != 0
  float weight_sens;      // 0: plain weights;  s != 0: w = sigmoid(s · weights[idx])
  int height, width;      //    (backbone_explicit_depth.py:38-41 fused into the gather)
};

struct Corr {
  float p[3], q[3], w;
  int idx;
  Taps taps;
  float ray_p[3];  // later-frame ray at idx   [depth-sourced]
  float z_p;
};

// `tap(row, col, u, v)` returns the EARLIER frame's depth at that pixel and its pixel-centre
// coordinates (global memory + two true divisions, or an LDS-staged window with coordinate tables
// in the dense tiled kernels); only used when the surfaces are depth-sourced.
// The later-frame pixel of a correspondence: flat index and pixel-centre coordinates.
struct PixelRef {
  int idx;
  float u, v;
};
FM_HD PixelRef pixel_ref(int idx, int height, int width) {
  const int row = idx / width, col = idx - row * width;
  return {idx, pixel_center(col, width), pixel_center(row, height)};
}

// sigmoid(x); FAST uses the hardware exp2 / rcp (about 2 ulp) on the device.
template <bool FAST>
FM_HD float fm_sigmoid(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (FAST) return fm_rcp(1.0f + __expf(-x));
#endif
  return 1.0f / (1.0f + expf(-x));
}

template <bool FAST, class DepthTap>
FM_HD Corr corr_load_with(const CorrSrc& s, const Mat3& kinv_e, const Mat3& kinv_l, const PixelRef& px, const DepthTap& tap) {
  Corr c;
  const int idx = px.idx;
  c.idx = idx;
  const float u = px.u, v = px.v;
  const float fx = s.bwd_flow[2 * (size_t)idx], fy = s.bwd_flow[2 * (size_t)idx + 1];
  c.w = s.weights[idx];
  if (s.weight_sens != 0.f) c.w = fm_sigmoid<FAST>(s.weight_sens * c.w);
  c.taps = bilinear_taps(u + fx, v + fy, s.height, s.width);
  c.q[0] = c.q[1] = c.q[2] = 0.f;
  c.z_p = 0.f;
  c.ray_p[0] = c.ray_p[1] = c.ray_p[2] = 0.f;
  if (s.surf_l == nullptr) {
    c.z_p = s.depth_l[idx];
    ray_dir(kinv_l, u, v, c.ray_p);
    c.p[0] = c.ray_p[0] * c.z_p;
    c.p[1] = c.ray_p[1] * c.z_p;
    c.p[2] = c.ray_p[2] * c.z_p;
    for (int k = 0; k < 4; ++k) {
      if (!c.taps.in[k]) continue;
      const int tc = tap_col(c.taps, k), tr = tap_row(c.taps, k);
      float ut, vt;
      const float z = tap(tr, tc, ut, vt);
      float ray[3];
      ray_dir(kinv_e, ut, vt, ray);
      c.q[0] += (ray[0] * z) * c.taps.w[k];
      c.q[1] += (ray[1] * z) * c.taps.w[k];
      c.q[2] += (ray[2] * z) * c.taps.w[k];
    }
  } else {
    const float* sl = s.surf_l + (size_t)idx * 3;
    c.p[0] = sl[0];
    c.p[1] = sl[1];
    c.p[2] = sl[2];
    for (int k = 0; k < 4; ++k) {
      if (!c.taps.in[k]) continue;
      const float* se = s.surf_e + ((size_t)tap_row(c.taps, k) * s.width + tap_col(c.taps, k)) * 3;
      c.q[0] += se[0] * c.taps.w[k];
      c.q[1] += se[1] * c.taps.w[k];
      c.q[2] += se[2] * c.taps.w[k];
    }
  }
  return c;
}

// One-pass Procrustes statistics.  The reference centres the clouds first and then forms
// M = Σ w (q−q̄)(p−p̄)ᵀ (procrustes.py:23-32): two passes over the correspondences, each paying the
// full gather chain.  Here ONE pass accumulates the raw moments about a per-pair reference point s
// (the later-frame point of the middle sample: inside the cloud, so nothing large cancels),
//   W = Σw,  P = Σ w p',  Q = Σ w q',  C = Σ w q' p'ᵀ      (p' = p − s, q' = q − s),
// and moments_finish recovers exactly the reference's quantities in fp64:
//   p − p̄ = p' − p̃,  p̃ = P/(W+ε) − s·ε/(W+ε)   (p̄ uses weights/(Σw+ε), ε = 1e-8),
//   M = C − q̃ Pᵀ − Q p̃ᵀ + W q̃ p̃ᵀ.
constexpr int kMomentCount = 16;

FM_HD void later_point(const CorrSrc& s, const Mat3& kinv_l, int idx, float out[3]) {
  if (s.surf_l == nullptr) {
    const int row = idx / s.width, col = idx - row * s.width;
    float ray[3];
    ray_dir(kinv_l, pixel_center(col, s.width), pixel_center(row, s.height), ray);
    const float z = s.depth_l[idx];
    out[0] = ray[0] * z; out[1] = ray[1] * z; out[2] = ray[2] * z;
  } else {
    out[0] = s.surf_l[(size_t)idx * 3]; out[1] = s.surf_l[(size_t)idx * 3 + 1]; out[2] = s.surf_l[(size_t)idx * 3 + 2];
  }
  for (int a = 0; a < 3; ++a)
    if (!(fabsf(out[a]) <= 3.0e38f)) out[a] = 0.f;  // a non-finite sample must not poison every term
}

FM_HD void moments_add(const Corr& c, const float s[3], float (&acc)[kMomentCount]) {
  const float p0 = c.p[0] - s[0], p1 = c.p[1] - s[1], p2 = c.p[2] - s[2];
  acc[0] += c.w;
  acc[1] = fmaf(c.w, p0, acc[1]);
  acc[2] = fmaf(c.w, p1, acc[2]);
  acc[3] = fmaf(c.w, p2, acc[3]);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float wq = c.w * (c.q[a] - s[a]);
    acc[4 + a] += wq;
    acc[7 + a * 3 + 0] = fmaf(wq, p0, acc[7 + a * 3 + 0]);
    acc[7 + a * 3 + 1] = fmaf(wq, p1, acc[7 + a * 3 + 1]);
    acc[7 + a * 3 + 2] = fmaf(wq, p2, acc[7 + a * 3 + 2]);
  }
}

// In place: raw moments -> [0] Σw, [1..3] Σw·p, [4..6] Σw·q, [7..15] M (the layout pose_solve_one reads).
FM_HD void moments_finish(double* st, const float s[3]) {
  const double w = st[0], inv = 1.0 / (w + 1e-8);
  double pt[3], qt[3], pm[3], qm[3];
  for (int a = 0; a < 3; ++a) {
    pm[a] = st[1 + a];
    qm[a] = st[4 + a];
    pt[a] = pm[a] * inv - (double)s[a] * 1e-8 * inv;
    qt[a] = qm[a] * inv - (double)s[a] * 1e-8 * inv;
  }
  for (int a = 0; a < 3; ++a)
    for (int d = 0; d < 3; ++d) st[7 + a * 3 + d] += -qt[a] * pm[d] - qm[a] * pt[d] + w * qt[a] * pt[d];
  for (int a = 0; a < 3; ++a) {
    st[1 + a] = pm[a] + w * (double)s[a];
    st[4 + a] = qm[a] + w * (double)s[a];
  }
}

FM_HD Corr corr_load(const CorrSrc& s, const Mat3& kinv_e, const Mat3& kinv_l, int idx) {
  return corr_load_with<false>(s, kinv_e, kinv_l, pixel_ref(idx, s.height, s.width), [&](int tr, int tc, float& ut, float& vt) {
    ut = pixel_center(tc, s.width);
    vt = pixel_center(tr, s.height);
    return s.depth_e[tr * s.width + tc];
  });
}

// Per-pair constants of the Procrustes backward (produced by the pose-solve backward).
struct PairGrad {
  float gM[9];      // dL/dM
  float gqbar[3];   // total dL/dq̄ (through t and through the centred vectors)
  float gpbar[3];   // total dL/dp̄
  float pbar[3], qbar[3];
  float dbar;       // gq̄·q̄ + gp̄·p̄
  float inv_wsum;   // 1/(Σw + 1e-8)
};

// dL/dq_j, dL/dp_j, dL/dw_j of one correspondence (derived in DESIGN.md).
FM_HD void corr_backward(const Corr& c, const PairGrad& g, float gq[3], float gp[3], float& gw) {
  float pc[3], qc[3];
  for (int a = 0; a < 3; ++a) {
    pc[a] = c.p[a] - g.pbar[a];
    qc[a] = c.q[a] - g.qbar[a];
  }
  const float wn = c.w * g.inv_wsum;
  float dj = 0.f;
  gw = 0.f;
  for (int a = 0; a < 3; ++a) {
    const float mp = g.gM[a * 3 + 0] * pc[0] + g.gM[a * 3 + 1] * pc[1] + g.gM[a * 3 + 2] * pc[2];  // (gM·pc)_a
    const float mq = g.gM[0 * 3 + a] * qc[0] + g.gM[1 * 3 + a] * qc[1] + g.gM[2 * 3 + a] * qc[2];  // (gMᵀ·qc)_a
    gq[a] = c.w * mp + wn * g.gqbar[a];
    gp[a] = c.w * mq + wn * g.gpbar[a];
    gw += qc[a] * mp;
    dj += g.gqbar[a] * c.q[a] + g.gpbar[a] * c.p[a];
  }
  gw += (dj - g.dbar) * g.inv_wsum;
}

// ---------------------------------------------------------------------------------
// 3x3 helpers in double (pose solve runs one thread per frame pair; cost is nil).
// ---------------------------------------------------------------------------------
FM_HD void mat3_mul(const double* a, const double* b, double* o) {  // o = a b
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}
FM_HD void mat3_mul_tn(const double* a, const double* b, double* o) {  // o = aᵀ b
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[0 * 3 + i] * b[0 * 3 + j] + a[1 * 3 + i] * b[1 * 3 + j] + a[2 * 3 + i] * b[2 * 3 + j];
}
FM_HD void mat3_mul_nt(const double* a, const double* b, double* o) {  // o = a bᵀ
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3 + 0] * b[j * 3 + 0] + a[i * 3 + 1] * b[j * 3 + 1] + a[i * 3 + 2] * b[j * 3 + 2];
}
FM_HD void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// Polar/SVD factorisation used by align_rigid (procrustes.py:35-39):
//   M = Ũ diag(σ̃) Ṽᵀ with Ũ, Ṽ ∈ SO(3), σ̃ = (σ1, σ2, ±σ3); R = Ũ Ṽᵀ equals the
//   reference's U·diag(1,1,sign(det U · det Vᵀ))·Vᵀ.  Computed from the Jacobi
//   eigen-decomposition of MᵀM; the third directions come from cross products so the
//   reflection fix is built in and R only depends on the two dominant singular pairs.
// Outputs are column-major-free: U[i*3+c] is row i, column c.
FM_HD void polar_svd3(const double* m, double* U, double* V, double* sig) {
  double a[9];
  mat3_mul_tn(m, m, a);  // a = MᵀM (symmetric)
  double v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(a[1]) + fabs(a[2]) + fabs(a[5]);
    const double diag = fabs(a[0]) + fabs(a[4]) + fabs(a[8]);
    if (off <= 1e-300 || off <= 1e-17 * diag) break;
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0;
      const int q = pq == 0 ? 1 : 2;
      const double apq = a[p * 3 + q];
      if (apq == 0.0) continue;
      const double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * apq);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      for (int k = 0; k < 3; ++k) {  // a <- a J
        const double akp = a[k * 3 + p], akq = a[k * 3 + q];
        a[k * 3 + p] = c * akp - s * akq;
        a[k * 3 + q] = s * akp + c * akq;
      }
      for (int k = 0; k < 3; ++k) {  // a <- Jᵀ a
        const double apk = a[p * 3 + k], aqk = a[q * 3 + k];
        a[p * 3 + k] = c * apk - s * aqk;
        a[q * 3 + k] = s * apk + c * aqk;
      }
      for (int k = 0; k < 3; ++k) {  // v <- v J
        const double vkp = v[k * 3 + p], vkq = v[k * 3 + q];
        v[k * 3 + p] = c * vkp - s * vkq;
        v[k * 3 + q] = s * vkp + c * vkq;
      }
    }
  }
  // order eigenvalues descending
  int o0 = 0, o1 = 1, o2 = 2;
  double l0 = a[0], l1 = a[4], l2 = a[8];
  if (l0 < l1) { double tl = l0; l0 = l1; l1 = tl; int ti = o0; o0 = o1; o1 = ti; }
  if (l0 < l2) { double tl = l0; l0 = l2; l2 = tl; int ti = o0; o0 = o2; o2 = ti; }
  if (l1 < l2) { double tl = l1; l1 = l2; l2 = tl; int ti = o1; o1 = o2; o2 = ti; }
  double v1[3] = {v[0 * 3 + o0], v[1 * 3 + o0], v[2 * 3 + o0]};
  double v2[3] = {v[0 * 3 + o1], v[1 * 3 + o1], v[2 * 3 + o1]};
  double v3[3];
  cross3(v1, v2, v3);
  const double s1 = sqrt(l0 > 0 ? l0 : 0), s2 = sqrt(l1 > 0 ? l1 : 0);
  double u1[3], u2[3], u3[3];
  if (!(s1 > 1e-150)) {  // M == 0: the reference's svd(0) yields U = V = I  ->  R = I
    for (int i = 0; i < 9; ++i) U[i] = V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    sig[0] = sig[1] = sig[2] = 0.0;
    return;
  }
  for (int i = 0; i < 3; ++i) u1[i] = (m[i * 3 + 0] * v1[0] + m[i * 3 + 1] * v1[1] + m[i * 3 + 2] * v1[2]) / s1;
  double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
  for (int i = 0; i < 3; ++i) u1[i] /= n1;
  for (int i = 0; i < 3; ++i) u2[i] = m[i * 3 + 0] * v2[0] + m[i * 3 + 1] * v2[1] + m[i * 3 + 2] * v2[2];
  double d12 = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
  for (int i = 0; i < 3; ++i) u2[i] -= d12 * u1[i];
  double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
  if (n2 > 1e-12 * s1) {
    for (int i = 0; i < 3; ++i) u2[i] /= n2;
  } else {  // rank-1 cloud: any unit vector orthogonal to u1 (rotation is not unique)
    int k = fabs(u1[0]) < fabs(u1[1]) ? (fabs(u1[0]) < fabs(u1[2]) ? 0 : 2) : (fabs(u1[1]) < fabs(u1[2]) ? 1 : 2);
    double e[3] = {0, 0, 0};
    e[k] = 1.0;
    cross3(u1, e, u2);
    n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    for (int i = 0; i < 3; ++i) u2[i] /= n2;
  }
  cross3(u1, u2, u3);
  for (int i = 0; i < 3; ++i) {
    U[i * 3 + 0] = u1[i];
    U[i * 3 + 1] = u2[i];
    U[i * 3 + 2] = u3[i];
    V[i * 3 + 0] = v1[i];
    V[i * 3 + 1] = v2[i];
    V[i * 3 + 2] = v3[i];
  }
  double mv3[3];
  for (int i = 0; i < 3; ++i) mv3[i] = m[i * 3 + 0] * v3[0] + m[i * 3 + 1] * v3[1] + m[i * 3 + 2] * v3[2];
  sig[0] = s1;
  sig[1] = s2;
  sig[2] = u3[0] * mv3[0] + u3[1] * mv3[1] + u3[2] * mv3[2];  // signed: ±σ3
}

// dL/dM from dL/dR for R = Ũ Ṽᵀ (polar-decomposition differential):
//   C = Ũᵀ G_R Ṽ ;  X_ab = (C_ab − C_ba)/(σ̃_a + σ̃_b), X_aa = 0 ;  G_M = Ũ X Ṽᵀ.
FM_HD void polar_svd3_bwd(const double* U, const double* V, const double* sig, const double* gR, double* gM) {
  double tmp[9], c[9], x[9];
  mat3_mul_tn(U, gR, tmp);
  mat3_mul(tmp, V, c);
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      const double den = sig[a] + sig[b];
      x[a * 3 + b] = (a == b || fabs(den) < 1e-300) ? 0.0 : (c[a * 3 + b] - c[b * 3 + a]) / den;
    }
  mat3_mul(U, x, tmp);
  mat3_mul_nt(tmp, V, gM);
}

// General 3x3 inverse via the adjugate (Tensor.inverse() on K, projection.py:86).
FM_HD void inv3(const float* k, float* o) {
  const double a = k[0], b = k[1], c = k[2], d = k[3], e = k[4], f = k[5], g = k[6], h = k[7], i = k[8];
  const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const double inv_det = 1.0 / (a * A + b * B + c * C);
  o[0] = (float)(A * inv_det);
  o[1] = (float)(-(b * i - c * h) * inv_det);
  o[2] = (float)((b * f - c * e) * inv_det);
  o[3] = (float)(B * inv_det);
  o[4] = (float)((a * i - c * g) * inv_det);
  o[5] = (float)(-(a * f - c * d) * inv_det);
  o[6] = (float)(C * inv_det);
  o[7] = (float)(-(a * h - b * g) * inv_det);
  o[8] = (float)((a * e - b * d) * inv_det);
}

// focal_lengths_to_intrinsics (flowmap/model/intrinsics/common.py:6-20): normalised K of one
// focal length — the reference's two roundings (f·√(hw) in fp32, then a true division).
FM_HD void focal_to_k(float focal, int height, int width, float* k) {
#pragma clang fp contract(off)
  const float scaled = focal * (float)sqrt((double)height * (double)width);
  k[0] = scaled / (float)width, k[1] = 0.f, k[2] = 0.5f;
  k[3] = 0.f, k[4] = scaled / (float)height, k[5] = 0.5f;
  k[6] = 0.f, k[7] = 0.f, k[8] = 1.f;
}

// ... and the part of its backward that one K contributes: d/d focal of (gK00·fx + gK11·fy).
FM_HD double focal_grad_term(const float* g_k, int height, int width) {
  return (double)g_k[0] / (double)width + (double)g_k[4] / (double)height;
}

// General 4x4 inverse in double (Tensor.inverse() on poses, projection.py:46,154,176,288): Gauss-Jordan with partial
// pivoting.  Returns false when singular (the output then holds non-finite values).
// Every index is a compile-time constant once the loops are unrolled: the pivot row is brought up by conditional swaps of
// whole rows (row `col` ends up holding the first row of maximal |a[r][col]|, as a search + one swap would leave it; the
// order of the other rows does not enter the result), so the 4x8 tableau lives in registers.  With a run-time pivot index it
// lived in scratch memory — 272 B per lane, every access a trip to the caches: pose_chain_bwd took 33 us for 150 poses.
FM_HD bool inv4(const double* m, double* o) {
  double a[4][8];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      a[r][c] = m[r * 4 + c];
      a[r][4 + c] = r == c ? 1.0 : 0.0;
    }
  bool ok = true;
#pragma unroll
  for (int col = 0; col < 4; ++col) {
#pragma unroll
    for (int r = col + 1; r < 4; ++r) {
      const bool up = fabs(a[r][col]) > fabs(a[col][col]);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const double x = a[col][c], y = a[r][c];
        a[col][c] = up ? y : x;
        a[r][c] = up ? x : y;
      }
    }
    ok = ok && a[col][col] != 0.0;
    const double ip = 1.0 / a[col][col];
#pragma unroll
    for (int c = 0; c < 8; ++c) a[col][c] *= ip;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r != col) {
        const double fct = a[r][col];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[r][c] -= fct * a[col][c];
      }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) o[r * 4 + c] = a[r][4 + c];
  return ok;
}

FM_HD void mat4_mul(const double* a, const double* b, double* o) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
      o[i * 4 + j] = s;
    }
}
FM_HD void mat4_mul_tn(const double* a, const double* b, double* o) {  // aᵀ b
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a[k * 4 + i] * b[k * 4 + j];
      o[i * 4 + j] = s;
    }
}
FM_HD void mat4_mul_nt(const double* a, const double* b, double* o) {  // a bᵀ
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[j * 4 + k];
      o[i * 4 + j] = s;
    }
}


// ---------------------------------------------------------------------------------
// Dense Procrustes (`num_points: null`: every pixel of every pair is a correspondence,
// config/experiment/ablation_explicit_depth.yaml:11-12) in PIXEL SPACE.
//
// With g = z_l·[u, v, 1] (the later pixel) and h = Σ_taps w_k·z_k·[u_k, v_k, 1] (the bilinear
// sample of the earlier frame, projection.py:231-242), the two points of a correspondence are
//   p = K⁻¹_l·g,   q = K⁻¹_e·h,
// linear maps that are CONSTANT per pair.  So the sums of align_rigid (procrustes.py:23-32) are
// accumulated on (g, h) — no ray, no camera-space point per pixel — and the intrinsics are applied
// once per pair in fp64 (dense_moments_finish).  The backward pass likewise: with the centred
// gc = g − K_l·p̄, hc = h − K_e·q̄ and the per-pair constants of DenseBwd,
//   t = Bm·gc + b0   ( = K⁻ᵀ_e·dL/dq / w )      s = Bmᵀ·hc + a0   ( = K⁻ᵀ_l·dL/dp / w )
//   dL/dw       = hc·t + a0·gc
//   dL/dz_l     = w·(s·[u, v, 1])
//   dL/dz_tap k = w_k·w·(t·[u_k, v_k, 1])
//   dL/dK⁻¹_l   = K_lᵀ·Σ (w·s) ⊗ g          dL/dK⁻¹_e = K_eᵀ·Σ (w·t) ⊗ h
// (corr_backward above, rewritten; ≈60 flops per correspondence instead of ≈250).
// ---------------------------------------------------------------------------------
#ifndef FM_DENSE_TILE_H  // (build-variant experiments: tools/dense_microbench.py)
#define FM_DENSE_TILE_H 32
#define FM_DENSE_TILE_W 64
#endif
constexpr int kDenseTileH = FM_DENSE_TILE_H, kDenseTileW = FM_DENSE_TILE_W;  // tile of one workgroup (and of the static tap lists)

// (i + 0.5) / n as sample_image_grid divides it (projection.py:109), by reciprocal + one Newton step:
// the correctly rounded quotient in all but rare double-rounding cases (then 1 ulp off).  Every dense
// kernel and the plan use THIS function, so their taps agree exactly with each other.
FM_HD float center_fast(int i, float n, float rcp_n) {
  const float a = (float)i + 0.5f;
  const float q = a * rcp_n;
  return fmaf(fmaf(-q, n, a), rcp_n, q);
}

// bilinear_taps for the dense kernels: after the border clamp the north-west tap is always inside, so
// only the east / south range tests remain.
FM_HD Taps dense_taps(float x01, float y01, int h, int w) {
  Taps t;
  const float gx = x01 * 2.f - 1.f, gy = y01 * 2.f - 1.f;
  float ix = ((gx + 1.f) * (float)w - 1.f) * 0.5f;
  float iy = ((gy + 1.f) * (float)h - 1.f) * 0.5f;
  ix = fminf((float)(w - 1), fmaxf(ix, 0.f));
  iy = fminf((float)(h - 1), fmaxf(iy, 0.f));
  const float fx = floorf(ix), fy = floorf(iy);
  t.x0 = (int)fx;
  t.y0 = (int)fy;
  const float ax = ix - fx, ay = iy - fy, bx = (fx + 1.f) - ix, by = (fy + 1.f) - iy;
  t.w[0] = bx * by;
  t.w[1] = ax * by;
  t.w[2] = bx * ay;
  t.w[3] = ax * ay;
  const bool xin1 = t.x0 + 1 < w, yin1 = t.y0 + 1 < h;
  t.in[0] = true;
  t.in[1] = xin1;
  t.in[2] = yin1;
  t.in[3] = xin1 && yin1;
  return t;
}

// h from the four taps (nw, ne, sw, se): values z[k], west / east column coordinates u0 / u1,
// north / south row coordinates v0 / v1.  Taps outside the image contribute nothing.
FM_HD void dense_h(const Taps& t, const float z[4], float u0, float u1, float v0, float v1, float h[3]) {
  const float t0 = t.in[0] ? t.w[0] * z[0] : 0.f, t1 = t.in[1] ? t.w[1] * z[1] : 0.f;
  const float t2 = t.in[2] ? t.w[2] * z[2] : 0.f, t3 = t.in[3] ? t.w[3] * z[3] : 0.f;
  h[0] = u0 * (t0 + t2) + u1 * (t1 + t3);
  h[1] = v0 * (t0 + t1) + v1 * (t2 + t3);
  h[2] = (t0 + t1) + (t2 + t3);
}

// Raw moments about the per-pair shift gs (a point inside both clouds):
//   [0] Σw  [1..3] Σ w g'  [4..6] Σ w h'  [7..15] Σ w h' g'ᵀ      g' = g − gs, h' = h − gs
FM_HD void dense_moments_add(const float g[3], const float h[3], float w, const float gs[3], float (&acc)[kMomentCount]) {
  const float g0 = g[0] - gs[0], g1 = g[1] - gs[1], g2 = g[2] - gs[2];
  acc[0] += w;
  acc[1] = fmaf(w, g0, acc[1]);
  acc[2] = fmaf(w, g1, acc[2]);
  acc[3] = fmaf(w, g2, acc[3]);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float wh = w * (h[a] - gs[a]);
    acc[4 + a] += wh;
    acc[7 + a * 3 + 0] = fmaf(wh, g0, acc[7 + a * 3 + 0]);
    acc[7 + a * 3 + 1] = fmaf(wh, g1, acc[7 + a * 3 + 1]);
    acc[7 + a * 3 + 2] = fmaf(wh, g2, acc[7 + a * 3 + 2]);
  }
}

// The shift of a pair: g of its middle pixel (non-finite depth -> 0, as later_point).
FM_HD void dense_shift(const float* depth_l, int height, int width, float gs[3]) {
  const int row = height / 2, col = width / 2;
  const float z = depth_l[(size_t)row * width + col];
  gs[0] = z * pixel_center(col, width);
  gs[1] = z * pixel_center(row, height);
  gs[2] = z;
  for (int a = 0; a < 3; ++a)
    if (!(fabsf(gs[a]) <= 3.0e38f)) gs[a] = 0.f;
}

// In place: raw pixel-space moments -> [0] Σw, [1..3] Σw·p, [4..6] Σw·q, [7..15] M = Σ w (q−q̄)(p−p̄)ᵀ
// (the layout pose_solve_one reads), all in fp64.
FM_HD void dense_moments_finish(double* st, const float gs[3], const float* kinv_e, const float* kinv_l) {
  const double w = st[0];
  double gsum[3], hsum[3], hg[9];
  for (int a = 0; a < 3; ++a) {
    gsum[a] = st[1 + a] + w * (double)gs[a];  // Σ w g
    hsum[a] = st[4 + a] + w * (double)gs[a];  // Σ w h
  }
  for (int a = 0; a < 3; ++a)
    for (int d = 0; d < 3; ++d)  // Σ w h gᵀ = Σ w (h'+gs)(g'+gs)ᵀ
      hg[a * 3 + d] = st[7 + a * 3 + d] + st[4 + a] * (double)gs[d] + (double)gs[a] * st[1 + d] + w * (double)gs[a] * (double)gs[d];
  double ke[9], kl[9], psum[3], qsum[3], tmp[9], c[9];
  for (int i = 0; i < 9; ++i) {
    ke[i] = kinv_e[i];
    kl[i] = kinv_l[i];
  }
  for (int a = 0; a < 3; ++a) {
    psum[a] = kl[a * 3 + 0] * gsum[0] + kl[a * 3 + 1] * gsum[1] + kl[a * 3 + 2] * gsum[2];
    qsum[a] = ke[a * 3 + 0] * hsum[0] + ke[a * 3 + 1] * hsum[1] + ke[a * 3 + 2] * hsum[2];
  }
  mat3_mul(ke, hg, tmp);
  mat3_mul_nt(tmp, kl, c);  // Σ w q pᵀ
  const double inv = 1.0 / (w + 1e-8);
  for (int a = 0; a < 3; ++a)
    for (int d = 0; d < 3; ++d) {
      const double qb = qsum[a] * inv, pb = psum[d] * inv;
      st[7 + a * 3 + d] = c[a * 3 + d] - qb * psum[d] - qsum[a] * pb + w * qb * pb;
    }
  for (int a = 0; a < 3; ++a) {
    st[1 + a] = psum[a];
    st[4 + a] = qsum[a];
  }
}

struct DenseBwd {
  float bm[9];    // K⁻ᵀ_e·dL/dM·K⁻¹_l
  float a0[3];    // K⁻ᵀ_l·dL/dp̄ / (Σw + 1e-8)
  float b0[3];    // K⁻ᵀ_e·dL/dq̄ / (Σw + 1e-8)
  float gbar[3];  // K_l·p̄
  float hbar[3];  // K_e·q̄
};

FM_HD void inv3d(const double* m, double* o) {
  const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const double inv_det = 1.0 / (a * A + b * B + c * C);
  o[0] = A * inv_det, o[1] = -(b * i - c * h) * inv_det, o[2] = (b * f - c * e) * inv_det;
  o[3] = B * inv_det, o[4] = (a * i - c * g) * inv_det, o[5] = -(a * f - c * d) * inv_det;
  o[6] = C * inv_det, o[7] = -(a * h - b * g) * inv_det, o[8] = (a * e - b * d) * inv_det;
}

// pg: the pair's row of pair_grad (gM 9, dL/dq̄ 3, dL/dp̄ 3, d̄, 1/(Σw+1e-8)); ax: its row of aux (p̄ at 21, q̄ at 24).
// k_e / k_l (out, fp64 3x3): the forward intrinsics of the two frames (inverse of K⁻¹).
FM_HD void dense_bwd_consts(const double* pg, const double* ax, const float* kinv_e, const float* kinv_l, DenseBwd& o, double* k_e,
                            double* k_l) {
  double ke[9], kl[9], tmp[9], bm[9];
  for (int i = 0; i < 9; ++i) {
    ke[i] = kinv_e[i];
    kl[i] = kinv_l[i];
  }
  mat3_mul_tn(ke, pg, tmp);
  mat3_mul(tmp, kl, bm);
  inv3d(ke, k_e);
  inv3d(kl, k_l);
  const double iw = pg[16];
  for (int i = 0; i < 9; ++i) o.bm[i] = (float)bm[i];
  for (int a = 0; a < 3; ++a) {
    o.a0[a] = (float)((kl[0 * 3 + a] * pg[12] + kl[1 * 3 + a] * pg[13] + kl[2 * 3 + a] * pg[14]) * iw);
    o.b0[a] = (float)((ke[0 * 3 + a] * pg[9] + ke[1 * 3 + a] * pg[10] + ke[2 * 3 + a] * pg[11]) * iw);
    o.gbar[a] = (float)(k_l[a * 3 + 0] * ax[21] + k_l[a * 3 + 1] * ax[22] + k_l[a * 3 + 2] * ax[23]);
    o.hbar[a] = (float)(k_e[a * 3 + 0] * ax[24] + k_e[a * 3 + 1] * ax[25] + k_e[a * 3 + 2] * ax[26]);
  }
}

// dL/dK⁻¹ of the two frames of a pair WITHOUT a per-pixel sum.  The per-point gradients of align_rigid's statistics are
//   dL/dq_i = w_i·(G_M·(p_i − p̄) + g_q̄/(Σw+1e-8)),     dL/dp_i = w_i·(G_Mᵀ·(q_i − q̄) + g_p̄/(Σw+1e-8))
// and dL/dK⁻¹_e = Σ dL/dq_i ⊗ h_i, dL/dK⁻¹_l = Σ dL/dp_i ⊗ g_i with h = K_e·q, g = K_l·p.  Both sums are linear in the
// statistics the FORWARD pass already holds (st: Σw, Σw·p, Σw·q, M = Σ w (q−q̄)(p−p̄)ᵀ):
//   Σ dL/dq ⊗ q = G_M·(Mᵀ + δ_p·q̄ᵀ) + g_q̄ ⊗ Σw·q /(Σw+1e-8),     δ_p = Σw·p − p̄·Σw  (= 1e-8·p̄)
//   Σ dL/dp ⊗ p = G_Mᵀ·(M + δ_q·p̄ᵀ) + g_p̄ ⊗ Σw·p /(Σw+1e-8),     δ_q = Σw·q − q̄·Σw
// mapped through K_eᵀ / K_lᵀ on the right.  (The dense backward used to accumulate 18 sums per pixel for this.)
// st: the pair's final statistics; pg: its row of pair_grad; ax: its row of aux; out_e / out_l: 3x3, to be ADDED to kinv_acc.
FM_HD void dense_kinv_grads(const double* st, const double* pg, const double* ax, const double* k_e, const double* k_l, double* out_e,
                            double* out_l) {
  const double w = st[0], iw = pg[16];
  const double* sp = st + 1;
  const double* sq = st + 4;
  const double* m = st + 7;   // [a*3 + d]: a indexes q, d indexes p
  const double* gm = pg;      // same layout
  const double* pbar = ax + 21;
  const double* qbar = ax + 24;
  double de[9], dl[9];
  for (int a = 0; a < 3; ++a)
    for (int c = 0; c < 3; ++c) {
      double e = 0.0, l = 0.0;
      for (int d = 0; d < 3; ++d) {
        e += gm[a * 3 + d] * (m[c * 3 + d] + (sp[d] - pbar[d] * w) * qbar[c]);  // G_M·(Mᵀ + δ_p q̄ᵀ)      [a][c]
        l += gm[d * 3 + a] * (m[d * 3 + c] + (sq[d] - qbar[d] * w) * pbar[c]);  // G_Mᵀ·(M + δ_q p̄ᵀ)     [a][c]
      }
      de[a * 3 + c] = e + iw * pg[9 + a] * sq[c];
      dl[a * 3 + c] = l + iw * pg[12 + a] * sp[c];
    }
  for (int a = 0; a < 3; ++a)
    for (int d = 0; d < 3; ++d) {  // (D·Kᵀ)[a][d] = Σ_c D[a][c]·K[d][c]
      if (out_e) out_e[a * 3 + d] = de[a * 3 + 0] * k_e[d * 3 + 0] + de[a * 3 + 1] * k_e[d * 3 + 1] + de[a * 3 + 2] * k_e[d * 3 + 2];
      if (out_l) out_l[a * 3 + d] = dl[a * 3 + 0] * k_l[d * 3 + 0] + dl[a * 3 + 1] * k_l[d * 3 + 1] + dl[a * 3 + 2] * k_l[d * 3 + 2];
    }
}

// The same from a pair's row of `aux` alone (pose_solve_one keeps Σw, p̄, q̄ and M there): what the sparse fit's backward and
// fm_pose_solve_bwd_kinv use.  out_e / out_l may be null (one role only; k of the absent role is not read).
FM_HD void pair_kinv_grads(const double* pg, const double* ax, const double* k_e, const double* k_l, double* out_e, double* out_l) {
  double st[16];
  const double w = ax[27];
  st[0] = w;
  for (int a = 0; a < 3; ++a) {
    st[1 + a] = ax[21 + a] * (w + 1e-8);  // p̄ = Σw·p / (Σw + 1e-8)
    st[4 + a] = ax[24 + a] * (w + 1e-8);
  }
  for (int k = 0; k < 9; ++k) st[7 + k] = ax[28 + k];
  dense_kinv_grads(st, pg, ax, out_e ? k_e : k_l, out_l ? k_l : k_e, out_e, out_l);
}

// t (needs only g: the earlier-role scatter does not sample the earlier frame)
FM_HD void dense_bwd_t(const DenseBwd& c, const float g[3], float t[3], float gc[3]) {
#pragma unroll
  for (int a = 0; a < 3; ++a) gc[a] = g[a] - c.gbar[a];
#pragma unroll
  for (int a = 0; a < 3; ++a) t[a] = fmaf(c.bm[a * 3 + 0], gc[0], fmaf(c.bm[a * 3 + 1], gc[1], fmaf(c.bm[a * 3 + 2], gc[2], c.b0[a])));
}

// s and dL/dw (need h as well)
FM_HD void dense_bwd_s(const DenseBwd& c, const float h[3], const float t[3], const float gc[3], float s[3], float& gw) {
  float hc[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) hc[a] = h[a] - c.hbar[a];
#pragma unroll
  for (int a = 0; a < 3; ++a) s[a] = fmaf(c.bm[0 * 3 + a], hc[0], fmaf(c.bm[1 * 3 + a], hc[1], fmaf(c.bm[2 * 3 + a], hc[2], c.a0[a])));
  gw = hc[0] * t[0] + hc[1] * t[1] + hc[2] * t[2] + c.a0[0] * gc[0] + c.a0[1] * gc[1] + c.a0[2] * gc[2];
}

}  // namespace fm
