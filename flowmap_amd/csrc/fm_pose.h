// Per-frame / per-pair small-matrix steps of the hot path (pose solve, pose chain,
// relative poses, finalisation of the flow kernel's reductions).  Each runs in ONE
// thread in fp64; kernels are thin wrappers (fm_procrustes.hip, fm_flow.hip) and
// tests/host_sim calls the very same functions on the CPU.
#pragma once
#include <cstring>

#include "fm_math.h"

namespace fm {

constexpr int kFlowAccStride = 20;   // kFlowAcc (13) padded; keeps the C-ABI workspace size
constexpr int kStatStride = 16;      // [0]=Σw [1..3]=Σw·p [4..6]=Σw·q [7..15]=M
constexpr int kAuxStride = 40;       // U(9) V(9) sig(3) pbar(3) qbar(3) wsum(1) M(9): [28..36]
constexpr int kPairGradStride = 20;  // gM(9) gqbar(3) gpbar(3) dbar(1) inv_wsum(1)

// ---- align_rigid, steps 1,4,5 (procrustes.py:23-25,35-51) -----------------------------
//   t_bwd = [R | t]: later-camera -> earlier-camera ("inverse relative transformation",
//   projection.py:190-197);  t_fwd = its rigid inverse [Rᵀ | −Rᵀt].
FM_HD void pose_solve_one(const double* st, float* tb, float* tf, double* ax) {
  const double inv = 1.0 / (st[0] + 1e-8);
  double pbar[3], qbar[3];
  for (int a = 0; a < 3; ++a) {
    pbar[a] = st[1 + a] * inv;
    qbar[a] = st[4 + a] * inv;
  }
  double U[9], V[9], sig[3], R[9];
  polar_svd3(st + 7, U, V, sig);
  mat3_mul_nt(U, V, R);
  double t[3];
  for (int a = 0; a < 3; ++a) t[a] = qbar[a] - (R[a * 3 + 0] * pbar[0] + R[a * 3 + 1] * pbar[1] + R[a * 3 + 2] * pbar[2]);
  for (int a = 0; a < 3; ++a) {
    for (int c = 0; c < 3; ++c) tb[a * 4 + c] = (float)R[a * 3 + c];
    tb[a * 4 + 3] = (float)t[a];
  }
  tb[12] = tb[13] = tb[14] = 0.f;
  tb[15] = 1.f;
  if (tf) {
    for (int a = 0; a < 3; ++a) {
      for (int c = 0; c < 3; ++c) tf[a * 4 + c] = (float)R[c * 3 + a];
      tf[a * 4 + 3] = (float)(-(R[0 * 3 + a] * t[0] + R[1 * 3 + a] * t[1] + R[2 * 3 + a] * t[2]));
    }
    tf[12] = tf[13] = tf[14] = 0.f;
    tf[15] = 1.f;
  }
  for (int k = 0; k < 9; ++k) {
    ax[k] = U[k];
    ax[9 + k] = V[k];
  }
  for (int a = 0; a < 3; ++a) {
    ax[18 + a] = sig[a];
    ax[21 + a] = pbar[a];
    ax[24 + a] = qbar[a];
  }
  ax[27] = st[0];
  for (int k = 0; k < 9; ++k) ax[28 + k] = st[7 + k];  // M: the backward derives dL/dK⁻¹ from the statistics (pair_kinv_grad)
}

// Backward of pose_solve_one.  g_tb / g_tf: dL/dT_bwd, dL/dT_fwd (4x4 row-major, bottom
// rows ignored; either may be null).  Writes gM, total centroid gradients and scalars.
FM_HD void pose_solve_bwd_one(const float* g_tb, const float* g_tf, const float* tb, const double* ax, double* out) {
  double R[9], t[3], gR[9], gt[3];
  for (int a = 0; a < 3; ++a) {
    for (int c = 0; c < 3; ++c) {
      R[a * 3 + c] = tb[a * 4 + c];
      gR[a * 3 + c] = g_tb ? (double)g_tb[a * 4 + c] : 0.0;
    }
    t[a] = tb[a * 4 + 3];
    gt[a] = g_tb ? (double)g_tb[a * 4 + 3] : 0.0;
  }
  if (g_tf) {
    // T_fwd = [Rᵀ | −Rᵀ t]:  gR += G_Rfᵀ − t·G_tfᵀ ;  gt += −R·G_tf
    const double gtf[3] = {g_tf[3], g_tf[7], g_tf[11]};
    for (int a = 0; a < 3; ++a) {
      for (int c = 0; c < 3; ++c) gR[a * 3 + c] += (double)g_tf[c * 4 + a] - t[a] * gtf[c];
      gt[a] -= R[a * 3 + 0] * gtf[0] + R[a * 3 + 1] * gtf[1] + R[a * 3 + 2] * gtf[2];
    }
  }
  const double* pbar = ax + 21;
  const double* qbar = ax + 24;
  const double wsum = ax[27];
  // t = q̄ − R p̄
  double gq[3], gp[3];
  for (int a = 0; a < 3; ++a) {
    gq[a] = gt[a];
    gp[a] = -(R[0 * 3 + a] * gt[0] + R[1 * 3 + a] * gt[1] + R[2 * 3 + a] * gt[2]);
    for (int c = 0; c < 3; ++c) gR[a * 3 + c] -= gt[a] * pbar[c];
  }
  double gM[9];
  polar_svd3_bwd(ax, ax + 9, ax + 18, gR, gM);
  // centred vectors also depend on the centroids: Σ_j w_j (p_j − p̄) = 1e-8·p̄ exactly
  const double eps = 1e-8;
  for (int a = 0; a < 3; ++a) {
    gq[a] -= eps * (gM[a * 3 + 0] * pbar[0] + gM[a * 3 + 1] * pbar[1] + gM[a * 3 + 2] * pbar[2]);
    gp[a] -= eps * (gM[0 * 3 + a] * qbar[0] + gM[1 * 3 + a] * qbar[1] + gM[2 * 3 + a] * qbar[2]);
  }
  for (int k = 0; k < 9; ++k) out[k] = gM[k];
  double dbar = 0;
  for (int a = 0; a < 3; ++a) {
    out[9 + a] = gq[a];
    out[12 + a] = gp[a];
    dbar += gq[a] * qbar[a] + gp[a] * pbar[a];
  }
  out[15] = dbar;
  out[16] = 1.0 / (wsum + 1e-8);
}

// ---- get_extrinsics (projection.py:187-210): E_0 = I, E_k = E_{k-1}·T_{k-1} ------------
FM_HD void pose_chain_fwd_one(const float* rel, int steps, float* e) {
  double cur[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int k = 0; k < 16; ++k) e[k] = (float)cur[k];
  for (int s = 0; s < steps; ++s) {
    double t[16], nxt[16];
    for (int k = 0; k < 16; ++k) t[k] = rel[(size_t)s * 16 + k];
    mat4_mul(cur, t, nxt);
    for (int k = 0; k < 16; ++k) {
      cur[k] = nxt[k];
      e[(size_t)(s + 1) * 16 + k] = (float)nxt[k];
    }
  }
}

FM_HD void pose_chain_bwd_one(const float* rel, const float* e, const float* ge, int steps, float* g_rel) {
  double carry[16];
  for (int k = 0; k < 16; ++k) carry[k] = ge[(size_t)steps * 16 + k];
  for (int s = steps - 1; s >= 0; --s) {
    double prev[16], t[16], gt[16], back[16];
    for (int k = 0; k < 16; ++k) {
      prev[k] = e[(size_t)s * 16 + k];
      t[k] = rel[(size_t)s * 16 + k];
    }
    mat4_mul_tn(prev, carry, gt);  // dL/dT_s = E_sᵀ · G_{s+1}
    mat4_mul_nt(carry, t, back);   // dL/dE_s += G_{s+1} · T_sᵀ
    for (int k = 0; k < 16; ++k) {
      g_rel[(size_t)s * 16 + k] = (float)gt[k];
      carry[k] = back[k] + (double)ge[(size_t)s * 16 + k];
    }
  }
}

// ---- relative poses with a GENERAL 4x4 inverse (projection.py:154,176) -----------------
//   fwd = inv(E1)·E0    bwd = inv(E0)·E1     (e01 points at E0; E1 follows)
FM_HD void relative_pose_fwd_one(const float* e01, float* fwd, float* bwd) {
  double e0[16], e1[16], inv0[16], inv1[16], o[16];
  for (int k = 0; k < 16; ++k) {
    e0[k] = e01[k];
    e1[k] = e01[16 + k];
  }
  inv4(e0, inv0);
  inv4(e1, inv1);
  mat4_mul(inv1, e0, o);
  for (int k = 0; k < 16; ++k) fwd[k] = (float)o[k];
  mat4_mul(inv0, e1, o);
  for (int k = 0; k < 16; ++k) bwd[k] = (float)o[k];
}

FM_HD void relative_pose_bwd_one(const float* e01, const float* g_fwd, const float* g_bwd, double* ge0, double* ge1) {
  double e0[16], e1[16], inv0[16], inv1[16];
  for (int k = 0; k < 16; ++k) {
    e0[k] = e01[k];
    e1[k] = e01[16 + k];
    ge0[k] = ge1[k] = 0.0;
  }
  inv4(e0, inv0);
  inv4(e1, inv1);
  double g[16], tmp[16], tmp2[16];
  if (g_fwd) {  // C = inv(E1)·E0 : dE0 += inv1ᵀ G ; dinv1 = G E0ᵀ ; dE1 += −inv1ᵀ dinv1 inv1ᵀ
    for (int k = 0; k < 16; ++k) g[k] = g_fwd[k];
    mat4_mul_tn(inv1, g, tmp);
    for (int k = 0; k < 16; ++k) ge0[k] += tmp[k];
    mat4_mul_nt(g, e0, tmp);
    mat4_mul_tn(inv1, tmp, tmp2);
    mat4_mul_nt(tmp2, inv1, tmp);
    for (int k = 0; k < 16; ++k) ge1[k] -= tmp[k];
  }
  if (g_bwd) {  // C = inv(E0)·E1
    for (int k = 0; k < 16; ++k) g[k] = g_bwd[k];
    mat4_mul_tn(inv0, g, tmp);
    for (int k = 0; k < 16; ++k) ge1[k] += tmp[k];
    mat4_mul_nt(g, e1, tmp);
    mat4_mul_tn(inv0, tmp, tmp2);
    mat4_mul_nt(tmp2, inv0, tmp);
    for (int k = 0; k < 16; ++k) ge0[k] -= tmp[k];
  }
}

// ---- all-pairs relative poses of a track segment (projection.py:288) ---------------------
//   rel[fs, ft] = inv(E_ft) · E_fs   for one batch element with f frames (ext: (f,4,4)).
FM_HD void allpairs_pose_fwd_one(const float* ext, int fs, int ft, float* out) {
  double es[16], et[16], inv[16], o[16];
  for (int k = 0; k < 16; ++k) {
    es[k] = ext[(size_t)fs * 16 + k];
    et[k] = ext[(size_t)ft * 16 + k];
  }
  inv4(et, inv);
  mat4_mul(inv, es, o);
  for (int k = 0; k < 16; ++k) out[k] = (float)o[k];
}

// dL/dE_frame from dL/drel (f,f,4,4):  as source  Σ_ft inv(E_ft)ᵀ·G[frame,ft]
//                                      as target  −inv(E_fr)ᵀ·(Σ_fs G[fs,frame]·E_fsᵀ)·inv(E_fr)ᵀ
FM_HD void allpairs_pose_bwd_one(const float* ext, const float* g_rel, int f, int frame, float* g_ext) {
  double acc[16], sum_t[16], e[16], inv[16], g[16], tmp[16], tmp2[16];
  for (int k = 0; k < 16; ++k) acc[k] = sum_t[k] = 0.0;
  for (int other = 0; other < f; ++other) {
    for (int k = 0; k < 16; ++k) e[k] = ext[(size_t)other * 16 + k];
    // source role: rel[frame, other] = inv(E_other)·E_frame
    inv4(e, inv);
    for (int k = 0; k < 16; ++k) g[k] = g_rel[((size_t)frame * f + other) * 16 + k];
    mat4_mul_tn(inv, g, tmp);
    for (int k = 0; k < 16; ++k) acc[k] += tmp[k];
    // target role: rel[other, frame] = inv(E_frame)·E_other
    for (int k = 0; k < 16; ++k) g[k] = g_rel[((size_t)other * f + frame) * 16 + k];
    mat4_mul_nt(g, e, tmp);
    for (int k = 0; k < 16; ++k) sum_t[k] += tmp[k];
  }
  for (int k = 0; k < 16; ++k) e[k] = ext[(size_t)frame * 16 + k];
  inv4(e, inv);
  mat4_mul_tn(inv, sum_t, tmp);
  mat4_mul_nt(tmp, inv, tmp2);
  for (int k = 0; k < 16; ++k) g_ext[k] = (float)(acc[k] - tmp2[k]);
}

// dK = −K⁻ᵀ · dKinv · K⁻ᵀ   (backward of Tensor.inverse(), projection.py:86)
FM_HD void kinv_grad_to_k(const double* g, const float* ki, double* gk) {
  double tmp[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0;
      for (int j = 0; j < 3; ++j) s += (double)ki[j * 3 + r] * g[j * 3 + c];
      tmp[r * 3 + c] = s;
    }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0;
      for (int j = 0; j < 3; ++j) s += tmp[r * 3 + j] * (double)ki[c * 3 + j];
      gk[r * 3 + c] = -s;
    }
}

// ---- flow kernel finalisation for one frame (see fm_math.h "One flow residual") ---------
// acc: (B*F, 2, kFlowAccStride) sums indexed by SOURCE frame and direction:
//   [0] Σρ·mask   [1..3] σ = Σω   [4..12] Ω = Σ ω ⊗ z[u,v,1]
// Everything below is per (frame, direction) 3x3 algebra in fp64.
struct FlowDirGrads {
  double s[9];    // S = A·Ω = Σ dL/dX' ⊗ z h
  double gt[3];   // dL/dt = A·σ
  double gkd[6];  // dL/dK_dst rows 0,1 (un-scaled intrinsics)
};

FM_HD void flow_dir_grads(const double* a, const float* pose44, const float* kinv_src, const float* k_dst, float ax, float ay,
                          FlowDirGrads& o) {
  const double* sig = a + 1;
  const double* om = a + 4;
  double kd[6];
  for (int i = 0; i < 3; ++i) {
    kd[i] = (double)k_dst[i] * (double)ax;
    kd[3 + i] = (double)k_dst[3 + i] * (double)ay;
  }
  const double A[9] = {kd[0], kd[3], 0.0, kd[1], kd[4], 0.0, kd[2], kd[5], -1.0};
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) o.s[r * 3 + c] = A[r * 3 + 0] * om[0 * 3 + c] + A[r * 3 + 1] * om[1 * 3 + c] + A[r * 3 + 2] * om[2 * 3 + c];
    o.gt[r] = A[r * 3 + 0] * sig[0] + A[r * 3 + 1] * sig[1] + A[r * 3 + 2] * sig[2];
  }
  // m = R·K⁻¹_src ;  dL/dkd'[a][b] = Σ_c m[b][c]·Ω[a][c] + t[b]·σ[a] ; un-scale rows by (ax, ay)
  double m[9], t[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c)
      m[r * 3 + c] = (double)pose44[r * 4 + 0] * kinv_src[0 * 3 + c] + (double)pose44[r * 4 + 1] * kinv_src[1 * 3 + c] +
                     (double)pose44[r * 4 + 2] * kinv_src[2 * 3 + c];
    t[r] = pose44[r * 4 + 3];
  }
  for (int aa = 0; aa < 2; ++aa)
    for (int bb = 0; bb < 3; ++bb) {
      double v = t[bb] * sig[aa];
      for (int c = 0; c < 3; ++c) v += m[bb * 3 + c] * om[aa * 3 + c];
      o.gkd[aa * 3 + bb] = v * (aa == 0 ? (double)ax : (double)ay);
    }
}

// One of the four roles frame bf plays in the flow loss — 0: source of the forward term (into bf+1), 1: source of the
// backward term (into bf−1), 2: destination of the forward term sourced at bf−1, 3: destination of the backward term sourced
// at bf+1.  A source role writes its pair's dL/dT (g_t_fwd / g_t_bwd row) and contributes −K⁻ᵀ·(Rᵀ·S)·K⁻ᵀ to dL/dK[bf]
// (kinv_grad_to_k is linear, so the two source roles need not be summed first); a destination role contributes rows 0, 1.
// gk9 receives this role's part of dL/dK[bf] (zeros when the role does not exist at the ends of the video).  The device
// kernel gives the four roles to four neighbouring lanes; flow_finalize_frame below (host double) runs them in turn.
FM_HD void flow_finalize_role(const double* acc, const float* k_all, const float* kinv_all, const float* t_fwd, const float* t_bwd, int frames,
                              int bf, int role, float ax, float ay, float* g_t_fwd, float* g_t_bwd, double* gk9) {
  const int f = bf % frames;
  const int b = bf / frames;
  const size_t pair_f = (size_t)b * (frames - 1) + f;
  for (int i = 0; i < 9; ++i) gk9[i] = 0.0;
  const bool needs_next = role == 0 || role == 3;  // roles that involve frame bf+1 / pair_f; the others frame bf−1 / pair_f−1
  if (needs_next ? f >= frames - 1 : f <= 0) return;
  FlowDirGrads d;
  if (role < 2) {
    const float* kinv = kinv_all + (size_t)bf * 9;
    const float* pose = role == 0 ? t_fwd + pair_f * 16 : t_bwd + (pair_f - 1) * 16;
    const float* kdst = k_all + (size_t)(role == 0 ? bf + 1 : bf - 1) * 9;
    flow_dir_grads(acc + ((size_t)bf * 2 + role) * kFlowAccStride, pose, kinv, kdst, ax, ay, d);
    float* g_t44 = role == 0 ? g_t_fwd + pair_f * 16 : g_t_bwd + (pair_f - 1) * 16;
    for (int r = 0; r < 3; ++r) {  // dL/dR = S·K⁻ᵀ ; dL/dt
      for (int c = 0; c < 3; ++c) {
        double v = 0;
        for (int j = 0; j < 3; ++j) v += d.s[r * 3 + j] * (double)kinv[c * 3 + j];
        g_t44[r * 4 + c] = (float)v;
      }
      g_t44[r * 4 + 3] = (float)d.gt[r];
    }
    for (int c = 0; c < 4; ++c) g_t44[12 + c] = 0.f;
    double gkinv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < 3; ++r)  // dL/dK⁻¹ = Rᵀ·S
      for (int c = 0; c < 3; ++c)
        for (int j = 0; j < 3; ++j) gkinv[r * 3 + c] += (double)pose[j * 4 + r] * d.s[j * 3 + c];
    kinv_grad_to_k(gkinv, kinv, gk9);
  } else if (role == 2) {
    flow_dir_grads(acc + ((size_t)(bf - 1) * 2 + 0) * kFlowAccStride, t_fwd + (pair_f - 1) * 16, kinv_all + (size_t)(bf - 1) * 9,
                   k_all + (size_t)bf * 9, ax, ay, d);
    for (int i = 0; i < 6; ++i) gk9[i] = d.gkd[i];
  } else {
    flow_dir_grads(acc + ((size_t)(bf + 1) * 2 + 1) * kFlowAccStride, t_bwd + pair_f * 16, kinv_all + (size_t)(bf + 1) * 9,
                   k_all + (size_t)bf * 9, ax, ay, d);
    for (int i = 0; i < 6; ++i) gk9[i] = d.gkd[i];
  }
}

FM_HD void flow_finalize_frame(const double* acc, const float* k_all, const float* kinv_all, const float* t_fwd, const float* t_bwd,
                               int batch, int frames, int bf, float ax, float ay, float* g_t_fwd, float* g_t_bwd, float* g_k) {
  (void)batch;
  double gk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, part[9];
  // (the order the device kernel's quad reduction adds them in: (0 + 1) + (2 + 3))
  double pair01[9], pair23[9];
  flow_finalize_role(acc, k_all, kinv_all, t_fwd, t_bwd, frames, bf, 0, ax, ay, g_t_fwd, g_t_bwd, pair01);
  flow_finalize_role(acc, k_all, kinv_all, t_fwd, t_bwd, frames, bf, 1, ax, ay, g_t_fwd, g_t_bwd, part);
  for (int i = 0; i < 9; ++i) pair01[i] += part[i];
  flow_finalize_role(acc, k_all, kinv_all, t_fwd, t_bwd, frames, bf, 2, ax, ay, g_t_fwd, g_t_bwd, pair23);
  flow_finalize_role(acc, k_all, kinv_all, t_fwd, t_bwd, frames, bf, 3, ax, ay, g_t_fwd, g_t_bwd, part);
  for (int i = 0; i < 9; ++i) gk[i] = pair01[i] + (pair23[i] + part[i]);
  for (int i = 0; i < 9; ++i) g_k[(size_t)bf * 9 + i] = (float)gk[i];
}

// ---------------------------------------------------------------------------------
// Point-tracking loss (compute_track_flow, projection.py:255-298; LossTracking,
// loss_tracking.py:28-61) — the per-residual, per-source and per-frame steps shared by
// fm_track.hip and the host double.
//
// Work-space per (frame-in-segment, point): ws[kTrackWs] =
//   [0..2] xyz   camera-space point sampled from the source frame's surface
//   [3..5] X_w   = E_fs · [xyz; 1]
//   [6..8] h     = Σ_taps w_k · z_k · [u_k, v_k, 1]   (xyz = K⁻¹ · h: carries dL/dK⁻¹)
// Per target frame, kTrackTgt constants (track_target): with Einv = inv(E_ft)[:3,:] (3x4),
//   au = K_row0 · Einv, av = K_row1 · Einv, c = Einv_row2   =>   u = q·(au·[X_w;1]),
//   v = q·(av·[X_w;1]), q = 1/(c·[X_w;1] + eps)   (the affine form of fm_math.h's flow term).
// Per-frame accumulators (fp64), all UNSCALED (the visible count is only known at the end):
//   acc [kTrackAccStride]  target role: [0..11] S = Σ ω ⊗ [X_w;1] with
//                          ω = (q·w_u, q·w_v, q·(w_u·u + w_v·v)), w = dL/d(u,v);  [18] Σ ρ, [19] count
//   acc2[kTrackAcc2Stride] source role: [0..11] Σ gX_w ⊗ [xyz;1], [12..20] Σ gxyz ⊗ h
// ---------------------------------------------------------------------------------
constexpr int kTrackWs = 9;
constexpr int kTrackTgt = 12;
constexpr int kTrackAccStride = 20;
constexpr int kTrackAcc2Stride = 24;
constexpr int kTrackSums = 14;  // values track_pair_term accumulates: S (12), Σρ, count

FM_HD void track_target(const float* ext_inv44, const float* k33, float* tgt) {
  for (int j = 0; j < 4; ++j) {
    double au = 0, av = 0;
    for (int r = 0; r < 3; ++r) {
      au += (double)k33[r] * ext_inv44[r * 4 + j];
      av += (double)k33[3 + r] * ext_inv44[r * 4 + j];
    }
    tgt[j] = (float)au;
    tgt[4 + j] = (float)av;
    tgt[8 + j] = ext_inv44[8 + j];
  }
}

// The bit pattern of a float as an unsigned integer: for non-negative, non-NaN x and positive b, x < b <=> bits(x) < bits(b); a negative x (sign
// bit) and a NaN compare above every positive b.  One unsigned compare stands for `x >= 0 && x < b`.
FM_HD unsigned track_float_bits(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __float_as_uint(x);
#else
  unsigned u;
  std::memcpy(&u, &x, sizeof(u));
  return u;
#endif
}

// Round 5 — the residual in SCALED image coordinates.  The mapping measures (u − u_gt)·ax, (v − v_gt)·ay (fix_aspect_ratio, mapping.py:9-24):
// with the target's projection rows pre-multiplied, au' = ax·au, av' = ay·av (track_scale_target), and the track position pre-multiplied
// per (target, point), the pair term works on u' = ax·u, v' = ay·v directly — r = (u' − u_gt', v' − v_gt'), dL/du' = coef·m·r_x — and sheds the
// four multiplications by ax / ay of every residual; the in-frame test u ∈ [0, 1) becomes u' ∈ [0, ax), ONE unsigned compare of bit patterns
// per coordinate (u' is formed by an fma with +0, so it is never −0); the Huber map is two `min`s instead of two compares and three selects:
//   c = min(1/n, 1/δ) (1/n = rsq(n²) = +inf at n = 0),  t = n²·c (= n²/δ below the knee, n above),  ρ = t − ½·min(t, δ),  dρ/dr = c·r
// — the same values as the two-branch form (½·n²/δ | n − ½δ; 1/δ | 1/n), bit for bit away from the knee.  The sums come out as S0/ax, S1/ay:
// the caller multiplies them back (track_pair_term below; the pair kernel when it stores a target's sums).
// Per residual in the pair kernel's loop: 49.3 -> 42 VALU instructions (profiles/r05_track_pairs_inner_loop_isa.txt).
FM_HD void track_scale_target(const float (&tg)[kTrackTgt], float ax, float ay, float (&ts)[kTrackTgt]) {
  for (int j = 0; j < 4; ++j) {
    ts[j] = tg[j] * ax;
    ts[4 + j] = tg[4 + j] * ay;
    ts[8 + j] = tg[8 + j];
  }
}

template <int KIND, bool GRAD>
FM_HD void track_pair_term_scaled(const float (&ts)[kTrackTgt], const float xw[3], float gt_xs, float gt_ys, float m, float delta,
                                  float inv_delta, float ax, float ay, float (&a)[kTrackSums], float gxw[3]) {
  const float xu = fmaf(ts[0], xw[0], fmaf(ts[1], xw[1], fmaf(ts[2], xw[2], ts[3])));
  const float xv = fmaf(ts[4], xw[0], fmaf(ts[5], xw[1], fmaf(ts[6], xw[2], ts[7])));
  const float x2 = fmaf(ts[8], xw[0], fmaf(ts[9], xw[1], fmaf(ts[10], xw[2], ts[11])));
  float q = fm_rcp(x2 + kProjEps);
  const bool ok = fabsf(q) <= 3.0e38f;
  q = ok ? q : 0.f;
  const float u = fmaf(xu, q, 0.f), v = fmaf(xv, q, 0.f);  // ax·u, ay·v; (−0) + (+0) = +0
  const bool inside = ok && track_float_bits(u) < track_float_bits(ax) && track_float_bits(v) < track_float_bits(ay);
  m = inside ? m : 0.f;
  const float rx = u - gt_xs, ry = v - gt_ys;
  const float ss = fmaf(rx, rx, ry * ry);
  float rho, coef;  // ρ and dρ/dr = coef·r
  if (KIND == kL2) {
    rho = 0.5f * ss;
    coef = 1.f;
  } else if (KIND == kL1) {
    const float inv_n = ss > 0.f ? fm_rsq(ss) : 0.f;
    rho = ss * inv_n;
    coef = inv_n;
  } else {
    coef = fminf(fm_rsq(ss), inv_delta);
    const float t = ss * coef;
    rho = fmaf(-0.5f, fminf(t, delta), t);
  }
  a[12] = fmaf(rho, m, a[12]);
  a[13] += m;
  if (!GRAD) return;
  const float gc = m * coef;
  const float wu = gc * rx, wv = gc * ry;  // dL/du', dL/dv' (unscaled by the loss normaliser)
  const float o0 = q * wu, o1 = q * wv, o2 = q * fmaf(wu, u, wv * v);
  a[0] = fmaf(o0, xw[0], a[0]);
  a[1] = fmaf(o0, xw[1], a[1]);
  a[2] = fmaf(o0, xw[2], a[2]);
  a[3] += o0;
  a[4] = fmaf(o1, xw[0], a[4]);
  a[5] = fmaf(o1, xw[1], a[5]);
  a[6] = fmaf(o1, xw[2], a[6]);
  a[7] += o1;
  a[8] = fmaf(o2, xw[0], a[8]);
  a[9] = fmaf(o2, xw[1], a[9]);
  a[10] = fmaf(o2, xw[2], a[10]);
  a[11] += o2;
  gxw[0] = fmaf(o0, ts[0], fmaf(o1, ts[4], fmaf(-o2, ts[8], gxw[0])));  // dL/dX_w = ω'·(au', av', −c)
  gxw[1] = fmaf(o0, ts[1], fmaf(o1, ts[5], fmaf(-o2, ts[9], gxw[1])));
  gxw[2] = fmaf(o0, ts[2], fmaf(o1, ts[6], fmaf(-o2, ts[10], gxw[2])));
}

// One (source fs, target ft, point) residual: adds into the target-role sums a[kTrackSums]
// (S in [0..11], Σρ in [12], count in [13]) and the source point's dL/dX_w.  Branch-free:
// hardware rcp / rsq, mapping kind a template parameter, invisibility (projection.py:290-296:
// the target must land inside [0,1)²) a 0/1 factor.  A point exactly on the camera plane
// (Z'+eps == 0) projects to ±1e8 in the reference and is invisible there too.
// (The un-scaled interface the host double calls: scales the target and the track position, runs the term above, scales the sums back.)
template <int KIND, bool GRAD>
FM_HD void track_pair_term(const float (&tg)[kTrackTgt], const float xw[3], float gt_x, float gt_y, float m, float delta,
                           float inv_delta, float ax, float ay, float (&a)[kTrackSums], float gxw[3]) {
  float ts[kTrackTgt], loc[kTrackSums];
  track_scale_target(tg, ax, ay, ts);
  for (int i = 0; i < kTrackSums; ++i) loc[i] = 0.f;
  track_pair_term_scaled<KIND, GRAD>(ts, xw, gt_x * ax, gt_y * ay, m, delta, inv_delta, ax, ay, loc, gxw);
  for (int j = 0; j < 4; ++j) {
    a[j] += loc[j] * ax;
    a[4 + j] += loc[4 + j] * ay;
    a[8 + j] += loc[8 + j];
  }
  a[12] += loc[12];
  a[13] += loc[13];
}

// Source-role step of one point once its dL/dX_w is complete: b[21] = this point's terms of
// acc2, gxyz = dL/dxyz = R_fsᵀ · gX_w (scattered into dL/ddepth by fm_track_scatter).
FM_HD void track_source_term(const Pose& e_s, const float* ws9, const float gxw[3], float (&b)[21], float gxyz[3]) {
  for (int r = 0; r < 3; ++r) {
    b[r * 4 + 0] = gxw[r] * ws9[0];
    b[r * 4 + 1] = gxw[r] * ws9[1];
    b[r * 4 + 2] = gxw[r] * ws9[2];
    b[r * 4 + 3] = gxw[r];
  }
  apply_rot_t(e_s, gxw, gxyz);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) b[12 + r * 3 + c] = gxyz[r] * ws9[6 + c];
}

// dL/dE (4x4) and dL/dK (3x3) of one frame from its two accumulators; sc = weight/count·upstream.
FM_HD void track_frame_grads(const double* a, const double* b, double sc, const float* ext_inv, const float* k, const float* kinv,
                             float* g_ext, float* g_k) {
  // target role.  au = K_row0·Einv, av = K_row1·Einv, c = Einv_row2 and S_i = Σ ω_i [X_w;1]:
  //   dL/dEinv[r][j] = K[0][r]·S0[j] + K[1][r]·S1[j] − (r == 2)·S2[j];   dL/dK[i][r] = Einv[r,:]·S_i
  // then dE = −Einvᵀ·dEinv·Einvᵀ (E⁻¹ of a general 4x4, bottom row of dEinv zero).
  double ginv[16], inv[16], tmp[16], tmp2[16], gk[9];
  for (int i = 0; i < 16; ++i) inv[i] = ext_inv[i];
  for (int r = 0; r < 3; ++r)
    for (int j = 0; j < 4; ++j)
      ginv[r * 4 + j] = ((double)k[r] * a[j] + (double)k[3 + r] * a[4 + j] - (r == 2 ? a[8 + j] : 0.0)) * sc;
  for (int i = 12; i < 16; ++i) ginv[i] = 0.0;
  mat4_mul_tn(inv, ginv, tmp);
  mat4_mul_nt(tmp, inv, tmp2);
  for (int i = 0; i < 16; ++i) g_ext[i] = (float)(-tmp2[i] + (i < 12 ? b[i] * sc : 0.0));
  // intrinsics: source role through K⁻¹ (rows of b[12..20]) + destination role (rows 0,1)
  kinv_grad_to_k(b + 12, kinv, gk);
  for (int i = 0; i < 9; ++i) gk[i] *= sc;
  for (int i = 0; i < 2; ++i)
    for (int r = 0; r < 3; ++r) {
      double d = 0;
      for (int j = 0; j < 4; ++j) d += inv[r * 4 + j] * a[i * 4 + j];
      gk[i * 3 + r] += d * sc;
    }
  for (int i = 0; i < 9; ++i) g_k[i] = (float)gk[i];
}

}  // namespace fm
