// TEST INFRASTRUCTURE ONLY — never shipped, never loaded by the product path.
//
// A serial CPU build of the SAME per-element math the HIP kernels run
// (flowmap_amd/csrc/fm_math.h, fm_pose.h), exported under the same C ABI
// (include/flowmap_hip.h).  tests/ inject it with
// `flowmap_amd._lib.set_library_for_testing()` so the Python host layer and the
// analytic gradients can be checked against the oracle in the GPU-less build
// container before GPU minutes are spent.  Launch geometry, coalescing, wave
// reductions and atomics are what it does NOT cover — the `-m gpu` tests do.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../flowmap_amd/csrc/fm_math.h"
#include "../../flowmap_amd/csrc/fm_pose.h"
#include "../../include/flowmap_hip.h"

using namespace fm;

template <int KIND>
static void sim_flow(const float* depth, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd,
                     const float* flow_fwd, const float* flow_bwd, const float* mask_fwd, const float* mask_bwd, const float* scale,
                     int batch, int frames, int height, int width, float delta, float ax, float ay, float* grad_depth, double* acc) {
  const size_t n = (size_t)height * width;
  const float sc = scale ? scale[0] : 0.f;
  const float inv_delta = KIND == kHuber ? 1.0f / delta : 0.f;
  for (int bf = 0; bf < batch * frames; ++bf) {
    const int f = bf % frames, b = bf / frames;
    const bool has_fwd = f < frames - 1, has_bwd = f > 0;
    Mat3 ki, kd;
    Pose t;
    DirConst df = {}, db = {};
    load_mat3(kinv + (size_t)bf * 9, ki);
    const size_t pair_f = (size_t)b * (frames - 1) + f, pair_b = pair_f - 1;
    if (has_fwd) {
      load_mat3(k + (size_t)(bf + 1) * 9, kd);
      load_pose44(t_fwd + pair_f * 16, t);
      make_dir(t, ki, kd, ax, ay, df);
    }
    if (has_bwd) {
      load_mat3(k + (size_t)(bf - 1) * 9, kd);
      load_pose44(t_bwd + pair_b * 16, t);
      make_dir(t, ki, kd, ax, ay, db);
    }
    double* dst = acc + (size_t)bf * 2 * kFlowAccStride;
    for (int row = 0; row < height; ++row) {
      const float v = pixel_center(row, height), v_ay = v * ay;
      const float rf0 = fmaf(df.a1, v, df.a2), rf1 = fmaf(df.b1, v, df.b2), rf2 = fmaf(df.c1, v, df.c2);
      const float rb0 = fmaf(db.a1, v, db.a2), rb1 = fmaf(db.b1, v, db.b2), rb2 = fmaf(db.c1, v, db.c2);
      for (int col = 0; col < width; ++col) {
        const size_t px = (size_t)row * width + col;
        const float u = pixel_center(col, width);
        const float z = depth[(size_t)bf * n + px];
        const float zu = z * u, zv = z * v, u_ax = u * ax;
        float gz = 0.f;
        float a_f[kFlowAcc] = {0}, a_b[kFlowAcc] = {0};
        if (has_fwd) {
          const float* fl = flow_fwd + (pair_f * n + px) * 2;
          const float m = mask_fwd[pair_f * n + px];
          if (scale) flow_term_fast<KIND, true>(df, rf0, rf1, rf2, z, u, zu, zv, u_ax, v_ay, fl[0], fl[1], m, sc, delta, inv_delta, ax, ay, a_f, gz);
          else flow_term_fast<KIND, false>(df, rf0, rf1, rf2, z, u, zu, zv, u_ax, v_ay, fl[0], fl[1], m, sc, delta, inv_delta, ax, ay, a_f, gz);
        }
        if (has_bwd) {
          const float* fl = flow_bwd + (pair_b * n + px) * 2;
          const float m = mask_bwd[pair_b * n + px];
          if (scale) flow_term_fast<KIND, true>(db, rb0, rb1, rb2, z, u, zu, zv, u_ax, v_ay, fl[0], fl[1], m, sc, delta, inv_delta, ax, ay, a_b, gz);
          else flow_term_fast<KIND, false>(db, rb0, rb1, rb2, z, u, zu, zv, u_ax, v_ay, fl[0], fl[1], m, sc, delta, inv_delta, ax, ay, a_b, gz);
        }
        for (int i = 0; i < kFlowAcc; ++i) {
          dst[i] += a_f[i];
          dst[kFlowAccStride + i] += a_b[i];
        }
        if (scale && grad_depth) grad_depth[(size_t)bf * n + px] = gz;
      }
    }
  }
}

extern "C" {

int fm_flow_loss_fused(const float* depth, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd,
                       const float* flow_fwd, const float* flow_bwd, const float* mask_fwd, const float* mask_bwd,
                       const float* packed, const float* scale, int batch, int frames, int height, int width, int mapping_kind,
                       float delta, float ax, float ay, float* grad_depth, double* acc, int, void*) {
  std::memset(acc, 0, sizeof(double) * (size_t)batch * frames * 2 * kFlowAccStride);
  std::vector<float> un_ff, un_fb, un_mf, un_mb;
  if (packed) {  // expand back to the reference layout the loop below reads
    const size_t n = (size_t)height * width, quads = n / 4, chunks = (quads + 63) / 64;
    const size_t pairs = (size_t)batch * (frames - 1);
    un_ff.assign(pairs * n * 2, 0.f); un_fb.assign(pairs * n * 2, 0.f); un_mf.assign(pairs * n, 0.f); un_mb.assign(pairs * n, 0.f);
    for (int b = 0; b < batch; ++b)
      for (int f = 0; f < frames; ++f)
        for (size_t q = 0; q < quads; ++q) {
          const float* src = packed + ((((size_t)b * frames + f) * chunks + q / 64) * 6 * 64 + q % 64) * 4;
          const size_t pair = (size_t)b * (frames - 1) + f;
          for (int e = 0; e < 4; ++e) {
            if (f < frames - 1) {
              un_ff[(pair * n + q * 4) * 2 + e] = src[0 * 256 + e];
              un_ff[(pair * n + q * 4) * 2 + 4 + e] = src[1 * 256 + e];
              un_mf[pair * n + q * 4 + e] = src[2 * 256 + e];
            }
            if (f > 0) {
              un_fb[((pair - 1) * n + q * 4) * 2 + e] = src[3 * 256 + e];
              un_fb[((pair - 1) * n + q * 4) * 2 + 4 + e] = src[4 * 256 + e];
              un_mb[(pair - 1) * n + q * 4 + e] = src[5 * 256 + e];
            }
          }
        }
    flow_fwd = un_ff.data(); flow_bwd = un_fb.data(); mask_fwd = un_mf.data(); mask_bwd = un_mb.data();
  }
  if (mapping_kind == kHuber)
    sim_flow<kHuber>(depth, k, kinv, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, scale, batch, frames, height, width, delta, ax, ay, grad_depth, acc);
  else if (mapping_kind == kL1)
    sim_flow<kL1>(depth, k, kinv, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, scale, batch, frames, height, width, delta, ax, ay, grad_depth, acc);
  else
    sim_flow<kL2>(depth, k, kinv, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, scale, batch, frames, height, width, delta, ax, ay, grad_depth, acc);
  return 0;
}

int fm_flow_loss_fused_adam(float* depth, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd, const float* flow_fwd,
                            const float* flow_bwd, const float* mask_fwd, const float* mask_bwd, const float* packed, const float* scale,
                            int batch, int frames, int height, int width, int mapping_kind, float delta, float ax, float ay,
                            float* grad_depth, double* acc, int items, float* exp_avg, float* exp_avg_sq, const uint8_t* touched, long step,
                            double lr, double beta1, double beta2, double eps, void* stream) {
  if (!exp_avg || !exp_avg_sq || !touched || !scale || !grad_depth || step < 1 || width % 4 != 0) return 1;
  const size_t total = (size_t)batch * frames * height * width;
  std::vector<float> g(total);  // the full gradient, then the update where the kernel applies it in its own pass
  if (fm_flow_loss_fused(depth, k, kinv, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, packed, scale, batch, frames, height, width,
                         mapping_kind, delta, ax, ay, g.data(), acc, items, stream) != 0)
    return 2;
  const AdamCoef c = adam_coefficients((double)step, lr, beta1, beta2, eps, 0.0);
  for (size_t i = 0; i < total; ++i) {
    if ((touched[i / 4] >> (i % 4)) & 1u) grad_depth[i] = g[i];
    else adam_update(c, depth[i], g[i], exp_avg[i], exp_avg_sq[i]);
  }
  return 0;
}

// The tap exchange, serially: the full gradient first, then every tap in rank order — frame by frame through chunk_base, whose entries are
// checked against the pixel list (the table the device kernel relies on to find a workgroup's run of taps).
int fm_flow_loss_fused_taps(float* depth, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd, const float* flow_fwd,
                            const float* flow_bwd, const float* mask_fwd, const float* mask_bwd, const float* packed, const float* scale,
                            int batch, int frames, int height, int width, int mapping_kind, float delta, float ax, float ay,
                            float* grad_depth, double* acc, int items, const fm_flow_taps* taps, float* exp_avg, float* exp_avg_sq,
                            const uint8_t* touched, long step, double lr, double beta1, double beta2, double eps, void* stream) {
  if (!taps || !taps->chunk_base || !taps->pixel || !scale || !grad_depth || width % 4 != 0 || (taps->grad && !taps->scale)) return 1;
  if (exp_avg && (!exp_avg_sq || !touched || step < 1)) return 1;
  if (!exp_avg && (exp_avg_sq || touched)) return 1;
  const size_t n = (size_t)height * width, total = (size_t)batch * frames * n;
  std::vector<float> g(total);
  if (fm_flow_loss_fused(depth, k, kinv, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, packed, scale, batch, frames, height, width,
                         mapping_kind, delta, ax, ay, g.data(), acc, items, stream) != 0)
    return 2;
  const size_t quads = n / 4, chunks = (quads + 63) / 64;
  const float ts = taps->grad ? taps->scale[0] : 0.f;
  std::vector<size_t> tap_at;  // flat pixel index of every tap, in rank order
  for (size_t bf = 0; bf < (size_t)batch * frames; ++bf) {
    for (size_t c = 0; c < chunks; ++c)
      for (int t = taps->chunk_base[bf * chunks + c]; t < taps->chunk_base[bf * chunks + c + 1]; ++t) {
        if ((size_t)t != tap_at.size() || taps->pixel[t] < 0 || (size_t)taps->pixel[t] / 256 != c) return 2;  // the rank table does not match the pixel list
        tap_at.push_back(bf * n + (size_t)taps->pixel[t]);
      }
  }
  if (taps->grad)
    for (size_t t = 0; t < tap_at.size(); ++t) g[tap_at[t]] += ts * taps->grad[t];
  const AdamCoef c = exp_avg ? adam_coefficients((double)step, lr, beta1, beta2, eps, 0.0) : AdamCoef{};
  for (size_t i = 0; i < total; ++i) {
    if (exp_avg) {
      if ((touched[i / 4] >> (i % 4)) & 1u) grad_depth[i] = g[i];
      else adam_update(c, depth[i], g[i], exp_avg[i], exp_avg_sq[i]);
    } else {
      grad_depth[i] = g[i];
    }
  }
  if (taps->depth)
    for (size_t t = 0; t < tap_at.size(); ++t) {
      if (taps->stale && !exp_avg && std::memcmp(&taps->depth[t], &depth[tap_at[t]], sizeof(float)) != 0) *taps->stale = 1;
      taps->depth[t] = depth[tap_at[t]];
    }
  return 0;
}

int fm_flow_loss_finalize(double* acc, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd,
                          const float* norm, int batch, int frames, float ax, float ay, float* loss, float* g_t_fwd, float* g_t_bwd,
                          float* g_k, void*) {
  double s = 0;
  for (int bf = 0; bf < batch * frames; ++bf) {
    flow_finalize_frame(acc, k, kinv, t_fwd, t_bwd, batch, frames, bf, ax, ay, g_t_fwd, g_t_bwd, g_k);
    s += acc[(size_t)bf * 2 * kFlowAccStride] + acc[(size_t)bf * 2 * kFlowAccStride + kFlowAccStride];
  }
  loss[0] = (float)(s * (double)norm[0]);
  std::memset(acc, 0, sizeof(double) * (size_t)batch * frames * 2 * kFlowAccStride);  // consumed: clean for the next launch
  return 0;
}

int fm_flow_valid_norm(const float* mask_fwd, const float* mask_bwd, long count, float weight, double* vsum, float* norm, void*) {
  double s = 0;
  for (long i = 0; i < count; ++i) s += (double)mask_fwd[i] + (mask_bwd ? (double)mask_bwd[i] : 0.0);
  vsum[0] = s;
  const double veff = s != 0.0 ? s : 1.0;
  norm[0] = (float)((double)weight / veff);
  norm[1] = (float)veff;
  return 0;
}

int fm_flow_pack_inputs(const float* flow_fwd, const float* flow_bwd, const float* mask_fwd, const float* mask_bwd, int batch,
                        int frames, int height, int width, float* packed, void*) {
  if (width % 4 != 0) return 1;
  const size_t n = (size_t)height * width, quads = n / 4, chunks = (quads + 63) / 64;
  std::memset(packed, 0, sizeof(float) * (size_t)batch * frames * chunks * 6 * 64 * 4);
  for (int b = 0; b < batch; ++b)
    for (int f = 0; f < frames; ++f)
      for (size_t q = 0; q < quads; ++q) {
        float* dst = packed + ((((size_t)b * frames + f) * chunks + q / 64) * 6 * 64 + q % 64) * 4;
        const size_t pair = (size_t)b * (frames - 1) + f;
        for (int e = 0; e < 4; ++e) {
          if (f < frames - 1) {
            dst[0 * 256 + e] = flow_fwd[(pair * n + q * 4) * 2 + e];
            dst[1 * 256 + e] = flow_fwd[(pair * n + q * 4) * 2 + 4 + e];
            dst[2 * 256 + e] = mask_fwd[pair * n + q * 4 + e];
          }
          if (f > 0) {
            dst[3 * 256 + e] = flow_bwd[((pair - 1) * n + q * 4) * 2 + e];
            dst[4 * 256 + e] = flow_bwd[((pair - 1) * n + q * 4) * 2 + 4 + e];
            dst[5 * 256 + e] = mask_bwd[(pair - 1) * n + q * 4 + e];
          }
        }
      }
  return 0;
}

static void sim_softmin_point(const float* depth, const float* weights, float sens, const float* flow, const int64_t* indices, int b, long j,
                              int height, int width, float& u, float& v, float& z, float& w, float& gx, float& gy, int& idx) {
  idx = (int)indices[j];
  const size_t n = (size_t)height * width;
  const int row = idx / width, col = idx - row * width;
  u = pixel_center(col, width);
  v = pixel_center(row, height);
  z = depth[((size_t)b * 2 + 1) * n + idx];
  const float raw = weights[(size_t)b * n + idx];
  w = sens != 0.f ? 1.0f / (1.0f + expf(-sens * raw)) : raw;
  gx = flow[((size_t)b * n + idx) * 2];
  gy = flow[((size_t)b * n + idx) * 2 + 1];
}

int fm_softmin_blend_fwd(const double* err, const float* cand_k, int batch, int n, int frames, float* soft, float* k, float* kinv, void*) {
  for (int b = 0; b < batch; ++b) {
    const double* e = err + (size_t)b * n;
    float lo = (float)e[0];
    for (int c = 1; c < n; ++c) lo = std::fmin(lo, (float)e[c]);
    double total = 0.0;
    for (int c = 0; c < n; ++c) total += std::exp(-(((float)e[c] - lo) * 10.f));
    float m[9] = {0};
    for (int c = 0; c < n; ++c) {
      const float sn = (float)(std::exp(-(((float)e[c] - lo) * 10.f)) / total);
      soft[(size_t)b * n + c] = sn;
      for (int i = 0; i < 9; ++i) m[i] += cand_k[(size_t)c * 9 + i] * sn;
    }
    float inv[9];
    inv3(m, inv);
    for (int f = 0; f < frames; ++f)
      for (int i = 0; i < 9; ++i) {
        k[((size_t)b * frames + f) * 9 + i] = m[i];
        if (kinv) kinv[((size_t)b * frames + f) * 9 + i] = inv[i];
      }
  }
  return 0;
}

int fm_softmin_blend_bwd(const float* g_k, const float* soft, const float* cand_k, int batch, int n, int frames, float* g_err, void*) {
  for (int b = 0; b < batch; ++b) {
    double g[9] = {0};
    for (int f = 0; f < frames; ++f)
      for (int i = 0; i < 9; ++i) g[i] += g_k[((size_t)b * frames + f) * 9 + i];
    std::vector<double> gs(n);
    double dot = 0.0;
    for (int c = 0; c < n; ++c) {
      gs[c] = 0.0;
      for (int i = 0; i < 9; ++i) gs[c] += cand_k[(size_t)c * 9 + i] * g[i];
      dot += soft[(size_t)b * n + c] * gs[c];
    }
    for (int c = 0; c < n; ++c) g_err[(size_t)b * n + c] = (float)(-10.0 * soft[(size_t)b * n + c] * (gs[c] - dot));
  }
  return 0;
}

int fm_softmin_score_fwd(const float* depth, const float* weights, float sens, const float* bwd_flow, const int64_t* indices, long points,
                         const float* k, const float* kinv, const float* rel, int batch, int candidates, int height, int width,
                         double* err, void*) {
  for (int bn = 0; bn < batch * candidates; ++bn) {
    const int b = bn / candidates, n = bn % candidates;
    Mat3 km, ki;
    Pose t;
    load_mat3(k + (size_t)n * 9, km);
    load_mat3(kinv + (size_t)n * 9, ki);
    load_pose44(rel + (size_t)bn * 16, t);
    double s = 0;
    for (long j = 0; j < points; ++j) {
      float u, v, z, w, gx, gy;
      int idx;
      sim_softmin_point(depth, weights, sens, bwd_flow, indices, b, j, height, width, u, v, z, w, gx, gy, idx);
      SoftminTerm o;
      s += softmin_term(km, ki, t, u, v, z, gx, gy, w, o);
    }
    err[bn] = s;
  }
  return 0;
}

int fm_softmin_score_bwd(const float* depth, const float* weights, float sens, const float* bwd_flow, const int64_t* indices, long points,
                         const float* k, const float* kinv, const float* rel, int batch, int candidates, int height, int width,
                         const float* g_err, float* g_depth, float* g_weights, double* g_rel_acc, float* g_rel, void*) {
  const size_t npx = (size_t)height * width;
  std::memset(g_rel_acc, 0, sizeof(double) * (size_t)batch * candidates * 12);
  for (int b = 0; b < batch; ++b)
    for (long j = 0; j < points; ++j) {
      float u, v, z, w, gx, gy, gz = 0.f, gw = 0.f;
      int idx;
      sim_softmin_point(depth, weights, sens, bwd_flow, indices, b, j, height, width, u, v, z, w, gx, gy, idx);
      for (int n = 0; n < candidates; ++n) {
        const int bn = b * candidates + n;
        Mat3 km, ki;
        Pose t;
        load_mat3(k + (size_t)n * 9, km);
        load_mat3(kinv + (size_t)n * 9, ki);
        load_pose44(rel + (size_t)bn * 16, t);
        SoftminTerm o;
        softmin_term(km, ki, t, u, v, z, gx, gy, w, o);
        float gt[12] = {};
        softmin_term_bwd(km, t, o, w, g_err[bn], gz, gw, gt);
        for (int e = 0; e < 12; ++e) g_rel_acc[(size_t)bn * 12 + e] += gt[e];
      }
      if (g_depth) g_depth[((size_t)b * 2 + 1) * npx + idx] += gz;
      if (g_weights) g_weights[(size_t)b * npx + idx] += sens != 0.f ? gw * sens * w * (1.f - w) : gw;
    }
  for (int bn = 0; bn < batch * candidates; ++bn) {
    for (int e = 0; e < 12; ++e) g_rel[(size_t)bn * 16 + e] = (float)g_rel_acc[(size_t)bn * 12 + e];
    for (int e = 12; e < 16; ++e) g_rel[(size_t)bn * 16 + e] = 0.f;
  }
  return 0;
}

int fm_random_subset(unsigned long long seed, long n, long count, int64_t* out, void*) {
  for (long i = 0; i < count; ++i) out[i] = (int64_t)permuted_index((uint64_t)i, (uint64_t)n, seed);
  return 0;
}

int fm_random_subset_stateful(unsigned long long* state, long n, long count, int64_t* out, void*) {
  for (long i = 0; i < count; ++i) out[i] = (int64_t)permuted_index((uint64_t)i, (uint64_t)n, state[0]);
  unsigned long long z = state[0] + 0x9e3779b97f4a7c15ULL;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  state[0] = z ^ (z >> 31);
  return 0;
}

int fm_world_points(const float* depth, const float* kinv, const float* ext, const float* colors, int frames, int height, int width,
                    float* out_xyz, float* out_rgb, void*) {
  const size_t n = (size_t)height * width;
  for (int fr = 0; fr < frames; ++fr) {
    Mat3 ki;
    Pose e;
    load_mat3(kinv + (size_t)fr * 9, ki);
    load_pose44(ext + (size_t)fr * 16, e);
    for (size_t i = 0; i < n; ++i) {
      float ray[3], xyz[3], xw[3];
      ray_dir(ki, pixel_center((int)(i % width), width), pixel_center((int)(i / width), height), ray);
      const float z = depth[(size_t)fr * n + i];
      xyz[0] = ray[0] * z; xyz[1] = ray[1] * z; xyz[2] = ray[2] * z;
      apply_pose(e, xyz, xw);
      for (int a = 0; a < 3; ++a) {
        out_xyz[((size_t)fr * n + i) * 3 + a] = xw[a];
        if (colors) out_rgb[((size_t)fr * n + i) * 3 + a] = colors[((size_t)fr * 3 + a) * n + i];
      }
    }
  }
  return 0;
}

int fm_consistency_mask(const float* videos, const float* flow, int batch, int frames, int height, int width, float* mask, void*) {
  const size_t n = (size_t)height * width;
  for (int bp = 0; bp < batch * (frames - 1); ++bp) {
    const int b = bp / (frames - 1), pair = bp % (frames - 1);
    const float* src = videos + ((size_t)b * frames + pair) * 3 * n;
    for (size_t i = 0; i < n; ++i)
      mask[(size_t)bp * n + i] = consistency_mask_at(src, src + 3 * n, height, width, (int)(i / width), (int)(i % width),
                                                     flow[((size_t)bp * n + i) * 2], flow[((size_t)bp * n + i) * 2 + 1]);
  }
  return 0;
}

int fm_flow_postprocess(const float* videos, const float* flow, int batch, int frames, int height, int width, int out_height,
                        int out_width, int reverse, float* out_flow, float* out_mask, void*) {
  const size_t n = (size_t)height * width, on = (size_t)out_height * out_width;
  for (int bp = 0; bp < batch * (frames - 1); ++bp) {
    const int b = bp / (frames - 1), pair = bp % (frames - 1);
    const int fs = reverse ? pair + 1 : pair, ft = reverse ? pair : pair + 1, raw = reverse ? frames - 2 - pair : pair;
    const float* src = videos + ((size_t)b * frames + fs) * 3 * n;
    const float* tgt = videos + ((size_t)b * frames + ft) * 3 * n;
    const float* fl = flow + ((size_t)b * (frames - 1) + raw) * n * 2;
    for (size_t i = 0; i < on; ++i) {
      float of[2], om;
      flow_postprocess_at(src, tgt, fl, height, width, out_height, out_width, (int)(i / out_width), (int)(i % out_width), of, om);
      out_flow[((size_t)bp * on + i) * 2] = of[0];
      out_flow[((size_t)bp * on + i) * 2 + 1] = of[1];
      out_mask[(size_t)bp * on + i] = om;
    }
  }
  return 0;
}

int fm_resize_crop(const float* in, long planes, int h, int w, int rh, int rw, int row0, int col0, int oh, int ow, float* out, void*) {
  for (long p = 0; p < planes; ++p)
    for (int y = 0; y < oh; ++y)
      for (int x = 0; x < ow; ++x) {
        const ResizeTap ty = resize_tap(y + row0, h, rh), tx = resize_tap(x + col0, w, rw);
        const float* s = in + (size_t)p * h * w;
        out[((size_t)p * oh + y) * ow + x] = ty.l0 * (tx.l0 * s[(size_t)ty.i0 * w + tx.i0] + tx.l1 * s[(size_t)ty.i0 * w + tx.i1]) +
                                            ty.l1 * (tx.l0 * s[(size_t)ty.i1 * w + tx.i0] + tx.l1 * s[(size_t)ty.i1 * w + tx.i1]);
      }
  return 0;
}

int fm_adam_step_elements(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const int64_t* elements, long count, long step,
                          double lr, double beta1, double beta2, double eps, double weight_decay, void*) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1 || (count > 0 && !elements)) return 1;
  const AdamCoef c = adam_coefficients((double)step, lr, beta1, beta2, eps, weight_decay);
  for (long i = 0; i < count; ++i) adam_update(c, param[elements[i]], grad[elements[i]], exp_avg[elements[i]], exp_avg_sq[elements[i]]);
  return 0;
}

int fm_fill_zero(float* x, long count, int, void*) {
  for (long i = 0; i < count; ++i) x[i] = 0.f;
  return 0;
}

int fm_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long count, long step, double lr, double beta1,
                 double beta2, double eps, double weight_decay, void*) {
  const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
  AdamCoef c{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)(lr / bc1), (float)std::sqrt(bc2), (float)eps,
             (float)weight_decay};
  for (long i = 0; i < count; ++i) adam_update(c, param[i], grad[i], exp_avg[i], exp_avg_sq[i]);
  return 0;
}

int fm_adam_step_capturable(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long count, const float* step, double lr,
                            double beta1, double beta2, double eps, double weight_decay, void* stream) {
  return fm_adam_step(param, grad, exp_avg, exp_avg_sq, count, (long)step[0], lr, beta1, beta2, eps, weight_decay, stream);
}

int fm_abi_version(void) { return FM_ABI_VERSION; }

int fm_scale_if_needed(float* x, long count, float* y, long count_y, const float* scalar, int* not_one, void*) {
  if (scalar[0] == 1.0f) return 0;
  if (not_one) *not_one = 1;
  for (long i = 0; i < count; ++i) x[i] *= scalar[0];
  for (long i = 0; i < count_y; ++i) y[i] *= scalar[0];
  return 0;
}

static CorrSrc make_src(const float* depth, const float* surfaces, const float* bwd_flow, const float* weights, float sens,
                        size_t pair, int b, int i, int frames, int height, int width, int repeat) {
  const size_t n = (size_t)height * width;
  const int bd = b / repeat;
  const size_t fe = (size_t)bd * frames + i, fl = fe + 1;
  pair = (size_t)bd * (frames - 1) + i;
  CorrSrc s;
  s.depth_e = surfaces ? nullptr : depth + fe * n;
  s.depth_l = surfaces ? nullptr : depth + fl * n;
  s.surf_e = surfaces ? surfaces + fe * n * 3 : nullptr;
  s.surf_l = surfaces ? surfaces + fl * n * 3 : nullptr;
  s.bwd_flow = bwd_flow + pair * n * 2;
  s.weights = weights + pair * n;
  s.weight_sens = sens;
  s.height = height;
  s.width = width;
  return s;
}

static bool dense_mode(const float* depth, const float* surfaces, const int64_t* indices, long points, int repeat, int height, int width) {
  return depth && !surfaces && !indices && repeat == 1 && points == (long)height * width && height <= 65535 && width <= 65535;
}

// One dense correspondence in pixel space (fm_math.h "Dense Procrustes"), taps read from memory.
struct SimDensePixel {
  float g[3], h[3], w, u, v;
  Taps taps;
};
static SimDensePixel sim_dense_pixel(const CorrSrc& src, int row, int col) {
  SimDensePixel o;
  const int idx = row * src.width + col;
  const float fw = (float)src.width, fh = (float)src.height;
  o.u = center_fast(col, fw, 1.0f / fw);
  o.v = center_fast(row, fh, 1.0f / fh);
  float w = src.weights[idx];
  if (src.weight_sens != 0.f) w = fm_sigmoid<false>(src.weight_sens * w);
  o.w = w;
  const float z = src.depth_l[idx];
  o.g[0] = z * o.u; o.g[1] = z * o.v; o.g[2] = z;
  o.taps = dense_taps(o.u + src.bwd_flow[2 * (size_t)idx], o.v + src.bwd_flow[2 * (size_t)idx + 1], src.height, src.width);
  const int x1 = std::min(o.taps.x0 + 1, src.width - 1), y1 = std::min(o.taps.y0 + 1, src.height - 1);
  const float zt[4] = {src.depth_e[(size_t)o.taps.y0 * src.width + o.taps.x0], src.depth_e[(size_t)o.taps.y0 * src.width + x1],
                       src.depth_e[(size_t)y1 * src.width + o.taps.x0], src.depth_e[(size_t)y1 * src.width + x1]};
  dense_h(o.taps, zt, center_fast(o.taps.x0, fw, 1.0f / fw), center_fast(o.taps.x0 + 1, fw, 1.0f / fw), center_fast(o.taps.y0, fh, 1.0f / fh),
          center_fast(o.taps.y0 + 1, fh, 1.0f / fh), o.h);
  return o;
}

int fm_procrustes_stats(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow,
                        const float* weights, float sens, const int64_t* indices, long points, int batch, int repeat, int frames,
                        int height, int width, double* stats, void*) {
  const int pairs = batch * (frames - 1);
  std::memset(stats, 0, sizeof(double) * (size_t)pairs * kStatStride);
  const bool dense = dense_mode(depth, surfaces, indices, points, repeat, height, width);
  for (int pr = 0; pr < pairs; ++pr) {
    const int b = pr / (frames - 1), i = pr % (frames - 1);
    Mat3 ke{}, kl{};
    if (!surfaces) {
      load_mat3(kinv + ((size_t)b * frames + i) * 9, ke);
      load_mat3(kinv + ((size_t)b * frames + i + 1) * 9, kl);
    }
    const CorrSrc src = make_src(depth, surfaces, bwd_flow, weights, sens, pr, b, i, frames, height, width, repeat);
    double* st = stats + (size_t)pr * kStatStride;
    if (dense) {  // pixel-space sums, intrinsics applied once per pair (as the tiled kernels)
      float gs[3];
      dense_shift(src.depth_l, height, width, gs);
      for (long j0 = 0; j0 < points; j0 += 256) {
        float acc[kMomentCount] = {};
        for (long j = j0; j < points && j < j0 + 256; ++j) {
          const SimDensePixel px = sim_dense_pixel(src, (int)(j / width), (int)(j % width));
          dense_moments_add(px.g, px.h, px.w, gs, acc);
        }
        for (int k = 0; k < kMomentCount; ++k) st[k] += acc[k];
      }
      dense_moments_finish(st, gs, kinv + ((size_t)b * frames + i) * 9, kinv + ((size_t)b * frames + i + 1) * 9);
      continue;
    }
    float shift[3];
    const long mid = points / 2;
    later_point(src, kl, indices ? (int)indices[mid] : (int)mid, shift);
    for (long j0 = 0; j0 < points; j0 += 256) {  // fp32 partial sums per 256 points, fp64 across (as the kernel)
      float acc[kMomentCount] = {};
      for (long j = j0; j < points && j < j0 + 256; ++j)
        moments_add(corr_load(src, ke, kl, indices ? (int)indices[j] : (int)j), shift, acc);
      for (int k = 0; k < kMomentCount; ++k) st[k] += acc[k];
    }
    moments_finish(st, shift);
  }
  return 0;
}

int fm_procrustes_dense_tiles(int height, int width, int* tiles) {
  if (!tiles || height < 1 || width < 1) return 1;
  *tiles = ((width + kDenseTileW - 1) / kDenseTileW) * ((height + kDenseTileH - 1) / kDenseTileH);
  return 0;
}

}  // extern "C"
template <class Fn>
static void sim_tap_tiles(const float* flow_pair, int height, int width, int row, int col, const Fn& fn) {
  const float* fl = flow_pair + 2 * ((size_t)row * width + col);
  const float fw = (float)width, fh = (float)height;
  const Taps t = dense_taps(center_fast(col, fw, 1.0f / fw) + fl[0], center_fast(row, fh, 1.0f / fh) + fl[1], height, width);
  const int tiles_x = (width + kDenseTileW - 1) / kDenseTileW;
  const int txa = t.x0 / kDenseTileW, tya = t.y0 / kDenseTileH;
  const int txb = (t.x0 + 1 < width) ? (t.x0 + 1) / kDenseTileW : txa, tyb = (t.y0 + 1 < height) ? (t.y0 + 1) / kDenseTileH : tya;
  fn(tya * tiles_x + txa);
  if (txb != txa) fn(tya * tiles_x + txb);
  if (tyb != tya) {
    fn(tyb * tiles_x + txa);
    if (txb != txa) fn(tyb * tiles_x + txb);
  }
}

extern "C" {
int fm_procrustes_dense_plan(const float* bwd_flow, int batch, int frames, int height, int width, int* counts, const int64_t* first,
                             uint32_t* list, void*) {
  if (!bwd_flow || !counts || (list == nullptr) != (first == nullptr) || height > 65535 || width > 65535) return 1;
  int tiles = 0;
  fm_procrustes_dense_tiles(height, width, &tiles);
  const size_t n = (size_t)height * width;
  for (int pr = 0; pr < batch * (frames - 1); ++pr)
    for (int row = 0; row < height; ++row)
      for (int col = 0; col < width; ++col)
        sim_tap_tiles(bwd_flow + (size_t)pr * n * 2, height, width, row, col, [&](int tile) {
          const int pos = counts[(size_t)pr * tiles + tile]++;
          if (list) list[first[(size_t)pr * tiles + tile] + pos] = ((uint32_t)row << 16) | (uint32_t)col;
        });
  // (row-major enumeration: every tile's list is already in ascending order, as the device's sort leaves it)
  return 0;
}

int fm_procrustes_scatter_dense(const float* depth, const float* kinv, const float* bwd_flow, const float* weights, float sens, int batch,
                                int frames, int height, int width, const double* aux, const double* pair_grad, float* grad_depth,
                                float* grad_weights, const int64_t* first, const uint32_t* list, double*, void*) {
  if (!depth || !kinv || !bwd_flow || !weights || !aux || !pair_grad || ((first == nullptr) != (list == nullptr))) return 1;
  int tiles = 0;
  fm_procrustes_dense_tiles(height, width, &tiles);
  const int tiles_x = (width + kDenseTileW - 1) / kDenseTileW;
  const size_t n = (size_t)height * width;
  for (int pr = 0; pr < batch * (frames - 1); ++pr) {
    const int b = pr / (frames - 1), i = pr % (frames - 1);
    const size_t fe = (size_t)b * frames + i, fl = fe + 1;
    const CorrSrc src = make_src(depth, nullptr, bwd_flow, weights, sens, pr, b, i, frames, height, width, 1);
    DenseBwd c;
    double k_e[9], k_l[9];
    dense_bwd_consts(pair_grad + (size_t)pr * kPairGradStride, aux + (size_t)pr * kAuxStride, kinv + fe * 9, kinv + fl * 9, c, k_e, k_l);
    double a_e[9] = {}, a_l[9] = {};
    // later role: every pixel of the later frame
    for (int row = 0; row < height; ++row)
      for (int col = 0; col < width; ++col) {
        const size_t idx = (size_t)row * width + col;
        const SimDensePixel px = sim_dense_pixel(src, row, col);
        float tv[3], gc[3], sv[3], gw;
        dense_bwd_t(c, px.g, tv, gc);
        dense_bwd_s(c, px.h, tv, gc, sv, gw);
        if (sens != 0.f) gw *= sens * px.w * (1.f - px.w);
        if (grad_weights) grad_weights[(size_t)pr * n + idx] = gw;
        if (grad_depth) grad_depth[fl * n + idx] += px.w * fmaf(sv[0], px.u, fmaf(sv[1], px.v, sv[2]));
        for (int a = 0; a < 3; ++a)
          for (int d = 0; d < 3; ++d) {
            a_e[a * 3 + d] += (double)(px.w * tv[a]) * px.h[d];
            a_l[a * 3 + d] += (double)(px.w * sv[a]) * px.g[d];
          }
      }
    {
      // the device derives dL/dK⁻¹ from the forward statistics (pair_kinv_grads, fm_pose_solve_bwd_kinv); the double also forms the
      // per-pixel sums it replaces and refuses to go on when the two disagree — every CPU test of the dense path checks the algebra
      double ge[9], gl[9], pe[9], pl[9], num = 0.0, den = 0.0;
      pair_kinv_grads(pair_grad + (size_t)pr * kPairGradStride, aux + (size_t)pr * kAuxStride, k_e, k_l, ge, gl);
      for (int r = 0; r < 3; ++r)
        for (int d = 0; d < 3; ++d) {
          pe[r * 3 + d] = k_e[0 * 3 + r] * a_e[0 * 3 + d] + k_e[1 * 3 + r] * a_e[1 * 3 + d] + k_e[2 * 3 + r] * a_e[2 * 3 + d];
          pl[r * 3 + d] = k_l[0 * 3 + r] * a_l[0 * 3 + d] + k_l[1 * 3 + r] * a_l[1 * 3 + d] + k_l[2 * 3 + r] * a_l[2 * 3 + d];
        }
      for (int k = 0; k < 9; ++k) {
        num += (ge[k] - pe[k]) * (ge[k] - pe[k]) + (gl[k] - pl[k]) * (gl[k] - pl[k]);
        den += pe[k] * pe[k] + pl[k] * pl[k];
      }
      if (num > 1e-6 * den + 1e-30) return 3;  // (1e-3 relative: the per-pixel sums are fp32 products)
    }
    if (!grad_depth) continue;
    if (!first) {  // the fused pass: every later pixel adds its four taps where they land (the device: LDS window sums + atomics)
      for (int row = 0; row < height; ++row)
        for (int col = 0; col < width; ++col) {
          const size_t idx = (size_t)row * width + col;
          const float fw = (float)width, fh = (float)height;
          const float u = center_fast(col, fw, 1.0f / fw), v = center_fast(row, fh, 1.0f / fh);
          float w = src.weights[idx];
          if (sens != 0.f) w = fm_sigmoid<false>(sens * w);
          const float z = src.depth_l[idx];
          const float g[3] = {z * u, z * v, z};
          float tv[3], gc[3];
          dense_bwd_t(c, g, tv, gc);
          const Taps tp = dense_taps(u + src.bwd_flow[2 * idx], v + src.bwd_flow[2 * idx + 1], height, width);
          for (int k = 0; k < 4; ++k) {
            if (!tp.in[k]) continue;
            const int y = tp.y0 + (k >> 1), x = tp.x0 + (k & 1);
            grad_depth[fe * n + (size_t)y * width + x] += tp.w[k] * fmaf(w * tv[0], center_fast(x, fw, 1.0f / fw), fmaf(w * tv[1], center_fast(y, fh, 1.0f / fh), w * tv[2]));
          }
        }
      continue;
    }
    // earlier role: per tile of the earlier frame, the listed later pixels (only taps inside the tile count)
    for (int tile = 0; tile < tiles; ++tile) {
      const int tx0 = (tile % tiles_x) * kDenseTileW, ty0 = (tile / tiles_x) * kDenseTileH;
      for (int64_t e = first[(size_t)pr * tiles + tile]; e < first[(size_t)pr * tiles + tile + 1]; ++e) {
        const int row = (int)(list[e] >> 16), col = (int)(list[e] & 0xffffu);
        const size_t idx = (size_t)row * width + col;
        const float fw = (float)width, fh = (float)height;
        const float u = center_fast(col, fw, 1.0f / fw), v = center_fast(row, fh, 1.0f / fh);
        float w = src.weights[idx];
        if (sens != 0.f) w = fm_sigmoid<false>(sens * w);
        const float z = src.depth_l[idx];
        const float g[3] = {z * u, z * v, z};
        float tv[3], gc[3];
        dense_bwd_t(c, g, tv, gc);
        const Taps tp = dense_taps(u + src.bwd_flow[2 * idx], v + src.bwd_flow[2 * idx + 1], height, width);
        for (int k = 0; k < 4; ++k) {
          const int rr = tp.y0 + (k >> 1) - ty0, cc = tp.x0 + (k & 1) - tx0;
          if (!tp.in[k] || rr < 0 || rr >= kDenseTileH || cc < 0 || cc >= kDenseTileW) continue;
          grad_depth[fe * n + (size_t)(ty0 + rr) * width + tx0 + cc] +=
              tp.w[k] * fmaf(w * tv[0], center_fast(tx0 + cc, fw, 1.0f / fw), fmaf(w * tv[1], center_fast(ty0 + rr, fh, 1.0f / fh), w * tv[2]));
        }
      }
    }
  }
  return 0;
}

int fm_procrustes_fit(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow, const float* weights,
                      float sens, const int64_t* indices, long points, int batch, int repeat, int frames, int height, int width,
                      double* stats, float* t_bwd, float* t_fwd, double* aux, void* stream) {
  fm_procrustes_stats(depth, kinv, surfaces, bwd_flow, weights, sens, indices, points, batch, repeat, frames, height, width, stats, stream);
  return fm_pose_solve(stats, batch * (frames - 1), t_bwd, t_fwd, aux, stream);
}

int fm_pose_chain_fwd(const float* rel, int batch, int steps, float* ext, void*);
int fm_procrustes_fit_chain(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow, const float* weights,
                            float sens, const int64_t* indices, long points, int batch, int frames, int height, int width, double* work,
                            float* t_bwd, float* t_fwd, double* aux, float* ext, float* corr_out, const float* tap_records, void* stream) {
  const int pairs = batch * (frames - 1);
  std::vector<double> stats((size_t)pairs * kStatStride);
  (void)work;  // stays zero, as the device leaves it
  if (tap_records) {  // the static tap records must be what the taps of (indices, flows) are: checked here, then not needed
    if (!depth || !indices || points > 4096) return 1;
    const size_t n = (size_t)height * width;
    for (int pr = 0; pr < pairs; ++pr)
      for (long j = 0; j < points; ++j) {
        const int idx = (int)indices[j];
        const PixelRef px = pixel_ref(idx, height, width);
        const float* fl = bwd_flow + ((size_t)pr * n + (size_t)idx) * 2;
        const Taps t = bilinear_taps(px.u + fl[0], px.v + fl[1], height, width);
        const float* r = tap_records + ((size_t)pr * points + (size_t)j) * 8;
        for (int k = 0; k < 4; ++k) {
          int off;
          std::memcpy(&off, r + k, sizeof(int));
          const int want = t.in[k] ? tap_row(t, k) * width + tap_col(t, k) : -1;
          if (off != want || (t.in[k] && r[4 + k] != t.w[k]) || (!t.in[k] && r[4 + k] != 0.f)) return 1;
        }
      }
  }
  if (fm_procrustes_fit(depth, kinv, surfaces, bwd_flow, weights, sens, indices, points, batch, 1, frames, height, width, stats.data(), t_bwd,
                        t_fwd, aux, stream) != 0)
    return 2;
  if (corr_out) {  // the record of every correspondence, as the device kernel leaves it (corr_record_core)
    if (points > 4096 || !tap_records) return 1;
    for (int b = 0; b < batch; ++b)
      for (int i = 0; i < frames - 1; ++i) {
        const size_t pair = (size_t)b * (frames - 1) + i;
        const CorrSrc src = make_src(depth, surfaces, bwd_flow, weights, sens, pair, b, i, frames, height, width, 1);
        Mat3 kinv_e{}, kinv_l{};
        if (!surfaces) {
          load_mat3(kinv + ((size_t)b * frames + i) * 9, kinv_e);
          load_mat3(kinv + ((size_t)b * frames + i + 1) * 9, kinv_l);
        }
        for (long j = 0; j < points; ++j) {
          const Corr c = corr_load(src, kinv_e, kinv_l, indices ? (int)indices[j] : (int)j);
          float* o = corr_out + (pair * (size_t)points + (size_t)j) * 8;
          o[0] = c.q[0], o[1] = c.q[1], o[2] = c.q[2], o[3] = c.p[0], o[4] = c.p[1], o[5] = c.p[2], o[6] = c.w;
          std::memcpy(o + 7, &c.idx, sizeof(float));
        }
      }
  }
  return ext ? fm_pose_chain_fwd(t_bwd, batch, frames - 1, ext, stream) : 0;  // (ext may be NULL: poses only)
}

int fm_procrustes_scatter_plan(const float* bwd_flow, const int64_t* indices, long points, int batch, int frames, int height, int width,
                               int64_t* keys, float* weights, void*) {
  const int64_t n = (int64_t)height * width;
  for (int pr = 0; pr < batch * (frames - 1); ++pr) {
    const int b = pr / (frames - 1), i = pr % (frames - 1);
    for (long j = 0; j < points; ++j) {
      const int idx = indices ? (int)indices[j] : (int)j;
      const PixelRef px = pixel_ref(idx, height, width);
      const float* fl = bwd_flow + ((size_t)pr * n + idx) * 2;
      const Taps t = bilinear_taps(px.u + fl[0], px.v + fl[1], height, width);
      const int64_t fe = (int64_t)b * frames + i;
      const size_t o = ((size_t)pr * points + j) * 5;
      for (int k = 0; k < 4; ++k) {
        keys[o + k] = t.in[k] ? fe * n + (int64_t)tap_row(t, k) * width + tap_col(t, k) : (int64_t)-1;
        weights[o + k] = t.in[k] ? t.w[k] : 0.f;
      }
      keys[o + 4] = (fe + 1) * n + idx;
      weights[o + 4] = 1.f;
    }
  }
  return 0;
}

int fm_sparse_store(const float* values, const int64_t* indices, long points, int groups, long stride, float* out, void*) {
  for (int g = 0; g < groups; ++g)
    for (long j = 0; j < points; ++j) out[(size_t)g * stride + indices[j]] = values[(size_t)g * points + j];
  return 0;
}

int fm_pose_solve(const double* stats, int pairs, float* t_bwd, float* t_fwd, double* aux, void*) {
  for (int pr = 0; pr < pairs; ++pr)
    pose_solve_one(stats + (size_t)pr * kStatStride, t_bwd + (size_t)pr * 16, t_fwd ? t_fwd + (size_t)pr * 16 : nullptr,
                   aux + (size_t)pr * kAuxStride);
  return 0;
}

int fm_pose_solve_bwd(const float* g_t_bwd, const float* g_t_fwd, const float* t_bwd, const double* aux, int pairs,
                      double* pair_grad, double* clear, long clear_count, void*) {
  for (long i = 0; i < clear_count; ++i) clear[i] = 0.0;
  for (int pr = 0; pr < pairs; ++pr)
    pose_solve_bwd_one(g_t_bwd ? g_t_bwd + (size_t)pr * 16 : nullptr, g_t_fwd ? g_t_fwd + (size_t)pr * 16 : nullptr,
                       t_bwd + (size_t)pr * 16, aux + (size_t)pr * kAuxStride, pair_grad + (size_t)pr * kPairGradStride);
  return 0;
}

int fm_pose_solve_bwd_kinv(const float* g_t_bwd, const float* g_t_fwd, const float* t_bwd, const double* aux, const float* kinv, int batch,
                           int frames, double* pair_grad, double* kinv_acc, void*) {
  for (int bf = 0; bf < batch * frames; ++bf) {
    const int b = bf / frames, f = bf % frames;
    double kd[9], kf[9], acc[9] = {};
    for (int k = 0; k < 9; ++k) kd[k] = kinv[(size_t)bf * 9 + k];
    inv3d(kd, kf);
    if (f < frames - 1) {
      const size_t pr = (size_t)b * (frames - 1) + f;
      double* pg = pair_grad + pr * kPairGradStride;
      pose_solve_bwd_one(g_t_bwd ? g_t_bwd + pr * 16 : nullptr, g_t_fwd ? g_t_fwd + pr * 16 : nullptr, t_bwd + pr * 16, aux + pr * kAuxStride, pg);
      double ge[9];
      pair_kinv_grads(pg, aux + pr * kAuxStride, kf, nullptr, ge, nullptr);
      for (int k = 0; k < 9; ++k) acc[k] += ge[k];
    }
    if (f > 0) {
      const size_t pr = (size_t)b * (frames - 1) + f - 1;
      double pg[kPairGradStride], gl[9];
      pose_solve_bwd_one(g_t_bwd ? g_t_bwd + pr * 16 : nullptr, g_t_fwd ? g_t_fwd + pr * 16 : nullptr, t_bwd + pr * 16, aux + pr * kAuxStride, pg);
      pair_kinv_grads(pg, aux + pr * kAuxStride, nullptr, kf, nullptr, gl);
      for (int k = 0; k < 9; ++k) acc[k] += gl[k];
    }
    for (int k = 0; k < 9; ++k) kinv_acc[(size_t)bf * 9 + k] = acc[k];
  }
  return 0;
}

int fm_procrustes_scatter(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow,
                          const float* weights, float sens, const int64_t* indices, long points, int batch, int repeat, int frames,
                          int height, int width, const double* aux, const double* pair_grad, float* grad_depth, float* grad_surfaces,
                          float* grad_weights, double* kinv_acc, float* point_grads, float* point_weight_grads, void*) {
  const int pairs = batch * (frames - 1);
  const size_t n = (size_t)height * width;
  if (point_grads) grad_depth = nullptr;  // planned form: the depth part goes through fm_depth_gather
  for (int pr = 0; pr < pairs; ++pr) {
    const int b = pr / (frames - 1), i = pr % (frames - 1);
    Mat3 ke{}, kl{};
    if (!surfaces) {
      load_mat3(kinv + ((size_t)b * frames + i) * 9, ke);
      load_mat3(kinv + ((size_t)b * frames + i + 1) * 9, kl);
    }
    const CorrSrc src = make_src(depth, surfaces, bwd_flow, weights, sens, pr, b, i, frames, height, width, repeat);
    const double* pg = pair_grad + (size_t)pr * kPairGradStride;
    const double* ax = aux + (size_t)pr * kAuxStride;
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
    const int bd = b / repeat;
    const size_t fe = (size_t)bd * frames + i, fl = fe + 1;
    const size_t fk = (size_t)b * frames + i, dpair = (size_t)bd * (frames - 1) + i;
    double ksum_e[9] = {}, ksum_l[9] = {};  // the per-point sums of dL/dK⁻¹ (what the closed form of pair_kinv_grads replaces)
    for (long j = 0; j < points; ++j) {
      const Corr c = corr_load(src, ke, kl, indices ? (int)indices[j] : (int)j);
      float gq[3], gp[3], gw;
      corr_backward(c, g, gq, gp, gw);
      if (sens != 0.f) gw *= sens * c.w * (1.f - c.w);
      // dense depth-sourced mode: every element is written exactly once and the real library STORES it
      // (the caller does not zero the buffer); every other mode accumulates into zeros
      const bool stores = (depth && !surfaces && !indices && repeat == 1 && points == (long)n) || point_grads;
      if (point_grads)
        for (int a = 0; a < 3; ++a) {
          point_grads[((size_t)pr * points + j) * 6 + a] = gq[a];
          point_grads[((size_t)pr * points + j) * 6 + 3 + a] = gp[a];
        }
      if (point_grads && point_weight_grads) {
        point_weight_grads[(size_t)pr * points + j] = gw;
      } else if (grad_weights) {
        if (stores) grad_weights[dpair * n + c.idx] = gw;
        else grad_weights[dpair * n + c.idx] += gw;
      }
      if (!surfaces) {
        const int row = c.idx / width, col = c.idx - row * width;
        const float u = pixel_center(col, width), v = pixel_center(row, height);
        if (grad_depth) grad_depth[fl * n + c.idx] += gp[0] * c.ray_p[0] + gp[1] * c.ray_p[1] + gp[2] * c.ray_p[2];
        const float zh[3] = {c.z_p * u, c.z_p * v, c.z_p};
        for (int a = 0; a < 3; ++a)
          for (int d = 0; d < 3; ++d) ksum_l[a * 3 + d] += (double)(gp[a] * zh[d]);
        for (int k = 0; k < 4; ++k) {
          if (!c.taps.in[k]) continue;
          const int tc = tap_col(c.taps, k), tr = tap_row(c.taps, k);
          const float ut = pixel_center(tc, width), vt = pixel_center(tr, height);
          const float z = depth[fe * n + (size_t)tr * width + tc];
          float ray[3];
          ray_dir(ke, ut, vt, ray);
          const float wt = c.taps.w[k];
          if (grad_depth) grad_depth[fe * n + (size_t)tr * width + tc] += wt * (gq[0] * ray[0] + gq[1] * ray[1] + gq[2] * ray[2]);
          const float zt[3] = {z * ut * wt, z * vt * wt, z * wt};
          for (int a = 0; a < 3; ++a)
            for (int d = 0; d < 3; ++d) ksum_e[a * 3 + d] += (double)(gq[a] * zt[d]);
        }
      } else if (grad_surfaces) {
        float* gl = grad_surfaces + (fl * n + c.idx) * 3;
        for (int a = 0; a < 3; ++a) gl[a] += gp[a];
        for (int k = 0; k < 4; ++k) {
          if (!c.taps.in[k]) continue;
          float* ge = grad_surfaces + (fe * n + (size_t)tap_row(c.taps, k) * width + tap_col(c.taps, k)) * 3;
          for (int a = 0; a < 3; ++a) ge[a] += gq[a] * c.taps.w[k];
        }
      }
    }
    if (!surfaces) {
      // the device derives dL/dK⁻¹ from the statistics kept in aux (pair_kinv_grads, fm_pose_solve_bwd_kinv); the double forms the
      // per-point sums as well and refuses to go on when the two disagree: every CPU test of a fit checks the algebra
      double kde[9], kdl[9], k_e[9], k_l[9], ge[9], gl[9], num = 0.0, den = 0.0;
      for (int k = 0; k < 9; ++k) {
        kde[k] = kinv[((size_t)b * frames + i) * 9 + k];
        kdl[k] = kinv[((size_t)b * frames + i + 1) * 9 + k];
      }
      inv3d(kde, k_e);
      inv3d(kdl, k_l);
      pair_kinv_grads(pg, ax, k_e, k_l, ge, gl);
      for (int k = 0; k < 9; ++k) {
        num += (ge[k] - ksum_e[k]) * (ge[k] - ksum_e[k]) + (gl[k] - ksum_l[k]) * (gl[k] - ksum_l[k]);
        den += ksum_e[k] * ksum_e[k] + ksum_l[k] * ksum_l[k];
      }
      if (num > 1e-6 * den + 1e-24) return 3;  // (1e-3 relative: the per-point products are fp32)
      if (kinv_acc)
        for (int k = 0; k < 9; ++k) {
          kinv_acc[fk * 9 + k] += ksum_e[k];
          kinv_acc[(fk + 1) * 9 + k] += ksum_l[k];
        }
    }
  }
  return 0;
}

int fm_pose_chain_fwd(const float* rel, int batch, int steps, float* ext, void*) {
  for (int b = 0; b < batch; ++b) pose_chain_fwd_one(rel + (size_t)b * steps * 16, steps, ext + (size_t)b * (steps + 1) * 16);
  return 0;
}

int fm_pose_chain_bwd(const float* rel, const float* ext, const float* g_ext, int batch, int steps, float* g_rel, void*) {
  for (int b = 0; b < batch; ++b)
    pose_chain_bwd_one(rel + (size_t)b * steps * 16, ext + (size_t)b * (steps + 1) * 16, g_ext + (size_t)b * (steps + 1) * 16, steps,
                       g_rel + (size_t)b * steps * 16);
  return 0;
}

int fm_relative_pose_fwd(const float* ext, int batch, int frames, float* fwd, float* bwd, void*) {
  for (int pr = 0; pr < batch * (frames - 1); ++pr) {
    const int b = pr / (frames - 1), i = pr % (frames - 1);
    relative_pose_fwd_one(ext + ((size_t)b * frames + i) * 16, fwd + (size_t)pr * 16, bwd + (size_t)pr * 16);
  }
  return 0;
}

int fm_relative_pose_bwd(const float* ext, const float* g_fwd, const float* g_bwd, int batch, int frames, float* g_ext, void*) {
  std::memset(g_ext, 0, sizeof(float) * (size_t)batch * frames * 16);
  for (int pr = 0; pr < batch * (frames - 1); ++pr) {
    const int b = pr / (frames - 1), i = pr % (frames - 1);
    double ge0[16], ge1[16];
    relative_pose_bwd_one(ext + ((size_t)b * frames + i) * 16, g_fwd ? g_fwd + (size_t)pr * 16 : nullptr,
                          g_bwd ? g_bwd + (size_t)pr * 16 : nullptr, ge0, ge1);
    float* o0 = g_ext + ((size_t)b * frames + i) * 16;
    for (int k = 0; k < 16; ++k) {
      o0[k] += (float)ge0[k];
      o0[16 + k] += (float)ge1[k];
    }
  }
  return 0;
}

int fm_allpairs_pose_fwd(const float* ext, int batch, int f, float* rel, void*) {
  for (int i = 0; i < batch * f * f; ++i) {
    const int b = i / (f * f), r = i % (f * f);
    allpairs_pose_fwd_one(ext + (size_t)b * f * 16, r / f, r % f, rel + (size_t)i * 16);
  }
  return 0;
}

int fm_allpairs_pose_bwd(const float* ext, const float* g_rel, int batch, int f, float* g_ext, void*) {
  for (int i = 0; i < batch * f; ++i) {
    const int b = i / f;
    allpairs_pose_bwd_one(ext + (size_t)b * f * 16, g_rel + (size_t)b * f * f * 16, f, i % f, g_ext + (size_t)i * 16);
  }
  return 0;
}

int fm_focal_intrinsics_fwd(const float* focal, long count, long repeat, int height, int width, float* k, float* kinv, void*) {
  for (long j = 0; j < count * repeat; ++j) {
    focal_to_k(focal[j / repeat], height, width, k + j * 9);
    if (kinv) inv3(k + j * 9, kinv + j * 9);
  }
  return 0;
}

int fm_focal_intrinsics_bwd(const float* grad_k, long count, long repeat, int height, int width, float* grad_focal, void*) {
  for (long i = 0; i < count; ++i) {
    double acc = 0.0;
    for (long r = 0; r < repeat; ++r) acc += focal_grad_term(grad_k + (i * repeat + r) * 9, height, width);
    grad_focal[i] = (float)(acc * (double)(float)sqrt((double)height * (double)width));
  }
  return 0;
}

int fm_intrinsics_inverse(const float* k, int count, float* kinv, void*) {
  for (int i = 0; i < count; ++i) inv3(k + (size_t)i * 9, kinv + (size_t)i * 9);
  return 0;
}

int fm_intrinsics_inverse_bwd(const double* kinv_acc, const float* kinv, int count, float* g_k, int accumulate, void*) {
  for (int i = 0; i < count; ++i) {
    double gk[9];
    kinv_grad_to_k(kinv_acc + (size_t)i * 9, kinv + (size_t)i * 9, gk);
    for (int e = 0; e < 9; ++e) g_k[(size_t)i * 9 + e] = (accumulate ? g_k[(size_t)i * 9 + e] : 0.f) + (float)gk[e];
  }
  return 0;
}


int fm_unproject_fwd(const float* xy, long xy_group_stride, const float* z, const float* kinv, int groups, long points,
                     float* out, void*) {
  for (int g = 0; g < groups; ++g) {
    Mat3 ki;
    load_mat3(kinv + (size_t)g * 9, ki);
    for (long i = 0; i < points; ++i) {
      const float* c = xy + (size_t)g * xy_group_stride + i * 2;
      float ray[3];
      ray_dir(ki, c[0], c[1], ray);
      const float zz = z[(size_t)g * points + i];
      for (int a = 0; a < 3; ++a) out[((size_t)g * points + i) * 3 + a] = ray[a] * zz;
    }
  }
  return 0;
}

int fm_unproject_bwd(const float* xy, long xy_group_stride, const float* z, const float* kinv, const float* g_out, int groups,
                     long points, float* g_z, double* kinv_acc, void*) {
  if (kinv_acc) std::memset(kinv_acc, 0, sizeof(double) * (size_t)groups * 9);
  for (int g = 0; g < groups; ++g) {
    Mat3 ki;
    load_mat3(kinv + (size_t)g * 9, ki);
    for (long i = 0; i < points; ++i) {
      const float* c = xy + (size_t)g * xy_group_stride + i * 2;
      float ray[3];
      ray_dir(ki, c[0], c[1], ray);
      const float zz = z[(size_t)g * points + i];
      const float* go = g_out + ((size_t)g * points + i) * 3;
      if (g_z) g_z[(size_t)g * points + i] = go[0] * ray[0] + go[1] * ray[1] + go[2] * ray[2];
      const float zh[3] = {zz * c[0], zz * c[1], zz};
      if (kinv_acc)
        for (int a = 0; a < 3; ++a)
          for (int d = 0; d < 3; ++d) kinv_acc[(size_t)g * 9 + a * 3 + d] += go[a] * zh[d];
    }
  }
  return 0;
}

int fm_reproject_fwd(const float* xyz, const float* t, const float* k, int groups, long points, float* xy, void*) {
  for (int g = 0; g < groups; ++g) {
    Pose tr;
    Mat3 kk;
    load_pose44(t + (size_t)g * 16, tr);
    load_mat3(k + (size_t)g * 9, kk);
    for (long i = 0; i < points; ++i) {
      const float* p = xyz + ((size_t)g * points + i) * 3;
      float xc[3];
      apply_pose(tr, p, xc);
      const Projected pr = project_point(xc, kk);
      xy[((size_t)g * points + i) * 2] = pr.u;
      xy[((size_t)g * points + i) * 2 + 1] = pr.v;
    }
  }
  return 0;
}

int fm_reproject_bwd(const float* xyz, const float* t, const float* k, const float* g_xy, int groups, long points, float* g_xyz,
                     float* g_t, float* g_k, double* acc, void*) {
  std::memset(acc, 0, sizeof(double) * (size_t)groups * 18);
  for (int g = 0; g < groups; ++g) {
    Pose tr;
    Mat3 kk;
    load_pose44(t + (size_t)g * 16, tr);
    load_mat3(k + (size_t)g * 9, kk);
    double* a = acc + (size_t)g * 18;
    for (long i = 0; i < points; ++i) {
      const float* p = xyz + ((size_t)g * points + i) * 3;
      float xc[3];
      apply_pose(tr, p, xc);
      const Projected pr = project_point(xc, kk);
      const float* go = g_xy + ((size_t)g * points + i) * 2;
      float gk[6] = {0, 0, 0, 0, 0, 0}, gxc[3];
      project_point_bwd(pr, kk, go[0], go[1], gk, gxc);
      for (int r = 0; r < 3; ++r) {
        a[r] += gxc[r];
        for (int d = 0; d < 3; ++d) a[3 + r * 3 + d] += gxc[r] * p[d];
      }
      for (int r = 0; r < 6; ++r) a[12 + r] += gk[r];
      if (g_xyz) {
        float gx[3];
        apply_rot_t(tr, gxc, gx);
        for (int r = 0; r < 3; ++r) g_xyz[((size_t)g * points + i) * 3 + r] = gx[r];
      }
    }
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
  return 0;
}

int fm_bilinear_sample_fwd(const float* img, const float* xy, int groups, int h, int w, int c, long points, float* out, void*) {
  for (int g = 0; g < groups; ++g) {
    const float* im = img + (size_t)g * h * w * c;
    for (long i = 0; i < points; ++i) {
      const float* q = xy + ((size_t)g * points + i) * 2;
      const Taps t = bilinear_taps(q[0], q[1], h, w);
      for (int ch = 0; ch < c; ++ch) {
        float s = 0.f;
        for (int k = 0; k < 4; ++k)
          if (t.in[k]) s += im[((size_t)tap_row(t, k) * w + tap_col(t, k)) * c + ch] * t.w[k];
        out[((size_t)g * points + i) * c + ch] = s;
      }
    }
  }
  return 0;
}

int fm_bilinear_sample_bwd(const float* g_out, const float* xy, int groups, int h, int w, int c, long points, float* g_img, void*) {
  for (int g = 0; g < groups; ++g) {
    float* gi = g_img + (size_t)g * h * w * c;
    for (long i = 0; i < points; ++i) {
      const float* q = xy + ((size_t)g * points + i) * 2;
      const Taps t = bilinear_taps(q[0], q[1], h, w);
      for (int ch = 0; ch < c; ++ch)
        for (int k = 0; k < 4; ++k)
          if (t.in[k]) gi[((size_t)tap_row(t, k) * w + tap_col(t, k)) * c + ch] += g_out[((size_t)g * points + i) * c + ch] * t.w[k];
    }
  }
  return 0;
}

int fm_mapping_fwd(const float* a, const float* b, long count, int kind, float delta, float ax, float ay, float* out, void*) {
  for (long i = 0; i < count; ++i) {
    float dx, dy;
    out[i] = robust_map(kind, delta, aspect_diff(a[2 * i], b[2 * i], ax), aspect_diff(a[2 * i + 1], b[2 * i + 1], ay), dx, dy);
  }
  return 0;
}

int fm_mapping_bwd(const float* a, const float* b, const float* g_out, long count, int kind, float delta, float ax, float ay,
                   float* g_a, float* g_b, void*) {
  for (long i = 0; i < count; ++i) {
    float dx, dy;
    robust_map(kind, delta, aspect_diff(a[2 * i], b[2 * i], ax), aspect_diff(a[2 * i + 1], b[2 * i + 1], ay), dx, dy);
    const float gx = g_out[i] * dx * ax, gy = g_out[i] * dy * ay;
    if (g_a) {
      g_a[2 * i] = gx;
      g_a[2 * i + 1] = gy;
    }
    if (g_b) {
      g_b[2 * i] = -gx;
      g_b[2 * i + 1] = -gy;
    }
  }
  return 0;
}

int fm_align_rigid_stats(const float* p, const float* q, const float* w, int groups, long points, double* stats, void*) {
  std::memset(stats, 0, sizeof(double) * (size_t)groups * kStatStride);
  for (int g = 0; g < groups; ++g) {
    double* st = stats + (size_t)g * kStatStride;
    for (long j = 0; j < points; ++j) {
      const size_t o = (size_t)g * points + j;
      st[0] += w[o];
      for (int a = 0; a < 3; ++a) {
        st[1 + a] += w[o] * p[o * 3 + a];
        st[4 + a] += w[o] * q[o * 3 + a];
      }
    }
    const double inv = 1.0 / (st[0] + 1e-8);
    float pbar[3], qbar[3];
    for (int a = 0; a < 3; ++a) {
      pbar[a] = (float)(st[1 + a] * inv);
      qbar[a] = (float)(st[4 + a] * inv);
    }
    for (long j = 0; j < points; ++j) {
      const size_t o = (size_t)g * points + j;
      for (int a = 0; a < 3; ++a) {
        const float wq = w[o] * (q[o * 3 + a] - qbar[a]);
        for (int d = 0; d < 3; ++d) st[7 + a * 3 + d] += wq * (p[o * 3 + d] - pbar[d]);
      }
    }
  }
  return 0;
}

int fm_align_rigid_bwd(const float* p, const float* q, const float* w, int groups, long points, const double* aux,
                       const double* pair_grad, float* g_p, float* g_q, float* g_w, void*) {
  for (int g = 0; g < groups; ++g) {
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
    for (long j = 0; j < points; ++j) {
      const size_t o = (size_t)g * points + j;
      Corr c;
      for (int a = 0; a < 3; ++a) {
        c.p[a] = p[o * 3 + a];
        c.q[a] = q[o * 3 + a];
      }
      c.w = w[o];
      float gq[3], gp[3], gw;
      corr_backward(c, gr, gq, gp, gw);
      for (int a = 0; a < 3; ++a) {
        if (g_p) g_p[o * 3 + a] = gp[a];
        if (g_q) g_q[o * 3 + a] = gq[a];
      }
      if (g_w) g_w[o] = gw;
    }
  }
  return 0;
}


// ---------------------------------------------------------------- fused tracking ------
int fm_extrinsics_inverse(const float* ext, int count, float* inv, void*) {
  for (int i = 0; i < count; ++i) {
    double a[16], o[16];
    for (int k = 0; k < 16; ++k) a[k] = ext[(size_t)i * 16 + k];
    inv4(a, o);
    for (int k = 0; k < 16; ++k) inv[(size_t)i * 16 + k] = (float)o[k];
  }
  return 0;
}

static int sim_track_points(const float* depth, int depth_frame0, const float* kinv, const float* ext, const float* ext_inv, const float* k,
                            int frames, const float* xy, const uint8_t* vis, const int32_t* seg, const int32_t* blocks, int nblocks,
                            int height, int width, float* ws, uint8_t* flag, float* tgt, const int32_t* tap_slot, const float* tap_depth);

int fm_track_points(const float* depth, int depth_frame0, const float* kinv, const float* ext, const float* ext_inv, const float* k,
                    int frames, const float* xy, const uint8_t* vis, const int32_t* seg, const int32_t* blocks, int nblocks, int,
                    int height, int width, float* ws, uint8_t* flag, float* tgt, void*) {
  return sim_track_points(depth, depth_frame0, kinv, ext, ext_inv, k, frames, xy, vis, seg, blocks, nblocks, height, width, ws, flag, tgt, nullptr,
                          nullptr);
}

// tap_slot / tap_depth: the tap depths come from the compact tap image (slot >= 0), are zero (slot -1) or are read from `depth` (slot <= -2)
static int sim_track_points(const float* depth, int depth_frame0, const float* kinv, const float* ext, const float* ext_inv, const float* k,
                            int frames, const float* xy, const uint8_t* vis, const int32_t* seg, const int32_t* blocks, int nblocks,
                            int height, int width, float* ws, uint8_t* flag, float* tgt, const int32_t* tap_slot, const float* tap_depth) {
  for (int fr = 0; fr < frames; ++fr) track_target(ext_inv + (size_t)fr * 16, k + (size_t)fr * 9, tgt + (size_t)fr * kTrackTgt);
  for (int blk = 0; blk < nblocks; ++blk) {
    const int sg = blocks[blk * 2], fl = blocks[blk * 2 + 1];
    const int start = seg[sg * 4], pc = seg[sg * 4 + 2], off = seg[sg * 4 + 3];
    const int frame = start + fl;
    Mat3 ki;
    Pose e;
    load_mat3(kinv + (size_t)frame * 9, ki);
    load_pose44(ext + (size_t)frame * 16, e);
    const float* d = depth + (size_t)(frame - depth_frame0) * height * width;
    for (int p = 0; p < pc; ++p) {
      const size_t idx = (size_t)off + (size_t)fl * pc + p;
      const float qx = xy[idx * 2], qy = xy[idx * 2 + 1];
      const Taps t = bilinear_taps(qx, qy, height, width);
      float xyz[3] = {0, 0, 0}, hh[3] = {0, 0, 0};
      for (int k = 0; k < 4; ++k) {
        if (!t.in[k]) continue;
        const int tc = tap_col(t, k), tr = tap_row(t, k);
        const float ut = pixel_center(tc, width), vt = pixel_center(tr, height);
        float ray[3];
        ray_dir(ki, ut, vt, ray);
        float z = d[tr * width + tc];
        if (tap_slot) {
          const int sl = tap_slot[idx * 4 + k];
          z = sl < 0 ? 0.f : ((sl & 0x20000000) ? z : tap_depth[sl & 0x1fffffff]);
        }
        for (int a = 0; a < 3; ++a) xyz[a] += (ray[a] * z) * t.w[k];
        hh[0] += z * ut * t.w[k];
        hh[1] += z * vt * t.w[k];
        hh[2] += z * t.w[k];
      }
      float xw[3];
      apply_pose(e, xyz, xw);
      for (int a = 0; a < 3; ++a) {
        ws[idx * kTrackWs + a] = xyz[a];
        ws[idx * kTrackWs + 3 + a] = xw[a];
        ws[idx * kTrackWs + 6 + a] = hh[a];
      }
      const bool inside = qx >= 0.f && qy >= 0.f && qx < 1.f && qy < 1.f;
      flag[idx] = (vis[idx] != 0 && inside) ? 1 : 0;
    }
  }
  return 0;
}

int fm_track_loss_fwd(const float* ws, const uint8_t* flag, const float* xy, const uint8_t* vis, const int32_t* seg,
                      const int32_t* tiles, int ntiles, int, int, const float* ext, const float* tgt, int frames, int, int, int kind,
                      float delta, float ax, float ay, float weight, float*, double* acc, float* loss, float* scale, double* totals,
                      float* gws, double* acc2, void*) {
  std::memset(acc, 0, sizeof(double) * (size_t)frames * kTrackAccStride);
  if (acc2) std::memset(acc2, 0, sizeof(double) * (size_t)frames * kTrackAcc2Stride);
  const float invd = 1.0f / delta;
  for (int tile = 0; tile < ntiles; ++tile) {
    const int sg = tiles[tile * 2], fs0 = tiles[tile * 2 + 1];
    const int start = seg[sg * 4], f = seg[sg * 4 + 1], pc = seg[sg * 4 + 2], off = seg[sg * 4 + 3];
    for (int fs = fs0; fs < fs0 + FM_TRACK_TILE && fs < f; ++fs) {
      Pose e;
      load_pose44(ext + (size_t)(start + fs) * 16, e);
      for (int p = 0; p < pc; ++p) {
        const size_t is = (size_t)off + (size_t)fs * pc + p;
        if (flag[is] == 0) continue;
        const float* w9 = ws + is * kTrackWs;
        const float* xw = w9 + 3;
        float gxw[3] = {0, 0, 0};
        for (int ft = 0; ft < f; ++ft) {
          const size_t it = (size_t)off + (size_t)ft * pc + p;
          if (vis[it] == 0) continue;
          float tg[kTrackTgt];
          for (int i = 0; i < kTrackTgt; ++i) tg[i] = tgt[(size_t)(start + ft) * kTrackTgt + i];
          float a[kTrackSums] = {};
          const float gx = xy[it * 2], gy = xy[it * 2 + 1];
          if (kind == kHuber) gws ? track_pair_term<kHuber, true>(tg, xw, gx, gy, 1.f, delta, invd, ax, ay, a, gxw)
                                  : track_pair_term<kHuber, false>(tg, xw, gx, gy, 1.f, delta, invd, ax, ay, a, gxw);
          else if (kind == kL1) gws ? track_pair_term<kL1, true>(tg, xw, gx, gy, 1.f, delta, invd, ax, ay, a, gxw)
                                    : track_pair_term<kL1, false>(tg, xw, gx, gy, 1.f, delta, invd, ax, ay, a, gxw);
          else gws ? track_pair_term<kL2, true>(tg, xw, gx, gy, 1.f, delta, invd, ax, ay, a, gxw)
                   : track_pair_term<kL2, false>(tg, xw, gx, gy, 1.f, delta, invd, ax, ay, a, gxw);
          double* dst = acc + (size_t)(start + ft) * kTrackAccStride;
          for (int i = 0; i < 12; ++i) dst[i] += a[i];
          dst[18] += a[12];
          dst[19] += a[13];
        }
        if (gws) {
          float b[21], gxyz[3];
          track_source_term(e, w9, gxw, b, gxyz);
          for (int r = 0; r < 3; ++r) gws[is * 3 + r] = gxyz[r];
          for (int i = 0; i < 21; ++i) acc2[(size_t)(start + fs) * kTrackAcc2Stride + i] += b[i];
        }
      }
    }
  }
  double sum = 0, cnt = 0;
  for (int fr = 0; fr < frames; ++fr) {
    sum += acc[(size_t)fr * kTrackAccStride + 18];
    cnt += acc[(size_t)fr * kTrackAccStride + 19];
  }
  if (totals) {
    totals[0] = sum;
    totals[1] = cnt;
  }
  const double den = cnt != 0.0 ? cnt : 1.0;
  loss[0] = (float)((double)weight * sum / den);
  scale[0] = (float)((double)weight / den);
  scale[1] = (float)cnt;
  return 0;
}

int fm_track_loss_fused_fwd(const float* depth, int depth_frame0, int own_first, int own_end, const float* kinv, const float* ext,
                            const float* ext_inv, const float* k, int frames, const float* xy, const uint8_t* vis, const int32_t* seg,
                            const int32_t* tiles, int ntiles, int pmax, int fmax, int height, int width, int kind, float delta, float ax,
                            float ay, float weight, float* ws, uint8_t* flag, float* tgt, float* partial, double* acc, float* loss,
                            float* scale, double* totals, float* gws, double* acc2, void* stream) {
  // the (segment, frame) entries of the listed tiles: owned ones are sampled, the others flagged invisible
  std::vector<int32_t> blocks;
  for (int tile = 0; tile < ntiles; ++tile) {
    const int sg = tiles[tile * 2], fs0 = tiles[tile * 2 + 1];
    const int start = seg[sg * 4], f = seg[sg * 4 + 1], pc = seg[sg * 4 + 2], off = seg[sg * 4 + 3];
    for (int fs = fs0; fs < fs0 + FM_TRACK_TILE && fs < f; ++fs) {
      if (start + fs >= own_first && start + fs < own_end) {
        blocks.push_back(sg);
        blocks.push_back(fs);
      } else {
        for (int p = 0; p < pc; ++p) flag[(size_t)off + (size_t)fs * pc + p] = 0;
      }
    }
  }
  if (fm_track_points(depth, depth_frame0, kinv, ext, ext_inv, k, frames, xy, vis, seg, blocks.data(), (int)(blocks.size() / 2), pmax, height,
                      width, ws, flag, tgt, stream) != 0)
    return 2;
  return fm_track_loss_fwd(ws, flag, xy, vis, seg, tiles, ntiles, pmax, fmax, ext, tgt, frames, height, width, kind, delta, ax, ay, weight,
                           partial, acc, loss, scale, totals, gws, acc2, stream);
}

int fm_track_loss_fused_fwd_taps(const float* depth, const float* kinv, const float* ext, const float* ext_inv, const float* k, int frames,
                                 const float* xy, const uint8_t* vis, const int32_t* seg, const int32_t* tiles, int ntiles, int pmax, int fmax,
                                 int height, int width, int kind, float delta, float ax, float ay, float weight, float* ws, uint8_t* flag,
                                 float* tgt, float* partial, double* acc, float* loss, float* scale, double* totals, float* gws, double* acc2,
                                 const int32_t* tap_slot, const float* tap_depth, const int64_t* plan_pixels, const int32_t* plan_first,
                                 const int32_t* plan_entries, const float* plan_weights, long plan_count, const int32_t* shared_ranks,
                                 long shared_count, float* tap_grad, void* stream) {
  if (tap_depth != nullptr && tap_slot == nullptr) return 1;
  if (shared_ranks != nullptr && (tap_slot == nullptr || shared_count < 0 || shared_count > plan_count)) return 1;
  if (shared_ranks != nullptr && tap_grad != nullptr) {  // the list must name exactly the taps with more than one plan entry, and the slots say so too
    long seen = 0;
    for (long m = 0; m < plan_count; ++m)
      if (plan_first[m + 1] - plan_first[m] > 1) {
        if (seen >= shared_count || shared_ranks[seen] != (int32_t)m) return 2;
        ++seen;
      }
    if (seen != shared_count) return 2;
  }
  if (tap_depth == nullptr) tap_slot = nullptr;  // (sampling from the depth images)
  if (tap_grad && !(gws && plan_pixels && plan_first && plan_entries && plan_weights && plan_count >= 0)) return 1;
  std::vector<int32_t> blocks;
  for (int tile = 0; tile < ntiles; ++tile) {
    const int sg = tiles[tile * 2], fs0 = tiles[tile * 2 + 1], f = seg[sg * 4 + 1];
    for (int fs = fs0; fs < fs0 + FM_TRACK_TILE && fs < f; ++fs) {
      blocks.push_back(sg);
      blocks.push_back(fs);
    }
  }
  if (sim_track_points(depth, 0, kinv, ext, ext_inv, k, frames, xy, vis, seg, blocks.data(), (int)(blocks.size() / 2), height, width, ws, flag, tgt,
                       tap_slot, tap_depth) != 0)
    return 2;
  if (fm_track_loss_fwd(ws, flag, xy, vis, seg, tiles, ntiles, pmax, fmax, ext, tgt, frames, height, width, kind, delta, ax, ay, weight, partial, acc,
                        loss, scale, totals, gws, acc2, stream) != 0)
    return 2;
  if (!tap_grad) return 0;
  const int64_t n = (int64_t)height * width;
  for (long m = 0; m < plan_count; ++m) {
    const int64_t frame = plan_pixels[m] / n;
    const int px = (int)(plan_pixels[m] - frame * n);
    Mat3 ki;
    load_mat3(kinv + (size_t)frame * 9, ki);
    float ray[3];
    ray_dir(ki, pixel_center(px % width, width), pixel_center(px / width, height), ray);
    float sum = 0.f;
    for (int e = plan_first[m]; e < plan_first[m + 1]; ++e) {
      const float* v = gws + (size_t)plan_entries[e] * 3;
      sum += plan_weights[e] * (v[0] * ray[0] + v[1] * ray[1] + v[2] * ray[2]);
    }
    tap_grad[m] = sum;
  }
  return 0;
}

int fm_tap_grad_apply(const float* tap_grad, const int64_t* pixels, long count, const float* scale, const float* upstream_plus,
                      const float* upstream_minus, float* grad_depth, int* mismatch_flag, void*) {
  if (count < 0 || !scale || (count > 0 && !(tap_grad && pixels && grad_depth))) return 1;
  const float factor = scale[0] * ((upstream_plus ? upstream_plus[0] : 0.f) - (upstream_minus ? upstream_minus[0] : 0.f));
  if (factor == 0.f) return 0;
  if (mismatch_flag && count > 0) *mismatch_flag = 1;
  for (long m = 0; m < count; ++m) grad_depth[pixels[m]] += factor * tap_grad[m];
  return 0;
}

int fm_track_loss_bwd(const double* acc, const double* acc2, const float* scale, const float* upstream, const float* ext_inv,
                      const float* k, const float* kinv, int frames, float* g_ext, float* g_k, void*) {
  const double sc = (double)scale[0] * (upstream ? (double)upstream[0] : 1.0);
  for (int fr = 0; fr < frames; ++fr)
    track_frame_grads(acc + (size_t)fr * kTrackAccStride, acc2 + (size_t)fr * kTrackAcc2Stride, sc, ext_inv + (size_t)fr * 16,
                      k + (size_t)fr * 9, kinv + (size_t)fr * 9, g_ext + (size_t)fr * 16, g_k + (size_t)fr * 9);
  return 0;
}

int fm_track_scatter(const float* gws, const uint8_t* flag, const float* xy, const uint8_t*, const int32_t* seg,
                     const int32_t* blocks, int nblocks, int, const float* kinv, const float* scale, const float* upstream, int height,
                     int width, int depth_frame0, float* grad_depth, void*) {
  const float sc = scale[0] * (upstream ? upstream[0] : 1.f);
  for (int blk = 0; blk < nblocks; ++blk) {
    const int sg = blocks[blk * 2], fs = blocks[blk * 2 + 1];
    const int start = seg[sg * 4], pc = seg[sg * 4 + 2], off = seg[sg * 4 + 3];
    const int frame_s = start + fs;
    Mat3 ki;
    load_mat3(kinv + (size_t)frame_s * 9, ki);
    for (int p = 0; p < pc; ++p) {
      const size_t is = (size_t)off + (size_t)fs * pc + p;
      if (flag[is] == 0) continue;
      const Taps t = bilinear_taps(xy[is * 2], xy[is * 2 + 1], height, width);
      for (int kk = 0; kk < 4; ++kk) {
        if (!t.in[kk]) continue;
        const int tc = tap_col(t, kk), tr = tap_row(t, kk);
        float ray[3];
        ray_dir(ki, pixel_center(tc, width), pixel_center(tr, height), ray);
        grad_depth[(size_t)(frame_s - depth_frame0) * height * width + tr * width + tc] +=
            t.w[kk] * (gws[is * 3] * sc * ray[0] + gws[is * 3 + 1] * sc * ray[1] + gws[is * 3 + 2] * sc * ray[2]);
      }
    }
  }
  return 0;
}

int fm_track_scatter_plan(const float* xy, const uint8_t* vis, const int32_t* seg, const int32_t* blocks, int nblocks, int, int height,
                          int width, int64_t* keys, float* weights, void*) {
  for (int blk = 0; blk < nblocks; ++blk) {
    const int sg = blocks[blk * 2], fs = blocks[blk * 2 + 1];
    const int start = seg[sg * 4], pc = seg[sg * 4 + 2], off = seg[sg * 4 + 3];
    for (int p = 0; p < pc; ++p) {
      const size_t is = (size_t)off + (size_t)fs * pc + p;
      const float x = xy[is * 2], y = xy[is * 2 + 1];
      const bool live = vis[is] != 0 && x >= 0.f && y >= 0.f && x < 1.f && y < 1.f;
      const Taps t = bilinear_taps(x, y, height, width);
      for (int kk = 0; kk < 4; ++kk) {
        const bool used = live && t.in[kk];
        keys[is * 4 + kk] = used ? (int64_t)(start + fs) * height * width + (int64_t)tap_row(t, kk) * width + tap_col(t, kk) : (int64_t)-1;
        weights[is * 4 + kk] = used ? t.w[kk] : 0.f;
      }
    }
  }
  return 0;
}

int fm_depth_gather(const float* vectors, const int64_t* pixels, const int32_t* first, const int32_t* entries, const float* weights,
                    long count, const float* kinv, const float* scale, const float* upstream, int height, int width, long frame0,
                    float* grad_depth, void*) {
  const float sc = (scale ? scale[0] : 1.f) * (upstream ? upstream[0] : 1.f);
  const int64_t n = (int64_t)height * width;
  for (long m = 0; m < count; ++m) {
    const int64_t frame = pixels[m] / n;
    const int px = (int)(pixels[m] - frame * n);
    Mat3 ki;
    load_mat3(kinv + (size_t)frame * 9, ki);
    float ray[3];
    ray_dir(ki, pixel_center(px % width, width), pixel_center(px / width, height), ray);
    float sum = 0.f;
    for (int e = first[m]; e < first[m + 1]; ++e) {
      const float* v = vectors + (size_t)entries[e] * 3;
      sum += weights[e] * (v[0] * sc * ray[0] + v[1] * sc * ray[1] + v[2] * sc * ray[2]);
    }
    grad_depth[(size_t)(frame - frame0) * n + px] += sum;
  }
  return 0;
}

int fm_depth_gather_kgrad(const float* vectors, const int64_t* pixels, const int32_t* first, const int32_t* entries, const float* weights,
                          long count, const float* kinv, int height, int width, float* grad_depth, const double* kinv_acc, int frames_k,
                          float* g_k, int accumulate, void* stream) {
  if (count > 0 && fm_depth_gather(vectors, pixels, first, entries, weights, count, kinv, nullptr, nullptr, height, width, 0, grad_depth, stream) != 0)
    return 2;
  return frames_k > 0 ? fm_intrinsics_inverse_bwd(kinv_acc, kinv, frames_k, g_k, accumulate, stream) : 0;
}

// The one-launch backward of the planned sparse fit from the forward's correspondence records: pose-solve backward + closed-form
// dL/dK⁻¹ (fm_pose_solve_bwd_kinv), corr_backward per record, the planned gather — plus a check that `frame_first` slices the plan
// the way the device kernel's per-frame blocks rely on.
int fm_procrustes_bwd_planned(const float* corr, const float* kinv, float sens, long points, int batch, int frames, int height, int width,
                              const double* aux, const float* t_bwd, const float* g_t_bwd, const float* g_t_fwd, const int64_t* plan_pixels,
                              const int32_t* plan_first, const int32_t* plan_vectors, const float* plan_weights, const int32_t* frame_first,
                              float* grad_depth, float* grad_weights, float* g_k, int accumulate_k, void* stream) {
  if (!(corr && kinv && aux && t_bwd) || points < 1 || points > FM_FIT_BWD_MAX_POINTS) return 1;
  if (grad_depth && !(plan_pixels && plan_first && plan_vectors && plan_weights && frame_first)) return 1;
  const int pairs = batch * (frames - 1);
  const int64_t n = (int64_t)height * width;
  long count = 0;
  if (grad_depth) {
    count = frame_first[batch * frames];
    if (frame_first[0] != 0) return 1;
    for (int bf = 0; bf < batch * frames; ++bf) {
      if (frame_first[bf] > frame_first[bf + 1]) return 1;
      for (int m = frame_first[bf]; m < frame_first[bf + 1]; ++m) {
        if (plan_pixels[m] / n != bf) return 1;
        for (int e = plan_first[m]; e < plan_first[m + 1]; ++e) {  // the entries of a frame's pixels come from its two pairs only
          const int v = plan_vectors[e], later = v & 1, pair = (v >> 1) / (int)points;
          const int b = bf / frames, f = bf % frames;
          if (pair != b * (frames - 1) + f - later) return 1;
        }
      }
    }
  }
  std::vector<double> pair_grad((size_t)pairs * kPairGradStride), kinv_acc((size_t)batch * frames * 9);
  std::vector<float> point_grads((size_t)pairs * points * 6);
  if (fm_pose_solve_bwd_kinv(g_t_bwd, g_t_fwd, t_bwd, aux, kinv, batch, frames, pair_grad.data(), kinv_acc.data(), stream) != 0) return 2;
  for (int pr = 0; pr < pairs; ++pr) {
    const double* pg = pair_grad.data() + (size_t)pr * kPairGradStride;
    const double* ax = aux + (size_t)pr * kAuxStride;
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
    for (long j = 0; j < points; ++j) {
      const float* r = corr + ((size_t)pr * points + (size_t)j) * 8;
      Corr c{};
      c.q[0] = r[0], c.q[1] = r[1], c.q[2] = r[2], c.p[0] = r[3], c.p[1] = r[4], c.p[2] = r[5], c.w = r[6];
      std::memcpy(&c.idx, r + 7, sizeof(int));
      float gq[3], gp[3], gw;
      corr_backward(c, g, gq, gp, gw);
      if (sens != 0.f) gw *= sens * c.w * (1.f - c.w);
      if (grad_weights) grad_weights[(size_t)pr * n + c.idx] = gw;
      float* o = point_grads.data() + ((size_t)pr * points + (size_t)j) * 6;
      o[0] = gq[0], o[1] = gq[1], o[2] = gq[2], o[3] = gp[0], o[4] = gp[1], o[5] = gp[2];
    }
  }
  if (grad_depth && count > 0 &&
      fm_depth_gather(point_grads.data(), plan_pixels, plan_first, plan_vectors, plan_weights, count, kinv, nullptr, nullptr, height, width, 0, grad_depth,
                      stream) != 0)
    return 2;
  return g_k ? fm_intrinsics_inverse_bwd(kinv_acc.data(), kinv, batch * frames, g_k, accumulate_k, stream) : 0;
}

// ---- frame windows (fm_layout): the double gathers every view into a dense temporary and runs its dense namesake — the
// semantics of the strides, none of the device kernels' pointer arithmetic ----
static std::vector<float> sim_densify(const float* p, const fm_layout* lay, int batch, int frames, size_t per_frame) {
  std::vector<float> out;
  if (!p) return out;
  const bool given = lay && (lay->frame_stride != 0 || lay->batch_stride != 0);
  const size_t fs = given ? (size_t)lay->frame_stride : per_frame, bs = given ? (size_t)lay->batch_stride : per_frame * frames;
  out.resize((size_t)batch * frames * per_frame);
  for (int b = 0; b < batch; ++b)
    for (int f = 0; f < frames; ++f) std::memcpy(out.data() + ((size_t)b * frames + f) * per_frame, p + b * bs + f * fs, per_frame * sizeof(float));
  return out;
}
static const float* sim_or_null(const std::vector<float>& v) { return v.empty() ? nullptr : v.data(); }

int fm_flow_loss_fused_views(const float* depth, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd,
                             const float* flow_fwd, const float* flow_bwd, const float* mask_fwd, const float* mask_bwd,
                             const float* packed, const float* scale, int batch, int frames, int height, int width, int mapping_kind,
                             float delta, float ax, float ay, float* grad_depth, double* acc, int items, const fm_layout* layouts, void* stream) {
  const size_t n = (size_t)height * width;
  const auto d = sim_densify(depth, layouts ? layouts + 0 : nullptr, batch, frames, n);
  const auto ff = sim_densify(flow_fwd, layouts ? layouts + 1 : nullptr, batch, frames - 1, 2 * n);
  const auto fb = sim_densify(flow_bwd, layouts ? layouts + 2 : nullptr, batch, frames - 1, 2 * n);
  const auto mf = sim_densify(mask_fwd, layouts ? layouts + 3 : nullptr, batch, frames - 1, n);
  const auto mb = sim_densify(mask_bwd, layouts ? layouts + 4 : nullptr, batch, frames - 1, n);
  return fm_flow_loss_fused(sim_or_null(d), k, kinv, t_fwd, t_bwd, sim_or_null(ff), sim_or_null(fb), sim_or_null(mf), sim_or_null(mb), packed, scale, batch,
                            frames, height, width, mapping_kind, delta, ax, ay, grad_depth, acc, items, stream);
}

int fm_flow_valid_norm_views(const float* mask_fwd, const float* mask_bwd, int batch, int pairs, long pixels, float weight, double* vsum,
                             float* norm, const fm_layout* layouts, void* stream) {
  const auto mf = sim_densify(mask_fwd, layouts ? layouts + 0 : nullptr, batch, pairs, (size_t)pixels);
  const auto mb = sim_densify(mask_bwd, layouts ? layouts + 1 : nullptr, batch, pairs, (size_t)pixels);
  return fm_flow_valid_norm(sim_or_null(mf), sim_or_null(mb), (long)batch * pairs * pixels, weight, vsum, norm, stream);
}

int fm_flow_pack_inputs_views(const float* flow_fwd, const float* flow_bwd, const float* mask_fwd, const float* mask_bwd, int batch,
                              int frames, int height, int width, float* packed, const fm_layout* layouts, void* stream) {
  const size_t n = (size_t)height * width;
  const auto ff = sim_densify(flow_fwd, layouts ? layouts + 0 : nullptr, batch, frames - 1, 2 * n);
  const auto fb = sim_densify(flow_bwd, layouts ? layouts + 1 : nullptr, batch, frames - 1, 2 * n);
  const auto mf = sim_densify(mask_fwd, layouts ? layouts + 2 : nullptr, batch, frames - 1, n);
  const auto mb = sim_densify(mask_bwd, layouts ? layouts + 3 : nullptr, batch, frames - 1, n);
  return fm_flow_pack_inputs(sim_or_null(ff), sim_or_null(fb), sim_or_null(mf), sim_or_null(mb), batch, frames, height, width, packed, stream);
}

struct SimProcViews {
  std::vector<float> depth, surfaces, flow, weights;
  SimProcViews(const float* d, const float* s, const float* fl, const float* w, const fm_layout* lay, int batch, int frames, int height, int width) {
    const size_t n = (size_t)height * width;
    depth = sim_densify(d, lay ? lay + 0 : nullptr, batch, frames, n);
    surfaces = sim_densify(s, lay ? lay + 1 : nullptr, batch, frames, 3 * n);
    flow = sim_densify(fl, lay ? lay + 2 : nullptr, batch, frames - 1, 2 * n);
    weights = sim_densify(w, lay ? lay + 3 : nullptr, batch, frames - 1, n);
  }
};

int fm_procrustes_fit_views(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow, const float* weights, float sens,
                            const int64_t* indices, long points, int batch, int frames, int height, int width, double* stats, float* t_bwd,
                            float* t_fwd, double* aux, const fm_layout* layouts, void* stream) {
  const SimProcViews v(depth, surfaces, bwd_flow, weights, layouts, batch, frames, height, width);
  return fm_procrustes_fit(sim_or_null(v.depth), kinv, sim_or_null(v.surfaces), sim_or_null(v.flow), sim_or_null(v.weights), sens, indices, points, batch, 1,
                           frames, height, width, stats, t_bwd, t_fwd, aux, stream);
}

int fm_procrustes_fit_chain_views(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow, const float* weights, float sens,
                                  const int64_t* indices, long points, int batch, int frames, int height, int width, double* work, float* t_bwd,
                                  float* t_fwd, double* aux, float* ext, float* corr_out, const float* tap_records, const fm_layout* layouts,
                                  void* stream) {
  const SimProcViews v(depth, surfaces, bwd_flow, weights, layouts, batch, frames, height, width);
  return fm_procrustes_fit_chain(sim_or_null(v.depth), kinv, sim_or_null(v.surfaces), sim_or_null(v.flow), sim_or_null(v.weights), sens, indices, points,
                                 batch, frames, height, width, work, t_bwd, t_fwd, aux, ext, corr_out, tap_records, stream);
}

int fm_procrustes_scatter_views(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow, const float* weights,
                                float sens, const int64_t* indices, long points, int batch, int frames, int height, int width, const double* aux,
                                const double* pair_grad, float* grad_depth, float* grad_surfaces, float* grad_weights, double* kinv_acc,
                                float* point_grads, const fm_layout* layouts, void* stream) {
  const SimProcViews v(depth, surfaces, bwd_flow, weights, layouts, batch, frames, height, width);
  return fm_procrustes_scatter(sim_or_null(v.depth), kinv, sim_or_null(v.surfaces), sim_or_null(v.flow), sim_or_null(v.weights), sens, indices, points, batch,
                               1, frames, height, width, aux, pair_grad, grad_depth, grad_surfaces, grad_weights, kinv_acc, point_grads, nullptr, stream);
}

int fm_procrustes_scatter_plan_views(const float* bwd_flow, const int64_t* indices, long points, int batch, int frames, int height, int width,
                                     int64_t* keys, float* weights, const fm_layout* flow_layout, void* stream) {
  const auto fl = sim_densify(bwd_flow, flow_layout, batch, frames - 1, 2 * (size_t)height * width);
  return fm_procrustes_scatter_plan(sim_or_null(fl), indices, points, batch, frames, height, width, keys, weights, stream);
}

// ---- frame sharding: the local work of the halo exchange ----
int fm_halo_copy(const float* grad, long n, int frames, float* sent_first, float* sent_last, void*) {
  if (!grad || n < 1 || frames < 1) return 1;
  if (sent_first) std::memcpy(sent_first, grad, sizeof(float) * n);
  if (sent_last) std::memcpy(sent_last, grad + (size_t)(frames - 1) * n, sizeof(float) * n);
  return 0;
}
// The ghost terms: a two-frame flow loss per side, evaluated by this file's own fused loss with the other direction masked out — the
// shared frame as source, a stand-in (never read where the mask is zero ... its term is multiplied by a zero mask) as the other frame.
int fm_flow_ghost_terms(const float* depth_first, const float* pose_first, const float* flow_first, const float* mask_first, float* grad_first,
                        const float* depth_last, const float* pose_last, const float* flow_last, const float* mask_last, float* grad_last,
                        const float* kinv, const float* k_dst, const float* norm, const float* upstream, int height, int width, int kind,
                        float delta, float ax, float ay, void* stream) {
  if (!kinv || !k_dst || !norm || height < 1 || width < 1 || kind < 0 || kind > 2) return 1;
  const size_t n = (size_t)height * width;
  const float scale = norm[0] * (upstream ? upstream[0] : 1.f);
  for (int side = 0; side < 2; ++side) {
    const float* depth = side == 0 ? depth_first : depth_last;
    const float* pose = side == 0 ? pose_first : pose_last;
    const float* flow = side == 0 ? flow_first : flow_last;
    const float* mask = side == 0 ? mask_first : mask_last;
    float* grad = side == 0 ? grad_first : grad_last;
    if (!grad) continue;
    if (!depth || !pose || !flow || !mask) return 1;
    // frames [shared, other] with the ghost term as the FORWARD direction of their pair (frame 0 -> camera 1): K of frame 0 = the shared
    // frame's, of frame 1 = the destination's; the backward direction (source frame 1) has a zero mask
    std::vector<float> d2(2 * n, 1.f), k2(18), ki2(18), tf(16), tb(16, 0.f), fb(2 * n, 0.f), mb(n, 0.f), g2(2 * n, 0.f);
    std::memcpy(d2.data(), depth, sizeof(float) * n);
    std::memcpy(ki2.data(), kinv, sizeof(float) * 9);
    std::memcpy(ki2.data() + 9, kinv, sizeof(float) * 9);
    std::memcpy(k2.data() + 9, k_dst, sizeof(float) * 9);
    std::memcpy(k2.data(), k_dst, sizeof(float) * 9);
    std::memcpy(tf.data(), pose, sizeof(float) * 16);
    for (int i = 0; i < 4; ++i) tb[i * 5] = 1.f;
    std::vector<double> acc(2 * 2 * kFlowAccStride, 0.0);
    if (fm_flow_loss_fused(d2.data(), k2.data(), ki2.data(), tf.data(), tb.data(), flow, fb.data(), mask, mb.data(), nullptr, &scale, 1, 2, height,
                           width, kind, delta, ax, ay, g2.data(), acc.data(), 0, stream) != 0)
      return 2;
    for (size_t i = 0; i < n; ++i) grad[i] += g2[i];
  }
  return 0;
}
int fm_halo_add(float* grad, long n, int frames, const float* dense_first, const float* dense_last, void*) {
  if (!grad || n < 1 || frames < 1 || (frames == 1 && dense_first && dense_last)) return 1;
  for (long i = 0; dense_first && i < n; ++i) grad[i] += dense_first[i];
  for (long i = 0; dense_last && i < n; ++i) grad[(size_t)(frames - 1) * n + i] += dense_last[i];
  return 0;
}
int fm_halo_delta(const float* grad, long n, int frames, const float* sent_first, const int64_t* pixels_first, long count_first, float* out_first,
                  const float* sent_last, const int64_t* pixels_last, long count_last, float* out_last, void*) {
  if (!grad || n < 1 || frames < 1 || count_first < 0 || count_last < 0) return 1;
  if ((out_first && (!sent_first || (!pixels_first && count_first))) || (out_last && (!sent_last || (!pixels_last && count_last)))) return 1;
  for (long i = 0; out_first && i < count_first; ++i) out_first[i] = grad[pixels_first[i]] - sent_first[pixels_first[i]];
  const float* last = grad + (size_t)(frames - 1) * n;
  for (long i = 0; out_last && i < count_last; ++i) out_last[i] = last[pixels_last[i]] - sent_last[pixels_last[i]];
  return 0;
}
int fm_halo_ghost_begin(const float* grad, long n, int frames, const int64_t* pixels_first, long count_first, float* base_first,
                        const int64_t* pixels_last, long count_last, float* base_last, const float* t_fwd, const float* t_bwd, int pairs, const float* k,
                        const float* kinv, float* pack, void*) {
  if (!grad || n < 1 || frames < 1 || count_first < 0 || count_last < 0) return 1;
  if ((base_first && !pixels_first && count_first) || (base_last && !pixels_last && count_last)) return 1;
  if (pack && (!t_fwd || !t_bwd || !k || !kinv || pairs < 1)) return 1;
  if (pack) {
    const size_t last = (size_t)(pairs - 1) * 16;
    std::memcpy(pack, t_fwd, sizeof(float) * 16);
    std::memcpy(pack + 16, t_bwd + last, sizeof(float) * 16);
    std::memcpy(pack + 32, t_bwd, sizeof(float) * 16);
    std::memcpy(pack + 48, t_fwd + last, sizeof(float) * 16);
    std::memcpy(pack + 64, k, sizeof(float) * 9);
    std::memcpy(pack + 73, kinv, sizeof(float) * 9);
  }
  for (long i = 0; base_first && i < count_first; ++i) base_first[i] = grad[pixels_first[i]];
  const float* last_frame = grad + (size_t)(frames - 1) * n;
  for (long i = 0; base_last && i < count_last; ++i) base_last[i] = last_frame[pixels_last[i]];
  return 0;
}
int fm_halo_delta_sparse(const float* grad, long n, int frames, const float* base_first, const int64_t* pixels_first, long count_first, float* out_first,
                         const float* base_last, const int64_t* pixels_last, long count_last, float* out_last, void*) {
  if (!grad || n < 1 || frames < 1 || count_first < 0 || count_last < 0) return 1;
  if ((out_first && count_first && (!base_first || !pixels_first)) || (out_last && count_last && (!base_last || !pixels_last))) return 1;
  for (long i = 0; out_first && i < count_first; ++i) out_first[i] = grad[pixels_first[i]] - base_first[i];
  const float* last = grad + (size_t)(frames - 1) * n;
  for (long i = 0; out_last && i < count_last; ++i) out_last[i] = last[pixels_last[i]] - base_last[i];
  return 0;
}
int fm_halo_scatter(float* grad, long n, int frames, const int64_t* pixels_first, const float* values_first, long count_first,
                    const int64_t* pixels_last, const float* values_last, long count_last, void*) {
  if (!grad || n < 1 || frames < 1 || count_first < 0 || count_last < 0) return 1;
  if ((values_first && !pixels_first && count_first) || (values_last && !pixels_last && count_last) || (frames == 1 && values_first && values_last)) return 1;
  for (long i = 0; values_first && i < count_first; ++i) grad[pixels_first[i]] += values_first[i];
  float* last = grad + (size_t)(frames - 1) * n;
  for (long i = 0; values_last && i < count_last; ++i) last[pixels_last[i]] += values_last[i];
  return 0;
}

}  // extern "C"
