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
// Launch plan (VALU/latency-bound, inputs ≈ 13 MB):
//   track_points   one thread per (frame-in-segment, point): sample xyz, lift to world
//                  X_w = E_fs·xyz, keep h = Σ w_k z_k [u_k,v_k,1], fold visibility ∧
//                  source-in-frame into one byte.
//   track_pairs    EVERY residual evaluated ONCE.  A thread owns one point and a tile of
//                  kTrackTile source frames held in registers (X_w and its running gradient),
//                  and loops over all target frames: the target-role sums (loss, count,
//                  Σ gX'⊗[X_w;1], dK) are block-reduced per target frame, the source-role
//                  gradient gX_w accumulates in registers and is finished in an epilogue
//                  (dL/dxyz per point, dL/dE_fs, dL/dK⁻¹ sums).  All UNSCALED: the visible
//                  count is not known until the launch ends.
//   finalize_fwd   loss, count, scale = weight / max(count, 1).
//   finalize_bwd   small-matrix chain rules -> dL/dE (F,4,4), dL/dK (F,3,3), scaled.
//   track_scatter  per-point dL/dxyz -> dL/ddepth through the 4 bilinear taps (atomics), scaled.
// (First version: a target-major and a source-major pass, each re-evaluating the whole
// chain: 0.47 ms at C2; this one: see DESIGN.md §3.4.)
#include "../../include/flowmap_hip.h"
#include "fm_device.h"
#include "fm_pose.h"

namespace fm {

// -DFM_TRACK_CLOCKS (tools/track_clocks.py): lane 0 of the first waves of track_pairs accumulates wall_clock64 (100 MHz) deltas per
// phase — prologue, per target: scalar constants / residual terms / reduction + store, epilogue — read back by fm_debug_track_clocks.
#ifdef FM_TRACK_CLOCKS
constexpr int kTrackClockSlots = 8, kTrackClockWaves = 2048;
__device__ long long fm_track_clock_buffer[kTrackClockWaves][kTrackClockSlots];
#define FM_TCLK(var) const long long var = wall_clock64()
#define FM_TCLK_ADD(slot, a, b) clk[slot] += (b) - (a)
#else
#define FM_TCLK(var)
#define FM_TCLK_ADD(slot, a, b)
#endif
constexpr unsigned kXcds = 8;  // accelerator complex dies of an MI355X, each with its own L2 (MI355X_MICROARCH.md)
constexpr int kTrackTile = FM_TRACK_TILE;  // source frames per thread (registers: 2 x 3 x kTrackTile floats)
__host__ __device__ constexpr size_t track_partial_stride(int fmax) { return (size_t)fmax * kTrackSums + kTrackTile * 21; }

struct TrackGeom {
  const float* xy;        // (total, 2) packed track positions
  const uint8_t* vis;     // (total)
  const int32_t* seg;     // (S, 4): start_frame, f, p, offset (in points)
  const int32_t* blocks;  // (NB, 2): segment, local frame
  int height, width;
};

// The work-space record of a track point (fm_pose.h: kTrackWs floats) is stored per (segment, frame) block as kTrackWs PLANES of p_count
// values — component c of point p of the block starting at point index `block` sits at (block·kTrackWs + c·p_count + p) — so that a wave's
// stores and loads of one component are contiguous (as records of 36 bytes every store instruction touched 18 lines).
__device__ __forceinline__ float* ws_plane(float* ws, size_t block, int p_count, int p) { return ws + block * kTrackWs + p; }
__device__ __forceinline__ const float* ws_plane(const float* ws, size_t block, int p_count, int p) { return ws + block * kTrackWs + p; }

// ---------------------------------------------------------------- track_points ------
// One track point of one frame: ws[idx] = [xyz | X_w | h], flag[idx]; returns the flag, X_w in xw.
__device__ __forceinline__ bool track_sample(const TrackGeom& g, const float* depth, int depth_frame0, const float* kinv, const float* ext,
                                             int frame, size_t idx, int p_count, int p, float* ws, uint8_t* flag, float xw[3]) {
  const float2 q = reinterpret_cast<const float2*>(g.xy)[idx];
  Mat3 ki;
  Pose e;
  load_mat3(kinv + (size_t)frame * 9, ki);
  load_pose44(ext + (size_t)frame * 16, e);
  const Taps t = bilinear_taps(q.x, q.y, g.height, g.width);
  const float* d = depth + (size_t)(frame - depth_frame0) * g.height * g.width;
  float xyz[3] = {0.f, 0.f, 0.f}, hh[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!t.in[k]) continue;
    const int tc = tap_col(t, k), tr = tap_row(t, k);
    const float ut = pixel_center(tc, g.width), vt = pixel_center(tr, g.height);
    float ray[3];
    ray_dir(ki, ut, vt, ray);
    const float z = d[tr * g.width + tc];
    xyz[0] += (ray[0] * z) * t.w[k];
    xyz[1] += (ray[1] * z) * t.w[k];
    xyz[2] += (ray[2] * z) * t.w[k];
    hh[0] += z * ut * t.w[k];
    hh[1] += z * vt * t.w[k];
    hh[2] += z * t.w[k];
  }
  apply_pose(e, xyz, xw);
  float* o = ws_plane(ws, idx - p, p_count, p);
  const size_t pc = (size_t)p_count;
  o[0] = xyz[0]; o[pc] = xyz[1]; o[2 * pc] = xyz[2];
  o[3 * pc] = xw[0];  o[4 * pc] = xw[1];  o[5 * pc] = xw[2];
  o[6 * pc] = hh[0];  o[7 * pc] = hh[1];  o[8 * pc] = hh[2];
  const bool inside = q.x >= 0.f && q.y >= 0.f && q.x < 1.f && q.y < 1.f;
  const bool live = g.vis[idx] != 0 && inside;
  flag[idx] = live ? 1 : 0;
  return live;
}

// track_sample for the N source frames of a tile at once, in three phases — every position first, then every tap depth, then the
// arithmetic and the stores — so that a lane has N·(1 + 4) loads in flight instead of walking N dependent load -> load chains one
// after the other (each link a cold round trip of 2.5-5 us: the prologue of track_pairs took 78 of a wave's 178 us that way;
// tools/track_clocks.py).  want[n] is wave-uniform (the frame exists and is sampled by this rank); idx[n] is in range for every lane
// (clamped point); `store` masks the lanes beyond the segment's points.  Same arithmetic per item as track_sample (a tap outside
// the image contributes an exact zero instead of being skipped).
// tap_slot / tap_depth (fm_track_loss_fused_fwd_taps): the four tap depths of point idx come from the compact tap image the flow pass
// leaves behind — tap_depth[rank], rank = the low 29 bits of tap_slot[4·idx + k]; bit 30: the pixel is shared with another track point;
// bit 29: read the depth image after all (a pixel another operator updates after the flow pass); slot -1: a tap that contributes
// nothing — instead of four cold lines of the depth images.
constexpr int kTapRank = 0x1fffffff, kTapDense = 0x20000000, kTapShared = 0x40000000;
template <int N>
__device__ __forceinline__ void track_sample_many(const TrackGeom& g, const float* depth, int depth_frame0, const float* kinv, const float* ext,
                                                  const int (&frame)[N], const size_t (&idx)[N], const bool (&want)[N], bool store, int p_count,
                                                  int p, float* ws, uint8_t* flag, float (&xw)[N][3], bool (&live)[N],
                                                  const int32_t* tap_slot = nullptr, const float* tap_depth = nullptr) {
  float2 q[N];
  uint8_t vis[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    q[n] = make_float2(0.f, 0.f);
    vis[n] = 0;
    if (want[n]) {
      q[n] = reinterpret_cast<const float2*>(g.xy)[idx[n]];
      vis[n] = g.vis[idx[n]];
    }
  }
  Taps t[N];
  float z[N][4];
  // The compact tap image, branch-free (the usual case): every slot row first, then — one wave-uniform test later — one 8-byte load per
  // image row of every sample, all in flight together.  Lane-divergent branches around the loads (a row with no first tap, a pixel
  // flagged "read the depth image") made the compiler wait for each sample's loads where its branches joined: a dozen dependent
  // round trips in the prologue instead of two.
  int4 s4[N];
  bool image_only = tap_depth != nullptr;
#ifdef FM_TRACK_SKIP_SAMPLE
  image_only = false;
#endif
  if (image_only) {
    bool dense = false;
#pragma unroll
    for (int n = 0; n < N; ++n) {
      s4[n] = make_int4(-1, -1, -1, -1);
      if (want[n]) s4[n] = reinterpret_cast<const int4*>(tap_slot)[idx[n]];
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const int sl[4] = {s4[n].x, s4[n].y, s4[n].z, s4[n].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) dense |= sl[k] >= 0 && (sl[k] & kTapDense) != 0;
    }
    image_only = __ballot(dense) == 0;  // (pixels an in-pass Adam update leaves to other operators are read from the depth image: the general form below)
  }
#pragma unroll
  for (int n = 0; n < N; ++n) {
    t[n] = bilinear_taps(q[n].x, q[n].y, g.height, g.width);
#pragma unroll
    for (int k = 0; k < 4; ++k) z[n][k] = 0.f;
    if (want[n]) {
      const float* d = depth + (size_t)(frame[n] - depth_frame0) * g.height * g.width;
      const int x1 = min(t[n].x0 + 1, g.width - 1), y1 = min(t[n].y0 + 1, g.height - 1);  // clamped reads; masked by in[] below
      const int a[4] = {t[n].y0 * g.width + t[n].x0, t[n].y0 * g.width + x1, y1 * g.width + t[n].x0, y1 * g.width + x1};
#ifdef FM_TRACK_SKIP_SAMPLE  // (timing experiments: what the tap gathers cost)
      if (true) {
#pragma unroll
        for (int k = 0; k < 4; ++k) z[n][k] = 1.5f;
      } else
#endif
      if (image_only) {
        const int sl[4] = {s4[n].x, s4[n].y, s4[n].z, s4[n].w};
#pragma unroll
        for (int r = 0; r < 2; ++r) {  // (the two taps of an image row are neighbouring ranks when both exist; the image is padded by one value)
          const int sa = sl[2 * r], sb = sl[2 * r + 1];
          const int base = sa >= 0 ? (sa & kTapRank) : (sb >= 0 ? (sb & kTapRank) : 0);
          float2 v;
          __builtin_memcpy(&v, tap_depth + base, sizeof(float2));
          z[n][2 * r] = sa >= 0 ? v.x : 0.f;
          z[n][2 * r + 1] = sb >= 0 ? (sa >= 0 ? v.y : v.x) : 0.f;
        }
      } else if (tap_depth != nullptr) {
        const int4 s4g = reinterpret_cast<const int4*>(tap_slot)[idx[n]];
        const int sl[4] = {s4g.x, s4g.y, s4g.z, s4g.w};
        // the two taps of an image row are neighbouring pixels, hence neighbouring ranks: one 8-byte load per row (the image is padded by one value)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int sa = sl[2 * r], sb = sl[2 * r + 1];
          if ((sa >= 0 && (sa & kTapDense)) || (sb >= 0 && (sb & kTapDense))) {  // (a pixel another operator updates after the flow pass: read the depth image)
            z[n][2 * r] = sa < 0 ? 0.f : (sa & kTapDense) ? d[a[2 * r]] : tap_depth[sa & kTapRank];
            z[n][2 * r + 1] = sb < 0 ? 0.f : (sb & kTapDense) ? d[a[2 * r + 1]] : tap_depth[sb & kTapRank];
          } else if (sa >= 0) {
            float2 v;
            __builtin_memcpy(&v, tap_depth + (sa & kTapRank), sizeof(float2));
            z[n][2 * r] = v.x;
            z[n][2 * r + 1] = sb >= 0 ? v.y : 0.f;
          } else {
            z[n][2 * r] = 0.f;
            z[n][2 * r + 1] = sb >= 0 ? tap_depth[sb & kTapRank] : 0.f;
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) z[n][k] = d[a[k]];
      }
    }
  }
#pragma unroll
  for (int n = 0; n < N; ++n) {
    live[n] = false;
    xw[n][0] = xw[n][1] = xw[n][2] = 0.f;
    if (!want[n]) continue;
    Mat3 ki;
    load_mat3(kinv + (size_t)frame[n] * 9, ki);
    float xyz[3] = {0.f, 0.f, 0.f}, hh[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int tc = tap_col(t[n], k), tr = tap_row(t[n], k);
      const float ut = pixel_center(tc, g.width), vt = pixel_center(tr, g.height);
      float ray[3];
      ray_dir(ki, ut, vt, ray);
      const float zk = t[n].in[k] ? z[n][k] : 0.f, wk = t[n].in[k] ? t[n].w[k] : 0.f;
      xyz[0] += (ray[0] * zk) * wk;
      xyz[1] += (ray[1] * zk) * wk;
      xyz[2] += (ray[2] * zk) * wk;
      hh[0] += zk * ut * wk;
      hh[1] += zk * vt * wk;
      hh[2] += zk * wk;
    }
    Pose e;
    load_pose44(ext + (size_t)frame[n] * 16, e);
    apply_pose(e, xyz, xw[n]);
    const bool inside = q[n].x >= 0.f && q[n].y >= 0.f && q[n].x < 1.f && q[n].y < 1.f;
    live[n] = store && vis[n] != 0 && inside;
    if (store) {
#ifndef FM_TRACK_SKIP_WS  // (timing experiments: what the nine strided stores cost)
      float* o = ws_plane(ws, idx[n] - p, p_count, p);
      const size_t pc = (size_t)p_count;
      o[0] = xyz[0]; o[pc] = xyz[1]; o[2 * pc] = xyz[2];
      o[3 * pc] = xw[n][0]; o[4 * pc] = xw[n][1]; o[5 * pc] = xw[n][2];
      o[6 * pc] = hh[0];  o[7 * pc] = hh[1];  o[8 * pc] = hh[2];
#endif
      flag[idx[n]] = live[n] ? 1 : 0;
    }
  }
}

__global__ void __launch_bounds__(256) track_points_kernel(TrackGeom g, const float* depth, int depth_frame0, const float* kinv,
                                                           const float* ext, float* ws, uint8_t* flag) {
  const int sg = g.blocks[blockIdx.x * 2], fl = g.blocks[blockIdx.x * 2 + 1];
  const int start = g.seg[sg * 4], p_count = g.seg[sg * 4 + 2], off = g.seg[sg * 4 + 3];
  const int p = blockIdx.y * blockDim.x + threadIdx.x;
  if (p >= p_count) return;
  float xw[3];
  track_sample(g, depth, depth_frame0, kinv, ext, start + fl, (size_t)off + (size_t)fl * p_count + p, p_count, p, ws, flag, xw);
}

// Per frame: the target-role constants (au, av, c) of track_target (fm_pose.h).
__global__ void __launch_bounds__(64) track_targets_kernel(const float* ext_inv, const float* k, int frames, float* tgt) {
  const int fr = blockIdx.x * blockDim.x + threadIdx.x;
  if (fr < frames) track_target(ext_inv + (size_t)fr * 16, k + (size_t)fr * 9, tgt + (size_t)fr * kTrackTgt);
}

// ----------------------------------------------------------------- track_pairs ------
// One wave per block: 64 points x kTrackTile source frames.  grid: (source tiles, point groups);
// tiles[(segment, first local source frame)].  Waves never synchronise with each other: the
// per-target sums are reduced inside the wave (DPP) and added by lane 63 with fp64 atomics.
// Wave sum of NV per-lane values, written by lane 63 as this wave's partial (plain stores: the
// partials of all waves are summed per frame by track_reduce_kernel — no atomics, deterministic).
template <int NV>
__device__ __forceinline__ void wave_store(const float (&v)[NV], float* dst) {
  float tot[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) tot[i] = wave_sum_lane63(v[i]);
  if ((threadIdx.x & (kWave - 1)) == kWave - 1) {
#pragma unroll
    for (int i = 0; i < NV; ++i) dst[i] = tot[i];
  }
}

// The per-target-frame reduction of the 14 sums runs once per target frame and wave: as compiled from
// wave_sum_lane63 it took 234 of the loop's 529 instructions (every DPP step a v_mov_b32 of the identity, a
// v_mov_b32_dpp and an add).  Here each step is ONE v_add_f32_dpp (lanes the row mask disables keep their
// value), the values interleaved so that no instruction reads a register written less than 7 instructions
// earlier — 7 registers per group — (the VALU -> DPP hazard needs 2 wait states; the leading s_nop covers the
// compiler's own last write).
__device__ __forceinline__ void wave_sum7_lane63(float& v0, float& v1, float& v2, float& v3, float& v4, float& v5, float& v6) {
#define FM_DPP7(ctrl)                                                                                                              \
  asm volatile("v_add_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\tv_add_f32_dpp %2, %2, %2 " ctrl          \
               "\n\tv_add_f32_dpp %3, %3, %3 " ctrl "\n\tv_add_f32_dpp %4, %4, %4 " ctrl "\n\tv_add_f32_dpp %5, %5, %5 " ctrl    \
               "\n\tv_add_f32_dpp %6, %6, %6 " ctrl                                                                               \
               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6))
  asm volatile("s_nop 1");
  FM_DPP7("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
  FM_DPP7("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
  FM_DPP7("row_half_mirror row_mask:0xf bank_mask:0xf");
  FM_DPP7("row_mirror row_mask:0xf bank_mask:0xf");
  FM_DPP7("row_bcast:15 row_mask:0xa bank_mask:0xf");
  FM_DPP7("row_bcast:31 row_mask:0xc bank_mask:0xf");
#undef FM_DPP7
}

// Round 3: a TRANSPOSING reduction for the 14 (padded to 16) per-target sums.  Summing each value across the wave separately costs 6
// DPP steps per value (84 instructions per target iteration of ~430).  Here every step halves the number of live registers instead:
// gfx950's v_permlane32_swap / v_permlane16_swap exchange register halves between lane groups, so after one swap + one add a
// register holds value i in one half of the lanes and value i+8 in the other, each already summed over the pair of lanes it came
// from (inline asm: this compiler lowers the SECOND result of __builtin_amdgcn_permlane32_swap / 16_swap to the first — it emitted
// v_add v, vdst, vdst; tools/probes/transpose_reduce_probe.hip).  16 registers -> 8 (lanes l, l+32) -> 4 (rows r, r+1) -> 2 (xor 8: two selects + one DPP add) -> 1 (half-row mirror) ->
// the quad's two DPP steps: 35 instructions, and lane L ends with the wave total of value L >> 2 (one masked store by every fourth
// lane instead of 14 stores by lane 63).  The fp32 summation tree differs from the per-value one in the last bits only.
// (wave_transpose_sum16: fm_device.h)
// NV (a multiple of 7) per-lane values -> their wave totals in lane 63, in place
template <int NV>
__device__ __forceinline__ void wave_sum_lane63_x7(float (&v)[NV]) {
  static_assert(NV % 7 == 0, "groups of seven registers");
#pragma unroll
  for (int i = 0; i < NV; i += 7) wave_sum7_lane63(v[i], v[i + 1], v[i + 2], v[i + 3], v[i + 4], v[i + 5], v[i + 6]);
}

// track_pair_term (fm_pose.h) for TWO source frames of the same point at once, the pair held as the two halves
// of packed registers: the multiply-adds of the projection, of the twelve S sums and of dL/dX_w become
// v_pk_fma_f32 / v_pk_mul_f32 (two fp32 lanes per instruction); reciprocal, reciprocal square root, the range
// tests and the selects of the robust kernel stay per element.  Same arithmetic per element as the scalar
// function (which the host double runs).  `a` holds a partial sum per half, added together after the tile.
typedef float v2f __attribute__((ext_vector_type(2)));

// (round 5) in SCALED image coordinates — track_pair_term_scaled (fm_pose.h): `ts` are the target's rows pre-multiplied by the aspect factors
// (track_scale_target), gt_xs / gt_ys the track position pre-multiplied; the sums S0 / S1 come out divided by ax / ay (the kernel multiplies them
// back where it stores a target's totals).
template <int KIND, bool GRAD>
__device__ __forceinline__ void track_pair_term2(const float (&ts)[kTrackTgt], const v2f (&xw)[3], float gt_xs, float gt_ys, v2f m, float delta,
                                                 float inv_delta, float ax, float ay, v2f (&a)[kTrackSums], v2f (&gxw)[3]) {
  const v2f xu = ts[0] * xw[0] + (ts[1] * xw[1] + (ts[2] * xw[2] + ts[3]));
  const v2f xv = ts[4] * xw[0] + (ts[5] * xw[1] + (ts[6] * xw[2] + ts[7]));
  const v2f x2 = ts[8] * xw[0] + (ts[9] * xw[1] + (ts[10] * xw[2] + ts[11]));
  v2f q;
  q.x = fm_rcp(x2.x + kProjEps);
  q.y = fm_rcp(x2.y + kProjEps);
  const bool ok0 = fabsf(q.x) <= 3.0e38f, ok1 = fabsf(q.y) <= 3.0e38f;
  q.x = ok0 ? q.x : 0.f;
  q.y = ok1 ? q.y : 0.f;
  const v2f zero = {0.f, 0.f};
  const v2f u = __builtin_elementwise_fma(xu, q, zero), v = __builtin_elementwise_fma(xv, q, zero);  // ax·u, ay·v; never −0
  const unsigned bx = __float_as_uint(ax), by = __float_as_uint(ay);
  m.x = (ok0 && __float_as_uint(u.x) < bx && __float_as_uint(v.x) < by) ? m.x : 0.f;
  m.y = (ok1 && __float_as_uint(u.y) < bx && __float_as_uint(v.y) < by) ? m.y : 0.f;
  const v2f rx = u - gt_xs, ry = v - gt_ys;
  const v2f ss = rx * rx + ry * ry;
  v2f rho, coef;  // ρ and dρ/dr = coef·r
  if (KIND == kL2) {
    rho = 0.5f * ss;
    coef = 1.f;
  } else if (KIND == kL1) {
    v2f inv_n;
    inv_n.x = ss.x > 0.f ? fm_rsq(ss.x) : 0.f;
    inv_n.y = ss.y > 0.f ? fm_rsq(ss.y) : 0.f;
    rho = ss * inv_n;
    coef = inv_n;
  } else {
    coef.x = fminf(fm_rsq(ss.x), inv_delta);
    coef.y = fminf(fm_rsq(ss.y), inv_delta);
    const v2f t = ss * coef;
    v2f knee;
    knee.x = fminf(t.x, delta);
    knee.y = fminf(t.y, delta);
    rho = -0.5f * knee + t;
  }
  a[12] = rho * m + a[12];
  a[13] += m;
  if (!GRAD) return;
  const v2f gc = m * coef;
  const v2f wu = gc * rx, wv = gc * ry;  // dL/du', dL/dv' (unscaled by the loss normaliser)
  const v2f o0 = q * wu, o1 = q * wv, o2 = q * (wu * u + wv * v);
  a[0] = o0 * xw[0] + a[0];
  a[1] = o0 * xw[1] + a[1];
  a[2] = o0 * xw[2] + a[2];
  a[3] += o0;
  a[4] = o1 * xw[0] + a[4];
  a[5] = o1 * xw[1] + a[5];
  a[6] = o1 * xw[2] + a[6];
  a[7] += o1;
  a[8] = o2 * xw[0] + a[8];
  a[9] = o2 * xw[1] + a[9];
  a[10] = o2 * xw[2] + a[10];
  a[11] += o2;
  gxw[0] = o0 * ts[0] + (o1 * ts[4] + (gxw[0] - o2 * ts[8]));  // dL/dX_w = ω'·(au', av', −c)
  gxw[1] = o0 * ts[1] + (o1 * ts[5] + (gxw[1] - o2 * ts[9]));
  gxw[2] = o0 * ts[2] + (o1 * ts[6] + (gxw[2] - o2 * ts[10]));
}

// (Round 5's matrix-pipe variant of the pair term — v_mfma_f32_4x4x1 for the pose transform and dL/dX_w: parity-green, 23 us slower —
// is kept as a patch, docs/history/patches/r05_track_pairs_mfma.patch; profiles/r05_track_pairs_mfma_ab.txt.)

static_assert(kTrackTile % 2 == 0, "the source frames of a tile are processed in pairs");

// Sampling inside the pair kernel (fm_track_loss_fused_fwd): the wave samples the points of its own tile's source
// frames in its prologue (every (segment, frame, point) belongs to exactly one tile, so nothing is sampled twice)
// and writes ws / flag for its epilogue and for the backward — the separate track_points launch (0.10 ms at C2,
// a chain of gathers with nothing to overlap) disappears into the first microseconds of 4 000 waves.
struct TrackSampling {
  const float* depth;  // null: ws / flag were filled by fm_track_points
  const float* kinv;
  int depth_frame0, own_first, own_end;  // frames [own_first, own_end) are sources on this rank (frame sharding)
  const int32_t* tap_slot;  // (total, 4) or null: where the compact tap image holds each tap's depth (track_sample_many)
  const float* tap_depth;   // the image, or null: sample the depth images
  float* tap_grad;          // (M) or null: the epilogue stores the unscaled dL/ddepth of every tap that belongs to ONE track point straight
                            // into the compact gradient (the taps several points share are summed by tap_grad_kernel afterwards)
};

// Points per lane (FM_TRACK_PG): with two, a wave covers 128 points and the per-target reduction of the 14 sums (a quarter
// of the loop's instructions with one) is shared by twice the residuals.
#ifndef FM_TRACK_PG
#define FM_TRACK_PG 2
#endif
constexpr int kTrackPG = FM_TRACK_PG;
#ifndef FM_TRACK_AHEAD
#define FM_TRACK_AHEAD 2
#endif
#ifndef FM_TRACK_WAVES  // resident waves per SIMD the register budget is set for (A/B: tools/ab_lib_variants.sh; profiles/r06_track_pairs_closure.txt)
#define FM_TRACK_WAVES (FM_TRACK_PG == 1 ? 4 : 2)
#endif
constexpr int kTrackAhead = FM_TRACK_AHEAD;  // target frames whose (visibility, position) loads are in flight

template <int KIND, bool GRAD>
__global__ void __launch_bounds__(64, FM_TRACK_WAVES) track_pairs_kernel(TrackGeom g, const int32_t* tiles, float* ws, uint8_t* flag,
                                                            const float* ext, const float* tgt, float delta, float ax, float ay, int fmax,
                                                            float* partial, float* gws, TrackSampling smp, int ntiles, int pgroups) {
#ifdef FM_TRACK_LDS_PAD  // (experiments: fewer resident waves per CU)
  __shared__ int lds_pad[FM_TRACK_LDS_PAD / 4];
  if (fmax < 0) lds_pad[threadIdx.x] = 1;
#endif
#ifdef FM_TRACK_CLOCKS
  long long clk[kTrackClockSlots] = {};
#endif
  FM_TCLK(c_begin);
  const float inv_delta = KIND == kHuber ? 1.0f / delta : 0.f;
  // Workgroups go to the eight XCDs round-robin in launch order, and every XCD has its own L2: consecutive launch indices are dealt to
  // the SAME XCD's share of the (tile, point group) list — tile-major, tiles in frame order — so that an XCD's waves sample neighbouring
  // frames (their taps: a few hundred KB of the compact tap image / of the depth images per frame) instead of all of them.
  const unsigned work_items = gridDim.x, per_xcd = (work_items + kXcds - 1) / kXcds;
  const unsigned work = (blockIdx.x % kXcds) * per_xcd + blockIdx.x / kXcds;
  if (work >= (unsigned)(ntiles * pgroups)) return;
  const unsigned tile_index = work / (unsigned)pgroups, group_index = work % (unsigned)pgroups;
  // this wave's slice of the partial-sum workspace: [fmax][14] target role, then [kTrackTile][21] source role
  float* mine = partial + (size_t)work * track_partial_stride(fmax);
  const int sg = tiles[tile_index * 2], fs0 = tiles[tile_index * 2 + 1];
  const int start = g.seg[sg * 4], f = g.seg[sg * 4 + 1], p_count = g.seg[sg * 4 + 2], off = g.seg[sg * 4 + 3];
  int p[kTrackPG], pp[kTrackPG];
  bool active[kTrackPG];
#pragma unroll
  for (int q = 0; q < kTrackPG; ++q) {
    p[q] = (group_index * kTrackPG + q) * blockDim.x + threadIdx.x;
    active[q] = p[q] < p_count;
    pp[q] = active[q] ? p[q] : p_count - 1;  // clamped: loads stay in bounds, results are masked
  }

  // source frames fs0 + 2j (.x) and fs0 + 2j + 1 (.y) share packed registers
  v2f xw[kTrackPG][kTrackTile / 2][3], gxw[kTrackPG][kTrackTile / 2][3];
  v2f live[kTrackPG][kTrackTile / 2];  // 1 when the source role is visible (projection.py:290-294), else 0
#pragma unroll
  for (int q = 0; q < kTrackPG; ++q) {
    float lvs[kTrackTile], xs[kTrackTile][3];
    if (smp.depth != nullptr) {  // sample this point's source frames of the tile (all loads of the tile in flight together)
      int frame[kTrackTile];
      size_t idx[kTrackTile];
      bool want[kTrackTile], lv[kTrackTile];
#pragma unroll
      for (int t = 0; t < kTrackTile; ++t) {
        const int fs = fs0 + t;
        frame[t] = start + fs;
        idx[t] = (size_t)off + (size_t)min(fs, f - 1) * p_count + pp[q];
        want[t] = fs < f && frame[t] >= smp.own_first && frame[t] < smp.own_end;
        if (fs < f && !want[t] && active[q]) flag[idx[t]] = 0;  // another rank's source
      }
      track_sample_many<kTrackTile>(g, smp.depth, smp.depth_frame0, smp.kinv, ext, frame, idx, want, active[q], p_count, pp[q], ws, flag, xs, lv,
                                    smp.tap_slot, smp.tap_depth);
#pragma unroll
      for (int t = 0; t < kTrackTile; ++t) lvs[t] = lv[t] ? 1.f : 0.f;
    } else {
#pragma unroll
      for (int t = 0; t < kTrackTile; ++t) {
        const int fs = fs0 + t;
        lvs[t] = 0.f;
        xs[t][0] = xs[t][1] = xs[t][2] = 0.f;
        if (active[q] && fs < f) {
          const size_t is = (size_t)off + (size_t)fs * p_count + p[q];
          if (flag[is] != 0) {
            lvs[t] = 1.f;
            const float* w9 = ws_plane(ws, is - p[q], p_count, p[q]);
            xs[t][0] = w9[3 * (size_t)p_count]; xs[t][1] = w9[4 * (size_t)p_count]; xs[t][2] = w9[5 * (size_t)p_count];
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < kTrackTile; ++t) {
      const float lv = lvs[t], x0 = lv != 0.f ? xs[t][0] : 0.f, x1 = lv != 0.f ? xs[t][1] : 0.f, x2 = lv != 0.f ? xs[t][2] : 0.f;
      if (t & 1) {
        live[q][t / 2].y = lv; xw[q][t / 2][0].y = x0; xw[q][t / 2][1].y = x1; xw[q][t / 2][2].y = x2;
      } else {
        live[q][t / 2].x = lv; xw[q][t / 2][0].x = x0; xw[q][t / 2][1].x = x1; xw[q][t / 2][2].x = x2;
      }
      gxw[q][t / 2][0] = gxw[q][t / 2][1] = gxw[q][t / 2][2] = 0.f;
    }
  }

  // The target's visibility and position are prefetched kTrackAhead iterations ahead (2; 1, 4 and 8 measure the same: the loop is
  // bound by VALU issue, 2.1 us per target with two waves on a SIMD — tools/track_clocks.py).
  size_t it[kTrackPG];
  uint8_t tv_q[kTrackAhead][kTrackPG];
  float2 gt_q[kTrackAhead][kTrackPG];
#pragma unroll
  for (int q = 0; q < kTrackPG; ++q) {
    it[q] = (size_t)off + pp[q];
#pragma unroll
    for (int d = 0; d < kTrackAhead; ++d) {
      tv_q[d][q] = 0;
      gt_q[d][q] = make_float2(0.f, 0.f);
      if (d < f) {
        tv_q[d][q] = g.vis[it[q] + (size_t)d * p_count];
        gt_q[d][q] = reinterpret_cast<const float2*>(g.xy)[it[q] + (size_t)d * p_count];
      }
    }
    it[q] += (size_t)(kTrackAhead - 1) * p_count;  // (the element the loop's next load reads is one frame further)
  }
  FM_TCLK(c_loop);
  FM_TCLK_ADD(0, c_begin, c_loop);
  for (int ft = 0; ft < f; ++ft) {
    FM_TCLK(c0);
    float tv[kTrackPG];
    float2 gt[kTrackPG];
#pragma unroll
    for (int q = 0; q < kTrackPG; ++q) {
      tv[q] = active[q] && tv_q[0][q] != 0 ? 1.f : 0.f;  // target role needs only the track's visibility (projection.py:290)
      gt[q] = make_float2(gt_q[0][q].x * ax, gt_q[0][q].y * ay);  // (scaled image coordinates: track_pair_term_scaled, fm_pose.h)
#pragma unroll
      for (int d = 0; d + 1 < kTrackAhead; ++d) {
        tv_q[d][q] = tv_q[d + 1][q];
        gt_q[d][q] = gt_q[d + 1][q];
      }
      it[q] += p_count;
      if (ft + kTrackAhead < f) {
        tv_q[kTrackAhead - 1][q] = g.vis[it[q]];
        gt_q[kTrackAhead - 1][q] = reinterpret_cast<const float2*>(g.xy)[it[q]];
      }
    }
    float tg[kTrackTgt];
#pragma unroll
    for (int i = 0; i < kTrackTgt; ++i)  // wave-uniform: scalar loads; the projection rows pre-multiplied by the aspect factors (track_scale_target)
      tg[i] = tgt[(size_t)(start + ft) * kTrackTgt + i] * (i < 4 ? ax : (i < 8 ? ay : 1.f));
#ifdef FM_TRACK_CLOCKS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    FM_TCLK(c1);
    FM_TCLK_ADD(1, c0, c1);
    float a[kTrackSums];
    v2f a2[kTrackSums];
#pragma unroll
    for (int i = 0; i < kTrackSums; ++i) a2[i] = 0.f;
#pragma unroll
    for (int q = 0; q < kTrackPG; ++q) {
#pragma unroll
      for (int j = 0; j < kTrackTile / 2; ++j)
        track_pair_term2<KIND, GRAD>(tg, xw[q][j], gt[q].x, gt[q].y, tv[q] * live[q][j], delta, inv_delta, ax, ay, a2, gxw[q][j]);
    }
#pragma unroll
    for (int i = 0; i < kTrackSums; ++i) a[i] = a2[i].x + a2[i].y;
#ifdef FM_TRACK_CLOCKS
    asm volatile("" :: "v"(a[0]), "v"(a[13]));
#endif
    FM_TCLK(c2);
    FM_TCLK_ADD(2, c1, c2);
    if (GRAD) {
#ifdef FM_TRACK_PLAIN_REDUCE
      wave_sum_lane63_x7<kTrackSums>(a);  // the totals are valid in lane 63
      if (threadIdx.x == kWave - 1) {
#pragma unroll
        for (int i = 0; i < kTrackSums; ++i) mine[(size_t)ft * kTrackSums + i] = a[i] * (i < 4 ? ax : (i < 8 ? ay : 1.f));
      }
#else
      static_assert(kTrackSums <= 16, "the transposing reduction handles 16 values");
      float a16[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) a16[i] = i < kTrackSums ? a[i] : 0.f;
      const float total = wave_transpose_sum16(a16);  // lane L: the wave total of sum L >> 2
      const int lane = threadIdx.x & (kWave - 1);
      const int which = lane >> 2;  // (S0 and S1 were accumulated divided by ax / ay: back to the un-scaled sums track_frame_grads reads)
      if ((lane & 3) == 0 && which < kTrackSums) mine[(size_t)ft * kTrackSums + which] = total * (which < 4 ? ax : (which < 8 ? ay : 1.f));
#endif
    } else {
      const float lc[2] = {a[12], a[13]};
      wave_store<2>(lc, mine + (size_t)ft * kTrackSums + 12);
    }
    FM_TCLK(c3);
    FM_TCLK_ADD(3, c2, c3);
  }
  FM_TCLK(c_epi);

  if (GRAD) {
#pragma unroll
    for (int t = 0; t < kTrackTile; ++t) {
      const int fs = fs0 + t;
      if (fs >= f) break;  // wave-uniform
      float b[21];
#pragma unroll
      for (int i = 0; i < 21; ++i) b[i] = 0.f;
      Pose e;
      load_pose44(ext + (size_t)(start + fs) * 16, e);
#pragma unroll
      for (int q = 0; q < kTrackPG; ++q) {
        if (((t & 1) ? live[q][t / 2].y : live[q][t / 2].x) != 0.f) {
          const size_t is = (size_t)off + (size_t)fs * p_count + p[q];
          float gxyz[3];
          const float gx[3] = {(t & 1) ? gxw[q][t / 2][0].y : gxw[q][t / 2][0].x, (t & 1) ? gxw[q][t / 2][1].y : gxw[q][t / 2][1].x,
                               (t & 1) ? gxw[q][t / 2][2].y : gxw[q][t / 2][2].x};
          float b1[21], w9[kTrackWs];
          const float* wp = ws_plane(ws, is - p[q], p_count, p[q]);
#pragma unroll
          for (int c = 0; c < kTrackWs; ++c) w9[c] = (c < 3 || c >= 6) ? wp[c * (size_t)p_count] : 0.f;  // (xyz and h: what the source role reads)
          track_source_term(e, w9, gx, b1, gxyz);
#pragma unroll
          for (int i = 0; i < 21; ++i) b[i] += b1[i];
          gws[is * 3 + 0] = gxyz[0];
          gws[is * 3 + 1] = gxyz[1];
          gws[is * 3 + 2] = gxyz[2];
          if (smp.tap_grad != nullptr) {  // dL/ddepth at this point's own taps: w_k · <dL/dxyz, K⁻¹·[u_k, v_k, 1]> (as tap_grad_kernel sums it)
            const float2 qq = reinterpret_cast<const float2*>(g.xy)[is];
            const Taps tp = bilinear_taps(qq.x, qq.y, g.height, g.width);
            const int4 s4 = reinterpret_cast<const int4*>(smp.tap_slot)[is];
            int sl[4] = {s4.x, s4.y, s4.z, s4.w};
            Mat3 ki;
            load_mat3(smp.kinv + (size_t)(start + fs) * 9, ki);
            float val[4];
            bool own[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float ray[3];
              ray_dir(ki, pixel_center(tap_col(tp, k), g.width), pixel_center(tap_row(tp, k), g.height), ray);
              val[k] = tp.w[k] * (gxyz[0] * ray[0] + gxyz[1] * ray[1] + gxyz[2] * ray[2]);
              own[k] = sl[k] >= 0 && !(sl[k] & kTapShared);
              sl[k] &= kTapRank;
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {  // the two taps of an image row have neighbouring ranks: one 8-byte store
              if (own[2 * r] && own[2 * r + 1]) {
                const float2 v = make_float2(val[2 * r], val[2 * r + 1]);
                __builtin_memcpy(smp.tap_grad + sl[2 * r], &v, sizeof(float2));
              } else {
                if (own[2 * r]) smp.tap_grad[sl[2 * r]] = val[2 * r];
                if (own[2 * r + 1]) smp.tap_grad[sl[2 * r + 1]] = val[2 * r + 1];
              }
            }
          }
        }
      }
      wave_sum_lane63_x7<21>(b);
      if (threadIdx.x == kWave - 1) {
#pragma unroll
        for (int i = 0; i < 21; ++i) mine[(size_t)fmax * kTrackSums + t * 21 + i] = b[i];
      }
    }
  }
#ifdef FM_TRACK_CLOCKS
  {
    const long long c_end = wall_clock64();
    clk[4] = c_end - c_epi;
    clk[5] = c_end - c_begin;
    clk[6] = (long long)f | ((c_begin & 0xffffffffffLL) << 8);  // (targets, and when the wave began: the host lines the waves up in time)
    // where the wave ran: HW_ID (wave / SIMD / CU / SH / SE) and XCC_ID, so that the host can count resident waves per SIMD
    clk[7] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    const unsigned wid = work;
    if (threadIdx.x == 0 && wid < (unsigned)kTrackClockWaves) {
#pragma unroll
      for (int i = 0; i < kTrackClockSlots; ++i) fm_track_clock_buffer[wid][i] = clk[i];
    }
  }
#endif
}

// Per frame: sum the partials of every wave that touched it (fp64, fixed order: bit-reproducible).
// grid: frames.  Each thread walks (tile, point group) entries, keeps the 14 target-role and 21
// source-role sums of the entries covering this frame, then the block reduces them through LDS.
__global__ void __launch_bounds__(256) track_reduce_kernel(const int32_t* seg, const int32_t* tiles, int ntiles, int pgroups, int fmax,
                                                           const float* partial, int grad, double* acc, double* acc2) {
  constexpr int kVals = kTrackSums + 21;
  __shared__ double red[kVals][kWave + 1];
  const int frame = blockIdx.x;
  const size_t stride = track_partial_stride(fmax);
  double v[kVals];
#pragma unroll
  for (int i = 0; i < kVals; ++i) v[i] = 0.0;
  for (int e = threadIdx.x; e < ntiles * pgroups; e += blockDim.x) {
    const int j = e / pgroups;
    const int sg = tiles[j * 2], fs0 = tiles[j * 2 + 1];
    const int ft = frame - seg[sg * 4];
    if (ft < 0 || ft >= seg[sg * 4 + 1]) continue;
    const float* w = partial + (size_t)e * stride;
    const float* tw = w + (size_t)ft * kTrackSums;
    if (grad) {
#pragma unroll
      for (int i = 0; i < 12; ++i) v[i] += (double)tw[i];
    }
    v[12] += (double)tw[12];
    v[13] += (double)tw[13];
    const int t = ft - fs0;  // this frame as a source of the tile
    if (grad && t >= 0 && t < kTrackTile) {
      const float* sw = w + (size_t)fmax * kTrackSums + t * 21;
#pragma unroll
      for (int i = 0; i < 21; ++i) v[kTrackSums + i] += (double)sw[i];
    }
  }
  // 256 -> 64 partial sums per value in registers/LDS, then one wave finishes each value
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  for (int w4 = 0; w4 < 4; ++w4) {
    if (wave == w4) {
#pragma unroll
      for (int i = 0; i < kVals; ++i) red[i][lane] = (w4 == 0 ? 0.0 : red[i][lane]) + v[i];
    }
    __syncthreads();
  }
  if (threadIdx.x < kVals) {
    double tot = 0.0;
    for (int l = 0; l < kWave; ++l) tot += red[threadIdx.x][l];
    const int i = threadIdx.x;
    // acc: [0..11] S, [12..17] unused (0), [18] Σρ, [19] count;  acc2: [0..20], [21..23] unused
    if (i < 12) acc[(size_t)frame * kTrackAccStride + i] = tot;
    else if (i < kTrackSums) acc[(size_t)frame * kTrackAccStride + i + 6] = tot;
    else if (grad) acc2[(size_t)frame * kTrackAcc2Stride + (i - kTrackSums)] = tot;
  }
  if (threadIdx.x >= 12 && threadIdx.x < 18) acc[(size_t)frame * kTrackAccStride + threadIdx.x] = 0.0;
  if (grad && threadIdx.x >= 21 && threadIdx.x < kTrackAcc2Stride) acc2[(size_t)frame * kTrackAcc2Stride + threadIdx.x] = 0.0;
}

// loss[0] = weight·Σρ/max(count,1); scale[0] = weight/max(count,1); scale[1] = count
__global__ void track_finalize_fwd_kernel(const double* acc, int frames, float weight, float* loss, float* scale, double* totals) {
  double sum = 0.0, cnt = 0.0;  // one wave
  for (int fr = threadIdx.x; fr < frames; fr += kWave) {
    sum += acc[(size_t)fr * kTrackAccStride + 18];
    cnt += acc[(size_t)fr * kTrackAccStride + 19];
  }
  sum = wave_sum(sum);
  cnt = wave_sum(cnt);
  if (threadIdx.x != 0) return;
  if (totals) {
    totals[0] = sum;
    totals[1] = cnt;
  }
  const double den = cnt != 0.0 ? cnt : 1.0;  // `valid_sum or 1` (loss_tracking.py:61)
  loss[0] = (float)((double)weight * sum / den);
  scale[0] = (float)((double)weight / den);
  scale[1] = (float)cnt;
}

// Scatter of the per-point surface gradients gws (total,3) into dL/ddepth through the
// bilinear taps: xyz = Σ_k w_k · z_k · Kinv·[u_k, v_k, 1].  Separate launch so the caller
// can aim it at whichever dense buffer will finally hold dL/ddepth.
__global__ void __launch_bounds__(256) track_scatter_kernel(TrackGeom g, const uint8_t* flag, const float* gws, const float* kinv,
                                                            const float* scale, const float* upstream, int depth_frame0,
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
  const float sc = scale[0] * (upstream ? upstream[0] : 1.f);
  const float gx = gws[is * 3] * sc, gy = gws[is * 3 + 1] * sc, gz = gws[is * 3 + 2] * sc;
  const float2 q = reinterpret_cast<const float2*>(g.xy)[is];
  const Taps t = bilinear_taps(q.x, q.y, g.height, g.width);
  float* gd = grad_depth + (size_t)(frame_s - depth_frame0) * g.height * g.width;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    if (!t.in[kk]) continue;
    const int tc = tap_col(t, kk), tr = tap_row(t, kk);
    float ray[3];
    ray_dir(ki, pixel_center(tc, g.width), pixel_center(tr, g.height), ray);
    atomicAdd(gd + tr * g.width + tc, t.w[kk] * (gx * ray[0] + gy * ray[1] + gz * ray[2]));
  }
}

// The scatter pattern is static (track positions and visibility are inputs of the optimisation):
// tap k of source point i lands on pixel keys[4i+k] = frame·H·W + row·W + col with weight
// weights[4i+k]; -1 marks a tap that contributes nothing.  Same arithmetic as track_scatter_kernel.
__global__ void __launch_bounds__(256) track_scatter_plan_kernel(TrackGeom g, int64_t* keys, float* weights) {
  const int sg = g.blocks[blockIdx.x * 2], fs = g.blocks[blockIdx.x * 2 + 1];
  const int start = g.seg[sg * 4], p_count = g.seg[sg * 4 + 2], off = g.seg[sg * 4 + 3];
  const int p = blockIdx.y * blockDim.x + threadIdx.x;
  if (p >= p_count) return;
  const size_t is = (size_t)off + (size_t)fs * p_count + p;
  const float2 q = reinterpret_cast<const float2*>(g.xy)[is];
  const bool live = g.vis[is] != 0 && q.x >= 0.f && q.y >= 0.f && q.x < 1.f && q.y < 1.f;  // the flag of track_points_kernel
  const Taps t = bilinear_taps(q.x, q.y, g.height, g.width);
  const int64_t base = (int64_t)(start + fs) * g.height * g.width;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const bool used = live && t.in[kk];
    keys[is * 4 + kk] = used ? base + (int64_t)tap_row(t, kk) * g.width + tap_col(t, kk) : (int64_t)-1;
    weights[is * 4 + kk] = used ? t.w[kk] : 0.f;
  }
}

// A planned scatter executed as a gather over the touched pixels: one thread owns one pixel of
// dL/ddepth and sums the contributions the plan lists for it,
//   grad_depth[pixel] += s · Σ_e weights[e] · < vectors[entries[e]], K⁻¹(frame)·[u, v, 1] >,
// — no atomics (device-scope float atomics run at the memory side on this part, ≈10-26 G/s however
// hot the line), one read-modify-write per pixel, and a result that does not depend on scheduling.
// Serves the tracking loss (vectors = dL/dxyz per source point) and the sparse Procrustes backward
// (vectors = dL/dq, dL/dp per correspondence).
__global__ void __launch_bounds__(256) depth_gather_kernel(const float* vectors, const int64_t* pixels, const int32_t* first,
                                                           const int32_t* entries, const float* weights, long count, const float* kinv,
                                                           const float* scale, const float* upstream, int height, int width,
                                                           long frame0, float* grad_depth) {
  const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= count) return;
  const int64_t n = (int64_t)height * width;
  const int64_t key = pixels[m];
  const int64_t frame = key / n;
  const int px = (int)(key - frame * n);
  const int row = px / width, col = px - row * width;
  Mat3 ki;
  load_mat3(kinv + (size_t)frame * 9, ki);
  float ray[3];
  ray_dir(ki, pixel_center(col, width), pixel_center(row, height), ray);
  const float sc = (scale ? scale[0] : 1.f) * (upstream ? upstream[0] : 1.f);
  float sum = 0.f;
  for (int e = first[m]; e < first[m + 1]; ++e) {
    const float* v = vectors + (size_t)entries[e] * 3;
    const float gx = v[0] * sc, gy = v[1] * sc, gz = v[2] * sc;
    sum += weights[e] * (gx * ray[0] + gy * ray[1] + gz * ray[2]);
  }
  grad_depth[(size_t)(frame - frame0) * n + px] += sum;
}

// The planned gather again, but INTO THE COMPACT TAP IMAGE instead of the depth gradient: tap_grad[m] = Σ_e weights[e]·<vectors[entries[e]],
// K⁻¹(frame)·[u, v, 1]> for the m-th touched pixel — unscaled; the fused flow pass (fm_flow_loss_fused_taps) adds scale·tap_grad at the
// pixel when it writes dL/ddepth, so the tracking loss never read-modify-writes a cold line of the gradient image.
__global__ void __launch_bounds__(256) tap_grad_kernel(const float* vectors, const int64_t* pixels, const int32_t* first, const int32_t* entries,
                                                       const float* weights, long count, const int32_t* ranks, const float* kinv, int height,
                                                       int width, float* tap_grad) {
  long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= count) return;
  if (ranks) m = ranks[m];  // (only the taps that several track points share: the others were stored by track_pairs' epilogue)
  const int64_t n = (int64_t)height * width;
  const int64_t key = pixels[m];
  const int64_t frame = key / n;
  const int px = (int)(key - frame * n);
  const int row = px / width, col = px - row * width;
  Mat3 ki;
  load_mat3(kinv + (size_t)frame * 9, ki);
  float ray[3];
  ray_dir(ki, pixel_center(col, width), pixel_center(row, height), ray);
  float sum = 0.f;
  for (int e = first[m]; e < first[m + 1]; ++e) {
    const float* v = vectors + (size_t)entries[e] * 3;
    sum += weights[e] * (v[0] * ray[0] + v[1] * ray[1] + v[2] * ray[2]);
  }
  tap_grad[m] = sum;
}

// grad_depth[pixels[m]] += scale[0]·(plus − minus)·tap_grad[m], nothing at all when the factor is zero (every block leaves after
// three scalar loads).  The corrections of the tap exchange: the flow pass added scale·tap_grad assuming the tracking loss would reach
// backward() with the same upstream gradient as the flow loss; `plus` = the tracking loss's upstream gradient (null: its backward never
// ran), `minus` = what the flow pass's share was multiplied by (null: it was never delivered).
__global__ void __launch_bounds__(256) tap_grad_apply_kernel(const float* tap_grad, const int64_t* pixels, long count, const float* scale,
                                                             const float* plus, const float* minus, float* grad_depth, int* mismatch) {
  const float factor = scale[0] * ((plus ? plus[0] : 0.f) - (minus ? minus[0] : 0.f));
  if (factor == 0.f) return;
  const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m == 0 && mismatch) *mismatch = 1;  // (an in-pass Adam update already used the absorbed gradient at factor 1: FusedAdam.step reports it)
  if (m >= count) return;
  grad_depth[pixels[m]] += factor * tap_grad[m];
}

// The planned gather (scale = upstream = 1, frame0 = 0) with the map of the K⁻¹ gradient back to K in the same launch:
// the blocks past the gather's do dK = [g_k] − K⁻ᵀ·dK⁻¹·K⁻ᵀ for 64 frames each.
__global__ void __launch_bounds__(256) depth_gather_kgrad_kernel(const float* vectors, const int64_t* pixels, const int32_t* first,
                                                                 const int32_t* entries, const float* weights, long count, const float* kinv,
                                                                 int height, int width, float* grad_depth, unsigned gather_blocks,
                                                                 const double* kinv_acc, int frames_k, float* g_k, int accumulate) {
  if (blockIdx.x >= gather_blocks) {
    const int i = (int)(blockIdx.x - gather_blocks) * 64 + (int)threadIdx.x;
    if (threadIdx.x >= 64 || i >= frames_k) return;
    double gk[9];
    kinv_grad_to_k(kinv_acc + (size_t)i * 9, kinv + (size_t)i * 9, gk);
    for (int e = 0; e < 9; ++e) {
      float* o = g_k + (size_t)i * 9 + e;
      *o = (accumulate ? *o : 0.f) + (float)gk[e];
    }
    return;
  }
  const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= count) return;
  const int64_t n = (int64_t)height * width;
  const int64_t key = pixels[m];
  const int64_t frame = key / n;
  const int px = (int)(key - frame * n);
  const int row = px / width, col = px - row * width;
  Mat3 ki;
  load_mat3(kinv + (size_t)frame * 9, ki);
  float ray[3];
  ray_dir(ki, pixel_center(col, width), pixel_center(row, height), ray);
  float sum = 0.f;
  for (int e = first[m]; e < first[m + 1]; ++e) {
    const float* v = vectors + (size_t)entries[e] * 3;
    sum += weights[e] * (v[0] * ray[0] + v[1] * ray[1] + v[2] * ray[2]);
  }
  grad_depth[(size_t)frame * n + px] += sum;
}

// dL/dE and dL/dK per frame from the two accumulators.
__global__ void __launch_bounds__(64) track_finalize_bwd_kernel(const double* acc, const double* acc2, const float* scale, const float* upstream,
                                          const float* ext_inv, const float* k, const float* kinv, int frames, float* g_ext,
                                          float* g_k) {
  const int fr = blockIdx.x * blockDim.x + threadIdx.x;
  if (fr >= frames) return;
  const double sc = (double)scale[0] * (upstream ? (double)upstream[0] : 1.0);
  track_frame_grads(acc + (size_t)fr * kTrackAccStride, acc2 + (size_t)fr * kTrackAcc2Stride, sc, ext_inv + (size_t)fr * 16,
                    k + (size_t)fr * 9, kinv + (size_t)fr * 9, g_ext + (size_t)fr * 16, g_k + (size_t)fr * 9);
}

__global__ void __launch_bounds__(64) inv4_kernel(const float* m, int count, float* out) {
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

#ifdef FM_TRACK_CLOCKS
int fm_debug_track_clocks(long long* host_out, int waves) {  // (waves, kTrackClockSlots) accumulated wall_clock64 deltas (100 MHz) of the last track_pairs launch
  if (waves > kTrackClockWaves) waves = kTrackClockWaves;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(fm_track_clock_buffer), sizeof(long long) * waves * kTrackClockSlots) == hipSuccess ? kTrackClockSlots : -1;
}
#endif

int fm_extrinsics_inverse(const float* ext, int count, float* inv, void* stream) {
  FM_CHECK_ARG(ext && inv && count >= 1);
  hipLaunchKernelGGL(inv4_kernel, dim3((count + 63) / 64), dim3(64), 0, (hipStream_t)stream, ext, count, inv);
  FM_LAUNCH_STATUS();
}

int fm_track_points(const float* depth, int depth_frame0, const float* kinv, const float* ext, const float* ext_inv, const float* k,
                    int frames, const float* xy, const uint8_t* vis, const int32_t* seg, const int32_t* blocks, int nblocks, int pmax,
                    int height, int width, float* ws, uint8_t* flag, float* tgt, void* stream) {
  FM_CHECK_ARG(depth && kinv && ext && ext_inv && k && xy && vis && seg && blocks && ws && flag && tgt);
  FM_CHECK_ARG(nblocks >= 1 && pmax >= 1 && frames >= 1 && depth_frame0 >= 0);
  hipStream_t st = (hipStream_t)stream;
  TrackGeom g{xy, vis, seg, blocks, height, width};
  hipLaunchKernelGGL(track_points_kernel, dim3(nblocks, (pmax + 255) / 256), dim3(256), 0, st, g, depth, depth_frame0, kinv, ext,
                     ws, flag);
  hipLaunchKernelGGL(track_targets_kernel, dim3((frames + 63) / 64), dim3(64), 0, st, ext_inv, k, frames, tgt);
  FM_LAUNCH_STATUS();
}

static int track_loss_launch(float* ws, uint8_t* flag, const float* xy, const uint8_t* vis, const int32_t* seg, const int32_t* tiles,
                             int ntiles, int pmax, int fmax, const float* ext, const float* tgt, int frames, int height, int width,
                             int mapping_kind, float delta, float aspect_x, float aspect_y, float weight, float* partial, double* acc,
                             float* loss, float* scale, double* totals, float* gws, double* acc2, TrackSampling smp, hipStream_t st) {
  TrackGeom g{xy, vis, seg, nullptr, height, width};
  const int pgroups = (pmax + kWave * kTrackPG - 1) / (kWave * kTrackPG);  // (the caller sized `partial` for groups of kWave points: enough)
  const unsigned items = (unsigned)ntiles * (unsigned)pgroups;
  const dim3 grid(((items + kXcds - 1) / kXcds) * kXcds);  // (a multiple of the XCD count: the kernel deals launch indices to XCD shares)
#define FM_TRACK_LAUNCH(K)                                                                                                          \
  do {                                                                                                                               \
    if (gws)                                                                                                                         \
      hipLaunchKernelGGL((track_pairs_kernel<K, true>), grid, dim3(kWave), 0, st, g, tiles, ws, flag, ext, tgt, delta,              \
                         aspect_x, aspect_y, fmax, partial, gws, smp, ntiles, pgroups);                                              \
    else                                                                                                                             \
      hipLaunchKernelGGL((track_pairs_kernel<K, false>), grid, dim3(kWave), 0, st, g, tiles, ws, flag, ext, tgt, delta,             \
                         aspect_x, aspect_y, fmax, partial, gws, smp, ntiles, pgroups);                                              \
  } while (0)
  if (mapping_kind == kHuber) FM_TRACK_LAUNCH(kHuber);
  else if (mapping_kind == kL1) FM_TRACK_LAUNCH(kL1);
  else FM_TRACK_LAUNCH(kL2);
#undef FM_TRACK_LAUNCH
  hipLaunchKernelGGL(track_reduce_kernel, dim3(frames), dim3(256), 0, st, seg, tiles, ntiles, pgroups, fmax, partial, gws ? 1 : 0, acc,
                     acc2);
  hipLaunchKernelGGL(track_finalize_fwd_kernel, dim3(1), dim3(kWave), 0, st, acc, frames, weight, loss, scale, totals);
  FM_LAUNCH_STATUS();
}

int fm_track_loss_fwd(const float* ws, const uint8_t* flag, const float* xy, const uint8_t* vis, const int32_t* seg,
                      const int32_t* tiles, int ntiles, int pmax, int fmax, const float* ext, const float* tgt, int frames, int height,
                      int width, int mapping_kind, float delta, float aspect_x, float aspect_y, float weight, float* partial,
                      double* acc, float* loss, float* scale, double* totals, float* gws, double* acc2, void* stream) {
  FM_CHECK_ARG(ws && flag && xy && vis && seg && tiles && ext && tgt && partial && acc && loss && scale);
  FM_CHECK_ARG(ntiles >= 1 && pmax >= 1 && fmax >= 1 && frames >= 1 && mapping_kind >= 0 && mapping_kind <= 2);
  FM_CHECK_ARG((gws == nullptr) == (acc2 == nullptr));
  return track_loss_launch(const_cast<float*>(ws), const_cast<uint8_t*>(flag), xy, vis, seg, tiles, ntiles, pmax, fmax, ext, tgt, frames,
                           height, width, mapping_kind, delta, aspect_x, aspect_y, weight, partial, acc, loss, scale, totals, gws, acc2,
                           TrackSampling{nullptr, nullptr, 0, 0, 0, nullptr, nullptr, nullptr}, (hipStream_t)stream);
}

int fm_track_loss_fused_fwd(const float* depth, int depth_frame0, int own_first, int own_end, const float* kinv, const float* ext,
                            const float* ext_inv, const float* k, int frames, const float* xy, const uint8_t* vis, const int32_t* seg,
                            const int32_t* tiles, int ntiles, int pmax, int fmax, int height, int width, int mapping_kind, float delta,
                            float aspect_x, float aspect_y, float weight, float* ws, uint8_t* flag, float* tgt, float* partial, double* acc,
                            float* loss, float* scale, double* totals, float* gws, double* acc2, void* stream) {
  FM_CHECK_ARG(depth && kinv && ext && ext_inv && k && xy && vis && seg && tiles && ws && flag && tgt && partial && acc && loss && scale);
  FM_CHECK_ARG(ntiles >= 1 && pmax >= 1 && fmax >= 1 && frames >= 1 && mapping_kind >= 0 && mapping_kind <= 2 && depth_frame0 >= 0);
  FM_CHECK_ARG((gws == nullptr) == (acc2 == nullptr) && own_first >= depth_frame0 && own_end >= own_first);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(track_targets_kernel, dim3((frames + 63) / 64), dim3(64), 0, st, ext_inv, k, frames, tgt);
  return track_loss_launch(ws, flag, xy, vis, seg, tiles, ntiles, pmax, fmax, ext, tgt, frames, height, width, mapping_kind, delta, aspect_x,
                           aspect_y, weight, partial, acc, loss, scale, totals, gws, acc2,
                           TrackSampling{depth, kinv, depth_frame0, own_first, own_end, nullptr, nullptr, nullptr}, st);
}

int fm_track_loss_fused_fwd_taps(const float* depth, const float* kinv, const float* ext, const float* ext_inv, const float* k, int frames,
                                 const float* xy, const uint8_t* vis, const int32_t* seg, const int32_t* tiles, int ntiles, int pmax, int fmax,
                                 int height, int width, int mapping_kind, float delta, float aspect_x, float aspect_y, float weight, float* ws,
                                 uint8_t* flag, float* tgt, float* partial, double* acc, float* loss, float* scale, double* totals, float* gws,
                                 double* acc2, const int32_t* tap_slot, const float* tap_depth, const int64_t* plan_pixels,
                                 const int32_t* plan_first, const int32_t* plan_entries, const float* plan_weights, long plan_count,
                                 const int32_t* shared_ranks, long shared_count, float* tap_grad, void* stream) {
  FM_CHECK_ARG(depth && kinv && ext && ext_inv && k && xy && vis && seg && tiles && ws && flag && tgt && partial && acc && loss && scale);
  FM_CHECK_ARG(ntiles >= 1 && pmax >= 1 && fmax >= 1 && frames >= 1 && mapping_kind >= 0 && mapping_kind <= 2);
  FM_CHECK_ARG((gws == nullptr) == (acc2 == nullptr) && (tap_depth == nullptr || tap_slot != nullptr));
  FM_CHECK_ARG(tap_grad == nullptr || (gws && plan_pixels && plan_first && plan_entries && plan_weights && plan_count >= 0));
  FM_CHECK_ARG(shared_ranks == nullptr || (tap_slot != nullptr && shared_count >= 0 && shared_count <= plan_count));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(track_targets_kernel, dim3((frames + 63) / 64), dim3(64), 0, st, ext_inv, k, frames, tgt);
  // with the list of shared taps the epilogue of track_pairs stores every other tap's gradient itself
  float* direct = (tap_grad != nullptr && shared_ranks != nullptr) ? tap_grad : nullptr;
  const int status = track_loss_launch(ws, flag, xy, vis, seg, tiles, ntiles, pmax, fmax, ext, tgt, frames, height, width, mapping_kind, delta,
                                       aspect_x, aspect_y, weight, partial, acc, loss, scale, totals, gws, acc2,
                                       TrackSampling{depth, kinv, 0, 0, frames, tap_slot, tap_depth, direct}, st);
  if (status != FM_OK || tap_grad == nullptr) return status;
  const long count = direct ? shared_count : plan_count;
  if (count == 0) return status;
  hipLaunchKernelGGL(tap_grad_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, gws, plan_pixels, plan_first, plan_entries,
                     plan_weights, count, direct ? shared_ranks : nullptr, kinv, height, width, tap_grad);
  FM_LAUNCH_STATUS();
}

int fm_tap_grad_apply(const float* tap_grad, const int64_t* pixels, long count, const float* scale, const float* upstream_plus,
                      const float* upstream_minus, float* grad_depth, int* mismatch_flag, void* stream) {
  FM_CHECK_ARG(count >= 0 && (count == 0 || (tap_grad && pixels && grad_depth)) && scale);
  if (count == 0) return FM_OK;
  hipLaunchKernelGGL(tap_grad_apply_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tap_grad, pixels, count, scale,
                     upstream_plus, upstream_minus, grad_depth, mismatch_flag);
  FM_LAUNCH_STATUS();
}

int fm_track_loss_bwd(const double* acc, const double* acc2, const float* scale, const float* upstream, const float* ext_inv,
                      const float* k, const float* kinv, int frames, float* g_ext, float* g_k, void* stream) {
  FM_CHECK_ARG(acc && acc2 && scale && ext_inv && k && kinv && g_ext && g_k && frames >= 1);
  hipLaunchKernelGGL(track_finalize_bwd_kernel, dim3((frames + 63) / 64), dim3(64), 0, (hipStream_t)stream, acc, acc2, scale, upstream,
                     ext_inv, k, kinv, frames, g_ext, g_k);
  FM_LAUNCH_STATUS();
}

int fm_track_scatter(const float* gws, const uint8_t* flag, const float* xy, const uint8_t* vis, const int32_t* seg,
                     const int32_t* blocks, int nblocks, int pmax, const float* kinv, const float* scale, const float* upstream,
                     int height, int width, int depth_frame0, float* grad_depth, void* stream) {
  FM_CHECK_ARG(gws && flag && xy && vis && seg && blocks && kinv && scale && grad_depth && nblocks >= 1 && pmax >= 1 && depth_frame0 >= 0);
  TrackGeom g{xy, vis, seg, blocks, height, width};
  hipLaunchKernelGGL(track_scatter_kernel, dim3(nblocks, (pmax + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, flag, gws, kinv,
                     scale, upstream, depth_frame0, grad_depth);
  FM_LAUNCH_STATUS();
}

int fm_track_scatter_plan(const float* xy, const uint8_t* vis, const int32_t* seg, const int32_t* blocks, int nblocks, int pmax, int height,
                          int width, int64_t* keys, float* weights, void* stream) {
  FM_CHECK_ARG(xy && vis && seg && blocks && keys && weights && nblocks >= 1 && pmax >= 1 && height >= 1 && width >= 1);
  TrackGeom g{xy, vis, seg, blocks, height, width};
  hipLaunchKernelGGL(track_scatter_plan_kernel, dim3(nblocks, (pmax + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, keys, weights);
  FM_LAUNCH_STATUS();
}

int fm_depth_gather(const float* vectors, const int64_t* pixels, const int32_t* first, const int32_t* entries, const float* weights,
                    long count, const float* kinv, const float* scale, const float* upstream, int height, int width, long frame0,
                    float* grad_depth, void* stream) {
  FM_CHECK_ARG(vectors && pixels && first && entries && weights && kinv && grad_depth && count >= 0 && frame0 >= 0);
  if (count == 0) return FM_OK;
  hipLaunchKernelGGL(depth_gather_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vectors, pixels, first,
                     entries, weights, count, kinv, scale, upstream, height, width, frame0, grad_depth);
  FM_LAUNCH_STATUS();
}

int fm_depth_gather_kgrad(const float* vectors, const int64_t* pixels, const int32_t* first, const int32_t* entries, const float* weights,
                          long count, const float* kinv, int height, int width, float* grad_depth, const double* kinv_acc, int frames_k,
                          float* g_k, int accumulate, void* stream) {
  FM_CHECK_ARG(kinv && count >= 0 && frames_k >= 0 && (count == 0 || (vectors && pixels && first && entries && weights && grad_depth)));
  FM_CHECK_ARG(frames_k == 0 || (kinv_acc && g_k));
  const unsigned gather_blocks = (unsigned)((count + 255) / 256), k_blocks = (unsigned)((frames_k + 63) / 64);
  if (gather_blocks + k_blocks == 0) return FM_OK;
  hipLaunchKernelGGL(depth_gather_kgrad_kernel, dim3(gather_blocks + k_blocks), dim3(256), 0, (hipStream_t)stream, vectors, pixels, first, entries,
                     weights, count, kinv, height, width, grad_depth, gather_blocks, kinv_acc, frames_k, g_k, accumulate);
  FM_LAUNCH_STATUS();
}

}  // extern "C"
