// Frame sharding (flowmap_amd/sharding.py): the local work of the halo exchange, as four small launches.
//
// Under frame-pair sharding the boundary frame of a shard is shared with the neighbour and both copies need the sum of the two
// partial dL/ddepth.  With FrameShard.enable_early_halo the dense part travels when the flow loss's forward pass ends and only a
// sparse correction — the values at the pixels the Procrustes fit / the tracks touched — after backward.  Done with torch's
// indexing operators that was ~14 launches per step (two copies, four gathers, two subtractions, two adds, two index_adds, ...),
// 25-40 us on the one-GPU proxy: about what the early exchange hides.  Here: one launch per stage, both boundaries at once.
//   fm_halo_copy     sent[side]      = grad[frame(side)]                                (what the link reads; backward adds to grad later)
//   fm_halo_delta    out[side][i]    = grad[frame(side)][px[side][i]] − sent[side][px[side][i]]     (after backward)
//   fm_halo_add      grad[frame(side)] += dense[side]                                   (the neighbour's dense part)
//   fm_halo_scatter  grad[frame(side)][px[side][i]] += values[side][i]                  (the neighbour's sparse part; px distinct)
// side 0 = first local frame (neighbour rank−1), side 1 = last local frame (rank+1); a NULL buffer switches a side off.
// There is no reference counterpart: the reference has no sharding (flowmap/overfit.py:94-108 replicates the video per rank).
#include "../../include/flowmap_hip.h"
#include "fm_device.h"
#include "fm_pose.h"

namespace fm {

// The ghost term (FrameShard.enable_ghost_halo): instead of RECEIVING the neighbour's dense dL/ddepth of the shared frame (3.7 MB per
// boundary and direction at 720p), a rank EVALUATES it.  The neighbour's dense part is one flow-loss direction of one pair — the
// backward term of pair L−1 (neighbour rank−1) or the forward term of pair L (neighbour rank+1), with frame L as its source: a function
// of depth[L] (held here), the constant flow and mask of that pair (handed over once, at set-up), K, and the pair's 4x4 pose, which
// the neighbour's Procrustes fit produces each step: 64 bytes travel.  Same per-pixel arithmetic as the fused flow pass (flow_term_fast,
// fm_math.h); only dL/ddepth is kept — the loss value and the pose / intrinsics sums of the term belong to the neighbour, who
// evaluates it too.  blockIdx.y = side (0: first local frame, backward term; 1: last local frame, forward term).
struct GhostSide {
  const float* depth;  // (H,W) the shared frame's depth
  const float* pose;   // (4,4) source camera -> destination camera
  const float* flow;   // (H,W,2) the ghost pair's flow for this direction
  const float* mask;   // (H,W)
  float* grad;         // (H,W) dL/ddepth of the shared frame: the term's part is ADDED
};
template <int KIND>
__global__ void __launch_bounds__(256) flow_ghost_kernel(GhostSide s0, GhostSide s1, const float* kinv, const float* k_dst, const float* norm,
                                                         const float* upstream, int height, int width, float delta, float ax, float ay) {
  const GhostSide s = blockIdx.y == 0 ? s0 : s1;
  if (s.grad == nullptr) return;
  Mat3 ki, kd;
  Pose t;
  load_mat3(kinv, ki);
  load_mat3(k_dst, kd);
  load_pose44(s.pose, t);
  DirConst d;
  make_dir(t, ki, kd, ax, ay, d);
  const float scale = norm[0] * (upstream ? upstream[0] : 1.f);
  const float inv_delta = KIND == kHuber ? 1.0f / delta : 0.f;
  const long n = (long)height * width;
  for (long px = (long)blockIdx.x * blockDim.x + threadIdx.x; px < n; px += (long)gridDim.x * blockDim.x) {
    const int row = (int)(px / width), col = (int)(px - (long)row * width);
    const float u = pixel_center(col, width), v = pixel_center(row, height);
    const float z = s.depth[px];
    float acc[kFlowAcc], gz = 0.f;
#pragma unroll
    for (int i = 0; i < kFlowAcc; ++i) acc[i] = 0.f;
    const float2 fl = reinterpret_cast<const float2*>(s.flow)[px];
    flow_term_fast<KIND, true>(d, fmaf(d.a1, v, d.a2), fmaf(d.b1, v, d.b2), fmaf(d.c1, v, d.c2), z, u, z * u, z * v, u * ax, v * ay, fl.x, fl.y,
                               s.mask[px], scale, delta, inv_delta, ax, ay, acc, gz);
    s.grad[px] += gz;
  }
}

typedef float v4f_s __attribute__((ext_vector_type(4)));

// blockIdx.y = side.  16-byte path when the frame size allows.
__global__ void __launch_bounds__(256) halo_copy_kernel(const float* g0, const float* g1, float* s0, float* s1, long n, int vec) {
  const float* src = blockIdx.y == 0 ? g0 : g1;
  float* dst = blockIdx.y == 0 ? s0 : s1;
  if (dst == nullptr) return;
  const long stride = (long)gridDim.x * blockDim.x, tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    for (long i = tid; i < n / 4; i += stride) reinterpret_cast<v4f_s*>(dst)[i] = reinterpret_cast<const v4f_s*>(src)[i];
  } else {
    for (long i = tid; i < n; i += stride) dst[i] = src[i];
  }
}

__global__ void __launch_bounds__(256) halo_add_kernel(float* g0, float* g1, const float* d0, const float* d1, long n, int vec) {
  float* dst = blockIdx.y == 0 ? g0 : g1;
  const float* src = blockIdx.y == 0 ? d0 : d1;
  if (src == nullptr) return;
  const long stride = (long)gridDim.x * blockDim.x, tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    for (long i = tid; i < n / 4; i += stride) {
      v4f_s a = reinterpret_cast<v4f_s*>(dst)[i];
      const v4f_s b = reinterpret_cast<const v4f_s*>(src)[i];
      a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
      reinterpret_cast<v4f_s*>(dst)[i] = a;
    }
  } else {
    for (long i = tid; i < n; i += stride) dst[i] += src[i];
  }
}

__global__ void __launch_bounds__(256) halo_delta_kernel(const float* g0, const float* g1, const float* s0, const float* s1, const int64_t* px0,
                                                         const int64_t* px1, long n0, long n1, float* o0, float* o1) {
  const bool first = blockIdx.y == 0;
  const float* g = first ? g0 : g1;
  const float* s = first ? s0 : s1;
  const int64_t* px = first ? px0 : px1;
  float* o = first ? o0 : o1;
  const long count = first ? n0 : n1;
  if (o == nullptr) return;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) o[i] = g[px[i]] - s[px[i]];
}

// The ghost halo's forward end in one launch.  blockIdx.y = side: base[i] = the boundary frame's gradient at the side's touched pixels (all
// that fm_halo_delta_sparse will ask for: a gather of a few thousand values instead of fm_halo_copy's two whole frames); block (0, 0)
// also copies what the ghost terms are evaluated from / what is sent — the boundary pairs' poses, K and K^-1 of the first frame — into `pack`:
//   [t_fwd[0] | t_bwd[pairs-1] | t_bwd[0] | t_fwd[pairs-1] | K | K^-1]  (4 x 16 + 2 x 9 floats)
constexpr int kGhostPack = 4 * 16 + 2 * 9;
__global__ void __launch_bounds__(256) halo_ghost_begin_kernel(const float* g0, const float* g1, const int64_t* px0, const int64_t* px1, long n0, long n1,
                                                               float* b0, float* b1, const float* t_fwd, const float* t_bwd, int pairs, const float* k,
                                                               const float* kinv, float* pack) {
  const bool first = blockIdx.y == 0;
  if (first && blockIdx.x == 0 && pack != nullptr && threadIdx.x < kGhostPack) {
    const int t = threadIdx.x;
    const size_t last = (size_t)(pairs - 1) * 16;
    float v;
    if (t < 16) v = t_fwd[t];
    else if (t < 32) v = t_bwd[last + (t - 16)];
    else if (t < 48) v = t_bwd[t - 32];
    else if (t < 64) v = t_fwd[last + (t - 48)];
    else if (t < 73) v = k[t - 64];
    else v = kinv[t - 73];
    pack[t] = v;
  }
  const float* g = first ? g0 : g1;
  const int64_t* px = first ? px0 : px1;
  float* b = first ? b0 : b1;
  const long count = first ? n0 : n1;
  if (b == nullptr) return;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) b[i] = g[px[i]];
}

// fm_halo_delta against that compact baseline: out[i] = grad[boundary frame][px[i]] - base[i]
__global__ void __launch_bounds__(256) halo_delta_sparse_kernel(const float* g0, const float* g1, const float* b0, const float* b1, const int64_t* px0,
                                                                const int64_t* px1, long n0, long n1, float* o0, float* o1) {
  const bool first = blockIdx.y == 0;
  const float* g = first ? g0 : g1;
  const float* b = first ? b0 : b1;
  const int64_t* px = first ? px0 : px1;
  float* o = first ? o0 : o1;
  const long count = first ? n0 : n1;
  if (o == nullptr) return;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) o[i] = g[px[i]] - b[i];
}

__global__ void __launch_bounds__(256) halo_scatter_kernel(float* g0, float* g1, const int64_t* px0, const int64_t* px1, const float* v0,
                                                           const float* v1, long n0, long n1) {
  const bool first = blockIdx.y == 0;
  float* g = first ? g0 : g1;
  const int64_t* px = first ? px0 : px1;
  const float* v = first ? v0 : v1;
  const long count = first ? n0 : n1;
  if (v == nullptr) return;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) g[px[i]] += v[i];
}

static inline unsigned halo_blocks(long work) {
  long blocks = (work + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  return (unsigned)blocks;
}
static inline bool halo_aligned(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace fm

using namespace fm;

extern "C" {

int fm_flow_ghost_terms(const float* depth_first, const float* pose_first, const float* flow_first, const float* mask_first, float* grad_first,
                        const float* depth_last, const float* pose_last, const float* flow_last, const float* mask_last, float* grad_last,
                        const float* kinv, const float* k_dst, const float* norm, const float* upstream, int height, int width,
                        int mapping_kind, float delta, float aspect_x, float aspect_y, void* stream) {
  FM_CHECK_ARG(kinv && k_dst && norm && height >= 1 && width >= 1 && mapping_kind >= 0 && mapping_kind <= 2);
  FM_CHECK_ARG(grad_first == nullptr || (depth_first && pose_first && flow_first && mask_first));
  FM_CHECK_ARG(grad_last == nullptr || (depth_last && pose_last && flow_last && mask_last));
  if (grad_first == nullptr && grad_last == nullptr) return FM_OK;
  const GhostSide s0{depth_first, pose_first, flow_first, mask_first, grad_first}, s1{depth_last, pose_last, flow_last, mask_last, grad_last};
  const long n = (long)height * width;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const dim3 grid((unsigned)blocks, 2);
  hipStream_t st = (hipStream_t)stream;
  if (mapping_kind == kHuber)
    hipLaunchKernelGGL(flow_ghost_kernel<kHuber>, grid, dim3(256), 0, st, s0, s1, kinv, k_dst, norm, upstream, height, width, delta, aspect_x, aspect_y);
  else if (mapping_kind == kL1)
    hipLaunchKernelGGL(flow_ghost_kernel<kL1>, grid, dim3(256), 0, st, s0, s1, kinv, k_dst, norm, upstream, height, width, delta, aspect_x, aspect_y);
  else
    hipLaunchKernelGGL(flow_ghost_kernel<kL2>, grid, dim3(256), 0, st, s0, s1, kinv, k_dst, norm, upstream, height, width, delta, aspect_x, aspect_y);
  FM_LAUNCH_STATUS();
}

int fm_halo_copy(const float* grad, long frame_elements, int frames, float* sent_first, float* sent_last, void* stream) {
  FM_CHECK_ARG(grad && frame_elements >= 1 && frames >= 1);
  if (!sent_first && !sent_last) return FM_OK;
  const float* last = grad + (size_t)(frames - 1) * frame_elements;
  const int vec = frame_elements % 4 == 0 && halo_aligned(grad) && halo_aligned(sent_first) && halo_aligned(sent_last);
  hipLaunchKernelGGL(halo_copy_kernel, dim3(halo_blocks(frame_elements / (vec ? 4 : 1)), 2), dim3(256), 0, (hipStream_t)stream, grad, last, sent_first,
                     sent_last, frame_elements, vec);
  FM_LAUNCH_STATUS();
}

int fm_halo_add(float* grad, long frame_elements, int frames, const float* dense_first, const float* dense_last, void* stream) {
  FM_CHECK_ARG(grad && frame_elements >= 1 && frames >= 1);
  if (!dense_first && !dense_last) return FM_OK;
  float* last = grad + (size_t)(frames - 1) * frame_elements;
  FM_CHECK_ARG(frames > 1 || !(dense_first && dense_last));  // (one frame cannot be both boundaries in one launch: two writers)
  const int vec = frame_elements % 4 == 0 && halo_aligned(grad) && halo_aligned(dense_first) && halo_aligned(dense_last);
  hipLaunchKernelGGL(halo_add_kernel, dim3(halo_blocks(frame_elements / (vec ? 4 : 1)), 2), dim3(256), 0, (hipStream_t)stream, grad, last, dense_first,
                     dense_last, frame_elements, vec);
  FM_LAUNCH_STATUS();
}

int fm_halo_delta(const float* grad, long frame_elements, int frames, const float* sent_first, const int64_t* pixels_first, long count_first,
                  float* out_first, const float* sent_last, const int64_t* pixels_last, long count_last, float* out_last, void* stream) {
  FM_CHECK_ARG(grad && frame_elements >= 1 && frames >= 1 && count_first >= 0 && count_last >= 0);
  FM_CHECK_ARG(!out_first || (sent_first && (pixels_first || count_first == 0)));
  FM_CHECK_ARG(!out_last || (sent_last && (pixels_last || count_last == 0)));
  if ((!out_first || count_first == 0) && (!out_last || count_last == 0)) return FM_OK;
  const float* last = grad + (size_t)(frames - 1) * frame_elements;
  const long most = count_first > count_last ? count_first : count_last;
  hipLaunchKernelGGL(halo_delta_kernel, dim3(halo_blocks(most), 2), dim3(256), 0, (hipStream_t)stream, grad, last, sent_first, sent_last, pixels_first,
                     pixels_last, out_first ? count_first : 0L, out_last ? count_last : 0L, out_first, out_last);
  FM_LAUNCH_STATUS();
}

int fm_halo_ghost_begin(const float* grad, long frame_elements, int frames, const int64_t* pixels_first, long count_first, float* base_first,
                        const int64_t* pixels_last, long count_last, float* base_last, const float* t_fwd, const float* t_bwd, int pairs,
                        const float* k, const float* kinv, float* pack, void* stream) {
  FM_CHECK_ARG(grad && frame_elements >= 1 && frames >= 1 && count_first >= 0 && count_last >= 0);
  FM_CHECK_ARG(!base_first || pixels_first || count_first == 0);
  FM_CHECK_ARG(!base_last || pixels_last || count_last == 0);
  FM_CHECK_ARG(!pack || (t_fwd && t_bwd && k && kinv && pairs >= 1));
  const float* last = grad + (size_t)(frames - 1) * frame_elements;
  const long n0 = base_first ? count_first : 0L, n1 = base_last ? count_last : 0L;
  if (!pack && n0 == 0 && n1 == 0) return FM_OK;
  const long most = n0 > n1 ? n0 : n1;
  hipLaunchKernelGGL(halo_ghost_begin_kernel, dim3(halo_blocks(most), 2), dim3(256), 0, (hipStream_t)stream, grad, last, pixels_first, pixels_last, n0, n1,
                     base_first, base_last, t_fwd, t_bwd, pairs, k, kinv, pack);
  FM_LAUNCH_STATUS();
}

int fm_halo_delta_sparse(const float* grad, long frame_elements, int frames, const float* base_first, const int64_t* pixels_first, long count_first,
                         float* out_first, const float* base_last, const int64_t* pixels_last, long count_last, float* out_last, void* stream) {
  FM_CHECK_ARG(grad && frame_elements >= 1 && frames >= 1 && count_first >= 0 && count_last >= 0);
  FM_CHECK_ARG(!out_first || count_first == 0 || (base_first && pixels_first));
  FM_CHECK_ARG(!out_last || count_last == 0 || (base_last && pixels_last));
  if ((!out_first || count_first == 0) && (!out_last || count_last == 0)) return FM_OK;
  const float* last = grad + (size_t)(frames - 1) * frame_elements;
  const long most = count_first > count_last ? count_first : count_last;
  hipLaunchKernelGGL(halo_delta_sparse_kernel, dim3(halo_blocks(most), 2), dim3(256), 0, (hipStream_t)stream, grad, last, base_first, base_last,
                     pixels_first, pixels_last, out_first ? count_first : 0L, out_last ? count_last : 0L, out_first, out_last);
  FM_LAUNCH_STATUS();
}

int fm_halo_scatter(float* grad, long frame_elements, int frames, const int64_t* pixels_first, const float* values_first, long count_first,
                    const int64_t* pixels_last, const float* values_last, long count_last, void* stream) {
  FM_CHECK_ARG(grad && frame_elements >= 1 && frames >= 1 && count_first >= 0 && count_last >= 0);
  FM_CHECK_ARG(!values_first || pixels_first || count_first == 0);
  FM_CHECK_ARG(!values_last || pixels_last || count_last == 0);
  FM_CHECK_ARG(frames > 1 || !(values_first && values_last));
  if ((!values_first || count_first == 0) && (!values_last || count_last == 0)) return FM_OK;
  float* last = grad + (size_t)(frames - 1) * frame_elements;
  const long most = count_first > count_last ? count_first : count_last;
  hipLaunchKernelGGL(halo_scatter_kernel, dim3(halo_blocks(most), 2), dim3(256), 0, (hipStream_t)stream, grad, last, pixels_first, pixels_last,
                     values_first, values_last, values_first ? count_first : 0L, values_last ? count_last : 0L);
  FM_LAUNCH_STATUS();
}

}  // extern "C"
