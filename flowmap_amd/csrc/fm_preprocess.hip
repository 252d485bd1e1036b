// Flow post-processing — the producer of the hot path's largest inputs.
//
// FlowPredictor.compute_bidirectional_flow (flowmap/flow/flow_predictor.py:82-102) runs the
// optical-flow network at the video's resolution, derives a photometric consistency mask
// there (:60-80), and bilinearly resizes flow and mask to the optimisation's resolution
// (:39-57) — for both temporal directions, flipping the video and the results in between.
// fm_flow_postprocess does everything after the network in ONE launch per direction: one
// thread per OUTPUT pixel, mask evaluated only at the four taps the resize reads, results
// written straight into the final (un-flipped) pair order.  No full-resolution mask, no
// permuted copies.  All gathers; the outputs are the only streaming traffic.
#include "fm_device.h"
#include "fm_math.h"

namespace fm {

// frames of pair `pair`: forward = (pair, pair+1); reverse = (pair+1, pair), and the raw
// flow comes from the time-flipped video, i.e. raw index frames-2-pair (:94-100).
struct PairFrames {
  int src, tgt, raw;
};

__device__ __forceinline__ PairFrames pair_frames(int pair, int frames, int reverse) {
  PairFrames p;
  p.src = reverse ? pair + 1 : pair;
  p.tgt = reverse ? pair : pair + 1;
  p.raw = reverse ? frames - 2 - pair : pair;
  return p;
}

__global__ void __launch_bounds__(256) consistency_mask_kernel(const float* videos, const float* flow, int frames, int h, int w,
                                                               float* mask) {
  const int bp = blockIdx.y, b = bp / (frames - 1), pair = bp % (frames - 1);
  const size_t n = (size_t)h * w;
  const float* src = videos + ((size_t)b * frames + pair) * 3 * n;
  const float* tgt = src + 3 * n;
  const float* fl = flow + (size_t)bp * n * 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / w), col = (int)(i - (size_t)row * w);
    const float2 f = reinterpret_cast<const float2*>(fl)[i];
    mask[(size_t)bp * n + i] = consistency_mask_at(src, tgt, h, w, row, col, f.x, f.y);
  }
}

__global__ void __launch_bounds__(256) flow_postprocess_kernel(const float* videos, const float* flow, int frames, int h, int w, int oh,
                                                               int ow, int reverse, float* out_flow, float* out_mask) {
  const int bp = blockIdx.y, b = bp / (frames - 1), pair = bp % (frames - 1);
  const PairFrames pf = pair_frames(pair, frames, reverse);
  const size_t n = (size_t)h * w, on = (size_t)oh * ow;
  const float* src = videos + ((size_t)b * frames + pf.src) * 3 * n;
  const float* tgt = videos + ((size_t)b * frames + pf.tgt) * 3 * n;
  const float* fl = flow + ((size_t)b * (frames - 1) + pf.raw) * n * 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < on; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / ow), col = (int)(i - (size_t)row * ow);
    float of[2], om;
    flow_postprocess_at(src, tgt, fl, h, w, oh, ow, row, col, of, om);
    reinterpret_cast<float2*>(out_flow)[(size_t)bp * on + i] = make_float2(of[0], of[1]);
    out_mask[(size_t)bp * on + i] = om;
  }
}

// resize_batch + center_crop_images (flowmap/misc/cropping.py:19-51) in one pass: only the pixels
// that survive the crop are interpolated, and the uncropped resized video never exists.
//   out[p][y][x] = bilinear_resize(in[p] -> (rh, rw))[y + row0][x + col0]
constexpr int kResizeRows = 8;  // output rows per block of resize_crop_kernel

struct __attribute__((packed, aligned(4))) Window4 {  // 4 consecutive floats at any float address
  float v[4];
};

__device__ __forceinline__ float pick4(const Window4& win, int k) {
  return k == 0 ? win.v[0] : k == 1 ? win.v[1] : k == 2 ? win.v[2] : win.v[3];
}

template <bool VEC>
__global__ void __launch_bounds__(256) resize_crop_kernel(const float* in, int h, int w, int rh, int rw, int row0, int col0, int oh, int ow,
                                                          float* out) {
  // A thread owns 4 consecutive output columns and walks kResizeRows rows: its column taps are
  // computed once, the row taps are wave-uniform (scalar registers), and a row is written as one
  // 16-byte non-temporal store per thread.  The output is the big stream (an up-scaled source stays
  // in L2), so what is left per pixel is the 4 gathers and the 7-operation blend.
  const float* src = in + (size_t)blockIdx.z * h * w;
  float* dst = out + (size_t)blockIdx.z * oh * ow;
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x0 >= ow) return;
  int i0[4], i1[4];
  float l0[4], l1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const ResizeTap tx = resize_tap((x0 + j < ow ? x0 + j : ow - 1) + col0, w, rw);
    i0[j] = tx.i0, i1[j] = tx.i1, l0[j] = tx.l0, l1[j] = tx.l1;
  }
  // When the image is not shrunk, the 8 column taps of a thread fall inside 4 consecutive source
  // pixels: one 16-byte load per source row replaces 8 gathers, taps are picked from registers.
  const int base = min(i0[0], w - 4);
  const bool window = w >= 4 && i1[3] - base <= 3;
  const int y_end = min((int)(blockIdx.y + 1) * kResizeRows, oh);
  for (int y = blockIdx.y * kResizeRows; y < y_end; ++y) {
#pragma clang fp contract(off)
    const ResizeTap ty = resize_tap(y + row0, h, rh);
    const float* r0 = src + (size_t)ty.i0 * w;
    const float* r1 = src + (size_t)ty.i1 * w;
    float v[4];
    if (window) {
      const Window4 a = *reinterpret_cast<const Window4*>(r0 + base), b = *reinterpret_cast<const Window4*>(r1 + base);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k0 = i0[j] - base, k1 = i1[j] - base;
        v[j] = ty.l0 * (l0[j] * pick4(a, k0) + l1[j] * pick4(a, k1)) + ty.l1 * (l0[j] * pick4(b, k0) + l1[j] * pick4(b, k1));
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = ty.l0 * (l0[j] * r0[i0[j]] + l1[j] * r0[i1[j]]) + ty.l1 * (l0[j] * r1[i0[j]] + l1[j] * r1[i1[j]]);
    }
    float* o = dst + (size_t)y * ow + x0;
    if (VEC) {
      __builtin_nontemporal_store(v[0], o), __builtin_nontemporal_store(v[1], o + 1);
      __builtin_nontemporal_store(v[2], o + 2), __builtin_nontemporal_store(v[3], o + 3);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (x0 + j < ow) o[j] = v[j];
    }
  }
}

}  // namespace fm

using namespace fm;

extern "C" {

int fm_consistency_mask(const float* videos, const float* flow, int batch, int frames, int height, int width, float* mask,
                        void* stream) {
  FM_CHECK_ARG(videos && flow && mask && batch >= 1 && frames >= 2 && height >= 1 && width >= 1);
  FM_CHECK_ARG((long)batch * (frames - 1) <= 65535);
  const long n = (long)height * width;
  long bx = (n + 255) / 256;
  if (bx > 4096) bx = 4096;
  hipLaunchKernelGGL(consistency_mask_kernel, dim3((unsigned)bx, (unsigned)(batch * (frames - 1))), dim3(256), 0, (hipStream_t)stream,
                     videos, flow, frames, height, width, mask);
  FM_LAUNCH_STATUS();
}

int fm_flow_postprocess(const float* videos, const float* flow, int batch, int frames, int height, int width, int out_height,
                        int out_width, int reverse, float* out_flow, float* out_mask, void* stream) {
  FM_CHECK_ARG(videos && flow && out_flow && out_mask && batch >= 1 && frames >= 2 && height >= 1 && width >= 1);
  FM_CHECK_ARG(out_height >= 1 && out_width >= 1 && (long)batch * (frames - 1) <= 65535);
  const long on = (long)out_height * out_width;
  long bx = (on + 255) / 256;
  if (bx > 4096) bx = 4096;
  hipLaunchKernelGGL(flow_postprocess_kernel, dim3((unsigned)bx, (unsigned)(batch * (frames - 1))), dim3(256), 0, (hipStream_t)stream,
                     videos, flow, frames, height, width, out_height, out_width, reverse ? 1 : 0, out_flow, out_mask);
  FM_LAUNCH_STATUS();
}

int fm_resize_crop(const float* in, long planes, int height, int width, int resized_height, int resized_width, int row0, int col0,
                   int out_height, int out_width, float* out, void* stream) {
  FM_CHECK_ARG(in && out && planes >= 1 && planes <= 65535 && height >= 1 && width >= 1 && resized_height >= 1 && resized_width >= 1);
  FM_CHECK_ARG(row0 >= 0 && col0 >= 0 && out_height >= 1 && out_width >= 1 && row0 + out_height <= resized_height &&
               col0 + out_width <= resized_width);
  FM_CHECK_ARG((out_height + kResizeRows - 1) / kResizeRows <= 65535);
  const dim3 grid((unsigned)((out_width + 1023) / 1024), (unsigned)((out_height + kResizeRows - 1) / kResizeRows), (unsigned)planes);
  if (out_width % 4 == 0 && ((uintptr_t)out & 15) == 0)
    hipLaunchKernelGGL(resize_crop_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, in, height, width, resized_height, resized_width,
                       row0, col0, out_height, out_width, out);
  else
    hipLaunchKernelGGL(resize_crop_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, in, height, width, resized_height,
                       resized_width, row0, col0, out_height, out_width, out);
  FM_LAUNCH_STATUS();
}

}  // extern "C"
