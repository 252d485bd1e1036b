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
__global__ void __launch_bounds__(256) resize_crop_kernel(const float* in, int h, int w, int rh, int rw, int row0, int col0, int oh, int ow,
                                                          float* out) {
  const size_t plane = blockIdx.y;
  const float* src = in + plane * (size_t)h * w;
  float* dst = out + plane * (size_t)oh * ow;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)oh * ow; i += (size_t)gridDim.x * blockDim.x) {
#pragma clang fp contract(off)
    const int y = (int)(i / ow), x = (int)(i - (size_t)y * ow);
    const ResizeTap ty = resize_tap(y + row0, h, rh), tx = resize_tap(x + col0, w, rw);
    const float v00 = src[(size_t)ty.i0 * w + tx.i0], v01 = src[(size_t)ty.i0 * w + tx.i1];
    const float v10 = src[(size_t)ty.i1 * w + tx.i0], v11 = src[(size_t)ty.i1 * w + tx.i1];
    dst[i] = ty.l0 * (tx.l0 * v00 + tx.l1 * v01) + ty.l1 * (tx.l0 * v10 + tx.l1 * v11);
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
  const long n = (long)out_height * out_width;
  long bx = (n + 255) / 256;
  if (bx > 4096) bx = 4096;
  hipLaunchKernelGGL(resize_crop_kernel, dim3((unsigned)bx, (unsigned)planes), dim3(256), 0, (hipStream_t)stream, in, height, width,
                     resized_height, resized_width, row0, col0, out_height, out_width, out);
  FM_LAUNCH_STATUS();
}

}  // extern "C"
