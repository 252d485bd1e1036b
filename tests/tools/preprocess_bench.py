"""Time the fused flow post-processing against the reference's op sequence on the same GPU
(run through gpurun).  The reference sequence is restated with torch ops by the oracle
(oracle.bidirectional_flows = flow_predictor.py:82-102); the network is a stand-in.

    python tests/tools/preprocess_bench.py [--frames 12 --height 2880 --width 5120 --scale 4]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from flowmap_amd import _ops  # noqa: E402
from oracle import flowmap_oracle as orc  # noqa: E402  (comparison only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--height", type=int, default=2880)
    ap.add_argument("--width", type=int, default=5120)
    ap.add_argument("--scale", type=int, default=4)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--crop-frames", type=int, default=24)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    f, h, w = args.frames, args.height, args.width
    shape = (h // args.scale, w // args.scale)
    g = torch.Generator(device=dev).manual_seed(0)
    # smooth video and flows (i.i.d. noise at 5120 px makes the photometric mask ill-conditioned:
    # a 1-ulp difference in a sampling coordinate moves (1-d)^8 by 1e-3)
    def smooth(shape_low, channels):
        low = torch.rand((channels, 1, *shape_low), device=dev, generator=g)
        return torch.nn.functional.interpolate(low, size=(h, w), mode="bicubic", align_corners=False)[:, 0]

    videos = smooth((h // 64, w // 64), f * 3).clamp(0, 1).reshape(1, f, 3, h, w).contiguous()
    raw_f = (0.02 * (smooth((h // 128, w // 128), (f - 1) * 2) - 0.5)).reshape(1, f - 1, 2, h, w).permute(0, 1, 3, 4, 2).contiguous()
    raw_b = (0.02 * (smooth((h // 128, w // 128), (f - 1) * 2) - 0.5)).reshape(1, f - 1, 2, h, w).permute(0, 1, 3, 4, 2).contiguous()
    flipped = videos.flip(dims=(1,))

    def ours():
        a = _ops.flow_postprocess(videos, raw_f, shape, reverse=False)
        b = _ops.flow_postprocess(videos, raw_b, shape, reverse=True)
        return a, b

    def reference_ops():
        return orc.bidirectional_flows(videos, lambda v: raw_f if v is videos else raw_b, shape)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.iters * 1e3, out

    ms_ours, (fa, fb) = timed(ours)
    ms_ref, fl = timed(reference_ops)
    err_gpu = max(float((fa[0] - fl.forward).abs().max()), float((fa[1] - fl.forward_mask).abs().max()),
                  float((fb[0] - fl.backward).abs().max()), float((fb[1] - fl.backward_mask).abs().max()))
    # torch's GPU grid_sample un-normalises coordinates with a different rounding than its CPU
    # kernel; where a sample sits on a pixel or image border that flips a tap (isolated pixels,
    # up to 2e-3 in the mask).  The CPU op sequence is the reference semantics: check the first
    # pair against it.
    cpu_first = videos[:, :2].cpu()
    m_cpu = orc.consistency_mask(cpu_first, raw_f[:, :1].cpu())
    m_cpu = orc.resize_bilinear(m_cpu.reshape(1, 1, h, w), shape).reshape(shape)
    err_cpu = float((fa[1][0, 0].cpu() - m_cpu).abs().max())
    print(json.dumps({"frames": f, "full_res": [h, w], "flow_shape": list(shape), "ms_fused_hip": ms_ours,
                      "ms_reference_ops_on_gpu": ms_ref, "speedup": ms_ref / ms_ours, "max_abs_diff_vs_torch_gpu_ops": err_gpu,
                      "max_abs_diff_mask_vs_torch_cpu_ops_first_pair": err_cpu}))
    del raw_f, raw_b, flipped, fa, fb, fl

    # cropping.py: the loaded video (1080p) -> the flow network's input (4x the optimisation's
    # 720x1280, patch-cropped); reference = resize_batch then center_crop on the same GPU
    from flowmap_amd import Batch
    from flowmap_amd.misc import cropping

    cf = args.crop_frames
    src = smooth((1080 // 64, 1920 // 64), 3)[:, :1080, :1920].clamp(0, 1)
    src = src[None, None].expand(1, cf, 3, 1080, 1920).contiguous()
    cfg = cropping.CroppingCfg((h // args.scale, w // args.scale), args.scale, 32)

    def ours_crop():
        return cropping.crop_and_resize_batch_for_flow(Batch(src), cfg).videos

    def reference_crop():
        out, _, _ = orc.crop_and_resize(src, None, cfg.image_shape, cfg.patch_size, cfg.flow_scale_multiplier)
        return out.contiguous()  # the reference's crop is a view; its consumers copy it (Batch.to / the network)

    ms_crop, got = timed(ours_crop)
    ms_crop_ref, want = timed(reference_crop)
    print(json.dumps({"what": "crop_and_resize_batch_for_flow", "frames": cf, "source": [1080, 1920], "out": list(got.shape[-2:]),
                      "ms_fused_hip": ms_crop, "ms_reference_ops_on_gpu": ms_crop_ref, "speedup": ms_crop_ref / ms_crop,
                      "write_GBps": got.numel() * 4 / ms_crop / 1e6, "max_abs_diff_vs_torch_gpu_ops": float((got - want).abs().max())}))


if __name__ == "__main__":
    main()
