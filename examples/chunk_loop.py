#!/usr/bin/env python3
"""The transmittance side of the reference's chunk loop, as it stands, on the HIP library.

`custom_autograd_grouped_cumprod.forward` (reference: gs_model.py:666-692) cuts a camera's depth-ordered Gaussians into
chunks for memory (:428) and, per chunk, runs `_forward_batch` (:598-624): expand the boxes into the rect list, scan it per
pixel, drop what is exactly 0, and take every pixel's smallest transmittance along as carry rows of the next chunk
(`_create_alpha_brend_min`, `_cat_alpha_brend`, `cutting_number`).  The class in cuda_kernel.py carries those helpers under the
reference's own names, so the loop below is the reference's statement sequence (:601-615) with nothing but the module changed.
(The fused `custom_autograd_grouped_cumprod.apply` needs none of this — nothing M-sized exists there to chunk — and, unlike
this loop, it is exact across chunk boundaries: the reference's carry is the transmittance IN FRONT of a pixel's last pair,
so one factor per boundary is lost)

    python examples/chunk_loop.py [--gaussians 200000 --width 1919 --height 1079 --depth 40 --chunks 4]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_kernel import custom_autograd_grouped_cumprod as F  # noqa: E402
from simplegaussiansplat_tk71_amd import synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=200_000)
    ap.add_argument("--width", type=int, default=1919)
    ap.add_argument("--height", type=int, default=1079)
    ap.add_argument("--depth", type=float, default=40.0, help="mean splat-pixel pairs per pixel")
    ap.add_argument("--chunks", type=int, default=4)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    sc = synthetic.make_scene(args.gaussians, args.width, args.height, args.depth, seed=0, device=dev)
    n = args.gaussians
    ends = [n * (c + 1) // args.chunks for c in range(args.chunks)]
    g = torch.Generator(device=dev).manual_seed(1)
    # first touch of the library and of the allocator's pool (not part of what is timed)
    warm = F._create_rects(sc["start"][: ends[0]], sc["end"][: ends[0]])
    F._create_alpha_brend_min(warm, F._create_alpha_brend(warm, torch.ones(warm.size(0), device=dev), flag="cumprod")[0])
    del warm
    unique_rects, T_min = None, None
    pairs = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c, end in enumerate(ends):
        s0 = ends[c - 1] if c else 0
        tc = time.perf_counter()
        rects = F._create_rects(sc["start"][s0:end], sc["end"][s0:end])                       # gs_model.py:601
        # 1 - opacity * gauss kernel of every pair (gs_model.py:602-604): a stand-in with the same range here
        anti_opacity = 1.0 - 0.6 * torch.rand(rects.size(0), device=dev, generator=g)
        if unique_rects is None:                                                               # :606-609
            T, mask = F._create_alpha_brend(rects, anti_opacity, flag="cumprod")
            (rects,) = F._mask_tensor(mask, rects)
            unique_rects, T_min = F._create_alpha_brend_min(rects, T)
        else:                                                                                  # :610-615
            cat_a, cat_rects = F._cat_alpha_brend([T_min, anti_opacity], [unique_rects, rects])
            T, mask = F._create_alpha_brend(cat_rects, cat_a, flag="cumprod", cutting_number=len(unique_rects))
            (rects,) = F._mask_tensor(mask, rects)
            cat_a, cat_rects = F._cat_alpha_brend([T_min, T], [unique_rects, rects])
            unique_rects, T_min = F._create_alpha_brend_min(cat_rects, cat_a)
        torch.cuda.synchronize()
        pairs += int(T.numel())
        print(f"chunk {c}: {end - s0} Gaussians, {T.numel()} pairs kept, {unique_rects.size(0)} pixels carried, "
              f"smallest transmittance so far {float(T_min.min()):.3e}, {1e3 * (time.perf_counter() - tc):.2f} ms")
    dt = time.perf_counter() - t0
    print(f"{args.chunks} chunks, {pairs} pairs: {1e3 * dt:.2f} ms on the HIP helpers (torch.unique(rects, dim=0), the first statement of the "
          f"reference's own _create_alpha_brend_min, takes about 0.85 ms per million rows on this GPU: {0.85e-6 * pairs:.0f} ms for these)")


if __name__ == "__main__":
    main()
