#!/usr/bin/env python3
"""Where a training step of the caller (row f4) spends its time at cfg3 scale: projection (torch ops) vs the HIP
Function, forward and backward.  python tools/caller_bench.py [--gaussians 1000000 --cameras 1]"""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplegaussiansplat_tk71_amd import gs_model as gm  # noqa: E402
from simplegaussiansplat_tk71_amd.synthetic import make_world, ring_cameras  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--cameras", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sigma-px", type=float, default=2.0)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = a.gaussians
    P, K, wh = ring_cameras(max(a.cameras, 1), a.width, a.height, device=dev)
    P, K, wh = P[:a.cameras], K[:a.cameras], wh[:a.cameras]
    model = gm.GS_model_with_param(*make_world(n, a.width, a.sigma_px, seed=0, device=dev))
    names = list(range(a.cameras))

    t_proj, (cams, _, _) = timed(lambda: model.camera_inputs(P, K, wh))
    pairs = sum(int(c["boxsize"].sum()) for c in cams if c is not None)
    kept = sum(c["boxsize"].numel() for c in cams if c is not None)
    print(f"{n} Gaussians, {a.cameras} camera(s) {a.width}x{a.height}: {kept} kept, {pairs:.3e} splat-pixel pairs")
    with torch.no_grad():
        t_fwd_nograd, _ = timed(lambda: model(P, K, wh, names))
    t_fwd, out = timed(lambda: model(P, K, wh, names))
    target = torch.rand_like(out[0])

    def step():
        images = model(P, K, wh, names)[0]
        loss = gm.splat_loss(images, target)
        loss.backward()
        model._optimizer.zero_grad(set_to_none=True)

    def step_l1():
        images = model(P, K, wh, names)[0]
        (images - target).abs().mean().backward()
        model._optimizer.zero_grad(set_to_none=True)

    t_step, _ = timed(step)
    t_l1, _ = timed(step_l1)
    print(f"projection only (camera_inputs, autograd graph recorded) {t_proj:8.2f} ms")
    print(f"forward  (projection + Function), no grad                {t_fwd_nograd:8.2f} ms")
    print(f"forward  (projection + Function), grad                   {t_fwd:8.2f} ms")
    print(f"forward + L1 loss + backward                             {t_l1:8.2f} ms")
    print(f"forward + L1/D-SSIM loss (fused) + backward              {t_step:8.2f} ms")
    # the Function alone on the same inputs
    from cuda_kernel import custom_autograd_grouped_cumprod as F
    c = cams[0]
    w, h = wh[0, 0].to(torch.int32), wh[0, 1].to(torch.int32)
    vi, o, l = (c[k].detach().clone().requires_grad_(True) for k in ("variance_inverse", "opacity", "l_d"))
    batch = c["boxsize"].new_tensor([c["boxsize"].numel()])
    gimg = torch.rand(a.height + 1, a.width + 1, 3, device=dev)

    def fn_only():
        img = F.apply(c["boxsize"], batch, c["startpoint"], c["endpoint"], c["mean"], vi, o, l, w, h)
        img.backward(gimg)
        vi.grad = o.grad = l.grad = None

    t_fn, _ = timed(fn_only)
    print(f"Function alone, forward + backward, one camera           {t_fn:8.2f} ms")


if __name__ == "__main__":
    main()
