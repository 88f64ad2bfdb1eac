"""One training step of the caller at cfg3 scale with the REFERENCE's formulation of projection and loss — ~150
PyTorch ops for gs_model.py:277-425, PyTorch convolutions for gs_control.py:180-182 (oracle/gs_forward_torch.py,
oracle/loss_torch.py) — around the same HIP Function, beside the build's fused kernels.  Checker/baseline only
(tests/ may use oracle/); the number quoted in DESIGN.md §7 f4 and bench.py's caller_level comes from here.

    python tests/bench_reference_caller_gpu.py [--gaussians 1000000]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gs_forward_torch as gft  # noqa: E402
from oracle import loss_torch  # noqa: E402
from simplegaussiansplat_tk71_amd import gs_model as gm  # noqa: E402
from simplegaussiansplat_tk71_amd.synthetic import make_world, ring_cameras  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    width, height = 1920, 1080
    P, K, wh = ring_cameras(1, width, height, device=dev)
    model = gm.GS_model_with_param(*make_world(a.gaussians, width, 2.0, seed=0, device=dev))
    target = torch.rand(1, 3, height, width, device=dev)
    fused_inputs = model.camera_inputs

    def torch_inputs(P, K, wh):
        return gft.camera_inputs(model.mean, model.variance_q, model.variance_scale, model.opacity, model.color, P, K, wh,
                                 model.variance_pixel_tile_max_width)

    def step(loss_fn):
        images = model(P, K, wh, [0])[0]
        loss_fn(images, target, 0.2).backward()
        model._optimizer.zero_grad(set_to_none=True)

    rows = []
    for name, inputs, loss_fn, reps in (("fused projection + fused loss (the build)", fused_inputs, gm.splat_loss, 5),
                                        ("fused projection + PyTorch loss", fused_inputs, loss_torch.splat_loss, 3),
                                        ("PyTorch projection + fused loss", torch_inputs, gm.splat_loss, 2),
                                        ("PyTorch projection + PyTorch loss (the reference's formulation)", torch_inputs, loss_torch.splat_loss, 2)):
        model.camera_inputs = inputs
        rows.append((name, timed(lambda: step(loss_fn), reps)))
    print(f"{a.gaussians} Gaussians, one {width}x{height} camera; projection + Function + L1/D-SSIM, forward and backward")
    for name, ms in rows:
        print(f"  {name:68s} {ms:9.2f} ms")


if __name__ == "__main__":
    main()
