"""Rows a5 / a6 (the reference's only live callers of the scans, gs_model.py:544-566 and :716-722) at BASELINE scene
sizes: whole-call times of create_alpha_brend(rects), create_alpha_brend_boxes, grad_cumsum, and — with --stages — every
stage of the rects route timed on its own.  Run it under `rocprofv3 --kernel-trace --stats` for the kernel table that
profiles/r03_wrappers.md quotes.

  python tools/wrapper_bench.py [cfg2 cfg3] [--stages] [--iters 5]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_kernel as ck  # noqa: E402
from simplegaussiansplat_tk71_amd import raster, synthetic  # noqa: E402


def timeit(fn, iters=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def scene_inputs(cfg, dev, seed=0):
    """The pair list as the reference builds it for one camera: Gaussian-major rects (uitility.py:336-366) and the
    anti-opacity of every pair (gs_model.py:533-535) of a synthetic scene of the config's shape."""
    sc = synthetic.make_scene_config(cfg, seed=seed, device=dev)
    rects, owner = raster.expand_rects(sc["start"], sc["end"], sc["width"], sc["height"], with_gaussian=True)
    g = torch.Generator(device=dev).manual_seed(seed + 1)
    gk = torch.rand(rects.size(0), device=dev, generator=g)
    anti = 1.0 - sc["opacity"].reshape(-1)[owner.long()] * gk
    grad = torch.randn(rects.size(0), device=dev, generator=g)
    return sc, rects, anti, grad


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    stages = "--stages" in sys.argv
    iters = 5
    if "--iters" in sys.argv:
        iters = int(sys.argv[sys.argv.index("--iters") + 1])
    dev = torch.device("cuda", 0)
    for cfg in (args or ["cfg2", "cfg3"]):
        sc, rects, anti, grad = scene_inputs(cfg, dev)
        m = rects.size(0)
        w, h = sc["width"], sc["height"]
        out = {"workload": cfg, "pairs": m, "gaussians": int(sc["start"].size(0))}
        out["create_alpha_brend_ms"] = timeit(lambda: ck.create_alpha_brend(rects, anti, "cumprod"), iters)
        out["create_alpha_brend_cumsum_ms"] = timeit(lambda: ck.create_alpha_brend(rects, anti, "cumsum"), iters)
        out["grad_cumsum_ms"] = timeit(lambda: ck.grad_cumsum(rects, grad), iters)
        out["create_alpha_brend_boxes_ms"] = timeit(lambda: ck.create_alpha_brend_boxes(sc["start"], sc["end"], anti, w, h, "cumprod"), iters)
        out["grad_cumsum_boxes_ms"] = timeit(lambda: ck.grad_cumsum_boxes(sc["start"], sc["end"], grad, w, h), iters)
        if hasattr(ck, "wrapper_stage_times"):
            out["stages"] = ck.wrapper_stage_times(rects, anti, timeit, iters)
        elif stages:
            key = ck.unique(rects).contiguous()
            bits = int(key.max().item()).bit_length()
            st = {"unique_ms": timeit(lambda: ck.unique(rects).contiguous(), iters),
                  "sort_ms": timeit(lambda: raster.stable_sort_keys(key, key_bits=bits), iters),
                  "sort_with_readback_ms": timeit(lambda: raster.stable_sort_keys(key), iters)}
            sk, idx = raster.stable_sort_keys(key, key_bits=bits)
            st["gather_ms"] = timeit(lambda: raster.gather_f32(anti, idx), iters)
            sx = raster.gather_f32(anti, idx)
            y = torch.empty_like(sx)
            import grouped_cumprod as gc
            st["scan_ms"] = timeit(lambda: gc.grouped_cumprod_forward(sx, sk, y), iters)
            st["unsort_finish_ms"] = timeit(lambda: raster.unsort_finish(y, sx, idx, 0), iters)
            full, keep = raster.unsort_finish(y, sx, idx, 0)
            st["compaction_ms"] = timeit(lambda: full[keep], iters)
            st["dropped_pairs"] = int(m - keep.sum().item())
            st["torch_sort_stable_ms"] = timeit(lambda: torch.sort(key, stable=True), 3, 1)
            out["stages"] = st
        print(json.dumps(out), flush=True)
        del rects, anti, grad, sc
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
