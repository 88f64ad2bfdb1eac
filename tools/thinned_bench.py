"""The backward's call on the list the forward thinned: `rects = rects[mask]` (gs_model.py:608) leaves a list that is no longer whole
boxes wherever pairs were dropped; `grad_cumsum(rects, grad)` (:641) then has to cut THAT.  Clustered scenes drop a lot.
   python tools/thinned_bench.py [s ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_kernel as ck  # noqa: E402
from simplegaussiansplat_tk71_amd import raster, synthetic  # noqa: E402
from tools.wrapper_bench import timeit  # noqa: E402

dev = torch.device("cuda", 0)
W, H = 1919, 1079
for s in [float(a) for a in sys.argv[1:]] or [4.0, 8.0]:
    sc = synthetic.make_scene(1_000_000, W, H, 80.0, seed=0, device=dev)
    g = torch.Generator(device=dev).manual_seed(7)
    half = (sc["end"] - sc["start"]) // 2
    cx = (torch.randn(sc["start"].size(0), device=dev, generator=g) * (W / s) + W / 2).round().clamp(0, W).to(torch.int32)
    cy = (torch.randn(sc["start"].size(0), device=dev, generator=g) * (H / s) + H / 2).round().clamp(0, H).to(torch.int32)
    c = torch.stack([cx, cy], 1)
    lim = torch.tensor([W, H], dtype=torch.int32, device=dev)
    sc["start"] = (c - half).clamp(min=0)
    sc["end"] = torch.minimum(c + half, lim)
    rects, owner = raster.expand_rects(sc["start"], sc["end"], W, H, with_gaussian=True)
    m = rects.size(0)
    anti = 1.0 - sc["opacity"].reshape(-1)[owner.long()] * torch.rand(m, device=dev, generator=g)
    del owner
    T, mask = ck.create_alpha_brend(rects, anti, "cumprod")
    kept = rects[mask]
    grad = torch.randn(kept.size(0), device=dev, generator=g)
    out = {"sigma_divisor": s, "pairs": m, "kept_pairs": int(kept.size(0))}
    rb = raster.rects_to_boxes(kept)
    out["thinned_list_rectangles"] = None if rb is None else int(rb.start.size(0))
    out["thinned_rects_to_boxes_ms"] = timeit(lambda: raster.rects_to_boxes(kept), 5, 2)
    out["grad_cumsum_on_thinned_list_ms"] = timeit(lambda: ck.grad_cumsum(kept, grad), 5, 2)
    out["grad_cumsum_on_thinned_list_sort_route_ms"] = timeit(lambda: ck.grad_cumsum(kept, grad, image_size=(W, H), route="sort"), 3, 1)
    out["create_alpha_brend_min_on_thinned_list_ms"] = timeit(lambda: ck.create_alpha_brend_min(kept, T, image_size=(W, H)), 5, 2)
    # agreement of the two routes on the thinned list
    a, am = ck.grad_cumsum(kept, grad)
    b, bm = ck.grad_cumsum(kept, grad, image_size=(W, H), route="sort")
    out["mask_differences"] = int((am != bm).sum())
    print(json.dumps(out), flush=True)
    del rects, anti, sc, T, mask, kept, grad, a, b, am, bm
    torch.cuda.empty_cache()
