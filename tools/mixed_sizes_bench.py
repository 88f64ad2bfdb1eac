"""Rows a5 / a6 / f1 / f3 on a scene whose box sizes are heavy-tailed (log-normal half-sizes: most boxes a few pixels, a few hundreds of
pixels wide) at about the cfg3 pair count.   python tools/mixed_sizes_bench.py [sigma_log ...]"""
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
for sig in [float(a) for a in sys.argv[1:]] or [0.0, 0.6, 1.0, 1.4]:
    g = torch.Generator(device=dev).manual_seed(3)
    n = 1_000_000
    base = synthetic.make_scene(n, W, H, 80.0, seed=0, device=dev)
    if sig > 0:
        # half-sizes log-normal with the same MEAN AREA as the uniform scene: (2h + 1)^2 ~ 165 on average
        z = torch.randn(n, 2, device=dev, generator=g) * sig
        side = torch.exp(z)                                    # E[side^2] = exp(2 sig^2) per axis
        side = side / torch.exp(torch.tensor(sig * sig, device=dev)) * 12.85
        half = ((side - 1.0) / 2.0).clamp(min=0).round().to(torch.int32)
        c = (base["start"] + base["end"]) // 2
        lim = torch.tensor([W, H], dtype=torch.int32, device=dev)
        base["start"] = (c - half).clamp(min=0)
        base["end"] = torch.minimum(c + half, lim)
    sc = base
    rects, owner = raster.expand_rects(sc["start"], sc["end"], W, H, with_gaussian=True)
    m = rects.size(0)
    anti = 1.0 - sc["opacity"].reshape(-1)[owner.long()] * torch.rand(m, device=dev, generator=g)
    del owner
    size = ((sc["end"] - sc["start"] + 1).long().prod(1))
    bins = raster.bin_tiles(sc["start"], sc["end"], W, H)
    out = {"sigma_log": sig, "pairs": m, "largest_box_pairs": int(size.max()), "median_box_pairs": int(size.median()), "tile_entries": bins.n_tile_pairs}
    out["create_rects_ms"] = timeit(lambda: ck.create_rects(sc["start"], sc["end"]), 5, 2)
    out["rects_to_boxes_ms"] = timeit(lambda: raster.rects_to_boxes(rects), 5, 2)
    rb = raster.rects_to_boxes(rects)
    out["rectangles"] = None if rb is None else int(rb.start.size(0))
    out["create_alpha_brend_ms"] = timeit(lambda: ck.create_alpha_brend(rects, anti, "cumprod"), 5, 2)
    out["grad_cumsum_ms"] = timeit(lambda: ck.grad_cumsum(rects, anti), 5, 2)
    out["create_alpha_brend_min_ms"] = timeit(lambda: ck.create_alpha_brend_min(rects, anti, image_size=(W, H)), 5, 2)
    out["sort_route_ms"] = timeit(lambda: ck.create_alpha_brend(rects, anti, "cumprod", image_size=(W, H), route="sort"), 3, 1)
    mean = ((sc["start"] + sc["end"]) // 2).to(torch.int32)
    out["bin_tiles_ms"] = timeit(lambda: raster.bin_tiles(sc["start"], sc["end"], W, H), 5, 2)
    img, ckpt = raster.blend_forward(bins, sc["start"], sc["end"], mean, sc["vinv"], sc["opacity"], sc["l_d"], with_checkpoints=True)
    gimg = torch.randn_like(img)
    out["blend_forward_ms"] = timeit(lambda: raster.blend_forward(bins, sc["start"], sc["end"], mean, sc["vinv"], sc["opacity"], sc["l_d"], with_checkpoints=True), 5, 2)
    out["blend_backward_ms"] = timeit(lambda: raster.blend_backward(bins, sc["start"], sc["end"], mean, sc["vinv"], sc["opacity"], sc["l_d"], ckpt, gimg), 5, 2)
    print(json.dumps(out), flush=True)
    del rects, anti, sc, bins, img, ckpt, gimg, rb
    torch.cuda.empty_cache()
