"""create_alpha_brend / grad_cumsum on scenes of the cfg3 pair count but SMALLER boxes (more Gaussians): which cut the list takes,
what the call costs.   python tools/small_boxes_bench.py [gaussians ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_kernel as ck  # noqa: E402
from simplegaussiansplat_tk71_amd import raster, synthetic  # noqa: E402
from tools.wrapper_bench import timeit  # noqa: E402

dev = torch.device("cuda", 0)
for n_gauss in [int(a) for a in sys.argv[1:]] or [1_000_000, 2_600_000, 6_600_000, 18_000_000]:
    sc = synthetic.make_scene(n_gauss, 1919, 1079, 80.0, seed=0, device=dev)
    rects, owner = raster.expand_rects(sc["start"], sc["end"], 1919, 1079, with_gaussian=True)
    m = rects.size(0)
    g = torch.Generator(device=dev).manual_seed(1)
    anti = 1.0 - sc["opacity"].reshape(-1)[owner.long()] * torch.rand(m, device=dev, generator=g)
    del owner
    out = {"gaussians": n_gauss, "pairs": m, "pairs_per_box": m / n_gauss}
    once = raster._cut_rects_once(rects, False, 0, 0, 8)
    out["one_call_cut"] = once if isinstance(once, str) or once is None else "ok"
    rb = raster.rects_to_boxes(rects)
    out["rectangles"] = None if rb is None else int(rb.start.size(0))
    out["rects_to_boxes_ms"] = timeit(lambda: raster.rects_to_boxes(rects), 5, 2)
    out["create_alpha_brend_ms"] = timeit(lambda: ck.create_alpha_brend(rects, anti, "cumprod"), 5, 2)
    out["sort_route_ms"] = timeit(lambda: ck.create_alpha_brend(rects, anti, "cumprod", image_size=(1919, 1079), route="sort"), 3, 1)
    print(json.dumps(out), flush=True)
    del rects, anti, sc, rb
    torch.cuda.empty_cache()
