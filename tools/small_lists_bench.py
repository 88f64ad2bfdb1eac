"""Where the general sort route beats the default (cut + bin + walk: 26 launches) on SMALL lists: both timed on scenes of growing size.
   python tools/small_lists_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_kernel as ck  # noqa: E402
from simplegaussiansplat_tk71_amd import raster, synthetic  # noqa: E402
from tools.wrapper_bench import timeit  # noqa: E402

dev = torch.device("cuda", 0)
for n_gauss, w, h, depth in ((500, 255, 255, 2.0), (2000, 255, 255, 8.0), (4000, 511, 383, 6.0), (10000, 639, 479, 6.0), (20000, 959, 539, 6.0),
                             (30000, 1279, 719, 6.0), (50000, 1919, 1079, 4.0), (100000, 1919, 1079, 8.0)):
    sc = synthetic.make_scene(n_gauss, w, h, depth, seed=0, device=dev)
    rects = raster.expand_rects(sc["start"], sc["end"], w, h)
    m = rects.size(0)
    anti = 1.0 - 0.6 * torch.rand(m, device=dev)
    out = {"pairs": m, "image": f"{w + 1}x{h + 1}",
           "boxes_route_ms": timeit(lambda: ck.create_alpha_brend(rects, anti, "cumprod", route="boxes"), 9, 3),
           "sort_route_ms": timeit(lambda: ck.create_alpha_brend(rects, anti, "cumprod", route="sort"), 9, 3),
           "sort_route_image_size_ms": timeit(lambda: ck.create_alpha_brend(rects, anti, "cumprod", route="sort", image_size=(w, h)), 9, 3),
           "auto_ms": timeit(lambda: ck.create_alpha_brend(rects, anti, "cumprod"), 9, 3)}
    print(json.dumps(out), flush=True)
