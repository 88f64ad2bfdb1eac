"""rects_to_boxes (the rect list cut back into boxes) alone, at a BASELINE scene size, from the int32 and the int64 list:
A/B runs of library variants.

  GCP_LIBRARY=variants/x.so python tools/cut_bench.py [cfg3] [--iters 10]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplegaussiansplat_tk71_amd import raster, synthetic  # noqa: E402
from tools.wrapper_bench import timeit  # noqa: E402

argv = sys.argv[1:]
iters = 10
if "--iters" in argv:
    i = argv.index("--iters")
    iters = int(argv[i + 1])
    del argv[i:i + 2]
dev = torch.device("cuda", 0)
for cfg in (argv or ["cfg3"]):
    sc, rects, anti, grad = synthetic.make_scene_pairs(cfg, seed=0, device=dev)
    rb = raster.rects_to_boxes(rects)
    r64 = rects.long()
    rb64 = raster.rects_to_boxes(r64)
    assert torch.equal(rb.start, rb64.start) and torch.equal(rb.end, rb64.end) and torch.equal(rb.box_off, rb64.box_off)
    out = {"library": os.environ.get("GCP_LIBRARY", "in-tree"), "workload": cfg, "pairs": int(rects.size(0)), "boxes": int(rb.start.size(0)),
           "rects_to_boxes_ms": timeit(lambda: raster.rects_to_boxes(rects), iters, 3),
           "rects_to_boxes_int64_ms": timeit(lambda: raster.rects_to_boxes(r64), iters, 3),
           "bin_tiles_ms": timeit(lambda: raster.bin_tiles(rb.start, rb.end, rb.width, rb.height), iters, 3),
           "checksum": int(rb.start.long().sum() + rb.end.long().sum() + rb.box_off.long().sum())}
    print(json.dumps(out), flush=True)
