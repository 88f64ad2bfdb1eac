"""Tile-list walk (k_pairs_scan_boxes) alone, at a BASELINE scene size: A/B runs of library variants.

  GCP_LIBRARY=variants/walk8.so python tools/walk_bench.py [cfg3] [--iters 10]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplegaussiansplat_tk71_amd import raster, synthetic  # noqa: E402
from tools.wrapper_bench import timeit  # noqa: E402


def main():
    argv = sys.argv[1:]
    iters = 10
    if "--iters" in argv:
        i = argv.index("--iters")
        iters = int(argv[i + 1])
        del argv[i:i + 2]
    dev = torch.device("cuda", 0)
    for cfg in (argv or ["cfg3"]):
        sc, rects, anti, grad = synthetic.make_scene_pairs(cfg, seed=0, device=dev)
        w, h = sc["width"], sc["height"]
        bins = raster.bin_tiles(sc["start"], sc["end"], w, h)
        boff = raster.box_offsets(sc["start"], sc["end"], w, h)
        ref = raster.scan_boxes(bins, sc["start"], sc["end"], boff, anti, 0).clone()
        out = {"library": os.environ.get("GCP_LIBRARY", "in-tree"), "workload": cfg, "pairs": int(rects.size(0)),
               "tile_entries": bins.n_tile_pairs}
        for name, vals, mode in (("cumprod", anti, 0), ("cumsum", anti, 1), ("reverse", grad, 2)):
            out[name + "_ms"] = timeit(lambda: raster.scan_boxes(bins, sc["start"], sc["end"], boff, vals, mode), iters, 3)
            out[name + "_counting_ms"] = timeit(lambda: raster.scan_boxes(bins, sc["start"], sc["end"], boff, vals, mode, count_dropped=True), iters, 3)
        out["checksum"] = float(ref.double().sum())
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
