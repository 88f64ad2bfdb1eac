"""The one-call cut (gcp_rects_cut) against the step-by-step one at a BASELINE scene size: what it reports, how long each takes.

  python tools/cut_diag.py [cfg3]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplegaussiansplat_tk71_amd import raster, synthetic  # noqa: E402
from tools.wrapper_bench import timeit  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    only = None
    argv = sys.argv[1:]
    if "--only" in argv:  # --only one | step: just that cut, ten times (for a kernel trace)
        i = argv.index("--only")
        only = argv[i + 1]
        del argv[i:i + 2]
    for cfg in (argv or ["cfg3"]):
        sc, rects, anti, grad = synthetic.make_scene_pairs(cfg, seed=0, device=dev)
        m = rects.size(0)
        if only:
            for _ in range(10):
                if only == "one":
                    raster._cut_rects_once(rects, False, 0, 0, 8)
                else:
                    raster.rects_to_boxes(rects, one_call=False)
            torch.cuda.synchronize()
            continue
        once = raster._cut_rects_once(rects, False, 0, 0, 8)
        out = {"workload": cfg, "pairs": m, "one_call": once if isinstance(once, str) or once is None else
               {"rects": int(once.start.size(0)), "K": once.n_tile_pairs, "width": once.width, "height": once.height}}
        out["one_call_ms"] = timeit(lambda: raster._cut_rects_once(rects, False, 0, 0, 8), 7, 3)
        out["step_by_step_ms"] = timeit(lambda: raster.rects_to_boxes(rects, one_call=False), 7, 3)
        out["rects_to_boxes_ms"] = timeit(lambda: raster.rects_to_boxes(rects), 7, 3)
        rb = raster.rects_to_boxes(rects)
        out["bin_ms"] = timeit(lambda: rb.bin(), 7, 3)
        out["bin_with_count_ms"] = timeit(lambda: raster.bin_tiles(rb.start, rb.end, rb.width, rb.height), 7, 3)
        torch.cuda.reset_peak_memory_stats(dev)
        base = torch.cuda.memory_allocated(dev)
        raster.rects_to_boxes(rects)
        out["one_call_peak_scratch_bytes_per_pair"] = (torch.cuda.max_memory_allocated(dev) - base) / m
        torch.cuda.reset_peak_memory_stats(dev)
        raster.rects_to_boxes(rects, one_call=False)
        out["step_by_step_peak_scratch_bytes_per_pair"] = (torch.cuda.max_memory_allocated(dev) - base) / m
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
