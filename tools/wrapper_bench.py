"""Rows a5 / a6 (the reference's only live callers of the scans, gs_model.py:544-566 and :716-722) at BASELINE scene
sizes: whole-call times of create_alpha_brend(rects), create_alpha_brend_boxes, grad_cumsum, and — with --stages — every
stage of the rects route timed on its own.  Run it under `rocprofv3 --kernel-trace --stats` for the kernel table that
profiles/r03_wrappers.md quotes.

  python tools/wrapper_bench.py [cfg2 cfg3] [--stages] [--chunked] [--carry | --only-carry] [--iters 5]
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
    return synthetic.make_scene_pairs(cfg, seed=seed, device=dev)


def main():
    argv = sys.argv[1:]
    iters = 5
    if "--iters" in argv:
        i = argv.index("--iters")
        iters = int(argv[i + 1])
        del argv[i:i + 2]
    args = [a for a in argv if not a.startswith("--")]
    stages = "--stages" in argv
    if "--profile" in argv:  # only the four whole calls, `iters` times each: what tools/profile_round.sh runs under rocprofv3
        dev = torch.device("cuda", 0)
        for cfg in (args or ["cfg3"]):
            sc, rects, anti, grad = scene_inputs(cfg, dev)
            w, h = sc["width"], sc["height"]
            bits = ck.pixel_key_bits(w, h)
            for _ in range(iters):
                ck.create_alpha_brend(rects, anti, "cumprod")                                    # auto: cut into boxes, walk
                ck.grad_cumsum(rects, grad)
                ck.create_alpha_brend(rects, anti, "cumprod", image_size=(w, h), route="sort")  # the general route
                ck.grad_cumsum(rects, grad, image_size=(w, h), route="sort")
                ck.create_alpha_brend_boxes(sc["start"], sc["end"], anti, w, h, "cumprod")
                ck.grad_cumsum_boxes(sc["start"], sc["end"], grad, w, h)
            torch.cuda.synchronize()
            print(json.dumps({"workload": cfg, "pairs": rects.size(0), "iters": iters}))
        return
    dev = torch.device("cuda", 0)
    for cfg in (args or ["cfg2", "cfg3"]):
        sc, rects, anti, grad = scene_inputs(cfg, dev)
        m = rects.size(0)
        w, h = sc["width"], sc["height"]
        out = {"workload": cfg, "pairs": m, "gaussians": int(sc["start"].size(0))}
        bits = ck.pixel_key_bits(w, h)
        out["key_bits"] = bits
        only_carry = "--only-carry" in argv  # (A/B runs of csrc/gcp_pixels.hip: tools/pixels_ab.sh)
        out["create_alpha_brend_ms"] = timeit(lambda: ck.create_alpha_brend(rects, anti, "cumprod"), iters)  # auto: cut into boxes, walk
        if only_carry:
            argv = [a for a in argv if a not in ("--stages", "--chunked")] + ["--carry"]
            stages = False
        else:
            out["grad_cumsum_ms"] = timeit(lambda: ck.grad_cumsum(rects, grad), iters)
            out["rects_to_boxes_ms"] = timeit(lambda: raster.rects_to_boxes(rects), iters)
            out["sort_route_create_alpha_brend_ms"] = timeit(lambda: ck.create_alpha_brend(rects, anti, "cumprod", image_size=(w, h), route="sort"), iters)
            out["sort_route_create_alpha_brend_cumsum_ms"] = timeit(lambda: ck.create_alpha_brend(rects, anti, "cumsum", image_size=(w, h), route="sort"), iters)
            out["sort_route_grad_cumsum_ms"] = timeit(lambda: ck.grad_cumsum(rects, grad, image_size=(w, h), route="sort"), iters)
            out["sort_route_key_bits_ms"] = timeit(lambda: ck.create_alpha_brend(rects, anti, "cumprod", key_bits=bits, route="sort"), iters)
            out["sort_route_key_range_read_back_ms"] = timeit(lambda: ck.create_alpha_brend(rects, anti, "cumprod", route="sort"), iters)
            out["create_alpha_brend_boxes_ms"] = timeit(lambda: ck.create_alpha_brend_boxes(sc["start"], sc["end"], anti, w, h, "cumprod"), iters)
            out["grad_cumsum_boxes_ms"] = timeit(lambda: ck.grad_cumsum_boxes(sc["start"], sc["end"], grad, w, h), iters)
        if "--chunked" in argv:
            # a second chunk's call (gs_model.py:611-612): the image's pixels as carry rows in torch.unique's order, then the boxes
            xs = torch.arange(w + 1, device=dev, dtype=torch.int32)
            ys = torch.arange(h + 1, device=dev, dtype=torch.int32)
            carry = torch.stack([xs[:, None].expand(-1, h + 1).reshape(-1), ys[None, :].expand(w + 1, -1).reshape(-1)], 1)
            c = carry.size(0)
            lst = torch.cat([carry, rects])
            vals = torch.cat([torch.rand(c, device=dev) * 0.5 + 0.5, anti])
            out["chunked_call_carry_rows"] = c
            out["chunked_call_create_alpha_brend_ms"] = timeit(lambda: ck.create_alpha_brend(lst, vals, "cumprod", c), iters)
            out["chunked_call_sort_route_ms"] = timeit(lambda: ck.create_alpha_brend(lst, vals, "cumprod", c, image_size=(w, h), route="sort"), iters)
            del lst, vals, carry
        if "--carry" in argv:
            # row f3: the per-pixel carry of the chunk loop (gs_model.py:582-586, :724-730), and the reference's own statement
            # of it — torch.unique(dim=0) + scatter_reduce(amin) — run by torch on the same device
            T, _ = ck.create_alpha_brend(rects, anti, "cumprod")  # what the reference calls it on (gs_model.py:607-609): falls along a pixel's list
            if T.numel() != m:  # (pairs were dropped: the list would have to be thinned with them)
                T = anti
            out["create_alpha_brend_min_ms"] = timeit(lambda: ck.create_alpha_brend_min(rects, T, image_size=(w, h)), iters)
            out["create_alpha_brend_min_unordered_values_ms"] = timeit(lambda: ck.create_alpha_brend_min(rects, anti, image_size=(w, h)), iters)
            out["create_alpha_brend_min_extent_read_back_ms"] = timeit(lambda: ck.create_alpha_brend_min(rects, T), iters)
            out["create_grad_alphabrend_min_ms"] = timeit(lambda: ck.create_grad_alphabrend_min(rects, grad, image_size=(w, h)), iters)
            out["create_alpha_brend_min_int64_ms"] = None
            if m <= 200_000_000:
                r64 = rects.long()
                out["create_alpha_brend_min_int64_ms"] = timeit(lambda: ck.create_alpha_brend_min(r64, T, image_size=(w, h)), iters)
                del r64
            out["create_rects_ms"] = timeit(lambda: ck.create_rects(sc["start"], sc["end"]), iters)  # _create_rects (gs_model.py:480-482): expansion + the read-back of M
            F = ck.custom_autograd_grouped_cumprod

            def first_chunk():  # the scan side of `_forward_batch` for a first chunk (gs_model.py:601, :607-609), under the reference's names
                r = F._create_rects(sc["start"], sc["end"])
                t, mask = F._create_alpha_brend(r, anti, flag="cumprod")
                r = r[mask]
                return F._create_alpha_brend_min(r, t)

            out["first_chunk_scan_side_ms"] = timeit(first_chunk, iters)
            out["first_chunk_of_which_rects_mask_indexing_ms"] = timeit(lambda: rects[torch.ones(m, dtype=torch.bool, device=dev)], iters)
            u, _ = ck.create_alpha_brend_min(rects, T, image_size=(w, h))
            out["distinct_pixels"] = int(u.size(0))
            del u, T

            def torch_statement():
                unique_rects, inv = torch.unique(rects, return_inverse=True, dim=0)
                return torch.zeros_like(unique_rects[:, 0], dtype=torch.float32).scatter_reduce(0, inv, anti, reduce="amin", include_self=False)

            try:
                out["torch_unique_dim0_scatter_amin_ms"] = timeit(torch_statement, 2, 1)
            except RuntimeError as e:  # out of memory at the larger scenes: said, not hidden
                out["torch_unique_dim0_scatter_amin_ms"] = None
                out["torch_unique_dim0_error"] = str(e).splitlines()[0][:160]
                torch.cuda.empty_cache()
        if stages:
            import grouped_cumprod as gc

            st = {"sort_rects_ms": timeit(lambda: raster.sort_rects(rects, image_size=(w, h)), iters),
                  "sort_rects_key_bits_ms": timeit(lambda: raster.sort_rects(rects, bits), iters)}
            sk, idx = raster.sort_rects(rects, image_size=(w, h))
            incl = torch.empty_like(anti)
            st["indexed_scan_ms"] = timeit(lambda: gc.grouped_cumprod_forward_indexed(anti, sk, idx, incl), iters)
            st["indexed_reverse_scan_ms"] = timeit(lambda: gc.grouped_cumsum_reverse_indexed(grad, sk, idx, incl), iters)
            gc.grouped_cumprod_forward_indexed(anti, sk, idx, incl)
            st["compact_finish_ms"] = timeit(lambda: raster.compact_finish(incl, anti, 0), iters)
            vals, keep = raster.compact_finish(incl, anti, 0)
            st["dropped_pairs"] = int(m - vals.numel())
            st["key_range_ms"] = timeit(lambda: raster.rects_key_bits(rects), iters)
            # the boxes route, stage by stage
            st["boxes_bin_tiles_ms"] = timeit(lambda: raster.bin_tiles(sc["start"], sc["end"], w, h), iters)
            bins = raster.bin_tiles(sc["start"], sc["end"], w, h)
            st["boxes_box_offsets_ms"] = timeit(lambda: raster.box_offsets(sc["start"], sc["end"], w, h), iters)
            boff = raster.box_offsets(sc["start"], sc["end"], w, h)
            st["boxes_walk_cumprod_ms"] = timeit(lambda: raster.finish_boxes(bins, sc["start"], sc["end"], boff, anti, 0), iters)
            st["boxes_walk_reverse_ms"] = timeit(lambda: raster.finish_boxes(bins, sc["start"], sc["end"], boff, grad, 2), iters)
            fin_b, keep_b, dropped = raster.finish_boxes(bins, sc["start"], sc["end"], boff, anti, 0)
            st["boxes_kept_count_ms"] = timeit(lambda: raster.compact_kept(fin_b, keep_b, dropped), iters)
            st["boxes_dropped_pairs"] = int(m - raster.compact_kept(fin_b, keep_b, dropped)[0].numel())
            del fin_b, keep_b, dropped
            # the cut of the default route and its binning (the counting pass is the cut's)
            st["one_call_cut_ms"] = timeit(lambda: raster.rects_to_boxes(rects), iters)
            st["step_by_step_cut_ms"] = timeit(lambda: raster.rects_to_boxes(rects, one_call=False), iters)
            rb = raster.rects_to_boxes(rects)
            if rb is not None:
                st["bin_after_cut_ms"] = timeit(lambda: rb.bin(), iters)
                st["rectangles"] = int(rb.start.size(0))
            del rb
            st["tile_entries"] = bins.n_tile_pairs
            del bins, boff
            y = torch.empty_like(anti)
            st["plain_scan_same_size_ms"] = timeit(lambda: gc.grouped_cumprod_forward(anti, sk, y), iters)
            del sk, idx, incl, y, vals, keep
            key = ck.unique(rects).contiguous()
            st["torch_sort_stable_ms"] = timeit(lambda: torch.sort(key, stable=True), 3, 1)
            del key
            out["stages"] = st
        print(json.dumps(out), flush=True)
        del rects, anti, grad, sc
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
