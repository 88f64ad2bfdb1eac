#!/usr/bin/env python3
"""Headline benchmark: splat-pixel pairs blended per second, grouped cumprod forward + backward.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no launcher in the environment (no RANK / WORLD_SIZE) the first form starts the second one itself as
a child process and returns its exit code (`launch_ranks`): one process per GPU either way.

One "step" = one pass of the hot path over one synthetic pair list resident in HBM:
grouped_cumprod_forward (12 B/pair) then grouped_cumprod_backward (20 B/pair) — the two
kernels BASELINE.json's metric is quoted on.  Workload at every N: each rank
owns one cfg3-sized image band (1920x1080 pixels, mean 80 splats/pixel, heavy-tailed, M ~ 1.66e8
pairs); pixel groups are independent so there is NO data-path collective (weak scaling).

Beside the headline (outside its timed region) the line carries: `config.unclipped` (the same workload without the
4096 clip of SURVEY.md §8d: the few pixels deeper than one tile's raw look-back window go through the scans' descriptor
tree), `sharded_frames` (ONE cfg5 / cfg3 frame cut into N slices by sharding.partition_groups: strong scaling, with and
without the frame gather + gradient scatter — BASELINE.json config 5), `function_level` (rows f1/f2, with the blend
backward's issue roofline and lane use), `caller_level` (row f4).

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the backward scan):
algorithmic bytes (20 B x pairs per launch) / its mean duration measured with HIP events on the
launch stream inside the timed region.  `cpu_baseline` times the oracle's sequential C port of the
same two ops on the host cores, on a bounded sample of the same pair list (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

# host threads that finished a parallel region go to sleep instead of spinning: the CPU baseline's legs (OpenMP C port,
# torch's own pool) otherwise disturb each other on the shared cores of the GPU box
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_FWD, BYTES_BWD = 12, 20  # algorithmic bytes per pair


def _median_time(fn, reps=5, warmup=1, min_s=0.25):
    """median of `reps` after `warmup` untimed calls (the first call also takes the page faults of freshly
    allocated outputs and starts the thread pool: it is never the one reported).  A leg shorter than `min_s` is repeated
    inside every timed repetition until it lasts that long (a 10 ms call on a shared host is not a measurement);
    returns seconds per call: (median, min, max)."""
    t0 = time.perf_counter()
    for _ in range(warmup):
        fn()
    once = max((time.perf_counter() - t0) / max(warmup, 1), 1e-6)
    inner = max(1, min(64, int(min_s / once + 0.999)))
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(inner):
            fn()
        ts.append((time.perf_counter() - t0) / inner)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


def cpu_baseline(p, sample_pairs):
    """Oracle (C port of the reference kernels, oracle/gcp_oracle.c) on the first groups of the pair list, on the host
    cores of this box: OpenMP over pixel groups with up to 16 threads (one GPU's CPU share) = the headline figure,
    plus single-thread legs (fwd+bwd, and forward only beside the reference's OWN forward compiled for the host) and the
    pure-PyTorch torch.cumprod path.  Every leg: one warm-up call, then the median of 5 (min / max kept beside it)."""
    import torch

    from oracle import c_oracle as co
    from oracle import ref as oref
    from oracle import torch_path as tp

    inv_len = p.inv_len.cpu()

    def prefix(pairs):
        g = int(torch.searchsorted(inv_len, torch.tensor(pairs, dtype=inv_len.dtype)).item())
        g = max(1, min(g, inv_len.numel()))
        return g, int(inv_len[g - 1].item())

    def leg(value_pairs, t, **extra):
        med, lo, hi = t
        return {"value": value_pairs / med, "unit": "pairs/s", "median_ms": med * 1e3, "min_ms": lo * 1e3, "max_ms": hi * 1e3,
                "timing": "1 warm-up + median of 5 (calls shorter than 0.25 s repeated inside each timed repetition)", **extra}

    g, s = prefix(sample_pairs)
    x, key, inv, go = (t[:s].cpu().contiguous() for t in (p.x, p.key, p.inv, p.grad_out))
    il = inv_len[:g].contiguous()
    threads = max(1, min(16, os.cpu_count() or 1, co.max_threads()))

    class one_core:
        """Single-thread legs run pinned to one of the cores this process may use: on a shared host an unpinned thread
        is moved between cores (and memory domains) from one run to the next, worth 10 % of a 20 ms leg."""

        def __enter__(self):
            self.old = None
            try:
                self.old = os.sched_getaffinity(0)
                os.sched_setaffinity(0, {sorted(self.old)[len(self.old) // 2]})
            except (AttributeError, OSError):
                self.old = None

        def __exit__(self, *exc):
            if self.old:
                os.sched_setaffinity(0, self.old)

    # single thread first (pinned; before any OpenMP team exists), a quarter of the sample
    g1, s1 = prefix(max(1, sample_pairs // 4))
    x1, k1, i1, go1, il1 = x[:s1].contiguous(), key[:s1].contiguous(), inv[:s1].contiguous(), go[:s1].contiguous(), inv_len[:g1].contiguous()
    ref_fwd = None
    with one_core():
        y1 = co.cumprod_forward(x1, k1)
        t_f1 = _median_time(lambda: co.cumprod_forward(x1, k1), reps=5, warmup=2, min_s=0.5)
        t_b1 = _median_time(lambda: co.cumprod_backward(x1, y1, go1, i1, il1), reps=5, warmup=1)
        # the reference's OWN forward (grouped_cumprod_forward.cu, unmodified, rocThrust CPP backend: sequential) where the
        # build container compiled it into oracle/_ref/; output allocated and touched before the timed calls
        host = oref.load("grouped_cumprod_ref_host")
        if host is not None:
            y_ref = torch.zeros_like(x1)
            t_ref = _median_time(lambda: host.grouped_cumprod_forward(x1, k1, y_ref), reps=5, warmup=2, min_s=0.5)
            ref_fwd = leg(s1, t_ref, unit="pairs/s (forward only)", cores=1, kind="reference",
                          sample=f"{s1} pairs, cuda_kernel/grouped_cumprod_forward.cu compiled for the host (oracle/_ref)",
                          equals_port=bool(torch.equal(y_ref, y1)))
    # the "pure-PyTorch torch.cumprod path" of BASELINE.json (per-group torch.cumprod), forward only
    g8, s8 = prefix(min(4_000_000, s))
    x8, k8 = x[:s8].contiguous(), key[:s8].contiguous()
    # (one torch thread, pinned like the other single-thread legs: with torch's own pool the leg — large temporaries, the
    # allocator, page faults — moved by +-20 % from one run to the next on the shared host)
    old_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    with one_core():
        t_tp = _median_time(lambda: tp.grouped_cumprod(x8, k8), reps=5, warmup=2, min_s=0.5)
    torch.set_num_threads(old_threads)
    # OpenMP over pixel groups: the headline figure
    y = co.cumprod_forward_mt(x, il, threads)
    t_f = _median_time(lambda: co.cumprod_forward_mt(x, il, threads), reps=5, warmup=1)
    t_b = _median_time(lambda: co.cumprod_backward_mt(x, y, go, inv, il, threads), reps=5, warmup=1)
    med = t_f[0] + t_b[0]
    return {
        "value": s / med,
        "unit": "pairs/s",
        "cores": threads,
        "kind": "port",
        "timing": "per leg: 1 warm-up + median of 5; value = sample / (median fwd + median bwd)",
        "sample": f"first {g} pixel groups = {s} pairs of the same pair list; oracle/gcp_oracle.c (fp32, every group scanned "
        f"left to right, literal O(L^2) backward loop of the reference), OpenMP over groups; fwd {1e3*t_f[0]:.1f} ms "
        f"[{1e3*t_f[1]:.1f}, {1e3*t_f[2]:.1f}], bwd {1e3*t_b[0]:.1f} ms [{1e3*t_b[1]:.1f}, {1e3*t_b[2]:.1f}]; host has {os.cpu_count()} cpus",
        "single_thread": {"value": s1 / (t_f1[0] + t_b1[0]), "unit": "pairs/s", "cores": 1, "sample": f"{s1} pairs",
                          "fwd_ms": t_f1[0] * 1e3, "bwd_ms": t_b1[0] * 1e3},
        "port_forward_single_thread": leg(s1, t_f1, unit="pairs/s (forward only)", cores=1, kind="port",
                                          sample=f"{s1} pairs, oracle/gcp_oracle.c forward alone — compare with reference_host_forward"),
        "torch_cumprod_path_forward": leg(s8, t_tp, unit="pairs/s (forward only)", threads=1,
                                          sample=f"{s8} pairs, oracle/torch_path.py (per-group torch.cumprod)"),
        "reference_host_forward": ref_fwd,
    }


VALU_PEAK_WAVE_INSTS = 1024 * 2.4e9 / 4  # 1024 SIMDs, one wave64 VALU instruction per 4 cycles (16 lanes wide) at 2.4 GHz


def lane_use(bins, start, end):
    """How full the waves of the blend kernels are: a (wave, list entry) visit costs the same VALU issue slots whether 1
    or 64 of the wave's pixels (4 rows x 16 columns) lie inside the entry's box.  Computed from the tile lists: per
    (tile, Gaussian) entry the box's columns and rows inside the tile -> pixels inside, and the waves (4-row groups) it
    reaches.  Returns (wave-entry visits, lanes inside a box / lanes issued)."""
    import torch

    K = bins.n_tile_pairs
    if K == 0:
        return 0, None
    dev = start.device
    tile = torch.searchsorted(bins.tile_start[1:].contiguous(), torch.arange(K, device=dev, dtype=torch.int32), right=True)
    g = bins.tile_list.long()
    tx0 = (tile % bins.tiles_x) * 16
    ty0 = (tile // bins.tiles_x) * 16
    x0 = torch.maximum(start[g, 0].clamp(min=0).long(), tx0)
    x1 = torch.minimum(end[g, 0].clamp(max=bins.width).long(), tx0 + 15)
    y0 = torch.maximum(start[g, 1].clamp(min=0).long(), ty0)
    y1 = torch.minimum(end[g, 1].clamp(max=bins.height).long(), ty0 + 15)
    cols = (x1 - x0 + 1).clamp(min=0)
    rows = (y1 - y0 + 1).clamp(min=0)
    waves = ((y1 - ty0) // 4 - (y0 - ty0) // 4 + 1).clamp(min=0) * (rows > 0)
    visits = int(waves.sum())
    return visits, float((cols * rows).sum()) / (64.0 * max(visits, 1))


def pmc_function(kernel, workload, counter):
    """A counter of the committed rocprofv3 --pmc pass over tools/raster_bench.py (profiles/pmc_function.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_function.json")) as f:
            return json.load(f)[workload][kernel][counter]
    except (OSError, KeyError, ValueError):
        return None


# algorithmic bytes per pair of rows a5 / a6, stage by stage
WRAPPER_BYTES = {
    "sort": {"pass 0: histogram reads the rects": 8, "pass 0: scatter reads the rects, writes key + index": 16,
             "passes 1, 2: histogram reads the keys": 8, "passes 1, 2: scatter reads and writes key + index": 32},
    "indexed scan": {"sorted key": 4, "permutation": 4, "gathered value": 4, "scattered inclusive value": 4},
    "compaction": {"count pass reads the inclusive values": 4, "write pass reads inclusive + self": 8, "mask": 1, "kept values": 4},
}
# the walk writes the FINAL values (inclusive / self, inclusive - self) and clears the mask byte of a pair it drops: when nothing
# drops — the usual case, and this workload's — there is no compaction pass (else + 5 B read, 4 B written per pair)
WRAPPER_BOX_BYTES = {"tile-list walk": {"value read": 4, "final value written": 4}, "mask": {"fill with ones": 1}}
# the default route: the rect list cut back into rectangles (one read of the list; its row / rectangle records are ~1/13 of a
# pair each, rounded up to 1 B/pair), then the boxes route
WRAPPER_AUTO_BYTES = {"rect list -> rows -> rectangles": {"rects read once": 8, "row and rectangle records": 1}, **WRAPPER_BOX_BYTES}


def wrapper_level(dev, workload):
    """Rows a5 / a6 — the reference's only live callers of the scans (`_create_alpha_brend`, gs_model.py:544-566, and
    `grad_cumsum`, :716-722; call sites :607, :612, :636) — on the pair list of one camera of a scene of the workload's
    shape: whole-call times (HIP events around the Python call, median of 5 after 2 warm-ups — 9 after 0.5 s of warm-up calls for the first figure of the block, 7 for the second; each call ends with the one
    device->host read of the kept count that sizes its result, as the reference's boolean-mask indexing does), the stages
    on their own, and a roofline per route from the byte model above."""
    import torch

    import cuda_kernel as ck
    import grouped_cumprod as gc
    from simplegaussiansplat_tk71_amd import raster, synthetic

    if workload not in synthetic.CONFIGS:
        return None
    sc, rects, anti, grad = synthetic.make_scene_pairs(workload, seed=0, device=dev)
    m = rects.size(0)
    w, h = sc["width"], sc["height"]
    bits = ck.pixel_key_bits(w, h)

    def timed(fn, iters=5, warmup=2, warm_s=0.0):
        t_w = time.perf_counter()
        k = 0
        while k < warmup or time.perf_counter() - t_w < warm_s:  # warm_s: the first figure of the block also waits for the clocks
            fn()
            k += 1
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

    # the reference's own call, nothing but (rects, values, flag): the list is cut back into boxes and walked
    # (the first calls also settle the caching allocator's pool and follow a stretch of host work with the GPU idle: warm up for 0.5 s —
    # with 0.1 s the first figure came out 4 % above the second, which does the same work)
    t_a5_auto = timed(lambda: ck.create_alpha_brend(rects, anti, "cumprod"), iters=9, warmup=4, warm_s=0.5)
    t_a6_auto = timed(lambda: ck.grad_cumsum(rects, grad), iters=7, warmup=2)
    t_cut = timed(lambda: raster.rects_to_boxes(rects))
    # the dtype the reference's own make_rect_points_parallel returns (uitility.py:336-366): int64, 16 B per pair, read as it is
    rects64 = rects.long()
    t_a5_i64 = timed(lambda: ck.create_alpha_brend(rects64, anti, "cumprod"))
    del rects64
    prep = ck.PreparedRects(rects)  # cut + binning once for the forward's and the backward's call on the same list
    t_a5_prep = timed(lambda: ck.create_alpha_brend(prep, anti, "cumprod"))
    t_a6_prep = timed(lambda: ck.grad_cumsum(prep, grad))
    del prep
    # the general route (any list of coordinates): stable radix sort, indexed scan
    t_a5 = timed(lambda: ck.create_alpha_brend(rects, anti, "cumprod", image_size=(w, h), route="sort"))
    t_a6 = timed(lambda: ck.grad_cumsum(rects, grad, image_size=(w, h), route="sort"))
    t_a5_kb = timed(lambda: ck.create_alpha_brend(rects, anti, "cumprod", key_bits=bits, route="sort"))
    t_a5_rb = timed(lambda: ck.create_alpha_brend(rects, anti, "cumprod", route="sort"))
    t_a5b = timed(lambda: ck.create_alpha_brend_boxes(sc["start"], sc["end"], anti, w, h, "cumprod"))
    t_a6b = timed(lambda: ck.grad_cumsum_boxes(sc["start"], sc["end"], grad, w, h))
    # stages of the rects route
    t_sort = timed(lambda: raster.sort_rects(rects, image_size=(w, h)))
    sk, idx = raster.sort_rects(rects, image_size=(w, h))
    incl = torch.empty_like(anti)
    t_scan = timed(lambda: gc.grouped_cumprod_forward_indexed(anti, sk, idx, incl))
    t_rev = timed(lambda: gc.grouped_cumsum_reverse_indexed(grad, sk, idx, incl))
    gc.grouped_cumprod_forward_indexed(anti, sk, idx, incl)
    t_comp = timed(lambda: raster.compact_finish(incl, anti, 0))
    kept = raster.compact_finish(incl, anti, 0)[0].numel()
    del sk, idx
    # stages of the boxes route
    t_bin = timed(lambda: raster.bin_tiles(sc["start"], sc["end"], w, h))
    bins = raster.bin_tiles(sc["start"], sc["end"], w, h)
    boff = raster.box_offsets(sc["start"], sc["end"], w, h)
    t_walk = timed(lambda: raster.finish_boxes(bins, sc["start"], sc["end"], boff, anti, 0))
    fin_b, keep_b, dropped = raster.finish_boxes(bins, sc["start"], sc["end"], boff, anti, 0)
    t_comp_b = timed(lambda: raster.compact_kept(fin_b, keep_b, dropped))
    n_dropped = m - int(raster.compact_kept(fin_b, keep_b, dropped)[0].numel())
    del fin_b, keep_b, dropped
    # the cut of the default route: one call, one read-back; and its binning, whose counting pass the cut has done
    rb = raster.rects_to_boxes(rects)
    t_bin_counted = timed(lambda: rb.bin()) if rb is not None else None
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats(dev)
    held = torch.cuda.memory_allocated(dev)
    raster.rects_to_boxes(rects)
    cut_scratch = (torch.cuda.max_memory_allocated(dev) - held) / m
    del rb
    # row f3: the per-pixel carry every chunk of the reference's loop ends with (gs_model.py:582-586, :724-730)
    T_all = ck.create_alpha_brend(rects, anti, "cumprod")[0]
    T_all = T_all if T_all.numel() == m else anti  # (pairs dropped: the reference thins the list with them; values of any order do)
    t_min = timed(lambda: ck.create_alpha_brend_min(rects, T_all, image_size=(w, h)))
    t_min_ext = timed(lambda: ck.create_alpha_brend_min(rects, T_all))
    t_first = timed(lambda: ck.create_grad_alphabrend_min(rects, grad, image_size=(w, h)))
    n_pixels = int(ck.create_alpha_brend_min(rects, T_all, image_size=(w, h))[0].size(0))
    del T_all
    b_rects = sum(sum(v.values()) for v in WRAPPER_BYTES.values())
    b_boxes = sum(sum(v.values()) for v in WRAPPER_BOX_BYTES.values())
    b_auto = sum(sum(v.values()) for v in WRAPPER_AUTO_BYTES.values())

    def roof(bytes_per_pair, ms):
        ach = bytes_per_pair * m / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                "algorithmic_bytes_per_pair": bytes_per_pair}

    return {
        "what": "_create_alpha_brend / grad_cumsum (gs_model.py:544-566, :716-722) on one camera's pair list: Gaussian-major rects, "
                "stable sort by pixel key, per-pixel scan, un-sort, != 0 compaction, / self | - self",
        "default_route": "rect list cut back into boxes (one call), binned, walked; the walk writes final values and the mask",
        "workload": f"{workload} scene: {w + 1}x{h + 1}, {int(sc['start'].size(0))} Gaussians",
        "pairs": m,
        "kept_pairs": kept,
        "sort": f"compact pixel ids y*{w + 1}+x: {max(1, (h * (w + 1) + w).bit_length())} bits in 3 passes (image_size given); "
                f"{bits}-bit keys y*10000+x with key_bits",
        # the call as the reference makes it: create_alpha_brend(rects, anti_opacity, flag) / grad_cumsum(rects, grad)
        "create_alpha_brend_ms": t_a5_auto,
        "grad_cumsum_ms": t_a6_auto,
        "create_alpha_brend_pairs_per_s": m / (t_a5_auto * 1e-3),
        "grad_cumsum_pairs_per_s": m / (t_a6_auto * 1e-3),
        "route": "auto -> boxes: the rect list (a concatenation of row-major boxes, uitility.py:336-366) is cut back into rectangles, "
                 "binned into tiles and walked; no M-sized sort",
        "roofline": roof(b_auto, t_a5_auto),
        "create_alpha_brend_int64_rects_ms": t_a5_i64,
        "with_prepared_rects_ms": {"create_alpha_brend": t_a5_prep, "grad_cumsum": t_a6_prep,
                                   "what": "cuda_kernel.PreparedRects(rects): cut + binning done once for the calls of one step"},
        "stages_ms": {"rects_to_boxes (rows, rectangles, boxes, tiles per box: one call, one read-back)": t_cut,
                      "bin_tiles (counting pass done by the cut)": t_bin_counted,
                      "tile-list walk writing final values + mask": t_walk,
                      "kept count + the read-back that sizes the result (no compaction pass: nothing dropped)": t_comp_b},
        "dropped_pairs": n_dropped,
        "device_to_host_reads_per_call": 2,
        "cut_scratch_bytes_per_pair": cut_scratch,
        "byte_model": WRAPPER_AUTO_BYTES,
        "general_sort_route": {
            "what": "route='sort': any list of pixel coordinates — key-in-sort stable radix sort, one indexed scan, the same compaction",
            "create_alpha_brend_ms": t_a5,
            "create_alpha_brend_key_bits_ms": t_a5_kb,
            "grad_cumsum_ms": t_a6,
            "create_alpha_brend_key_range_read_back_ms": t_a5_rb,
            "pairs_per_s": m / (t_a5 * 1e-3),
            "roofline": roof(b_rects, t_a5),
            "stages_ms": {"sort_rects (3 radix passes)": t_sort, "indexed scan (cumprod)": t_scan, "indexed scan (suffix sum)": t_rev,
                          "compact_finish": t_comp},
            "stage_rooflines": {"sort": roof(sum(WRAPPER_BYTES["sort"].values()), t_sort),
                                "indexed scan": roof(sum(WRAPPER_BYTES["indexed scan"].values()), t_scan),
                                "compaction": roof(sum(WRAPPER_BYTES["compaction"].values()), t_comp)},
            "byte_model": WRAPPER_BYTES,
        },
        "chunk_carry": {
            "what": "_create_alpha_brend_min / create_grad_alphabrend_min (gs_model.py:582-586, :724-730; every chunk of the reference's "
                    "forward / backward): torch.unique(rects, dim=0) + scatter_reduce(amin) as ONE pass of per-pixel integer minima "
                    "into an image-sized table, read out in (x, y) order (csrc/gcp_pixels.hip)",
            "create_alpha_brend_min_ms": t_min,
            "create_alpha_brend_min_without_image_size_ms": t_min_ext,  # (the list's extent is measured once per device and remembered)
            "create_grad_alphabrend_min_ms": t_first,
            "distinct_pixels": n_pixels,
            "roofline": roof(12, t_min),
            "roofline_first_pair": roof(8, t_first),
            "byte_model": {"rects read once": 8, "value read (not for the first-pair index)": 4},
        },
        "from_boxes": {
            "what": "the same results from the boxes the rects were expanded from: tile binning + one walk of the tile lists + the same compaction",
            "create_alpha_brend_boxes_ms": t_a5b,
            "grad_cumsum_boxes_ms": t_a6b,
            "pairs_per_s": m / (t_a5b * 1e-3),
            "stages_ms": {"bin_tiles": t_bin, "tile-list walk writing final values + mask": t_walk,
                          "kept count + read-back (no compaction pass: nothing dropped)": t_comp_b},
            "tile_entries": bins.n_tile_pairs,
            "roofline": roof(b_boxes, t_a5b),
            "byte_model": WRAPPER_BOX_BYTES,
        },
    }


def function_level(dev, workload):
    """Rows f1/f2 (outside the timed region, informational): the whole rasterise-and-blend Function of the
    reference (gs_model.py:666-692, :786-820) on a scene of the same shape — tile binning, fused blend
    forward, fused blend backward — in splat-pixel pairs per second."""
    import torch

    from simplegaussiansplat_tk71_amd import raster, synthetic

    if workload not in synthetic.CONFIGS:
        return None
    sc = synthetic.make_scene_config(workload, seed=0, device=dev)
    w, h = sc["width"], sc["height"]
    pairs = int(sc["boxsize"].sum().item())
    params = (sc["start"], sc["end"], sc["mean"], sc["vinv"], sc["opacity"], sc["l_d"])

    def timed(fn, iters=5):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            r = fn()
        b.record()
        torch.cuda.synchronize()
        return r, a.elapsed_time(b) / iters

    bins, t_bin = timed(lambda: raster.bin_tiles(sc["start"], sc["end"], w, h))
    (img, ckpt), t_fwd = timed(lambda: raster.blend_forward(bins, *params, with_checkpoints=True))
    gimg = torch.randn_like(img)
    _, t_bwd = timed(lambda: raster.blend_backward(bins, *params, ckpt, gimg))
    visits, lane_frac = lane_use(bins, sc["start"], sc["end"])
    insts = pmc_function("k_blend_bwd", workload, "SQ_INSTS_VALU")
    kernel_us = pmc_function("k_blend_bwd", workload, "avg_us_rocprof")  # the kernel alone, same committed profile
    roof = {
        "kernel": "k_blend_bwd (fused blend backward; HBM traffic is K x 64 B + checkpoints + image: far from the HBM roof)",
        "bound": "valu-issue",
        "insts": insts,  # VALU wave-instructions per launch (rocprofv3 --pmc SQ_INSTS_VALU, profiles/pmc_function.json)
        "peak_wave_insts_per_s": VALU_PEAK_WAVE_INSTS,
        # live: HIP events around raster.blend_backward = k_blend_bwd + k_grad_reduce + the host call
        "achieved_wave_insts_per_s": None if insts is None else insts / (t_bwd * 1e-3),
        "frac": None if insts is None else insts / (t_bwd * 1e-3) / VALU_PEAK_WAVE_INSTS,
        # the kernel alone, from the committed rocprofv3 pass (profiles/r02_function_kernels.md)
        "kernel_us_rocprof": kernel_us,
        "frac_kernel_alone": None if insts is None or not kernel_us else insts / (kernel_us * 1e-6) / VALU_PEAK_WAVE_INSTS,
        "wave_entry_visits": visits,
        "valu_per_visit": None if insts is None or not visits else insts / visits,
        "active_lane_frac": lane_frac,  # pixels inside the visited entry's box / 64 lanes issued
        "avg_launch_us": t_bwd * 1e3,
    }
    return {
        "what": "custom_autograd_grouped_cumprod: tile binning + fused blend forward + backward (no pair list materialised)",
        "gaussians": int(sc["start"].size(0)),
        "pairs": pairs,
        "tile_pairs": bins.n_tile_pairs,
        "bin_ms": t_bin,
        "forward_ms": t_fwd,
        "backward_ms": t_bwd,
        "pairs_per_s": pairs / ((t_bin + t_fwd + t_bwd) * 1e-3),
        "roofline": roof,
    }


def caller_level(dev, workload):
    """Row f4 (outside the timed region, informational): one training step of the caller — camera projection,
    Function, L1 + D-SSIM loss, backward to the five parameter tensors — on 10^6 Gaussians and one 1920x1080 camera,
    with the fused projection and loss kernels.  (The reference's op-by-op PyTorch formulation of projection and loss,
    gs_model.py:277-425 / gs_control.py:180-182, around the same Function is timed by tests/bench_reference_caller_gpu.py.)"""
    import time

    import torch

    from simplegaussiansplat_tk71_amd import gs_model as gm
    from simplegaussiansplat_tk71_amd import synthetic

    if workload != "cfg3":
        return None
    width, height, n = 1920, 1080, 1_000_000
    P, K, wh = synthetic.ring_cameras(1, width, height, device=dev)
    model = gm.GS_model_with_param(*synthetic.make_world(n, width, 2.0, seed=0, device=dev))
    target = torch.rand(1, 3, height, width, device=dev)

    def step():
        images = model(P, K, wh, [0])[0]
        gm.splat_loss(images, target, 0.2).backward()  # gs_control.py:180-182
        model._optimizer.zero_grad(set_to_none=True)

    def timed(reps):
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    cams, _, _ = model.camera_inputs(P, K, wh)
    pairs = int(cams[0]["boxsize"].sum())
    fused_ms = timed(5)
    return {
        "what": "GS_model_with_param: projection + Function + L1/D-SSIM loss, forward and backward, one 1920x1080 camera",
        "gaussians": n,
        "visible": int(cams[0]["boxsize"].numel()),
        "pairs": pairs,
        "step_ms": fused_ms,
        "reference_formulation_ms": "179 (projection and loss as PyTorch ops around the same Function; tests/bench_reference_caller_gpu.py)",
        "pairs_per_s": pairs / (fused_ms * 1e-3),
    }


def sharded_frame(name, world, rank, device, scan_fwd_bwd, sync, steps=10, warmup=3, collectives=True, max_run=None):
    """BASELINE.json config 5's shape (and the strong-scaling form of config 3): ONE frame of config `name`, its pair
    list cut into `world` slices at pixel-group boundaries (sharding.partition_groups), every rank scanning its slice
    with no data-path collective; then the only exchange the design has — one gather of per-group rows to rank 0 (frame
    assembly) and one scatter back (dL/dI distribution) — timed on its own.  Returns a dict on every rank (identical
    numbers: times are max over ranks).  `scan_fwd_bwd(p)` runs grouped_cumprod forward + backward on a PairList (the HIP
    library in bench.py; tests inject the oracle to rehearse this exact control flow on CPU with gloo);
    `sync()` = barrier + device synchronise."""
    import torch
    import torch.distributed as dist

    from simplegaussiansplat_tk71_amd import sharding, synthetic

    kw = {} if max_run is None else {"max_run": max_run}
    # build this rank's slice; if ANY rank fails here (out of memory, ...) every rank skips the block together — a rank
    # that raised on its own would leave the others waiting in the next collective
    p = shards = None
    total_pairs, fail = 0, None
    try:
        p, shards, total_pairs = synthetic.make_config_slice(name, world, rank, seed=0, device=device, **kw)
    except Exception as e:  # noqa: BLE001 - reported in the JSON line
        fail = repr(e)
    if world > 1:
        ok = torch.tensor([0.0 if fail else 1.0], dtype=torch.float64, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 0.0:
            return {"workload": name, "skipped": fail or "another rank failed to build its slice"}
    elif fail:
        return {"workload": name, "skipped": fail}
    for _ in range(warmup):
        scan_fwd_bwd(p)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        scan_fwd_bwd(p)
    sync()
    t_scan = (time.perf_counter() - t0) / steps
    t_scan_local = t_scan
    out = {
        "workload": f"{name}: ONE {synthetic.CONFIGS[name]['width']}x{synthetic.CONFIGS[name]['height']} frame, "
                    f"{total_pairs} pairs, cut into {world} slice(s) at pixel-group boundaries nearest k*M/N",
        "total_pairs": total_pairs,
        "pairs_per_rank": [s.n_pairs for s in shards],
        "groups_per_rank": [s.n_groups for s in shards],
        "steps": steps,
    }
    gather_s = scatter_s = None
    err = None
    # rank 0's rate on its own slice (its slice pairs / its own time, before the max over ranks) = the single-GPU reference
    slice_rate0 = shards[0].n_pairs / t_scan_local if rank == 0 else 0.0
    if world > 1:
        t = torch.tensor([t_scan, slice_rate0], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_scan, slice_rate0 = float(t[0].item()), float(t[1].item())
        if collectives:
            # everything a rank does on its own before a collective is agreed on collectively first: a rank that raised
            # alone would leave the others waiting inside gather / scatter / all_reduce
            rows, fail = None, None
            try:
                rows = torch.rand(p.n_groups, 3, device=device)  # per-pixel colour of this slice's groups
            except Exception as e:  # noqa: BLE001 - reported in the JSON line
                fail = repr(e)
            ok = torch.tensor([0.0 if fail else 1.0], dtype=torch.float64, device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) == 0.0:
                err = fail or "another rank could not allocate its rows; collectives skipped on every rank"
            else:
                # a failure INSIDE a collective (RCCL refusing to come up hits every rank alike) is recorded in the JSON
                # line, never swallowed; the rank then takes no further part in this block's collectives
                try:
                    tg = ts = 0.0
                    reps = 3
                    for it in range(reps + 1):
                        sync()
                        t1 = time.perf_counter()
                        full = sharding.gather_groups(rows, shards, dst=0)
                        sync()
                        t2 = time.perf_counter()
                        back = sharding.scatter_groups(full, shards, like=rows, src=0)
                        sync()
                        t3 = time.perf_counter()
                        if it:  # first round = RCCL channel set-up
                            tg += t2 - t1
                            ts += t3 - t2
                    bad = 0.0 if torch.equal(back, rows) else 1.0  # round trip must be the identity
                    tt = torch.tensor([tg / reps, ts / reps, bad], dtype=torch.float64, device=device)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    if float(tt[2]) != 0.0:
                        err = "gather/scatter round trip is not the identity"
                    else:
                        gather_s, scatter_s = float(tt[0]), float(tt[1])
                except Exception as e:  # noqa: BLE001
                    err = repr(e)
    bytes_pp = BYTES_FWD + BYTES_BWD
    out.update({
        "scan_ms_per_step": t_scan * 1e3,
        "pairs_per_s_scan_only": total_pairs / t_scan,
        "per_gpu_pairs_per_s": total_pairs / t_scan / world,
        "per_gpu_GBps": bytes_pp * total_pairs / t_scan / 1e9 / world,
        # strong scaling: frame rate / (N x the rate rank 0 reached on ITS slice, timed on its own before the max over ranks)
        "efficiency_vs_single_gpu": (total_pairs / t_scan) / (world * slice_rate0) if slice_rate0 > 0 else None,
        "rank0_slice_pairs_per_s": slice_rate0,
        "frame_gather_ms": None if gather_s is None else gather_s * 1e3,
        "grad_scatter_ms": None if scatter_s is None else scatter_s * 1e3,
        "pairs_per_s_with_gather_scatter": None if gather_s is None else total_pairs / (t_scan + gather_s + scatter_s),
        "collective_error": err,
    })
    return out


def pmc_traffic(kernel, workload):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)[workload][kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a fresh child — `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> bench.py <same arguments>`, the command
    the module docstring gives — wait for it and return its exit code (rank 0's JSON line goes straight to the inherited
    stdout).  This parent never touches the GPU (no torch.cuda / HIP call, and no exec from a process that did): it only
    makes sure the HIP library is built once, so that N ranks do not compile it side by side."""
    import signal
    import socket
    import subprocess

    from simplegaussiansplat_tk71_amd import _build

    if _build.is_stale():
        _build.build_hip_library(force=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    print(f"[bench] no launcher in the environment: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env)

    def forward(signum, _frame):  # a driver that stops this process stops the ranks too
        if child.poll() is None:
            child.send_signal(signum)

    old = {sg: signal.signal(sg, forward) for sg in (signal.SIGTERM, signal.SIGINT)}
    try:
        rc = child.wait()
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)
    return rc if rc >= 0 else 128 - rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg3",
                    help="cfg2 | cfg3 | cfg5band (one 8th of the 4K frame) | cfg3_unclipped (cfg3's run-length mix without the "
                         "clip at 4096 splats per pixel: a few pixels deeper than one scan tile)")
    ap.add_argument("--cpu-sample", type=int, default=64_000_000, help="pairs in the CPU baseline sample (0 = skip)")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit(f"--gpus {args.gpus}: expected a positive rank count")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: no launcher has set up the ranks, so this process becomes the launcher
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # the HIP library is prebuilt in-tree; if it is missing, local rank 0 builds it and the others wait
    from simplegaussiansplat_tk71_amd import _build

    if _build.is_stale():  # missing, or built from other sources than this checkout's
        if local_rank == 0:
            _build.build_hip_library(force=True)
        else:
            t_wait = time.time()
            while _build.is_stale() and time.time() - t_wait < 600:
                time.sleep(2.0)
    import grouped_cumprod as gc
    from simplegaussiansplat_tk71_amd import synthetic

    if args.gpus != world:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one process per GPU; GCP_BENCH_BACKEND=gloo (+ ranks sharing a GPU) exists only to rehearse the N > 1 control
    # flow on a one-GPU box
    backend = os.environ.get("GCP_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > max(1, torch.cuda.device_count()):
        sys.exit(f"--gpus {world} with RCCL needs {world} GPUs, this node shows {torch.cuda.device_count()} "
                 "(GCP_BENCH_BACKEND=gloo lets ranks share a GPU to rehearse the control flow)")
    dev = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where the few collective operands live
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus or dist.get_rank() != rank:
            sys.exit(f"process group has {dist.get_world_size()} ranks (rank {dist.get_rank()}), expected --gpus {args.gpus} (RANK {rank})")
        if rank == 0:
            print(f"[bench] process group up: backend {backend}, world size {dist.get_world_size()}", file=sys.stderr, flush=True)

    # this rank's band of the frame; bands are independent pair lists (no exchange on the scan path)
    unclipped_headline = args.workload.endswith("_unclipped")
    if unclipped_headline:
        args.workload = args.workload[: -len("_unclipped")]
    if args.workload == "cfg5band":
        rows = synthetic.CONFIGS["cfg5"]["height"] // 8
        p = synthetic.make_config("cfg5", seed=rank, device=dev, rows=rows, row_start=rank * rows)
        wl = f"cfg5 band: 3840x{rows} rows of the 3840x2160 / 5M-Gaussian frame, mean 100 splats/pixel (deep)"
    else:
        c = synthetic.CONFIGS[args.workload]
        p = synthetic.make_config(args.workload, seed=rank, device=dev, row_start=rank * c["height"],
                                  **({"max_run": None} if unclipped_headline else {}))
        wl = ("" if not unclipped_headline else "UNCLIPPED (no limit of 4096 splats per pixel) ") + (
            f"{args.workload}: {c['width']}x{c['height']}, {c['gaussians']} Gaussians, mean {c['mean_depth']:g} "
            f"splats/pixel ({'deep heavy-tailed lists' if c['deep'] else 'Poisson'})"
        )
    m = p.n_pairs
    y = torch.empty_like(p.x)
    gin = torch.empty_like(p.x)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]

    def step(e=None):
        if e is not None:
            e[0].record()
        gc.grouped_cumprod_forward(p.x, p.key, y)
        if e is not None:
            e[1].record()
        gc.grouped_cumprod_backward(p.x, y, p.grad_out, p.inv, gin, p.inv_len)
        if e is not None:
            e[2].record()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    # every rank first times its own band ALONE (device synchronisation only, no barrier): the single-GPU rate that
    # `efficiency_vs_single_gpu` is taken against — at N = 1 it is the headline itself
    torch.cuda.synchronize()
    ta = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    alone_rate = m * args.steps / (time.perf_counter() - ta)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(ev[k])
    sync()
    elapsed = time.perf_counter() - t0

    t_fwd = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps * 1e-3  # seconds per launch
    t_bwd = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps * 1e-3
    fallback = gc.last_fallback_tiles(dev)

    stats = torch.tensor([elapsed, float(m)], dtype=torch.float64, device=cdev)
    alone0 = torch.tensor([alone_rate if rank == 0 else 0.0], dtype=torch.float64, device=cdev)
    if world > 1:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        msum = stats[1:].clone()
        dist.all_reduce(msum, op=dist.ReduceOp.SUM)
        dist.all_reduce(alone0, op=dist.ReduceOp.MAX)  # rank 0's figure, on every rank
        elapsed, total_pairs = float(tmax.item()), float(msum.item())
    else:
        total_pairs = float(m)
    rank0_alone = float(alone0.item())

    def headline():
        """The contract's fields: everything the timed region decided."""
        ach = BYTES_BWD * m / t_bwd / 1e9
        return {
            "metric": "splat-pixel pairs blended per second (grouped cumprod fwd+bwd)",
            "value": total_pairs * args.steps / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "per_gpu_pairs_per_s": total_pairs * args.steps / elapsed / world,
            "per_gpu_GBps": (BYTES_FWD + BYTES_BWD) * total_pairs * args.steps / elapsed / 1e9 / world,
            "per_gpu_frac_of_hbm_peak": (BYTES_FWD + BYTES_BWD) * total_pairs * args.steps / elapsed / 1e9 / world / HBM_PEAK_GBPS,
            # aggregate rate / (N x the rate rank 0 reached on its own band before the first barrier); 1.0 at N = 1 up to timer noise
            "efficiency_vs_single_gpu": total_pairs * args.steps / elapsed / (world * rank0_alone),
            "rank0_alone_pairs_per_s": rank0_alone,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": wl,
                "pairs_per_gpu": m,
                "pixel_groups_per_gpu": p.n_groups,
                "sharding": "one image band per GPU, no data-path collective" if world > 1 else "single GPU",
                "bytes_per_pair": BYTES_FWD + BYTES_BWD,
                "aggregate_algorithmic_GBps": (BYTES_FWD + BYTES_BWD) * total_pairs * args.steps / elapsed / 1e9,
                "tiles_left_to_the_follow_up_kernel": fallback,
                "world_size_checked": world,
            },
            "roofline": {
                "kernel": "gcp_scan_main<CUMPROD_BWD> (grouped_cumprod_backward)",
                "bound": "hbm",
                "achieved": ach,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBPS,
                "traffic": pmc_traffic("cumprod_bwd", args.workload),
                "avg_launch_us": t_bwd * 1e6,
                "algorithmic_bytes_per_launch": BYTES_BWD * m,
                "forward": {
                    "kernel": "gcp_scan_main<CUMPROD_FWD> (grouped_cumprod_forward)",
                    "achieved": BYTES_FWD * m / t_fwd / 1e9,
                    "frac": BYTES_FWD * m / t_fwd / 1e9 / HBM_PEAK_GBPS,
                    "avg_launch_us": t_fwd * 1e6,
                    "traffic": pmc_traffic("cumprod_fwd", args.workload),
                },
                "fwd_plus_bwd_frac": (BYTES_FWD + BYTES_BWD) * m / (t_fwd + t_bwd) / 1e9 / HBM_PEAK_GBPS,
            },
        }

    # Everything below is outside the timed region and optional.  With N > 1 it contains collectives; one that never
    # returns (a rank lost, a fabric problem) must not cost the run its headline: after GCP_BENCH_EXTRAS_TIMEOUT seconds
    # (default 420) rank 0 prints the line it has and every rank leaves — with exit code 3, so that a hang shows in the
    # run's return code and not only in a field of the line.
    watchdog = None
    if world > 1:
        import threading

        limit = float(os.environ.get("GCP_BENCH_EXTRAS_TIMEOUT", "420"))

        def give_up():
            if rank == 0:
                out = headline()
                out["sharded_frames"] = {"error": f"the optional blocks did not finish within {limit:g} s; headline only"}
                print(json.dumps(out), flush=True)
                sys.stdout.flush()
            else:
                time.sleep(1.0)  # the launcher ends every rank as soon as one has left: rank 0 prints first
            os._exit(3)

        watchdog = threading.Timer(limit, give_up)
        watchdog.daemon = True
        watchdog.start()

    # ---- outside the timed region ------------------------------------------------------------------------------
    walked = gc.last_lookback_tiles(dev)
    # the pair the reference's live path actually calls: grouped_cumprod_forward, and
    # grouped_cumsum_forward on flipped arrays = one reverse scan here (gs_model.py:716-722); 12 + 12 B/pair
    er = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    gc.grouped_cumsum_reverse(p.grad_out, p.key, gin)
    er[0].record()
    for _ in range(10):
        gc.grouped_cumsum_reverse(p.grad_out, p.key, gin)
    er[1].record()
    torch.cuda.synchronize()
    t_rev = er[0].elapsed_time(er[1]) / 10 * 1e-3
    live_pair = {
        "what": "grouped_cumprod_forward + suffix-sum scan (the reference's flip / grouped_cumsum_forward / flip), 24 B/pair",
        "reverse_scan_us": t_rev * 1e6,
        "pairs_per_s": m / (t_fwd + t_rev),
        "frac_of_hbm_peak": 24 * m / (t_fwd + t_rev) / 1e9 / HBM_PEAK_GBPS,
    }

    def scan_fwd_bwd(q):
        yy = scan_fwd_bwd.buf.setdefault(("y", q.n_pairs), torch.empty_like(q.x))
        gg = scan_fwd_bwd.buf.setdefault(("g", q.n_pairs), torch.empty_like(q.x))
        gc.grouped_cumprod_forward(q.x, q.key, yy)
        gc.grouped_cumprod_backward(q.x, yy, q.grad_out, q.inv, gg, q.inv_len)

    scan_fwd_bwd.buf = {}

    # BASELINE.json config 5 (and config 3 in strong-scaling form): ONE frame cut into `world` slices
    sharded = {}
    if os.environ.get("GCP_BENCH_NO_SHARDED", "0") != "1":
        for name in ("cfg5", "cfg3"):
            if backend != "nccl" and world > 1:
                break  # the gloo rehearsal of this control flow lives in tests/test_sharding.py (CPU tensors)
            free, _ = torch.cuda.mem_get_info(dev)
            c = synthetic.CONFIGS[name]
            est = c["height"] * c["width"] * c["mean_depth"] / world * 4 * 12  # slice arrays + generation temporaries
            enough = torch.tensor([1.0 if free >= est else 0.0], dtype=torch.float64, device=dev)
            if world > 1:  # the decision must be the same on every rank: the block contains collectives
                dist.all_reduce(enough, op=dist.ReduceOp.MIN)
            if float(enough.item()) == 0.0:
                sharded[name] = {"skipped": f"needs ~{est / 2**30:.0f} GiB of HBM per rank, {free / 2**30:.0f} GiB free on rank {rank}"}
                continue
            sharded[name] = sharded_frame(name, world, rank, dev, scan_fwd_bwd, sync, steps=10, warmup=3,
                                          collectives=os.environ.get("GCP_BENCH_NO_COLLECTIVES", "0") != "1")
            scan_fwd_bwd.buf.clear()
            torch.cuda.empty_cache()

    # the same workload without the 4096 clip: a handful of pixels deeper than one tile's raw look-back window
    unclipped = None
    if world == 1 and args.workload in synthetic.CONFIGS:
        q = synthetic.make_config(args.workload, seed=rank, device=dev, max_run=None)
        for _ in range(3):
            scan_fwd_bwd(q)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        yy, gg = scan_fwd_bwd.buf[("y", q.n_pairs)], scan_fwd_bwd.buf[("g", q.n_pairs)]
        reps = 10
        tf = tb = 0.0
        for _ in range(reps):
            e0.record()
            gc.grouped_cumprod_forward(q.x, q.key, yy)
            e1.record()
            gc.grouped_cumprod_backward(q.x, yy, q.grad_out, q.inv, gg, q.inv_len)
            e2.record()
            torch.cuda.synchronize()
            tf += e0.elapsed_time(e1)
            tb += e1.elapsed_time(e2)
        unclipped = {
            "workload": f"{args.workload} without the clip at 4096 splats per pixel",
            "pairs": q.n_pairs,
            "longest_pixel_list": int(q.run_len.max()),
            "pixels_deeper_than_4096": int((q.run_len > 4096).sum()),
            "pairs_per_s": q.n_pairs / ((tf + tb) / reps * 1e-3),
            "forward_us": tf / reps * 1e3,
            "backward_us": tb / reps * 1e3,
            "fwd_plus_bwd_frac_of_hbm_peak": (BYTES_FWD + BYTES_BWD) * q.n_pairs / ((tf + tb) / reps * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "tiles_resolved_through_the_descriptor_tree": gc.last_lookback_tiles(dev),
            "tiles_left_to_the_follow_up_kernel": gc.last_fallback_tiles(dev),
        }
        del q, yy, gg
        scan_fwd_bwd.buf.clear()
        torch.cuda.empty_cache()

    # BASELINE.json configs[1] (1920x1080, 100k Gaussians, mean 8 splats per pixel) beside the headline: the same two
    # launches on its pair list (66 MB per array: cache-resident, which is why the metric is not quoted on it)
    cfg2_block = None
    if world == 1 and args.workload != "cfg2":
        q = synthetic.make_config("cfg2", seed=rank, device=dev)
        for _ in range(5):
            scan_fwd_bwd(q)
        yy, gg = scan_fwd_bwd.buf[("y", q.n_pairs)], scan_fwd_bwd.buf[("g", q.n_pairs)]
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        reps, tf, tb = 50, 0.0, 0.0
        for _ in range(reps):
            e0.record()
            gc.grouped_cumprod_forward(q.x, q.key, yy)
            e1.record()
            gc.grouped_cumprod_backward(q.x, yy, q.grad_out, q.inv, gg, q.inv_len)
            e2.record()
            torch.cuda.synchronize()
            tf += e0.elapsed_time(e1)
            tb += e1.elapsed_time(e2)
        cfg2_block = {
            "workload": "cfg2: 1920x1080, 100000 Gaussians, mean 8 splats/pixel (Poisson) — BASELINE.json configs[1]",
            "pairs": q.n_pairs,
            "pairs_per_s": q.n_pairs / ((tf + tb) / reps * 1e-3),
            "forward_us": tf / reps * 1e3,
            "backward_us": tb / reps * 1e3,
            "fwd_plus_bwd_algorithmic_GBps": (BYTES_FWD + BYTES_BWD) * q.n_pairs / ((tf + tb) / reps * 1e-3) / 1e9,
            "fwd_plus_bwd_frac_of_hbm_peak": (BYTES_FWD + BYTES_BWD) * q.n_pairs / ((tf + tb) / reps * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "note": "arrays of 66 MB each stay in the 256 MB Infinity Cache between launches: not an HBM figure",
        }
        del q, yy, gg
        scan_fwd_bwd.buf.clear()
        torch.cuda.empty_cache()

    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        out = headline()
        out["config"].update({
            "tiles_resolved_through_the_descriptor_tree": walked,
            "unclipped": unclipped,
            "live_path_pair": live_pair,
            "cfg2": cfg2_block,
        })
        out["sharded_frames"] = sharded
        out["roofline"] = out.pop("roofline")  # keep the order of the earlier rounds' lines: roofline after sharded_frames
        if world == 1:
            out["wrapper_level"] = wrapper_level(dev, args.workload)
            out["function_level"] = function_level(dev, args.workload)
            out["caller_level"] = caller_level(dev, args.workload)
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(p, args.cpu_sample)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
