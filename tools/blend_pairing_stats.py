"""Could two list entries share one (wave, entry) visit of the blend kernels?  

k_blend_bwd issues ~57 VALU instructions per (wave = 16x4 pixel strip, list entry) visit whether 1 or 64 of the strip's
pixels lie inside the entry's box; 37.5 % of the issued lanes are inside a box at the cfg3 scene.  "Entry-pair packing"
would let two CONSECUTIVE visits of a strip share the per-lane arithmetic when their pixel sets inside the strip are
disjoint (the per-pixel transmittance chain is then untouched).  This script counts, from the real tile lists of a scene,
how often that is possible:

  row-disjoint pair    the two boxes touch different pixel rows of the strip: arithmetic AND the 16-lane row reductions
                       (15 of the 57 instructions) can be shared; cost of the pair ~ 57 + 12 (per-lane selects of the
                       entry's 12 parameters) instead of 114
  pixel-disjoint pair  same rows, different columns: only the arithmetic (42) can be shared, the row sums of the two
                       entries must stay apart: cost ~ 42 + 12 + 2 x 15 = 84 instead of 114

and the largest number of such pairs that can be formed at once (a maximum matching along each strip's visit sequence).

  python tools/blend_pairing_stats.py [cfg2|cfg3]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplegaussiansplat_tk71_amd import raster, synthetic  # noqa: E402


def max_matching_on_paths(edge_ok, same_seq):
    """edge_ok[i]: visits i and i+1 may pair; same_seq[i]: they belong to the same strip sequence.  Maximum number of
    disjoint pairs = sum over maximal runs of L consecutive usable edges of ceil(L / 2)."""
    ok = (edge_ok & same_seq).to(torch.int64)
    if ok.numel() == 0:
        return 0
    # run lengths of consecutive ones
    pad = torch.cat([ok.new_zeros(1), ok, ok.new_zeros(1)])
    d = pad[1:] - pad[:-1]
    starts = torch.nonzero(d == 1).flatten()
    ends = torch.nonzero(d == -1).flatten()
    lens = ends - starts
    return int(((lens + 1) // 2).sum())


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    dev = torch.device("cuda", 0)
    sc = synthetic.make_scene_config(cfg, seed=0, device=dev)
    w, h = sc["width"], sc["height"]
    bins = raster.bin_tiles(sc["start"], sc["end"], w, h)
    K = bins.n_tile_pairs
    tile = torch.searchsorted(bins.tile_start[1:].contiguous(), torch.arange(K, device=dev, dtype=torch.int32), right=True)
    g = bins.tile_list.long()
    tx0 = (tile % bins.tiles_x) * 16
    ty0 = (tile // bins.tiles_x) * 16
    x0 = (sc["start"][g, 0].clamp(min=0).long() - tx0).clamp(min=0)
    x1 = (sc["end"][g, 0].clamp(max=w).long() - tx0).clamp(max=15)
    y0 = (sc["start"][g, 1].clamp(min=0).long() - ty0).clamp(min=0)
    y1 = (sc["end"][g, 1].clamp(max=h).long() - ty0).clamp(max=15)
    cm = ((2 << x1) - (1 << x0)) * (x1 >= x0)
    rm = ((2 << y1) - (1 << y0)) * (y1 >= y0)
    out = {"workload": cfg, "tile_entries": K, "visits": 0, "row_disjoint_adjacent": 0, "pixel_disjoint_adjacent": 0,
           "max_pairs_row_disjoint": 0, "max_pairs_pixel_disjoint": 0}
    lanes_in = 0
    pop4 = torch.tensor([bin(i).count("1") for i in range(16)], device=dev)
    pop16 = torch.tensor([bin(i).count("1") for i in range(1 << 16)], device=dev)
    for strip in range(4):
        rb = (rm >> (4 * strip)) & 0xF
        vis = torch.nonzero(rb != 0).flatten()
        out["visits"] += int(vis.numel())
        rows = rb[vis]
        cols = cm[vis]
        lanes_in += int((pop4[rows] * pop16[cols]).sum())
        same = tile[vis][1:] == tile[vis][:-1]
        row_dis = (rows[1:] & rows[:-1]) == 0
        pix_dis = row_dis | ((cols[1:] & cols[:-1]) == 0)
        out["row_disjoint_adjacent"] += int((row_dis & same).sum())
        out["pixel_disjoint_adjacent"] += int((pix_dis & same).sum())
        out["max_pairs_row_disjoint"] += max_matching_on_paths(row_dis, same)
        out["max_pairs_pixel_disjoint"] += max_matching_on_paths(pix_dis, same)
    v = out["visits"]
    out["active_lane_frac"] = lanes_in / (64.0 * v)
    pr, pp = out["max_pairs_row_disjoint"], out["max_pairs_pixel_disjoint"]
    # VALU wave-instructions per visit today, and with every possible pair formed (row-disjoint pairs first: they save more)
    base = 57.0 * v
    only_pix = max(pp - pr, 0)
    packed = base - pr * (114 - 69) - only_pix * (114 - 84)
    out["valu_per_visit_now"] = 57.0
    out["valu_saving_frac_upper_bound"] = 1.0 - packed / base
    print(json.dumps(out))


if __name__ == "__main__":
    main()
