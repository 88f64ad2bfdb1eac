"""Pixel-group sharding of the pair list across the GPUs of one node.

The reference is single-GPU (no communication backend); this is the MI355X-native
addition of SURVEY.md §8(e).  Pixel groups (runs of equal key = one pixel's depth-sorted splats) are
independent, so the scans need NO collective: the pair list is cut at group boundaries and every
rank scans its slice.  Only frame assembly (per-group results -> one image on the loss rank) and
its backward (dL/dI -> owning ranks) move data: ONE gather and ONE scatter of per-group rows.

One process per GPU, `torch.distributed`; backend "nccl" is RCCL on ROCm.  A gather/scatter to one
root over the xGMI full mesh uses the 7 point-to-point links into the root concurrently (≈153 GB/s
each), so it is bound by the per-link size of ONE rank's band, not by a ring: at 4K (99.5 MB frame,
8 ranks) ≈12.4 MB per link ≈ 81 µs.  Works unchanged on CPU tensors with backend "gloo" (tests).
"""
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class Shard:
    rank: int
    group_start: int  # first pixel group owned
    group_end: int    # one past the last
    pair_start: int   # first pair owned (a group boundary)
    pair_end: int

    @property
    def n_groups(self):
        return self.group_end - self.group_start

    @property
    def n_pairs(self):
        return self.pair_end - self.pair_start


def partition_groups(inv_len: torch.Tensor, world_size: int) -> List[Shard]:
    """Cut [0, M) into `world_size` contiguous slices at the group boundaries nearest k*M/R
    (balance by pairs, not by pixels or rows).  `inv_len` = exclusive end offset per group
    (the reference's convention, cuda_test.py:27).  Pure index arithmetic: identical on every rank."""
    g = inv_len.numel()
    ends = inv_len.detach().to("cpu", torch.int64)
    m = int(ends[-1].item()) if g else 0
    cuts = [0]
    for k in range(1, world_size):
        target = (m * k) // world_size
        # group boundary nearest to target: boundaries are 0 and ends[i]
        i = int(torch.searchsorted(ends, torch.tensor(target, dtype=torch.int64), right=False).item())
        i = min(i, g - 1) if g else 0
        hi = int(ends[i].item()) if g else 0
        lo = int(ends[i - 1].item()) if i > 0 else 0
        cut_groups = i + 1 if (hi - target) <= (target - lo) else i
        cuts.append(max(cuts[-1], min(cut_groups, g)))
    cuts.append(g)
    shards = []
    for r in range(world_size):
        g0, g1 = cuts[r], cuts[r + 1]
        p0 = int(ends[g0 - 1].item()) if g0 > 0 else 0
        p1 = int(ends[g1 - 1].item()) if g1 > 0 else 0
        shards.append(Shard(r, g0, g1, p0, p1))
    return shards


def shards_from_counts(group_counts: Sequence[int], pair_counts: Sequence[int]) -> List[Shard]:
    """Shard table for ranks that already own disjoint bands (e.g. one image band generated per rank)."""
    out, g, p = [], 0, 0
    for r, (gc, pc) in enumerate(zip(group_counts, pair_counts)):
        out.append(Shard(r, g, g + int(gc), p, p + int(pc)))
        g += int(gc)
        p += int(pc)
    return out


def local_arrays(shard: Shard, key, x, inv, inv_len, *more):
    """This rank's slice of the flat arrays, with group ids and end offsets rebased to the slice
    (contiguous views: no copy).  Extra per-pair arrays in `more` are sliced alike."""
    p = slice(shard.pair_start, shard.pair_end)
    g = slice(shard.group_start, shard.group_end)
    out = [key[p], x[p], inv[p] - shard.group_start, inv_len[g] - shard.pair_start]
    out += [t[p] for t in more]
    return out


def _padded(t: torch.Tensor, rows: int) -> torch.Tensor:
    if t.size(0) == rows:
        return t.contiguous()
    out = t.new_zeros((rows,) + tuple(t.shape[1:]))
    out[: t.size(0)] = t
    return out


def _via_host(t: torch.Tensor, group) -> bool:
    """gloo moves host memory: device rows go through the host when the N > 1 path is rehearsed without RCCL peers
    (ranks sharing one GPU).  With RCCL ("nccl") the rows stay in HBM and travel over xGMI."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def gather_groups(local: torch.Tensor, shards: Sequence[Shard], dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """ONE gather of per-group rows (e.g. pixel colours [G_r, 3]) to `dst`; returns the
    concatenation [G, ...] there and None elsewhere.  Bands differ in group count, so rows are
    padded to the largest band for the collective and trimmed after."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(shards) == world and local.size(0) == shards[rank].n_groups
    rows = max(s.n_groups for s in shards)
    send = _padded(local, rows)
    device = send.device
    if _via_host(send, group):
        send = send.cpu()
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, bufs, dst=dst, group=group)
        return torch.cat([b[: s.n_groups] for b, s in zip(bufs, shards)], 0).to(device)
    dist.gather(send, None, dst=dst, group=group)
    return None


def scatter_groups(full: Optional[torch.Tensor], shards: Sequence[Shard], like: torch.Tensor, src: int = 0, group=None):
    """ONE scatter of per-group rows from `src` (e.g. dL/dI rows of the loss rank) to their owners.
    `like` gives dtype/device/trailing shape on the receiving ranks."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rows = max(s.n_groups for s in shards)
    host = _via_host(like, group)
    recv = torch.empty((rows,) + tuple(like.shape[1:]), dtype=like.dtype, device="cpu" if host else like.device)
    if rank == src:
        assert full is not None and full.size(0) == shards[-1].group_end
        parts = [_padded(full[s.group_start : s.group_end], rows) for s in shards]
        if host:
            parts = [p.cpu() for p in parts]
        dist.scatter(recv, parts, src=src, group=group)
    else:
        dist.scatter(recv, None, src=src, group=group)
    return recv[: shards[rank].n_groups].to(like.device)


def frame_from_groups(values: torch.Tensor, group_key: torch.Tensor, height: int, width: int) -> torch.Tensor:
    """Per-group rows -> image in the reference's (H+1, W+1, C) layout (gs_model.py:505), pixel
    key = y*10000 + x (gs_model.py:538-541).  Groups are unique pixels, so this is a plain
    scatter: no atomics (the reference accumulates per pair with index_put_(accumulate=True), :510-514)."""
    img = values.new_zeros((height + 1, width + 1) + tuple(values.shape[1:]))
    k = group_key.long()
    img[k // 10000, k % 10000] = values
    return img


def groups_from_frame(frame: torch.Tensor, group_key: torch.Tensor) -> torch.Tensor:
    """Inverse gather: dL/dI at every group's pixel (gs_model.py:703-706)."""
    k = group_key.long()
    return frame[k // 10000, k % 10000]


# ---------------------------------------------------------------------------------------------
# Function-level sharding (row f1): image row bands
# ---------------------------------------------------------------------------------------------
def row_bands(height: int, world_size: int, tile: int = 16) -> List[tuple]:
    """Split image rows 0..height (inclusive, the reference's (H+1)-row image, gs_model.py:505) into
    `world_size` contiguous bands [y0, y1] whose starts are multiples of the 16-pixel tile, so a band's
    tiles are whole tiles of the full frame.  Bands may be empty (y1 < y0) when there are more ranks than
    tile rows."""
    n_tile_rows = (height + 1 + tile - 1) // tile
    bands = []
    for r in range(world_size):
        t0 = (n_tile_rows * r) // world_size
        t1 = (n_tile_rows * (r + 1)) // world_size
        y0, y1 = t0 * tile, min(t1 * tile - 1, height)
        bands.append((y0, y1))
    return bands


def row_bands_by_pairs(startpoint: torch.Tensor, endpoint: torch.Tensor, height: int, world_size: int, tile: int = 16) -> List[tuple]:
    """`row_bands` with the cuts placed by WORK: contiguous bands of whole tile rows holding about the same number of
    splat-pixel pairs each, instead of the same number of rows — a scene whose Gaussians crowd one part of the frame would
    otherwise leave the ranks of the other parts idle (the same imbalance the tile-list walk had between XCDs, DESIGN.md
    §3.4).  Pairs per pixel row come from the boxes alone (a difference array over the rows: O(N + H)); the cut in front of
    rank r is the tile-row boundary nearest to r / world_size of the pairs.  Pure index arithmetic on the boxes: identical on
    every rank.  Bands may be empty, as in `row_bands`."""
    n_tile_rows = (height + 1 + tile - 1) // tile
    s = startpoint.detach().to("cpu", torch.int64)
    e = endpoint.detach().to("cpu", torch.int64)
    y0, y1 = s[:, 1].clamp(min=0), e[:, 1].clamp(max=height)
    wd = (e[:, 0] - s[:, 0] + 1).clamp(min=0)
    ok = (y1 >= y0) & (wd > 0)
    diff = torch.zeros(height + 2, dtype=torch.int64)
    diff.index_add_(0, y0[ok], wd[ok])
    diff.index_add_(0, y1[ok] + 1, -wd[ok])
    per_row = torch.cumsum(diff, 0)[: height + 1]
    pad = n_tile_rows * tile - (height + 1)
    per_tile_row = torch.cat([per_row, torch.zeros(pad, dtype=torch.int64)]).reshape(n_tile_rows, tile).sum(1)
    ends = torch.cumsum(per_tile_row, 0)  # pairs in tile rows [0, t]
    total = int(ends[-1].item()) if n_tile_rows else 0
    if total == 0:
        return row_bands(height, world_size, tile)
    cuts = [0]
    for r in range(1, world_size):
        target = (total * r) // world_size
        t = int(torch.searchsorted(ends, torch.tensor(target, dtype=torch.int64), right=False).item())  # first t with ends[t] >= target
        t = min(t, n_tile_rows - 1)
        before = int(ends[t - 1].item()) if t > 0 else 0
        cut = t + 1 if (int(ends[t].item()) - target) <= (target - before) else t
        cuts.append(max(cuts[-1], min(cut, n_tile_rows)))
    cuts.append(n_tile_rows)
    return [(cuts[r] * tile, min(cuts[r + 1] * tile - 1, height)) for r in range(world_size)]


def band_view(startpoint: torch.Tensor, endpoint: torch.Tensor, mean: torch.Tensor, band: tuple):
    """Inputs of the blend for one band: y coordinates shifted so the band starts at row 0 and boxes cut
    to the band (a box that misses the band becomes empty and is skipped by the kernels; depth order is
    untouched, so per-pixel results are exactly those of the full frame).  Returns
    (startpoint', endpoint', mean', band_height) with band_height = y1 - y0 (the `image_height` argument)."""
    y0, y1 = band
    shift = torch.tensor([0, y0], dtype=startpoint.dtype, device=startpoint.device)
    s = startpoint - shift
    e = endpoint - shift
    s = torch.stack([s[:, 0], s[:, 1].clamp(min=0)], 1)
    e = torch.stack([e[:, 0], e[:, 1].clamp(max=y1 - y0)], 1)
    m = mean - shift.to(mean.dtype)
    return s, e, m, y1 - y0


def band_is_empty(band: tuple) -> bool:
    return band[1] < band[0]


def _hip_blend(s, e, m, variance_inverse, opacity, l_d, width, height, grad_band):
    from . import raster

    bins = raster.bin_tiles(s, e, width, height)
    if grad_band is None:
        return raster.blend_forward(bins, s, e, m, variance_inverse, opacity, l_d), None
    img, ckpt = raster.blend_forward(bins, s, e, m, variance_inverse, opacity, l_d, with_checkpoints=True)
    g_mean, g_vinv, g_op, g_l = raster.blend_backward(bins, s, e, m, variance_inverse, opacity, l_d, ckpt, grad_band)
    return img, (g_vinv, g_op, g_l)


def blend_band(band, startpoint, endpoint, mean, variance_inverse, opacity, l_d, width, grad_band=None, blend=None):
    """One rank's share of the band-sharded Function: the image rows of `band` and, when `grad_band` (dL/dI rows of
    the band) is given, this band's share of the per-Gaussian gradients (variance_inverse, opacity, l_d).
    A rank whose band is EMPTY (more ranks than 16-pixel tile rows) launches nothing and returns a zero-row image and
    zero gradients, so it still joins gather_bands / scatter_bands / allreduce_gaussian_grads with the right shapes
    instead of raising while the others wait in the collective.
    `blend(s, e, m, vinv, opacity, l_d, width, height, grad_band)` defaults to the HIP kernels (raster.py)."""
    n = startpoint.size(0)
    if band_is_empty(band):
        f = variance_inverse
        img = f.new_zeros((0, int(width) + 1, 3))
        if grad_band is None:
            return img, None
        return img, (f.new_zeros((n, 2, 2)), f.new_zeros(tuple(opacity.shape)), f.new_zeros((n, 3)))
    s, e, m, bh = band_view(startpoint, endpoint, mean, band)
    img, grads = (blend or _hip_blend)(s, e, m, variance_inverse, opacity, l_d, int(width), bh, grad_band)
    if grads is not None:
        grads = (grads[0], grads[1].reshape(opacity.shape), grads[2])
    return img, grads


def gather_bands(band_image: torch.Tensor, bands: Sequence[tuple], dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """ONE gather of row bands [(y1-y0+1), W+1, C] into the frame [(H+1), W+1, C] on `dst`."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rows = max(max(b[1] - b[0] + 1, 0) for b in bands)
    send = _padded(band_image, rows)
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, bufs, dst=dst, group=group)
        return torch.cat([bufs[r][: max(bands[r][1] - bands[r][0] + 1, 0)] for r in range(world)], 0)
    dist.gather(send, None, dst=dst, group=group)
    return None


def scatter_bands(frame: Optional[torch.Tensor], bands: Sequence[tuple], like: torch.Tensor, src: int = 0, group=None):
    """ONE scatter of dL/dI row bands from `src` to their owners (`like`: this rank's band image)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rows = max(max(b[1] - b[0] + 1, 0) for b in bands)
    recv = like.new_empty((rows,) + tuple(like.shape[1:]))
    if rank == src:
        parts = [_padded(frame[b[0] : b[1] + 1], rows) for b in bands]
        dist.scatter(recv, parts, src=src, group=group)
    else:
        dist.scatter(recv, None, src=src, group=group)
    y0, y1 = bands[rank]
    return recv[: max(y1 - y0 + 1, 0)]


def allreduce_gaussian_grads(*grads: torch.Tensor, group=None):
    """Per-Gaussian gradients are sums over pixels; a box that straddles bands has a share on several ranks.
    ONE all-reduce of the concatenated N x (2 + 4 + 1 + 3) floats (reduce-scatter + all-gather over xGMI)."""
    flat = torch.cat([g.reshape(g.size(0), -1) for g in grads], 1).contiguous()
    if flat.is_cuda and dist.get_backend(group) == "gloo":  # rehearsals of the N > 1 path on a box without RCCL peers
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    out, c = [], 0
    for g in grads:
        k = g[0].numel() if g.size(0) else 0
        out.append(flat[:, c : c + k].reshape(g.shape))
        c += k
    return out
