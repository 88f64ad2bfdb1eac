"""Tile binning and fused alpha blending on the HIP library.

Host side of csrc/gcp_raster.hip: allocates buffers with torch, passes raw pointers through the C ABI
(include/grouped_cumprod_hip.h).  Everything the reference does around its scan in
`custom_autograd_grouped_cumprod` (reference: gs_model.py:598-663) happens in three launches here:
bin (f2), blend forward (f1), blend backward (f1).  No CPU path.
"""
import ctypes
from dataclasses import dataclass

import torch

from . import _lib

TILE = 16


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _dev_tensor(t, name, dtype, shape_tail=None):
    _require(isinstance(t, torch.Tensor), f"{name}: expected a torch.Tensor")
    _require(t.is_cuda, f"{name}: expected a ROCm device tensor, got {t.device} (no CPU path)")
    if t.requires_grad:
        t = t.detach()
    if t.dtype != dtype:
        t = t.to(dtype)
    if not t.is_contiguous():
        t = t.contiguous()
    if shape_tail is not None:
        _require(tuple(t.shape[1:]) == tuple(shape_tail), f"{name}: shape {tuple(t.shape)}, expected [N,{shape_tail}]")
    return t


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


class _on:
    """`with _on(device):` — torch.cuda.device(device) only where it is not the current one already (the calls of rows a5 / a6
    run behind a read-back with the GPU waiting for the next launch: the host work between two launches is on the clock)."""
    __slots__ = ("ctx",)

    def __init__(self, device):
        self.ctx = None if device.index == torch.cuda.current_device() else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


@dataclass
class TileBins:
    """Per-tile depth-ordered Gaussian lists of one image (the f2 product)."""
    width: int
    height: int
    n_gauss: int
    n_tile_pairs: int
    tiles_x: int
    tiles_y: int
    tile_off: torch.Tensor    # int32[N+1]  first Gaussian-major entry of every Gaussian
    tile_start: torch.Tensor  # int32[n_tiles+1] first sorted entry of every tile
    tile_list: torch.Tensor   # int32[K] Gaussian ids, tile-major, depth order inside a tile
    info: torch.Tensor = None  # capture-safe mode only: int32[2] on the device = (entries listed, capacity exceeded)

    @property
    def tile_capacity(self):
        """Upper bound of the (tile, Gaussian) entry count that sizes every buffer derived from the bins (== the exact
        count when the bins were built with the host read of K; the caller's bound in the capture-safe mode)."""
        return self.n_tile_pairs

    def overflowed(self):
        """Capture-safe mode: True if the capacity was too small (one device->host read; call it once per step, after
        the work has been queued).  Always False for bins built with the exact count."""
        return bool(self.info[1].item()) if self.info is not None else False


def bin_tiles(startpoint, endpoint, width, height, capacity=None, counted=None):
    """startpoint/endpoint: int [N,2] (x,y) inclusive boxes in depth order.

    capacity=None: two calls with one device->host read of the entry count K in between (exactly sized buffers).
    capacity=K_max: ONE call, no read-back, graph-capturable; `bins.n_tile_pairs` is then the capacity, the true count
    stays on the device (`bins.info`) and `bins.overflowed()` tells after the fact whether the bound was too small.
    counted=(tile_off int32[N+1], K): the counting pass has been done already (`rects_to_boxes` does it while it cuts the
    list): the second call alone, nothing read back."""
    start = _dev_tensor(startpoint, "startpoint", torch.int32, (2,))
    end = _dev_tensor(endpoint, "endpoint", torch.int32, (2,))
    n = start.size(0)
    _require(end.size(0) == n, "endpoint: row count differs from startpoint")
    dev = start.device
    width, height = int(width), int(height)
    lib = _lib.load()
    tx, ty = ctypes.c_int32(0), ctypes.c_int32(0)
    _lib.check(lib.gcp_tile_grid(width, height, ctypes.byref(tx), ctypes.byref(ty)), "gcp_tile_grid")
    n_tiles = tx.value * ty.value
    if counted is not None:
        _require(capacity is None, "bin_tiles: `counted` and `capacity` exclude each other")
        tile_off, K = counted
        tile_off = _dev_tensor(tile_off, "tile_off", torch.int32)
        K = int(K)
        _require(tile_off.numel() >= n + 1 and K >= 0, "counted: expected (int32[N + 1] prefix sums, their total)")
        with torch.cuda.device(dev):
            tile_start = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
            tile_list = torch.empty(max(K, 1), dtype=torch.int32, device=dev)
            ws = torch.empty(lib.gcp_bin_workspace_bytes(n, K), dtype=torch.uint8, device=dev)
            _lib.check(
                lib.gcp_bin_tiles_fill(start.data_ptr(), end.data_ptr(), n, width, height, tile_off.data_ptr(), K,
                                       tile_start.data_ptr(), tile_list.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)),
                "gcp_bin_tiles_fill",
            )
        return TileBins(width, height, n, K, tx.value, ty.value, tile_off, tile_start, tile_list[:K])
    if capacity is not None:
        cap = int(capacity)
        _require(cap >= 1, "capacity: must be >= 1")
        with torch.cuda.device(dev):
            tile_off = torch.empty(n + 1, dtype=torch.int32, device=dev)
            tile_start = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
            tile_list = torch.empty(cap, dtype=torch.int32, device=dev)
            info = torch.empty(2, dtype=torch.int32, device=dev)
            ws = torch.empty(lib.gcp_bin_workspace_bytes(n, cap), dtype=torch.uint8, device=dev)
            _lib.check(
                lib.gcp_bin_tiles(start.data_ptr(), end.data_ptr(), n, width, height, cap, tile_off.data_ptr(),
                                  tile_start.data_ptr(), tile_list.data_ptr(), info.data_ptr(), ws.data_ptr(), ws.numel(),
                                  _stream(dev)),
                "gcp_bin_tiles",
            )
        return TileBins(width, height, n, cap, tx.value, ty.value, tile_off, tile_start, tile_list, info)
    with torch.cuda.device(dev):
        st = _stream(dev)
        tile_off = torch.empty(n + 1, dtype=torch.int32, device=dev)
        ws = torch.empty(lib.gcp_bin_workspace_bytes(n, 0), dtype=torch.uint8, device=dev)
        k = ctypes.c_int64(0)
        _lib.check(
            lib.gcp_bin_tiles_count(start.data_ptr(), end.data_ptr(), n, width, height, tile_off.data_ptr(),
                                    ctypes.byref(k), ws.data_ptr(), ws.numel(), st),
            "gcp_bin_tiles_count",
        )
        K = k.value
        tile_start = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
        tile_list = torch.empty(max(K, 1), dtype=torch.int32, device=dev)
        ws = torch.empty(lib.gcp_bin_workspace_bytes(n, K), dtype=torch.uint8, device=dev)
        _lib.check(
            lib.gcp_bin_tiles_fill(start.data_ptr(), end.data_ptr(), n, width, height, tile_off.data_ptr(), K,
                                   tile_start.data_ptr(), tile_list.data_ptr(), ws.data_ptr(), ws.numel(), st),
            "gcp_bin_tiles_fill",
        )
    return TileBins(width, height, n, K, tx.value, ty.value, tile_off, tile_start, tile_list[:K])


def _params(startpoint, endpoint, mean, variance_inverse, opacity, l_d):
    start = _dev_tensor(startpoint, "startpoint", torch.int32, (2,))
    end = _dev_tensor(endpoint, "endpoint", torch.int32, (2,))
    mean_f = _dev_tensor(mean, "mean", torch.float32, (2,))  # reference: mean.to(torch.float32), gs_model.py:679
    vinv = _dev_tensor(variance_inverse, "variance_inverse", torch.float32, (2, 2))
    n = start.size(0)
    op = _dev_tensor(opacity, "opacity", torch.float32).reshape(-1)
    col = _dev_tensor(l_d, "l_d", torch.float32, (3,))
    for t, name in ((end, "endpoint"), (mean_f, "mean"), (vinv, "variance_inverse"), (op, "opacity"), (col, "l_d")):
        _require(t.size(0) == n, f"{name}: {t.size(0)} rows, expected {n}")
        _require(t.device == start.device, f"{name}: on {t.device}, expected {start.device}")
    return start, end, mean_f, vinv, op, col


def blend_forward(bins, startpoint, endpoint, mean, variance_inverse, opacity, l_d, with_checkpoints=False):
    """-> image f32[(H+1),(W+1),3] (reference layout, gs_model.py:505); with_checkpoints=True returns
    (image, t_ckpt), t_ckpt being the per-pixel transmittance checkpoints `blend_backward` restarts from."""
    start, end, mean_f, vinv, op, col = _params(startpoint, endpoint, mean, variance_inverse, opacity, l_d)
    dev = start.device
    lib = _lib.load()
    image = torch.empty(bins.height + 1, bins.width + 1, 3, dtype=torch.float32, device=dev)
    ckpt = None
    if with_checkpoints:
        ckpt = torch.empty(lib.gcp_blend_checkpoint_floats(bins.tile_capacity, bins.width, bins.height), dtype=torch.float32,
                           device=dev)
    with torch.cuda.device(dev):
        _lib.check(
            lib.gcp_blend_forward(start.data_ptr(), end.data_ptr(), mean_f.data_ptr(), vinv.data_ptr(), op.data_ptr(),
                                  col.data_ptr(), bins.n_gauss, bins.width, bins.height, bins.tile_start.data_ptr(),
                                  bins.tile_list.data_ptr(), image.data_ptr(), ckpt.data_ptr() if with_checkpoints else None,
                                  _stream(dev)),
            "gcp_blend_forward",
        )
    return (image, ckpt) if with_checkpoints else image


def blend_backward(bins, startpoint, endpoint, mean, variance_inverse, opacity, l_d, t_ckpt, grad_image):
    """-> (grad_mean [N,2], grad_variance_inverse [N,2,2], grad_opacity [N,1], grad_l_d [N,3]).
    `t_ckpt`: the checkpoints `blend_forward(..., with_checkpoints=True)` returned for the same bins and inputs."""
    start, end, mean_f, vinv, op, col = _params(startpoint, endpoint, mean, variance_inverse, opacity, l_d)
    dev = start.device
    n = bins.n_gauss
    gimg = _dev_tensor(grad_image, "grad_image", torch.float32)
    shape = (bins.height + 1, bins.width + 1, 3)
    _require(tuple(gimg.shape) == shape, f"grad_image: expected shape {shape}")
    lib = _lib.load()
    ck = _dev_tensor(t_ckpt, "t_ckpt", torch.float32)
    _require(ck.numel() >= lib.gcp_blend_checkpoint_floats(bins.tile_capacity, bins.width, bins.height),
             "t_ckpt: too small for these bins (pass what blend_forward(..., with_checkpoints=True) returned)")
    g_mean = torch.empty(n, 2, dtype=torch.float32, device=dev)
    g_vinv = torch.empty(n, 2, 2, dtype=torch.float32, device=dev)
    g_op = torch.empty(n, 1, dtype=torch.float32, device=dev)
    g_l = torch.empty(n, 3, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        ws = torch.empty(lib.gcp_blend_backward_workspace_bytes(bins.tile_capacity), dtype=torch.uint8, device=dev)
        _lib.check(
            lib.gcp_blend_backward(start.data_ptr(), end.data_ptr(), mean_f.data_ptr(), vinv.data_ptr(), op.data_ptr(),
                                   col.data_ptr(), n, bins.width, bins.height, bins.tile_off.data_ptr(),
                                   bins.tile_capacity, bins.tile_start.data_ptr(), bins.tile_list.data_ptr(),
                                   ck.data_ptr(), gimg.data_ptr(), g_mean.data_ptr(), g_vinv.data_ptr(),
                                   g_op.data_ptr(), g_l.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)),
            "gcp_blend_backward",
        )
    return g_mean, g_vinv, g_op, g_l


def exclusive_scan_i32(x):
    """int32[n] -> int32[n+1] exclusive prefix sums (last entry = total)."""
    x = _dev_tensor(x, "x", torch.int32)
    dev = x.device
    lib = _lib.load()
    out = torch.empty(x.numel() + 1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        ws = torch.empty(lib.gcp_scan_i32_workspace_bytes(x.numel()), dtype=torch.uint8, device=dev)
        _lib.check(lib.gcp_exclusive_scan_i32(x.data_ptr(), out.data_ptr(), x.numel(), ws.data_ptr(), ws.numel(),
                                              _stream(dev)), "gcp_exclusive_scan_i32")
    return out


@dataclass
class PixelLists:
    """Per-pixel CSR of the splat-pixel pairs: what the reference obtains from `unique` + a stable
    sort of M pixel keys (gs_model.py:546-548), built from the tile lists instead."""
    pixel_off: torch.Tensor   # int32[P+1], P = (H+1)*(W+1) pixels row-major (= ascending key y*10000+x)
    pair_gauss: torch.Tensor  # int32[M] Gaussian of every pair, pixel-major, depth order inside a pixel
    pair_index: torch.Tensor  # int32[M] == `index` of torch.sort(key, stable=True) (gs_model.py:547)
    pair_key: torch.Tensor    # int32[M] == `sorted_inv` of the same sort: y*10000+x of every pair
    box_off: torch.Tensor     # int32[N+1] first Gaussian-major pair of every Gaussian


def pixel_lists(bins, startpoint, endpoint):
    start = _dev_tensor(startpoint, "startpoint", torch.int32, (2,))
    end = _dev_tensor(endpoint, "endpoint", torch.int32, (2,))
    dev = start.device
    lib = _lib.load()
    n_pix = (bins.height + 1) * (bins.width + 1)
    with torch.cuda.device(dev):
        st = _stream(dev)
        count = torch.empty(n_pix, dtype=torch.int32, device=dev)
        bsize = torch.empty(max(bins.n_gauss, 1), dtype=torch.int32, device=dev)
        _lib.check(
            lib.gcp_pixel_lists_count(start.data_ptr(), end.data_ptr(), bins.n_gauss, bins.width, bins.height,
                                      bins.tile_start.data_ptr(), bins.tile_list.data_ptr(), count.data_ptr(),
                                      bsize.data_ptr(), st),
            "gcp_pixel_lists_count",
        )
        total = int(bsize[: bins.n_gauss].sum(dtype=torch.int64).item())
        _require(total < 2**31, f"pixel lists: {total} pairs do not fit the int32 offsets of the pair arrays")
        pixel_off = exclusive_scan_i32(count)
        box_off = exclusive_scan_i32(bsize[: bins.n_gauss])
        m = int(pixel_off[-1].item())
        _require(m == int(box_off[-1].item()), "pixel lists: pair count mismatch between pixel- and box-major views")
        pair_gauss = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
        pair_index = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
        pair_key = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
        _lib.check(
            lib.gcp_pixel_lists_fill(start.data_ptr(), end.data_ptr(), bins.n_gauss, bins.width, bins.height,
                                     bins.tile_start.data_ptr(), bins.tile_list.data_ptr(), pixel_off.data_ptr(),
                                     box_off.data_ptr(), pair_gauss.data_ptr(), pair_index.data_ptr(),
                                     pair_key.data_ptr(), st),
            "gcp_pixel_lists_fill",
        )
    return PixelLists(pixel_off, pair_gauss[:m], pair_index[:m], pair_key[:m], box_off)


def expand_rects(startpoint, endpoint, width, height, with_gaussian=False):
    """The reference's rect list: every pixel of every box, Gaussian-major, row-major inside a box
    (Utilities.make_rect_points_parallel, uitility.py:336-366 -> _create_rects, gs_model.py:480-482).
    Returns rects int32[M,2] (x,y) [, pair_gauss int32[M] = gause_points_inv, gs_model.py:768-773]."""
    start = _dev_tensor(startpoint, "startpoint", torch.int32, (2,))
    end = _dev_tensor(endpoint, "endpoint", torch.int32, (2,))
    n = start.size(0)
    dev = start.device
    lib = _lib.load()
    width, height = int(width), int(height)
    with torch.cuda.device(dev):
        st = _stream(dev)
        bsize = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        _lib.check(lib.gcp_box_sizes(start.data_ptr(), end.data_ptr(), n, width, height, bsize.data_ptr(), st), "gcp_box_sizes")
        box_off = exclusive_scan_i32(bsize[:n])
        m = int(box_off[-1].item())  # the reference synchronises here too (uitility.py:348)
        rects = torch.empty(m, 2, dtype=torch.int32, device=dev)
        owner = torch.empty(m, dtype=torch.int32, device=dev) if with_gaussian else None
        _lib.check(
            lib.gcp_expand_rects(start.data_ptr(), end.data_ptr(), box_off.data_ptr(), n, m, width, height, rects.data_ptr(),
                                 owner.data_ptr() if with_gaussian else None, st),
            "gcp_expand_rects",
        )
    return (rects, owner) if with_gaussian else rects


def box_offsets(startpoint, endpoint, width, height):
    """int32[N+1]: first Gaussian-major pair of every Gaussian in the reference's rect list (uitility.py:336-366), i.e. the
    exclusive prefix sums of the box sizes clamped to the image; [-1] = M."""
    start = _dev_tensor(startpoint, "startpoint", torch.int32, (2,))
    end = _dev_tensor(endpoint, "endpoint", torch.int32, (2,))
    n = start.size(0)
    dev = start.device
    with torch.cuda.device(dev):
        bsize = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        _lib.check(_lib.load().gcp_box_sizes(start.data_ptr(), end.data_ptr(), n, int(width), int(height), bsize.data_ptr(), _stream(dev)),
                   "gcp_box_sizes")
        return exclusive_scan_i32(bsize[:n])


def scan_boxes(bins, startpoint, endpoint, box_off, values, mode, count_dropped=False):
    """Inclusive per-pixel scan of `values` (f32[M], Gaussian-major rect order) in depth order, result in the same order:
    mode 0 product, 1 sum, 2 suffix sum — the sort / gather / scan / un-sort of _create_alpha_brend (gs_model.py:546-555)
    as one walk of the tile lists (gcp_pairs_scan_boxes).  Rows of `values` whose pair lies outside every listed box do
    not exist by construction (box_off comes from the same boxes).  count_dropped: also return int32[ceil(M / 4096)], how
    many results are exactly 0 in every 4096 consecutive pairs — hand it to `compact_finish(..., dropped=)`, which then
    skips its counting launch."""
    start = _dev_tensor(startpoint, "startpoint", torch.int32, (2,))
    end = _dev_tensor(endpoint, "endpoint", torch.int32, (2,))
    x = _dev_tensor(values, "values", torch.float32)
    off = _dev_tensor(box_off, "box_off", torch.int32)
    _require(x.dim() == 1, "values: expected a 1-D tensor")
    _require(off.numel() == bins.n_gauss + 1, "box_off: expected n_gauss + 1 offsets")
    out = torch.empty_like(x)
    dropped = torch.empty((x.numel() + 4095) // 4096, dtype=torch.int32, device=x.device) if count_dropped else None
    if x.numel() == 0 or bins.n_tile_pairs == 0:  # boxes that expand to nothing (all outside the image): nothing to scan
        _require(x.numel() == 0, "values: the boxes expand to no pair at all")
        return (out, dropped) if count_dropped else out
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().gcp_pairs_scan_boxes(start.data_ptr(), end.data_ptr(), bins.n_gauss, bins.width, bins.height,
                                                    bins.tile_start.data_ptr(), bins.tile_list.data_ptr(), off.data_ptr(), x.data_ptr(),
                                                    out.data_ptr(), x.numel(), int(mode), dropped.data_ptr() if count_dropped else None,
                                                    _stream(x.device)), "gcp_pairs_scan_boxes")
    return (out, dropped) if count_dropped else out


def finish_buffers(values):
    """The outputs of `finish_boxes` for `values` (f32[M]), allocated and — the mask set to ones, the zero counts cleared —
    initialised NOW: nothing of it depends on the boxes, so a caller that still has to wait for the cut's read-back queues the
    two fills first and they run while the host waits.  -> (final, keep, dropped) to pass as `finish_boxes(..., buffers=)`."""
    x = _dev_tensor(values, "values", torch.float32)
    _require(x.dim() == 1, "values: expected a 1-D tensor")
    m = x.numel()
    out = torch.empty_like(x)
    keep = torch.empty(m, dtype=torch.uint8, device=x.device)
    dropped = torch.empty((m + 4095) // 4096, dtype=torch.int32, device=x.device)
    if m:
        with _on(x.device):
            _lib.check(_lib.load().gcp_pairs_finish_prepare(keep.data_ptr(), dropped.data_ptr(), m, _stream(x.device)), "gcp_pairs_finish_prepare")
    return out, keep, dropped


def finish_boxes(bins, startpoint, endpoint, box_off, values, mode, buffers=None):
    """The walk of `scan_boxes` writing the FINAL values of _create_alpha_brend (gs_model.py:557-564) instead of the inclusive
    ones: -> (final f32[M] = inclusive / self (mode 0) or inclusive - self (modes 1, 2), keep uint8[M] = inclusive != 0,
    dropped int32[ceil(M / 4096)]).  Hand all three to `compact_kept`: when nothing was dropped they ARE the result.
    buffers: what `finish_buffers(values)` returned on this stream (already initialised)."""
    start = _dev_tensor(startpoint, "startpoint", torch.int32, (2,))
    end = _dev_tensor(endpoint, "endpoint", torch.int32, (2,))
    x = _dev_tensor(values, "values", torch.float32)
    off = _dev_tensor(box_off, "box_off", torch.int32)
    _require(x.dim() == 1, "values: expected a 1-D tensor")
    _require(off.numel() >= bins.n_gauss + 1, "box_off: expected n_gauss + 1 offsets")
    m = x.numel()
    prepared = buffers is not None
    out, keep, dropped = buffers if prepared else (torch.empty_like(x), torch.empty(m, dtype=torch.uint8, device=x.device),
                                                   torch.empty((m + 4095) // 4096, dtype=torch.int32, device=x.device))
    _require(out.numel() == m and keep.numel() == m and dropped.numel() == (m + 4095) // 4096, "buffers: not those of finish_buffers(values)")
    if m == 0:
        return out, keep, dropped
    _require(bins.n_tile_pairs > 0, "values: the boxes expand to no pair at all")
    with _on(x.device):
        _lib.check(_lib.load().gcp_pairs_finish_boxes(start.data_ptr(), end.data_ptr(), bins.n_gauss, bins.width, bins.height,
                                                      bins.tile_start.data_ptr(), bins.tile_list.data_ptr(), off.data_ptr(), x.data_ptr(),
                                                      out.data_ptr(), keep.data_ptr(), m, int(mode), dropped.data_ptr(), 1 if prepared else 0,
                                                      _stream(x.device)), "gcp_pairs_finish_boxes")
    return out, keep, dropped


def mark_all_kept(mask):
    """A `!= 0` mask that is known (from the kept count the call read back anyway) to be all ones: `cuda_kernel.mask_tensor` —
    the reference's `_mask_tensor`, gs_model.py:525-531 — then hands its tensors through instead of indexing each of them."""
    mask._gcp_all_kept = True
    return mask


def all_kept(mask):
    return bool(getattr(mask, "_gcp_all_kept", False))


def compact_kept(final, keep, dropped=None, begin=0, end=None):
    """[values, mask] of _create_alpha_brend from what `finish_boxes` wrote, rows [begin, end) (the `cutting_number` slice,
    gs_model.py:557-559): ONE device->host read — the kept count, which sizes the result as the reference's `output[mask]`
    does — and, only if something was dropped, one pass that moves the kept values together.  When nothing was dropped the
    returned tensors are views of `final` and `keep`."""
    fin = _dev_tensor(final, "final", torch.float32)
    kp = _dev_tensor(keep, "keep", torch.uint8)
    _require(fin.dim() == 1 and kp.shape == fin.shape, "final / keep: expected two 1-D tensors of one length")
    m = fin.numel()
    end = m if end is None else int(end)
    begin = int(begin)
    _require(0 <= begin <= end <= m, "compact_kept: bad row range")
    n = end - begin
    dev = fin.device
    mask = kp[begin:end].view(torch.bool)
    if n == 0:
        return fin[begin:end], mask
    lib = _lib.load()
    if dropped is not None:
        dropped = _dev_tensor(dropped, "dropped", torch.int32)
        _require(dropped.numel() == (m + 4095) // 4096, "dropped: expected one count per 4096 rows")
    with _on(dev):
        st = _stream(dev)
        count = torch.empty(1, dtype=torch.int32, device=dev)
        ws = torch.empty(lib.gcp_compact_kept_workspace_bytes(n), dtype=torch.uint8, device=dev)
        _lib.check(lib.gcp_compact_kept_count(kp.data_ptr(), dropped.data_ptr() if dropped is not None else None, m, begin, end,
                                              count.data_ptr(), ws.data_ptr(), ws.numel(), st), "gcp_compact_kept_count")
        kept = int(count.item())
        if kept == n:
            return fin[begin:end], mark_all_kept(mask)
        values = torch.empty(kept, dtype=torch.float32, device=dev)
        if kept:
            _lib.check(lib.gcp_compact_kept_write(fin.data_ptr(), kp.data_ptr(), begin, end, values.data_ptr(), ws.data_ptr(), ws.numel(), st),
                       "gcp_compact_kept_write")
    return values, mask


@dataclass
class RectBoxes:
    """A rect list cut back into rectangles (`rects_to_boxes`): the boxes, in list order, that expand to it."""
    start: torch.Tensor    # int32[R,2] (x, y) inclusive
    end: torch.Tensor      # int32[R,2]
    box_off: torch.Tensor  # int32[R+1]: first pair of every rectangle, [-1] = M
    width: int             # largest x in the list
    height: int            # largest y
    tile_off: torch.Tensor = None  # int32[R+1]: the binning's counting pass, done with the cut (None: not done)
    n_tile_pairs: int = None       # its total K

    def bin(self):
        """The rectangles binned into 16x16 tiles (`bin_tiles`), without a second counting pass where the cut has done it.
        (This runs right behind the cut's read-back, with the GPU idle until its first launch: the arrays are the cut's own —
        int32, contiguous, on one device — so the checks of `bin_tiles` are skipped.)"""
        if self.tile_off is None:
            return bin_tiles(self.start, self.end, self.width, self.height)
        lib = _lib.load()
        n, K, dev = self.start.size(0), int(self.n_tile_pairs), self.start.device
        if dev.index != torch.cuda.current_device():
            return bin_tiles(self.start, self.end, self.width, self.height, counted=(self.tile_off, K))
        ctx, cty = ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.check(lib.gcp_tile_grid(self.width, self.height, ctypes.byref(ctx), ctypes.byref(cty)), "gcp_tile_grid")
        tx, ty = ctx.value, cty.value
        tile_start = torch.empty(tx * ty + 1, dtype=torch.int32, device=dev)
        tile_list = torch.empty(max(K, 1), dtype=torch.int32, device=dev)
        ws = torch.empty(lib.gcp_bin_workspace_bytes(n, K), dtype=torch.uint8, device=dev)
        _lib.check(
            lib.gcp_bin_tiles_fill(self.start.data_ptr(), self.end.data_ptr(), n, self.width, self.height, self.tile_off.data_ptr(), K,
                                   tile_start.data_ptr(), tile_list.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)),
            "gcp_bin_tiles_fill",
        )
        return TileBins(self.width, self.height, n, K, tx, ty, self.tile_off, tile_start, tile_list[:K])


CUT_SLOT_ROWS = 512   # row records a 4096-pair tile may park in the one-call cut: boxes of 8 columns and more
CUT_MIN_RECT = 64     # pairs per rectangle its arrays are sized for
CUT_SLOT_ROWS_ANY = 2048  # the second attempt: room for every list that is made of boxes at all (rows of 2 pixels on average,
                          # rectangles of `min_mean_size` pairs: beyond that the list is sorted anyway)
_cut_sizing = {}  # device index -> (slot_rows, min_rect): what the last list cut there needed, with a margin


def _pow2_at_least(x):
    return 1 << max(0, int(x) - 1).bit_length()


def _cut_rects_once(r, i64, carry_front, carry_back, min_mean_size, slot_rows=CUT_SLOT_ROWS, min_rect=CUT_MIN_RECT):
    """gcp_rects_cut: rows, rectangles, boxes and the binning's counts with ONE device->host read.  Returns a RectBoxes, None
    (not a list the walk can take: sort) or "retry" (more rows per tile or smaller rectangles than this attempt made room
    for).  slot_rows / min_rect: the room made — row records per 4096-pair tile, pairs per rectangle."""
    lib = _lib.load()
    m = r.size(0)
    dev = r.device
    cap = (m - carry_front - carry_back) // max(1, int(min_rect)) + carry_front + carry_back + 16
    with _on(dev):
        st = _stream(dev)
        start = torch.empty(cap, 2, dtype=torch.int32, device=dev)
        end = torch.empty(cap, 2, dtype=torch.int32, device=dev)
        box_off = torch.empty(cap + 1, dtype=torch.int32, device=dev)
        tile_off = torch.empty(cap + 1, dtype=torch.int32, device=dev)
        info = torch.empty(8, dtype=torch.int32, device=dev)
        ws = torch.empty(lib.gcp_rects_cut_workspace_bytes(m, carry_front, carry_back, slot_rows, cap), dtype=torch.uint8, device=dev)
        _lib.check(lib.gcp_rects_cut(r.data_ptr(), 1 if i64 else 0, m, carry_front, carry_back, slot_rows, cap, start.data_ptr(),
                                     end.data_ptr(), box_off.data_ptr(), tile_off.data_ptr(), info.data_ptr(), ws.data_ptr(), ws.numel(), st),
                   "gcp_rects_cut")
        n_rows, max_x, max_y, mn, flags, n_rects, k, _ = info.tolist()
        del ws
    _require(mn >= 0, "rects: coordinates must lie in [0, 2^31) (negative ones are not supported)")
    if flags & 1:
        return None
    if flags:
        return "retry"
    if n_rects * min_mean_size > m:
        return None
    # what this list needed, with a margin, for the next list cut on this device (a training loop's lists resemble each other)
    body = max(1, m - carry_front - carry_back)
    tiles = (m + 4095) // 4096
    _cut_sizing[dev.index] = (min(CUT_SLOT_ROWS_ANY, max(CUT_SLOT_ROWS, _pow2_at_least(5 * n_rows // (4 * max(1, tiles))))),
                              max(int(min_mean_size), min(CUT_MIN_RECT, _pow2_at_least(body // max(1, n_rects)) // 4)))
    return RectBoxes(start[:n_rects], end[:n_rects], box_off[: n_rects + 1], int(max_x), int(max_y), tile_off[: n_rects + 1], int(k))


def rects_to_boxes(rects, min_mean_size=8, carry_rows=0, carry_at_end=False, one_call=True):
    """Cut the reference's rect list (int [M,2] (x, y), gs_model.py:480-482) back into the row-major rectangles it is a
    concatenation of (csrc/gcp_pairs.hip).  Works on any list; returns None when the list does not look like boxes at all
    (more than one row per two pairs, or fewer than `min_mean_size` pairs per rectangle on average: the caller sorts
    instead).  carry_rows: how many single-pixel carry rows the list starts — or, carry_at_end, ends — with (a chunked call's
    `cutting_number` rows, gs_model.py:611, :636 — sorted unique pixels, which come out as one-pixel-wide rectangles): they
    are allowed for on top.

    Default: the whole cut AND the binning's counting pass in one call (gcp_rects_cut) with ONE device->host read and 2.5 B of
    scratch per pair — sized for what the last list cut on the device needed (at first: rows of 8 pixels and more, rectangles
    of 64 pairs and more on average).  A list of smaller boxes than that repeats the call once with room for anything that
    is boxes at all (rows of 2 pixels, rectangles of `min_mean_size` pairs: 8 B of scratch per pair + 3.5 B in the box
    arrays), and the next list starts from what this one needed.  one_call=False forces the step-by-step cut (three reads,
    14 B of scratch per pair)."""
    # int64 lists — what the reference's own make_rect_points_parallel returns (uitility.py:336-366) — are read as they are
    i64 = isinstance(rects, torch.Tensor) and rects.dtype == torch.int64
    r = _dev_tensor(rects, "rects", torch.int64 if i64 else torch.int32, (2,))
    m = r.size(0)
    dev = r.device
    if m == 0:
        z = torch.zeros(0, 2, dtype=torch.int32, device=dev)
        return RectBoxes(z, z.clone(), torch.zeros(1, dtype=torch.int32, device=dev), 0, 0)
    lib = _lib.load()
    cut = lib.gcp_rects_rows_i64 if i64 else lib.gcp_rects_rows
    carry_rows = min(max(int(carry_rows), 0), m)
    if one_call:
        # first with the room the last list cut on this device needed (default: rows of 8 pixels, rectangles of 64 pairs),
        # then — a list of smaller boxes — with room for anything that is boxes at all; only then step by step
        cf, cb = (0, carry_rows) if carry_at_end else (carry_rows, 0)
        sizing = _cut_sizing.get(dev.index, (CUT_SLOT_ROWS, CUT_MIN_RECT))
        for slot_rows, min_rect in (sizing, (CUT_SLOT_ROWS_ANY, max(1, int(min_mean_size)))):
            out = _cut_rects_once(r, i64, cf, cb, min_mean_size, slot_rows, min_rect)
            if not isinstance(out, str):
                return out
    cap = min(carry_rows + lib.gcp_rects_rows_capacity(m - carry_rows), m + 1)
    with torch.cuda.device(dev):
        st = _stream(dev)
        row_start = torch.empty(cap, dtype=torch.int32, device=dev)
        row_xy = torch.empty(cap, 2, dtype=torch.int32, device=dev)
        info = torch.empty(5, dtype=torch.int32, device=dev)
        ws = torch.empty(lib.gcp_rects_rows_workspace_bytes(m), dtype=torch.uint8, device=dev)
        _lib.check(cut(r.data_ptr(), m, cap, row_start.data_ptr(), row_xy.data_ptr(), info.data_ptr(), ws.data_ptr(),
                       ws.numel(), st), "gcp_rects_rows")
        n_rows, max_x, max_y, mn, not_boxes = info.tolist()
        del ws
        _require(mn >= 0, "rects: coordinates must lie in [0, 2^31) (negative ones are not supported)")
        if not_boxes:  # rows shorter than 2 pairs on average (the carry rows apart): the general route
            return None
        rect_row = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
        info2 = torch.empty(2, dtype=torch.int32, device=dev)
        ws2 = torch.empty(lib.gcp_rows_rectangles_workspace_bytes(n_rows), dtype=torch.uint8, device=dev)
        _lib.check(lib.gcp_rows_rectangles(row_start.data_ptr(), row_xy.data_ptr(), n_rows, rect_row.data_ptr(), info2.data_ptr(),
                                           ws2.data_ptr(), ws2.numel(), st), "gcp_rows_rectangles")
        n_rects = int(info2[0].item())
        if n_rects * min_mean_size > m:
            return None
        start = torch.empty(n_rects, 2, dtype=torch.int32, device=dev)
        end = torch.empty(n_rects, 2, dtype=torch.int32, device=dev)
        box_off = torch.empty(n_rects + 1, dtype=torch.int32, device=dev)
        _lib.check(lib.gcp_rectangle_boxes(rect_row.data_ptr(), row_start.data_ptr(), row_xy.data_ptr(), n_rects, m, start.data_ptr(),
                                           end.data_ptr(), box_off.data_ptr(), st), "gcp_rectangle_boxes")
    return RectBoxes(start, end, box_off, int(max_x), int(max_y))


_extent_seen = {}  # device index -> (width, height): the largest list extent pixels_min has measured there
_EXTENT_CACHE_CELLS = 1 << 24


def pixels_min(rects, values=None, image_size=None):
    """The distinct pixels of a rect list and the minimum of `values` over each one's pairs (csrc/gcp_pixels.hip): what
    the reference's `_create_alpha_brend_min` (gs_model.py:582-586) gets from `torch.unique(rects, dim=0)` +
    `scatter_reduce(amin)`.  Returns (unique_rects [U,2] in the dtype of `rects`, rows in (x, y) ascending order — the
    order torch.unique(dim=0) returns — and out f32[U]).  values=None: `out` is the index of every pixel's FIRST pair as
    the float `create_grad_alphabrend_min` carries it in (gs_model.py:728).
    image_size=(width, height) with every x <= width, y <= height sizes the pixel table.  Without it the list's extent is
    measured (one more pass over the list and one more device->host read) and remembered per device: the next call there
    starts from the remembered extent — any table that holds the list gives the same result — and measures again only if a
    coordinate falls outside it (the kernel says so), so a training loop pays for the measurement once.  The read that sizes
    the result (the number of distinct pixels, as torch.unique has it) remains in every case."""
    i64 = isinstance(rects, torch.Tensor) and rects.dtype == torch.int64
    r = _dev_tensor(rects, "rects", torch.int64 if i64 else torch.int32, (2,))
    n = r.size(0)
    dev = r.device
    v = None
    if values is not None:
        v = _dev_tensor(values, "values", torch.float32)
        _require(v.dim() == 1 and v.numel() == n, f"values: shape {tuple(v.shape)}, rects has {n} rows")
        _require(v.device == dev, "values: not on the device of rects")
    if n == 0:
        return r.new_empty(0, 2), torch.empty(0, dtype=torch.float32, device=dev)
    lib = _lib.load()

    def run(w, h):
        ws_bytes = lib.gcp_pixels_min_workspace_bytes(w, h)
        _require(ws_bytes > 0, f"rects: a {w + 1} x {h + 1} pixel table is beyond what pixels_min holds")
        cap = min((w + 1) * (h + 1), n)
        out_xy = torch.empty(cap, 2, dtype=r.dtype, device=dev)
        out_val = torch.empty(cap, dtype=torch.float32, device=dev)
        info = torch.empty(4, dtype=torch.int32, device=dev)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.gcp_pixels_min(r.data_ptr(), 1 if i64 else 0, v.data_ptr() if v is not None else None, n, w, h, out_xy.data_ptr(),
                                      out_val.data_ptr(), cap, info.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)), "gcp_pixels_min")
        u, outside, _, _ = info.tolist()
        return None if outside else (out_xy[:u], out_val[:u])

    with _on(dev):
        if image_size is not None:
            w, h = int(image_size[0]), int(image_size[1])
            _require(w >= 0 and h >= 0, "image_size: expected width >= 0 and height >= 0")
            out = run(w, h)
            _require(out is not None, f"rects: a coordinate lies outside [0, {w}] x [0, {h}]")
            return out
        seen = _extent_seen.get(dev.index)
        if seen is not None:
            out = run(*seen)
            if out is not None:
                return out
        ext = torch.empty(3, dtype=torch.int32, device=dev)
        _lib.check(lib.gcp_pixels_range(r.data_ptr(), 1 if i64 else 0, n, ext.data_ptr(), _stream(dev)), "gcp_pixels_range")
        w, h, mn = ext.tolist()
        _require(mn >= 0, "rects: coordinates must lie in [0, 2^31) (negative ones are not supported)")
        if seen is not None and (max(w, seen[0]) + 1) * (max(h, seen[1]) + 1) <= _EXTENT_CACHE_CELLS:
            w, h = max(w, seen[0]), max(h, seen[1])
        if (w + 1) * (h + 1) <= _EXTENT_CACHE_CELLS:
            _extent_seen[dev.index] = (w, h)
        out = run(w, h)
        _require(out is not None, "pixels_min: the measured extent does not hold the list")
        return out


def stable_sort_keys(keys, key_bits=None):
    """Stable sort of non-negative int32 keys on the HIP library: returns (sorted_keys int32[n], index int32[n]) with
    sorted_keys == keys[index] and equal keys in input order — `torch.sort(keys, stable=True)` as the reference needs
    it for its pixel keys (gs_model.py:546-547), in ceil(key_bits/8) radix passes.  `key_bits` defaults to the bits of
    keys.max() (one device->host read); pass it to stay asynchronous."""
    k = _dev_tensor(keys, "keys", torch.int32)
    _require(k.dim() == 1, "keys: expected a 1-D tensor")
    n = k.numel()
    dev = k.device
    lib = _lib.load()
    out_k = torch.empty_like(k)
    out_i = torch.empty_like(k)
    if n == 0:
        return out_k, out_i
    if key_bits is None:
        mx = int(k.max().item())
        _require(int(k.min().item()) >= 0, "keys: negative keys are not supported")
        key_bits = max(1, mx.bit_length())
    with torch.cuda.device(dev):
        ws = torch.empty(lib.gcp_sort_workspace_bytes(n), dtype=torch.uint8, device=dev)
        _lib.check(lib.gcp_sort_pairs_u32(k.data_ptr(), n, int(key_bits), out_k.data_ptr(), out_i.data_ptr(), ws.data_ptr(),
                                          ws.numel(), _stream(dev)), "gcp_sort_pairs_u32")
    return out_k, out_i


def rects_key_bits(rects):
    """Bits of the largest pixel key y*10000 + x of a rect list (one device->host read of two ints; raises on negative
    coordinates).  Callers that know the image size pass key_bits = (height*10000 + width).bit_length() instead."""
    r = _dev_tensor(rects, "rects", torch.int32, (2,))
    out = torch.empty(2, dtype=torch.int32, device=r.device)
    with torch.cuda.device(r.device):
        _lib.check(_lib.load().gcp_rects_key_range(r.data_ptr(), r.size(0), out.data_ptr(), _stream(r.device)), "gcp_rects_key_range")
    mx, mn = out.tolist()
    _require(r.size(0) == 0 or mn >= 0, "rects: negative coordinates are not supported")
    return max(1, int(mx).bit_length())


def sort_rects(rects, key_bits=None, image_size=None):
    """Stable sort of the reference's pixel keys straight from its rect list (gs_model.py:538-541, :546-547): returns
    (sorted_key int32[n], index int32[n]) = torch.sort(rects[:,1]*10000 + rects[:,0], stable=True) without ever writing
    the unsorted key array.  `image_size` = (width, height) with every x <= width, y <= height (the Function knows them,
    gs_model.py:666): the passes run on compact pixel ids (21 bits, three passes of 7 at 1920x1080).  Else `key_bits`:
    bits of the largest key ((H*10000+W).bit_length()); neither = one read-back of the key range."""
    r = _dev_tensor(rects, "rects", torch.int32, (2,))
    n = r.size(0)
    dev = r.device
    out_k = torch.empty(n, dtype=torch.int32, device=dev)
    out_i = torch.empty(n, dtype=torch.int32, device=dev)
    if n == 0:
        return out_k, out_i
    id_width = 0
    if image_size is not None:
        w, h = int(image_size[0]), int(image_size[1])
        _require(0 <= w < 10000 and h >= 0, "image_size: expected 0 <= width < 10000 and height >= 0")
        bits = max(1, (h * (w + 1) + w).bit_length())
        if bits <= 24:
            id_width, key_bits = w + 1, bits
        else:  # too large for the compact form: plain keys
            key_bits = max(1, (h * 10000 + w).bit_length())
    elif key_bits is None:
        key_bits = rects_key_bits(r)
    lib = _lib.load()
    with torch.cuda.device(dev):
        ws = torch.empty(lib.gcp_sort_workspace_bytes(n), dtype=torch.uint8, device=dev)
        _lib.check(lib.gcp_sort_rects(r.data_ptr(), n, int(key_bits), int(id_width), out_k.data_ptr(), out_i.data_ptr(), ws.data_ptr(),
                                      ws.numel(), _stream(dev)), "gcp_sort_rects")
    return out_k, out_i


def compact_finish(inclusive, self_values, mode, begin=0, end=None, dropped=None):
    """The tail of _create_alpha_brend (gs_model.py:557-564) on the un-sorted inclusive values, rows [begin, end):
    -> (values f32[n_kept] = inclusive / self (mode 0) or inclusive - self (mode 1) of the rows whose inclusive value is
    not 0, keep bool[end - begin]).  One device->host read (the kept count sizes the returned tensor, as the reference's
    boolean-mask indexing does).  dropped: the per-4096 zero counts `scan_boxes(..., count_dropped=True)` took while it
    wrote `inclusive` (used when begin is a multiple of 4096, ignored otherwise)."""
    inc = _dev_tensor(inclusive, "inclusive", torch.float32)
    sv = _dev_tensor(self_values, "self_values", torch.float32)
    _require(inc.dim() == 1 and sv.shape == inc.shape, "inclusive / self_values: expected two 1-D tensors of one length")
    end = inc.numel() if end is None else int(end)
    begin = int(begin)
    _require(0 <= begin <= end <= inc.numel(), "compact_finish: bad row range")
    dev = inc.device
    n = end - begin
    values = torch.empty(n, dtype=torch.float32, device=dev)
    keep = torch.empty(n, dtype=torch.uint8, device=dev)
    if n == 0:
        return values, keep.view(torch.bool)
    lib = _lib.load()
    count = torch.empty(1, dtype=torch.int32, device=dev)
    if dropped is not None:
        dropped = _dev_tensor(dropped, "dropped", torch.int32)
        _require(dropped.numel() == (inc.numel() + 4095) // 4096, "dropped: expected one count per 4096 rows of `inclusive`")
        if begin % 4096:
            dropped = None
    with torch.cuda.device(dev):
        ws = torch.empty(lib.gcp_compact_workspace_bytes(n), dtype=torch.uint8, device=dev)
        _lib.check(lib.gcp_compact_finish(inc.data_ptr(), sv.data_ptr(), begin, end, int(mode), values.data_ptr(), keep.data_ptr(),
                                          count.data_ptr(), dropped.data_ptr() if dropped is not None else None, ws.data_ptr(),
                                          ws.numel(), _stream(dev)), "gcp_compact_finish")
    kept = int(count.item())
    mask = keep.view(torch.bool)
    return values[:kept], (mark_all_kept(mask) if kept == n else mask)


def gather_f32(src, index):
    """src[index] for an int32 permutation (gs_model.py:548)."""
    src = _dev_tensor(src, "src", torch.float32)
    index = _dev_tensor(index, "index", torch.int32)
    dst = torch.empty(index.numel(), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        _lib.check(_lib.load().gcp_gather_f32(src.data_ptr(), index.data_ptr(), dst.data_ptr(), index.numel(), _stream(src.device)),
                   "gcp_gather_f32")
    return dst


def unsort_finish(inclusive, sorted_x, index, mode):
    """Un-sort fused with `/ self` (mode 0) or `- self` (mode 1) and the `!= 0` test: -> (full f32[n], keep bool[n]) in the
    original pair order (gs_model.py:555-564)."""
    dev = inclusive.device
    n = inclusive.numel()
    full = torch.empty(n, dtype=torch.float32, device=dev)
    keep = torch.empty(n, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().gcp_unsort_finish(inclusive.data_ptr(), sorted_x.data_ptr(), index.data_ptr(), full.data_ptr(),
                                                 keep.data_ptr(), n, int(mode), _stream(dev)), "gcp_unsort_finish")
    return full, keep.view(torch.bool)


def render_cameras(cameras, n_streams=3):
    """Forward-render several cameras of one batch (the reference loops over them one after the other,
    gs_model.py:402-449) on `n_streams` HIP streams: the binning of one camera has to hand its entry count to the
    host (a per-STREAM synchronisation, like the reference's `.item()`), and while that stream waits the other
    streams keep the GPU busy with their blends.  `cameras`: iterable of dicts with keys start, end, mean, vinv,
    opacity, l_d, width, height.  Returns the list of images [(H+1, W+1, 3)] in input order."""
    cameras = list(cameras)
    if not cameras:
        return []
    dev = cameras[0]["start"].device
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, min(n_streams, len(cameras))))]
    cur = torch.cuda.current_stream(dev)
    for st in streams:
        st.wait_stream(cur)
    images = [None] * len(cameras)
    for i, c in enumerate(cameras):
        st = streams[i % len(streams)]
        with torch.cuda.stream(st):
            bins = bin_tiles(c["start"], c["end"], c["width"], c["height"])
            images[i] = blend_forward(bins, c["start"], c["end"], c["mean"], c["vinv"], c["opacity"], c["l_d"])
            for t in (c["start"], c["end"], c["mean"], c["vinv"], c["opacity"], c["l_d"]):
                t.record_stream(st)
    for st in streams:
        cur.wait_stream(st)
    for img in images:
        img.record_stream(cur)
    return images
