"""Autograd layer and the scan call sites of the reference, on the HIP library.

The reference's `cuda_kernel.py` (reference: cuda_kernel.py:1-19) is a stale JIT
loader; its autograd code lives in gs_model.py.  This module is what
BASELINE.json's north_star asks `cuda_kernel.py` to be: real
`torch.autograd.Function`s over the grouped scans, plus the two helpers of
`custom_autograd_grouped_cumprod` that are the only in-repo callers of the
extension:

  create_alpha_brend  <- gs_model.py:544-566 (_create_alpha_brend)
  grad_cumsum         <- gs_model.py:716-722
  unique              <- gs_model.py:538-541 (pixel key = y*10000 + x, int32)

and the helpers its chunk loop (`_forward_batch` / `_backward_batch`, gs_model.py:598-663) calls either side of them:

  create_rects                <- gs_model.py:480-482 (_create_rects -> uitility.py:336-366)
  create_alpha_brend_min      <- gs_model.py:582-586 (_create_alpha_brend_min: torch.unique(dim=0) + scatter_reduce(amin))
  create_grad_alphabrend_min  <- gs_model.py:724-730
  cat_alpha_brend, mask_tensor, sort_tensor, mask_zero_T  <- gs_model.py:589-594, :525-531, :517-523, :575-578

all of them also static methods of `custom_autograd_grouped_cumprod` under the reference's names.

Deliberate differences from the reference, all result-preserving:
  * `torch.sort(..., stable=True)`: depth order inside a pixel is carried only by
    sort stability; the reference calls torch.sort without it (gs_model.py:547),
    which is stable on CUDA in practice and NOT on CPU for small inputs.
  * the sort itself is the library's own stable LSD radix sort (raster.sort_rects -> gcp_sort_rects: the
    pixel key is computed from `rects` inside the first pass, only the significant key bits are sorted,
    int32 payload; bit-identical to torch.sort(stable=True) of the keys).
  * gather (gs_model.py:548), scan (:551/:553) and un-sort `output[torch.argsort(index)]` (:555, a
    second radix sort in the reference) are ONE indexed scan (gcp_cum*_indexed): values are gathered
    through the permutation on the way in and the inclusive results scattered through it on the way out.
  * the `!= 0` mask, the boolean-mask compaction and `/ self` | `- self` (:560-564) are one stable
    stream compaction over the original order (gcp_compact_finish); the element-wise step commutes with
    the compaction the reference applies first.
  * grad_cumsum's flip / scan / flip is one reverse scan on the same sorted keys.
"""
import contextlib

import torch

from . import grouped_cumprod as _ext
from . import raster as _raster

__all__ = [
    "GroupedCumprod",
    "GroupedCumsum",
    "grouped_cumprod",
    "grouped_cumsum",
    "unique",
    "pixel_key_bits",
    "create_alpha_brend",
    "create_alpha_blend",
    "PreparedRects",
    "grad_cumsum",
    "create_alpha_brend_min",
    "create_grad_alphabrend_min",
    "cat_alpha_brend",
    "create_rects",
    "mask_zero_T",
    "mask_tensor",
    "sort_tensor",
    "create_alpha_brend_boxes",
    "grad_cumsum_boxes",
    "custom_autograd_grouped_cumprod",
    "tile_capacity",
    "capacity_exceeded",
    "GraphedStep",
    "capture_leaf",
]

# Capture-safe mode of the Function (no device->host read inside forward / backward): the caller bounds the number of
# (tile, Gaussian) entries of a camera; whether a bound was too small is OR-ed into one sticky int32 per device that
# lives outside every graph's memory pool (so a replayed graph keeps reporting) and is read back once per step.
_capacity = None
_sticky = {}  # device index -> int32[1], never freed: captured graphs hold its address


def _sticky_flag(device):
    flag = _sticky.get(device.index)
    if flag is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("tile_capacity: enter the context BEFORE the capture begins (the overflow flag of "
                               f"{device} must not be allocated inside a graph's private pool)")
        flag = _sticky[device.index] = torch.zeros(1, dtype=torch.int32, device=device)
    return flag


@contextlib.contextmanager
def tile_capacity(n_entries, device=None):
    """Inside this context `custom_autograd_grouped_cumprod.apply` bins with a caller-given bound on the (tile, Gaussian)
    entry count (about 3 per Gaussian at BASELINE's box sizes; a Gaussian covers ceil(w/16+1) x ceil(h/16+1) tiles at
    most) instead of reading the exact count back from the device — the reference synchronises on `.item()` per chunk
    (gs_model.py:677,793,801-802).  Forward and backward then queue their kernels without ever waiting for the GPU and
    can be captured into a HIP graph (`GraphedStep`).  Pass image_width / image_height as Python ints or CPU tensors (a
    device tensor would have to be read back).  Call `capacity_exceeded()` once per step — also after every replay of a
    captured step: the flag is written by the captured kernels themselves."""
    global _capacity
    if torch.cuda.is_available():
        _sticky_flag(torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device))
    old, _capacity = _capacity, int(n_entries)
    try:
        yield
    finally:
        _capacity = old


def capacity_exceeded():
    """True if any Function call — eager, or replayed from a graph — since the last check ran out of its
    `tile_capacity` (those calls dropped the Gaussians that did not fit, from the back of the depth order, with zero
    gradients: rerun the step with a larger bound).  One device->host read per device in use; resets the flags."""
    hit = False
    for flag in _sticky.values():
        with torch.cuda.device(flag.device):
            if bool(flag.item()):
                hit = True
                flag.zero_()
    return hit


def capture_leaf(t):
    """Mark `t` as an autograd leaf created FOR a graph capture — under the capturing stream, with no history from eager steps
    (`p.detach().requires_grad_()` is one) — so that `custom_autograd_grouped_cumprod` accepts it inside `torch.cuda.graph`
    (see `_refuse_stale_leaves_in_capture`).  `GraphedStep` marks its own aliases.  Returns `t`."""
    t._gcp_capture_leaf = True
    return t


def _refuse_stale_leaves_in_capture(*tensors):
    """While the current stream is capturing, every autograd leaf the differentiable inputs descend from must have been
    made for this capture (`capture_leaf`; `GraphedStep` does it).  A leaf that has been through an eager step keeps its
    AccumulateGrad node — bound to the stream of that step, alive as long as any output of it is — and a backward captured on
    another stream then pulls that stream into the capture: `capture_end` of this ROCm stack dies with SIGSEGV
    (tools/capture_repro.py).  Python cannot see a node's stream, so the test is the conservative one: an
    unmarked leaf raises — a RuntimeError that says how to proceed instead of a crash at the end of the capture."""
    if not torch.cuda.is_available() or not torch.cuda.is_current_stream_capturing():
        return
    stack, seen = [], set()

    def check(leaf):
        if leaf.requires_grad and not getattr(leaf, "_gcp_capture_leaf", False):
            raise RuntimeError(
                "custom_autograd_grouped_cumprod inside torch.cuda.graph: a differentiable input descends from an autograd leaf "
                f"(shape {tuple(leaf.shape)}) that was not created for this capture.  If that leaf has been through an eager step "
                "whose outputs are still alive, its AccumulateGrad node belongs to another stream and capture_end crashes on this "
                "ROCm stack.  Capture the step with cuda_kernel.GraphedStep(fn, params, capacity=...) — it traces through fresh "
                "aliases of the parameters — or, if you made fresh leaves yourself (p.detach().requires_grad_() under the "
                "capturing stream), mark them with cuda_kernel.capture_leaf(t).")

    for t in tensors:
        if isinstance(t, torch.Tensor) and t.requires_grad:
            if t.grad_fn is None:
                check(t)
            else:
                stack.append(t.grad_fn)
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        var = getattr(fn, "variable", None)  # AccumulateGrad: the leaf it accumulates into
        if isinstance(var, torch.Tensor):
            check(var)
        for nxt, _ in fn.next_functions:
            stack.append(nxt)


class GraphedStep:
    """One training-step body — `fn(*params)` -> loss, or a tuple whose first entry is the loss — with its backward,
    captured into ONE HIP graph and replayed with `replay()` -> (outputs of fn, gradients w.r.t. params).

    The step is traced through fresh ALIASES of the parameters (`p.detach().requires_grad_()`: same storage, so
    in-place updates of `p` are seen by every replay; new autograd leaves).  Why: a leaf's AccumulateGrad node is
    created lazily under the stream that is current when the leaf first enters an autograd graph, and stays alive as
    long as any such graph does (the previous step's loss, logged outputs, ...).  After eager steps on the default
    stream a backward captured on a side stream hands its gradients to that old node; the engine then makes the
    node's stream — the legacy default stream — wait on an event recorded inside the capture, which pulls the default
    stream into it, and `capture_end` of this ROCm stack segfaults (tools/capture_repro.py reproduces it with three
    lines of plain PyTorch; torch only warns "AccumulateGrad node's stream does not match").  Aliases created on the
    capture stream have no history, so whatever the caller still holds cannot reach the capture.

    `capacity`: bound for `tile_capacity` around every trace of `fn` (None: `fn` must not contain a host read).  After
    `replay()` call `capacity_exceeded()` as after an eager step."""

    def __init__(self, fn, params, capacity=None, warmup=2, stream=None):
        self.params = list(params)
        if not self.params:
            raise ValueError("GraphedStep: no parameters")
        dev = self.params[0].device
        self._fn = fn
        self._capacity = capacity
        self.stream = torch.cuda.Stream(device=dev) if stream is None else stream
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            for _ in range(max(1, int(warmup))):  # allocator warm-up and lazy initialisations outside the capture
                self._trace()
        torch.cuda.current_stream(dev).wait_stream(self.stream)
        self.graph = torch.cuda.CUDAGraph()
        with self._bound(), torch.cuda.graph(self.graph, stream=self.stream):
            self.outputs, self.grads = self._trace(bounded=False)

    def _bound(self):
        return tile_capacity(self._capacity, self.params[0].device) if self._capacity is not None else contextlib.nullcontext()

    def _trace(self, bounded=True):
        with (self._bound() if bounded else contextlib.nullcontext()):
            leaves = [capture_leaf(p.detach().requires_grad_(True)) for p in self.params]
            out = self._fn(*leaves)
            loss = out[0] if isinstance(out, (tuple, list)) else out
            grads = torch.autograd.grad(loss, leaves, allow_unused=True)
        outs = tuple(o.detach() if isinstance(o, torch.Tensor) else o for o in out) if isinstance(out, (tuple, list)) else out.detach()
        return outs, grads

    def replay(self):
        self.graph.replay()
        return self.outputs, self.grads


class GroupedCumprod(torch.autograd.Function):
    """y = inclusive product scan of x inside runs of equal adjacent `key` (int32).

    backward = grouped_cumprod_backward (reference kernel semantics,
    cuda_kernel/grouped_cumprod_backward.cu:22-29, incl. the 0 -> 1e-8 divisor).
    """

    @staticmethod
    def forward(ctx, x, key):
        y = torch.empty_like(x)
        _ext.grouped_cumprod_forward(x, key, y)
        ctx.save_for_backward(x, y, key)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, y, key = ctx.saved_tensors
        grad_x = torch.empty_like(x)
        # The kernel reads group ends from the runs of `inv`; the key array itself has
        # the same runs, and inv_len is only part of the reference signature.
        inv_len = torch.zeros(1, dtype=torch.int32, device=x.device)
        _ext.grouped_cumprod_backward(x, y, grad_y.contiguous(), key, grad_x, inv_len)
        return grad_x, None


class GroupedCumsum(torch.autograd.Function):
    """y = inclusive sum scan inside runs of equal adjacent `key`; backward = suffix sums."""

    @staticmethod
    def forward(ctx, x, key):
        y = torch.empty_like(x)
        _ext.grouped_cumsum_forward(x, key, y)
        ctx.save_for_backward(key)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        (key,) = ctx.saved_tensors
        grad_x = torch.empty_like(grad_y)
        _ext.grouped_cumsum_reverse(grad_y.contiguous(), key, grad_x)
        return grad_x, None


def grouped_cumprod(x, key):
    return GroupedCumprod.apply(x, key)


def grouped_cumsum(x, key):
    return GroupedCumsum.apply(x, key)


def unique(rects):
    """Pixel key of every pair: y*10000 + x in int32 (reference: gs_model.py:538-541)."""
    with torch.no_grad():
        rects = rects.to(torch.int32)
        return rects[:, 1] * 10000 + rects[:, 0]


def pixel_key_bits(image_width, image_height):
    """Bits of the largest pixel key y*10000 + x of an image (24 at 1920x1080, 25 at 3840x2160): pass it as `key_bits` to
    `create_alpha_brend` / `grad_cumsum` and the sort never has to read the key range back from the device."""
    return max(1, (int(image_height) * 10000 + int(image_width)).bit_length())


def create_alpha_brend(rects, anti_opacity, flag, cutting_number=None, *, key_bits=None, image_size=None, route="auto"):
    """Per-pixel exclusive transmittance (flag="cumprod") or exclusive prefix sum
    (flag="cumsum") of `anti_opacity`, returned in the ORIGINAL pair order.

    reference: gs_model.py:544-566.  Same results, two routes:

    route="boxes" — what `rects` always is when the reference calls this: `_create_rects` (gs_model.py:480-482,
      uitility.py:336-366) writes one row-major box per Gaussian in depth order.  The list is cut back into rectangles
      (raster.rects_to_boxes: two one-pass stream compactions), the rectangles are binned into 16x16 tiles, and every
      pixel walks its tile's list front to back reading and writing each pair in place (raster.scan_boxes) — no M-sized
      sort, every pixel scanned in the CPU path's own order.
    route="sort" — any list of pixel coordinates: (1) key + stable sort (:546-547), the pixel key computed inside the first
      radix pass straight from `rects`; (2) gather, grouped scan, un-sort (:548-555) as ONE indexed scan.
    route="auto" (default) tries the cut and sorts when the list does not come apart into boxes (fewer than 8 pairs per
      rectangle on average).
    Either way the tail — drop `cutting_number` carry rows, compact the entries whose inclusive value is exactly 0
    (:557-560, :575-578), inclusive / self (:562) or inclusive - self (:564) — is ONE stable stream compaction in
    original order.  Returns [values, mask].  Keyword-only extensions for the sort route: `image_size=(width, height)` —
    what the Function holds as image_width / image_height (gs_model.py:666); the sort then runs on compact pixel ids —
    or `key_bits` (`pixel_key_bits(width, height)`); with neither the key range is read back once."""
    if route not in ("auto", "boxes", "sort"):
        raise ValueError(route)
    with torch.no_grad():
        if route != "sort":
            out = _rects_as_boxes(rects, anti_opacity, flag, cutting_number)
            if out is not None:
                return out
            if route == "boxes":
                raise RuntimeError("create_alpha_brend: rects do not come apart into boxes (route='boxes')")
        sorted_inv, index = _raster.sort_rects(rects.rects if isinstance(rects, PreparedRects) else rects, key_bits, image_size)
        return _scan_unsort_compact(sorted_inv, index, anti_opacity, flag, cutting_number)


_MAX_WALK_PIXELS = 1 << 26  # beyond this the tile grid of the walk is mostly empty tiles: sort instead


class PreparedRects:
    """A rect list cut into boxes and binned into tiles ONCE, for the several calls a step makes on the same list — the
    reference's forward calls `_create_alpha_brend(rects, ...)` and its backward `grad_cumsum(rects, ...)` on the rects it
    rebuilds from the same boxes (gs_model.py:601-612, :630-643).  Pass it wherever `rects` is expected:

        prep = PreparedRects(rects)
        T, mask = create_alpha_brend(prep, anti_opacity, "cumprod")
        S, mask = grad_cumsum(prep, grad)

    `boxes` is None when the list is not a concatenation of boxes; the calls then sort (`rects` is kept for that).
    carry_rows: the number of `cutting_number` rows the list starts — or, carry_at_end (`grad_cumsum`), ends — with in a
    chunked call (gs_model.py:611, :636): single pixels, which are cut into one-pixel-wide rectangles and walked like the
    boxes."""

    def __init__(self, rects, carry_rows=0, carry_at_end=False):
        self.rects = rects
        self.shape = rects.shape
        with torch.no_grad():
            rb = _raster.rects_to_boxes(rects, carry_rows=carry_rows, carry_at_end=carry_at_end)
            if rb is not None and (rb.width + 1) * (rb.height + 1) > _MAX_WALK_PIXELS:
                rb = None
            self.boxes = rb
            self.bins = rb.bin() if rb is not None else None


def _rects_as_boxes(rects, values, flag, cutting_number=None):
    """The boxes route from nothing but the rect list; None if the list is not a concatenation of boxes."""
    if flag not in ("cumprod", "cumsum", "cumsum_reverse"):
        raise ValueError(flag)
    values = values.detach().contiguous()
    n = values.numel()
    if n != int(rects.shape[0]):
        raise RuntimeError(f"values: {n} rows, rects has {int(rects.shape[0])}")
    # the walk's output buffers and their two fills are queued BEFORE the cut: they run while the host waits for its read-back
    buffers = _raster.finish_buffers(values) if (values.is_cuda and values.dtype == torch.float32 and n) else None
    prep = rects if isinstance(rects, PreparedRects) else PreparedRects(rects, carry_rows=int(cutting_number) if cutting_number else 0,
                                                                       carry_at_end=flag == "cumsum_reverse")
    rb, bins = prep.boxes, prep.bins
    if rb is None:
        return None
    return _walk_and_finish(bins, rb.start, rb.end, rb.box_off, values, flag, cutting_number, buffers)


def _walk_and_finish(bins, start, end, box_off, values, flag, cutting_number=None, buffers=None):
    """Walk + tail of _create_alpha_brend (gs_model.py:546-564) from binned boxes.  The walk writes the FINAL values
    (inclusive / self, inclusive - self) and clears the mask byte of every pair whose inclusive value is exactly 0; one
    device->host read of the kept count — the one that sizes the result — decides: nothing dropped (the usual case: a
    factor 1 - alpha G is 0 only for alpha G == 1, a product underflows only behind ~100 opaque layers) and the walk's
    output IS the result, else one pass moves the kept values together."""
    n = values.numel()
    mode = {"cumprod": 0, "cumsum": 1, "cumsum_reverse": 2}[flag]
    cut = int(cutting_number) if cutting_number else 0
    begin, stop = (0, n - cut) if flag == "cumsum_reverse" else (cut, n)
    if n == 0 or bins.n_tile_pairs == 0:
        if n:
            raise RuntimeError("values: the boxes expand to no pair at all")
        return [values.new_empty(0), torch.zeros(0, dtype=torch.bool, device=values.device)]
    final, keep, dropped = _raster.finish_boxes(bins, start, end, box_off, values, mode, buffers=buffers)
    values_out, mask = _raster.compact_kept(final, keep, dropped, max(0, min(begin, n)), max(0, stop))
    return [values_out, mask]


def _scan_unsort_compact(sorted_key, index, anti_opacity, flag, cutting_number=None):
    """Everything of _create_alpha_brend after the sort (gs_model.py:548-566) on the HIP library.
    flag "cumsum_reverse" is grad_cumsum's suffix form (gs_model.py:716-722), whose carry rows sit at the END."""
    if flag not in ("cumprod", "cumsum", "cumsum_reverse"):
        raise ValueError(flag)
    anti_opacity = anti_opacity.detach().contiguous()
    inclusive = torch.empty_like(anti_opacity)
    if flag == "cumprod":
        _ext.grouped_cumprod_forward_indexed(anti_opacity, sorted_key, index, inclusive)
    elif flag == "cumsum":
        _ext.grouped_cumsum_forward_indexed(anti_opacity, sorted_key, index, inclusive)
    else:
        _ext.grouped_cumsum_reverse_indexed(anti_opacity, sorted_key, index, inclusive)
    n = anti_opacity.numel()
    cut = int(cutting_number) if cutting_number else 0
    begin, end = (0, n - cut) if flag == "cumsum_reverse" else (cut, n)
    values, keep = _raster.compact_finish(inclusive, anti_opacity, 0 if flag == "cumprod" else 1, begin, end)
    return [values, keep]


create_alpha_blend = create_alpha_brend  # spelling alias


def _scan_boxes_compact(startpoint, endpoint, values, image_width, image_height, flag):
    w, h = int(image_width), int(image_height)
    values = values.detach().contiguous()
    bins = _raster.bin_tiles(startpoint, endpoint, w, h)
    box_off = _raster.box_offsets(startpoint, endpoint, w, h)
    m = int(box_off[-1].item())
    if values.numel() != m:
        raise RuntimeError(f"values: {values.numel()} rows, but the boxes expand to {m} pairs")
    return _walk_and_finish(bins, startpoint, endpoint, box_off, values, flag)


def create_alpha_brend_boxes(startpoint, endpoint, anti_opacity, image_width, image_height, flag="cumprod"):
    """`create_alpha_brend` for callers that still hold the boxes the rects were expanded from (reference: gs_model.py:601
    `_create_rects(startpoint, endpoint)` feeds :607).  Nothing M-sized is sorted: the Gaussians are binned into 16x16
    tiles (K ~ 3 entries per Gaussian) and every pixel walks its tile's depth-ordered list, reading and writing each
    pair at its Gaussian-major position (raster.scan_boxes), then the same stream compaction.  `anti_opacity` is in the
    reference's Gaussian-major rect order; returns the same [values, mask] as `create_alpha_brend(rects, ...)`, every
    pixel scanned strictly in depth order (the association of the CPU path) instead of the tree order of the flat scan:
    values equal within fp32 round-off; the mask (inclusive value != 0, gs_model.py:560) equal except where a value is
    exactly 0 in one summation order only — signed sums that cancel, products at the edge of underflow: a handful among
    1.65e8 pairs, as between the reference's CPU and GPU runs."""
    if flag not in ("cumprod", "cumsum"):
        raise ValueError(flag)
    with torch.no_grad():
        return _scan_boxes_compact(startpoint, endpoint, anti_opacity, image_width, image_height, flag)


def _mask_in_order(out, mask_order):
    """[values, mask] with the mask in the caller's order: "original" (row i of the mask is row i of the inputs) or
    "reference" — what gs_model.py:716-722 returns: the mask of the FLIPPED arrays, never flipped back (:721-722), row i
    of it being row n - 1 - i of the kept input range; `_backward_batch` applies it as it is (:642-645)."""
    if mask_order == "reference":
        flipped = out[1].flip(0)
        return [out[0], _raster.mark_all_kept(flipped) if _raster.all_kept(out[1]) else flipped]
    return out


def grad_cumsum(rects, grad, cutting_number=None, *, key_bits=None, image_size=None, route="auto", mask_order="original"):
    """Per-pixel exclusive SUFFIX sum of `grad` in original pair order.

    reference: gs_model.py:716-722 (flip, _create_alpha_brend(flag="cumsum"), flip).
    Flipping a stably sorted list and summing forward equals summing backward on the
    un-flipped list, so this walks the tile lists back to front (route "boxes") or runs one
    indexed reverse scan (route "sort"); routes as for `create_alpha_brend`.  `cutting_number` counts
    rows at the START of the flipped arrays, i.e. the LAST rows of the inputs
    (gs_model.py:636 appends the carry rows at the end before the flip).
    mask_order="original" (default): the mask is returned in the order of the inputs; mask_order="reference": in the
    reference's order — flipped, as gs_model.py:721-722 leaves it — bit for bit what the reference returns.
    """
    if route not in ("auto", "boxes", "sort"):
        raise ValueError(route)
    if mask_order not in ("original", "reference"):
        raise ValueError(mask_order)
    with torch.no_grad():
        if route != "sort":
            out = _rects_as_boxes(rects, grad, "cumsum_reverse", cutting_number)
            if out is not None:
                return _mask_in_order(out, mask_order)
            if route == "boxes":
                raise RuntimeError("grad_cumsum: rects do not come apart into boxes (route='boxes')")
        sorted_inv, index = _raster.sort_rects(rects.rects if isinstance(rects, PreparedRects) else rects, key_bits, image_size)
        return _mask_in_order(_scan_unsort_compact(sorted_inv, index, grad, "cumsum_reverse", cutting_number), mask_order)


def grad_cumsum_boxes(startpoint, endpoint, grad, image_width, image_height, *, mask_order="original"):
    """`grad_cumsum` (gs_model.py:716-722) from the boxes: the tile lists walked back to front; same [values, mask] as
    `grad_cumsum(rects, grad)` (values within fp32 round-off, masks as explained at `create_alpha_brend_boxes`)."""
    if mask_order not in ("original", "reference"):
        raise ValueError(mask_order)
    with torch.no_grad():
        return _mask_in_order(_scan_boxes_compact(startpoint, endpoint, grad, image_width, image_height, "cumsum_reverse"), mask_order)


def _rects_and_extent(rects, image_size):
    """(the rect tensor, (width, height) or None) of a tensor or a PreparedRects (whose cut has found the extent)"""
    if isinstance(rects, PreparedRects):
        if image_size is None and rects.boxes is not None and rects.shape[0]:
            image_size = (rects.boxes.width, rects.boxes.height)
        rects = rects.rects
    return rects, image_size


def create_alpha_brend_min(rects, T, *, image_size=None):
    """The distinct pixels of `rects` and every pixel's smallest `T` — the carry the reference's chunked forward hands
    from one chunk to the next (a pixel's transmittance behind the chunk: T only falls along its list).

    reference: gs_model.py:582-586 (`_create_alpha_brend_min`; called by every chunk, :609 / :615):
    `torch.unique(rects, return_inverse=True, dim=0)` — a lexicographic sort of the M rows — and
    `scatter_reduce(0, inv, T, "amin", include_self=False)`.  Here: ONE pass over the list, every pair taking the minimum
    with its pixel's cell of an image-sized table, and the table read out in (x, y) order (csrc/gcp_pixels.hip); nothing
    M-sized is sorted or written.  Returns [unique_rects (dtype of rects, the rows torch.unique returns in its order),
    T_min f32] bit for bit — a minimum does not depend on the order it is taken in (one exception that is none: -0 == +0, and a
    pixel holding both returns -0 here, whichever its reduction met first in the reference).  Keyword-only extension:
    image_size=(width, height), what the Function holds (gs_model.py:666), spares the pass that finds the list's extent;
    a PreparedRects brings it along."""
    with torch.no_grad():
        r, image_size = _rects_and_extent(rects, image_size)
        u, t_min = _raster.pixels_min(r, T, image_size)
        return [u, t_min]


def create_grad_alphabrend_min(rects, grad, *, image_size=None):
    """The distinct pixels of `rects` and `grad` at every pixel's FIRST pair — the backward's carry between chunks.

    reference: gs_model.py:724-730: `_create_alpha_brend_min(rects, arange(n).float())`, then `grad[min.to(int32)]` — the
    index travels as a float32, exact below 2^24 pairs and rounded to nearest-even above (the reference then reads a
    neighbouring row; reproduced as it is, except that an index rounded up past the last row — a device-side assert
    there — reads the last row).  Same single pass as `create_alpha_brend_min`, the pair's index in place of a value."""
    with torch.no_grad():
        r, image_size = _rects_and_extent(rects, image_size)
        u, first = _raster.pixels_min(r, None, image_size)
        n = int(r.shape[0])
        return [u, grad[first.to(torch.int64).clamp_(max=max(n - 1, 0))]]


def cat_alpha_brend(values, rects):
    """reference: gs_model.py:589-594 (`_cat_alpha_brend`): the carry rows put in front of (forward, :611) or behind
    (backward, :635) a chunk's own.  Two concatenations."""
    with torch.no_grad():
        return [torch.cat((values[0], values[1]), dim=0), torch.cat((rects[0], rects[1]), dim=0)]


def create_rects(startpoint, endpoint):
    """reference: gs_model.py:480-482 (`_create_rects` -> uitility.py:336-366): every pixel of every box, box after box,
    row-major inside a box, int32 [M,2] (x, y).  One expansion kernel (raster.expand_rects) instead of five M-sized
    temporaries."""
    with torch.no_grad():
        return _raster.expand_rects(startpoint, endpoint, 1 << 30, 1 << 30)


def mask_tensor(mask, *tensors):
    """reference: gs_model.py:525-531 (`_mask_tensor`): `tensor[mask]` for each of the M-sized arrays of a chunk (six of them
    at :618, seven at :645).  A mask that `create_alpha_brend` / `grad_cumsum` returned from a call that dropped nothing — the
    usual case, known from the kept count the call read back anyway — is all ones: the tensors are then handed through as they
    are (no copies; each boolean-mask indexing of 1.65e8 rows costs 1-2 ms)."""
    with torch.no_grad():
        if _raster.all_kept(mask):
            return list(tensors)
        return [t[mask] for t in tensors]


def sort_tensor(index, *tensors):
    """reference: gs_model.py:517-523 (`_sort_tensor`)."""
    with torch.no_grad():
        return [t[index] for t in tensors]


def mask_zero_T(T):
    """reference: gs_model.py:575-578."""
    with torch.no_grad():
        mask = T != 0
        return [T[mask], mask]


class custom_autograd_grouped_cumprod(torch.autograd.Function):
    """The reference's rasterise-and-blend Function, same name and call signature
    (reference: gs_model.py:477-820; call site gs_model.py:449):

        image = custom_autograd_grouped_cumprod.apply(boxsize, batch, startpoint, endpoint, mean,
                    variance_inverse, opacity, l_d, image_width, image_height)   # -> (H+1, W+1, 3)

    Inputs are the Gaussians of one camera in depth order.  Instead of expanding them into M
    splat-pixel pairs, sorting, scanning, un-sorting and scatter-adding (gs_model.py:598-624), the
    Gaussians are binned into 16x16 tiles and blended per pixel in one fused kernel
    (csrc/gcp_raster.hip); backward returns the same four gradients (gs_model.py:820).

    Differences, all documented in DESIGN.md:
      * `boxsize` and `batch` are accepted and ignored: they exist to chunk the pair list for
        memory (gs_model.py:428); nothing of size M is allocated here, so the result is always the
        single-chunk one (the reference's multi-chunk carry drops a factor);
      * dL/dl_d is the true gradient (the reference's is channel-collapsed);
      * deterministic (no atomics); the reference accumulates with index_put_(accumulate=True).
    """

    # The reference's helpers around the scans, under their own names (gs_model.py:480-594, :716-730): code written against
    # `custom_autograd_grouped_cumprod._create_alpha_brend(...)` etc. — its `_forward_batch` / `_backward_batch` — runs on
    # the HIP library as it stands.
    unique = staticmethod(unique)
    _create_rects = staticmethod(create_rects)
    _create_alpha_brend = staticmethod(create_alpha_brend)
    _mask_zero_T = staticmethod(mask_zero_T)
    _mask_tensor = staticmethod(mask_tensor)
    _sort_tensor = staticmethod(sort_tensor)
    _create_alpha_brend_min = staticmethod(create_alpha_brend_min)
    _cat_alpha_brend = staticmethod(cat_alpha_brend)
    create_grad_alphabrend_min = staticmethod(create_grad_alphabrend_min)

    @staticmethod
    def grad_cumsum(rects, grad, cutting_number=None):
        """gs_model.py:716-722, with the mask in the reference's (flipped) order: `_backward_batch` applies it as it is."""
        return grad_cumsum(rects, grad, cutting_number, mask_order="reference")

    @staticmethod
    def forward(ctx, boxsize, batch, startpoint, endpoint, mean, variance_inverse, opacity, l_d, image_width,
                image_height):
        _refuse_stale_leaves_in_capture(mean, variance_inverse, opacity, l_d)
        with torch.no_grad():
            w, h = int(image_width), int(image_height)
            bins = _raster.bin_tiles(startpoint, endpoint, w, h, capacity=_capacity)
            if bins.info is not None:  # sticky: survives the call, and is re-executed by every replay of a captured step
                flag = _sticky_flag(bins.info.device)
                torch.maximum(flag, bins.info[1:2], out=flag)
            image, t_ckpt = _raster.blend_forward(bins, startpoint, endpoint, mean, variance_inverse, opacity, l_d,
                                                  with_checkpoints=True)
        # the reference saves its inputs plus per-chunk (unique_rects, T_min, sizes) and recomputes the M-length pair
        # arrays in backward (gs_model.py:691, :786-820); here: the inputs, one transmittance per pixel and 32 list
        # entries, and the tile lists.  Every tensor goes through save_for_backward (autograd then owns its lifetime and
        # checks it for in-place changes); the node itself keeps only integers.
        ctx.bins_meta = (bins.width, bins.height, bins.n_gauss, bins.n_tile_pairs, bins.tiles_x, bins.tiles_y)
        ctx.save_for_backward(startpoint, endpoint, mean, variance_inverse, opacity, l_d, t_ckpt, bins.tile_off,
                              bins.tile_start, bins.tile_list)
        return image

    @staticmethod
    def backward(ctx, pixel_sum_grad):
        startpoint, endpoint, mean, variance_inverse, opacity, l_d, t_ckpt, tile_off, tile_start, tile_list = ctx.saved_tensors
        bins = _raster.TileBins(*ctx.bins_meta, tile_off, tile_start, tile_list)
        with torch.no_grad():
            g_mean, g_vinv, g_op, g_l = _raster.blend_backward(
                bins, startpoint, endpoint, mean, variance_inverse, opacity, l_d, t_ckpt, pixel_sum_grad
            )
        g_mean = g_mean if mean.is_floating_point() else None  # integer means carry no gradient
        return None, None, None, None, g_mean, g_vinv, g_op.reshape(opacity.shape), g_l, None, None
