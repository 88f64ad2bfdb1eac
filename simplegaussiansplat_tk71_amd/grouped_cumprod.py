"""Drop-in for the reference's compiled module `grouped_cumprod`.

Same three functions, same positional arguments, same in-place-output convention
(reference: cuda_kernel/cuda_kernel.cpp:5-22; callers gs_model.py:551,553 and
cuda_test.py:23,29).  Each call is one asynchronous launch of the HIP library on
the CURRENT PyTorch stream (the reference uses the legacy default stream,
grouped_cumprod_backward.cu:56, and Thrust's blocking default policy).

Contract (reference: data_ptr<float>()/data_ptr<int>() in
grouped_cumprod_forward.cu:8-10 raise c10::Error -> RuntimeError on a dtype
mismatch; nothing else is checked there).  Here every violation raises
RuntimeError before anything is launched:
  values float32, keys / ids / offsets int32, all on the same ROCm device,
  contiguous, equal element counts.
There is no CPU path: CPU tensors raise.  The CPU statement of these ops lives in
oracle/ and is test infrastructure only.
"""
import ctypes

import torch

from . import _lib

__all__ = [
    "grouped_cumprod_forward",
    "grouped_cumsum_forward",
    "grouped_cumprod_backward",
    "grouped_cumsum_reverse",
    "grouped_cumprod_forward_indexed",
    "grouped_cumsum_forward_indexed",
    "grouped_cumsum_reverse_indexed",
    "grouped_cumprod_forward_carry",
    "grouped_cumsum_forward_carry",
    "grouped_cumsum_reverse_carry",
    "Workspace",
    "check_groups",
    "check_permutation",
    "check_group_ids",
    "set_validate_operands",
    "last_fallback_tiles",
    "last_lookback_tiles",
    "set_lookback_wait_us",
    "tile_elems",
]

class Workspace:
    """Scratch of the scans (include/grouped_cumprod_hip.h "Workspace"): a launch counter and two sets of tile descriptors,
    zeroed once and self-maintaining afterwards.  Launches that share one must be ORDERED (one stream); give every stream
    that scans concurrently its own.  Pass it as `workspace=` to any scan of this module, or let the module keep one per
    (device, stream) — see `_workspace`.  A HIP graph that captured a scan holds this buffer's address: keep the object
    alive as long as the graph (the module's own cache never drops a workspace that was used during a capture)."""

    def __init__(self, device, n_elements):
        device = torch.device(device)
        self.capacity = int(n_elements)
        need = _lib.load().gcp_workspace_bytes(self.capacity)
        # launch counter + two alternating sets of tile descriptors: must start zeroed
        self.tensor = torch.zeros(max(need, 1 << 16), dtype=torch.uint8, device=device)
        self.pinned = False  # a captured graph has baked the address in

    def fits(self, n):
        return self.capacity >= n


_workspaces = {}   # (device index, stream handle) -> Workspace, in least-recently-used order
_retired = []      # workspaces a captured graph may still point at, replaced by larger ones: never freed
_MAX_CACHED = 16   # per process; torch's stream pool hands out 32 streams per device and priority at most


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _check_tensor(t, name, dtype, device, numel=None):
    _require(isinstance(t, torch.Tensor), f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    want = "Float" if dtype is torch.float32 else "Int"
    _require(t.dtype is dtype, f"{name}: expected scalar type {want} but found {t.dtype}")
    _require(t.is_cuda, f"{name}: expected a ROCm device tensor, got device {t.device} (no CPU path)")
    _require(t.device == device, f"{name}: on {t.device}, expected {device}")
    _require(t.is_contiguous(), f"{name}: must be contiguous")
    if numel is not None:
        _require(t.numel() == numel, f"{name}: {t.numel()} elements, expected {numel}")


def _no_alias(out, out_name, *inputs):
    """The scans re-read raw inputs of the NEIGHBOURING tile (look-back) while other blocks already store their
    outputs, so an output that shares bytes with an input races between blocks.  Only the exactly-in-place call of the
    forward scans (out is x, as thrust::inclusive_scan_by_key allows, grouped_cumprod_forward.cu:17-23) has a mode of its
    own; everything else that overlaps is rejected loudly instead of returning nondeterministic values."""
    o0 = out.data_ptr()
    o1 = o0 + out.numel() * out.element_size()
    for t, name in inputs:
        t0 = t.data_ptr()
        t1 = t0 + t.numel() * t.element_size()
        _require(not (o0 < t1 and t0 < o1),
                 f"{out_name}: overlaps {name} in memory; in-place / aliased outputs are not supported "
                 "(allocate a separate output tensor)")


def _workspace(device, stream_handle, n):
    """The module's own scratch of the current (device, stream): reused across calls, grown geometrically, at most
    _MAX_CACHED of them alive (least recently used first out).  Keyed by the HIP stream HANDLE: torch's streams come
    from a per-device pool and are never destroyed, so one handle is one stream for the life of the process — two
    `torch.cuda.Stream` objects that share a handle ARE the same stream, and their launches are ordered.  (An external
    stream that is destroyed and whose handle is later reused has drained by then: hipStreamDestroy releases the handle
    only when its work has completed.)  A workspace in use while the stream is capturing is pinned: the graph replays
    with its address."""
    key = (device.index, stream_handle)
    ws = _workspaces.pop(key, None)
    capturing = torch.cuda.is_current_stream_capturing()
    if ws is None or not ws.fits(n):
        if ws is not None and ws.pinned:
            _retired.append(ws)
        ws = Workspace(device, n + n // 2)
    ws.pinned = ws.pinned or capturing
    _workspaces[key] = ws  # most recently used last
    while len(_workspaces) > _MAX_CACHED:
        for k, w in _workspaces.items():
            if not w.pinned and k != key:
                del _workspaces[k]
                break
        else:
            break
    return ws.tensor


def _launch(fn_name, device, n, ptrs_before_n, ptrs_after_n=(), workspace=None):
    lib = _lib.load()
    if device.index != torch.cuda.current_device():
        with torch.cuda.device(device):
            return _launch(fn_name, device, n, ptrs_before_n, ptrs_after_n, workspace)
    stream = torch.cuda.current_stream(device).cuda_stream
    if workspace is not None:
        _require(isinstance(workspace, Workspace), "workspace: expected a grouped_cumprod.Workspace")
        _require(workspace.tensor.device == device, f"workspace: on {workspace.tensor.device}, expected {device}")
        _require(workspace.fits(n), f"workspace: sized for {workspace.capacity} elements, this call has {n}")
        workspace.pinned = workspace.pinned or torch.cuda.is_current_stream_capturing()
        ws = workspace.tensor
    else:
        ws = _workspace(device, stream, n)
    status = getattr(lib, fn_name)(*ptrs_before_n, n, *ptrs_after_n, ws.data_ptr(), ws.numel(), stream)
    if status:
        _lib.check(status, fn_name)


def _forward(fn_name, x, key, y, workspace=None):
    _require(isinstance(key, torch.Tensor), "pixel_index: expected a torch.Tensor")
    n = key.numel()  # reference: n = pixel_index.numel() (grouped_cumprod_forward.cu:13)
    dev = key.device
    _check_tensor(key, "pixel_index", torch.int32, dev)
    _check_tensor(x, "unti_opacity", torch.float32, dev, n)
    _check_tensor(y, "out", torch.float32, dev, n)
    if n == 0:
        return
    # exactly in place (out is unti_opacity) is legal, as with the reference's Thrust scans (grouped_cumprod_forward.cu:17-23):
    # the library then takes every carry from its tile descriptors instead of re-reading the neighbouring tile's inputs
    # (about two thirds of the out-of-place rate); any partial overlap is refused
    inplace = y.data_ptr() == x.data_ptr()
    _no_alias(y, "out", *(() if inplace else ((x, "unti_opacity"),)), (key, "pixel_index"))
    _launch(fn_name, dev, n, (x.data_ptr(), key.data_ptr(), y.data_ptr()), workspace=workspace)


def grouped_cumprod_forward(unti_opacity, pixel_index, out, *, workspace=None):
    """out[i] = prod of unti_opacity over the run of equal adjacent pixel_index up to i.

    reference: cuda_kernel/grouped_cumprod_forward.cu:6-24.
    """
    _forward("gcp_cumprod_forward", unti_opacity, pixel_index, out, workspace)


def grouped_cumsum_forward(unti_opacity, pixel_index, out, *, workspace=None):
    """Same with a running sum.  reference: cuda_kernel/grouped_cumsum_forward.cu:6-24."""
    _forward("gcp_cumsum_forward", unti_opacity, pixel_index, out, workspace)


def grouped_cumsum_reverse(x, key, out, *, workspace=None):
    """Suffix sums inside each run: flip -> grouped_cumsum_forward -> flip of the
    reference (gs_model.py:716-722) in one pass.  Not in the reference module."""
    _forward("gcp_cumsum_reverse", x, key, out, workspace)


def _forward_indexed(fn_name, x, sorted_key, index, y):
    _require(isinstance(sorted_key, torch.Tensor), "sorted_key: expected a torch.Tensor")
    n = sorted_key.numel()
    dev = sorted_key.device
    _check_tensor(sorted_key, "sorted_key", torch.int32, dev)
    _check_tensor(index, "index", torch.int32, dev, n)
    _check_tensor(x, "x", torch.float32, dev, n)
    _check_tensor(y, "out", torch.float32, dev, n)
    if n == 0:
        return
    _no_alias(y, "out", (x, "x"), (sorted_key, "sorted_key"), (index, "index"))
    _launch(fn_name, dev, n, (x.data_ptr(), sorted_key.data_ptr(), index.data_ptr(), y.data_ptr()))


def grouped_cumprod_forward_indexed(x, sorted_key, index, out):
    """out[index[i]] = running product, inside runs of equal adjacent sorted_key, of x[index[i]]: gather, grouped scan and
    un-sort of `_create_alpha_brend` (gs_model.py:548-555) in one pass; `x` / `out` in the original pair order,
    `sorted_key` / `index` from the stable sort of the pixel keys.  `index` must be a permutation: not checked unless
    `set_validate_operands(True)` (see `check_permutation`).  Not in the reference module."""
    _forward_indexed("gcp_cumprod_forward_indexed", x, sorted_key, index, out)


def grouped_cumsum_forward_indexed(x, sorted_key, index, out):
    _forward_indexed("gcp_cumsum_forward_indexed", x, sorted_key, index, out)


def grouped_cumsum_reverse_indexed(x, sorted_key, index, out):
    """Suffix-sum form (grad_cumsum, gs_model.py:716-722, without the flips)."""
    _forward_indexed("gcp_cumsum_reverse_indexed", x, sorted_key, index, out)


def _forward_carry(fn_name, x, inv, carry, y):
    _require(isinstance(inv, torch.Tensor), "inv: expected a torch.Tensor")
    n = inv.numel()
    dev = inv.device
    _check_tensor(inv, "inv", torch.int32, dev)
    _check_tensor(x, "x", torch.float32, dev, n)
    _check_tensor(y, "out", torch.float32, dev, n)
    _check_tensor(carry, "carry", torch.float32, dev)
    if n == 0:
        return
    _require(carry.numel() > 0, "carry: empty for a non-empty input")
    _no_alias(y, "out", (x, "x"), (inv, "inv"), (carry, "carry"))
    _launch(fn_name, dev, n, (x.data_ptr(), inv.data_ptr(), carry.data_ptr(), y.data_ptr()), (carry.numel(),))


def grouped_cumprod_forward_carry(x, inv, carry, out):
    """out[i] = carry[inv[i]] * prod of x over the group up to i (`inv` = dense group id).  Exact version
    of the reference's chunk carry (gs_model.py:606-615).  Not in the reference module."""
    _forward_carry("gcp_cumprod_forward_carry", x, inv, carry, out)


def grouped_cumsum_forward_carry(x, inv, carry, out):
    _forward_carry("gcp_cumsum_forward_carry", x, inv, carry, out)


def grouped_cumsum_reverse_carry(x, inv, carry, out):
    """Suffix sums that start from carry[inv[i]] (the suffix entering from a deeper chunk,
    gs_model.py:634-643)."""
    _forward_carry("gcp_cumsum_reverse_carry", x, inv, carry, out)


def grouped_cumprod_backward(param, param_cumprod, grad_out, inv, grad_in, inv_len, *, workspace=None):
    """grad_in[j] = sum_{i=j}^{inv_len[inv[j]]-1} grad_out[i] * param_cumprod[i] / p'_j.

    reference: cuda_kernel/grouped_cumprod_backward.cu:9-65 (p'_j = param[j] or 1e-8
    when it is 0, :25).  `inv` = dense group id per element, `inv_len` = exclusive
    end offset per group (cuda_test.py:27).
    """
    _require(isinstance(param, torch.Tensor), "param: expected a torch.Tensor")
    n = param.numel()  # reference: n = param.numel() (grouped_cumprod_backward.cu:52)
    dev = param.device
    _check_tensor(param, "param", torch.float32, dev)
    _check_tensor(param_cumprod, "param_cumprod", torch.float32, dev, n)
    _check_tensor(grad_out, "grad_out", torch.float32, dev, n)
    _check_tensor(inv, "inv", torch.int32, dev, n)
    _check_tensor(grad_in, "grad_in", torch.float32, dev, n)
    _check_tensor(inv_len, "inv_len", torch.int32, dev)
    if n == 0:
        return
    _require(inv_len.numel() > 0, "inv_len: empty for a non-empty input")
    _no_alias(grad_in, "grad_in", (param, "param"), (param_cumprod, "param_cumprod"), (grad_out, "grad_out"), (inv, "inv"),
              (inv_len, "inv_len"))
    _launch(
        "gcp_cumprod_backward",
        dev,
        n,
        (param.data_ptr(), param_cumprod.data_ptr(), grad_out.data_ptr(), inv.data_ptr(), grad_in.data_ptr(),
         inv_len.data_ptr()),
        (inv_len.numel(),),
        workspace=workspace,
    )


def check_groups(inv, inv_len):
    """Number of places where (inv, inv_len) is not a consistent dense partition (debug aid, synchronises)."""
    dev = inv.device
    n = inv.numel()
    _check_tensor(inv, "inv", torch.int32, dev)
    _check_tensor(inv_len, "inv_len", torch.int32, dev)
    lib = _lib.load()
    bad = ctypes.c_int64(0)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        status = lib.gcp_check_groups(inv.data_ptr(), inv_len.data_ptr(), n, inv_len.numel(), ctypes.byref(bad), stream)
    _lib.check(status, "gcp_check_groups")
    return bad.value


def check_permutation(index):
    """Number of entries of `index` (int32[n]) that are outside [0, n) or repeat an earlier value: 0 iff it is a
    permutation — what the indexed scans read and write through, unchecked (debug aid, synchronises)."""
    dev = index.device
    _check_tensor(index, "index", torch.int32, dev)
    bad = ctypes.c_int64(0)
    with torch.cuda.device(dev):
        status = _lib.load().gcp_check_permutation(index.data_ptr(), index.numel(), ctypes.byref(bad),
                                                   torch.cuda.current_stream(dev).cuda_stream)
    if status not in (0, 1):
        _lib.check(status, "gcp_check_permutation")
    return bad.value


def check_group_ids(inv, n_groups):
    """Number of entries of `inv` (int32[n]) outside [0, n_groups): the carry forms index `carry` with them, unchecked
    (debug aid, synchronises)."""
    dev = inv.device
    _check_tensor(inv, "inv", torch.int32, dev)
    _require(int(n_groups) > 0 or inv.numel() == 0, "n_groups: must be positive for a non-empty input")
    bad = ctypes.c_int64(0)
    with torch.cuda.device(dev):
        status = _lib.load().gcp_check_group_ids(inv.data_ptr(), inv.numel(), int(n_groups), ctypes.byref(bad),
                                                 torch.cuda.current_stream(dev).cuda_stream)
    if status not in (0, 1):
        _lib.check(status, "gcp_check_group_ids")
    return bad.value


def set_validate_operands(on):
    """Process-wide, default off (environment GCP_VALIDATE_OPERANDS=1 turns it on): the indexed scans check that `index`
    is a permutation and the carry forms that `inv` lies in [0, len(carry)) BEFORE they launch, and raise RuntimeError
    ("invalid argument") instead of reading or writing out of bounds — one extra pass and a host synchronisation per
    call."""
    _lib.check(_lib.load().gcp_set_validate_operands(1 if on else 0), "gcp_set_validate_operands")


def _last_stat(fn_name, device):
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    lib = _lib.load()
    out = ctypes.c_int64(0)
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        hit = _workspaces.get((device.index, stream))
        if hit is None:
            return 0
        status = getattr(lib, fn_name)(hit.tensor.data_ptr(), stream, ctypes.byref(out))
    _lib.check(status, fn_name)
    return out.value


def last_fallback_tiles(device=None):
    """Tiles of the most recent scan on the current stream that gave up waiting for another tile's descriptor and
    were finished by the follow-up launch (synchronises).  0 in practice."""
    return _last_stat("gcp_last_fallback_tiles", device)


def last_lookback_tiles(device=None):
    """Tiles of the most recent scan on the current stream whose group started more than one tile back and that
    resolved their carry through the in-kernel descriptor look-back (synchronises)."""
    return _last_stat("gcp_last_lookback_tiles", device)


def set_lookback_wait_us(us):
    """Longest wait for another tile's descriptor (default 200 us); negative = two-pass behaviour (tests, A/B timing)."""
    _lib.check(_lib.load().gcp_set_lookback_wait_us(int(us)), "gcp_set_lookback_wait_us")


def tile_elems():
    return _lib.load().gcp_tile_elems()
