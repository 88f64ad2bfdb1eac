"""The "pure-PyTorch torch.cumprod CPU path" BASELINE.json names.  TEST INFRASTRUCTURE ONLY.

The reference has no live pure-PyTorch scan: two attempts survive as comments
(uitility.py:369-379 needs torch_scatter; uitility.py:382-428 is a log/cumsum/exp
trick around torch.cumprod, :416).  This restates what they were for — a grouped
cumulative product / sum over runs of equal adjacent keys — with plain
torch.cumprod / torch.cumsum applied per group: groups are bucketed by length,
each bucket is padded to a dense [groups, Lmax] matrix (pad = identity) and scanned
along dim 1, so every group is scanned left to right by torch itself.
"""
import torch


def _runs(key):
    n = key.numel()
    head = torch.ones(n, dtype=torch.bool)
    head[1:] = key[1:] != key[:-1]
    starts = torch.nonzero(head).flatten()
    lens = torch.diff(torch.cat([starts, torch.tensor([n])]))
    return starts, lens


def _grouped_scan(x, key, fn, pad):
    n = key.numel()
    y = torch.empty_like(x)
    if n == 0:
        return y
    starts, lens = _runs(key)
    lo = 0
    maxlen = int(lens.max())
    while lo < maxlen:
        hi = max(2 * lo, 8) if lo else 8
        sel = (lens > lo) & (lens <= hi)
        if sel.any():
            s, l = starts[sel], lens[sel]
            width = int(l.max())
            col = torch.arange(width)
            valid = col[None, :] < l[:, None]
            idx = (s[:, None] + col[None, :]).clamp(max=n - 1)
            m = torch.where(valid, x[idx], torch.full((), pad, dtype=x.dtype))
            sc = fn(m, 1)
            y[idx[valid]] = sc[valid]
        lo = hi
    return y


def grouped_cumprod(x, key):
    return _grouped_scan(x, key, torch.cumprod, 1.0)


def grouped_cumsum(x, key):
    return _grouped_scan(x, key, torch.cumsum, 0.0)


def grouped_cumprod_backward_autograd(x, key, grad_out):
    """True VJP of grouped_cumprod by torch autograd (differs from the reference kernel
    only where x == 0, where the reference substitutes 1e-8: grouped_cumprod_backward.cu:25)."""
    xr = x.clone().requires_grad_(True)
    y = _grouped_scan_autograd(xr, key)
    (g,) = torch.autograd.grad(y, xr, grad_out)
    return g


def _grouped_scan_autograd(x, key):
    starts, lens = _runs(key)
    outs = []
    for s, l in zip(starts.tolist(), lens.tolist()):
        outs.append(torch.cumprod(x[s : s + l], 0))
    return torch.cat(outs) if outs else x.new_zeros(0)
