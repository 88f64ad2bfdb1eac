"""ctypes binding of oracle/libgcp_oracle.so.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import torch

_DIR = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_DIR, "gcp_oracle.c")
_SO = os.path.join(_DIR, "libgcp_oracle.so")
_lib = None


def build(force=False):
    """gcc -O2 -ffp-contract=off: strict left-to-right fp32, no fused multiply-add."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.run(
            ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-o", _SO, _SRC], check=True
        )
    return _SO


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        for name in dir(_lib):
            pass
    return _lib


def _f32(t):
    assert t.dtype == torch.float32 and t.device.type == "cpu" and t.is_contiguous(), (t.dtype, t.device)
    return ctypes.c_void_p(t.data_ptr())


def _i32(t):
    assert t.dtype == torch.int32 and t.device.type == "cpu" and t.is_contiguous(), (t.dtype, t.device)
    return ctypes.c_void_p(t.data_ptr())


def _n(t):
    return ctypes.c_int64(t.numel())


def cumprod_forward(x, key):
    """reference: grouped_cumprod_forward.cu:17-23, sequential fp32."""
    y = torch.empty_like(x)
    _load().oracle_cumprod_forward(_f32(x), _i32(key), _f32(y), _n(key))
    return y


def cumsum_forward(x, key):
    """reference: grouped_cumsum_forward.cu:17-23, sequential fp32."""
    y = torch.empty_like(x)
    _load().oracle_cumsum_forward(_f32(x), _i32(key), _f32(y), _n(key))
    return y


def cumsum_reverse(x, key):
    """flip / grouped_cumsum_forward / flip of gs_model.py:716-722, sequential fp32."""
    y = torch.empty_like(x)
    _load().oracle_cumsum_reverse(_f32(x), _i32(key), _f32(y), _n(key))
    return y


def cumprod_backward(param, param_cumprod, grad_out, inv, inv_len):
    """reference: grouped_cumprod_backward.cu:18-29, literal O(sum L^2) loops, fp32."""
    g = torch.empty_like(param)
    _load().oracle_cumprod_backward(
        _f32(param), _f32(param_cumprod), _f32(grad_out), _i32(inv), _f32(g), _i32(inv_len), _n(param)
    )
    return g


def cumprod_backward_f64(param, param_cumprod, grad_out, inv):
    g = torch.empty(param.numel(), dtype=torch.float64)
    _load().oracle_cumprod_backward_f64(
        _f32(param), _f32(param_cumprod), _f32(grad_out), _i32(inv), ctypes.c_void_p(g.data_ptr()), _n(param)
    )
    return g


def cumprod_forward_f64(x, key):
    y = torch.empty(x.numel(), dtype=torch.float64)
    _load().oracle_cumprod_forward_f64(_f32(x), _i32(key), ctypes.c_void_p(y.data_ptr()), _n(key))
    return y


def cumsum_forward_f64(x, key):
    y = torch.empty(x.numel(), dtype=torch.float64)
    _load().oracle_cumsum_forward_f64(_f32(x), _i32(key), ctypes.c_void_p(y.data_ptr()), _n(key))
    return y


def groups_from_key(key):
    """(inv, inv_len) in the reference's convention (cuda_test.py:19-27): dense group id
    per element and exclusive end offset per group, both int32."""
    n = key.numel()
    if n == 0:
        return torch.zeros(0, dtype=torch.int32), torch.zeros(0, dtype=torch.int32)
    head = torch.ones(n, dtype=torch.bool)
    head[1:] = key[1:] != key[:-1]
    inv = (torch.cumsum(head.to(torch.int64), 0) - 1).to(torch.int32)
    starts = torch.nonzero(head).flatten()
    inv_len = torch.cat([starts[1:], torch.tensor([n])]).to(torch.int32)
    return inv, inv_len


def max_threads():
    return int(_load().oracle_max_threads())


def cumprod_forward_mt(x, inv_len, threads):
    """Same results as cumprod_forward, groups spread over `threads` OpenMP threads (bench.py's CPU baseline)."""
    y = torch.empty_like(x)
    _load().oracle_cumprod_forward_mt(_f32(x), _i32(inv_len), _f32(y), _n(inv_len), ctypes.c_int(threads))
    return y


def cumprod_backward_mt(param, param_cumprod, grad_out, inv, inv_len, threads):
    g = torch.empty_like(param)
    _load().oracle_cumprod_backward_mt(_f32(param), _f32(param_cumprod), _f32(grad_out), _i32(inv), _f32(g), _i32(inv_len),
                                       _n(param), ctypes.c_int(threads))
    return g
