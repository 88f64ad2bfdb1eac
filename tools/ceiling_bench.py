"""How close are the scans to an ideal streaming kernel of the same traffic mix?  One process, interleaved
rounds: tools/stream_ceiling.hip (no dependencies, same bytes per element) vs the product's scan kernels."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grouped_cumprod as gc  # noqa: E402
from simplegaussiansplat_tk71_amd import synthetic  # noqa: E402

V = ctypes.c_void_p


def main():
    dev = torch.device("cuda", 0)
    so = os.path.join(ROOT, "tools", "libstream_ceiling.so")
    if not os.path.exists(so):
        import subprocess

        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so,
                        os.path.join(ROOT, "tools", "stream_ceiling.hip")], check=True)
    lib = ctypes.CDLL(so)
    lib.ceiling12.argtypes = [V, V, V, ctypes.c_longlong, ctypes.c_int, V]
    lib.ceiling20.argtypes = [V, V, V, V, V, ctypes.c_longlong, ctypes.c_int, V]
    p = synthetic.make_config("cfg3", seed=0, device=dev)
    m = p.n_pairs
    y, g = torch.empty_like(p.x), torch.empty_like(p.x)
    st = torch.cuda.current_stream().cuda_stream
    ops = {
        "ideal 12B (2R+1W)": (lambda: lib.ceiling12(p.x.data_ptr(), p.key.data_ptr(), g.data_ptr(), m, 0, st), 12),
        "ideal 12B nt-store": (lambda: lib.ceiling12(p.x.data_ptr(), p.key.data_ptr(), g.data_ptr(), m, 1, st), 12),
        "ideal 12B nt-ld+st": (lambda: lib.ceiling12(p.x.data_ptr(), p.key.data_ptr(), g.data_ptr(), m, 2, st), 12),
        "cumprod_fwd": (lambda: gc.grouped_cumprod_forward(p.x, p.key, y), 12),
        "ideal 20B (4R+1W)": (lambda: lib.ceiling20(p.x.data_ptr(), y.data_ptr(), p.grad_out.data_ptr(), p.inv.data_ptr(), g.data_ptr(), m, 0, st), 20),
        "ideal 20B nt-store": (lambda: lib.ceiling20(p.x.data_ptr(), y.data_ptr(), p.grad_out.data_ptr(), p.inv.data_ptr(), g.data_ptr(), m, 1, st), 20),
        "ideal 20B nt-ld+st": (lambda: lib.ceiling20(p.x.data_ptr(), y.data_ptr(), p.grad_out.data_ptr(), p.inv.data_ptr(), g.data_ptr(), m, 2, st), 20),
        "cumprod_bwd": (lambda: gc.grouped_cumprod_backward(p.x, y, p.grad_out, p.inv, g, p.inv_len), 20),
    }
    res = {k: [] for k in ops}
    for _ in range(3):
        for fn, _b in ops.values():
            fn()
    torch.cuda.synchronize()
    for r in range(7):
        for name, (fn, _b) in ops.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn(); a.record()
            for _ in range(10):
                fn()
            b.record(); torch.cuda.synchronize()
            res[name].append(a.elapsed_time(b) / 10 * 1e3)
    for name, (_f, bpe) in ops.items():
        t = sorted(res[name])
        med = t[len(t) // 2]
        print(f"{name:22s} median {med:7.1f} us  min {t[0]:7.1f} us  {bpe*m/med/1e3:6.0f} GB/s = {bpe*m/med/1e3/80:.1f}% of 8 TB/s")


if __name__ == "__main__":
    main()
