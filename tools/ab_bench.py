"""A/B micro-benchmark of several builds of the HIP library in ONE process, interleaved rounds
(cdna_hip_programming.md §5.4 rule 24).  Calls the C ABI directly through ctypes.

  python tools/ab_bench.py --config cfg3 --rounds 7 --iters 10 variants/a.so variants/b.so
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplegaussiansplat_tk71_amd import synthetic  # noqa: E402

V = ctypes.c_void_p


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name in ("gcp_cumprod_forward", "gcp_cumsum_forward", "gcp_cumsum_reverse"):
        getattr(lib, name).argtypes = [V, V, V, ctypes.c_int64, V, ctypes.c_size_t, V]
    lib.gcp_cumprod_backward.argtypes = [V, V, V, V, V, V, ctypes.c_int64, ctypes.c_int64, V, ctypes.c_size_t, V]
    lib.gcp_workspace_bytes.restype = ctypes.c_size_t
    lib.gcp_workspace_bytes.argtypes = [ctypes.c_int64]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    p = synthetic.make_config(args.config, seed=0, device=dev)
    m = p.n_pairs
    y = torch.empty_like(p.x)
    g = torch.empty_like(p.x)
    libs = [(os.path.basename(q), load(q)) for q in args.libs]
    ws = torch.zeros(max(l.gcp_workspace_bytes(m) for _, l in libs) + 4096, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def fwd(lib):
        assert lib.gcp_cumprod_forward(p.x.data_ptr(), p.key.data_ptr(), y.data_ptr(), m, ws.data_ptr(), ws.numel(), stream) == 0

    def bwd(lib):
        assert lib.gcp_cumprod_backward(p.x.data_ptr(), y.data_ptr(), p.grad_out.data_ptr(), p.inv.data_ptr(), g.data_ptr(),
                                        p.inv_len.data_ptr(), m, p.n_groups, ws.data_ptr(), ws.numel(), stream) == 0

    res = {(n, k): [] for n, _ in libs for k in ("fwd", "bwd")}
    for _, lib in libs:
        for _ in range(3):
            fwd(lib), bwd(lib)
    torch.cuda.synchronize()
    for r in range(args.rounds):
        for name, lib in libs:
            for kind, fn in (("fwd", fwd), ("bwd", bwd)):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fn(lib)
                a.record()
                for _ in range(args.iters):
                    fn(lib)
                b.record()
                torch.cuda.synchronize()
                res[(name, kind)].append(a.elapsed_time(b) / args.iters * 1e3)
    print(f"{args.config}: M={m}")
    for name, _ in libs:
        f, b = sorted(res[(name, "fwd")]), sorted(res[(name, "bwd")])
        fm, bm = f[len(f) // 2], b[len(b) // 2]
        print(f"{name:28s} fwd med {fm:7.1f} min {f[0]:7.1f} us ({12*m/fm/1e3:6.0f} GB/s) | bwd med {bm:7.1f} min {b[0]:7.1f} us "
              f"({20*m/bm/1e3:6.0f} GB/s) | fwd+bwd {32*m/(fm+bm)/1e3:6.0f} GB/s = {32*m/(fm+bm)/1e3/80:.1f}%")


if __name__ == "__main__":
    main()
