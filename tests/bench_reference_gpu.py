"""The reference's OWN kernels on this MI355X beside ours (needs oracle/_ref/grouped_cumprod_ref_gfx950.so,
built in the build container by `make -C oracle ref_gfx950` from the reference's unmodified sources:
cuda_kernel.cpp + grouped_cumprod_forward.cu / grouped_cumsum_forward.cu (rocThrust inclusive_scan_by_key)
+ grouped_cumprod_backward.cu (one thread per element, serial loop to the group end)).
Checker/baseline only — not part of the product."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ may use oracle/ (the reference build is a checker)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
import grouped_cumprod as gc  # noqa: E402
from simplegaussiansplat_tk71_amd import synthetic  # noqa: E402


def timeit(fn, iters, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    args = ap.parse_args()
    import grouped_cumprod_ref_gfx950 as ref

    dev = torch.device("cuda", 0)
    p = synthetic.make_config(args.config, seed=0, device=dev)
    m = p.n_pairs
    y, g, yr, gr = (torch.empty_like(p.x) for _ in range(4))
    print(f"{args.config}: M={m} pairs, longest group {int(p.run_len.max())}")
    rows = [
        ("cumprod forward", lambda: ref.grouped_cumprod_forward(p.x, p.key, yr), lambda: gc.grouped_cumprod_forward(p.x, p.key, y), 12, 10),
        ("cumsum forward", lambda: ref.grouped_cumsum_forward(p.x, p.key, gr), lambda: gc.grouped_cumsum_forward(p.x, p.key, g), 12, 10),
        ("cumprod backward", lambda: ref.grouped_cumprod_backward(p.x, yr, p.grad_out, p.inv, gr, p.inv_len),
         lambda: gc.grouped_cumprod_backward(p.x, y, p.grad_out, p.inv, g, p.inv_len), 20, 3),
    ]
    for name, fr, fo, bpe, iters in rows:
        tr = timeit(fr, iters)
        to = timeit(fo, max(iters, 10))
        print(f"{name:17s} reference@gfx950 {tr*1e3:12.1f} us ({bpe*m/tr/1e6:7.0f} GB/s)   this repo {to*1e3:9.1f} us ({bpe*m/to/1e6:7.0f} GB/s)   x{tr/to:.1f}")
    print("max |cumprod - reference| =", float((y - yr).abs().max()), "  max rel |backward - reference| =",
          float(((g - gr).abs() / (1 + gr.abs())).max()))


if __name__ == "__main__":
    main()
