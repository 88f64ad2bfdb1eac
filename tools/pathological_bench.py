"""Worst-case inputs for the scan: groups far longer than the look-back window (fallback kernel active)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grouped_cumprod as gc  # noqa: E402


def timeit(fn, iters=10, warmup=3):
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
    dev = torch.device("cuda", 0)
    n = 166_000_000
    g = torch.Generator(device=dev).manual_seed(0)
    x = 1.0 - 1e-3 * torch.rand(n, device=dev, generator=g)
    go = torch.randn(n, device=dev, generator=g)
    y, gi = torch.empty_like(x), torch.empty_like(x)
    cases = {}
    cases["one group = whole array"] = torch.zeros(n, dtype=torch.int32, device=dev)
    for lo, hi in ((5000, 13000), (20000, 60000), (300000, 900000)):
        lens = torch.randint(lo, hi, (n // lo + 2,), device=dev, generator=g)
        ids = torch.arange(lens.numel(), device=dev, dtype=torch.int32)
        cases[f"groups of {lo}-{hi}"] = torch.repeat_interleave(ids, lens)[:n].contiguous()
    lens = torch.full((n // 80 + 2,), 80, device=dev)
    cases["groups of 80 (reference point)"] = torch.repeat_interleave(torch.arange(lens.numel(), device=dev, dtype=torch.int32), lens)[:n].contiguous()
    only = sys.argv[1:]  # optional substrings selecting cases (a --pmc run wants one case per process)
    for mode, wait in (("descriptor walk", 200), ("two-pass (walk off)", -1)):
        gc.set_lookback_wait_us(wait)
        print(f"--- {mode}")
        for name, key in cases.items():
            if only and not any(o in name for o in only):
                continue
            inv_len = torch.zeros(1, dtype=torch.int32, device=dev)
            tf = timeit(lambda: gc.grouped_cumprod_forward(x, key, y))
            fb, lb = gc.last_fallback_tiles(dev), gc.last_lookback_tiles(dev)
            tb = timeit(lambda: gc.grouped_cumprod_backward(x, y, go, key, gi, inv_len))
            print(f"{name:32s} fwd {tf*1e3:8.1f} us ({12*n/tf/1e6:6.0f} GB/s = {12*n/tf/1e6/80:4.1f} %)  bwd {tb*1e3:8.1f} us "
                  f"({20*n/tb/1e6:6.0f} GB/s = {20*n/tb/1e6/80:4.1f} %)  tiles: walked {lb}, left to the follow-up kernel {fb}", flush=True)
    gc.set_lookback_wait_us(200)
    if not only or any("in place" in o for o in only):
        # exactly in place (out is x): every continuing tile takes its carry from the descriptor tree
        key = cases["groups of 80 (reference point)"]
        xs = torch.randn(n, device=dev, generator=g)
        buf = xs.clone()
        t_in = timeit(lambda: gc.grouped_cumsum_forward(buf, key, buf))   # (a running sum of running sums: values irrelevant here)
        t_out = timeit(lambda: gc.grouped_cumsum_forward(xs, key, y))
        print(f"--- in place\ngroups of 80, cumsum: in place {t_in*1e3:8.1f} us ({12*n/t_in/1e6/80:4.1f} %)   out of place {t_out*1e3:8.1f} us "
              f"({12*n/t_out/1e6/80:4.1f} %)", flush=True)


if __name__ == "__main__":
    main()
