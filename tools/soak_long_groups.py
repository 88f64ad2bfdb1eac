"""Soak test of the descriptor-tree path: many launches of the scans on long-group inputs, checking that every launch is
bit-identical to the first and that no tile ever ran out of patience (nothing left to the follow-up kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grouped_cumprod as gc  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    n = 40_000_003
    g = torch.Generator(device=dev).manual_seed(1)
    x = 1.0 - 1e-3 * torch.rand(n, device=dev, generator=g)
    z = torch.randn(n, device=dev, generator=g)
    cases = {"one group": torch.zeros(n, dtype=torch.int32, device=dev)}
    lens = torch.randint(5000, 60000, (n // 5000 + 2,), device=dev, generator=g)
    cases["groups of 5e3-6e4"] = torch.repeat_interleave(torch.arange(lens.numel(), device=dev, dtype=torch.int32), lens)[:n].contiguous()
    for name, key in cases.items():
        ref_f, ref_r, ref_b = (torch.empty(n, device=dev) for _ in range(3))
        gc.grouped_cumprod_forward(x, key, ref_f)
        gc.grouped_cumsum_reverse(z, key, ref_r)
        gc.grouped_cumprod_backward(x, ref_f, z, key, ref_b, torch.zeros(1, dtype=torch.int32, device=dev))
        out = torch.empty(n, device=dev)
        left = diff = 0
        for _ in range(iters):
            gc.grouped_cumprod_forward(x, key, out)
            left += gc.last_fallback_tiles(dev)
            diff += int(not torch.equal(out, ref_f))
            gc.grouped_cumsum_reverse(z, key, out)
            left += gc.last_fallback_tiles(dev)
            diff += int(not torch.equal(out, ref_r))
            gc.grouped_cumprod_backward(x, ref_f, z, key, out, torch.zeros(1, dtype=torch.int32, device=dev))
            left += gc.last_fallback_tiles(dev)
            diff += int(not torch.equal(out, ref_b))
        print(f"{name}: {3 * iters} launches of {n} elements ({(n + 4095) // 4096} tiles): {left} tiles left to the follow-up kernel, "
              f"{diff} launches that differ from the first in any bit", flush=True)


if __name__ == "__main__":
    main()
