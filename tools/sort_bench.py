"""Native stable (key, index) sort vs torch.sort(stable=True) on the reference's pixel keys (cfg2 / cfg3 sized)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplegaussiansplat_tk71_amd import raster, synthetic  # noqa: E402


def timeit(fn, iters=5, warmup=2):
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
    for cfg in ("cfg2", "cfg3"):
        sc = synthetic.make_scene_config(cfg, seed=0, device=dev)
        rects = raster.expand_rects(sc["start"], sc["end"], sc["width"], sc["height"])  # Gaussian-major, as the reference sorts it
        key = (rects[:, 1] * 10000 + rects[:, 0]).contiguous()
        bits = int(key.max().item()).bit_length()
        t_torch = timeit(lambda: torch.sort(key, stable=True))
        t_ours = timeit(lambda: raster.stable_sort_keys(key, key_bits=bits))
        a, ai = torch.sort(key, stable=True)
        b, bi = raster.stable_sort_keys(key, key_bits=bits)
        ok = torch.equal(a, b) and torch.equal(ai, bi.long())
        print(f"{cfg}: M={key.numel()} keys ({bits} bits)  torch.sort(stable) {t_torch:8.3f} ms   native {t_ours:8.3f} ms   identical={ok}")
        import cuda_kernel as ck

        anti = 1.0 - 0.9 * torch.rand(key.numel(), device=dev)
        t_a5 = timeit(lambda: ck.create_alpha_brend(rects, anti, "cumprod"), iters=3, warmup=1)
        t_a5b = timeit(lambda: ck.create_alpha_brend_boxes(sc["start"], sc["end"], anti, sc["width"], sc["height"], "cumprod"), iters=3, warmup=1)
        print(f"      create_alpha_brend(rects) {t_a5:8.3f} ms   create_alpha_brend_boxes {t_a5b:8.3f} ms")


if __name__ == "__main__":
    main()
