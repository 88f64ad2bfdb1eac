"""Rows a5 / a6 / f3 at the size of BASELINE configs[4] on ONE GPU (3840 x 2160, 5e6 Gaussians, ~8.3e8 pairs): the default route against
the general sort route (two independent implementations), the chunk carry against torch's scatter-amin.  A one-off check of the 32-bit
index arithmetic at the largest list one GPU is asked to take; prints one JSON line.

  python tools/cfg5_wrappers_check.py
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_kernel as ck  # noqa: E402
from simplegaussiansplat_tk71_amd import synthetic  # noqa: E402
from tools.wrapper_bench import timeit  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    t0 = time.time()
    sc, rects, anti, grad = synthetic.make_scene_pairs("cfg5", seed=1, device=dev)
    m = rects.size(0)
    w, h = sc["width"], sc["height"]
    anti[::1000003] = 0.0  # a few opaque pairs: something is dropped, the compaction pass runs
    out = {"workload": "cfg5", "pairs": m, "gaussians": int(sc["start"].size(0)), "setup_s": round(time.time() - t0, 1)}
    def compare(name, got_a, got_b, tol):
        """two routes: masks may differ where a value is exactly 0 in ONE association only (products at the edge of underflow,
        sums that cancel: a handful per 1e9 pairs); values compared on the pairs both keep"""
        (va, ma), (vb, mb) = got_a, got_b
        both = ma & mb
        fa = va[(torch.cumsum(ma, 0) - 1).clamp_(min=0)][both]
        fb = vb[(torch.cumsum(mb, 0) - 1).clamp_(min=0)][both]
        out[name] = {"kept_walk": int(ma.sum()), "kept_sort_route": int(mb.sum()), "mask_differences": int((ma != mb).sum()),
                     "max_abs_difference_on_common_pairs": float((fa - fb).abs().max())}
        return out[name]["mask_differences"] <= 1e-6 * m and out[name]["max_abs_difference_on_common_pairs"] <= tol

    ok = compare("create_alpha_brend", ck.create_alpha_brend(rects, anti, "cumprod", route="boxes"),
                 ck.create_alpha_brend(rects, anti, "cumprod", image_size=(w, h), route="sort"), 1e-5)
    ok &= compare("grad_cumsum", ck.grad_cumsum(rects, grad, route="boxes"), ck.grad_cumsum(rects, grad, image_size=(w, h), route="sort"), 2e-3)
    out["create_alpha_brend_ms"] = timeit(lambda: ck.create_alpha_brend(rects, anti, "cumprod"), 3, 1)
    out["grad_cumsum_ms"] = timeit(lambda: ck.grad_cumsum(rects, grad), 3, 1)
    T, mask = ck.create_alpha_brend(rects, anti, "cumprod")
    if T.numel() != m:  # pairs were dropped (opaque ones, products that underflow in the deepest pixels): the list is thinned with
        rects = rects[mask]  # them, as gs_model.py:608 does — no longer a list of whole boxes
        grad = grad[mask]
    del mask
    u, t_min = ck.create_alpha_brend_min(rects, T, image_size=(w, h))
    key = rects[:, 0].long() * (h + 1) + rects[:, 1].long()
    tab = torch.full(((w + 1) * (h + 1),), float("inf"), device=dev).scatter_reduce_(0, key, T, reduce="amin")
    covered = torch.nonzero(torch.isfinite(tab)).flatten()
    out["distinct_pixels"] = int(u.size(0))
    out["min_pixels_equal"] = bool(u.size(0) == covered.numel() and torch.equal(u[:, 0].long() * (h + 1) + u[:, 1].long(), covered))
    out["min_values_equal"] = bool(torch.equal(t_min, tab[covered]))
    del tab, key
    out["create_alpha_brend_min_ms"] = timeit(lambda: ck.create_alpha_brend_min(rects, T, image_size=(w, h)), 3, 1)
    out["create_grad_alphabrend_min_ms"] = timeit(lambda: ck.create_grad_alphabrend_min(rects, grad, image_size=(w, h)), 3, 1)
    _, g_first = ck.create_grad_alphabrend_min(rects, grad, image_size=(w, h))
    out["first_pair_rows"] = int(g_first.numel())
    print(json.dumps(out), flush=True)
    ok &= out["min_pixels_equal"] and out["min_values_equal"]
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
