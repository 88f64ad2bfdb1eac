#!/usr/bin/env python3
"""Generate tests/golden/carry_golden.npz by RUNNING the reference's own chunk-carry helpers on CPU:
`_create_alpha_brend_min` (gs_model.py:582-586), `_cat_alpha_brend` (:589-594), `create_grad_alphabrend_min` (:724-730),
alone and in the order `_forward_batch` (:606-615) and `_backward_batch` (:634-643) call them over the chunks of a
depth-chunked camera, together with `_create_alpha_brend` (:544-566) and `grad_cumsum` (:716-722).

Runs only in the build container (needs /root/reference and oracle/_ref/):
    make -C oracle ref_host && python -B tests/golden/make_carry_golden.py

Import recipe, stable-sort patch and scene generator are those of make_function_golden.py (imported from it).  Only data is
written: inputs and the reference's outputs.  No reference source is copied."""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_function_golden as mg  # noqa: E402


def main():
    gs_model = mg.import_reference()
    mg.patch_stable_sort()
    F = gs_model.custom_autograd_grouped_cumprod
    out = {}
    # ---- the helpers alone ------------------------------------------------------------------------------------------
    cases = {"m_tiny": (5, 10, 8, 2, 51), "m_small": (40, 33, 17, 4, 52), "m_mid": (260, 64, 48, 5, 53)}
    for name, (n_gauss, w, h, mh, seed) in cases.items():
        sc = mg.make_scene(n_gauss, w, h, mh, seed)
        with mg.CudaToCpu():
            rects = F._create_rects(sc["start"], sc["end"])
            g = torch.Generator().manual_seed(seed + 100)
            n = rects.size(0)
            T = torch.rand(n, generator=g).to(torch.float32)
            T[::11] = 0.0
            signed = torch.randn(n, generator=g)  # minima of negative values and of both zeros' neighbourhood
            signed[::7] = -0.0
            grad = torch.randn(n, generator=g)
            out[name + "/rects"] = rects.numpy()
            out[name + "/T"] = T.numpy()
            out[name + "/signed"] = signed.numpy()
            out[name + "/grad"] = grad.numpy()
            out[name + "/width_height"] = np.array([w, h], dtype=np.int32)
            for tag, vals in (("T", T), ("signed", signed)):
                u, m = F._create_alpha_brend_min(rects, vals)
                out[f"{name}/min_{tag}/unique_rects"] = u.numpy()
                out[f"{name}/min_{tag}/values"] = m.numpy()
            u, gmin = F.create_grad_alphabrend_min(rects, grad)
            out[name + "/grad_min/unique_rects"] = u.numpy()
            out[name + "/grad_min/values"] = gmin.numpy()
            # a list that is no longer made of whole boxes: what `rects = rects[mask]` (gs_model.py:608) leaves
            keep = torch.rand(n, generator=g) > 0.3
            u, m = F._create_alpha_brend_min(rects[keep], T[keep])
            out[name + "/masked/keep"] = keep.numpy()
            out[name + "/masked/unique_rects"] = u.numpy()
            out[name + "/masked/values"] = m.numpy()
            # int64 lists (what make_rect_points_parallel returns before the cast of :482)
            u, m = F._create_alpha_brend_min(rects.to(torch.int64), T)
            assert u.dtype == torch.int64
            out[name + "/min_T_i64/unique_rects"] = u.numpy()

    # ---- the chain over the chunks of one camera, in the reference's own order ------------------------------------------
    for name, (n_gauss, w, h, mh, seed, n_chunks) in {"chain_small": (60, 24, 18, 4, 61, 3), "chain_mid": (400, 64, 48, 6, 62, 4)}.items():
        sc = mg.make_scene(n_gauss, w, h, mh, seed)
        g = torch.Generator().manual_seed(seed + 200)
        bounds = [n_gauss * (c + 1) // n_chunks for c in range(n_chunks)]
        out[name + "/width_height"] = np.array([w, h], dtype=np.int32)
        out[name + "/chunk_ends"] = np.array(bounds, dtype=np.int64)
        out[name + "/start"] = sc["start"].numpy()
        out[name + "/end"] = sc["end"].numpy()
        with mg.CudaToCpu():
            # forward (gs_model.py:601-615)
            unique_rects, T_min = None, None
            kept_rects = []
            for c in range(n_chunks):
                s0 = bounds[c - 1] if c else 0
                rects = F._create_rects(sc["start"][s0:bounds[c]], sc["end"][s0:bounds[c]])
                anti = (1.0 - 0.9 * torch.rand(rects.size(0), generator=g)).to(torch.float32)
                anti[::19] = 0.0
                out[f"{name}/fwd{c}/anti_opacity"] = anti.numpy()
                if unique_rects is None:
                    T, mask = F._create_alpha_brend(rects, anti, flag="cumprod")
                    rects = rects[mask]
                    unique_rects, T_min = F._create_alpha_brend_min(rects, T)
                else:
                    cat_anti, cat_rects = F._cat_alpha_brend([T_min, anti], [unique_rects, rects])
                    T, mask = F._create_alpha_brend(cat_rects, cat_anti, flag="cumprod", cutting_number=len(unique_rects))
                    rects = rects[mask]
                    cat_anti, cat_rects = F._cat_alpha_brend([T_min, T], [unique_rects, rects])
                    unique_rects, T_min = F._create_alpha_brend_min(cat_rects, cat_anti)
                kept_rects.append(rects)
                out[f"{name}/fwd{c}/T"] = T.numpy()
                out[f"{name}/fwd{c}/mask"] = mask.numpy()
                out[f"{name}/fwd{c}/unique_rects"] = unique_rects.numpy()
                out[f"{name}/fwd{c}/T_min"] = T_min.numpy()
            # backward, last chunk first (gs_model.py:634-643, :799-809)
            unique_rects, grad_cumsum_0 = None, None
            for c in reversed(range(n_chunks)):
                rects = kept_rects[c]
                pixel_grad = torch.randn(rects.size(0), generator=g)
                pixel_grad[::23] = 0.0
                out[f"{name}/bwd{c}/pixel_grad"] = pixel_grad.numpy()
                if unique_rects is not None:
                    cat_grad, cat_rects = F._cat_alpha_brend([pixel_grad, grad_cumsum_0], [rects, unique_rects])
                    pixel_grad_cumsum, mask = F.grad_cumsum(cat_rects, cat_grad, len(unique_rects))
                    rects = rects[mask]
                    cat_grad, cat_rects = F._cat_alpha_brend([pixel_grad_cumsum, grad_cumsum_0], [rects, unique_rects])
                    unique_rects, grad_cumsum_0 = F.create_grad_alphabrend_min(cat_rects, cat_grad)
                else:
                    pixel_grad_cumsum, mask = F.grad_cumsum(rects, pixel_grad)
                    rects = rects[mask]
                    unique_rects, grad_cumsum_0 = F.create_grad_alphabrend_min(rects, pixel_grad_cumsum)
                out[f"{name}/bwd{c}/pixel_grad_cumsum"] = pixel_grad_cumsum.numpy()
                out[f"{name}/bwd{c}/mask"] = mask.numpy()
                out[f"{name}/bwd{c}/unique_rects"] = unique_rects.numpy()
                out[f"{name}/bwd{c}/grad_cumsum_0"] = grad_cumsum_0.numpy()
    path = os.path.join(HERE, "carry_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
