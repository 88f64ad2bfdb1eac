#!/usr/bin/env python3
"""Generate tests/golden/wrappers_golden.npz by RUNNING the reference's own `_create_alpha_brend` (gs_model.py:544-566) and
`grad_cumsum` (:716-722) on CPU, for the branches tests/golden/function_golden.npz does not hold: flag="cumsum", and
`cutting_number` (the carry rows of the reference's chunked calls, :557-559) with both flags and through grad_cumsum.

Runs only in the build container (needs /root/reference and oracle/_ref/):
    make -C oracle ref_host && python -B tests/golden/make_wrappers_golden.py

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
    cases = {"w_tiny": (5, 10, 8, 2, 41, (None, 1, 7)), "w_small": (40, 33, 17, 4, 42, (None, 5, 64)),
             "w_mid": (220, 64, 48, 5, 43, (None, 3, 1001))}
    for name, (n_gauss, w, h, mh, seed, cuts) in cases.items():
        sc = mg.make_scene(n_gauss, w, h, mh, seed)
        with mg.CudaToCpu():
            rects = F._create_rects(sc["start"], sc["end"])
            g = torch.Generator().manual_seed(seed + 100)
            anti = (1.0 - 0.95 * torch.rand(rects.size(0), generator=g)).to(torch.float32)
            anti[::13] = 0.0
            grad = torch.randn(rects.size(0), generator=g)
            grad[::5] = 0.0
            out[name + "/rects"] = rects.numpy()
            out[name + "/anti_opacity"] = anti.numpy()
            out[name + "/grad"] = grad.numpy()
            out[name + "/width_height"] = np.array([w, h], dtype=np.int32)
            out[name + "/cuts"] = np.array([-1 if c is None else c for c in cuts], dtype=np.int64)
            for c in cuts:
                if c is not None and c >= rects.size(0):
                    continue
                tag = "none" if c is None else str(c)
                for flag in ("cumprod", "cumsum"):
                    v, m = F._create_alpha_brend(rects, anti, flag=flag, cutting_number=c)
                    out[f"{name}/{flag}_{tag}/values"] = v.numpy()
                    out[f"{name}/{flag}_{tag}/mask"] = m.numpy()
                s, sm = F.grad_cumsum(rects, grad, cutting_number=c)
                out[f"{name}/grad_cumsum_{tag}/values"] = s.numpy()
                out[f"{name}/grad_cumsum_{tag}/mask_flipped"] = sm.numpy()  # the reference leaves this mask in flipped order
    path = os.path.join(HERE, "wrappers_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
