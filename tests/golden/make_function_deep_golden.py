#!/usr/bin/env python3
"""Generate tests/golden/function_deep_golden.npz: the reference's own `custom_autograd_grouped_cumprod`
(gs_model.py:477-820) RUN on CPU on scenes whose pixel lists are hundreds of layers deep — what pins the blend
backward's behaviour at depth to the reference's reverse scan (`grad_cumsum`, gs_model.py:716-722) rather than to the
builder's dense oracle alone.  Same import recipe as make_function_golden.py (run in the build container):

    make -C oracle ref_host && python -B tests/golden/make_function_deep_golden.py

Scenes: every Gaussian's box covers the whole image, wide kernels (every pixel sees every layer), opacities chosen so
that the transmittance runs from 1 down to ~1e-9 (300 layers) / ~1e-20 (700 layers) along every pixel's list.
Only data is written: inputs and the reference's outputs."""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_function_golden as mfg  # noqa: E402  (the import recipe and run_function)


def deep_scene(n_layers, width, height, op_lo, op_hi, seed):
    g = torch.Generator().manual_seed(seed)
    n = n_layers
    start = torch.zeros(n, 2, dtype=torch.int32)
    end = torch.tensor([[width, height]], dtype=torch.int32).repeat(n, 1)
    mean = torch.stack([torch.randint(2, width - 1, (n,), generator=g), torch.randint(2, height - 1, (n,), generator=g)], 1).to(torch.int32)
    s = 5.0 + 9.0 * torch.rand(n, generator=g)
    vinv = torch.zeros(n, 2, 2)
    vinv[:, 0, 0] = 1.0 / (s * s)
    vinv[:, 1, 1] = 1.0 / (s * s)
    opacity = (op_lo + (op_hi - op_lo) * torch.rand(n, 1, generator=g)).to(torch.float32)
    l_d = (0.1 + 0.9 * torch.rand(n, 3, generator=g)).to(torch.float32)
    wimg = torch.randn(height + 1, width + 1, 3, generator=g)
    boxsize = torch.prod((end - start + 1).to(torch.int64), dim=1)
    return dict(boxsize=boxsize, start=start, end=end, mean=mean, vinv=vinv.contiguous(), opacity=opacity, l_d=l_d, wimg=wimg)


def main():
    gs_model = mfg.import_reference()
    mfg.patch_stable_sort()
    F = gs_model.custom_autograd_grouped_cumprod
    _bwd = F.backward

    def backward_in_mode(ctx, g):
        with mfg.CudaToCpu():
            return _bwd(ctx, g)

    F.backward = staticmethod(backward_in_mode)
    out = {}
    for name, (n, w, h, lo, hi, seed) in {"deep_300": (300, 19, 17, 0.02, 0.2, 41), "deep_700": (700, 17, 15, 0.02, 0.2, 42)}.items():
        sc = deep_scene(n, w, h, lo, hi, seed)
        img, gv, go, gl = mfg.run_function(F, sc, w, h, [n])
        for k in ("boxsize", "start", "end", "mean", "vinv", "opacity", "l_d", "wimg"):
            out[f"{name}/{k}"] = sc[k].numpy()
        out[name + "/width_height"] = np.array([w, h], dtype=np.int32)
        out[name + "/image"] = img.numpy()
        out[name + "/grad_vinv"] = gv.numpy()
        out[name + "/grad_opacity"] = go.numpy()
        print(name, "image max", float(img.max()), "|grad_opacity| range", float(go.abs().min()), float(go.abs().max()))
    path = os.path.join(HERE, "function_deep_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
