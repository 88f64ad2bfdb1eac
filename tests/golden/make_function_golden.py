#!/usr/bin/env python3
"""Generate tests/golden/function_golden.npz by RUNNING the reference's own Python path on CPU.

Runs only in the build container (needs /root/reference and oracle/_ref/):
    make -C oracle ref_host && python -B tests/golden/make_function_golden.py

What runs is the reference's `custom_autograd_grouped_cumprod` (gs_model.py:477-820), its
`_create_alpha_brend` (:544-566) and `grad_cumsum` (:716-722), imported from /root/reference,
with the reference's own forward kernels compiled for the host (oracle/_ref) registered as
the module `grouped_cumprod` it imports (gs_model.py:8).  Import recipe:
  * empty stand-in modules for packages the reference imports at module level but never
    touches on this path (kornia, pycolmap, torchvision, sh_utility) — absent offline;
  * a TorchFunctionMode that maps the hard-coded device="cuda" (gs_model.py:505,728,771,792)
    to CPU, re-entered around Function.backward (autograd calls it outside the mode);
  * torch.sort / torch.argsort default to stable=True: depth order inside a pixel rides on
    sort stability (gs_model.py:547), which holds for the CUDA radix sort the author ran and
    not for the CPU default.
Only data is written: inputs and the reference's outputs.  No reference source is copied.
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def import_reference():
    for name in ("kornia", "kornia.metrics", "pycolmap", "torchvision", "sh_utility"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["kornia"].metrics = sys.modules["kornia.metrics"]

    def eval_sh(*a, **k):
        raise RuntimeError("sh_utility is not part of the reference checkout")

    sys.modules["sh_utility"].eval_sh = eval_sh
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    import grouped_cumprod_ref_host

    sys.modules["grouped_cumprod"] = grouped_cumprod_ref_host
    sys.path.insert(0, REF)
    import gs_model  # the reference

    return gs_model


class CudaToCpu(torch.overrides.TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        dev = kwargs.get("device")
        if dev is not None and "cuda" in str(dev):
            kwargs["device"] = "cpu"
        args = tuple("cpu" if isinstance(a, str) and a == "cuda" else a for a in args)
        return func(*args, **kwargs)


def patch_stable_sort():
    _sort, _argsort = torch.sort, torch.argsort

    def sort(input, *a, **k):
        if not a and "dim" not in k:
            k.setdefault("stable", True)
        elif "stable" not in k and len(a) <= 1:
            k["stable"] = True
            if a:
                k["dim"] = a[0]
                a = ()
        return _sort(input, *a, **k)

    def argsort(input, *a, **k):
        k.setdefault("stable", True)
        return _argsort(input, *a, **k)

    torch.sort, torch.argsort = sort, argsort


def make_scene(n_gauss, width, height, max_half, seed, sig_min=0.6):
    g = torch.Generator().manual_seed(seed)
    mean = torch.stack(
        [torch.randint(0, width + 1, (n_gauss,), generator=g), torch.randint(0, height + 1, (n_gauss,), generator=g)], 1
    ).to(torch.int32)
    half = torch.randint(1, max_half + 1, (n_gauss, 2), generator=g).to(torch.int32)
    lim = torch.tensor([width, height], dtype=torch.int32)
    start = torch.minimum((mean - half).clamp(min=0), lim)
    end = torch.minimum((mean + half).clamp(min=0), lim)
    boxsize = torch.prod((end - start + 1).to(torch.int64), dim=1)
    sx = sig_min + 1.8 * torch.rand(n_gauss, generator=g)
    sy = sig_min + 1.8 * torch.rand(n_gauss, generator=g)
    rho = 0.8 * (torch.rand(n_gauss, generator=g) - 0.5)
    cov = torch.stack([sx * sx, rho * sx * sy, rho * sx * sy, sy * sy], 1).reshape(-1, 2, 2)
    vinv = torch.linalg.inv(cov).to(torch.float32).contiguous()
    opacity = (0.05 + 0.9 * torch.rand(n_gauss, 1, generator=g)).to(torch.float32)
    l_d = (0.05 + 0.95 * torch.rand(n_gauss, 3, generator=g)).to(torch.float32)
    wimg = torch.randn(height + 1, width + 1, 3, generator=g)
    return dict(boxsize=boxsize, start=start, end=end, mean=mean, vinv=vinv, opacity=opacity, l_d=l_d, wimg=wimg)


def run_function(F, sc, width, height, chunk_ends):
    vinv = sc["vinv"].clone().requires_grad_(True)
    opacity = sc["opacity"].clone().requires_grad_(True)
    l_d = sc["l_d"].clone().requires_grad_(True)
    batch = torch.tensor(chunk_ends, dtype=torch.int64)
    with CudaToCpu():
        img = F.apply(
            sc["boxsize"], batch, sc["start"], sc["end"], sc["mean"], vinv, opacity, l_d,
            torch.tensor(width, dtype=torch.int32), torch.tensor(height, dtype=torch.int32),
        )
        loss = (img * sc["wimg"]).sum()
    loss.backward()
    return img.detach(), vinv.grad, opacity.grad, l_d.grad


def main():
    gs_model = import_reference()
    patch_stable_sort()
    F = gs_model.custom_autograd_grouped_cumprod
    _bwd = F.backward

    def backward_in_mode(ctx, g):
        with CudaToCpu():
            return _bwd(ctx, g)

    F.backward = staticmethod(backward_in_mode)

    out = {}
    # ---- a5 / a6: the scan call sites -------------------------------------------------
    for name, (n_gauss, w, h, mh, seed) in {"wrap_small": (12, 16, 12, 3, 11), "wrap_mid": (300, 64, 48, 5, 12)}.items():
        sc = make_scene(n_gauss, w, h, mh, seed)
        with CudaToCpu():
            rects = F._create_rects(sc["start"], sc["end"])
            g = torch.Generator().manual_seed(seed + 100)
            anti = (1.0 - 0.9 * torch.rand(rects.size(0), generator=g)).to(torch.float32)
            anti[:: 17] = 0.0  # exact zeros exercise the `!= 0` compaction (gs_model.py:575-578)
            T, mask = F._create_alpha_brend(rects, anti, flag="cumprod")
            grad = torch.randn(rects.size(0), generator=g)
            grad[:: 23] = 0.0
            S, smask = F.grad_cumsum(rects, grad)
            inv = F.unique(rects)
            sorted_inv, index = torch.sort(inv)
        out[name + "/rects"] = rects.numpy()
        out[name + "/anti_opacity"] = anti.numpy()
        out[name + "/T"] = T.numpy()
        out[name + "/T_mask"] = mask.numpy()
        out[name + "/grad"] = grad.numpy()
        out[name + "/S"] = S.numpy()
        out[name + "/S_mask_flipped"] = smask.numpy()  # the reference leaves this mask in flipped order
        out[name + "/sorted_inv"] = sorted_inv.numpy()
        out[name + "/index"] = index.numpy()

    # ---- a7: whole Function, single chunk (the parity contract) and one 2-chunk case ----
    scenes = {
        "fn_6g_16x12": (6, 16, 12, 3, 21, None),
        "fn_200g_64x48": (200, 64, 48, 6, 22, None),
        "fn_1500g_128x96": (1500, 128, 96, 8, 23, None, 1.2),
        # narrow Gaussians in wide boxes: g underflows to exactly 0 at box corners, a pixel's deepest pair then has
        # an exactly-zero suffix sum, grad_cumsum's mask is not all True and — being returned in FLIPPED order
        # (gs_model.py:720-722 vs its use at :642-645) — mis-selects rows: the reference's gradients are garbage here
        "fn_300g_64x48_Q9_INFORMATIONAL": (300, 64, 48, 8, 30, None, 0.5),
        "fn_200g_64x48_2chunks_INFORMATIONAL": (200, 64, 48, 6, 22, 90),
    }
    for name, spec in scenes.items():
        n_gauss, w, h, mh, seed, split = spec[:6]
        sc = make_scene(n_gauss, w, h, mh, seed, *spec[6:])
        ends = [n_gauss] if split is None else [split, n_gauss]
        img, gv, go, gl = run_function(F, sc, w, h, ends)
        for k in ("boxsize", "start", "end", "mean", "vinv", "opacity", "l_d", "wimg"):
            out[f"{name}/{k}"] = sc[k].numpy()
        out[name + "/width_height"] = np.array([w, h], dtype=np.int32)
        out[name + "/chunk_ends"] = np.array(ends, dtype=np.int64)
        out[name + "/image"] = img.numpy()
        out[name + "/grad_vinv"] = gv.numpy()
        out[name + "/grad_opacity"] = go.numpy()
        out[name + "/grad_l_REFERENCE_BUGGY"] = gl.numpy()  # channel-collapsed (gs_model.py:710-712, :763-766)
    path = os.path.join(HERE, "function_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
