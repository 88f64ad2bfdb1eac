#!/usr/bin/env python3
"""Generate tests/golden/forward_golden.npz by RUNNING the reference's model forward on CPU.

Runs only in the build container (needs /root/reference and oracle/_ref/):
    make -C oracle ref_host && python -B tests/golden/make_forward_golden.py

What runs is the reference's `GS_model_with_param.forward` (gs_model.py:277-460) — camera projection, pixel
covariance, 3-sigma boxes, depth sort, culling — imported from /root/reference with the recipe of
make_function_golden.py.  What is recorded is every argument list it hands to
`custom_autograd_grouped_cumprod.apply` (gs_model.py:449) and the image batch it returns.

One stand-in is unavoidable: `sh_utility.eval_sh` (gs_model.py:9,335) is not part of the reference's checkout.
The build's own `eval_sh` (oracle/gs_forward_torch.py; the HIP kernel evaluates the same basis) is plugged in, so the colour argument `l_d`
is NOT a reference output (it is stored for completeness and flagged); every other recorded array is.
Only data is written: inputs and the reference's outputs.  No reference source is copied.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_function_golden as mfg  # noqa: E402


def make_world(n_gauss, n_cam, width, height, seed):
    """Points in a unit-ish blob, cameras on a ring looking at the origin (COLMAP convention: x right, y down, z forward)."""
    g = torch.Generator().manual_seed(seed)
    mean = 0.6 * torch.randn(n_gauss, 3, generator=g)
    variance_q = torch.randn(n_gauss, 4, generator=g)
    variance_scale = torch.log(0.03 + 0.09 * torch.rand(n_gauss, 3, generator=g))
    opacity = torch.logit(0.05 + 0.9 * torch.rand(n_gauss, 1, generator=g))
    color = 0.4 * torch.randn(n_gauss, 9, 3, generator=g)
    color[:, 0, :] += 1.77
    P, K = [], []
    for c in range(n_cam):
        ang = 2 * np.pi * c / n_cam + 0.3
        eye = torch.tensor([3.2 * np.cos(ang), 0.5 * np.sin(2 * ang), 3.2 * np.sin(ang)], dtype=torch.float32)
        fwd = -eye / eye.norm()
        right = torch.linalg.cross(torch.tensor([0.0, -1.0, 0.0]), fwd)
        right = right / right.norm()
        down = torch.linalg.cross(fwd, right)
        R = torch.stack([right, down, fwd])  # world -> camera
        P.append(torch.cat([R, (-R @ eye)[:, None]], dim=1))
        f = 0.9 * width
        K.append(torch.tensor([[f, 0, width / 2], [0, f, height / 2], [0, 0, 1]], dtype=torch.float32))
    wh = torch.tensor([[width, height]] * n_cam, dtype=torch.float32)
    return dict(mean=mean, variance_q=variance_q, variance_scale=variance_scale, opacity=opacity, color=color,
                P=torch.stack(P), K=torch.stack(K), wh=wh)


def main():
    gs_model = mfg.import_reference()
    mfg.patch_stable_sort()
    from oracle.gs_forward_torch import eval_sh

    gs_model.eval_sh = eval_sh  # stand-in for the missing sh_utility (see the module docstring)
    gs_model.Utilities.gpu_mem = staticmethod(lambda tag="": None)  # prints torch.cuda statistics only
    F = gs_model.custom_autograd_grouped_cumprod
    captured = []
    _apply = F.apply

    def recording_apply(*args):
        captured.append([a.detach().clone() if torch.is_tensor(a) else a for a in args])
        return _apply(*args)

    F.apply = staticmethod(recording_apply)

    out = {}
    for name, (n_gauss, n_cam, w, h, seed) in {"fwd_40g_2cam_32x24": (40, 2, 32, 24, 5), "fwd_600g_3cam_96x64": (600, 3, 96, 64, 6)}.items():
        wd = make_world(n_gauss, n_cam, w, h, seed)
        captured.clear()
        with mfg.CudaToCpu():
            model = gs_model.GS_model_with_param(
                wd["mean"].clone(), wd["variance_q"].clone(), wd["variance_scale"].clone(), wd["opacity"].clone(),
                1e-12, 0.0004, 0.01, 0.005, 0.04, 0.00016, 0.0000016, 0.01, 30_000, 0.0025, 0.025, 0.005, 0.001)
            with torch.no_grad():
                model.color.copy_(wd["color"])
                images, names, grad_iter = model(wd["P"], wd["K"], wd["wh"], [f"cam{c}" for c in range(n_cam)])
        for k, v in wd.items():
            out[f"{name}/{k}"] = v.numpy()
        out[f"{name}/images_reference_layout"] = images.detach().numpy()
        out[f"{name}/grad_iter"] = grad_iter.numpy()
        out[f"{name}/n_rendered"] = np.array(len(captured))
        keys = ("boxsize", "batch", "startpoint", "endpoint", "mean_pixel", "variance_inverse", "opacity_sigmoid",
                "l_d_STAND_IN_SH", "width", "height")
        for c, args in enumerate(captured):
            for k, a in zip(keys, args):
                out[f"{name}/cam{c}/{k}"] = a.numpy()
        print(name, "rendered", len(captured), "cameras;", [int(a[0].numel()) for a in captured], "Gaussians kept")
    # ---- densify / split / clone / prune / opacity reset (gs_model.py:190-271) on the same CPU recipe -----------------
    n_gauss, extent = 160, 3.0
    g = torch.Generator().manual_seed(77)
    wd = make_world(n_gauss, 1, 32, 24, 9)
    wd["variance_scale"] = torch.log(0.004 + 0.5 * torch.rand(n_gauss, 3, generator=g) ** 3)  # small, medium and huge
    wd["opacity"] = torch.logit(0.0005 + 0.3 * torch.rand(n_gauss, 1, generator=g) ** 2)        # some below the prune threshold
    grads_norm = 0.002 * torch.rand(n_gauss, generator=g)
    grads_iter = torch.randint(0, 4, (n_gauss,), generator=g).to(torch.int16)
    for k in ("mean", "variance_q", "variance_scale", "opacity", "color"):
        out[f"densify/{k}"] = wd[k].numpy()
    out["densify/mean_grads_norm"], out["densify/mean_grads_iter"] = grads_norm.numpy(), grads_iter.numpy()
    out["densify/extent"] = np.array(extent)

    def fresh():
        with mfg.CudaToCpu():
            m = gs_model.GS_model_with_param(
                wd["mean"].clone(), wd["variance_q"].clone(), wd["variance_scale"].clone(), wd["opacity"].clone(),
                1e-12, 0.0004, 0.01, 0.005, 0.04, 0.00016, 0.0000016, 0.01, 30_000, 0.0025, 0.025, 0.005, 0.001)
        with torch.no_grad():
            m.color.copy_(wd["color"])
        m.mean_grads_norm, m.mean_grads_iter = grads_norm.clone(), grads_iter.clone()
        return m

    def record(tag, m):
        for k in ("mean", "variance_q", "variance_scale", "opacity", "color"):
            out[f"densify/{tag}/{k}"] = getattr(m, k).detach().numpy()
        out[f"densify/{tag}/mean_grads_norm"] = m.mean_grads_norm.numpy()
        out[f"densify/{tag}/mean_grads_iter"] = m.mean_grads_iter.numpy()

    with mfg.CudaToCpu():
        m = fresh()
        out["densify/grads_per_iter_norm"] = m.param_grads_per_iter_norm().numpy()
        m.densify_and_clone(extent)
        record("clone", m)
        m = fresh()
        torch.manual_seed(123)
        m.densify_and_split(extent)
        record("split_seed123", m)
        m = fresh()
        torch.manual_seed(321)
        m.densify_and_prune(extent, 0.001)
        record("prune_seed321", m)
        m = fresh()
        m.reset_opacity(0.01)
        record("reset_opacity", m)
    print("densify: clone", out["densify/clone/mean"].shape[0], "split", out["densify/split_seed123/mean"].shape[0], "prune",
          out["densify/prune_seed321/mean"].shape[0], "of", n_gauss)
    path = os.path.join(HERE, "forward_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
