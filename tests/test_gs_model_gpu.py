"""Caller side on the GPU: the model's forward against the images the reference's own model
forward produced, its gradients against the dense autograd oracle, and the training loop."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import dense_render
from simplegaussiansplat_tk71_amd import gs_model as gm

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "forward_golden.npz")
CASES = ("fwd_40g_2cam_32x24", "fwd_600g_3cam_96x64")
TILE_LOGIT = math.log(0.04 / 0.96)
TOL = 1e-5  # fp32 image values, as for the Function itself


def load(name, device):
    z = np.load(GOLDEN)
    w = {k: torch.from_numpy(z[f"{name}/{k}"]).to(device) for k in ("mean", "variance_q", "variance_scale", "opacity", "color", "P", "K", "wh")}
    return z, w


def make_model(w, **kw):
    model = gm.GS_model_with_param(w["mean"].clone(), w["variance_q"].clone(), w["variance_scale"].clone(), w["opacity"].clone(), **kw)
    with torch.no_grad():
        model.color.copy_(w["color"])
    return model


@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference_model_images(name, device):
    """`GS_model_with_param.forward` of the reference, run on CPU with its own Function (make_forward_golden.py),
    against the build's model on the HIP Function — in the reference's (scrambled, Q6) output layout."""
    z, w = load(name, device)
    n_cam = w["P"].shape[0]
    images, names, grad_iter = make_model(w, reference_layout=True)(w["P"], w["K"], w["wh"], [f"cam{c}" for c in range(n_cam)])
    want = torch.from_numpy(z[f"{name}/images_reference_layout"])
    assert images.shape == want.shape and names == [f"cam{c}" for c in range(n_cam)]
    assert np.array_equal(grad_iter.cpu().numpy(), z[f"{name}/grad_iter"])
    torch.testing.assert_close(images.detach().cpu(), want, atol=TOL, rtol=TOL)
    # the honest layout is a permutation of the same pixels
    proper = make_model(w)(w["P"], w["K"], w["wh"], list(range(n_cam)))[0]
    h, wd = int(w["wh"][0, 1]), int(w["wh"][0, 0])
    torch.testing.assert_close(proper.permute(0, 2, 3, 1).reshape(-1, 3, h, wd), images, atol=0, rtol=0)
    # the device projection gives the integers the reference computed on the CPU
    cams, _, _ = gm.camera_inputs(w["mean"], w["variance_q"], w["variance_scale"], w["opacity"], w["color"], w["P"], w["K"], w["wh"], TILE_LOGIT)
    for c, cam in enumerate(cams):
        for mine, theirs in (("startpoint", "startpoint"), ("endpoint", "endpoint"), ("mean", "mean_pixel"), ("boxsize", "boxsize")):
            assert np.array_equal(cam[mine].cpu().numpy(), z[f"{name}/cam{c}/{theirs}"]), (c, mine)


def test_parameter_gradients_match_dense_oracle(device):
    """d(loss)/d(mean, q, scale, opacity, colour) through projection + HIP Function (fp32) against the same projection
    followed by the dense autograd renderer in fp64 on the CPU."""
    name = CASES[0]
    z, w = load(name, device)
    n_cam = w["P"].shape[0]
    h, wd = int(w["wh"][0, 1]), int(w["wh"][0, 0])
    wimg = torch.randn(n_cam, 3, h, wd, generator=torch.Generator().manual_seed(4))
    model = make_model(w)
    images = model(w["P"], w["K"], w["wh"], list(range(n_cam)))[0]
    (images * wimg.to(device)).sum().backward()

    wc = {k: v.cpu() for k, v in w.items()}
    leaves = {k: wc[k].clone().requires_grad_(True) for k in ("mean", "variance_q", "variance_scale", "opacity", "color")}
    cams, _, _ = gm.camera_inputs(leaves["mean"], leaves["variance_q"], leaves["variance_scale"], leaves["opacity"], leaves["color"],
                                  wc["P"], wc["K"], wc["wh"], TILE_LOGIT)
    dense = torch.stack([dense_render.render(c["startpoint"], c["endpoint"], c["mean"], c["variance_inverse"], c["opacity"], c["l_d"],
                                             wd, h, dtype=torch.float64) for c in cams])
    dense = dense[:, 1:, 1:, :].permute(0, 3, 1, 2)
    torch.testing.assert_close(images.detach().cpu().double(), dense.detach(), atol=TOL, rtol=TOL)
    (dense * wimg.double()).sum().backward()
    for k, leaf in leaves.items():
        got, want = getattr(model, k).grad.cpu().double(), leaf.grad.double()
        assert torch.isfinite(got).all(), k
        scale = want.abs().max().item()
        assert scale > 0, k
        assert (got - want).abs().max().item() <= 2e-4 * scale, (k, (got - want).abs().max().item(), scale)


def test_invisible_cameras_are_dropped(device):
    """A camera that sees nothing leaves the batch, with its name (gs_model.py:414-417, :456)."""
    z, w = load(CASES[0], device)
    P = w["P"].clone()
    P[1, :, 3] = torch.tensor([0.0, 0.0, -50.0], device=device)  # everything behind camera 1
    images, names, grad_iter = make_model(w)(P, w["K"], w["wh"], ["a", "b"])
    assert images.shape[0] == 1 and names == ["a"]


def test_training_loop_learns_and_densifies(device):
    from examples.train_cameras import synthetic_scene, train

    start, P, K, wh, targets = synthetic_scene(600, 6, 64, 48, 0, device)
    model, losses = train(start, P, K, wh, targets, iterations=150, densify_from_iter=60, densification_interval=45,
                          opacity_reset_interval=0, log=lambda *_: None)
    assert all(l == l for l in losses)
    assert np.mean(losses[-10:]) < 0.75 * np.mean(losses[:10]), (np.mean(losses[:10]), np.mean(losses[-10:]))
    n = model.mean.shape[0]
    assert model.variance_q.shape[0] == n and model.color.shape[0] == n and model.mean_grads_iter.shape[0] == n
    model.reset_opacity(0.01)
    assert float(torch.sigmoid(model.opacity.detach()).max()) <= 0.01 + 1e-6
