"""Caller side on the GPU: the model's forward against the images the reference's own model
forward produced, its gradients against the dense autograd oracle, and the training loop."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import dense_render
from oracle import gs_forward_torch as gft
from oracle import loss_torch
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
    for fused in (True, False):
        cams, _, _ = (gm if fused else gft).camera_inputs(w["mean"], w["variance_q"], w["variance_scale"], w["opacity"], w["color"], w["P"],
                                                           w["K"], w["wh"], TILE_LOGIT)
        for c, cam in enumerate(cams):
            for mine, theirs in (("startpoint", "startpoint"), ("endpoint", "endpoint"), ("mean", "mean_pixel"), ("boxsize", "boxsize")):
                got, want = cam[mine].cpu().numpy(), z[f"{name}/cam{c}/{theirs}"]
                assert got.dtype == want.dtype and np.array_equal(got, want), (fused, c, mine)
            for mine, theirs in (("variance_inverse", "variance_inverse"), ("opacity", "opacity_sigmoid"), ("l_d", "l_d_STAND_IN_SH")):
                np.testing.assert_allclose(cam[mine].cpu().numpy(), z[f"{name}/cam{c}/{theirs}"], rtol=2e-5, atol=1e-6, err_msg=f"{fused} {c} {mine}")


@pytest.mark.parametrize("fused", [True, False])
def test_parameter_gradients_match_dense_oracle(fused, device):
    """d(loss)/d(mean, q, scale, opacity, colour) through projection + HIP Function (fp32) against the same projection
    followed by the dense autograd renderer in fp64 on the CPU."""
    name = CASES[0]
    z, w = load(name, device)
    n_cam = w["P"].shape[0]
    h, wd = int(w["wh"][0, 1]), int(w["wh"][0, 0])
    wimg = torch.randn(n_cam, 3, h, wd, generator=torch.Generator().manual_seed(4))
    model = make_model(w)
    if not fused:  # the oracle's PyTorch projection in front of the same HIP Function
        model.camera_inputs = lambda P, K, wh: gft.camera_inputs(model.mean, model.variance_q, model.variance_scale, model.opacity,
                                                                 model.color, P, K, wh, model.variance_pixel_tile_max_width)
    images = model(w["P"], w["K"], w["wh"], list(range(n_cam)))[0]
    (images * wimg.to(device)).sum().backward()

    wc = {k: v.cpu() for k, v in w.items()}
    leaves = {k: wc[k].clone().requires_grad_(True) for k in ("mean", "variance_q", "variance_scale", "opacity", "color")}
    cams, _, _ = gft.camera_inputs(leaves["mean"], leaves["variance_q"], leaves["variance_scale"], leaves["opacity"], leaves["color"],
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


def random_world(n, n_cam, width, height, seed, device, sigma=0.05):
    from simplegaussiansplat_tk71_amd.synthetic import ring_cameras

    g = torch.Generator().manual_seed(seed)
    P, K, wh = ring_cameras(n_cam, width, height, device=device)
    w = {"mean": torch.randn(n, 3, generator=g) * torch.tensor([0.9, 0.6, 0.9]), "variance_q": torch.randn(n, 4, generator=g),
         "variance_scale": torch.log(sigma * (0.4 + 1.2 * torch.rand(n, 3, generator=g))),
         "opacity": torch.logit(0.02 + 0.96 * torch.rand(n, 1, generator=g)), "color": 0.5 * torch.randn(n, 9, 3, generator=g)}
    w["mean"][: n // 20] *= 6  # some behind / beside the cameras
    w = {k: v.to(device) for k, v in w.items()}
    w.update(P=P, K=K, wh=wh)
    return w


@pytest.mark.parametrize("n,n_cam,width,height,sh_degree", [(20000, 3, 160, 120, 2), (5000, 2, 64, 48, 1), (300, 1, 40, 30, 0)])
def test_fused_projection_equals_torch_formulation(n, n_cam, width, height, sh_degree, device):
    """gcp_project_forward / _backward against the PyTorch formulation (itself pinned to the reference's forward):
    same cull set and depth order, integers equal, floats and all five parameter gradients within fp32 round-off."""
    w = random_world(n, n_cam, width, height, 7 + n, device)
    names = ("mean", "variance_q", "variance_scale", "opacity", "color")
    results = {}
    for fused in (True, False):
        leaves = {k: w[k].clone().requires_grad_(True) for k in names}
        cams, grad_iter, _ = (gm if fused else gft).camera_inputs(*(leaves[k] for k in names), w["P"], w["K"], w["wh"], TILE_LOGIT, L_max=sh_degree)
        gen = torch.Generator().manual_seed(1)
        loss = 0
        for cam in cams:  # upstream gradients belong to Gaussians, not to rows: the row order may differ by near-ties in depth
            for k in ("variance_inverse", "opacity", "l_d"):
                loss = loss + (cam[k] * torch.randn((n, *cam[k].shape[1:]), generator=gen).to(device)[cam["index"]]).sum()
        loss.backward()
        results[fused] = (cams, grad_iter, {k: v.grad for k, v in leaves.items()})
    (cf, gf, gradf), (ct, gt, gradt) = results[True], results[False]
    assert torch.equal(gf, gt)
    flips = 0
    for c, (a, b) in enumerate(zip(cf, ct)):
        # same cull set (up to values that sit within an ulp of a threshold) ...
        sa, sb = set(a["index"].tolist()), set(b["index"].tolist())
        flips += len(sa ^ sb)
        # ... in depth order: depths that differ by an ulp between the formulations may swap neighbours
        z = (torch.cat([w["mean"], torch.ones(n, 1, device=device)], 1) @ w["P"][c].T)[:, 2].double()
        za = z[a["index"]]
        assert bool((za[1:] - za[:-1] >= -1e-5 * za[1:].abs().clamp_min(1.0)).all())
        common = torch.tensor(sorted(sa & sb), device=device)
        ra = torch.full((n,), -1, device=device, dtype=torch.long)
        rb = ra.clone()
        ra[a["index"]] = torch.arange(a["index"].numel(), device=device)
        rb[b["index"]] = torch.arange(b["index"].numel(), device=device)
        for k in ("startpoint", "endpoint", "mean", "boxsize"):
            assert a[k].dtype == b[k].dtype
            flips += int((a[k][ra[common]] != b[k][rb[common]]).sum())
        for k in ("variance_inverse", "opacity", "l_d"):
            torch.testing.assert_close(a[k][ra[common]], b[k][rb[common]], rtol=2e-4, atol=1e-6)
    # a float that lands within an ulp of an integer may truncate differently in the two formulations
    assert flips <= max(2, n * n_cam // 5000), flips
    for k in names:
        scale = gradt[k].abs().max().item()
        assert scale > 0, k
        err = (gradf[k] - gradt[k]).abs().max().item()
        assert err <= 5e-4 * scale, (k, err, scale)
    if sh_degree < 2:  # unused coefficients get no gradient
        assert float(gradf["color"][:, (sh_degree + 1) ** 2:].abs().max()) == 0.0


def test_capture_safe_projection_and_function_equal_the_default_and_run_as_one_graph(device):
    """camera_inputs(capture_safe=True) reads nothing back: every camera's list keeps all N Gaussians, the culled ones
    with empty boxes behind the kept ones.  Image and all five parameter gradients equal the default mode's bit for
    bit, and projection + Function forward + backward are captured into ONE HIP graph and replayed on moved Gaussians."""
    import cuda_kernel as ck

    n, width, height = 4000, 96, 64
    w = random_world(n, 2, width, height, 11, device)
    names = ("mean", "variance_q", "variance_scale", "opacity", "color")
    wh_host = [[width, height]] * 2
    target = torch.rand(2, height + 1, width + 1, 3, device=device)

    def run(leaves, capture_safe, with_grads=True):
        cams, grad_iter, (wd, ht) = gm.camera_inputs(*(leaves[k] for k in names), w["P"], w["K"], wh_host if capture_safe else w["wh"],
                                                     TILE_LOGIT, capture_safe=capture_safe)
        imgs = []
        for cam in cams:
            imgs.append(ck.custom_autograd_grouped_cumprod.apply(cam["boxsize"], None, cam["startpoint"], cam["endpoint"], cam["mean"],
                                                                 cam["variance_inverse"], cam["opacity"], cam["l_d"], wd - 1, ht - 1))
        img = torch.stack(imgs)
        if not with_grads:
            return img, None, grad_iter, cams
        grads = torch.autograd.grad(((img - target[:, : img.shape[1], : img.shape[2]]) ** 2).sum(), [leaves[k] for k in names])
        return img, grads, grad_iter, cams

    leaves = {k: w[k].clone().requires_grad_(True) for k in names}
    img0, g0, it0, cams0 = run(leaves, False)
    with ck.tile_capacity(8 * n):
        img1, g1, it1, cams1 = run(leaves, True)
    assert not ck.capacity_exceeded()
    assert all(c["index"].numel() == n for c in cams1) and all(c["index"].numel() < n for c in cams0)  # something was culled
    assert torch.equal(it0, it1) and torch.equal(img0, img1)
    for a, b, k in zip(g0, g1, names):
        assert torch.equal(a, b), k
    # One graph: projection, binning, blend forward, blend backward, projection backward.  The outputs of the eager steps
    # above (img0, cams0, ...: live autograd graphs that reach the leaves, as a training loop's previous loss does) are
    # deliberately KEPT across the capture: cuda_kernel.GraphedStep traces through fresh aliases of the leaves, so the
    # AccumulateGrad nodes those graphs keep alive on the default stream cannot be pulled into the capture (the cause of
    # round 2's capture_end crash, tools/capture_repro.py) — and torch's stream-mismatch warning must not appear.
    import warnings

    def body(*ls):
        img, _, _, _ = run(dict(zip(names, ls)), True, with_grads=False)
        return ((img - target[:, : img.shape[1], : img.shape[2]]) ** 2).sum(), img

    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        step = ck.GraphedStep(body, [leaves[k] for k in names], capacity=8 * n)
        with torch.no_grad():
            leaves["mean"].add_(0.05 * torch.randn_like(leaves["mean"]))
            leaves["opacity"].sub_(0.3)
        (_, got_img), got_grads = step.replay()
        torch.cuda.synchronize()
    assert not [str(r.message) for r in rec if "AccumulateGrad" in str(r.message)]
    assert not ck.capacity_exceeded()
    got_img, got_grads = got_img.clone(), [g.clone() for g in got_grads]
    img2, g2, _, _ = run(leaves, False)   # the default (synchronising) mode on the moved Gaussians
    assert torch.equal(got_img, img2)
    for a, b, k in zip(got_grads, g2, names):
        assert torch.equal(a, b), k
    assert img0.grad_fn is not None and cams0 and cams1 and it0 is not None and g1  # still referenced here


def test_fused_projection_argument_checks(device):
    w = random_world(50, 1, 32, 24, 3, device)
    args = [w[k] for k in ("mean", "variance_q", "variance_scale", "opacity", "color")]
    with pytest.raises(RuntimeError, match="no CPU path"):
        gm.camera_inputs(*(a.cpu() for a in args), w["P"].cpu(), w["K"].cpu(), w["wh"].cpu(), TILE_LOGIT)
    with pytest.raises(RuntimeError):  # degree 3 is refused by the library
        gm.camera_inputs(*args, w["P"], w["K"], w["wh"], TILE_LOGIT, L_max=3)
    with pytest.raises(RuntimeError):
        gm.camera_inputs(args[0].double(), *args[1:], w["P"], w["K"], w["wh"], TILE_LOGIT)
    empty = [a[:0] for a in args]
    cams, grad_iter, _ = gm.camera_inputs(*empty, w["P"], w["K"], w["wh"], TILE_LOGIT)
    assert cams == [None] and grad_iter.numel() == 0


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


@pytest.mark.parametrize("shape", [(2, 3, 48, 64), (1, 3, 37, 53), (1, 1, 6, 6), (1, 2, 16, 7), (1, 3, 270, 480)])
@pytest.mark.parametrize("lamda", [0.2, 1.0, 0.0])
def test_fused_loss_equals_torch_formulation(shape, lamda, device):
    """gcp_ssim_l1_forward / _backward against the PyTorch formulation of gs_control.py:180-182 (Gaussian 11-tap SSIM,
    reflect padding, + L1): value to 1e-6, gradient — including the exact adjoint of the reflect padding at the image
    border — to 1e-4 of its scale."""
    g = torch.Generator().manual_seed(sum(shape))
    b = torch.rand(shape, generator=g).to(device)
    a0 = (b + 0.25 * torch.randn(shape, generator=g).to(device)).clamp(0, 1)
    out = {}
    for fused in (True, False):
        a = a0.clone().requires_grad_(True)
        loss = (gm if fused else loss_torch).splat_loss(a, b, lamda)
        (3.0 * loss).backward()  # a non-unit upstream gradient
        out[fused] = (loss.detach(), a.grad)
    torch.testing.assert_close(out[True][0], out[False][0], rtol=1e-5, atol=1e-6)
    scale = out[False][1].abs().max().item()
    assert scale > 0
    err = (out[True][1] - out[False][1]).abs().max().item()
    assert err <= 1e-4 * scale, (err, scale)
    with torch.no_grad():  # no maps are written when nothing needs a gradient
        torch.testing.assert_close(gm.splat_loss(a0, b, lamda), out[False][0], rtol=1e-5, atol=1e-6)


def test_fused_loss_argument_checks(device):
    a = torch.rand(1, 3, 16, 16, device=device)
    with pytest.raises(RuntimeError, match="no CPU path"):
        gm.splat_loss(a.cpu(), a.cpu())
    with pytest.raises(RuntimeError):
        gm.splat_loss(a, a[:, :2])
    with pytest.raises(RuntimeError):  # smaller than the reflection the window needs
        gm.splat_loss(a[..., :5], a[..., :5])
    assert float(gm.splat_loss(a, a)) == pytest.approx(0.0, abs=1e-6)


def _dp_worker(rank, world, port, out_path):
    """One rank of the camera-parallel step: render batch[rank::world], all-reduce, compare with the whole batch."""
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # both ranks share the one GPU of the test box
    try:
        device = torch.device("cuda", 0)
        w = random_world(1500, 3, 96, 64, 11, device)
        target = torch.rand(3, 3, 64, 96, generator=torch.Generator().manual_seed(2)).to(device)
        names = ("mean", "variance_q", "variance_scale", "opacity", "color")

        def grads(cams, weight):
            model = make_model(w)
            images, kept, grad_iter = model(w["P"][cams], w["K"][cams], w["wh"][cams], cams.tolist())
            (gm.splat_loss(images, target[cams]) * weight).backward()
            return model, grad_iter

        batch = torch.arange(3, device=device)
        mine = batch[rank::world]
        model, grad_iter = grads(mine, mine.numel() / batch.numel())
        grad_iter = model.allreduce_grads(grad_iter)
        if rank == 0:
            full, full_iter = grads(batch, 1.0)
            worst = 0.0
            for k in names:
                a, b = getattr(model, k).grad, getattr(full, k).grad
                worst = max(worst, ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item())
            torch.save({"worst": worst, "iter_equal": bool(torch.equal(grad_iter, full_iter))}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_camera_data_parallel_step_equals_whole_batch(device, tmp_path):
    """Two processes (gloo; RCCL needs one GPU per rank) each render their cameras of a 3-camera batch; after
    GS_model_with_param.allreduce_grads every parameter gradient equals the single-process whole-batch one."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dp.pt")
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    assert res["iter_equal"]
    assert res["worst"] <= 1e-4, res


def _rotmat_to_qvec_wxyz(R):
    """3x3 rotation -> COLMAP quaternion (w, x, y, z), w >= 0."""
    R = np.asarray(R, dtype=np.float64)
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    x = np.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2
    y = np.sqrt(max(0.0, 1 - R[0, 0] + R[1, 1] - R[2, 2])) / 2
    z = np.sqrt(max(0.0, 1 - R[0, 0] - R[1, 1] + R[2, 2])) / 2
    x, y, z = np.copysign(x, R[2, 1] - R[1, 2]), np.copysign(y, R[0, 2] - R[2, 0]), np.copysign(z, R[1, 0] - R[0, 1])
    return np.array([w, x, y, z])


def test_training_from_a_colmap_directory(device, tmp_path):
    """BASELINE config 4's shape (COLMAP sparse model + images -> training loop) on a synthetic scene written to disk:
    cameras.bin / images.bin / points3D.bin through colmap_io, PNG photographs, then examples/train_cameras.py's
    --colmap path.  (The reference's own scene has no images.bin)"""
    from PIL import Image

    from examples.train_cameras import load_colmap, synthetic_scene, train
    from simplegaussiansplat_tk71_amd import colmap_io

    n_cam, width, height = 5, 64, 48
    start, P, K, wh, targets = synthetic_scene(500, n_cam, width, height, 1, device)
    root = tmp_path / "scene"
    (root / "images").mkdir(parents=True)
    cameras, images = {}, {}
    for c in range(n_cam):
        cameras[c + 1] = {"model": "PINHOLE", "width": width, "height": height,
                          "params": np.array([K[c, 0, 0].item(), K[c, 1, 1].item(), K[c, 0, 2].item(), K[c, 1, 2].item()])}
        Pc = P[c].cpu().numpy().astype(np.float64)
        q = _rotmat_to_qvec_wxyz(Pc[:, :3])
        assert np.allclose(colmap_io.qvec_to_rotmat(q), Pc[:, :3], atol=1e-4)
        images[c + 1] = {"qvec": q, "tvec": Pc[:, 3], "camera_id": c + 1, "name": f"img{c}.png"}
        Image.fromarray((targets[c].permute(1, 2, 0).clamp(0, 1) * 255).round().byte().cpu().numpy()).save(root / "images" / f"img{c}.png")
    xyz = start.cpu().numpy().astype(np.float64)
    colmap_io.write_model(root / "sparse" / "0", cameras, images,
                          {"id": np.arange(1, len(xyz) + 1), "xyz": xyz, "rgb": np.zeros((len(xyz), 3), np.uint8), "error": np.zeros(len(xyz))})
    xyz2, P2, K2, wh2, photos = load_colmap(str(root), device)
    torch.testing.assert_close(xyz2, start, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(P2, P, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(K2, K)
    torch.testing.assert_close(wh2, wh)
    assert photos.shape == (n_cam, 3, height, width) and float((photos - targets).abs().max()) <= 0.5 / 255 + 1e-6
    _, losses = train(xyz2, P2, K2, wh2, photos, iterations=60, densify_from_iter=1000, opacity_reset_interval=0, log=lambda *_: None)
    assert np.mean(losses[-5:]) < 0.85 * np.mean(losses[:5]), (losses[:5], losses[-5:])


def test_render_is_invariant_to_gaussian_order_and_transparent_extras(device):
    """Size-independent properties of projection + depth sort + Function: the picture does not depend on the order the
    Gaussians are stored in (gradients follow the permutation), fully transparent Gaussians change nothing, and
    swapping cameras swaps pictures."""
    n = 4000
    w = random_world(n, 2, 96, 64, 21, device)
    names = ("mean", "variance_q", "variance_scale", "opacity", "color")
    wimg = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(5)).to(device)

    def run(world, cams=(0, 1)):
        model = make_model(world)
        idx = torch.tensor(cams, device=device)
        images = model(world["P"][idx], world["K"][idx], world["wh"][idx], list(cams))[0]
        (images * wimg[: len(cams)]).sum().backward()
        return images.detach(), {k: getattr(model, k).grad for k in names}

    base_img, base_grad = run(w)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(6)).to(device)
    shuffled = dict(w, **{k: w[k][perm].contiguous() for k in names})
    img, grad = run(shuffled)
    torch.testing.assert_close(img, base_img, atol=TOL, rtol=TOL)
    for k in names:
        scale = base_grad[k].abs().max().item()
        assert (grad[k] - base_grad[k][perm]).abs().max().item() <= 1e-4 * scale, k

    extra = 500  # opacity logit -inf -> alpha exactly 0: no contribution, zero gradient for everything but nothing breaks
    g = torch.Generator().manual_seed(7)
    more = {k: torch.cat([w[k], w[k][torch.randint(0, n, (extra,), generator=g).to(device)]]) for k in names}
    more["opacity"][n:] = -float("inf")
    img2, grad2 = run(dict(w, **more))
    torch.testing.assert_close(img2, base_img, atol=TOL, rtol=TOL)
    for k in names:
        assert torch.isfinite(grad2[k]).all(), k
        scale = base_grad[k].abs().max().item()
        assert (grad2[k][:n] - base_grad[k]).abs().max().item() <= 1e-4 * scale, k
    assert float(grad2["color"][n:].abs().max()) == 0.0

    swapped = run(w, cams=(1, 0))[0]
    torch.testing.assert_close(swapped[0], base_img[1], atol=0, rtol=0)
    torch.testing.assert_close(swapped[1], base_img[0], atol=0, rtol=0)


@pytest.mark.parametrize("n", [1, 3, 1001, 3 * 4096 + 2])
def test_hip_adam_equals_torch_adam(n, device):
    """gcp_adam_step against torch.optim.Adam (the optimiser the reference builds, gs_model.py:43-47): ten steps with
    changing gradients and a learning-rate change in between, parameters and both moments."""
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g).to(device)
    a, b = p0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
    ours, theirs = gm.HipAdam([{"params": a, "lr": 0.01}]), torch.optim.Adam([{"params": b, "lr": 0.01}])
    for it in range(10):
        grad = (torch.randn(n, generator=g) * (10.0 ** torch.randint(-3, 2, (1,), generator=g).item())).to(device)
        a.grad, b.grad = grad.clone(), grad.clone()
        if it == 5:
            ours.param_groups[0]["lr"] = theirs.param_groups[0]["lr"] = 0.002
        ours.step()
        theirs.step()
        ours.zero_grad()
        theirs.zero_grad(set_to_none=True)
        assert a.grad is None
    # fp32 round-off only (torch forms the first moment with lerp, m + (g - m)(1 - b1), and divides where this
    # multiplies): a few ulp of the largest magnitude in each tensor
    for got, want in ((a.detach(), b.detach()), (ours.state[a]["exp_avg"], theirs.state[b]["exp_avg"]),
                      (ours.state[a]["exp_avg_sq"], theirs.state[b]["exp_avg_sq"])):
        assert (got - want).abs().max().item() <= 1e-6 * max(want.abs().max().item(), 1e-30)
    with pytest.raises(RuntimeError):
        cpu = torch.zeros(4, requires_grad=True)
        cpu.grad = torch.ones(4)
        gm.HipAdam([{"params": cpu, "lr": 0.1}]).step()
