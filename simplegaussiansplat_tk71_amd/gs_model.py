"""Caller of the hot path: 3-D Gaussians + cameras -> the inputs of the rasterise-and-blend Function -> images.

Mirrors the reference's model class (reference: gs_model.py:123-460, `GS_model_with_param`): same parameter set
(mean, variance_q, variance_scale, opacity, color), same `forward(P, K, wh, image_sample)` return value
`[images, image_sample, grad_iter]`, same per-parameter Adam learning rates, densify / prune / opacity reset.
Row f4 of SURVEY.md §8: this is caller integration, PyTorch on the GPU around the HIP Function — not a kernel.

The projection math (`camera_inputs`, reference: gs_model.py:277-425) is pinned against the reference's own
forward run on CPU (tests/golden/forward_golden.npz: the arguments the reference hands to
`custom_autograd_grouped_cumprod.apply`).  Differences, all deliberate:
  * the 3-sigma box comes from a closed-form 2x2 eigen-decomposition on the device; the reference moves every
    covariance to the CPU for `torch.linalg.eigh` and back (gs_model.py:327-332).  For a positive semi-definite
    matrix `V^2 |lambda|` is just its diagonal, so the box is 3*sqrt(diag) exactly;
  * the depth sort is stable (the reference's `torch.argsort`, :356, leaves ties undefined);
  * images are permuted to (B, 3, H, W); the reference `reshape`s (H, W, 3) memory into (3, H, W), scrambling
    channels (gs_model.py:454) — `reference_layout=True` reproduces that;
  * one Function call per camera, never chunked (nothing of pair-list size exists here; gs_model.py:428);
  * `eval_sh` below stands in for the reference's `sh_utility.eval_sh`, which is not in its checkout
    (gs_model.py:9,335): real spherical harmonics up to degree 2 in the usual 3DGS convention — parity unpinned;
  * tensors live on the parameters' device instead of a hard-coded "cuda";
  * on the GPU the whole per-Gaussian chain is ONE HIP kernel per camera and direction (`gcp_project_forward`,
    `gcp_project_backward`, csrc/gcp_project.hip) instead of ~150 PyTorch kernels: at 10^6 Gaussians the reference's
    formulation costs 64 ms forward + 110 ms backward around a 1.8 ms Function.  `camera_inputs(fused=False)` keeps
    the PyTorch formulation: it runs anywhere and is what the kernels are tested against.
"""
import math

import torch

from . import _lib
from . import raster as _raster
from .cuda_kernel import custom_autograd_grouped_cumprod

__all__ = [
    "GS_dataset",
    "GS_model_with_param",
    "camera_inputs",
    "eval_sh",
    "qvec_to_rotmat_batch",
    "pixel_jacobian_batch",
    "invert_2x2_batch",
    "box_halfsize",
    "get_expon_lr_func",
    "mean_neighbour_distance",
    "ssim",
    "splat_loss",
]

_SH_C0 = 0.28209479177387814  # 1 / (2 sqrt(pi))
_SH_C1 = 0.4886025119029199  # sqrt(3 / (4 pi))
_SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)


def eval_sh(deg, sh, dirs):
    """Colour of a real-SH expansion in direction `dirs`: sh (..., 3, (deg+1)^2), dirs (..., 3) unit -> (..., 3).
    Stand-in for the reference's missing sh_utility.eval_sh (call site gs_model.py:335-338); degree <= 2."""
    if not 0 <= deg <= 2:
        raise ValueError("eval_sh supports degrees 0..2")
    if sh.shape[-1] < (deg + 1) ** 2:
        raise ValueError("not enough SH coefficients for the degree")
    out = _SH_C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        out = out - _SH_C1 * y * sh[..., 1] + _SH_C1 * z * sh[..., 2] - _SH_C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            out = (out + _SH_C2[0] * (x * y) * sh[..., 4] + _SH_C2[1] * (y * z) * sh[..., 5]
                   + _SH_C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] + _SH_C2[3] * (x * z) * sh[..., 7]
                   + _SH_C2[4] * (xx - yy) * sh[..., 8])
    return out


def qvec_to_rotmat_batch(q):
    """(N, 4) unit quaternions in (x, y, z, w) order -> (N, 3, 3) (reference: uitility.py:231-254)."""
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    r0 = torch.stack([1 - 2 * (y**2 + z**2), 2 * (x * y - w * z), 2 * (x * z + w * y)], dim=1)
    r1 = torch.stack([2 * (x * y + w * z), 1 - 2 * (x**2 + z**2), 2 * (y * z - w * x)], dim=1)
    r2 = torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x**2 + y**2)], dim=1)
    return torch.stack([r0, r1, r2], dim=1)


def pixel_jacobian_batch(K, xyz):
    """d(pixel)/d(camera point): K (C, 3, 3), xyz (C, N, 3) -> (C, N, 2, 3) (reference: uitility.py:257-287)."""
    fx, fy = K[:, 0, 0].unsqueeze(1), K[:, 1, 1].unsqueeze(1)
    X, Y, Z = xyz[..., 0], xyz[..., 1], xyz[..., 2].clamp_min(1e-2)
    zero = torch.zeros_like(Z)
    row0 = torch.stack([fx / Z, zero, -fx * X / (Z**2)], dim=-1)
    row1 = torch.stack([zero, fy / Z, -fy * Y / (Z**2)], dim=-1)
    return torch.stack([row0, row1], dim=-2)


def invert_2x2_batch(A, eps=1e-6):
    """Closed-form inverse with `det + eps` (reference: uitility.py:431-462)."""
    a, b, c, d = A[..., 0, 0], A[..., 0, 1], A[..., 1, 0], A[..., 1, 1]
    det = a * d - b * c + eps
    return torch.stack([torch.stack([d / det, -b / det], dim=-1), torch.stack([-c / det, a / det], dim=-1)], dim=-2)


def box_halfsize(cov):
    """3-sigma half extents `3*sqrt(V^2 |lambda|)` of 2x2 covariances (..., 2, 2) -> (..., 2), on the device
    (reference: gs_model.py:327-332 via CPU eigh, lower triangle)."""
    a, b, c = cov[..., 0, 0], cov[..., 1, 0], cov[..., 1, 1]
    m, d = 0.5 * (a + c), 0.5 * (a - c)
    r = torch.sqrt(d * d + b * b)
    lo, hi = m - r, m + r
    ratio = torch.where(r > 0, d / r.clamp_min(torch.finfo(cov.dtype).tiny), torch.zeros_like(d))
    w_hi, w_lo = 0.5 * (1.0 + ratio), 0.5 * (1.0 - ratio)  # squared x-components of the two eigenvectors
    ex = torch.where(lo >= 0, a, w_lo * lo.abs() + w_hi * hi.abs())
    ey = torch.where(lo >= 0, c, w_hi * lo.abs() + w_lo * hi.abs())
    # r == 0: eigh returns the identity basis, eigenvalues (a, a)
    return 3.0 * torch.sqrt(torch.stack([ex, ey], dim=-1).abs())


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear learning-rate decay with an optional eased start (reference: uitility.py:573-607)."""

    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        delay = 1.0
        if lr_delay_steps > 0:
            delay = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0), 1))
        t = min(max(step / max_steps, 0.0), 1.0)
        return delay * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)

    return helper


def mean_neighbour_distance(n, cloud, batch_size=2000):
    """Mean distance to the n nearest points (self included), repeated on 3 axes: the initial scale
    (reference: uitility.py:68-78, `kyori2`)."""
    out = torch.zeros((cloud.shape[0], 1), device=cloud.device, dtype=cloud.dtype)
    for i in range(0, cloud.shape[0], batch_size):
        d = torch.cdist(cloud[i:i + batch_size], cloud)
        out[i:i + batch_size] = torch.topk(d, min(n, d.shape[1]), dim=1, largest=False).values.mean(dim=1, keepdim=True)
    return out.repeat(1, 3)


def _box_clamp(wh, tile_max_width, dev):
    """Upper bound of the 3-sigma half extents: 10 * sqrt(W*H) * sigmoid(tile_max_width) (gs_model.py:364-365)."""
    tile_max = torch.sqrt((wh[0, 0] * wh[0, 1]).to(torch.int32).to(torch.float32)) * torch.sigmoid(
        torch.as_tensor(tile_max_width, dtype=torch.float32, device=dev))
    return (tile_max * 10).item()


class _ProjectCamera(torch.autograd.Function):
    """One camera of `camera_inputs` on the HIP library (csrc/gcp_project.hip): gcp_project_forward, the library's
    stable radix sort on the depth keys, gcp_project_gather; backward = gcp_project_backward."""

    @staticmethod
    def forward(ctx, mean, variance_q, variance_scale, opacity, color, cam_P, cam_K, width, height, box_clamp, L_max):
        dev, n = mean.device, mean.shape[0]
        args = [t.detach().contiguous() for t in (mean, variance_q, variance_scale, opacity, color, cam_P, cam_K)]
        for t in args:
            if t.dtype != torch.float32 or t.device != dev:
                raise RuntimeError("projection expects float32 tensors on one device")
        if not mean.is_cuda:
            raise RuntimeError("the fused projection is a HIP kernel: tensors must live on the GPU (no CPU path)")
        lib = _lib.load()
        f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
        i32 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)  # noqa: E731
        record, sort_key, row_of = f32(n, 16), i32(n), i32(n)
        keep = torch.empty(n, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.gcp_project_forward(*(t.data_ptr() for t in args), n, L_max, color.shape[1], width, height, box_clamp,
                                               record.data_ptr(), sort_key.data_ptr(), keep.data_ptr(), row_of.data_ptr(), stream),
                       "gcp_project_forward")
            m = int(keep.sum()) if n else 0  # the one device->host read: sizes of the outputs
            # culled Gaussians carry the largest key: the first m entries of the stable permutation are the kept ones in
            # depth order, ties in the Gaussians' own order
            perm = _raster.stable_sort_keys(sort_key, key_bits=31)[1] if n else sort_key
            start, end, mean_xy, boxsize = i32(m, 2), i32(m, 2), i32(m, 2), torch.empty(m, dtype=torch.int64, device=dev)
            vinv, alpha, l_d, index = f32(m, 2, 2), f32(m, 1), f32(m, 3), torch.empty(m, dtype=torch.int64, device=dev)
            _lib.check(lib.gcp_project_gather(record.data_ptr(), perm.data_ptr(), m, start.data_ptr(), end.data_ptr(),
                                              mean_xy.data_ptr(), boxsize.data_ptr(), vinv.data_ptr(), alpha.data_ptr(),
                                              l_d.data_ptr(), index.data_ptr(), row_of.data_ptr(), stream), "gcp_project_gather")
        keep = keep.view(torch.bool)
        ctx.save_for_backward(*args, row_of)
        ctx.L_max = L_max
        out = (vinv, alpha, l_d, start, end, mean_xy, boxsize, index, keep)
        ctx.mark_non_differentiable(*out[3:])
        return out

    @staticmethod
    def backward(ctx, g_vinv, g_alpha, g_ld, *_):
        *args, row_of = ctx.saved_tensors
        mean, variance_q, variance_scale, opacity, color = args[:5]
        grads = [torch.empty_like(t) for t in (mean, variance_q, variance_scale, opacity, color)]  # every row is written
        g = [t.contiguous().float() for t in (g_vinv, g_alpha, g_ld)]
        with torch.cuda.device(mean.device):
            stream = torch.cuda.current_stream(mean.device).cuda_stream
            _lib.check(_lib.load().gcp_project_backward(
                *(t.data_ptr() for t in args), mean.shape[0], ctx.L_max, color.shape[1], row_of.data_ptr(),
                *(t.data_ptr() for t in g), *(t.data_ptr() for t in grads), stream), "gcp_project_backward")
        return (*grads, None, None, None, None, None, None)


def camera_inputs(mean, variance_q, variance_scale, opacity, color, P, K, wh, tile_max_width, L_max=2, sh=eval_sh, fused=None):
    """Per camera, the depth-ordered, culled arguments of the Function (reference: gs_model.py:277-425).

    mean (N,3), variance_q (N,4 xyzw), variance_scale (N,3 log), opacity (N,1 logit), color (N,(L+1)^2,3),
    P (C,3,4) world->camera, K (C,3,3), wh (C,2), tile_max_width = logit of the box clamp as a fraction of
    sqrt(W*H)/10.  Returns a list with one dict per camera (None where nothing is visible, :414-417) holding
    boxsize, startpoint, endpoint, mean, variance_inverse, opacity, l_d, index (Gaussian ids, depth order),
    and the (N,) bool `grad_iter` of Gaussians seen by any camera (:401-407).

    `fused` (default: on for GPU tensors with the built-in `eval_sh`) runs one HIP kernel per camera and direction;
    otherwise the chain is the PyTorch formulation below, differentiated by autograd."""
    dev = mean.device
    if fused is None:
        fused = mean.is_cuda and sh is eval_sh
    if fused:
        if sh is not eval_sh:
            raise ValueError("the fused projection evaluates the built-in real SH basis")
        width, height = wh[0, 0].to(torch.int32), wh[0, 1].to(torch.int32)
        clamp = _box_clamp(wh, tile_max_width, dev)
        grad_iter = torch.zeros(mean.shape[0], device=dev, dtype=torch.bool)
        cams = []
        for c in range(P.shape[0]):
            vinv, alpha, l_d, start, end, mean_xy, boxsize, index, keep = _ProjectCamera.apply(
                mean, variance_q, variance_scale, opacity, color, P[c], K[c], int(width), int(height), clamp, L_max)
            grad_iter |= keep
            cams.append(None if index.numel() == 0 else {
                "boxsize": boxsize, "startpoint": start, "endpoint": end, "mean": mean_xy, "variance_inverse": vinv,
                "opacity": alpha, "l_d": l_d, "index": index})
        return cams, grad_iter, (width, height)
    return _camera_inputs_torch(mean, variance_q, variance_scale, opacity, color, P, K, wh, tile_max_width, L_max, sh)


def _camera_inputs_torch(mean, variance_q, variance_scale, opacity, color, P, K, wh, tile_max_width, L_max, sh):
    """The reference's formulation, op for op (gs_model.py:277-425), on whatever device the tensors live."""
    dev = mean.device
    n, n_cam = mean.shape[0], P.shape[0]
    width, height = wh[0, 0].to(torch.int32), wh[0, 1].to(torch.int32)
    fmax, fmin = torch.finfo(torch.float32).max, torch.finfo(torch.float32).min
    imax, imin = torch.iinfo(torch.int32).max, torch.iinfo(torch.int32).min

    homo = torch.hstack((mean, torch.ones((n, 1), device=dev, dtype=mean.dtype)))[None]
    mean_camera = homo @ P.transpose(1, 2)  # (C, N, 3)
    pix_h = mean_camera @ K.transpose(1, 2)
    mean_pixel = pix_h[:, :, 0:2] / pix_h[:, :, 2][:, :, None].clamp_min(1e-2)

    q = variance_q / torch.norm(variance_q, dim=1, keepdim=True).clamp_min(1e-8)
    rot = qvec_to_rotmat_batch(q)
    s_diag = torch.eye(3, dtype=torch.float32, device=dev)[None] * torch.exp(variance_scale)[:, None, :]
    cov = rot @ s_diag @ s_diag.transpose(1, 2) @ rot.transpose(1, 2)
    cov_cam = P[:, None, :, 0:3] @ cov[None] @ P.transpose(1, 2)[:, None, 0:3, :]
    J = pixel_jacobian_batch(K, mean_camera)
    cov_pix = (J @ cov_cam @ J.transpose(2, 3)).clamp(max=fmax / 1000, min=fmin / 1000) \
        + 1e-6 * torch.eye(2, dtype=torch.float32, device=dev)[None, None]
    half = box_halfsize(cov_pix.detach())  # boxes are integers downstream: no gradient path (:365)

    view = -mean_camera / torch.norm(mean_camera, dim=-1, keepdim=True).clamp_min(1e-8)
    l_d = sh(L_max, color[None].expand(n_cam, -1, -1, -1).transpose(2, 3), view)
    vinv = invert_2x2_batch(cov_pix)

    z_index = torch.argsort(mean_camera[:, :, 2].detach(), dim=1, stable=True)
    cam = torch.arange(n_cam, device=dev)[:, None]
    mc_z = mean_camera[cam, z_index, 2]
    op_z = torch.sigmoid(opacity)[None].expand(n_cam, -1, -1)[cam, z_index]
    mp_z = mean_pixel[cam, z_index].clamp(max=imax / 1000, min=imin / 1000).to(torch.int32)
    vinv_z = vinv[cam, z_index]
    l_z = l_d[cam, z_index]
    half_z = half[cam, z_index].clamp(max=_box_clamp(wh, tile_max_width, dev)).to(torch.int32)

    grad_iter = torch.zeros(n, device=dev, dtype=torch.bool)
    cams = []
    for c in range(n_cam):
        bw, bh, mx, my = half_z[c, :, 0], half_z[c, :, 1], mp_z[c, :, 0], mp_z[c, :, 1]
        keep = (mc_z[c] > 0) & (bw != 0) & (mx - bw < width) & (mx + bw > 0) & (my - bh < height) & (my + bh > 0)
        grad_iter[z_index[c, keep]] = True
        if not bool(keep.any()):
            cams.append(None)
            continue
        m, b = mp_z[c][keep], half_z[c][keep]
        lim = torch.stack([width, height]).to(dev)
        start = torch.minimum((m - b).clamp(min=0), lim)
        end = torch.minimum((m + b).clamp(min=0), lim)
        cams.append({
            "boxsize": torch.prod(end - start + 1, dim=1),
            "startpoint": start, "endpoint": end, "mean": m,
            "variance_inverse": vinv_z[c][keep].contiguous(), "opacity": op_z[c][keep].contiguous(),
            "l_d": l_z[c][keep].contiguous(), "index": z_index[c, keep],
        })
    return cams, grad_iter, (width, height)


class GS_dataset(torch.utils.data.Dataset):
    """Cameras and their image names (reference: gs_model.py:13-30)."""

    def __init__(self, P, K, wh, image_sample):
        self.P, self.K, self.wh, self.image_sample = P, K, wh, image_sample

    def __len__(self):
        return len(self.P)

    def __getitem__(self, idx):
        return [self.P[idx], self.K[idx], self.wh[idx], self.image_sample[idx]]

    def get_camera_extent(self, reference_translation=False):
        """Largest distance of a camera from the mean camera position.  The reference measures it on the translation
        column of [R|t] (gs_model.py:23-30) — the world origin in camera coordinates, which is the same point for every
        camera that looks at the origin; the camera centres -R^T t are used here (`reference_translation=True`
        restores the reference's quantity)."""
        t = self.P[:, :, 3]
        if not reference_translation:
            t = -(self.P[:, :, 0:3].transpose(1, 2) @ t[:, :, None]).squeeze(-1)
        return torch.max((t.mean(dim=0)[None] - t).norm(dim=1)).item()


class GS_model_with_param(torch.nn.Module):
    """Trainable scene (reference: gs_model.py:123-460).  Hyper-parameters the reference wraps in two nested
    parameter modules (:76-119) are plain floats here: nothing ever trains them."""

    def __init__(self, mean, variance_q, variance_scale, opacity, grad_delta_upper_limit=1e-12, grad_threshold=0.0004,
                 percent_dense=0.01, prunning_min_opacity=0.005, variance_pixel_tile_max_width=0.04,
                 position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
                 position_lr_max_steps=30_000, feature_lr=0.0025, opacity_lr=0.025, scaling_lr=0.005,
                 rotation_lr=0.001, c_00=1.77, L_max=2, lr=0.1, reference_layout=False):
        super().__init__()
        self.grad_delta_upper_limit, self.grad_threshold = grad_delta_upper_limit, grad_threshold
        self.percent_dense, self.prunning_min_opacity = percent_dense, prunning_min_opacity
        self.variance_pixel_tile_max_width = math.log(variance_pixel_tile_max_width / (1 - variance_pixel_tile_max_width))
        self.mean = torch.nn.Parameter(mean)
        self.variance_q = torch.nn.Parameter(variance_q)
        self.variance_scale = torch.nn.Parameter(variance_scale)
        self.opacity = torch.nn.Parameter(opacity)
        color = torch.zeros((mean.size(0), (L_max + 1) ** 2, 3), device=mean.device, dtype=torch.float32)
        color[:, 0, :] = c_00  # mid grey: 0.2821 * 1.77 = 0.5 (:156-158)
        self.color = torch.nn.Parameter(color)
        self.mean_lr_setfunc = get_expon_lr_func(position_lr_init, position_lr_final, lr_delay_steps=0,
                                                 lr_delay_mult=position_lr_delay_mult, max_steps=position_lr_max_steps)
        self.lr = {"mean": self.mean_lr_setfunc(0), "variance_q": rotation_lr, "variance_scale": scaling_lr,
                   "opacity": opacity_lr, "color": feature_lr}
        self._L_max = L_max
        self.reference_layout = reference_layout
        self.mean_grads_norm = torch.zeros(mean.shape[0], device=mean.device, dtype=torch.float32)
        self.mean_grads_iter = torch.zeros(mean.shape[0], device=mean.device, dtype=torch.int16)
        self.changing_optimizer()

    # ---- optimiser plumbing (reference: gs_model.py:43-67) -------------------------------------------------
    def changing_optimizer(self):
        groups = [{"params": p, "lr": float(self.lr[name])} for name, p in self.named_parameters(recurse=False)]
        # one multi-tensor kernel per step on the GPU instead of ~30 element-wise ones (0.95 -> 0.25 ms at 10^6 Gaussians)
        self._optimizer = torch.optim.Adam(groups, fused=True) if self.mean.is_cuda else torch.optim.Adam(groups)

    def set_mean_lr(self, iteration):
        """The reference rebuilds Adam (and drops its moments) every step to change one rate (gs_control.py:195-197);
        here the rate of the `mean` group is updated in place."""
        self.lr["mean"] = self.mean_lr_setfunc(iteration)
        for group, (name, _) in zip(self._optimizer.param_groups, self.named_parameters(recurse=False)):
            if name == "mean":
                group["lr"] = float(self.lr["mean"])

    def train_step(self):
        self._optimizer.step()
        self._optimizer.zero_grad(set_to_none=True)
        return self

    # ---- camera data parallelism (one process per GPU; the reference is single-GPU) ----------------------------
    def allreduce_grads(self, grad_iter=None, group=None):
        """Cameras of a batch are independent: each rank renders its share (`batch[rank::world]`, loss weighted by its
        share of the batch) and the parameter gradients are summed with ONE all-reduce of the concatenated N x 38
        floats (sharding.allreduce_gaussian_grads; RCCL over xGMI under backend "nccl").  `grad_iter` (seen by any
        camera) is OR-ed across ranks.  Every rank then takes the same optimiser step."""
        import torch.distributed as dist

        from . import sharding

        params = [p for _, p in self.named_parameters(recurse=False)]
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
        for p, g in zip(params, sharding.allreduce_gaussian_grads(*grads, group=group)):
            p.grad = g.contiguous()
        if grad_iter is not None:
            seen = grad_iter.to(torch.uint8)
            if seen.is_cuda and dist.get_backend(group) == "gloo":
                host = seen.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
                seen = host.to(grad_iter.device)
            else:
                dist.all_reduce(seen, op=dist.ReduceOp.MAX, group=group)
            grad_iter = seen.bool()
        return grad_iter

    # ---- densification statistics (:190-199) ----------------------------------------------------------------
    def param_iter_update(self, grad_iter):
        if self.mean.grad is not None:
            self.mean_grads_norm += self.mean.grad.norm(dim=1)
            self.mean_grads_iter += grad_iter.to(torch.int16)

    def param_grads_per_iter_norm(self):
        return self.mean_grads_norm / (self.mean_grads_iter + (self.mean_grads_iter == 0).int())

    def _replace(self, keep, extra=None):
        """Keep rows `keep` of every per-Gaussian tensor and append `extra` (dict name -> rows)."""
        names = ("mean", "variance_q", "variance_scale", "opacity", "color")
        for name in names:
            rows = getattr(self, name).data[keep]
            if extra is not None:
                rows = torch.cat((rows, extra[name]), dim=0)
            setattr(self, name, torch.nn.Parameter(rows))
        for name in ("mean_grads_norm", "mean_grads_iter"):
            rows = getattr(self, name)[keep]
            if extra is not None:
                rows = torch.cat((rows, extra[name]), dim=0)
            setattr(self, name, rows)

    def _rows(self, mask, repeat=1):
        out = {k: getattr(self, k).data[mask].repeat(repeat, *([1] * (getattr(self, k).dim() - 1)))
               for k in ("mean", "variance_q", "variance_scale", "opacity", "color")}
        out["mean_grads_norm"] = self.mean_grads_norm[mask].repeat(repeat)
        out["mean_grads_iter"] = self.mean_grads_iter[mask].repeat(repeat)
        return out

    def densify_and_split(self, scene_extent, N=2):
        """Large Gaussians with a big positional gradient are replaced by N samples of themselves (:201-227)."""
        scale = torch.exp(self.variance_scale.data)
        sel = (self.param_grads_per_iter_norm() >= self.grad_threshold) & (scale.max(dim=1).values > self.percent_dense * scene_extent)
        new = self._rows(sel, N)
        stds = scale[sel].repeat(N, 1)
        q = self.variance_q.data[sel] / torch.norm(self.variance_q.data[sel], dim=1, keepdim=True).clamp_min(1e-8)
        rots = qvec_to_rotmat_batch(q).repeat(N, 1, 1)
        new["mean"] = torch.bmm(rots, torch.normal(torch.zeros_like(stds), stds).unsqueeze(-1)).squeeze(-1) + new["mean"]
        new["variance_scale"] = torch.log(stds / (0.8 * N))
        self._replace(~sel, new)

    def densify_and_clone(self, scene_extent):
        """Small Gaussians with a big positional gradient are duplicated (:229-243)."""
        sel = (self.param_grads_per_iter_norm() >= self.grad_threshold) & (
            torch.exp(self.variance_scale.data).max(dim=1).values <= self.percent_dense * scene_extent)
        self._replace(torch.ones_like(sel), self._rows(sel))

    def densify_and_prune(self, extent):
        """(:245-265)"""
        self.densify_and_split(extent)
        self.densify_and_clone(extent)
        prune = (torch.sigmoid(self.opacity.data) < self.prunning_min_opacity).squeeze(1)
        prune |= torch.exp(self.variance_scale.data).max(dim=1).values > 0.1 * extent
        self._replace(~prune)
        self.changing_optimizer()

    def reset_opacity(self, reset_opacity):
        """(:267-271)"""
        cap = torch.full_like(self.opacity.data, reset_opacity)
        self.opacity = torch.nn.Parameter(torch.logit(torch.minimum(torch.sigmoid(self.opacity.data), cap)))
        self.changing_optimizer()

    # ---- forward (:277-460) ------------------------------------------------------------------------------------
    def camera_inputs(self, P, K, wh, fused=None):
        return camera_inputs(self.mean, self.variance_q, self.variance_scale, self.opacity, self.color, P, K, wh,
                             self.variance_pixel_tile_max_width, self._L_max, fused=fused)

    def forward(self, P, K, wh, image_sample):
        cams, grad_iter, (width, height) = self.camera_inputs(P, K, wh)
        images, names = [], []
        for cam, name in zip(cams, image_sample):
            if cam is None:
                continue  # nothing visible: the reference drops the image from the batch (:414-417)
            batch = cam["boxsize"].new_tensor([cam["boxsize"].numel()])
            images.append(custom_autograd_grouped_cumprod.apply(
                cam["boxsize"], batch, cam["startpoint"], cam["endpoint"], cam["mean"], cam["variance_inverse"],
                cam["opacity"], cam["l_d"], width, height))
            names.append(name)
        if not images:
            raise RuntimeError("no camera of the batch sees any Gaussian")  # the reference fails in torch.stack (:454)
        out = torch.stack(images, dim=0)[:, 1:, 1:, :]
        h, w = int(height), int(width)
        out = out.reshape(-1, 3, h, w) if self.reference_layout else out.permute(0, 3, 1, 2).contiguous()
        return [out, names, grad_iter]


def _gaussian_window(size, sigma, device, dtype):
    x = torch.arange(size, device=device, dtype=dtype) - (size - 1) / 2
    g = torch.exp(-(x * x) / (2 * sigma * sigma))
    return g / g.sum()


def ssim(img1, img2, window_size=11, max_val=1.0, sigma=1.5):
    """Structural-similarity map (B, C, H, W), Gaussian window, reflect padding — the quantity the reference takes
    from kornia (`metrics.ssim(..., max_val=1.0, window_size=11)`, gs_control.py:180).  kornia is not installed
    here, so this is the published formula (Wang et al. 2004), parity unpinned."""
    c = img1.shape[1]
    g = _gaussian_window(window_size, sigma, img1.device, img1.dtype)
    kx, ky = g.view(1, 1, 1, -1).expand(c, 1, 1, -1), g.view(1, 1, -1, 1).expand(c, 1, -1, 1)
    pad = window_size // 2

    def blur(t):
        t = torch.nn.functional.pad(t, (pad, pad, pad, pad), mode="reflect")
        return torch.nn.functional.conv2d(torch.nn.functional.conv2d(t, kx, groups=c), ky, groups=c)

    c1, c2 = (0.01 * max_val) ** 2, (0.03 * max_val) ** 2
    mu1, mu2 = blur(img1), blur(img2)
    s11, s22, s12 = blur(img1 * img1) - mu1 * mu1, blur(img2 * img2) - mu2 * mu2, blur(img1 * img2) - mu1 * mu2
    return ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2) + 1e-12)


_WINDOW_CACHE = {}


def _host_window(window_size, sigma):
    """The normalised window as a ctypes float array (host memory, handed to the loss kernels by value)."""
    import ctypes

    key = (window_size, sigma)
    if key not in _WINDOW_CACHE:
        w = _gaussian_window(window_size, sigma, "cpu", torch.float32)
        _WINDOW_CACHE[key] = (ctypes.c_float * window_size)(*w.tolist())
    return _WINDOW_CACHE[key]


class _SplatLoss(torch.autograd.Function):
    """(1 - lambda) L1 + lambda (1 - mean SSIM) in one HIP kernel per direction (csrc/gcp_loss.hip); differentiable in
    the first image only (the second is the target photograph)."""

    @staticmethod
    def forward(ctx, images, targets, lamda, max_val):
        if not images.is_cuda:
            raise RuntimeError("the fused loss is a HIP kernel: tensors must live on the GPU (no CPU path)")
        if images.shape != targets.shape or images.dim() != 4:
            raise RuntimeError("splat_loss expects two (B, C, H, W) tensors of one shape")
        a, b = images.detach().contiguous().float(), targets.detach().contiguous().float()
        bsz, ch, h, w = a.shape
        lib = _lib.load()
        win = _host_window(11, 1.5)
        need_grad = images.requires_grad
        maps = [torch.empty_like(a) for _ in range(3)] if need_grad else [None] * 3
        with torch.cuda.device(a.device):
            stream = torch.cuda.current_stream(a.device).cuda_stream
            partial = torch.empty(lib.gcp_ssim_blocks(bsz * ch, h, w), 2, dtype=torch.float32, device=a.device)
            _lib.check(lib.gcp_ssim_l1_forward(a.data_ptr(), b.data_ptr(), bsz * ch, h, w, win, (0.01 * max_val) ** 2,
                                               (0.03 * max_val) ** 2, *(m.data_ptr() if need_grad else None for m in maps),
                                               partial.data_ptr(), stream), "gcp_ssim_l1_forward")
        sums = partial.double().sum(dim=0) / a.numel()  # [mean SSIM, mean |a - b|]
        ctx.save_for_backward(a, b, *(maps if need_grad else []))
        ctx.lamda, ctx.dtype = lamda, images.dtype
        return ((1 - lamda) * sums[1] + lamda * (1 - sums[0])).to(images.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        a, b, m1, m2, m3 = ctx.saved_tensors
        n = a.numel()
        scales = (grad_out.reshape(1).float() * torch.tensor([-ctx.lamda / n, (1 - ctx.lamda) / n], device=a.device)).contiguous()
        grad = torch.empty_like(a)
        bsz, ch, h, w = a.shape
        with torch.cuda.device(a.device):
            stream = torch.cuda.current_stream(a.device).cuda_stream
            _lib.check(_lib.load().gcp_ssim_l1_backward(a.data_ptr(), b.data_ptr(), m1.data_ptr(), m2.data_ptr(), m3.data_ptr(), bsz * ch,
                                                        h, w, _host_window(11, 1.5), scales.data_ptr(), grad.data_ptr(), stream),
                       "gcp_ssim_l1_backward")
        return grad.to(ctx.dtype), None, None, None


def splat_loss(images, targets, lamda=0.2, fused=None):
    """(1 - lambda) L1 + lambda (1 - mean SSIM) (reference: gs_control.py:180-182).  On the GPU (`fused`, the default
    there) both terms and their gradient are one HIP kernel per direction; `fused=False` is the PyTorch formulation
    the kernel is tested against."""
    if fused is None:
        fused = images.is_cuda
    if fused:
        return _SplatLoss.apply(images, targets, float(lamda), 1.0)
    l1 = torch.nn.functional.l1_loss(images, targets, reduction="mean")
    return (1 - lamda) * l1 + lamda * (1 - ssim(images, targets, max_val=1.0, window_size=11).mean())
