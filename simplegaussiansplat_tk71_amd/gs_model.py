"""Caller of the hot path: 3-D Gaussians + cameras -> the inputs of the rasterise-and-blend Function -> images.

Mirrors the reference's model class (reference: gs_model.py:123-460, `GS_model_with_param`): same parameter set
(mean, variance_q, variance_scale, opacity, color), same `forward(P, K, wh, image_sample)` return value
`[images, image_sample, grad_iter]`, same per-parameter Adam learning rates, densify / prune / opacity reset.
Row f4 of SURVEY.md §8: this is caller integration, PyTorch on the GPU around the HIP Function — not a kernel.

The projection (`camera_inputs`, reference: gs_model.py:277-425) is pinned, through its PyTorch restatement in
oracle/gs_forward_torch.py, against the reference's own forward run on CPU (tests/golden/forward_golden.npz: the
arguments the reference hands to `custom_autograd_grouped_cumprod.apply`).  Differences, all deliberate:
  * the 3-sigma box comes from a closed-form 2x2 eigen-decomposition on the device; the reference moves every
    covariance to the CPU for `torch.linalg.eigh` and back (gs_model.py:327-332).  For a positive semi-definite
    matrix `V^2 |lambda|` is just its diagonal, so the box is 3*sqrt(diag) exactly;
  * the depth sort is stable (the reference's `torch.argsort`, :356, leaves ties undefined);
  * images are permuted to (B, 3, H, W); the reference `reshape`s (H, W, 3) memory into (3, H, W), scrambling
    channels (gs_model.py:454) — `reference_layout=True` reproduces that;
  * one Function call per camera, never chunked (nothing of pair-list size exists here; gs_model.py:428);
  * the SH colour stands in for the reference's `sh_utility.eval_sh`, which is not in its checkout
    (gs_model.py:9,335): real spherical harmonics up to degree 2 in the usual 3DGS convention — parity unpinned;
  * tensors live on the parameters' device instead of a hard-coded "cuda";
  * on the GPU the whole per-Gaussian chain is ONE HIP kernel per camera and direction (`gcp_project_forward`,
    `gcp_project_backward`, csrc/gcp_project.hip) instead of ~150 PyTorch kernels: at 10^6 Gaussians the reference's
    formulation costs 64 ms forward + 110 ms backward around a 1.8 ms Function.  That formulation is kept, as the
    checker the kernels are tested against, in oracle/gs_forward_torch.py — not here: projection and loss have no
    CPU path (CPU tensors raise).
"""
import math

import torch

from . import _lib
from . import raster as _raster
from .cuda_kernel import custom_autograd_grouped_cumprod

__all__ = [
    "GS_dataset",
    "GS_model_with_param",
    "HipAdam",
    "camera_inputs",
    "qvec_to_rotmat_batch",
    "get_expon_lr_func",
    "mean_neighbour_distance",
    "splat_loss",
]

def qvec_to_rotmat_batch(q):
    """(N, 4) unit quaternions in (x, y, z, w) order -> (N, 3, 3) (reference: uitility.py:231-254)."""
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    r0 = torch.stack([1 - 2 * (y**2 + z**2), 2 * (x * y - w * z), 2 * (x * z + w * y)], dim=1)
    r1 = torch.stack([2 * (x * y + w * z), 1 - 2 * (x**2 + z**2), 2 * (y * z - w * x)], dim=1)
    r2 = torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x**2 + y**2)], dim=1)
    return torch.stack([r0, r1, r2], dim=1)


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


_CLAMP_CACHE = {}


def _box_clamp(width, height, tile_max_width):
    """Upper bound of the 3-sigma half extents: 10 * sqrt(W*H) * sigmoid(tile_max_width) in float32, as the reference forms
    it (gs_model.py:364-365).  Evaluated once per (W, H, setting) on the host: no device work, no read-back."""
    key = (width, height, float(tile_max_width))
    if key not in _CLAMP_CACHE:
        t = torch.sqrt(torch.tensor(width * height, dtype=torch.int32).to(torch.float32)) * torch.sigmoid(
            torch.tensor(float(tile_max_width), dtype=torch.float32))
        _CLAMP_CACHE[key] = (t * 10).item()
    return _CLAMP_CACHE[key]


class _ProjectCamera(torch.autograd.Function):
    """One camera of `camera_inputs` on the HIP library (csrc/gcp_project.hip): gcp_project_forward, the library's
    stable radix sort on the depth keys, gcp_project_gather; backward = gcp_project_backward."""

    @staticmethod
    def forward(ctx, mean, variance_q, variance_scale, opacity, color, cam_P, cam_K, width, height, box_clamp, L_max,
                capture_safe=False):
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
            # the one device->host read: sizes of the outputs.  capture_safe: none — the list keeps all n Gaussians, the
            # culled ones behind the kept ones with empty boxes (gcp_project_gather with the keep mask)
            m = n if capture_safe else (int(keep.sum()) if n else 0)
            # culled Gaussians carry the largest key: the first m entries of the stable permutation are the kept ones in
            # depth order, ties in the Gaussians' own order
            perm = _raster.stable_sort_keys(sort_key, key_bits=31)[1] if n else sort_key
            start, end, mean_xy, boxsize = i32(m, 2), i32(m, 2), i32(m, 2), torch.empty(m, dtype=torch.int64, device=dev)
            vinv, alpha, l_d, index = f32(m, 2, 2), f32(m, 1), f32(m, 3), torch.empty(m, dtype=torch.int64, device=dev)
            _lib.check(lib.gcp_project_gather(record.data_ptr(), perm.data_ptr(), m, start.data_ptr(), end.data_ptr(),
                                              mean_xy.data_ptr(), boxsize.data_ptr(), vinv.data_ptr(), alpha.data_ptr(),
                                              l_d.data_ptr(), index.data_ptr(), row_of.data_ptr(),
                                              keep.data_ptr() if capture_safe else None, stream), "gcp_project_gather")
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
        return (*grads, None, None, None, None, None, None, None)


def camera_inputs(mean, variance_q, variance_scale, opacity, color, P, K, wh, tile_max_width, L_max=2, capture_safe=False):
    """Per camera, the depth-ordered, culled arguments of the Function (reference: gs_model.py:277-425).

    mean (N,3), variance_q (N,4 xyzw), variance_scale (N,3 log), opacity (N,1 logit), color (N,(L+1)^2,3),
    P (C,3,4) world->camera, K (C,3,3), wh (C,2), tile_max_width = logit of the box clamp as a fraction of
    sqrt(W*H)/10.  Returns a list with one dict per camera (None where nothing is visible, :414-417) holding
    boxsize, startpoint, endpoint, mean, variance_inverse, opacity, l_d, index (Gaussian ids, depth order),
    and the (N,) bool `grad_iter` of Gaussians seen by any camera (:401-407).

    One HIP kernel per camera and direction (csrc/gcp_project.hip); GPU tensors only — there is no CPU path.  The
    reference's op-by-op PyTorch formulation lives in oracle/gs_forward_torch.py as the checker.

    capture_safe=True: no device->host read at all (pass `wh` as a CPU tensor or a list): every camera's list keeps all N
    Gaussians in depth order, the culled ones behind the kept ones with EMPTY boxes (binned into no tile, zero
    gradients), and no camera is ever dropped; images and gradients are those of the default mode.  Together with
    `cuda_kernel.tile_capacity` the projection + Function forward and backward queue without waiting for the GPU."""
    width, height = (int(v) for v in (wh[0].tolist() if isinstance(wh, torch.Tensor) else wh[0]))  # device `wh`: one read (.to(int32) truncates, :279)
    clamp = _box_clamp(width, height, tile_max_width)
    grad_iter = None
    cams = []
    for c in range(P.shape[0]):
        vinv, alpha, l_d, start, end, mean_xy, boxsize, index, keep = _ProjectCamera.apply(
            mean, variance_q, variance_scale, opacity, color, P[c], K[c], width, height, clamp, L_max, capture_safe)
        grad_iter = keep if grad_iter is None else grad_iter | keep
        cams.append(None if index.numel() == 0 else {
            "boxsize": boxsize, "startpoint": start, "endpoint": end, "mean": mean_xy, "variance_inverse": vinv,
            "opacity": alpha, "l_d": l_d, "index": index})
    if grad_iter is None:
        grad_iter = torch.zeros(mean.shape[0], device=mean.device, dtype=torch.bool)
    return cams, grad_iter, (width, height)


class HipAdam:
    """torch.optim.Adam's update (default betas / eps, no weight decay, no amsgrad — what the reference constructs at
    gs_model.py:43-47) on the HIP library: one streaming kernel per parameter tensor (csrc/gcp_optim.hip, gcp_adam_step)
    instead of torch's multi-tensor launches (0.38 -> 0.2 ms per step at 10^6 Gaussians).  Same interface as far as the
    model uses it: `param_groups` with one tensor and an `lr` each, `step()`, `zero_grad()`."""

    def __init__(self, param_groups, betas=(0.9, 0.999), eps=1e-8):
        self.param_groups = [dict(g, params=list(g["params"]) if isinstance(g["params"], (list, tuple)) else [g["params"]])
                             for g in param_groups]
        self.betas, self.eps = betas, eps
        self.state = {}

    @torch.no_grad()
    def step(self):
        lib = _lib.load()
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("HipAdam updates contiguous float32 GPU tensors")
                st = self.state.setdefault(p, {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)})
                st["step"] += 1
                grad = p.grad.contiguous()
                with torch.cuda.device(p.device):
                    _lib.check(lib.gcp_adam_step(p.data_ptr(), grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                                 p.numel(), float(group["lr"]), self.betas[0], self.betas[1], self.eps, st["step"],
                                                 torch.cuda.current_stream(p.device).cuda_stream), "gcp_adam_step")

    def zero_grad(self, set_to_none=True):
        for group in self.param_groups:
            for p in group["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()


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
        # on the GPU one streaming kernel per tensor (torch's per-op Adam: 0.95 ms per step at 10^6 Gaussians, its fused
        # multi-tensor form 0.38, HipAdam 0.2)
        self._optimizer = HipAdam(groups) if self.mean.is_cuda else torch.optim.Adam(groups)

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
    def camera_inputs(self, P, K, wh):
        return camera_inputs(self.mean, self.variance_q, self.variance_scale, self.opacity, self.color, P, K, wh,
                             self.variance_pixel_tile_max_width, self._L_max)

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


_WINDOW_CACHE = {}


def _host_window(window_size, sigma):
    """The normalised window as a ctypes float array (host memory, handed to the loss kernels by value)."""
    import ctypes

    key = (window_size, sigma)
    if key not in _WINDOW_CACHE:
        x = torch.arange(window_size, dtype=torch.float32) - (window_size - 1) / 2
        w = torch.exp(-(x * x) / (2 * sigma * sigma))
        _WINDOW_CACHE[key] = (ctypes.c_float * window_size)(*(w / w.sum()).tolist())
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


def splat_loss(images, targets, lamda=0.2):
    """(1 - lambda) L1 + lambda (1 - mean SSIM), 11-tap Gaussian window, reflect padding (reference:
    gs_control.py:180-182, kornia.metrics.ssim).  Both terms and their gradient are one HIP kernel per direction
    (csrc/gcp_loss.hip); GPU tensors only.  The PyTorch formulation it is tested against: oracle/loss_torch.py."""
    return _SplatLoss.apply(images, targets, float(lamda), 1.0)
