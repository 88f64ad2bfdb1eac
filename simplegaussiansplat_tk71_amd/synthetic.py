"""Synthetic depth-sorted splat-pixel pair lists.

The scan never sees Gaussians, only the flat pair arrays the reference builds in
`_create_alpha_brend` after its sort (reference: gs_model.py:546-548):
  key      int32[M]  y*10000 + x (gs_model.py:538-541), sorted row-major, one run per pixel
  x        f32[M]    1 - alpha * G  ("anti opacity", gs_model.py:533-535)
  inv      int32[M]  dense group id 0..G-1          (cuda_test.py:21)
  inv_len  int32[G]  exclusive end offset per group (cuda_test.py:27)
  grad_out f32[M]    N(0,1)

Named workloads = BASELINE.json `configs`:
  cfg1  256x256,    D=8,   Poisson           (CPU plumbing case)
  cfg2  1920x1080,  D=8,   Poisson           (100k Gaussians)
  cfg3  1920x1080,  D=80,  deep heavy tail   (1M Gaussians)   <- the metric's configuration
  cfg5  3840x2160,  D=100, deep heavy tail   (5M Gaussians, 8 GPUs)
"""
from dataclasses import dataclass

import torch

CONFIGS = {
    "cfg1": dict(height=256, width=256, mean_depth=8.0, deep=False, gaussians=2_000),
    "cfg2": dict(height=1080, width=1920, mean_depth=8.0, deep=False, gaussians=100_000),
    "cfg3": dict(height=1080, width=1920, mean_depth=80.0, deep=True, gaussians=1_000_000),
    "cfg5": dict(height=2160, width=3840, mean_depth=100.0, deep=True, gaussians=5_000_000),
}

MAX_RUN = 4096


@dataclass
class PairList:
    key: torch.Tensor
    x: torch.Tensor
    inv: torch.Tensor
    inv_len: torch.Tensor
    grad_out: torch.Tensor
    run_len: torch.Tensor  # int64[P] splats per pixel (zeros allowed)
    height: int
    width: int

    @property
    def n_pairs(self):
        return self.key.numel()

    @property
    def n_groups(self):
        return self.inv_len.numel()


def run_lengths(n_pixels, mean_depth, deep, generator, device, max_run=MAX_RUN):
    """Splats per pixel.  Poisson(D), or for "deep per-pixel lists" the mix
    0.9*Poisson(D/2) + 0.1*Geometric(mean 5.5*D) (same mean D), clipped at `max_run` (4096 = SURVEY.md §8d; None = no
    clip: the "cfg3_unclipped" workload, whose few pixels deeper than 4096 exercise the scans' descriptor tree)."""
    if max_run is None:
        max_run = 1 << 30
    if not deep:
        lam = torch.full((n_pixels,), float(mean_depth), device=device)
        return torch.poisson(lam, generator=generator).long().clamp_(max=max_run)
    lam = torch.full((n_pixels,), float(mean_depth) / 2.0, device=device)
    body = torch.poisson(lam, generator=generator)
    u = torch.rand(n_pixels, device=device, generator=generator)
    p = 1.0 / (5.5 * float(mean_depth))
    v = torch.rand(n_pixels, device=device, generator=generator).clamp_(min=1e-12)
    tail = torch.floor(torch.log(v) / torch.log1p(torch.tensor(-p, device=device))) + 1.0
    return torch.where(u < 0.1, tail, body).long().clamp_(min=0, max=max_run)


def pixel_keys(height, width, device, row_start=0):
    """The reference's pixel key y*10000 + x (gs_model.py:538-541) of every pixel of a band, row-major."""
    ys = torch.arange(row_start, row_start + height, device=device, dtype=torch.int32)
    xs = torch.arange(width, device=device, dtype=torch.int32)
    return (ys[:, None] * 10000 + xs[None, :]).reshape(-1)


def pairs_from_runs(L, pixel_key, generator, height, width):
    """Pair list of the pixels `pixel_key` with `L` splats each (zeros allowed): keys, dense group ids, end offsets and
    seeded values."""
    device = L.device
    key = torch.repeat_interleave(pixel_key, L)
    m = key.numel()
    nz = L > 0
    inv_len = torch.cumsum(L[nz], 0).to(torch.int32)
    inv = torch.repeat_interleave(torch.arange(int(nz.sum()), device=device, dtype=torch.int32), L[nz])
    # opacity ~ sigmoid(N(1.7, 2.0)) clipped to [0.005, 0.995]: quantiles of the reference's opacity.pt
    a = torch.sigmoid(torch.randn(m, device=device, generator=generator) * 2.0 + 1.7).clamp_(0.005, 0.995)
    gk = torch.rand(m, device=device, generator=generator)
    x = 1.0 - a * gk
    grad_out = torch.randn(m, device=device, generator=generator)
    return PairList(key, x, inv, inv_len, grad_out, L, height, width)


def make_pairs(height, width, mean_depth, deep=False, seed=0, device="cpu", row_start=0, gaussians=None, max_run=MAX_RUN):
    """Pair list of an image band of `height` rows starting at image row `row_start`."""
    del gaussians  # enters only through mean_depth
    device = torch.device(device)
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    L = run_lengths(height * width, mean_depth, deep, g, device, max_run)
    return pairs_from_runs(L, pixel_keys(height, width, device, row_start), g, height, width)


def make_config(name, seed=0, device="cpu", rows=None, row_start=0, max_run=MAX_RUN):
    """One of BASELINE.json's configs; `rows` restricts it to an image band."""
    c = CONFIGS[name]
    h = c["height"] if rows is None else rows
    return make_pairs(h, c["width"], c["mean_depth"], c["deep"], seed, device, row_start, max_run=max_run)


def make_config_slice(name, world_size, rank, seed=0, device="cpu", max_run=MAX_RUN):
    """Rank `rank`'s share of ONE frame of config `name` cut into `world_size` contiguous slices at the pixel-group
    boundaries nearest k*M/R (sharding.partition_groups: balance by pairs, not by rows — SURVEY.md §8e).  Every rank
    draws the same per-pixel run lengths (same seed), so all ranks agree on the cut without communicating; only the
    owned slice's pair arrays are materialised.  Returns (PairList of the slice with ids / offsets rebased to it, the
    Shard table, total pairs of the frame)."""
    from . import sharding

    c = CONFIGS[name]
    device = torch.device(device)
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    h, w = c["height"], c["width"]
    L = run_lengths(h * w, c["mean_depth"], c["deep"], g, device, max_run)
    nz = torch.nonzero(L > 0).flatten()
    ends = torch.cumsum(L[nz], 0)
    shards = sharding.partition_groups(ends, world_size)
    sh = shards[rank]
    own = nz[sh.group_start : sh.group_end]
    gl = torch.Generator(device=device)
    gl.manual_seed(int(seed) * 1000003 + 17 * rank + 1)
    p = pairs_from_runs(L[own], pixel_keys(h, w, device)[own], gl, h, w)
    assert p.n_pairs == sh.n_pairs and p.n_groups == sh.n_groups
    return p, shards, int(ends[-1].item()) if ends.numel() else 0


def make_scene(n_gauss, width, height, mean_depth, seed=0, device="cpu"):
    """Function-level synthetic input: Gaussians of one camera in depth order
    (= index order) as `custom_autograd_grouped_cumprod` receives them (reference: gs_model.py:419-425,
    :449).  Integer centres uniform in the image; half-sizes drawn so that the mean box area is
    P*D/N pixels, i.e. the pair list has mean depth D."""
    device = torch.device(device)
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    p = (width + 1) * (height + 1)
    area = max(1.0, p * float(mean_depth) / n_gauss)
    side = area ** 0.5                      # mean (2h+1)
    hmean = max(0.5, (side - 1.0) / 2.0)
    half = (torch.rand(n_gauss, 2, device=device, generator=g) * 2.0 * hmean).round().to(torch.int32)
    mean = torch.stack([torch.randint(0, width + 1, (n_gauss,), device=device, generator=g),
                        torch.randint(0, height + 1, (n_gauss,), device=device, generator=g)], 1).to(torch.int32)
    lim = torch.tensor([width, height], dtype=torch.int32, device=device)
    start = torch.minimum((mean - half).clamp(min=0), lim)
    end = torch.minimum((mean + half).clamp(min=0), lim)
    boxsize = torch.prod((end - start + 1).to(torch.int64), dim=1)
    sig = (0.6 + 0.45 * half.float() * (0.5 + torch.rand(n_gauss, 2, device=device, generator=g)))
    rho = 0.8 * (torch.rand(n_gauss, device=device, generator=g) - 0.5)
    sx, sy = sig[:, 0], sig[:, 1]
    det = (sx * sy) ** 2 * (1 - rho * rho)
    vinv = torch.stack([sy * sy / det, -rho * sx * sy / det, -rho * sx * sy / det, sx * sx / det], 1).reshape(-1, 2, 2)
    opacity = torch.sigmoid(torch.randn(n_gauss, 1, device=device, generator=g) * 2.0 + 1.7).clamp_(0.005, 0.995)
    l_d = 0.05 + 0.95 * torch.rand(n_gauss, 3, device=device, generator=g)
    return dict(boxsize=boxsize, start=start, end=end, mean=mean, vinv=vinv.contiguous(), opacity=opacity, l_d=l_d,
                width=width, height=height)


def make_scene_config(name, seed=0, device="cpu"):
    c = CONFIGS[name]
    return make_scene(c["gaussians"], c["width"] - 1, c["height"] - 1, c["mean_depth"], seed, device)


def make_scene_pairs(name, seed=0, device="cuda"):
    """The pair list the reference's Function builds for one camera of a synthetic scene of config `name`, as its
    `_create_alpha_brend` / `grad_cumsum` receive it (gs_model.py:601-607, :630-636): Gaussian-major rects int32[M,2]
    (uitility.py:336-366), the anti-opacity 1 - o*g of every pair (gs_model.py:533-535) and a gradient per pair.
    Returns (scene dict, rects, anti_opacity, grad).  Needs the HIP library (rect expansion runs on the device)."""
    from . import raster

    sc = make_scene_config(name, seed=seed, device=device)
    rects, owner = raster.expand_rects(sc["start"], sc["end"], sc["width"], sc["height"], with_gaussian=True)
    g = torch.Generator(device=rects.device).manual_seed(int(seed) + 1)
    gk = torch.rand(rects.size(0), device=rects.device, generator=g)
    anti = 1.0 - sc["opacity"].reshape(-1)[owner.long()] * gk
    grad = torch.randn(rects.size(0), device=rects.device, generator=g)
    return sc, rects, anti, grad


def ring_cameras(n_cam, width, height, radius=3.2, device="cpu"):
    """World->camera [R|t] (n,3,4), intrinsics (n,3,3) and (n,2) sizes of cameras on a ring looking at the origin
    (COLMAP convention: x right, y down, z forward) — the camera side of the caller tests and benches (row f4)."""
    import math

    P, K = [], []
    for c in range(n_cam):
        ang = 2 * math.pi * c / n_cam + 0.3
        eye = torch.tensor([radius * math.cos(ang), 0.5 * math.sin(2 * ang), radius * math.sin(ang)])
        fwd = -eye / eye.norm()
        right = torch.linalg.cross(torch.tensor([0.0, -1.0, 0.0]), fwd)
        right = right / right.norm()
        R = torch.stack([right, torch.linalg.cross(fwd, right), fwd])
        P.append(torch.cat([R, (-R @ eye)[:, None]], dim=1))
        K.append(torch.tensor([[0.9 * width, 0, width / 2], [0, 0.9 * width, height / 2], [0, 0, 1.0]]))
    wh = torch.tensor([[width, height]] * n_cam, dtype=torch.float32)
    return torch.stack(P).to(device), torch.stack(K).to(device), wh.to(device)


def make_world(n_gauss, width, sigma_px=2.0, seed=0, device="cpu", radius=3.2):
    """3-D Gaussians for `ring_cameras`: a blob around the origin whose members project to about `sigma_px` pixels
    (cfg3-like at 10^6 Gaussians, 1920x1080, sigma_px = 2: ~0.9e6 visible, ~1.6e8 splat-pixel pairs per camera).
    Returns the parameter tensors of gs_model.GS_model_with_param: mean, variance_q, variance_scale, opacity (logit)."""
    g = torch.Generator().manual_seed(seed)
    mean = torch.randn(n_gauss, 3, generator=g) * torch.tensor([0.9, 0.5, 0.6])
    sigma_world = sigma_px * radius / (0.9 * width)
    scale = torch.log(sigma_world * (0.6 + 0.8 * torch.rand(n_gauss, 3, generator=g)))
    q = torch.randn(n_gauss, 4, generator=g)
    op = torch.logit(0.05 + 0.9 * torch.rand(n_gauss, 1, generator=g))
    return [t.to(device) for t in (mean, q, scale, op)]
