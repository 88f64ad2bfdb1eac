"""Shared helpers for the parity tests: seeded inputs and the tolerance rule."""
import torch

TOL = 1e-5  # BASELINE.json north_star: "within 1e-5 fp32 on transmittance and colour"

DISTRIBUTIONS = ("all1", "poisson8", "geo80", "one_run", "runs3000", "runs9000", "mixed")


def make_keys(n, dist, seed):
    """int32 keys of length n whose runs follow `dist`; keys are NOT globally sorted
    on purpose for some distributions (the contract is runs of equal adjacent keys)."""
    g = torch.Generator().manual_seed(seed)
    if n == 0:
        return torch.zeros(0, dtype=torch.int32)
    if dist == "all1":
        lens = torch.ones(n, dtype=torch.long)
    elif dist == "poisson8":
        lens = torch.poisson(torch.full((n // 4 + 8,), 8.0), generator=g).long()
    elif dist == "geo80":
        u = torch.rand(n // 8 + 8, generator=g).clamp_(min=1e-12)
        lens = (torch.floor(torch.log(u) / torch.log1p(torch.tensor(-1.0 / 80.0))) + 1).long()
    elif dist == "one_run":
        lens = torch.tensor([n])
    elif dist == "runs3000":
        lens = torch.randint(2000, 4000, (n // 2000 + 2,), generator=g)
    elif dist == "runs9000":
        lens = torch.randint(5000, 13000, (n // 5000 + 2,), generator=g)
    elif dist == "mixed":
        a = torch.poisson(torch.full((n // 16 + 8,), 8.0), generator=g).long()
        b = torch.randint(1, 20000, (n // 16 + 8,), generator=g)
        pick = torch.rand(n // 16 + 8, generator=g) < 0.01
        lens = torch.where(pick, b, a)
    else:
        raise ValueError(dist)
    lens = torch.cat([lens[lens > 0], torch.tensor([n])])  # last entry guarantees coverage
    csum = torch.cumsum(lens, 0)
    k = int(torch.searchsorted(csum, torch.tensor(n)).item()) + 1
    lens = lens[:k].clone()
    lens[-1] -= int(csum[k - 1].item()) - n
    assert int(lens.sum()) == n and int(lens.min()) > 0
    # alternate a small set of key values so equal keys recur in non-adjacent runs
    ids = torch.arange(lens.numel())
    vals = ((ids * 7919) % 1000 + (ids % 2) * 1000003).to(torch.int32)
    return torch.repeat_interleave(vals, lens)


def make_values(n, seed, kind="alpha"):
    g = torch.Generator().manual_seed(seed + 1)
    if kind == "alpha":  # 1 - a*g with a ~ sigmoid(N(1.7,2)) as in BASELINE.md
        a = torch.sigmoid(torch.randn(n, generator=g) * 2.0 + 1.7).clamp_(0.005, 0.995)
        return 1.0 - a * torch.rand(n, generator=g)
    if kind == "near1":  # slow decay: long products stay O(1)
        return 1.0 - 1e-3 * torch.rand(n, generator=g)
    if kind == "normal":
        return torch.randn(n, generator=g)
    raise ValueError(kind)


def assert_parity(got, want32, scale, what="", tol=TOL):
    """The tolerance rule of the parity tests.

    scale=None — transmittance, colour and every other product of factors in [0, 1] (row a1, the T of a5, the image):
      |got - want32| <= tol ABSOLUTE, BASELINE.json north_star's "within 1e-5 fp32 on transmittance and colour" to the letter.
    scale given — sums and gradients, which are not bounded by 1: |got - want32| <= tol * (1 + scale) elementwise, `scale`
      being the magnitude of the quantity with every term taken in absolute value (fp64): the condition scale of the sum."""
    got = got.detach().cpu().double()
    want = want32.detach().cpu().double()
    scale = torch.zeros_like(want) if scale is None else scale.detach().cpu().double().abs()
    err = (got - want).abs()
    bound = tol * (1.0 + scale)
    bad = err > bound
    if bad.any():
        i = int(torch.nonzero(bad)[0])
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.numel()} outside {tol:g}*(1+scale); first at {i}: "
            f"got {got[i].item():.9g} want {want[i].item():.9g} scale {scale[i].item():.4g}; "
            f"max err {err.max().item():.3g}"
        )


def make_scene(n_gauss, width, height, max_half, seed, opacity_one_every=0):
    """Gaussians of one camera in depth order, as `custom_autograd_grouped_cumprod` receives them
    (reference: gs_model.py:419-425, :449): integer means, inclusive integer boxes clamped to the image."""
    g = torch.Generator().manual_seed(seed)
    mean = torch.stack(
        [torch.randint(0, width + 1, (n_gauss,), generator=g), torch.randint(0, height + 1, (n_gauss,), generator=g)], 1
    ).to(torch.int32)
    half = torch.randint(1, max_half + 1, (n_gauss, 2), generator=g).to(torch.int32)
    lim = torch.tensor([width, height], dtype=torch.int32)
    start = torch.minimum((mean - half).clamp(min=0), lim)
    end = torch.minimum((mean + half).clamp(min=0), lim)
    boxsize = torch.prod((end - start + 1).to(torch.int64), dim=1)
    sx = 0.6 + 0.35 * max_half * torch.rand(n_gauss, generator=g)
    sy = 0.6 + 0.35 * max_half * torch.rand(n_gauss, generator=g)
    rho = 0.8 * (torch.rand(n_gauss, generator=g) - 0.5)
    cov = torch.stack([sx * sx, rho * sx * sy, rho * sx * sy, sy * sy], 1).reshape(-1, 2, 2)
    vinv = torch.linalg.inv(cov).to(torch.float32).contiguous()
    opacity = (0.05 + 0.9 * torch.rand(n_gauss, 1, generator=g)).to(torch.float32)
    if opacity_one_every:
        opacity[::opacity_one_every] = 1.0  # alpha == 1 at the centre pixel: inclusive product exactly 0
    l_d = (0.05 + 0.95 * torch.rand(n_gauss, 3, generator=g)).to(torch.float32)
    wimg = torch.randn(height + 1, width + 1, 3, generator=g)
    return dict(boxsize=boxsize, start=start, end=end, mean=mean, vinv=vinv, opacity=opacity, l_d=l_d, wimg=wimg,
                width=width, height=height)


def full_cover_scales(sc, n):
    """Scenes whose boxes all cover the whole image: the per-(Gaussian, pixel) terms have a closed dense form (fp64), and
    with them the CONDITION SCALE of every gradient — the sum over pixels of the |terms| whose signed sum it is — so a
    bound can be stated relative to the Gaussian's own terms instead of the scene's largest gradient.
    Returns (scales per gradient name, exclusive transmittance T[n, P])."""
    w, h = sc["width"], sc["height"]
    ys, xs = torch.meshgrid(torch.arange(h + 1, dtype=torch.float64), torch.arange(w + 1, dtype=torch.float64), indexing="ij")
    dx = xs.reshape(1, -1) - sc["mean"][:, 0:1].double()
    dy = ys.reshape(1, -1) - sc["mean"][:, 1:2].double()
    v = sc["vinv"].double()
    gk = torch.exp(-0.5 * (dx * dx * v[:, 0, 0, None] + dx * dy * (v[:, 0, 1, None] + v[:, 1, 0, None]) + dy * dy * v[:, 1, 1, None]))
    a = sc["opacity"].double() * gk                                                   # [n, P]
    T = torch.cumprod(torch.cat([torch.ones(1, a.size(1), dtype=torch.float64), 1.0 - a[:-1]]), 0)   # exclusive
    c = sc["l_d"].double() @ sc["wimg"].double().reshape(-1, 3).T                      # dL/dI . l   [n, P]
    S = torch.flip(torch.cumsum(torch.flip(T * a * c, [0]), 0), [0]) - T * a * c       # exclusive suffix sums
    sa = S / (1.0 - a)
    abs_o = (T * gk * c.abs() + gk * sa.abs()).sum(1)                                  # scale of dL/do (gs_model.py:733-740)
    abs_c = T * a * c.abs() + a * sa.abs()                                             # scale of the "common" factor
    abs_l = (T * a)[:, :, None] * sc["wimg"].double().reshape(1, -1, 3).abs()          # [n, P, 3]
    scales = {"opacity": abs_o[:, None], "l_d": abs_l.sum(1),
              "vinv": 0.5 * torch.stack([(abs_c * dx * dx).sum(1), (abs_c * (dx * dy).abs()).sum(1), (abs_c * (dx * dy).abs()).sum(1),
                                         (abs_c * dy * dy).sum(1)], 1)}
    return scales, T
