"""Rows f1/f2: tile binning + fused blend (csrc/gcp_raster.hip) vs the reference's own Function outputs
(tests/golden/function_golden.npz) and vs the dense autograd oracle (oracle/dense_render.py)."""
import os

import numpy as np
import pytest
import torch

from tests.util import TOL, full_cover_scales, make_scene

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _apply(device, sc):
    import cuda_kernel as ck

    vinv = sc["vinv"].to(device).requires_grad_(True)
    op = sc["opacity"].to(device).requires_grad_(True)
    l_d = sc["l_d"].to(device).requires_grad_(True)
    n = sc["start"].size(0)
    img = ck.custom_autograd_grouped_cumprod.apply(
        sc["boxsize"].to(device), torch.tensor([n], device=device), sc["start"].to(device), sc["end"].to(device),
        sc["mean"].to(device), vinv, op, l_d, torch.tensor(sc["width"], dtype=torch.int32),
        torch.tensor(sc["height"], dtype=torch.int32),
    )
    (img * sc["wimg"].to(device)).sum().backward()
    return img.detach().cpu(), vinv.grad.cpu(), op.grad.cpu(), l_d.grad.cpu()


def _close(got, want, scale_ref, what, tol=TOL):
    """|got - want| <= tol * (1 + |want| + typical magnitude): gradients are sums over thousands of pairs."""
    err = (got.double() - want.double()).abs()
    bound = tol * (1.0 + want.double().abs() + scale_ref)
    assert bool((err <= bound).all()), f"{what}: max err {err.max().item():.3g}, bound {bound.min().item():.3g}"


@pytest.mark.parametrize("name", ["fn_6g_16x12", "fn_200g_64x48", "fn_1500g_128x96"])
def test_function_vs_reference_golden(device, name):
    """Same inputs the reference's own custom_autograd_grouped_cumprod was run on (CPU, single chunk)."""
    z = np.load(os.path.join(GOLD, "function_golden.npz"))
    g = lambda k: torch.from_numpy(z[f"{name}/{k}"])  # noqa: E731
    w, h = (int(v) for v in z[name + "/width_height"])
    sc = dict(boxsize=g("boxsize"), start=g("start"), end=g("end"), mean=g("mean"), vinv=g("vinv"), opacity=g("opacity"),
              l_d=g("l_d"), wimg=g("wimg"), width=w, height=h)
    img, gv, go, gl = _apply(device, sc)
    torch.testing.assert_close(img, g("image"), atol=TOL, rtol=TOL)
    _close(go, g("grad_opacity"), g("grad_opacity").abs().mean().item(), "grad_opacity vs reference")
    _close(gv, g("grad_vinv"), g("grad_vinv").abs().mean().item(), "grad_vinv vs reference")
    # colour gradient: the reference's is known-buggy; ours must be the true one (dense autograd oracle)
    from oracle import dense_render as dr

    _, _, _, gl64 = dr.render_with_grads(sc["start"], sc["end"], sc["mean"], sc["vinv"], sc["opacity"], sc["l_d"], w, h, sc["wimg"])
    _close(gl, gl64, gl64.abs().mean().item(), "grad_l vs dense oracle")


@pytest.mark.parametrize("n_gauss,w,h,mh,seed,one", [(1, 8, 8, 2, 1, 0), (40, 33, 17, 4, 2, 0), (400, 100, 70, 9, 3, 0),
                                                      (300, 64, 64, 20, 4, 0), (250, 50, 40, 6, 5, 7), (2500, 255, 191, 10, 6, 0),
                                                      (900, 48, 40, 24, 8, 0)])  # ~800 entries per tile: several staging rounds
def test_function_vs_dense_oracle(device, n_gauss, w, h, mh, seed, one):
    """Random scenes incl. boxes spanning many tiles, tile lists longer than one staging round of either kernel (256
    forward, 32 backward; per-wave hit words of 64 entries) and (one=7) opacity-1 Gaussians whose centre pixel has an
    inclusive product of exactly 0 (dropped pair, gs_model.py:560)."""
    from oracle import dense_render as dr

    sc = make_scene(n_gauss, w, h, mh, seed, one)
    img, gv, go, gl = _apply(device, sc)
    i64, gv64, go64, gl64 = dr.render_with_grads(sc["start"], sc["end"], sc["mean"], sc["vinv"], sc["opacity"], sc["l_d"], w, h, sc["wimg"])
    if one:
        assert bool((sc["opacity"] == 1.0).any())
    torch.testing.assert_close(img.double(), i64, atol=TOL, rtol=TOL)
    _close(go, go64, go64.abs().mean().item(), "grad_opacity")
    _close(gv, gv64, gv64.abs().mean().item(), "grad_vinv")
    _close(gl, gl64, gl64.abs().mean().item(), "grad_l")


def test_tile_lists_are_depth_ordered_and_complete(device):
    from simplegaussiansplat_tk71_amd import raster

    sc = make_scene(500, 90, 75, 12, 9)
    bins = raster.bin_tiles(sc["start"].to(device), sc["end"].to(device), 90, 75)
    ts, tl, toff = bins.tile_start.cpu(), bins.tile_list.cpu(), bins.tile_off.cpu()
    assert bins.tiles_x == (90 + 1 + 15) // 16 and bins.tiles_y == (75 + 1 + 15) // 16
    s, e = sc["start"], sc["end"]
    total = 0
    for ty in range(bins.tiles_y):
        for tx in range(bins.tiles_x):
            hit = (s[:, 0] // 16 <= tx) & (e[:, 0] // 16 >= tx) & (s[:, 1] // 16 <= ty) & (e[:, 1] // 16 >= ty)
            want = torch.nonzero(hit).flatten().to(torch.int32)  # ascending index = depth order
            t = ty * bins.tiles_x + tx
            got = tl[int(ts[t]) : int(ts[t + 1])]
            assert torch.equal(got, want), (tx, ty)
            total += want.numel()
    assert total == bins.n_tile_pairs == int(toff[-1])


def test_pixel_lists_equal_stable_sort(device):
    """f2: the CSR built from tile lists == what the reference derives with unique + torch.sort (stable)
    over the M pixel keys (gs_model.py:538-548): `index` bit for bit."""
    from simplegaussiansplat_tk71_amd import raster

    sc = make_scene(300, 70, 45, 8, 13)
    s, e = sc["start"], sc["end"]
    bins = raster.bin_tiles(s.to(device), e.to(device), 70, 45)
    pl = raster.pixel_lists(bins, s.to(device), e.to(device))
    # reference-style expansion (uitility.py:336-366): Gaussian-major, row-major inside the box
    rect_g, rect_x, rect_y = [], [], []
    for g in range(s.size(0)):
        xs = torch.arange(int(s[g, 0]), int(e[g, 0]) + 1)
        ys = torch.arange(int(s[g, 1]), int(e[g, 1]) + 1)
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        rect_x.append(xx.flatten()); rect_y.append(yy.flatten()); rect_g.append(torch.full((xx.numel(),), g))
    rx, ry, rg = torch.cat(rect_x), torch.cat(rect_y), torch.cat(rect_g)
    key = (ry * 10000 + rx).to(torch.int32)
    sorted_key, index = torch.sort(key, stable=True)
    assert pl.pair_index.numel() == key.numel()
    assert torch.equal(pl.pair_index.cpu().long(), index)
    assert torch.equal(pl.pair_gauss.cpu().long(), rg[index])
    counts = torch.zeros(46 * 71, dtype=torch.int64).index_add_(0, (ry * 71 + rx), torch.ones_like(rx))
    assert torch.equal(pl.pixel_off.cpu().long(), torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)]))


def test_fused_blend_equals_scan_path(device):
    """The fused kernel's per-pixel transmittance is the grouped cumprod of the scan path: feed the CSR order
    to grouped_cumprod_forward and rebuild the image from its exclusive products."""
    import grouped_cumprod as gc
    from simplegaussiansplat_tk71_amd import raster

    sc = make_scene(350, 80, 60, 7, 21)
    d = {k: v.to(device) for k, v in sc.items() if isinstance(v, torch.Tensor)}
    bins = raster.bin_tiles(d["start"], d["end"], 80, 60)
    img = raster.blend_forward(bins, d["start"], d["end"], d["mean"], d["vinv"], d["opacity"], d["l_d"])
    pl = raster.pixel_lists(bins, d["start"], d["end"])
    g = pl.pair_gauss.long()
    pix = torch.repeat_interleave(torch.arange(61 * 81, device=device), torch.diff(pl.pixel_off).long())
    py, px = pix // 81, pix % 81
    dx = px.float() - d["mean"][g, 0].float()
    dy = py.float() - d["mean"][g, 1].float()
    v = d["vinv"][g]
    q = (dx * v[:, 0, 0] + dy * v[:, 1, 0]) * dx + (dx * v[:, 0, 1] + dy * v[:, 1, 1]) * dy
    gk = torch.exp(-0.5 * q)
    anti = 1.0 - d["opacity"][g, 0] * gk
    incl = torch.empty_like(anti)
    gc.grouped_cumprod_forward(anti.contiguous(), (py * 10000 + px).to(torch.int32).contiguous(), incl)
    T = incl / anti
    p = (T * d["opacity"][g, 0] * gk)[:, None] * d["l_d"][g]
    want = torch.zeros(61 * 81, 3, device=device).index_add_(0, pix, p).reshape(61, 81, 3)
    torch.testing.assert_close(img, want, atol=TOL, rtol=TOL)


def test_exclusive_scan_i32(device):
    from simplegaussiansplat_tk71_amd import raster

    for n in (0, 1, 255, 2048, 2049, 1_000_003):
        x = torch.randint(0, 50, (n,), dtype=torch.int32, generator=torch.Generator().manual_seed(n))
        got = raster.exclusive_scan_i32(x.to(device)).cpu()
        want = torch.cat([torch.zeros(1, dtype=torch.int64), x.long().cumsum(0)]).to(torch.int32)
        assert torch.equal(got, want), n


def test_blend_is_deterministic_and_empty_scene(device):
    from simplegaussiansplat_tk71_amd import raster

    sc = make_scene(600, 120, 90, 10, 33)
    a = _apply(device, sc)
    b = _apply(device, sc)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    e2 = torch.zeros(0, 2, dtype=torch.int32, device=device)
    bins = raster.bin_tiles(e2, e2, 31, 17)
    assert bins.n_tile_pairs == 0
    img = raster.blend_forward(bins, e2, e2, e2.float(), torch.zeros(0, 2, 2, device=device), torch.zeros(0, 1, device=device),
                               torch.zeros(0, 3, device=device))
    assert img.shape == (18, 32, 3) and float(img.abs().sum()) == 0.0


def test_band_sharded_blend_equals_full_frame(device):
    """The per-band inputs of sharding.band_view reproduce the full-frame image rows bit for bit (depth order
    and per-pixel arithmetic are unchanged) and the per-Gaussian gradients sum to the full-frame ones."""
    from simplegaussiansplat_tk71_amd import raster, sharding

    sc = make_scene(700, 130, 100, 14, 41)
    d = {k: v.to(device) for k, v in sc.items() if isinstance(v, torch.Tensor)}
    w, h = 130, 100
    full_bins = raster.bin_tiles(d["start"], d["end"], w, h)
    full, fck = raster.blend_forward(full_bins, d["start"], d["end"], d["mean"], d["vinv"], d["opacity"], d["l_d"], with_checkpoints=True)
    gfull = raster.blend_backward(full_bins, d["start"], d["end"], d["mean"], d["vinv"], d["opacity"], d["l_d"], fck, d["wimg"])
    acc = None
    rows = []
    for band in sharding.row_bands(h, 3):
        s, e, m, bh = sharding.band_view(d["start"], d["end"], d["mean"], band)
        bins = raster.bin_tiles(s, e, w, bh)
        img, ck = raster.blend_forward(bins, s, e, m, d["vinv"], d["opacity"], d["l_d"], with_checkpoints=True)
        g = raster.blend_backward(bins, s, e, m, d["vinv"], d["opacity"], d["l_d"], ck, d["wimg"][band[0] : band[1] + 1].contiguous())
        rows.append(img)
        acc = g if acc is None else tuple(a + b for a, b in zip(acc, g))
    assert torch.equal(torch.cat(rows, 0), full)
    for a, b in zip(acc, gfull):
        torch.testing.assert_close(a, b, atol=2e-4, rtol=1e-4)


def test_create_alpha_brend_from_boxes_equals_sort_route(device):
    """a5 with the sort replaced by a walk of the tile lists: the same mask bit for bit and the same values within fp32
    round-off as the stable-sort route, and the CPU statement's inclusive products bit for bit."""
    import cuda_kernel as ck
    from simplegaussiansplat_tk71_amd import raster

    sc = make_scene(400, 90, 70, 8, 51)
    s, e = sc["start"].to(device), sc["end"].to(device)
    # the reference's Gaussian-major rect list (uitility.py:336-366)
    wh = (e - s + 1).long()
    npts = wh[:, 0] * wh[:, 1]
    gid = torch.repeat_interleave(torch.arange(s.size(0), device=device), npts)
    local = torch.arange(int(npts.sum()), device=device) - torch.repeat_interleave(torch.cumsum(npts, 0) - npts, npts)
    rects = torch.stack([s[gid, 0] + local % wh[gid, 0], s[gid, 1] + local // wh[gid, 0]], 1).to(torch.int32)
    anti = 1.0 - 0.9 * torch.rand(rects.size(0), device=device, generator=torch.Generator(device=device).manual_seed(1))
    anti[::19] = 0.0
    from oracle import wrappers as ow

    for flag in ("cumprod", "cumsum"):
        a_vals, a_mask = ck.create_alpha_brend(rects, anti, flag)
        b_vals, b_mask = ck.create_alpha_brend_boxes(s, e, anti, 90, 70, flag)
        assert torch.equal(a_mask, b_mask)
        torch.testing.assert_close(a_vals, b_vals, atol=1e-5, rtol=1e-5)  # tree order vs strictly sequential per pixel
        w_vals, w_mask, _, _ = ow.create_alpha_brend(rects.cpu(), anti.cpu(), flag)
        assert torch.equal(b_mask.cpu(), w_mask)
        if flag == "cumprod":  # the walk multiplies in the CPU statement's own order: the inclusive products agree bit for bit
            assert torch.equal(b_vals.cpu(), w_vals)
    grad = torch.randn(rects.size(0), device=device, generator=torch.Generator(device=device).manual_seed(2))
    grad[::23] = 0.0
    a_vals, a_mask = ck.grad_cumsum(rects, grad)
    b_vals, b_mask = ck.grad_cumsum_boxes(s, e, grad, 90, 70)
    assert torch.equal(a_mask, b_mask)
    torch.testing.assert_close(a_vals, b_vals, atol=1e-5, rtol=1e-5)
    pl = raster.pixel_lists(raster.bin_tiles(s, e, 90, 70), s, e)
    assert torch.equal(pl.pair_key, torch.sort(ck.unique(rects), stable=True).values)


def test_function_on_the_reference_q9_case(device):
    """Narrow Gaussians whose kernel underflows to exactly 0 inside their box: the reference's forward is right and
    its backward is not (tests/test_oracle.py::test_reference_backward_breaks_...).  Ours: image equals the
    reference's, gradients equal the dense oracle's."""
    from oracle import dense_render as dr

    z = np.load(os.path.join(GOLD, "function_golden.npz"))
    name = "fn_300g_64x48_Q9_INFORMATIONAL"
    g = lambda k: torch.from_numpy(z[f"{name}/{k}"])  # noqa: E731
    w, h = (int(v) for v in z[name + "/width_height"])
    sc = dict(boxsize=g("boxsize"), start=g("start"), end=g("end"), mean=g("mean"), vinv=g("vinv"), opacity=g("opacity"),
              l_d=g("l_d"), wimg=g("wimg"), width=w, height=h)
    img, gv, go, gl = _apply(device, sc)
    torch.testing.assert_close(img, g("image"), atol=TOL, rtol=TOL)
    _, gv64, go64, gl64 = dr.render_with_grads(sc["start"], sc["end"], sc["mean"], sc["vinv"], sc["opacity"], sc["l_d"], w, h, sc["wimg"])
    _close(go, go64, go64.abs().mean().item(), "grad_opacity vs dense oracle")
    _close(gv, gv64, gv64.abs().mean().item(), "grad_vinv vs dense oracle")
    _close(gl, gl64, gl64.abs().mean().item(), "grad_l vs dense oracle")


def test_blend_extreme_exponents(device):
    """exp() of the blend kernels at the ends of its range: a huge positive quadratic form (g underflows to
    exactly 0 -> the pair contributes nothing) and a negative-definite 'precision' (g overflows to inf like expf;
    the reference would propagate the same inf/nan) must not disturb the other pixels."""
    from simplegaussiansplat_tk71_amd import raster

    start = torch.tensor([[0, 0], [8, 8]], dtype=torch.int32, device=device)
    end = torch.tensor([[15, 15], [8, 8]], dtype=torch.int32, device=device)
    mean = torch.tensor([[0.0, 0.0], [8.0, 8.0]], device=device)
    vinv = torch.tensor([[[40.0, 0.0], [0.0, 40.0]], [[1.0, 0.0], [0.0, 1.0]]], device=device)
    op = torch.tensor([[0.9], [0.5]], device=device)
    col = torch.ones(2, 3, device=device)
    bins = raster.bin_tiles(start, end, 15, 15)
    img = raster.blend_forward(bins, start, end, mean, vinv, op, col)
    assert torch.isfinite(img).all()
    torch.testing.assert_close(img[0, 0], torch.full((3,), 0.9, device=device))            # g = 1 at the centre
    assert float(img[15, 15].abs().max()) == 0.0                                           # exp(-9000) -> 0
    torch.testing.assert_close(img[8, 8], torch.full((3,), 0.5, device=device))            # untouched by Gaussian 0's tail


def test_expand_rects_matches_reference_order(device):
    """Utilities.make_rect_points_parallel (uitility.py:336-366): Gaussian-major, row-major inside each box."""
    from simplegaussiansplat_tk71_amd import raster

    sc = make_scene(250, 70, 45, 9, 61)
    s, e = sc["start"], sc["end"]
    rects, owner = raster.expand_rects(s.to(device), e.to(device), 70, 45, with_gaussian=True)
    want, wg = [], []
    for g in range(s.size(0)):
        ys = torch.arange(int(s[g, 1]), int(e[g, 1]) + 1)
        xs = torch.arange(int(s[g, 0]), int(e[g, 0]) + 1)
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        want.append(torch.stack([xx.flatten(), yy.flatten()], 1))
        wg.append(torch.full((xx.numel(),), g))
    assert torch.equal(rects.cpu().long(), torch.cat(want))
    assert torch.equal(owner.cpu().long(), torch.cat(wg))
    assert rects.size(0) == int(sc["boxsize"].sum())


def test_expand_rects_every_box_size_class_and_the_reference_names(device):
    """One-pixel boxes, boxes that miss the image (no pairs), boxes written by one wave and by the whole block (> 8192
    pairs), more boxes than one block takes, a box of more than 2^24 pairs (the exact-division branch) — against the
    closed form: pair i of box g is (x0 + i % w, y0 + i // w); and `custom_autograd_grouped_cumprod._create_rects` against the
    rect lists the reference itself produced (tests/golden/carry_golden.npz)."""
    import numpy as np

    from simplegaussiansplat_tk71_amd import cuda_kernel as ck
    from simplegaussiansplat_tk71_amd import raster
    from tests.test_oracle import _carry_golden

    def closed_form(s, e, w, h):
        x0, y0 = s[:, 0].clamp(min=0).long(), s[:, 1].clamp(min=0).long()
        x1, y1 = e[:, 0].clamp(max=w).long(), e[:, 1].clamp(max=h).long()
        bw, bh = (x1 - x0 + 1).clamp(min=0), (y1 - y0 + 1).clamp(min=0)
        size = torch.where((bw > 0) & (bh > 0), bw * bh, torch.zeros_like(bw))
        owner = torch.repeat_interleave(torch.arange(s.size(0)), size)
        local = torch.arange(int(size.sum())) - torch.repeat_interleave(torch.cumsum(size, 0) - size, size)
        return torch.stack((x0[owner] + local % bw[owner], y0[owner] + local // bw[owner]), 1), owner

    g = torch.Generator().manual_seed(5)
    n = 700
    s = torch.stack((torch.randint(0, 3000, (n,), generator=g), torch.randint(0, 2000, (n,), generator=g)), 1).to(torch.int32)
    half = torch.randint(0, 12, (n, 2), generator=g).to(torch.int32)
    e = s + half
    e[::7] = s[::7]                      # one-pixel boxes
    e[3::50, 0] = s[3::50, 0] - 1        # empty
    s[5], e[5] = torch.tensor([10, 20]), torch.tensor([200, 120])      # 191 x 101 = 19 291 pairs: the whole block
    s[640], e[640] = torch.tensor([0, 0]), torch.tensor([2999, 1999])  # 6e6 pairs
    s[77], e[77] = torch.tensor([2990, 1990]), torch.tensor([3100, 2100])  # cut by the image
    want, wown = closed_form(s, e, 2999, 1999)
    rects, owner = raster.expand_rects(s.to(device), e.to(device), 2999, 1999, with_gaussian=True)
    assert torch.equal(rects.cpu().long(), want) and torch.equal(owner.cpu().long(), wown)
    # more than 2^24 pairs in one box
    s2 = torch.tensor([[1, 2], [0, 0], [4, 4]], dtype=torch.int32)
    e2 = torch.tensor([[3, 2], [4999, 3999], [4, 5]], dtype=torch.int32)
    want, _ = closed_form(s2, e2, 5000, 4000)
    rects = raster.expand_rects(s2.to(device), e2.to(device), 5000, 4000)
    assert rects.size(0) == 3 + 20_000_000 + 2 and torch.equal(rects.cpu().long(), want)
    # the reference's own lists, under its own name
    z = _carry_golden()
    for name in ("chain_small", "chain_mid"):
        start, end = torch.from_numpy(z[name + "/start"]), torch.from_numpy(z[name + "/end"])
        got = ck.custom_autograd_grouped_cumprod._create_rects(start.to(device), end.to(device))
        want, _ = closed_form(start, end, 1 << 30, 1 << 30)
        assert got.dtype == torch.int32 and torch.equal(got.cpu().long(), want)
    for name in ("m_tiny", "m_small", "m_mid"):  # (rects stored by the reference's _create_rects; the scene it came from is make_scene's)
        from tests.golden.make_function_golden import make_scene as golden_scene

        n_gauss, w, h, mh, seed = {"m_tiny": (5, 10, 8, 2, 51), "m_small": (40, 33, 17, 4, 52), "m_mid": (260, 64, 48, 5, 53)}[name]
        sc = golden_scene(n_gauss, w, h, mh, seed)
        got = ck.create_rects(sc["start"].to(device), sc["end"].to(device))
        assert np.array_equal(got.cpu().numpy(), z[name + "/rects"])


def test_bin_tiles_refuses_more_than_int32_entries(device):
    """80 000 Gaussians that each cover a 3840x2160 frame would need 2.6e9 (tile, Gaussian) entries."""
    from simplegaussiansplat_tk71_amd import raster

    n = 80_000
    start = torch.zeros(n, 2, dtype=torch.int32, device=device)
    end = torch.tensor([[3839, 2159]], dtype=torch.int32, device=device).repeat(n, 1)
    with pytest.raises(RuntimeError, match="invalid argument"):
        raster.bin_tiles(start, end, 3839, 2159)


def test_gradient_wrt_float_means(device):
    """The reference computes dL/dmean (gs_model.py:743-750) but feeds integer means, so autograd drops it
    (SURVEY §0 Q5).  With float means the Function returns it; for symmetric precision matrices it is the true
    gradient: compare with the dense autograd oracle."""
    import cuda_kernel as ck
    from oracle import dense_render as dr

    sc = make_scene(300, 70, 50, 6, 71)
    mean_f = sc["mean"].float() + 0.25  # off-grid centres
    vinv = sc["vinv"].to(device)
    op = sc["opacity"].to(device)
    l_d = sc["l_d"].to(device)
    m = mean_f.to(device).requires_grad_(True)
    n = sc["start"].size(0)
    img = ck.custom_autograd_grouped_cumprod.apply(sc["boxsize"].to(device), torch.tensor([n], device=device), sc["start"].to(device),
                                                   sc["end"].to(device), m, vinv, op, l_d, 70, 50)
    (img * sc["wimg"].to(device)).sum().backward()
    _, _, _, _, gm64 = dr.render_with_grads(sc["start"], sc["end"], mean_f, sc["vinv"], sc["opacity"], sc["l_d"], 70, 50, sc["wimg"], with_mean=True)
    _close(m.grad.cpu(), gm64, gm64.abs().mean().item(), "grad_mean")


def test_all_boxes_empty(device):
    """Gaussians whose boxes miss the image entirely: no tile entries, zero image, zero gradients."""
    from simplegaussiansplat_tk71_amd import raster

    n = 50
    start = torch.full((n, 2), 40, dtype=torch.int32, device=device)
    end = torch.full((n, 2), 39, dtype=torch.int32, device=device)  # end < start
    bins = raster.bin_tiles(start, end, 31, 17)
    assert bins.n_tile_pairs == 0
    args = (start, end, start.float(), torch.eye(2, device=device).repeat(n, 1, 1), torch.full((n, 1), 0.5, device=device),
            torch.ones(n, 3, device=device))
    img, ck = raster.blend_forward(bins, *args, with_checkpoints=True)
    assert float(img.abs().sum()) == 0.0
    grads = raster.blend_backward(bins, *args, ck, torch.ones_like(img))
    assert all(float(g.abs().sum()) == 0.0 for g in grads)


@pytest.mark.parametrize("n,hi", [(1, 5), (63, 3), (4096, 1 << 8), (4097, 1 << 16), (100003, 10_800_000), (3_000_017, 21_600_000),
                                  (50_000, 1), (200_000, 2_000_000_000)])
def test_stable_sort_keys_equals_torch_stable_sort(device, n, hi):
    """The native (key, index) sort: same sorted keys AND the same permutation as torch.sort(stable=True)."""
    from simplegaussiansplat_tk71_amd import raster

    g = torch.Generator().manual_seed(n)
    keys = torch.randint(0, hi + 1, (n,), generator=g, dtype=torch.int32)
    want_k, want_i = torch.sort(keys, stable=True)
    got_k, got_i = raster.stable_sort_keys(keys.to(device))
    assert torch.equal(got_k.cpu(), want_k)
    assert torch.equal(got_i.cpu().long(), want_i)


def test_render_cameras_on_two_streams_equals_sequential(device):
    from simplegaussiansplat_tk71_amd import raster

    cams = []
    for seed in range(5):
        sc = make_scene(400 + 50 * seed, 100, 80, 8, 80 + seed)
        cams.append({k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in sc.items()})
    seq = []
    for c in cams:
        bins = raster.bin_tiles(c["start"], c["end"], c["width"], c["height"])
        seq.append(raster.blend_forward(bins, c["start"], c["end"], c["mean"], c["vinv"], c["opacity"], c["l_d"]))
    par = raster.render_cameras(cams, n_streams=2)
    torch.cuda.synchronize()
    for a, b in zip(seq, par):
        assert torch.equal(a, b)


def test_gaussians_behind_an_opaque_stack_get_accurate_gradients(device):
    """60 layers of opacity 0.6 over the whole image leave T ~ 1e-24 for the 40 Gaussians behind them.  The reference's
    reverse scan (gs_model.py:716-722) gives those their true, tiny gradients; a backward that forms suffix sums as
    (pixel total) - (prefix) hands them ~1e-7 of the total instead — noise that Adam normalises into full-size steps.
    Here every gradient term is T_k times a bounded quantity (csrc/gcp_raster.hip, k_blend_bwd), so the hidden
    Gaussians' gradients must be right to a RELATIVE 1e-3 although they are ~1e-20."""
    w = h = 31
    n_front, n_back = 60, 40
    n = n_front + n_back
    g = torch.Generator().manual_seed(3)
    start = torch.zeros(n, 2, dtype=torch.int32)
    end = torch.full((n, 2), w, dtype=torch.int32)
    mean = torch.randint(8, 24, (n, 2), generator=g).to(torch.int32)
    vinv = (torch.eye(2) * 2e-3).repeat(n, 1, 1)  # wide: g ~ 1 over the whole image
    opacity = torch.cat([torch.full((n_front, 1), 0.6), 0.2 + 0.6 * torch.rand(n_back, 1, generator=g)])
    sc = {"start": start, "end": end, "mean": mean, "vinv": vinv, "opacity": opacity, "l_d": 0.1 + torch.rand(n, 3, generator=g),
          "boxsize": torch.prod((end - start + 1).long(), 1), "width": w, "height": h,
          "wimg": torch.randn(h + 1, w + 1, 3, generator=g)}
    from oracle import dense_render as dr

    img, gv, go, gl = _apply(device, sc)
    i64, gv64, go64, gl64 = dr.render_with_grads(start, end, mean, vinv, opacity, sc["l_d"], w, h, sc["wimg"])
    torch.testing.assert_close(img.double(), i64, atol=TOL, rtol=TOL)
    for name, got, want in (("opacity", go, go64), ("vinv", gv, gv64), ("l_d", gl, gl64)):
        hidden_true = want[n_front:].abs().max().item()
        assert hidden_true < 1e-8, (name, hidden_true)                   # far below the ~1e-6 round-off of the pixel totals
        assert want[:n_front].abs().max().item() > 1e-3, name              # the visible ones still learn
        # per Gaussian: error relative to that Gaussian's own gradient magnitude (sum of |components|), hidden ones included
        got, want = got.double().reshape(n, -1), want.reshape(n, -1)
        rel = (got - want).abs().sum(1) / want.abs().sum(1).clamp(min=1e-300)
        assert float(rel.max()) < 1e-3, (name, float(rel.max()), int(rel.argmax()))


def test_layers_behind_an_exactly_opaque_layer_add_nothing_and_get_zero_gradients(device):
    """A flat layer of opacity exactly 1 (Λ = 0: g = 1 everywhere, 1 - αG = 0) over the upper half of a 32 x 48 image, 70 layers
    behind it, 20 in front: every pixel of the upper tiles' strips is behind an exact zero — the forward skips the rest of
    their lists and the backward writes their zeros without evaluating them (§5 item 9) — while the lower half goes on
    through all 91 layers; both against the dense oracle, the hidden layers' gradients exactly 0 in the covered tiles."""
    from oracle import dense_render as dr

    w, h = 31, 47
    n_front, n_back = 20, 70
    n = n_front + 1 + n_back
    g = torch.Generator().manual_seed(11)
    start = torch.zeros(n, 2, dtype=torch.int32)
    end = torch.tensor([[w, h]], dtype=torch.int32).repeat(n, 1)
    end[n_front] = torch.tensor([w, 15])                      # the opaque layer: the upper tile row only
    mean = torch.stack([torch.randint(4, 28, (n,), generator=g), torch.randint(4, 44, (n,), generator=g)], 1).to(torch.int32)
    vinv = (torch.eye(2) * 3e-3).repeat(n, 1, 1)
    vinv[n_front] = 0.0
    opacity = 0.05 + 0.5 * torch.rand(n, 1, generator=g)
    opacity[n_front] = 1.0
    sc = {"start": start, "end": end, "mean": mean, "vinv": vinv, "opacity": opacity, "l_d": 0.1 + torch.rand(n, 3, generator=g),
          "boxsize": torch.prod((end - start + 1).long(), 1), "width": w, "height": h, "wimg": torch.randn(h + 1, w + 1, 3, generator=g)}
    img, gv, go, gl = _apply(device, sc)
    i64, gv64, go64, gl64 = dr.render_with_grads(start, end, mean, vinv, opacity, sc["l_d"], w, h, sc["wimg"])
    torch.testing.assert_close(img.double(), i64, atol=TOL, rtol=TOL)
    for name, got, want in (("opacity", go, go64), ("vinv", gv, gv64), ("l_d", gl, gl64)):
        got, want = got.double().reshape(n, -1), want.reshape(n, -1)
        scale = want.abs().max().item()
        assert float((got - want).abs().max()) <= 2e-5 * (1.0 + scale), name
    # the same scene with the opaque layer over the WHOLE image: nothing behind it reaches the image or gets a gradient
    end2 = end.clone()
    end2[n_front] = torch.tensor([w, h])
    sc2 = dict(sc, end=end2, boxsize=torch.prod((end2 - start + 1).long(), 1))
    img2, gv2, go2, gl2 = _apply(device, sc2)
    sc3 = {k: (v[: n_front + 1] if torch.is_tensor(v) and v.size(0) == n else v) for k, v in sc2.items()}
    img3, gv3, go3, gl3 = _apply(device, sc3)
    assert torch.equal(img2, img3)
    assert float(go2[n_front + 1:].abs().max()) == 0.0 and float(gv2[n_front + 1:].abs().max()) == 0.0 and float(gl2[n_front + 1:].abs().max()) == 0.0
    assert torch.equal(go2[: n_front + 1], go3) and torch.equal(gl2[: n_front + 1], gl3) and torch.equal(gv2[: n_front + 1], gv3)


def test_boxes_over_hundreds_of_tiles_among_small_ones(device):
    """Background splats: six Gaussians whose boxes cover a whole 304 x 208 image (19 x 13 = 247 tiles each) spread through 600
    small ones.  Their (tile, Gaussian) entries are emitted, and their 247 gradient slots summed, by a whole wave instead of one
    thread (k_tile_emit, k_grad_reduce); the tile lists against the sort the reference would do, image and gradients against
    the dense oracle."""
    from oracle import dense_render as dr
    from simplegaussiansplat_tk71_amd import raster

    w, h = 303, 207
    sc = make_scene(606, w, h, 6, seed=17)
    big = torch.arange(0, 606, 101)
    sc["start"][big] = 0
    sc["end"][big] = torch.tensor([w, h], dtype=torch.int32)
    sc["opacity"][big] = 0.05
    sc["vinv"][big] = torch.eye(2) * 1e-4
    sc["boxsize"] = torch.prod((sc["end"] - sc["start"] + 1).long(), 1)
    bins = raster.bin_tiles(sc["start"].to(device), sc["end"].to(device), w, h)
    # reference order: (tile, Gaussian) pairs sorted by tile, Gaussians ascending inside a tile
    tx0, ty0 = sc["start"][:, 0] // 16, sc["start"][:, 1] // 16
    tx1, ty1 = sc["end"][:, 0] // 16, sc["end"][:, 1] // 16
    want = []
    for g in range(606):
        for ty in range(int(ty0[g]), int(ty1[g]) + 1):
            for tx in range(int(tx0[g]), int(tx1[g]) + 1):
                want.append((ty * bins.tiles_x + tx, g))
    want.sort()
    assert bins.n_tile_pairs == len(want)
    assert bins.tile_list.cpu().tolist() == [g for _, g in want]
    img, gv, go, gl = _apply(device, sc)
    i64, gv64, go64, gl64 = dr.render_with_grads(sc["start"], sc["end"], sc["mean"], sc["vinv"], sc["opacity"], sc["l_d"], w, h, sc["wimg"])
    torch.testing.assert_close(img.double(), i64, atol=TOL, rtol=TOL)
    for name, got, want64 in (("opacity", go, go64), ("vinv", gv, gv64), ("l_d", gl, gl64)):
        got, want64 = got.double().reshape(606, -1), want64.reshape(606, -1)
        # a background splat's gradient is a sum over 63 000 pixels: the bound is relative to the sum of its |terms|, which
        # the largest component over-estimates by little here
        scale = want64.abs().max(1, keepdim=True).values
        assert bool(((got - want64).abs() <= 1e-4 * (1.0 + scale)).all()), name


def _stack_scene(n_layers, opacity_lo, opacity_hi, seed, w=15, h=15):
    """`n_layers` wide Gaussians over one 16x16 tile: every pixel's list is n_layers deep."""
    g = torch.Generator().manual_seed(seed)
    n = n_layers
    start = torch.zeros(n, 2, dtype=torch.int32)
    end = torch.tensor([[w, h]], dtype=torch.int32).repeat(n, 1)
    mean = torch.randint(2, 14, (n, 2), generator=g).to(torch.int32)
    sx = 4.0 + 8.0 * torch.rand(n, generator=g)
    vinv = torch.zeros(n, 2, 2)
    vinv[:, 0, 0] = 1.0 / (sx * sx)
    vinv[:, 1, 1] = 1.0 / (sx * sx)
    opacity = opacity_lo + (opacity_hi - opacity_lo) * torch.rand(n, 1, generator=g)
    return {"start": start, "end": end, "mean": mean, "vinv": vinv, "opacity": opacity, "l_d": 0.1 + torch.rand(n, 3, generator=g),
            "boxsize": torch.prod((end - start + 1).long(), 1), "width": w, "height": h,
            "wimg": torch.randn(h + 1, w + 1, 3, generator=g)}


@pytest.mark.parametrize("n_layers,op_lo,op_hi", [(4096, 0.0005, 0.004), (4096, 0.002, 0.02), (1200, 0.005, 0.05)])
def test_deep_pixel_columns_relative_gradient_accuracy(device, n_layers, op_lo, op_hi):
    """Pixel lists as deep as BASELINE's configs allow (4096 layers: 128 backward chunks, 16 forward staging rounds),
    with opacities chosen so that the transmittance falls through [1, 1e-2 .. 1e-20] along the list.  Every Gaussian's
    gradient must be accurate RELATIVE to its own size — the semi-occluded ones (T in [1e-5, 1e-2]) are exactly where a
    total-minus-prefix backward loses all digits — against the dense fp64 renderer."""
    from oracle import dense_render as dr

    sc = _stack_scene(n_layers, op_lo, op_hi, seed=n_layers + int(op_hi * 1e4))
    img, gv, go, gl = _apply(device, sc)
    i64, gv64, go64, gl64 = dr.render_with_grads(sc["start"], sc["end"], sc["mean"], sc["vinv"], sc["opacity"], sc["l_d"],
                                                 sc["width"], sc["height"], sc["wimg"])
    torch.testing.assert_close(img.double(), i64, atol=TOL, rtol=TOL)
    n = n_layers
    scales, T = full_cover_scales(sc, n)
    assert float((T[:, 120] < 1e-2).double().mean()) > 0.3 and float(T[-1, 120]) < 1e-3    # the list really runs into occlusion
    seen = {}
    for name, got, want in (("opacity", go, go64), ("vinv", gv, gv64), ("l_d", gl, gl64)):
        got, want, scale = got.double().reshape(n, -1), want.reshape(n, -1), scales[name]
        err = (got - want).abs()
        bound = 2e-5 * scale + 1e-30
        assert bool((err <= bound).all()), (name, float((err / (scale + 1e-300)).max()), int((err > bound).any(1).nonzero()[0]))
        seen[name] = (float(scale.max()), float(scale.min()), float((err / (scale + 1e-300)).max()))
    print(f"{n_layers} layers, opacity [{op_lo}, {op_hi}], T at the list end {float(T[-1, 120]):.1e}: per-Gaussian (largest "
          f"condition scale, smallest, max err / scale): {seen}")


@pytest.mark.parametrize("name", ["deep_300", "deep_700"])
def test_deep_stacks_vs_the_reference_functions_own_outputs(device, name):
    """tests/golden/function_deep_golden.npz = the reference's own custom_autograd_grouped_cumprod (its reverse scan
    `grad_cumsum`, gs_model.py:716-722) run on CPU on 300- and 700-layer pixel lists whose transmittance falls to 1e-9 /
    1e-20 (generator: tests/golden/make_function_deep_golden.py).  The HIP Function's image and its opacity / covariance
    gradients must agree with the REFERENCE's — each Gaussian within 4e-5 of its own condition scale (the reference is
    fp32 itself) — over the 18 orders of magnitude those gradients span."""
    z = np.load(os.path.join(GOLD, "function_deep_golden.npz"))
    g = lambda k: torch.from_numpy(z[f"{name}/{k}"])  # noqa: E731
    w, h = (int(v) for v in z[name + "/width_height"])
    sc = dict(boxsize=g("boxsize"), start=g("start"), end=g("end"), mean=g("mean"), vinv=g("vinv"), opacity=g("opacity"),
              l_d=g("l_d"), wimg=g("wimg"), width=w, height=h)
    n = sc["start"].size(0)
    img, gv, go, gl = _apply(device, sc)
    torch.testing.assert_close(img, g("image"), atol=TOL, rtol=TOL)
    scales, T = full_cover_scales(sc, n)
    assert float(T[-1].median()) < (1e-7 if name == "deep_300" else 1e-15)  # the lists really run deep into occlusion
    for what, got, want in (("opacity", go, g("grad_opacity")), ("vinv", gv, g("grad_vinv"))):
        got, want = got.double().reshape(n, -1), want.double().reshape(n, -1)
        err = (got - want).abs()
        bound = 4e-5 * scales[what] + 1e-30
        assert bool((err <= bound).all()), (what, float((err / (scales[what] + 1e-300)).max()), int((err > bound).any(1).nonzero()[0]))
    assert float(g("grad_opacity").abs().min()) < 1e-8 < 1.0 < float(g("grad_opacity").abs().max())  # the span is real


def test_random_small_scenes_against_the_dense_oracle(device):
    """Fuzz of the binning / hit-list / staging logic: 40 seeded scenes of random size (images that are not multiples of
    the 16-pixel tile, 1-pixel boxes, boxes larger than the image, lists of 1 to ~300 entries per tile), image and all
    three gradients against the dense fp64 renderer."""
    from oracle import dense_render as dr

    rng = np.random.default_rng(12345)
    for case in range(40):
        w, h = int(rng.integers(3, 70)), int(rng.integers(3, 70))
        n = int(rng.choice([1, 2, 7, 33, 64, 65, 129, 300]))
        mh = int(rng.choice([1, 2, 5, 17, 40]))
        sc = make_scene(n, w, h, mh, 1000 + case, opacity_one_every=int(rng.choice([0, 0, 5])))
        img, gv, go, gl = _apply(device, sc)
        i64, gv64, go64, gl64 = dr.render_with_grads(sc["start"], sc["end"], sc["mean"], sc["vinv"], sc["opacity"], sc["l_d"], w, h, sc["wimg"])
        what = f"case {case}: {n} Gaussians, {w}x{h}, half <= {mh}"
        torch.testing.assert_close(img.double(), i64, atol=TOL, rtol=TOL, msg=lambda m: f"{what}: {m}")
        # one bound for every depth (1 to 300 layers per pixel here): 1e-5 relative to the mean gradient component
        _close(go, go64, go64.abs().mean().item(), what + " grad_opacity")
        _close(gv, gv64, gv64.abs().mean().item(), what + " grad_vinv")
        _close(gl, gl64, gl64.abs().mean().item(), what + " grad_l")


def test_capture_safe_binning_equals_exact_binning_and_reports_overflow(device):
    """gcp_bin_tiles (one call, caller-bounded entry count, nothing read back) builds the same tile lists as the
    count / read K / fill route, for any sufficient capacity; a capacity that is too small is reported in `info` and the
    Gaussians that did not fit are dropped from the BACK of the depth order, with zero gradients."""
    from simplegaussiansplat_tk71_amd import raster

    sc = make_scene(700, 120, 90, 14, 31)
    d = {k: v.to(device) for k, v in sc.items() if isinstance(v, torch.Tensor)}
    exact = raster.bin_tiles(d["start"], d["end"], 120, 90)
    K = exact.n_tile_pairs
    for cap in (K, K + 1, 2 * K + 4097):
        b = raster.bin_tiles(d["start"], d["end"], 120, 90, capacity=cap)
        assert b.info.tolist() == [K, 0] and not b.overflowed()
        assert torch.equal(b.tile_off, exact.tile_off) and torch.equal(b.tile_start, exact.tile_start)
        assert torch.equal(b.tile_list[:K], exact.tile_list)
        img_e, ck_e = raster.blend_forward(exact, d["start"], d["end"], d["mean"], d["vinv"], d["opacity"], d["l_d"], with_checkpoints=True)
        img_b, ck_b = raster.blend_forward(b, d["start"], d["end"], d["mean"], d["vinv"], d["opacity"], d["l_d"], with_checkpoints=True)
        assert torch.equal(img_b, img_e)
        ge = raster.blend_backward(exact, d["start"], d["end"], d["mean"], d["vinv"], d["opacity"], d["l_d"], ck_e, d["wimg"])
        gb = raster.blend_backward(b, d["start"], d["end"], d["mean"], d["vinv"], d["opacity"], d["l_d"], ck_b, d["wimg"])
        assert all(torch.equal(x, y) for x, y in zip(gb, ge))
    # too small: the first Gaussians (front of the depth order) that fit are kept, the rest dropped
    cap = K // 2
    b = raster.bin_tiles(d["start"], d["end"], 120, 90, capacity=cap)
    listed, flag = b.info.tolist()
    assert flag == 1 and b.overflowed() and 0 < listed <= cap
    toff = exact.tile_off.cpu()
    n_kept = int((toff[1:] <= cap).sum())          # Gaussians whose entries end below the capacity
    assert listed == int(toff[n_kept])
    part = raster.bin_tiles(d["start"][:n_kept], d["end"][:n_kept], 120, 90)
    assert torch.equal(b.tile_start, part.tile_start) and torch.equal(b.tile_list[:listed], part.tile_list)
    img_b, ck_b = raster.blend_forward(b, d["start"], d["end"], d["mean"], d["vinv"], d["opacity"], d["l_d"], with_checkpoints=True)
    img_p = raster.blend_forward(part, d["start"][:n_kept], d["end"][:n_kept], d["mean"][:n_kept], d["vinv"][:n_kept],
                                 d["opacity"][:n_kept], d["l_d"][:n_kept])
    assert torch.equal(img_b, img_p)
    gb = raster.blend_backward(b, d["start"], d["end"], d["mean"], d["vinv"], d["opacity"], d["l_d"], ck_b, d["wimg"])
    assert all(float(g[n_kept:].abs().sum()) == 0.0 for g in gb)


def test_function_forward_backward_in_a_hip_graph(device):
    """With a tile capacity the whole Function — binning, blend forward, blend backward — queues its kernels without a
    single device->host read, so forward + backward are captured into ONE HIP graph and replayed on new parameter values
    (the reference synchronises on .item() several times per camera, gs_model.py:677,793,801-802)."""
    import cuda_kernel as ck
    from oracle import dense_render as dr

    sc = make_scene(400, 100, 70, 9, 3)
    n = sc["start"].size(0)
    st = {k: sc[k].to(device) for k in ("boxsize", "start", "end", "mean")}
    vinv = sc["vinv"].to(device).clone().requires_grad_(True)
    op = sc["opacity"].to(device).clone().requires_grad_(True)
    l_d = sc["l_d"].to(device).clone().requires_grad_(True)
    wimg = sc["wimg"].to(device)
    batch = torch.tensor([n], device=device)

    def body(v, o, l):
        img = ck.custom_autograd_grouped_cumprod.apply(st["boxsize"], batch, st["start"], st["end"], st["mean"], v, o, l,
                                                       sc["width"], sc["height"])
        return (img * wimg).sum(), img

    def step():
        loss, img = body(vinv, op, l_d)
        return (img, *torch.autograd.grad(loss, (vinv, op, l_d)))

    eager = step()  # an eager step on the default stream whose outputs (and autograd graph) stay referenced below
    cap = 4 * n + 1024
    gs = ck.GraphedStep(body, [vinv, op, l_d], capacity=cap)
    assert not ck.capacity_exceeded()
    # new parameter values in the captured buffers, then replay: no Python, no host read
    with torch.no_grad():
        op.mul_(0.7).add_(0.1)
        l_d.copy_(torch.rand_like(l_d))
        vinv.mul_(1.3)
    (_, g_img), g_grads = gs.replay()
    torch.cuda.synchronize()
    assert not ck.capacity_exceeded()
    outs = (g_img, *g_grads)
    assert eager[0].grad_fn is not None
    img, gv, go, gl = (t.detach().cpu() for t in outs)
    i64, gv64, go64, gl64 = dr.render_with_grads(sc["start"], sc["end"], sc["mean"], vinv.detach().cpu(), op.detach().cpu(),
                                                 l_d.detach().cpu(), sc["width"], sc["height"], sc["wimg"])
    torch.testing.assert_close(img.double(), i64, atol=TOL, rtol=TOL)
    _close(go, go64, go64.abs().mean().item(), "graph replay grad_opacity")
    _close(gv, gv64, gv64.abs().mean().item(), "graph replay grad_vinv")
    _close(gl, gl64, gl64.abs().mean().item(), "graph replay grad_l")
    # a bound that is too small is reported, not silently wrong
    with ck.tile_capacity(50):
        step()
    assert ck.capacity_exceeded()
    assert not ck.capacity_exceeded()  # read and reset


def test_capacity_overflow_is_reported_by_every_replay_of_a_captured_step(device):
    """the overflow flag of a captured step must not go blind after the first check.  Capture with a bound
    that fits, let the boxes grow in place (Gaussians move between steps), replay: every replay whose lists outgrow the
    captured bound reports it, and shrinking the boxes again clears it."""
    import cuda_kernel as ck
    from simplegaussiansplat_tk71_amd import raster

    sc = make_scene(400, 100, 70, 9, 5)
    n = sc["start"].size(0)
    st = {k: sc[k].to(device) for k in ("boxsize", "start", "end", "mean")}
    vinv = sc["vinv"].to(device).clone().requires_grad_(True)
    op = sc["opacity"].to(device).clone().requires_grad_(True)
    l_d = sc["l_d"].to(device).clone().requires_grad_(True)
    wimg = sc["wimg"].to(device)
    K = raster.bin_tiles(st["start"], st["end"], sc["width"], sc["height"]).n_tile_pairs

    def body(v, o, l):
        img = ck.custom_autograd_grouped_cumprod.apply(st["boxsize"], None, st["start"], st["end"], st["mean"], v, o, l,
                                                       sc["width"], sc["height"])
        return (img * wimg).sum(), img

    gs = ck.GraphedStep(body, [vinv, op, l_d], capacity=K + 8)
    gs.replay()
    assert not ck.capacity_exceeded()
    small = st["end"].clone()
    st["end"].add_(48).clamp_(max=torch.tensor([sc["width"], sc["height"]], device=device, dtype=st["end"].dtype))
    assert raster.bin_tiles(st["start"], st["end"], sc["width"], sc["height"]).n_tile_pairs > K + 8
    for _ in range(3):  # every replay reports, not only the first
        gs.replay()
        assert ck.capacity_exceeded()
    gs.replay()
    gs.replay()
    assert ck.capacity_exceeded() and not ck.capacity_exceeded()  # sticky across replays until read
    st["end"].copy_(small)
    gs.replay()
    assert not ck.capacity_exceeded()


_STALE_CAPTURE = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
import cuda_kernel as ck
from tests.util import make_scene

dev = torch.device("cuda", 0)
sc = make_scene(200, 63, 47, 9, seed=5)
w, h = sc["width"], sc["height"]
ints = [sc[k].to(dev) for k in ("boxsize", "start", "end", "mean")]
op = sc["opacity"].to(dev).requires_grad_(True)
vinv = sc["vinv"].to(dev).requires_grad_(True)
l_d = sc["l_d"].to(dev).requires_grad_(True)
batch = torch.tensor([200], device=dev)

def step(v, o, l):
    img = ck.custom_autograd_grouped_cumprod.apply(ints[0], batch, ints[1], ints[2], ints[3], v, o * 1.0, l, w, h)
    return torch.autograd.grad(img.square().sum(), [v, o, l])

eager = step(vinv, op, l_d)          # an eager step on the default stream; `kept` keeps its graph (and the leaves' nodes) alive
kept = (op * 2.0).sum()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
mode = sys.argv[2]
with ck.tile_capacity(8 * 200, dev):
    with torch.cuda.stream(side):
        pass
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(graph, stream=side):
            if mode == "stale":
                step(vinv, op, l_d)                  # the leaves of the eager step: refused, not crashed on
            else:
                fresh = [ck.capture_leaf(t.detach().requires_grad_(True)) for t in (vinv, op, l_d)]
                got = step(*fresh)
    except RuntimeError as e:
        print("RAISED", "GraphedStep" in str(e) and "capture_leaf" in str(e))
        sys.exit(0)
graph.replay()
torch.cuda.synchronize()
print("CAPTURED", all(torch.allclose(a, b, rtol=1e-4, atol=1e-5) for a, b in zip(got, eager)))
'''


@pytest.mark.parametrize("mode", ["stale", "marked"])
def test_bare_capture_of_the_function_refuses_leaves_of_eager_steps(device, tmp_path, mode):
    """A bare `torch.cuda.graph` around the Function after eager steps used to end in SIGSEGV inside capture_end (stale
    AccumulateGrad nodes pull the default stream into the capture: tools/capture_repro.py).  The Function now refuses leaves
    that were not made for the capture with a RuntimeError that names `GraphedStep` and `capture_leaf`; fresh leaves marked
    with `capture_leaf` capture and replay to the eager gradients.  (In a subprocess: a crash must not take the suite down.)"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "capture_case.py"
    script.write_text(_STALE_CAPTURE)
    res = subprocess.run([sys.executable, "-X", "faulthandler", str(script), root, mode], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert ("RAISED True" if mode == "stale" else "CAPTURED True") in res.stdout, res.stdout + res.stderr[-2000:]


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 1023, 1024, 1025, 4099, 40610, 65535, 65536, 65537, 200001])
def test_exclusive_scan_i32_every_size_class(device, n):
    """The int32 prefix sums under every stream compaction, across the block-count boundaries of its two launches (2048
    elements per block) — aligned and unaligned inputs.  (A one-launch form for short arrays — a single 1024-thread block —
    was measured at 12 us per 40 000 elements against 11 us for the two launches: not kept.)"""
    from simplegaussiansplat_tk71_amd import raster

    g = torch.Generator().manual_seed(n)
    x = torch.randint(0, 4097, (n + 3,), generator=g, dtype=torch.int32)
    for off in (0, 1):
        xs = x[off: off + n].to(device)
        if off:
            xs = x.to(device)[off: off + n]  # a view that starts 4 bytes into the allocation: not 16-byte aligned
        got = raster.exclusive_scan_i32(xs).cpu()
        want = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(x[off: off + n].long(), 0)]).to(torch.int32)
        assert torch.equal(got, want), (n, off)
