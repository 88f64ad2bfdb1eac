"""Row f3 widened: the per-pixel carry helpers the reference's chunk loop calls either side of the scans —
`_create_alpha_brend_min` (gs_model.py:582-586), `_cat_alpha_brend` (:589-594), `create_grad_alphabrend_min` (:724-730) —
on the HIP library (csrc/gcp_pixels.hip), against the reference's own outputs (tests/golden/carry_golden.npz: the helpers
alone and the whole chunk loop of `_forward_batch` / `_backward_batch`, run on CPU by make_carry_golden.py), against the
literal restatement oracle/wrappers.py at larger sizes, and through size-independent properties at the cfg3 scene."""
import numpy as np
import pytest
import torch

from tests.test_oracle import _carry_golden, carry_chain
from tests.util import TOL, assert_parity, make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["cells", "filter"], autouse=True)
def look_table(request, monkeypatch):
    """Every test of this file twice: with the pairs looking at the cells themselves (what lists below 2^25 pairs do) and at
    the 16-bit filter table (what larger lists do) — GCP_PIXELS_FILTER_FROM moves the switch (csrc/gcp_pixels.hip)."""
    # the three long tests run once, under the table their list sizes take anyway (or, the scene-size oracle test, the other one)
    once = {"test_cfg3_scene_properties": "filter", "test_first_pair_index_travels_as_a_float_like_the_reference": "cells",
            "test_against_the_oracle_at_scene_size_and_on_lists_of_unrelated_coordinates": "filter"}
    name = request.node.originalname or request.node.name
    if name in once and once[name] != request.param:
        pytest.skip("long test: run under the other look table only")
    monkeypatch.setenv("GCP_PIXELS_FILTER_FROM", "0" if request.param == "filter" else str(1 << 40))
    return request.param


def _bits(t):
    return t.detach().cpu().numpy().view(np.int32)


@pytest.mark.parametrize("name", ["m_tiny", "m_small", "m_mid"])
def test_helpers_equal_the_reference_bit_for_bit(device, name):
    from simplegaussiansplat_tk71_amd import cuda_kernel as ck

    z = _carry_golden()
    F = ck.custom_autograd_grouped_cumprod
    rects = torch.from_numpy(z[name + "/rects"]).to(device)
    w, h = z[name + "/width_height"].tolist()
    for tag in ("T", "signed"):
        vals = torch.from_numpy(z[f"{name}/{tag}"]).to(device)
        for size in (None, (w, h), (w + 5, h + 9)):  # the list's own extent, the image's, a larger image
            u, m = ck.create_alpha_brend_min(rects, vals, image_size=size)
            assert u.dtype == torch.int32 and np.array_equal(u.cpu().numpy(), z[f"{name}/min_{tag}/unique_rects"]), (tag, size)
            assert np.array_equal(_bits(m), z[f"{name}/min_{tag}/values"].view(np.int32)), (tag, size)
        u, m = F._create_alpha_brend_min(rects, vals)  # under the reference's name
        assert np.array_equal(u.cpu().numpy(), z[f"{name}/min_{tag}/unique_rects"]) and np.array_equal(_bits(m), z[f"{name}/min_{tag}/values"].view(np.int32))
    grad = torch.from_numpy(z[name + "/grad"]).to(device)
    u, gm = F.create_grad_alphabrend_min(rects, grad)
    assert np.array_equal(u.cpu().numpy(), z[name + "/grad_min/unique_rects"]) and np.array_equal(_bits(gm), z[name + "/grad_min/values"].view(np.int32))
    # a list thinned by a mask (what rects[mask] leaves, gs_model.py:608): not boxes any more — the table does not care
    keep = torch.from_numpy(z[name + "/masked/keep"]).to(device)
    T = torch.from_numpy(z[name + "/T"]).to(device)
    u, m = ck.create_alpha_brend_min(rects[keep], T[keep])
    assert np.array_equal(u.cpu().numpy(), z[name + "/masked/unique_rects"]) and np.array_equal(_bits(m), z[name + "/masked/values"].view(np.int32))
    # int64 lists: read where they lie, returned as int64
    u, m = ck.create_alpha_brend_min(rects.long(), T)
    assert u.dtype == torch.int64 and np.array_equal(u.cpu().numpy(), z[name + "/min_T_i64/unique_rects"])
    assert np.array_equal(_bits(m), z[f"{name}/min_T/values"].view(np.int32))
    # a PreparedRects brings the extent along
    prep = ck.PreparedRects(rects)
    u, m = ck.create_alpha_brend_min(prep, T)
    assert np.array_equal(u.cpu().numpy(), z[f"{name}/min_T/unique_rects"]) and np.array_equal(_bits(m), z[f"{name}/min_T/values"].view(np.int32))
    # views that start in the middle of a 16-byte word
    u, m = ck.create_alpha_brend_min(rects[1:], T[1:])
    from oracle import wrappers as ow

    wu, wm = ow.create_alpha_brend_min(rects[1:].cpu(), T[1:].cpu())
    assert torch.equal(u.cpu(), wu) and np.array_equal(_bits(m), _bits(wm))


@pytest.mark.parametrize("name", ["chain_small", "chain_mid"])
def test_the_reference_chunk_loop_runs_on_the_hip_functions_under_its_own_names(device, name):
    """`_forward_batch` / `_backward_batch`'s calls (gs_model.py:601-615, :634-643), chunk after chunk, on
    `custom_autograd_grouped_cumprod._create_rects / _create_alpha_brend / _create_alpha_brend_min / _cat_alpha_brend /
    grad_cumsum / create_grad_alphabrend_min` of the HIP module — against what the reference returned at every step.
    Masks, pixel lists and the rows picked as carries bit for bit; values within the tolerance rule."""
    from simplegaussiansplat_tk71_amd import cuda_kernel as ck

    z = _carry_golden()
    F = ck.custom_autograd_grouped_cumprod
    got = carry_chain(z, name, F, F._create_rects, to_dev=lambda t: t.to(device))
    n_steps = 0
    for k, v in got.items():
        want = z[f"{name}/{k}"]
        if k.endswith(("mask", "unique_rects")):
            assert v.dtype == want.dtype and np.array_equal(v, want), (name, k)
        elif k.endswith(("/T", "/T_min")):
            assert_parity(torch.from_numpy(v), torch.from_numpy(want), None, what=f"{name}/{k}")
        else:  # sums of N(0, 1) terms over a pixel's list: bounded by the list's |terms|
            assert v.shape == want.shape, (name, k)
            scale = torch.full((v.size,), 1.0) * max(1.0, float(np.abs(want).max()))
            assert_parity(torch.from_numpy(v), torch.from_numpy(want), scale, what=f"{name}/{k}")
        n_steps += 1
    assert n_steps == 8 * len(z[name + "/chunk_ends"])


def test_against_the_oracle_at_scene_size_and_on_lists_of_unrelated_coordinates(device):
    from oracle import wrappers as ow
    from simplegaussiansplat_tk71_amd import cuda_kernel as ck
    from simplegaussiansplat_tk71_amd import raster

    sc = make_scene(20000, 639, 359, 9, seed=71)
    rects = raster.expand_rects(sc["start"].to(device), sc["end"].to(device), 639, 359)
    n = rects.size(0)
    assert n > 2_000_000
    g = torch.Generator().manual_seed(72)
    T = torch.rand(n, generator=g)
    grad = torch.randn(n, generator=g)
    wu, wm = ow.create_alpha_brend_min(rects.cpu(), T)
    for size in (None, (639, 359)):
        u, m = ck.create_alpha_brend_min(rects, T.to(device), image_size=size)
        assert torch.equal(u.cpu(), wu) and np.array_equal(_bits(m), _bits(wm))
    wu, wg = ow.create_grad_alphabrend_min(rects.cpu(), grad)
    u, gm = ck.create_grad_alphabrend_min(rects, grad.to(device))
    assert torch.equal(u.cpu(), wu) and np.array_equal(_bits(gm), _bits(wg))
    # any list of coordinates: shuffled pairs, few distinct pixels (every cell contended), partial last block
    perm = torch.randperm(n, generator=g)[: n - 1234]
    r2, t2 = rects.cpu()[perm].contiguous(), T[perm].contiguous()
    wu, wm = ow.create_alpha_brend_min(r2, t2)
    u, m = ck.create_alpha_brend_min(r2.to(device), t2.to(device))
    assert torch.equal(u.cpu(), wu) and np.array_equal(_bits(m), _bits(wm))
    r3 = torch.stack((torch.randint(0, 3, (300001,), generator=g), torch.randint(0, 2, (300001,), generator=g)), 1).to(torch.int32)
    t3 = torch.randn(300001, generator=g)
    wu, wm = ow.create_alpha_brend_min(r3, t3)
    u, m = ck.create_alpha_brend_min(r3.to(device), t3.to(device))
    assert torch.equal(u.cpu(), wu) and np.array_equal(_bits(m), _bits(wm))
    wu, wg = ow.create_grad_alphabrend_min(r3, t3)
    u, gm = ck.create_grad_alphabrend_min(r3.to(device), t3.to(device))
    assert torch.equal(u.cpu(), wu) and np.array_equal(_bits(gm), _bits(wg))


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_lists_against_the_oracle(device, seed):
    """Seeded fuzz of gcp_pixels_min: lists of 1 ... 3e5 pairs over images from 1 x 1 to 700 x 500 — so from every cell contended
    by thousands of pairs to most cells untouched — in int32 and int64, values of both signs with repeats, infinities and both
    zeros, in random order, falling or rising along the list; with and without the image size; minima and first-pair rows
    bit for bit against torch.unique(dim=0) + scatter_reduce(amin)."""
    from oracle import wrappers as ow
    from simplegaussiansplat_tk71_amd import cuda_kernel as ck

    g = torch.Generator().manual_seed(4000 + seed)
    n = int([1, 7, 300, 4096, 4097, 65_537, 300_000][seed % 7] * (1 + seed // 7))
    w, h = [(0, 0), (3, 2), (40, 30), (699, 499)][seed % 4]
    dt = torch.int64 if seed % 3 == 0 else torch.int32
    r = torch.stack((torch.randint(0, w + 1, (n,), generator=g), torch.randint(0, h + 1, (n,), generator=g)), 1).to(dt)
    kind = seed % 5
    if kind == 0:
        v = torch.randn(n, generator=g)
    elif kind == 1:
        v = torch.sort(torch.rand(n, generator=g), descending=True).values            # falling along the list, like transmittances
    elif kind == 2:
        v = torch.sort(torch.rand(n, generator=g)).values                                # rising: every pair a new maximum
    elif kind == 3:
        v = torch.randint(-3, 4, (n,), generator=g).float() * 0.5                        # few distinct values, many ties, +-0
        v[v == 0] = torch.where(torch.rand(int((v == 0).sum()), generator=g) < 0.5, torch.tensor(-0.0), torch.tensor(0.0))
    else:
        v = torch.randn(n, generator=g) * 1e30
        v[::17] = float("inf")
        v[5::29] = float("-inf")
    wu, wm = ow.create_alpha_brend_min(r, v)
    for size in (None, (w, h), (w + 3, h + 1)):
        u, m = ck.create_alpha_brend_min(r.to(device), v.to(device), image_size=size)
        assert u.dtype == dt and torch.equal(u.cpu(), wu), (seed, size)
        # bit for bit — except that a pixel holding BOTH zeros returns -0 here and, in the reference, whichever of the two its
        # reduction met first (-0 == +0: unspecified on its GPU path, list order on the CPU): such minima are compared as values
        both_zero = (m.cpu() == 0) & (wm == 0)
        assert np.array_equal(_bits(m)[~both_zero.numpy()], _bits(wm)[~both_zero.numpy()]), (seed, size)
        assert torch.equal(m.cpu() == 0, wm == 0), (seed, size)
    grad = torch.randn(n, generator=g)
    wu, wg = ow.create_grad_alphabrend_min(r, grad)
    u, gm = ck.create_grad_alphabrend_min(r.to(device), grad.to(device))
    assert torch.equal(u.cpu(), wu) and np.array_equal(_bits(gm), _bits(wg)), seed


@pytest.mark.parametrize("w,h", [(30, 2160), (70, 127), (5, 128), (40, 255), (1, 9000), (4000, 3), (15, 15), (16, 16), (17, 1), (0, 0), (0, 300), (300, 0)])
def test_read_out_on_images_of_every_shape(device, w, h):
    """The table is read out in patches of 16 columns x 128 rows through LDS, the (column, row band) pieces ranked by one scan:
    narrow, tall, wide and one-row images, widths around the 16-column patch, heights around the 128-row band."""
    from oracle import wrappers as ow
    from simplegaussiansplat_tk71_amd import cuda_kernel as ck

    g = torch.Generator().manual_seed(w * 10007 + h)
    n = 50_000
    r = torch.stack((torch.randint(0, w + 1, (n,), generator=g), torch.randint(0, h + 1, (n,), generator=g)), 1).to(torch.int32)
    r[:7] = torch.tensor([[0, 0], [w, h], [w, 0], [0, h], [w // 2, h // 2], [min(15, w), h], [min(16, w), 0]], dtype=torch.int32)
    v = torch.rand(n, generator=g)
    wu, wm = ow.create_alpha_brend_min(r, v)
    u, m = ck.create_alpha_brend_min(r.to(device), v.to(device), image_size=(w, h))
    assert torch.equal(u.cpu(), wu) and np.array_equal(_bits(m), _bits(wm))
    wu, wg = ow.create_grad_alphabrend_min(r.long(), v)
    u, gm = ck.create_grad_alphabrend_min(r.long().to(device), v.to(device))
    assert u.dtype == torch.int64 and torch.equal(u.cpu(), wu) and np.array_equal(_bits(gm), _bits(wg))


def test_edges_empty_one_pixel_nan_and_coordinates_outside(device):
    from simplegaussiansplat_tk71_amd import cuda_kernel as ck

    e = torch.zeros(0, 2, dtype=torch.int32, device=device)
    u, m = ck.create_alpha_brend_min(e, torch.zeros(0, device=device))
    assert u.shape == (0, 2) and u.dtype == torch.int32 and m.shape == (0,) and m.dtype == torch.float32
    u, gm = ck.create_grad_alphabrend_min(e.long(), torch.zeros(0, device=device))
    assert u.shape == (0, 2) and u.dtype == torch.int64 and gm.shape == (0,)
    one = torch.tensor([[7, 3]], dtype=torch.int32, device=device)
    u, m = ck.create_alpha_brend_min(one, torch.tensor([0.25], device=device))
    assert u.tolist() == [[7, 3]] and m.tolist() == [0.25]
    # amin hands a NaN through (scatter_reduce does), whatever else the pixel holds; +-inf are ordinary values
    r = torch.tensor([[1, 1], [2, 1], [1, 1], [2, 1], [0, 0], [0, 0]], dtype=torch.int32, device=device)
    v = torch.tensor([0.5, float("inf"), float("nan"), 3.0, float("-inf"), 1.0], device=device)
    u, m = ck.create_alpha_brend_min(r, v)
    assert u.tolist() == [[0, 0], [1, 1], [2, 1]]
    assert m[0].item() == float("-inf") and torch.isnan(m[1]).item() and m[2].item() == 3.0
    want = torch.zeros(3).scatter_reduce(0, torch.tensor([1, 2, 1, 2, 0, 0]), v.cpu(), reduce="amin", include_self=False)
    assert torch.equal(torch.isnan(m.cpu()), torch.isnan(want)) and torch.equal(m.cpu()[[0, 2]], want[[0, 2]])
    # a coordinate outside the image the caller named, and a negative one: refused, not written somewhere
    with pytest.raises(RuntimeError, match="outside"):
        ck.create_alpha_brend_min(r, v, image_size=(1, 1))
    bad = r.clone()
    bad[3, 0] = -2
    with pytest.raises(RuntimeError, match="negative|outside"):
        ck.create_alpha_brend_min(bad, v)
    with pytest.raises(RuntimeError, match="outside"):
        ck.create_alpha_brend_min(bad, v, image_size=(4, 4))
    with pytest.raises(RuntimeError, match="rows"):
        ck.create_alpha_brend_min(r, v[:-1])
    with pytest.raises(RuntimeError, match="no CPU path"):
        ck.create_alpha_brend_min(r.cpu(), v.cpu())


def test_mask_tensor_hands_tensors_through_only_when_the_call_dropped_nothing(device):
    """`_mask_tensor` (gs_model.py:525-531) under its own name: `tensor[mask]` — except for a mask that came from a call whose
    kept count (read back anyway) said nothing was dropped, where the tensors pass through without the six M-sized copies."""
    from simplegaussiansplat_tk71_amd import cuda_kernel as ck
    from simplegaussiansplat_tk71_amd import raster

    F = ck.custom_autograd_grouped_cumprod
    sc = make_scene(300, 63, 47, 5, seed=81)
    rects = F._create_rects(sc["start"].to(device), sc["end"].to(device))
    n = rects.size(0)
    g = torch.Generator().manual_seed(82)
    anti = (1.0 - 0.9 * torch.rand(n, generator=g)).to(device)
    other = torch.arange(n, device=device)
    for route in ("auto", "sort"):
        T, mask = ck.create_alpha_brend(rects, anti, "cumprod", route=route)
        assert T.numel() == n and bool(mask.all()) and raster.all_kept(mask), route
        a, b = F._mask_tensor(mask, anti, other)
        assert a is anti and b is other
        assert not raster.all_kept(mask[1:])          # a slice of it is an ordinary mask again
        S, smask = F.grad_cumsum(rects, anti)          # the reference's (flipped) mask keeps the mark
        assert raster.all_kept(smask) and F._mask_tensor(smask, other)[0] is other
    anti2 = anti.clone()
    anti2[::9] = 0.0
    for route in ("auto", "sort"):
        T, mask = ck.create_alpha_brend(rects, anti2, "cumprod", route=route)
        assert T.numel() < n and not raster.all_kept(mask), route
        a, b = F._mask_tensor(mask, anti2, other)
        assert torch.equal(a, anti2[mask]) and torch.equal(b, other[mask]) and a.numel() == T.numel()
    idx = torch.randperm(n, generator=g).to(device)
    assert torch.equal(F._sort_tensor(idx, other)[0], other[idx])
    hand_made = torch.ones(n, dtype=torch.bool, device=device)   # a mask of unknown origin is indexed with, whatever it holds
    assert F._mask_tensor(hand_made, other)[0] is not other


def test_the_remembered_extent_never_changes_a_result(device):
    """Without image_size the extent of the list is measured once per device and remembered; a later list is first tried
    against the remembered extent and measured only if a coordinate falls outside it.  Same results either way."""
    from oracle import wrappers as ow
    from simplegaussiansplat_tk71_amd import cuda_kernel as ck
    from simplegaussiansplat_tk71_amd import raster

    g = torch.Generator().manual_seed(9)

    def check(w, h, n):
        r = torch.stack((torch.randint(0, w + 1, (n,), generator=g), torch.randint(0, h + 1, (n,), generator=g)), 1).to(torch.int32)
        t = torch.rand(n, generator=g)
        wu, wm = ow.create_alpha_brend_min(r, t)
        u, m = ck.create_alpha_brend_min(r.to(device), t.to(device))
        assert torch.equal(u.cpu(), wu) and np.array_equal(_bits(m), _bits(wm)), (w, h, n)

    raster._extent_seen.pop(device.index, None)
    check(40, 30, 5000)            # measured: (<= 40, <= 30) remembered
    seen = raster._extent_seen[device.index]
    assert seen[0] <= 40 and seen[1] <= 30
    check(20, 10, 3000)            # fits the remembered extent: no measurement, a larger table than needed
    assert raster._extent_seen[device.index] == seen
    check(90, 12, 4000)            # wider: the kernel reports a coordinate outside, the extent is measured again and grows
    grown = raster._extent_seen[device.index]
    assert grown[0] > seen[0] and grown[1] >= seen[1]
    check(15, 70, 4000)            # taller
    assert raster._extent_seen[device.index][1] > grown[1]
    bad = torch.tensor([[3, 4], [-1, 2]], dtype=torch.int32, device=device)
    with pytest.raises(RuntimeError, match="negative"):
        ck.create_alpha_brend_min(bad, torch.ones(2, device=device))
    raster._extent_seen.pop(device.index, None)


def test_first_pair_index_travels_as_a_float_like_the_reference(device):
    """gs_model.py:728: the index goes through fp32 — exact below 2^24, rounded to nearest-even above, and the reference
    then reads the row it was rounded to.  The expectation is the statement itself, on the first indices constructed."""
    from simplegaussiansplat_tk71_amd import cuda_kernel as ck
    from simplegaussiansplat_tk71_amd import raster

    n = (1 << 24) + 4099
    fresh = 3000  # the last `fresh` pairs lie on pixels of their own: their first pair has an index above 2^24
    x = torch.zeros(n, dtype=torch.int32)
    y = torch.zeros(n, dtype=torch.int32)
    x[n - fresh:] = torch.arange(1, fresh + 1, dtype=torch.int32)
    y[n - fresh:] = 1
    rects = torch.stack((x, y), 1).to(device)
    u, first = raster.pixels_min(rects, None)
    want_first = torch.cat((torch.zeros(1), torch.arange(n - fresh, n, dtype=torch.int32).to(torch.float32)))
    assert u.size(0) == fresh + 1 and u[0].tolist() == [0, 0] and u[-1].tolist() == [fresh, 1]
    assert torch.equal(first.cpu(), want_first)
    assert (want_first[1:].long() != torch.arange(n - fresh, n)).any()  # the rounding is really exercised
    grad = torch.arange(n, dtype=torch.float32, device=device) * 0.5
    _, picked = ck.create_grad_alphabrend_min(rects, grad)
    assert torch.equal(picked.cpu(), grad.cpu()[want_first.long().clamp(max=n - 1)])


def test_cfg3_scene_properties(device):
    """1920x1080, 1M Gaussians, 1.65e8 pairs (BASELINE configs[2]): the distinct pixels are exactly the pixels the boxes
    cover, in (x, y) order; the minimum equals torch's own scatter-amin on the device; the first-pair index of every pixel is
    the smallest index that holds the pixel (checked through the pair it names: its coordinates are the pixel's, and no
    earlier pair of the pixel exists = scatter-amin of the indices in int64)."""
    from simplegaussiansplat_tk71_amd import cuda_kernel as ck
    from simplegaussiansplat_tk71_amd import raster, synthetic

    sc, rects, T, _ = synthetic.make_scene_pairs("cfg3", seed=5, device=device)
    w, h = int(sc["width"]), int(sc["height"])
    n = rects.size(0)
    assert n > 150_000_000
    u, m = ck.create_alpha_brend_min(rects, T, image_size=(w, h))
    key = rects[:, 0].long() * (h + 1) + rects[:, 1].long()  # (x, y) order
    cells = (w + 1) * (h + 1)
    tab = torch.full((cells,), float("inf"), device=device).scatter_reduce_(0, key, T, reduce="amin")
    covered = torch.nonzero(torch.isfinite(tab)).flatten()
    assert u.size(0) == covered.numel()
    assert torch.equal(u[:, 0].long() * (h + 1) + u[:, 1].long(), covered)
    assert torch.equal(m, tab[covered])
    del tab
    u2, first = raster.pixels_min(rects, None, (w, h))
    assert torch.equal(u2, u)
    idx = torch.full((cells,), n, dtype=torch.int64, device=device).scatter_reduce_(0, key, torch.arange(n, device=device), reduce="amin")
    assert torch.equal(first, idx[covered].to(torch.int32).to(torch.float32))
