"""HIP path vs the committed golden vectors (outputs of the reference's own kernels / Python path),
and vs the reference's whole extension compiled for gfx950 where oracle/_ref/ travelled along."""
import os
import sys

import numpy as np
import pytest
import torch

from tests.util import TOL, assert_parity, make_keys, make_values

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF_DIR = os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref")


def _cases():
    z = np.load(os.path.join(GOLD, "scan_golden.npz"))
    return z, sorted({k.split("/")[0] for k in z.files if k.startswith("n")})


@pytest.mark.parametrize("name", _cases()[1])
def test_scan_vs_reference_golden(device, name):
    import grouped_cumprod as gc

    z, _ = _cases()
    lens = torch.from_numpy(z[name + "/lens"].astype(np.int64))
    key = torch.repeat_interleave(torch.from_numpy(z[name + "/vals"]), lens).contiguous()
    x = ((torch.from_numpy(z[name + "/xq"].astype(np.int64)) + 1).to(torch.float32) / 65536.0).contiguous()
    kd, xd = key.to(device), x.to(device)
    y = torch.empty_like(xd)
    sel = slice(None)
    if name.endswith("_sampled"):
        sel = torch.from_numpy(z[name + "/idx"].astype(np.int64))
    gc.grouped_cumprod_forward(xd, kd, y)
    want = torch.from_numpy(z[name + "/cumprod"])
    assert_parity(y.cpu()[sel], want, None, f"{name} cumprod")  # transmittance: absolute 1e-5
    gc.grouped_cumsum_forward(xd, kd, y)
    want = torch.from_numpy(z[name + "/cumsum"])
    assert_parity(y.cpu()[sel], want, want, f"{name} cumsum")  # x > 0: the sum is its own scale


def _ref_gfx950():
    so = os.path.join(REF_DIR, "grouped_cumprod_ref_gfx950.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/grouped_cumprod_ref_gfx950.so not present")
    if REF_DIR not in sys.path:
        sys.path.insert(0, REF_DIR)
    import grouped_cumprod_ref_gfx950 as ref

    return ref


@pytest.mark.parametrize("dist", ["poisson8", "geo80", "mixed", "all1"])
def test_vs_reference_extension_on_this_gpu(device, dist):
    """The reference's cuda_kernel.cpp + three .cu files, compiled unmodified for gfx950 (rocThrust HIP
    backend) by oracle/Makefile, run on the same GPU and inputs: forward scans and the backward kernel."""
    import grouped_cumprod as gc
    from oracle import c_oracle as co

    ref = _ref_gfx950()
    n = 200_003
    key = make_keys(n, dist, 21)
    x = make_values(n, 21)
    go = make_values(n, 22, "normal")
    inv, inv_len = co.groups_from_key(key)
    kd, xd, god, invd, ild = (t.to(device) for t in (key, x, go, inv, inv_len))
    mine, theirs = torch.empty_like(xd), torch.zeros_like(xd)
    gc.grouped_cumprod_forward(xd, kd, mine)
    ref.grouped_cumprod_forward(xd, kd, theirs)
    torch.cuda.synchronize()
    assert_parity(mine, theirs.cpu(), None, "cumprod vs reference@gfx950")
    cp = theirs.clone()
    gc.grouped_cumsum_forward(god, kd, mine)
    ref.grouped_cumsum_forward(god, kd, theirs)
    torch.cuda.synchronize()
    assert_parity(mine, theirs.cpu(), co.cumsum_forward_f64(go.abs(), key), "cumsum vs reference@gfx950")
    gc.grouped_cumprod_backward(xd, cp, god, invd, mine, ild)
    ref.grouped_cumprod_backward(xd, cp, god, invd, theirs, ild)
    torch.cuda.synchronize()
    scale = co.cumprod_backward_f64(x, cp.cpu(), go.abs(), inv)
    assert_parity(mine, theirs.cpu(), scale, "backward vs reference@gfx950")


def test_wrappers_vs_reference_function_golden(device):
    """a5/a6 on the GPU vs what the reference's `_create_alpha_brend` / `grad_cumsum` returned:
    masks (index work) bit-exact, values within 1e-5."""
    import cuda_kernel as ck

    z = np.load(os.path.join(GOLD, "function_golden.npz"))
    for name in ("wrap_small", "wrap_mid"):
        rects = torch.from_numpy(z[name + "/rects"]).to(device)
        anti = torch.from_numpy(z[name + "/anti_opacity"]).to(device)
        grad = torch.from_numpy(z[name + "/grad"]).to(device)
        for route in ("sort", "boxes", "auto"):
            T, mask = ck.create_alpha_brend(rects, anti, "cumprod", route=route)
            assert np.array_equal(mask.cpu().numpy(), z[name + "/T_mask"]), route
            torch.testing.assert_close(T.cpu(), torch.from_numpy(z[name + "/T"]), atol=TOL, rtol=0)  # transmittance: absolute 1e-5
            S, smask = ck.grad_cumsum(rects, grad, route=route)
            # deliberate deviation: our mask is in ORIGINAL order, the reference leaves it flipped
            assert np.array_equal(smask.flip(0).cpu().numpy(), z[name + "/S_mask_flipped"]), route
            S_ref, smask_ref = ck.grad_cumsum(rects, grad, route=route, mask_order="reference")  # the reference's order, as is
            assert np.array_equal(smask_ref.cpu().numpy(), z[name + "/S_mask_flipped"]), route
            assert torch.equal(S_ref, S)
            torch.testing.assert_close(S.cpu(), torch.from_numpy(z[name + "/S"]), atol=TOL, rtol=TOL)


def test_autograd_functions(device):
    import cuda_kernel as ck
    from oracle import c_oracle as co
    from oracle import torch_path as tp

    n = 30000
    key = make_keys(n, "poisson8", 31)
    x = make_values(n, 31)
    go = make_values(n, 32, "normal")
    xd = x.to(device).requires_grad_(True)
    y = ck.grouped_cumprod(xd, key.to(device))
    y.backward(go.to(device))
    want = tp.grouped_cumprod_backward_autograd(x, key, go)  # torch autograd through torch.cumprod
    inv, _ = co.groups_from_key(key)
    scale = co.cumprod_backward_f64(x, co.cumprod_forward(x, key), go.abs(), inv)
    assert_parity(xd.grad, want, scale, "GroupedCumprod.backward")

    xs = make_values(n, 33, "normal")
    xsd = xs.to(device).requires_grad_(True)
    s = ck.grouped_cumsum(xsd, key.to(device))
    s.backward(go.to(device))
    want = co.cumsum_reverse(go, key)  # d/dx of a prefix sum is the suffix sum of the cotangent
    scale = co.cumsum_forward_f64(go.abs().flip(0).contiguous(), key.flip(0).contiguous()).flip(0)
    assert_parity(xsd.grad, want, scale, "GroupedCumsum.backward")


def test_full_size_properties_cfg3(device):
    """BASELINE.json's metric configuration (1920x1080, mean 80 splats/pixel, M ~ 1.66e8): checks that do
    not need the oracle at full size.
      * cumsum of ones == position inside the group (exact integers in fp32 up to 4096);
      * cumprod of a constant c == c^k; last element of each group == c^len;
      * backward with grad_out = 1, param = 1: grad_in = remaining length (exact);
      * a 16M-pair prefix agrees with the C oracle."""
    import grouped_cumprod as gc
    from oracle import c_oracle as co
    from simplegaussiansplat_tk71_amd import synthetic

    p = synthetic.make_config("cfg3", seed=1, device=device)
    m = p.n_pairs
    ones = torch.ones(m, device=device)
    y = torch.empty(m, device=device)
    gc.grouped_cumsum_forward(ones, p.key, y)
    starts = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), p.inv_len[:-1].long()])
    pos = torch.arange(m, device=device) - starts[p.inv.long()] + 1
    assert torch.equal(y, pos.float())
    gc.grouped_cumsum_reverse(ones, p.key, y)
    rem = p.inv_len.long()[p.inv.long()] - torch.arange(m, device=device)
    assert torch.equal(y, rem.float())
    gc.grouped_cumprod_backward(ones, ones, ones, p.inv, y, p.inv_len)
    assert torch.equal(y, rem.float())
    half = torch.full((m,), 0.5, device=device)
    gc.grouped_cumprod_forward(half, p.key, y)
    # powers of two are exact in fp32, denormals included (2^-149 is the last non-zero one)
    table = torch.pow(torch.tensor(0.5, dtype=torch.float64), torch.arange(0, 4098, dtype=torch.float64)).float()
    assert table[149] > 0 and table[150] == 0
    assert torch.equal(y, table.to(device)[pos])
    # runs are clipped at 4096 = one tile: only the few tiles whose group
    # already spans the 256 elements behind them and wave 0's whole share take their carry from the descriptor tree
    assert gc.last_fallback_tiles(device) + gc.last_lookback_tiles(device) < 0.06 * (m // 4096)
    del ones, half, pos, rem
    cut = int(p.inv_len[int(torch.searchsorted(p.inv_len, 16_000_000))].item())
    gc.grouped_cumprod_forward(p.x, p.key, y)
    want = co.cumprod_forward(p.x[:cut].cpu(), p.key[:cut].cpu())
    assert_parity(y[:cut], want, None, "cfg3 prefix vs oracle")


def test_gpu_error_vs_fp64_is_no_worse_than_the_sequential_fp32_path(device):
    """The GPU associates differently from the reference's left-to-right order (lane-serial, then tree); measured
    against fp64 it must not be LESS accurate than the sequential fp32 CPU path it is compared with."""
    import grouped_cumprod as gc
    from oracle import c_oracle as co
    from simplegaussiansplat_tk71_amd import synthetic

    p = synthetic.make_pairs(300, 400, 80.0, deep=True, seed=6)  # ~9.6M pairs, cfg3's run-length mix
    x, key, inv, inv_len, go = p.x, p.key, p.inv, p.inv_len, p.grad_out
    y = torch.empty(p.n_pairs, device=device)
    gc.grouped_cumprod_forward(x.to(device), key.to(device), y)
    f64 = co.cumprod_forward_f64(x, key)
    o32 = co.cumprod_forward(x, key)
    rms = lambda a: float(((a.double() - f64) ** 2).mean().sqrt())  # noqa: E731
    e_gpu, e_cpu = rms(y.cpu()), rms(o32)
    assert e_gpu <= 1.25 * e_cpu + 1e-12, (e_gpu, e_cpu)

    g = torch.empty(p.n_pairs, device=device)
    gc.grouped_cumprod_backward(x.to(device), o32.to(device), go.to(device), inv.to(device), g, inv_len.to(device))
    b64 = co.cumprod_backward_f64(x, o32, go, inv)
    b32 = co.cumprod_backward(x, o32, go, inv, inv_len)
    rmsb = lambda a: float(((a.double() - b64) ** 2).mean().sqrt())  # noqa: E731
    e_gpu, e_cpu = rmsb(g.cpu()), rmsb(b32)
    assert e_gpu <= 1.25 * e_cpu + 1e-12, (e_gpu, e_cpu)
    print(f"rms error vs fp64: forward gpu {rms(y.cpu()):.3g} cpu-fp32 {rms(o32):.3g}; backward gpu {e_gpu:.3g} cpu-fp32 {e_cpu:.3g}")


def test_wrappers_vs_reference_golden_cumsum_and_cutting_number(device):
    """a5/a6 on the GPU — sort route (image size given / key range read back) and boxes route — vs what the reference's own
    `_create_alpha_brend` / `grad_cumsum` returned on CPU for flag="cumsum" and for `cutting_number` (the carry rows of its
    chunked calls, gs_model.py:557-559; tests/golden/wrappers_golden.npz): masks bit-exact, values within 1e-5."""
    import cuda_kernel as ck

    z = np.load(os.path.join(GOLD, "wrappers_golden.npz"))
    for name in ("w_tiny", "w_small", "w_mid"):
        rects = torch.from_numpy(z[name + "/rects"]).to(device)
        anti = torch.from_numpy(z[name + "/anti_opacity"]).to(device)
        grad = torch.from_numpy(z[name + "/grad"]).to(device)
        w, h = (int(v) for v in z[name + "/width_height"])
        for c in z[name + "/cuts"].tolist():
            cut, tag = (None, "none") if c < 0 else (c, str(c))
            if f"{name}/cumprod_{tag}/values" not in z.files:
                continue
            for kw in ({"image_size": (w, h), "route": "sort"}, {"route": "sort"}, {"route": "boxes"}, {}):
                for flag in ("cumprod", "cumsum"):
                    v, m = ck.create_alpha_brend(rects, anti, flag, cut, **kw)
                    assert np.array_equal(m.cpu().numpy(), z[f"{name}/{flag}_{tag}/mask"]), (name, flag, tag)
                    # transmittance: 1e-5 absolute (north_star); prefix sums of values in [0, 1]: 1e-5 relative to their size
                    tol = dict(atol=TOL, rtol=0) if flag == "cumprod" else dict(atol=2e-5, rtol=TOL)
                    torch.testing.assert_close(v.cpu(), torch.from_numpy(z[f"{name}/{flag}_{tag}/values"]), **tol)
                s, sm = ck.grad_cumsum(rects, grad, cut, **kw)
                # deliberate deviation: our mask is in ORIGINAL order, the reference leaves it flipped
                assert np.array_equal(sm.flip(0).cpu().numpy(), z[f"{name}/grad_cumsum_{tag}/mask_flipped"]), (name, tag)
                # mask_order="reference": the reference's own (flipped) mask, bit for bit, and the same values
                s_ref, sm_ref = ck.grad_cumsum(rects, grad, cut, mask_order="reference", **kw)
                assert np.array_equal(sm_ref.cpu().numpy(), z[f"{name}/grad_cumsum_{tag}/mask_flipped"]), (name, tag)
                assert torch.equal(s_ref, s)
                torch.testing.assert_close(s.cpu(), torch.from_numpy(z[f"{name}/grad_cumsum_{tag}/values"]), atol=2e-5, rtol=TOL)
