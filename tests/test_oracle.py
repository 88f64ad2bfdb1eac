"""CPU tests that pin the oracle: reference known-answer vectors, golden vectors produced by the
reference's own kernels (tests/golden/), the live host build of the reference where present, and
the pure-PyTorch torch.cumprod statement."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import torch_path as tp
from oracle import wrappers as ow
from tests.util import make_keys, make_values

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_kat_cuda_test_py():
    """reference: cuda_test.py:19-34; expected backward is printed at :34."""
    param = torch.tensor([0.4, 0.2, 0.1, 0.8, 0.2])
    index = torch.tensor([0, 0, 1, 1, 2], dtype=torch.int32)
    index_len = torch.tensor([2, 4, 5], dtype=torch.int32)
    cp = co.cumprod_forward(param, index)
    torch.testing.assert_close(cp, torch.tensor([0.4, 0.08, 0.1, 0.08, 0.2]), atol=1e-7, rtol=1e-6)
    g = co.cumprod_backward(param, cp, param.clone(), index, index_len)
    torch.testing.assert_close(g, torch.tensor([0.44, 0.08, 0.74, 0.08, 0.2]), atol=1e-6, rtol=1e-6)
    inv, inv_len = co.groups_from_key(index)
    assert torch.equal(inv, index) and torch.equal(inv_len, index_len)


def test_kat_uitility_worked_example():
    """reference: uitility.py:383-393 — A=[1..7], groups {(1,1),(1,2),(1,1),(1,2),(1,3),(1,1),(1,3)}
    -> [1,2,3,8,5,18,35]: grouped cumprod returned in the ORIGINAL (unsorted) order, i.e. the
    sort -> scan -> unsort of gs_model.py:547-555."""
    A = torch.arange(1, 8, dtype=torch.float32)
    G = torch.tensor([[1, 1], [1, 2], [1, 1], [1, 2], [1, 3], [1, 1], [1, 3]], dtype=torch.int32)
    T, mask, _, _ = ow.create_alpha_brend(G, A, "cumprod")
    assert bool(mask.all())
    # the wrapper returns inclusive/self = exclusive product; inclusive = that * A
    torch.testing.assert_close(T * A, torch.tensor([1.0, 2.0, 3.0, 8.0, 5.0, 18.0, 35.0]))


def _golden_cases():
    z = np.load(os.path.join(GOLD, "scan_golden.npz"))
    names = sorted({k.split("/")[0] for k in z.files if k.startswith("n")})
    return z, names


def _golden_inputs(z, name):
    lens = torch.from_numpy(z[name + "/lens"].astype(np.int64))
    vals = torch.from_numpy(z[name + "/vals"])
    key = torch.repeat_interleave(vals, lens).contiguous()
    x = ((torch.from_numpy(z[name + "/xq"].astype(np.int64)) + 1).to(torch.float32) / 65536.0).contiguous()
    return x, key


def test_golden_file_kat():
    z, _ = _golden_cases()
    np.testing.assert_allclose(z["kat/cumprod"], [0.4, 0.08, 0.1, 0.08, 0.2], rtol=1e-6)
    np.testing.assert_allclose(z["kat/cumsum"], [0.4, 0.6, 0.1, 0.9, 0.2], rtol=1e-6)


@pytest.mark.parametrize("name", _golden_cases()[1])
def test_oracle_bit_exact_vs_reference_golden(name):
    """Golden outputs come from thrust::inclusive_scan_by_key (reference .cu, host backend):
    sequential fp32, so the C oracle must reproduce them BIT FOR BIT."""
    z, _ = _golden_cases()
    x, key = _golden_inputs(z, name)
    cp, cs = co.cumprod_forward(x, key), co.cumsum_forward(x, key)
    if name.endswith("_sampled"):
        idx = torch.from_numpy(z[name + "/idx"].astype(np.int64))
        assert np.array_equal(cp[idx].numpy(), z[name + "/cumprod"])
        assert np.array_equal(cs[idx].numpy(), z[name + "/cumsum"])
        assert cp.double().sum().item() == float(z[name + "/cumprod_sum64"])
        assert cs.double().sum().item() == float(z[name + "/cumsum_sum64"])
    else:
        assert np.array_equal(cp.numpy(), z[name + "/cumprod"])
        assert np.array_equal(cs.numpy(), z[name + "/cumsum"])


def test_oracle_vs_live_reference_host_build():
    """Where oracle/_ref/ exists (built from /root/reference by oracle/Makefile) run the reference's
    own forward kernels live against the C oracle on fresh inputs."""
    ref_dir = os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref")
    if not os.path.exists(os.path.join(ref_dir, "grouped_cumprod_ref_host.so")):
        pytest.skip("oracle/_ref/grouped_cumprod_ref_host.so not built")
    sys.path.insert(0, ref_dir)
    import grouped_cumprod_ref_host as ref

    for n, dist in [(1, "all1"), (777, "poisson8"), (50000, "geo80"), (50000, "mixed"), (9000, "one_run")]:
        key, x = make_keys(n, dist, n), make_values(n, n)
        y = torch.zeros_like(x)
        ref.grouped_cumprod_forward(x, key, y)
        assert torch.equal(y, co.cumprod_forward(x, key))
        ref.grouped_cumsum_forward(x, key, y)
        assert torch.equal(y, co.cumsum_forward(x, key))


@pytest.mark.parametrize("dist", ["all1", "poisson8", "geo80", "mixed"])
def test_torch_cumprod_path_matches_c_oracle(dist):
    """BASELINE config 1 (256x256, ~8 splats/pixel scale): the pure-PyTorch torch.cumprod path and the
    C restatement agree to fp32 rounding (torch.cumprod is also a left-to-right product)."""
    n = 65536 * 8 if dist == "poisson8" else 60000
    key, x = make_keys(n, dist, 3), make_values(n, 3)
    torch.testing.assert_close(tp.grouped_cumprod(x, key), co.cumprod_forward(x, key), atol=1e-6, rtol=1e-5)
    xs = make_values(n, 4, "normal")
    want = co.cumsum_forward(xs, key)
    scale = co.cumsum_forward_f64(xs.abs(), key)
    assert bool(((tp.grouped_cumsum(xs, key).double() - want.double()).abs() <= 1e-5 * (1 + scale)).all())


def test_backward_is_the_vjp_of_cumprod():
    """For non-zero params the reference kernel (grouped_cumprod_backward.cu:22-29) is the true VJP:
    compare the literal restatement with torch autograd through per-group torch.cumprod, and with
    the O(n) fp64 form."""
    n = 4000
    key, x = make_keys(n, "poisson8", 9), make_values(n, 9)
    inv, inv_len = co.groups_from_key(key)
    go = make_values(n, 10, "normal")
    cp = co.cumprod_forward(x, key)
    lit = co.cumprod_backward(x, cp, go, inv, inv_len)
    auto = tp.grouped_cumprod_backward_autograd(x, key, go)
    f64 = co.cumprod_backward_f64(x, cp, go, inv)
    scale = co.cumprod_backward_f64(x, cp, go.abs(), inv)
    assert bool(((lit.double() - f64).abs() <= 1e-5 * (1 + scale)).all())
    assert bool(((auto.double() - f64).abs() <= 1e-5 * (1 + scale)).all())


def test_backward_zero_param_rule():
    """grouped_cumprod_backward.cu:25: a zero param is replaced by 1e-8 in the divisor."""
    x = torch.tensor([0.5, 0.0, 0.25])
    key = torch.zeros(3, dtype=torch.int32)
    inv, inv_len = co.groups_from_key(key)
    cp = co.cumprod_forward(x, key)  # [0.5, 0, 0]
    go = torch.tensor([1.0, 1.0, 1.0])
    g = co.cumprod_backward(x, cp, go, inv, inv_len)
    torch.testing.assert_close(g, torch.tensor([1.0, 0.0, 0.0]))


def test_wrappers_vs_reference_function_golden():
    """a5/a6: the restated wrappers reproduce what the reference's own `_create_alpha_brend` and
    `grad_cumsum` returned (tests/golden/function_golden.npz), indices bit-exact."""
    z = np.load(os.path.join(GOLD, "function_golden.npz"))
    for name in ("wrap_small", "wrap_mid"):
        rects = torch.from_numpy(z[name + "/rects"])
        anti = torch.from_numpy(z[name + "/anti_opacity"])
        T, mask, sorted_inv, index = ow.create_alpha_brend(rects, anti, "cumprod")
        assert np.array_equal(sorted_inv.numpy(), z[name + "/sorted_inv"])
        assert np.array_equal(index.numpy(), z[name + "/index"])
        assert np.array_equal(mask.numpy(), z[name + "/T_mask"])
        assert np.array_equal(T.numpy(), z[name + "/T"])
        grad = torch.from_numpy(z[name + "/grad"])
        S, smask = ow.grad_cumsum(rects, grad)
        assert np.array_equal(smask.numpy(), z[name + "/S_mask_flipped"])
        assert np.array_equal(S.numpy(), z[name + "/S"])


def test_empty_inputs():
    e, k = torch.zeros(0), torch.zeros(0, dtype=torch.int32)
    assert co.cumprod_forward(e, k).numel() == 0
    assert co.cumsum_forward(e, k).numel() == 0
    assert co.cumprod_backward(e, e, e, k, k).numel() == 0
    assert tp.grouped_cumprod(e, k).numel() == 0


@pytest.mark.parametrize("name", ["fn_6g_16x12", "fn_200g_64x48", "fn_1500g_128x96"])
def test_dense_renderer_vs_reference_function_golden(name):
    """a7: the dense autograd restatement reproduces the image and the opacity / precision-matrix
    gradients the reference's own Function produced (single chunk = the parity contract).  The
    reference's colour gradient is channel-collapsed (gs_model.py:710-712, :763-766): the golden holds
    it as informational, and the dense oracle's is the hypothesis sum_pairs(g.p)/l_c of SURVEY §0 Q2."""
    from oracle import dense_render as dr

    z = np.load(os.path.join(GOLD, "function_golden.npz"))
    g = lambda k: torch.from_numpy(z[f"{name}/{k}"])  # noqa: E731
    w, h = (int(v) for v in z[name + "/width_height"])
    img, gv, go, gl = dr.render_with_grads(g("start"), g("end"), g("mean"), g("vinv"), g("opacity"), g("l_d"), w, h, g("wimg"))
    torch.testing.assert_close(img.float(), g("image"), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(go.float(), g("grad_opacity"), atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(gv.float(), g("grad_vinv"), atol=2e-5, rtol=1e-4)
    # fp32 forward of the same restatement
    img32 = dr.render(g("start"), g("end"), g("mean"), g("vinv"), g("opacity"), g("l_d"), w, h)
    torch.testing.assert_close(img32, g("image"), atol=1e-5, rtol=1e-5)
    # the reference's colour gradient differs from the true one by design
    assert not torch.allclose(gl.float(), g("grad_l_REFERENCE_BUGGY"), atol=1e-3, rtol=1e-2)


@pytest.mark.parametrize("name", ["deep_300", "deep_700"])
def test_dense_renderer_vs_reference_function_deep_golden(name):
    """The dense fp64 oracle against the reference's own Function at DEPTH (300 / 700 layers per pixel, gradients from
    1e-18 to 19; tests/golden/function_deep_golden.npz): image, and every Gaussian's opacity / precision-matrix gradient
    within 2e-5 of its own condition scale (tests/util.full_cover_scales: the sum over pixels of the |terms| whose signed sum
    it is; the reference is fp32) — the oracle the HIP backward is held to is itself pinned to the reference where the old
    total-minus-prefix backward went wrong."""
    from oracle import dense_render as dr
    from tests.util import full_cover_scales

    z = np.load(os.path.join(GOLD, "function_deep_golden.npz"))
    g = lambda k: torch.from_numpy(z[f"{name}/{k}"])  # noqa: E731
    w, h = (int(v) for v in z[name + "/width_height"])
    img, gv, go, gl = dr.render_with_grads(g("start"), g("end"), g("mean"), g("vinv"), g("opacity"), g("l_d"), w, h, g("wimg"))
    torch.testing.assert_close(img.float(), g("image"), atol=1e-5, rtol=1e-5)
    n = go.shape[0]
    sc = dict(mean=g("mean"), vinv=g("vinv"), opacity=g("opacity"), l_d=g("l_d"), wimg=g("wimg"), width=w, height=h)
    scales, _ = full_cover_scales(sc, n)
    for what, got, want in (("opacity", go, g("grad_opacity")), ("vinv", gv, g("grad_vinv"))):
        got, want = got.reshape(n, -1), want.double().reshape(n, -1)
        assert bool(((got - want).abs() <= 2e-5 * scales[what] + 1e-30).all()), what
    assert float(g("grad_opacity").abs().min()) < 1e-8


def test_reference_backward_breaks_when_a_suffix_sum_is_exactly_zero():
    """SURVEY §0 Q9 (found while building the goldens): `grad_cumsum` returns its mask in FLIPPED order
    (gs_model.py:720-722) and `_backward_batch` applies it to un-flipped tensors (:642-645).  Harmless while the
    mask is all True; with narrow Gaussians in wide boxes g underflows to exactly 0, a pixel's deepest pair then
    has an exactly-zero suffix sum and the reference's gradients come out wrong, while its forward image is fine.
    The golden keeps that case as information; the dense oracle (true gradients) is what the GPU path is held to."""
    from oracle import dense_render as dr

    z = np.load(os.path.join(GOLD, "function_golden.npz"))
    name = "fn_300g_64x48_Q9_INFORMATIONAL"
    g = lambda k: torch.from_numpy(z[f"{name}/{k}"])  # noqa: E731
    w, h = (int(v) for v in z[name + "/width_height"])
    img, gv, go, gl = dr.render_with_grads(g("start"), g("end"), g("mean"), g("vinv"), g("opacity"), g("l_d"), w, h, g("wimg"))
    torch.testing.assert_close(img.float(), g("image"), atol=1e-5, rtol=1e-5)
    assert (go.float() - g("grad_opacity")).abs().max() > 0.1  # the reference's own gradient is off by O(1)


def test_multithreaded_oracle_is_bit_identical():
    """bench.py's CPU baseline uses the OpenMP forms; they must give exactly the single-thread results."""
    n = 200_000
    key, x = make_keys(n, "mixed", 12), make_values(n, 12)
    inv, inv_len = co.groups_from_key(key)
    go = make_values(n, 13, "normal")
    y1 = co.cumprod_forward(x, key)
    assert torch.equal(co.cumprod_forward_mt(x, inv_len, 4), y1)
    g1 = co.cumprod_backward(x, y1, go, inv, inv_len)
    assert torch.equal(co.cumprod_backward_mt(x, y1, go, inv, inv_len, 4), g1)
    assert co.max_threads() >= 1


def test_wrappers_oracle_equals_the_reference_with_cumsum_and_cutting_number():
    """oracle/wrappers.py against the reference's own `_create_alpha_brend` / `grad_cumsum` run on CPU
    (tests/golden/wrappers_golden.npz, generator make_wrappers_golden.py): flag="cumsum" and `cutting_number` with both
    flags and through grad_cumsum — the branches function_golden.npz does not hold.  Masks bit-exact, values exact (the C
    oracle reproduces the reference's host Thrust scans bit for bit)."""
    import os

    import numpy as np

    from oracle import wrappers as ow

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "wrappers_golden.npz"))
    n_checked = 0
    for name in ("w_tiny", "w_small", "w_mid"):
        rects = torch.from_numpy(z[name + "/rects"])
        anti = torch.from_numpy(z[name + "/anti_opacity"])
        grad = torch.from_numpy(z[name + "/grad"])
        for c in z[name + "/cuts"].tolist():
            cut, tag = (None, "none") if c < 0 else (c, str(c))
            if f"{name}/cumprod_{tag}/values" not in z.files:
                continue
            for flag in ("cumprod", "cumsum"):
                v, m, _, _ = ow.create_alpha_brend(rects, anti, flag, cut)
                assert np.array_equal(m.numpy(), z[f"{name}/{flag}_{tag}/mask"]), (name, flag, tag)
                assert np.array_equal(v.numpy(), z[f"{name}/{flag}_{tag}/values"]), (name, flag, tag)
            s, sm = ow.grad_cumsum(rects, grad, cut)
            assert np.array_equal(sm.numpy(), z[f"{name}/grad_cumsum_{tag}/mask_flipped"]), (name, tag)
            assert np.array_equal(s.numpy(), z[f"{name}/grad_cumsum_{tag}/values"]), (name, tag)
            n_checked += 1
    assert n_checked >= 8


def _carry_golden():
    import os

    import numpy as np

    return np.load(os.path.join(os.path.dirname(__file__), "golden", "carry_golden.npz"))


def test_carry_oracle_equals_the_reference_helpers():
    """oracle/wrappers.py's `create_alpha_brend_min` / `create_grad_alphabrend_min` against the reference's own
    (gs_model.py:582-586, :724-730) run on CPU (tests/golden/carry_golden.npz, generator make_carry_golden.py): lists of
    boxes, a list thinned by a mask, int64 lists, values of both signs and both zeros.  Bit for bit."""
    import numpy as np

    from oracle import wrappers as ow

    z = _carry_golden()
    for name in ("m_tiny", "m_small", "m_mid"):
        rects = torch.from_numpy(z[name + "/rects"])
        for tag in ("T", "signed"):
            u, m = ow.create_alpha_brend_min(rects, torch.from_numpy(z[f"{name}/{tag}"]))
            assert np.array_equal(u.numpy(), z[f"{name}/min_{tag}/unique_rects"]), (name, tag)
            assert np.array_equal(m.numpy().view(np.int32), z[f"{name}/min_{tag}/values"].view(np.int32)), (name, tag)
        u, gm = ow.create_grad_alphabrend_min(rects, torch.from_numpy(z[name + "/grad"]))
        assert np.array_equal(u.numpy(), z[name + "/grad_min/unique_rects"]) and np.array_equal(gm.numpy(), z[name + "/grad_min/values"])
        keep = torch.from_numpy(z[name + "/masked/keep"])
        u, m = ow.create_alpha_brend_min(rects[keep], torch.from_numpy(z[name + "/T"])[keep])
        assert np.array_equal(u.numpy(), z[name + "/masked/unique_rects"]) and np.array_equal(m.numpy(), z[name + "/masked/values"])
        u, _ = ow.create_alpha_brend_min(rects.long(), torch.from_numpy(z[name + "/T"]))
        assert u.dtype == torch.int64 and np.array_equal(u.numpy(), z[name + "/min_T_i64/unique_rects"])


def carry_chain(z, name, F, rects_of, to_dev=lambda t: t):
    """The reference's chunk loop (gs_model.py:601-615 forward, :634-643 backward, last chunk first) on the functions of
    `F` (the oracle's, or the HIP module's under the reference's names), compared step by step with what the reference
    itself returned (carry_golden.npz).  rects_of(start, end) expands a chunk's boxes; exact=False: values within the
    tolerance rule (the HIP scans associate differently), masks / pixels / carried rows bit for bit."""
    import numpy as np

    ends = z[name + "/chunk_ends"].tolist()
    start, end = torch.from_numpy(z[name + "/start"]), torch.from_numpy(z[name + "/end"])
    got = {}
    unique_rects, T_min, kept = None, None, []
    for c in range(len(ends)):
        s0 = ends[c - 1] if c else 0
        rects = rects_of(to_dev(start[s0:ends[c]]), to_dev(end[s0:ends[c]]))
        anti = to_dev(torch.from_numpy(z[f"{name}/fwd{c}/anti_opacity"]))
        if unique_rects is None:
            T, mask = F._create_alpha_brend(rects, anti, flag="cumprod")[:2]
            rects = rects[mask]
            unique_rects, T_min = F._create_alpha_brend_min(rects, T)
        else:
            cat_anti, cat_rects = F._cat_alpha_brend([T_min, anti], [unique_rects, rects])
            T, mask = F._create_alpha_brend(cat_rects, cat_anti, flag="cumprod", cutting_number=len(unique_rects))[:2]
            rects = rects[mask]
            cat_anti, cat_rects = F._cat_alpha_brend([T_min, T], [unique_rects, rects])
            unique_rects, T_min = F._create_alpha_brend_min(cat_rects, cat_anti)
        kept.append(rects)
        got[f"fwd{c}/T"], got[f"fwd{c}/mask"], got[f"fwd{c}/unique_rects"], got[f"fwd{c}/T_min"] = T, mask, unique_rects, T_min
    unique_rects, grad_cumsum_0 = None, None
    for c in reversed(range(len(ends))):
        rects = kept[c]
        pixel_grad = to_dev(torch.from_numpy(z[f"{name}/bwd{c}/pixel_grad"]))
        if unique_rects is not None:
            cat_grad, cat_rects = F._cat_alpha_brend([pixel_grad, grad_cumsum_0], [rects, unique_rects])
            pixel_grad_cumsum, mask = F.grad_cumsum(cat_rects, cat_grad, len(unique_rects))
            rects = rects[mask]
            cat_grad, cat_rects = F._cat_alpha_brend([pixel_grad_cumsum, grad_cumsum_0], [rects, unique_rects])
            unique_rects, grad_cumsum_0 = F.create_grad_alphabrend_min(cat_rects, cat_grad)
        else:
            pixel_grad_cumsum, mask = F.grad_cumsum(rects, pixel_grad)
            rects = rects[mask]
            unique_rects, grad_cumsum_0 = F.create_grad_alphabrend_min(rects, pixel_grad_cumsum)
        got[f"bwd{c}/pixel_grad_cumsum"], got[f"bwd{c}/mask"] = pixel_grad_cumsum, mask
        got[f"bwd{c}/unique_rects"], got[f"bwd{c}/grad_cumsum_0"] = unique_rects, grad_cumsum_0
    return {k: v.cpu().numpy() for k, v in got.items()}


def test_carry_oracle_runs_the_reference_chunk_loop_to_the_reference_results():
    import types

    import numpy as np

    from oracle import wrappers as ow

    F = types.SimpleNamespace(_create_alpha_brend=ow.create_alpha_brend, _create_alpha_brend_min=ow.create_alpha_brend_min,
                              _cat_alpha_brend=ow.cat_alpha_brend, grad_cumsum=ow.grad_cumsum,
                              create_grad_alphabrend_min=ow.create_grad_alphabrend_min)
    z = _carry_golden()

    def rects_of(start, end):  # uitility.py:336-366, box after box
        rows = []
        for (x0, y0), (x1, y1) in zip(start.tolist(), end.tolist()):
            ys, xs = torch.meshgrid(torch.arange(y0, y1 + 1), torch.arange(x0, x1 + 1), indexing="ij")
            rows.append(torch.stack((xs.flatten(), ys.flatten()), 1))
        return torch.cat(rows).to(torch.int32)

    for name in ("chain_small", "chain_mid"):
        got = carry_chain(z, name, F, rects_of)
        for k, v in got.items():
            assert np.array_equal(v, z[f"{name}/{k}"]), (name, k)
