"""PyTorch restatement, op for op, of the reference's model forward up to the Function call
(reference: gs_model.py:277-425; helpers uitility.py:231-287, :431-462) — the checker of the fused projection kernels
(csrc/gcp_project.hip).  Pinned against the reference's own forward run on CPU: tests/golden/forward_golden.npz
(tests/test_gs_model.py::test_camera_inputs_match_reference_forward): integers bit-exact, floats to 1e-6.

Differences from the reference's text, none of which changes a result on the golden scenes: the 3-sigma box comes
from a closed-form 2x2 eigen-decomposition (`box_halfsize`; the reference calls torch.linalg.eigh on the CPU,
gs_model.py:327-332), the depth sort is stable, tensors live where the inputs live.  `eval_sh` stands in for the
reference's `sh_utility.eval_sh`, which is absent from its checkout (gs_model.py:9,335): real spherical harmonics up
to degree 2 in the usual 3DGS convention — parity unpinned.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the product never imports this.
"""
import torch

def qvec_to_rotmat_batch(q):
    """(N, 4) unit quaternions in (x, y, z, w) order -> (N, 3, 3) (reference: uitility.py:231-254)."""
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    r0 = torch.stack([1 - 2 * (y**2 + z**2), 2 * (x * y - w * z), 2 * (x * z + w * y)], dim=1)
    r1 = torch.stack([2 * (x * y + w * z), 1 - 2 * (x**2 + z**2), 2 * (y * z - w * x)], dim=1)
    r2 = torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x**2 + y**2)], dim=1)
    return torch.stack([r0, r1, r2], dim=1)


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


def box_clamp(wh, tile_max_width, dev):
    """Upper bound of the 3-sigma half extents: 10 * sqrt(W*H) * sigmoid(tile_max_width) (gs_model.py:364-365)."""
    tile_max = torch.sqrt((wh[0, 0] * wh[0, 1]).to(torch.int32).to(torch.float32)) * torch.sigmoid(
        torch.as_tensor(tile_max_width, dtype=torch.float32, device=dev))
    return (tile_max * 10).item()


def camera_inputs(mean, variance_q, variance_scale, opacity, color, P, K, wh, tile_max_width, L_max=2, sh=eval_sh):
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
    half_z = half[cam, z_index].clamp(max=box_clamp(wh, tile_max_width, dev)).to(torch.int32)

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
