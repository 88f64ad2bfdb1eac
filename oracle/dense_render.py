"""Dense autograd restatement of the reference's rasterise-and-blend Function.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Spec: 自動微分の成分表示.md:50-74 (eq. 6-9) and the forward of
`custom_autograd_grouped_cumprod` (gs_model.py:598-624, :666-692), single chunk:

  for every Gaussian i in depth order (= input order) and every pixel r inside its
  integer box [startpoint_i, endpoint_i] (inclusive, uitility.py:336-366):
      g   = exp(-0.5 (r-m_i) Λ_i (r-m_i)^T)                       eq. 8 / gs_model.py:495
      T   = Π_{k<i, r in box_k} (1 - o_k g_k)                     eq. 6 (exclusive)
      p   = T · l_i · o_i · g                                     eq. 9 / gs_model.py:500
      I[r_y, r_x, :] += p                                         gs_model.py:510-514
  a pair whose INCLUSIVE product T·(1 - o g) is exactly 0 is dropped (gs_model.py:560,:575-578)
  image layout (H+1, W+1, 3) (gs_model.py:505)

No pair lists, no sort: one dense [H+1, W+1] update per Gaussian, differentiable by torch autograd
(so its gradients are the true ones; the reference's hand-written ∂L/∂l is known-buggy).
Pinned against the reference's own outputs in tests/golden/function_golden.npz
(tests/test_oracle.py::test_dense_renderer_vs_reference_function_golden).
"""
import torch


def render(start, end, mean, vinv, opacity, l_d, width, height, dtype=torch.float32):
    """start/end/mean: int [N,2] (x,y); vinv [N,2,2]; opacity [N,1]; l_d [N,3] -> image [H+1, W+1, 3]."""
    n = start.shape[0]
    ys = torch.arange(height + 1, dtype=dtype)[:, None]
    xs = torch.arange(width + 1, dtype=dtype)[None, :]
    T = torch.ones(height + 1, width + 1, dtype=dtype)
    img = torch.zeros(height + 1, width + 1, 3, dtype=dtype)
    vinv = vinv.to(dtype)
    opacity = opacity.to(dtype)
    l_d = l_d.to(dtype)
    # a float `mean` that requires grad is differentiated through (the reference passes ints); plain numbers otherwise
    mean_t = mean.to(dtype) if mean.requires_grad else mean.to(dtype).tolist()
    for i in range(n):
        x0, y0 = int(start[i, 0]), int(start[i, 1])
        x1, y1 = int(end[i, 0]), int(end[i, 1])
        if x1 < x0 or y1 < y0:
            continue
        dx = xs[:, x0 : x1 + 1] - mean_t[i][0]
        dy = ys[y0 : y1 + 1, :] - mean_t[i][1]
        a, b, c, d = vinv[i, 0, 0], vinv[i, 0, 1], vinv[i, 1, 0], vinv[i, 1, 1]
        q = (dx * a + dy * c) * dx + (dx * b + dy * d) * dy  # (d Λ) d^T, association of gs_model.py:495
        g = torch.exp(-0.5 * q)
        Tb = T[y0 : y1 + 1, x0 : x1 + 1]
        incl = Tb * (1.0 - opacity[i, 0] * g)
        keep = incl != 0
        contrib = torch.where(keep, Tb * opacity[i, 0] * g, torch.zeros((), dtype=dtype))
        pad = torch.zeros(height + 1, width + 1, dtype=dtype)
        pad = torch.nn.functional.pad(contrib, (x0, width - x1, y0, height - y1))
        img = img + pad[:, :, None] * l_d[i][None, None, :]
        T = T * torch.nn.functional.pad(1.0 - opacity[i, 0] * g, (x0, width - x1, y0, height - y1), value=1.0)
    return img


def render_with_grads(start, end, mean, vinv, opacity, l_d, width, height, grad_image, dtype=torch.float64, with_mean=False):
    """Image and the true gradients w.r.t. (vinv, opacity, l_d[, mean]) for the loss <image, grad_image>."""
    v = vinv.detach().to(dtype).clone().requires_grad_(True)
    o = opacity.detach().to(dtype).clone().requires_grad_(True)
    l = l_d.detach().to(dtype).clone().requires_grad_(True)
    m = mean.detach().to(dtype).clone().requires_grad_(with_mean)
    img = render(start, end, m if with_mean else mean, v, o, l, width, height, dtype)
    (img * grad_image.to(dtype)).sum().backward()
    if with_mean:
        return img.detach(), v.grad, o.grad, l.grad, m.grad
    return img.detach(), v.grad, o.grad, l.grad
