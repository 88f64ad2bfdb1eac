"""PyTorch formulation of the training loss (reference: gs_control.py:180-182:
(1 - lambda) * l1_loss + lambda * (1 - kornia.metrics.ssim(..., max_val=1.0, window_size=11).mean())) — the checker of
the fused loss kernels (csrc/gcp_loss.hip).  kornia is not installed here, so `ssim` is the published formula (Wang
et al. 2004) with kornia's conventions (Gaussian window sigma 1.5, reflect padding): parity unpinned.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the product never imports this.
"""
import torch

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


def splat_loss(images, targets, lamda=0.2):
    """(1 - lambda) L1 + lambda (1 - mean SSIM) (reference: gs_control.py:180-182)."""
    l1 = torch.nn.functional.l1_loss(images, targets, reduction="mean")
    return (1 - lamda) * l1 + lamda * (1 - ssim(images, targets, max_val=1.0, window_size=11).mean())
