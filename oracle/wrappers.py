"""Literal CPU restatement of the two scan call sites of the reference's autograd Function.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  create_alpha_brend  <- gs_model.py:544-566  (_create_alpha_brend)
  grad_cumsum         <- gs_model.py:716-722
  unique              <- gs_model.py:538-541
  mask_zero_T         <- gs_model.py:575-578
  create_alpha_brend_min / cat_alpha_brend / create_grad_alphabrend_min  <- gs_model.py:582-594, :724-730 (below)

Statement-for-statement, on CPU tensors, with the C oracle in place of the compiled
extension.  One thing is made explicit: the sort is STABLE.  The reference calls
torch.sort(inv) (gs_model.py:547) whose 1-D CUDA radix sort is stable in practice; the
CPU default is not stable for small inputs, and depth order inside a pixel is carried only
by that stability.
"""
import torch

from . import c_oracle as co


def unique(rects):
    rects = rects.to(torch.int32)
    return rects[:, 1] * 10000 + rects[:, 0]


def mask_zero_T(T):
    mask = T != 0
    return [T[mask], mask]


def create_alpha_brend(rects, anti_opacity, flag, cutting_number=None):
    inv = unique(rects)
    sorted_inv, index = torch.sort(inv, stable=True)
    sorted_anti_opacity = anti_opacity[index].contiguous()
    key = sorted_inv.to(torch.int32).contiguous()
    if flag == "cumprod":
        output = co.cumprod_forward(sorted_anti_opacity, key)
    elif flag == "cumsum":
        output = co.cumsum_forward(sorted_anti_opacity, key)
    else:
        raise ValueError(flag)
    # gs_model.py:555 — back to the original order
    output = output[torch.argsort(index, stable=True)]
    if cutting_number:
        output = output[cutting_number:]
        anti_opacity = anti_opacity[cutting_number:]
    output, mask = mask_zero_T(output)
    if flag == "cumprod":
        output = output / anti_opacity[mask]
    else:
        output = output - anti_opacity[mask]
    return [output, mask, sorted_inv, index]


def grad_cumsum(rects, grad, cutting_number=None):
    """gs_model.py:716-722, literally: flip, cumsum wrapper, flip the VALUES back.
    NB the reference returns the mask in FLIPPED order (it is not flipped back, :721-722)."""
    rects = rects.flip(0)
    grad = grad.flip(0)
    output, mask, _, _ = create_alpha_brend(rects, grad.contiguous(), "cumsum", cutting_number)
    output = output.flip(0)
    return [output, mask]


# ---- the per-pixel carry of the chunked calls ----------------------------------------------------
#   create_alpha_brend_min      <- gs_model.py:582-586  (_create_alpha_brend_min)
#   cat_alpha_brend             <- gs_model.py:589-594  (_cat_alpha_brend)
#   create_grad_alphabrend_min  <- gs_model.py:724-730
# pinned to the reference's own outputs by tests/test_oracle.py (tests/golden/carry_golden.npz)
def create_alpha_brend_min(rects, T):
    unique_rects, inv = torch.unique(rects, return_inverse=True, dim=0)
    T_min = torch.zeros_like(unique_rects[:, 0], dtype=torch.float32).scatter_reduce(0, inv, T, reduce="amin", include_self=False)
    return [unique_rects, T_min]


def cat_alpha_brend(values, rects):
    return [torch.cat((values[0], values[1]), dim=0), torch.cat((rects[0], rects[1]), dim=0)]


def create_grad_alphabrend_min(rects, grad):
    index = torch.arange(rects.size(0), dtype=torch.int32)
    unique_rects, first = create_alpha_brend_min(rects, index.to(torch.float32))  # the index travels as a float (:728)
    return [unique_rects, grad[first.to(torch.int32)]]
