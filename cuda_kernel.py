"""`cuda_kernel.py` — the file BASELINE.json's north_star names for the autograd
Function (the reference's file of that name is a stale JIT loader,
cuda_kernel.py:1-19).  Re-exports the package module."""
from simplegaussiansplat_tk71_amd.cuda_kernel import *  # noqa: F401,F403
from simplegaussiansplat_tk71_amd.cuda_kernel import __all__  # noqa: F401
