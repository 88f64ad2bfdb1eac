"""`import grouped_cumprod` — the module name the reference imports (gs_model.py:8,
cuda_test.py:6).  Re-exports the HIP-backed drop-in."""
from simplegaussiansplat_tk71_amd.grouped_cumprod import *  # noqa: F401,F403
from simplegaussiansplat_tk71_amd.grouped_cumprod import __all__  # noqa: F401
