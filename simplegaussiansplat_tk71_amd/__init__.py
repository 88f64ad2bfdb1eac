"""MI355X-native grouped cumprod / cumsum (per-pixel alpha-compositing scan) of
TaiseiNiman/SimpleGaussianSplat_tk71.

Layout (only what the hot path needs):
  csrc/gcp_scan.hip   hand-written HIP kernels + the C ABI (include/grouped_cumprod_hip.h)
  _build.py, _lib.py  in-tree hipcc build and ctypes binding (no fallback)
  grouped_cumprod.py  drop-in for the reference's compiled module `grouped_cumprod`
  cuda_kernel.py      torch.autograd.Functions + the reference's scan call sites
  synthetic.py        synthetic (H x W, D splats/pixel) pair lists of BASELINE.md
  sharding.py         pixel-group sharding across GPUs (RCCL gather / scatter)
"""
__version__ = "0.1.0"
