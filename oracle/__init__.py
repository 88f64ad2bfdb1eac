"""oracle/ — CPU restatement of the reference's grouped-scan path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (simplegaussiansplat_tk71_amd, grouped_cumprod.py,
cuda_kernel.py) never does, and has no CPU fallback.

  gcp_oracle.c   sequential fp32 C statement of the three kernels (+ fp64 variants)
  c_oracle.py    ctypes binding of it (torch CPU tensors in, torch CPU tensors out)
  torch_path.py  the "pure-PyTorch torch.cumprod path": per-group torch.cumprod / cumsum
  wrappers.py    literal restatement of gs_model.py:544-566 and :716-722 around the scans
  ref.py         loader for oracle/_ref/ (the reference's own sources compiled here)
  dense_render.py       dense autograd restatement of the rasterise-and-blend Function (rows f1/f2)
  gs_forward_torch.py   op-for-op PyTorch restatement of the model forward up to the Function call (row f4):
                        the checker of the fused projection kernels, pinned to the reference's own forward
  loss_torch.py         PyTorch formulation of the L1 + D-SSIM loss: the checker of the fused loss kernels

Pinned by: the reference's known-answer vectors (cuda_test.py:19-34,
uitility.py:383-393), the reference's forward .cu files compiled for the host
(oracle/_ref/grouped_cumprod_ref_host.so) live and via tests/golden/, and the
reference's whole extension compiled for gfx950
(oracle/_ref/grouped_cumprod_ref_gfx950.so) on the GPU box.
"""
