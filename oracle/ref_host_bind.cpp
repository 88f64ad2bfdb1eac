// ref_host_bind.cpp — pybind11 registration for the HOST build of the reference's two
// forward scans (oracle/_ref/grouped_cumprod_ref_host*.so).
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  This file is ours; the two
// functions it registers are compiled straight from the reference's sources
// where they lie (/root/reference/cuda_kernel/grouped_cumprod_forward.cu and
// grouped_cumsum_forward.cu) with rocThrust's sequential CPP backend selected,
// so thrust::inclusive_scan_by_key runs on CPU tensors.  The reference's own
// registration file (cuda_kernel/cuda_kernel.cpp:17-22) also declares the
// backward, which is a __global__ kernel and has no host form — hence this
// two-function registration for the host module.  Same names, same signatures.
#include <torch/extension.h>

void grouped_cumprod_forward(torch::Tensor x, torch::Tensor key, torch::Tensor y);
void grouped_cumsum_forward(torch::Tensor x, torch::Tensor key, torch::Tensor y);

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("grouped_cumprod_forward", &grouped_cumprod_forward,
        "reference grouped_cumprod_forward.cu, Thrust CPP (host) backend");
  m.def("grouped_cumsum_forward", &grouped_cumsum_forward,
        "reference grouped_cumsum_forward.cu, Thrust CPP (host) backend");
}
