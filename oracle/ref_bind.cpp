// Test infrastructure (oracle) -- NOT product code.
// pybind11 entry for the reference's *CPU* DepthCov ops, compiled from the
// reference sources where they lie (/root/reference/como/backend/src/cov_cpu.cpp).
// The reference's own dispatcher (src/cov.cpp) cannot be compiled under a ROCm
// torch (it includes <c10/cuda/CUDAGuard.h> unconditionally, cov.cpp:2), so this
// file binds the two CPU entry points declared in include/cov.h:8,12-14 directly.
#include <torch/extension.h>

torch::Tensor cross_covariance_cpu(torch::Tensor x1, torch::Tensor E1, torch::Tensor x2,
                                   torch::Tensor E2, float scale);
void get_new_chol_obs_info_cpu(torch::Tensor L, torch::Tensor obs_info, torch::Tensor var,
                               torch::Tensor k_ni, torch::Tensor k_id, float k_ii, int N);

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("cross_covariance", &cross_covariance_cpu, "reference CPU cross covariance");
  m.def("get_new_chol_obs_info", &get_new_chol_obs_info_cpu, "reference CPU chol/obs_info append");
}
