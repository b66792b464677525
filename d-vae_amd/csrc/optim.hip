// Adam step over a flat fp32 buffer: the optimizer of the reference's train step (dvae_model.py:471-475,
// torch.optim.Adam(lr=1e-4)) as one streaming pass -- p, g, m, v read once, p, m, v written once (28 B per
// parameter).  Same update as torch's (non-amsgrad, L2 weight decay folded into the gradient):
//   g' = g + wd p;  m = m + (1 - b1)(g' - m);  v = b2 v + (1 - b2) g'^2
//   p = p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include <algorithm>
#include <cmath>

#include "common.h"

namespace {

struct AdamK { float lr_over_bc1, inv_bc2_sqrt, one_minus_b1, b2, one_minus_b2, eps, wd; };   // 1 - beta formed in double, as torch does

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamK& k) {
  g = fmaf(k.wd, p, g);
  m = fmaf(k.one_minus_b1, g - m, m);
  v = fmaf(k.one_minus_b2, g * g, k.b2 * v);
  const float denom = sqrtf(v) * k.inv_bc2_sqrt + k.eps;
  p -= k.lr_over_bc1 * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ P, const float* __restrict__ Gr, float* __restrict__ M,
                                                   float* __restrict__ V, size_t n, AdamK k) {
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 p = reinterpret_cast<float4*>(P)[i], m = reinterpret_cast<float4*>(M)[i], v = reinterpret_cast<float4*>(V)[i];
    const float4 g = reinterpret_cast<const float4*>(Gr)[i];
    adam_one(p.x, g.x, m.x, v.x, k);
    adam_one(p.y, g.y, m.y, v.y, k);
    adam_one(p.z, g.z, m.z, v.z, k);
    adam_one(p.w, g.w, m.w, v.w, k);
    reinterpret_cast<float4*>(P)[i] = p;
    reinterpret_cast<float4*>(M)[i] = m;
    reinterpret_cast<float4*>(V)[i] = v;
  }
  // the last n % 4 parameters
  const size_t t = (n4 << 2) + blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && t < n) adam_one(P[t], Gr[t], M[t], V[t], k);
}

}  // namespace

extern "C" int dvae_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                              double beta1, double beta2, double eps, double weight_decay, long long step, void* stream) {
  DVAE_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adam_step: null pointer");
  DVAE_REQUIRE(step >= 1 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0, "adam_step: step >= 1, betas in [0, 1)");
  DVAE_REQUIRE(((uintptr_t)param & 15) == 0 && ((uintptr_t)grad & 15) == 0 && ((uintptr_t)exp_avg & 15) == 0 &&
                   ((uintptr_t)exp_avg_sq & 15) == 0, "adam_step: buffers must be 16-byte aligned");
  if (n == 0) return DVAE_OK;
  const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
  const AdamK k{(float)(lr / bc1), (float)(1.0 / std::sqrt(bc2)), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                (float)eps, (float)weight_decay};
  const int grid = (int)std::min<size_t>(((n >> 2) + 256) / 256, 4096);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, k);
  DVAE_LAUNCH_CHECK();
  return DVAE_OK;
}
