// bare v_mfma_f32_32x32x16_bf16 loop: what rate does this box hold with every CU issuing MFMAs only?
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  uint4 x = make_uint4(0x3f803f80u + threadIdx.x * seed, 0x3f003f80u, 0x3e803f80u ^ seed, 0x3f803c80u);
  uint4 y = make_uint4(0x3f803f80u, 0x3f003f00u + seed, 0x3e803f80u, 0x3f803e80u);
  bf16x8 a = __builtin_bit_cast(bf16x8, x), b = __builtin_bit_cast(bf16x8, y);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[q], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  if (s == 12345.f) out[0] = s;
}
// the same loop on random operands (data toggling costs power: does the clock hold?)
__global__ __launch_bounds__(256) void krand(float* out, int iters, const uint4* rnd) {
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 a[4], b[4];
  for (int q = 0; q < 4; ++q) {
    a[q] = __builtin_bit_cast(bf16x8, rnd[(threadIdx.x * 8 + q) & 4095]);
    b[q] = __builtin_bit_cast(bf16x8, rnd[(threadIdx.x * 8 + 4 + q + blockIdx.x) & 4095]);
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], b[q], acc[q], 0, 0, 0);
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[(q + 1) & 3], a[(q + 2) & 3], acc[q], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  if (s == 12345.f) out[0] = s;
}
extern "C" int lab_bf16_rand(int wg_per_cu, int iters, float* out, const void* rnd, void* st) {
  hipLaunchKernelGGL(krand, dim3(256 * wg_per_cu), dim3(256), 0, (hipStream_t)st, out, iters, (const uint4*)rnd);
  return (int)hipGetLastError();
}
extern "C" int lab_bf16(int wg_per_cu, int iters, float* out, void* st) {
  hipLaunchKernelGGL(k<4>, dim3(256 * wg_per_cu), dim3(256), 0, (hipStream_t)st, out, iters, 1u);
  return (int)hipGetLastError();
}
