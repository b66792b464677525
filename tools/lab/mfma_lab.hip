// lab: what rate does v_mfma_f32_32x32x2_f32 sustain (a) from registers, (b) with the GEMM's LDS reads,
// at 1..3 waves per SIMD.  grid = 256*wg_per_cu workgroups of 256 threads.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void lab(int iters, float* out, int lds_pad_kb) {
  extern __shared__ float dyn[];
  __shared__ __attribute__((aligned(16))) float lds[2 * 128 * 36];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 128 * 36; i += 256) lds[i] = (float)(i & 7) * 0.125f;
  if (lds_pad_kb > 0) dyn[tid] = 0.f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const float* As = lds; const float* Bs = lds + 128 * 36;
  const int wm = wave >> 1, wn = wave & 1;
  float4 af[2], bf[2];
  af[0] = af[1] = bf[0] = bf[1] = make_float4(1.f, 0.5f, 0.25f, 0.125f);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      if (MODE >= 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const float4*>(As + (wm * 64 + i * 32 + (lane & 31)) * 36 + kc * 8 + (lane >> 5) * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const float4*>(Bs + (wn * 64 + j * 32 + (lane & 31)) * 36 + kc * 8 + (lane >> 5) * 4);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (MODE >= 2) __syncthreads();
    if (MODE >= 3) {   // the store half of the single-stage scheme (values irrelevant)
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const int idx = tid + v * 256;
        *reinterpret_cast<float4*>(lds + (idx >> 3) * 36 + (idx & 7) * 4) = make_float4(1.f, 0.5f, 0.25f, 0.125f);
      }
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.f) out[tid] = s;
}
extern "C" int lab_run(int mode, int wg_per_cu, int iters, float* out, int lds_pad_kb, void* stream) {
  dim3 g(256 * wg_per_cu), b(256);
  size_t dyn = (size_t)lds_pad_kb * 1024;
  switch (mode) {
    case 0: hipLaunchKernelGGL(lab<0>, g, b, dyn, (hipStream_t)stream, iters, out, lds_pad_kb); break;
    case 1: hipLaunchKernelGGL(lab<1>, g, b, dyn, (hipStream_t)stream, iters, out, lds_pad_kb); break;
    case 2: hipLaunchKernelGGL(lab<2>, g, b, dyn, (hipStream_t)stream, iters, out, lds_pad_kb); break;
    default: hipLaunchKernelGGL(lab<3>, g, b, dyn, (hipStream_t)stream, iters, out, lds_pad_kb); break;
  }
  return (int)hipGetLastError();
}
