// diagnostic: one wave spins for `cycles` core clocks and reports (core clocks, 100 MHz ticks)
#include <hip/hip_runtime.h>
__global__ void clock_probe(long long cycles, long long* out) {
  const long long c0 = clock64(), r0 = wall_clock64();
  long long c = c0;
  while (c - c0 < cycles) { __builtin_amdgcn_s_sleep(8); c = clock64(); }
  const long long r1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c - c0; out[1] = r1 - r0; }
}
extern "C" int clock_probe_launch(long long cycles, long long* out, void* stream) {
  hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, cycles, out);
  return (int)hipGetLastError();
}
