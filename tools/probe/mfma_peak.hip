// fp32 MFMA issue-rate probe (not product code): waves x iters x 8 independent 32x32x2 MFMAs
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256, 2) k_peak(float* out, int iters, float a, float b) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = threadIdx.x * 1e-9f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256, 2) k_peak_rand(float* out, const float* in, int iters) {
  f32x16 acc[8];
  const float a0 = in[threadIdx.x], b0 = in[256 + threadIdx.x];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = in[(threadIdx.x * 16 + r + i * 7) & 4095];
  float a = a0, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    a = -a; b = b * -1.0001f;
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
extern "C" int probe_peak_rand(float* out, const float* in, int blocks, int iters, void* stream) {
  hipLaunchKernelGGL(k_peak_rand, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, in, iters);
  return (int)hipGetLastError();
}
extern "C" int probe_peak(float* out, int blocks, int iters, void* stream) {
  hipLaunchKernelGGL(k_peak, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 1.0001f, 0.9999f);
  return (int)hipGetLastError();
}
