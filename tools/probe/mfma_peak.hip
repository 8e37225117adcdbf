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
// MFMA + LDS operand reads in the conv kernel's pattern: per 8 MFMAs, 2 A + 4 B dwords from LDS,
// software-pipelined one group ahead.
__global__ void __launch_bounds__(256, 2) k_peak_lds(float* out, const float* in, int iters) {
  __shared__ float lds[8192];
  for (int t = threadIdx.x; t < 8192; t += 256) lds[t] = in[t & 4095];
  __syncthreads();
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float* A = lds + (lane >> 5) * 576 + (lane & 31);
  const float* B = lds + 4096 + (lane >> 5) * 639 + wid * 128 + (lane & 31);
  float a0 = A[0], a1 = A[32], b0 = B[0], b1 = B[32], b2 = B[64], b3 = B[96];
  for (int it = 0; it < iters; ++it) {
    const int o = (it & 31) * 64, ob = (it & 15) * 3;
    const float na0 = A[o], na1 = A[o + 32], nb0 = B[ob], nb1 = B[ob + 32], nb2 = B[ob + 64], nb3 = B[ob + 96];
    __builtin_amdgcn_sched_barrier(0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b2, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b3, acc[3], 0, 0, 0);
    acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[4], 0, 0, 0);
    acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[5], 0, 0, 0);
    acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b2, acc[6], 0, 0, 0);
    acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b3, acc[7], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; b2 = nb2; b3 = nb3;
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
extern "C" int probe_peak_lds(float* out, const float* in, int blocks, int iters, void* stream) {
  hipLaunchKernelGGL(k_peak_lds, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, in, iters);
  return (int)hipGetLastError();
}
extern "C" int probe_peak(float* out, int blocks, int iters, void* stream) {
  hipLaunchKernelGGL(k_peak, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 1.0001f, 0.9999f);
  return (int)hipGetLastError();
}
