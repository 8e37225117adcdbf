// bf16 MFMA shape probe (not product code): the same FLOPs issued as v_mfma_f32_32x32x16_bf16 or as v_mfma_f32_16x16x32_bf16 on
// random operands, operands in registers or re-read from LDS, one wave per SIMD (256 threads per CU, one workgroup per CU).
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probe/mfma_shapes.hip -o tools/probe/libmfmashapes.so
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, bool LDS>
__global__ void __launch_bounds__(256, 1) k_shape(float* out, const bf16x8* in, int iters) {
  __shared__ bf16x8 tile[2048];                                  // 32 KB of operand chunks
  for (int t = threadIdx.x; t < 2048; t += 256) tile[t] = in[t];
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  bf16x8 a[2], b[2];
  for (int i = 0; i < 2; ++i) { a[i] = tile[(wid * 64 + lane + 17 * i) & 2047]; b[i] = tile[(wid * 64 + lane + 901 + 29 * i) & 2047]; }
  float s = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[4];                                               // a 64x64 output block = 4 tiles of 32x32
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if (LDS) for (int i = 0; i < 2; ++i) { a[i] = tile[(it * 64 + lane + 256 * i) & 2047]; b[i] = tile[(it * 64 + lane + 1024 + 256 * i) & 2047]; }
      // K = 32 per iteration: two k-steps of 16
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[3], 0, 0, 0);
      }
    }
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x4 acc[16];                                               // the same 64x64 block = 16 tiles of 16x16
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    bf16x8 a4[4], b4[4];
    for (int i = 0; i < 4; ++i) { a4[i] = tile[(wid * 64 + lane + 17 * i) & 2047]; b4[i] = tile[(wid * 64 + lane + 901 + 29 * i) & 2047]; }
    for (int it = 0; it < iters; ++it) {
      if (LDS) for (int i = 0; i < 4; ++i) { a4[i] = tile[(it * 64 + lane + 256 * i) & 2047]; b4[i] = tile[(it * 64 + lane + 1024 + 256 * i) & 2047]; }
      // K = 32 per iteration: one k-step of 32
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a4[i], b4[j], acc[i * 4 + j], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" int probe_shape(float* out, const void* in, int blocks, int iters, int shape, int lds, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const bf16x8* p = (const bf16x8*)in;
  if (shape == 32 && !lds) hipLaunchKernelGGL((k_shape<32, false>), dim3(blocks), dim3(256), 0, st, out, p, iters);
  else if (shape == 32) hipLaunchKernelGGL((k_shape<32, true>), dim3(blocks), dim3(256), 0, st, out, p, iters);
  else if (!lds) hipLaunchKernelGGL((k_shape<16, false>), dim3(blocks), dim3(256), 0, st, out, p, iters);
  else hipLaunchKernelGGL((k_shape<16, true>), dim3(blocks), dim3(256), 0, st, out, p, iters);
  return (int)hipGetLastError();
}
