// Boundary probe: C-ABI .so launched on torch's stream via ctypes. Not product code.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k_mfma(float* o, const float* A, const float* B){
  // A: 32x2 row-major [i][k], B: 2x32 row-major [k][j]; out 32x32 row-major
  int l = threadIdx.x;
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l&31)*2 + (l>>5)], B[(l>>5)*32 + (l&31)], acc, 0,0,0);
  for(int r=0;r<16;r++){ int row=(r&3)+8*(r>>2)+4*(l>>5); int col=l&31; o[row*32+col]=acc[r]; }
}
__global__ void k_copy(float4* __restrict__ o, const float4* __restrict__ a, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x; size_t s=(size_t)gridDim.x*blockDim.x;
  for(;i<n;i+=s) o[i]=a[i];
}
extern "C" int probe_mfma(float* o,const float* a,const float* b, void* stream){
  hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, (hipStream_t)stream, o,a,b);
  return (int)hipGetLastError();
}
extern "C" int probe_copy(void* o,const void* a,size_t nbytes, void* stream){
  hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, (hipStream_t)stream, (float4*)o,(const float4*)a,nbytes/16);
  return (int)hipGetLastError();
}
