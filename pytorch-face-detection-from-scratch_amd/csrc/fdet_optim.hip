// Optimiser step and dropout-scale generation (HBM-bound elementwise kernels, 16 B/lane).
#include "fdet_common.h"
#include <cmath>

using namespace fdet;

// torch.optim._multi_tensor.Adam.step (torch 1.10.1) on one flat buffer:
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
//   p = p - (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// 16 B read + 12 B written per parameter besides the gradient.
__global__ void __launch_bounds__(256)
k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
       size_t n, float b1, float b2, float one_m_b1, float one_m_b2, float step_size, float inv_bc2_sqrt,
       float eps, float gscale) {
  const size_t nv = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* pa = &pp.x; const float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = ga[k] * gscale;
      ma[k] = ma[k] * b1 + gr * one_m_b1;
      va[k] = va[k] * b2 + (gr * gr) * one_m_b2;
      const float denom = sqrtf(va[k]) * inv_bc2_sqrt + eps;
      pa[k] = pa[k] - step_size * (ma[k] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0) {
    for (size_t i = nv * 4 + threadIdx.x; i < n; i += blockDim.x) {
      const float gr = g[i] * gscale;
      const float mk = m[i] * b1 + gr * one_m_b1;
      const float vk = v[i] * b2 + (gr * gr) * one_m_b2;
      m[i] = mk; v[i] = vk;
      p[i] = p[i] - step_size * (mk / (sqrtf(vk) * inv_bc2_sqrt + eps));
    }
  }
}

extern "C" int fdet_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                              int step, double lr, double beta1, double beta2, double eps, float grad_scale,
                              void* stream) {
  FDET_REQUIRE(param && grad && exp_avg && exp_avg_sq && step >= 1, "adam_step: bad arguments (step=%d)", step);
  FDET_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16) == 0,
               "adam_step: buffers must be 16-byte aligned");
  if (n == 0) return FDET_OK;
  const double bc1 = 1.0 - std::pow(beta1, (double)step);
  const double bc2 = 1.0 - std::pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / std::sqrt(bc2));
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, n, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2),
                     step_size, inv_bc2_sqrt, (float)eps, grad_scale);
  return check_launch("fdet_adam_step");
}

// Counter-based generator (splitmix64 finaliser) -> uniform [0,1) with 24 bits.
__device__ __forceinline__ float u01(uint64_t seed, uint64_t ctr) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (ctr + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ void __launch_bounds__(256)
k_dropout_scales(float* __restrict__ out, size_t n, float p, float keep_scale, uint64_t seed, uint64_t offset) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (u01(seed, offset + i) >= p) ? keep_scale : 0.f;
}

extern "C" int fdet_dropout_scales(float* out, size_t n, float p, uint64_t seed, uint64_t offset, void* stream) {
  FDET_REQUIRE(out && p >= 0.f && p < 1.f, "dropout_scales: bad arguments (p=%f)", (double)p);
  if (n == 0) return FDET_OK;
  hipLaunchKernelGGL(k_dropout_scales, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out,
                     n, p, 1.0f / (1.0f - p), seed, offset);
  return check_launch("fdet_dropout_scales");
}

// Dropout2d scales of EVERY dropout layer of a model in one launch, indexed by the GLOBAL image number so that
// data-parallel ranks draw exactly the planes a single process would draw for the concatenated batch
// (per-rank streams; rank r passes first_image = r * n):
//   counter(image g, layer k, channel c) = base + g * LS + pre_k + c,   LS = sum_k C_k,  pre_k = sum_{j<k} C_j
// out: layer k is a dense [n][C_k] array at float offset n * pre_k.
struct DropLayers {
  int nlayers, ls;
  int pre[33];
  float p[32];
};

__global__ void __launch_bounds__(256)
k_dropout_layers(float* __restrict__ out, int n, DropLayers L, uint64_t seed, uint64_t base, uint64_t first_image) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * L.ls) return;
  const int i = (int)(t / L.ls), r = (int)(t - (size_t)i * L.ls);
  int k = 0;
  while (k + 1 < L.nlayers && r >= L.pre[k + 1]) ++k;
  const int c = r - L.pre[k], C = L.pre[k + 1] - L.pre[k];
  const float p = L.p[k];
  const float u = u01(seed, base + (first_image + (uint64_t)i) * (uint64_t)L.ls + (uint64_t)r);
  out[(size_t)n * L.pre[k] + (size_t)i * C + c] = (u >= p) ? 1.0f / (1.0f - p) : 0.f;
}

extern "C" int fdet_dropout_scales_layers(float* out, int n, int nlayers, const int* channels, const float* p,
                                          uint64_t seed, uint64_t base, uint64_t first_image, void* stream) {
  FDET_REQUIRE(out && channels && p && n >= 0 && nlayers >= 1 && nlayers <= 32,
               "dropout_scales_layers: bad arguments (n=%d, nlayers=%d; at most 32 layers)", n, nlayers);
  DropLayers L;
  L.nlayers = nlayers;
  L.pre[0] = 0;
  for (int k = 0; k < nlayers; ++k) {
    FDET_REQUIRE(channels[k] > 0 && p[k] >= 0.f && p[k] < 1.f, "dropout_scales_layers: layer %d: channels=%d p=%f", k,
                 channels[k], (double)p[k]);
    L.pre[k + 1] = L.pre[k] + channels[k];
    L.p[k] = p[k];
  }
  L.ls = L.pre[nlayers];
  if (n == 0) return FDET_OK;
  const size_t total = (size_t)n * L.ls;
  hipLaunchKernelGGL(k_dropout_layers, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out,
                     n, L, seed, base, first_image);
  return check_launch("fdet_dropout_scales_layers");
}
