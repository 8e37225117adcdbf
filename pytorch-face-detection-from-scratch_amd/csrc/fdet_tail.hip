// Elementwise tails of the residual block (HBM-bound): dropout scale + skip add + 2x2 max
// pool forward, and their backward fused with the second LeakyReLU's derivative.
// models/PoolResnet.py:37-42 (models/Resnet.py:34-39) and ATen's max_pool2d backward
// (first maximum in window scan order wins; NaN is a maximum).
#include "fdet_common.h"

using namespace fdet;

namespace {

__device__ __forceinline__ void upd(float v, int k, float& m, int& arg) {
  if (v > m || v != v) { m = v; arg = k; }
}

// pool == 2: one thread per pooled output (floor(H/2) x floor(W/2): an odd last row / column belongs to
// no window, as nn.MaxPool2d(2)); pool == 1: one thread per element
__global__ void __launch_bounds__(256)
k_tail_fwd(const float* __restrict__ c, const float* __restrict__ x, const float* __restrict__ scale,
           float* __restrict__ out, int NF, int H, int W, int pool) {
  const int Ho = H / pool, Wo = W / pool;
  const size_t total = (size_t)NF * Ho * Wo;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int ox = (int)(t % Wo);
    const size_t r = t / Wo;
    const int oy = (int)(r % Ho);
    const size_t nf = r / Ho;
    const float sc = scale ? scale[nf] : 1.f;
    if (pool == 1) {
      out[t] = c[t] * sc + x[t];
    } else {
      const size_t base = (nf * H + (size_t)oy * 2) * W + (size_t)ox * 2;
      float m = -INFINITY; int arg = 0;
      upd(c[base] * sc + x[base], 0, m, arg);
      upd(c[base + 1] * sc + x[base + 1], 1, m, arg);
      upd(c[base + W] * sc + x[base + W], 2, m, arg);
      upd(c[base + W + 1] * sc + x[base + W + 1], 3, m, arg);
      out[t] = m;
    }
  }
}

__global__ void __launch_bounds__(256)
k_tail_bwd(const float* __restrict__ dout, const float* __restrict__ c, const float* __restrict__ x,
           const float* __restrict__ scale, float* __restrict__ dz2, float* __restrict__ de, int NF, int H,
           int W, int pool, float slope) {
  const int Ho = H / pool, Wo = W / pool;
  const size_t total = (size_t)NF * Ho * Wo;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int ox = (int)(t % Wo);
    const size_t r = t / Wo;
    const int oy = (int)(r % Ho);
    const size_t nf = r / Ho;
    const float sc = scale ? scale[nf] : 1.f;
    const float g = dout[t];
    if (pool == 1) {
      const float cv = c[t];
      dz2[t] = g * sc * (cv > 0.f ? 1.f : slope);
    } else {
      const size_t base = (nf * H + (size_t)oy * 2) * W + (size_t)ox * 2;
      const size_t off[4] = {0, 1, (size_t)W, (size_t)W + 1};
      float cv[4];
      float m = -INFINITY; int arg = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) { cv[k] = c[base + off[k]]; upd(cv[k] * sc + x[base + off[k]], k, m, arg); }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = (k == arg) ? g : 0.f;
        de[base + off[k]] = d;
        dz2[base + off[k]] = d * sc * (cv[k] > 0.f ? 1.f : slope);
      }
    }
  }
}

}  // namespace

extern "C" int fdet_block_tail_fwd(const float* c, const float* x, const float* drop_scale, float* out, int N,
                                   int F, int H, int W, int pool, void* stream) {
  FDET_REQUIRE(c && x && out && N > 0 && F > 0 && H > 0 && W > 0, "block_tail_fwd: bad arguments");
  FDET_REQUIRE(pool == 1 || (pool == 2 && H >= 2 && W >= 2), "block_tail_fwd: pool=%d needs H,W >= 2 (H=%d W=%d)", pool, H, W);
  const size_t total = (size_t)N * F * (H / pool) * (W / pool);
  size_t blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_tail_fwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, c, x, drop_scale, out,
                     N * F, H, W, pool);
  return check_launch("fdet_block_tail_fwd");
}

extern "C" int fdet_block_tail_bwd(const float* dout, const float* c, const float* x, const float* drop_scale,
                                   float* dz2, float* de, int N, int F, int H, int W, int pool, float slope,
                                   void* stream) {
  FDET_REQUIRE(dout && c && dz2 && N > 0 && F > 0 && H > 0 && W > 0, "block_tail_bwd: bad arguments");
  FDET_REQUIRE(pool == 1 || (pool == 2 && H >= 2 && W >= 2 && x && de),
               "block_tail_bwd: pool=%d needs H,W >= 2 and x,de buffers", pool);
  if (pool == 2 && ((H | W) & 1)) {      // odd map: the last row / column is in no window -> zero gradient
    (void)hipMemsetAsync(dz2, 0, (size_t)N * F * H * W * sizeof(float), (hipStream_t)stream);
    (void)hipMemsetAsync(de, 0, (size_t)N * F * H * W * sizeof(float), (hipStream_t)stream);
  }
  const size_t total = (size_t)N * F * (H / pool) * (W / pool);
  size_t blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_tail_bwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dout, c, x, drop_scale,
                     dz2, de, N * F, H, W, pool, slope);
  return check_launch("fdet_block_tail_bwd");
}
