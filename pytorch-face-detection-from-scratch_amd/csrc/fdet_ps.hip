// Pre-split (PS) activation format: allocation geometry and the fp32 NCHW <-> PS converters (fdet_ps.h).
// The converters sit at the boundary of the PS region of the conv stack (and in the tests); inside the region the
// producing epilogues write PS directly.
#include "fdet_ps.h"

using namespace fdet;

namespace {

__global__ void __launch_bounds__(256)
k_ps_from_f32(const float* __restrict__ x, ps_bf16x8* __restrict__ ps, PsGeo g, int total) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int xx = t % g.W, r = t / g.W;
  const int y = r % g.H, r2 = r / g.H;
  const int gr = r2 % g.C8, n = r2 / g.C8;
  const size_t HW = (size_t)g.H * g.W;
  const float* src = x + ((size_t)n * g.C + gr * 8) * HW + (size_t)y * g.W + xx;
  ps_bf16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float f = src[j * HW];
    const __bf16 h = (__bf16)f;
    hi[j] = h;
    lo[j] = (__bf16)(f - (float)h);
  }
  const size_t u = (size_t)n * g.img + (size_t)(gr * g.HP + y) * g.WP + xx + 1;
  ps[u] = hi;
  ps[u + g.plane] = lo;
}

__global__ void __launch_bounds__(256)
k_ps_to_f32(const ps_bf16x8* __restrict__ ps, float* __restrict__ x, PsGeo g, int total) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int xx = t % g.W, r = t / g.W;
  const int y = r % g.H, r2 = r / g.H;
  const int gr = r2 % g.C8, n = r2 / g.C8;
  const size_t HW = (size_t)g.H * g.W;
  const size_t u = (size_t)n * g.img + (size_t)(gr * g.HP + y) * g.WP + xx + 1;
  const ps_bf16x8 hi = ps[u], lo = ps[u + g.plane];
  float* dst = x + ((size_t)n * g.C + gr * 8) * HW + (size_t)y * g.W + xx;
#pragma unroll
  for (int j = 0; j < 8; ++j) dst[j * HW] = (float)hi[j] + (float)lo[j];
}

// backward of the fused pooled-block tail into a PS tensor: dz2 = unpool(dout) * drop_scale * lrelu'(c) from the pooled
// gradient (fp32 NCHW) and the channel-innermost routing bytes (route8 [N][C/8][Hp][Wp][8]) of the forward pass.
// One thread per (image, channel group, window): 8 channels x 4 positions = four hi and four lo units.
template <bool P16>
__global__ void __launch_bounds__(256)
k_pool_route_bwd_ps(const float* __restrict__ dout, const unsigned char* __restrict__ route8, const float* __restrict__ scale,
                    ps_bf16x8* __restrict__ dz, PsGeo g, int Hp, int Wp, float slope, int total) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int xp = t % Wp, r = t / Wp;
  const int yp = r % Hp, r2 = r / Hp;
  const int gr = r2 % g.C8, n = r2 / g.C8;
  const size_t HWp = (size_t)Hp * Wp;
  const float* src = dout + ((size_t)n * g.C + gr * 8) * HWp + (size_t)yp * Wp + xp;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 rb = *reinterpret_cast<const u32x2*>(route8 + (((size_t)(n * g.C8 + gr) * Hp + yp) * Wp + xp) * 8);
  float gv[8];
  unsigned mk[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    gv[j] = src[j * HWp] * (scale ? scale[n * g.C + gr * 8 + j] : 1.f);
    mk[j] = (rb[j >> 2] >> (8 * (j & 3))) & 0xffu;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ps_bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = ((int)((mk[j] >> 4) & 3) == k) ? gv[j] * (((mk[j] >> k) & 1) ? 1.f : slope) : 0.f;
      const __bf16 h = (__bf16)f;
      hi[j] = h;
      if (!P16) lo[j] = (__bf16)(f - (float)h);
    }
    const size_t u = (size_t)n * g.img + (size_t)(gr * g.HP + 2 * yp + (k >> 1)) * g.WP + 2 * xp + (k & 1) + 1;
    dz[u] = hi;
    if (!P16) dz[u + g.plane] = lo;                        // precision16: the hi plane only
  }
}

}  // namespace

namespace {
int pool_route_bwd_ps_run(const float* dout_pooled, const unsigned char* route8, const float* drop_scale,
                          void* dz2_ps, int N, int C, int H, int W, float slope, void* stream, bool p16) {
  PsGeo g;
  FDET_REQUIRE(dout_pooled && route8 && dz2_ps && !(H & 1) && !(W & 1) && ps_geo(N, C, H, W, g),
               "pool_route_bwd_ps: unsupported shape N=%d C=%d H=%d W=%d (even H, W; C %% 8 == 0; W <= 62)", N, C, H, W);
  const long long total = (long long)N * g.C8 * (H / 2) * (W / 2);
  FDET_REQUIRE(total < (1ll << 31), "pool_route_bwd_ps: tensor too large");
  if (p16)
    hipLaunchKernelGGL(k_pool_route_bwd_ps<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dout_pooled, route8, drop_scale, reinterpret_cast<ps_bf16x8*>(dz2_ps), g, H / 2, W / 2, slope, (int)total);
  else
    hipLaunchKernelGGL(k_pool_route_bwd_ps<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dout_pooled, route8, drop_scale, reinterpret_cast<ps_bf16x8*>(dz2_ps), g, H / 2, W / 2, slope, (int)total);
  return check_launch("fdet_pool_route_bwd_ps");
}
}  // namespace

extern "C" int fdet_pool_route_bwd_ps(const float* dout_pooled, const unsigned char* route8, const float* drop_scale,
                                      void* dz2_ps, int N, int C, int H, int W, float slope, void* stream) {
  return pool_route_bwd_ps_run(dout_pooled, route8, drop_scale, dz2_ps, N, C, H, W, slope, stream, false);
}
extern "C" int fdet_pool_route_bwd_ps_p16(const float* dout_pooled, const unsigned char* route8, const float* drop_scale,
                                          void* dz2_ps, int N, int C, int H, int W, float slope, void* stream) {
  return pool_route_bwd_ps_run(dout_pooled, route8, drop_scale, dz2_ps, N, C, H, W, slope, stream, true);
}

extern "C" size_t fdet_ps_bytes(int N, int C, int H, int W) {
  PsGeo g;
  if (!ps_geo(N, C, H, W, g)) return 0;
  return (size_t)(N + 2) * g.img * 16;
}

extern "C" size_t fdet_ps_image0_offset(int N, int C, int H, int W) {
  PsGeo g;
  if (!ps_geo(N, C, H, W, g)) return 0;
  return (size_t)g.img * 16;
}

extern "C" int fdet_ps_from_f32(const float* x, void* ps, int N, int C, int H, int W, void* stream) {
  PsGeo g;
  FDET_REQUIRE(x && ps && ps_geo(N, C, H, W, g), "ps_from_f32: unsupported shape N=%d C=%d H=%d W=%d", N, C, H, W);
  const long long total = (long long)N * g.C8 * H * W;
  FDET_REQUIRE(total < (1ll << 31), "ps_from_f32: tensor too large");
  hipLaunchKernelGGL(k_ps_from_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     reinterpret_cast<ps_bf16x8*>(ps), g, (int)total);
  return check_launch("fdet_ps_from_f32");
}

extern "C" int fdet_ps_to_f32(const void* ps, float* x, int N, int C, int H, int W, void* stream) {
  PsGeo g;
  FDET_REQUIRE(x && ps && ps_geo(N, C, H, W, g), "ps_to_f32: unsupported shape N=%d C=%d H=%d W=%d", N, C, H, W);
  const long long total = (long long)N * g.C8 * H * W;
  FDET_REQUIRE(total < (1ll << 31), "ps_to_f32: tensor too large");
  hipLaunchKernelGGL(k_ps_to_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const ps_bf16x8*>(ps), x, g, (int)total);
  return check_launch("fdet_ps_to_f32");
}
