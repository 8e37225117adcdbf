// Pre-split (PS) activation format: allocation geometry and the fp32 NCHW <-> PS converters (fdet_ps.h).
// The converters sit at the boundary of the PS region of the conv stack (and in the tests); inside the region the
// producing epilogues write PS directly.
#include "fdet_ps.h"

using namespace fdet;

namespace {

// (strips: t runs over the columns of the FULL rows; an element next to a strip edge is also its neighbour's halo)
__global__ void __launch_bounds__(256)
k_ps_from_f32(const float* __restrict__ x, ps_bf16x8* __restrict__ ps, PsGeo g, PsStrips st, int total) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int xx = t % st.Wf, r = t / st.Wf;
  const int y = r % g.H, r2 = r / g.H;
  const int gr = r2 % g.C8, n = r2 / g.C8;
  const size_t HW = (size_t)g.H * st.Wf;
  const float* src = x + ((size_t)n * g.C + gr * 8) * HW + (size_t)y * st.Wf + xx;
  ps_bf16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float f = src[j * HW];
    const __bf16 h = (__bf16)f;
    hi[j] = h;
    lo[j] = (__bf16)(f - (float)h);
  }
  const int sidx = xx / st.Ws, xs = xx - sidx * st.Ws;
  const size_t u = (size_t)(sidx * st.Nimg + n) * g.img + (size_t)(gr * g.HP + y) * g.WP + xs + 1;
  ps[u] = hi;
  ps[u + g.plane] = lo;
  if (xs == 0 && sidx > 0) {                                 // right halo (slot Ws + 1) of the strip to the left
    const size_t v = u - (size_t)st.Nimg * g.img + st.Ws;
    ps[v] = hi;
    ps[v + g.plane] = lo;
  }
  if (xs == st.Ws - 1 && sidx + 1 < st.S) {                  // left halo (slot 0) of the strip to the right
    const size_t v = u + (size_t)st.Nimg * g.img - st.Ws;
    ps[v] = hi;
    ps[v + g.plane] = lo;
  }
}

// halos of a strip tensor whose producer wrote real elements only: mode 0 = copy the neighbour strips' edge columns into
// slot 0 / slot Ws + 1 (what a 3x3 conv over the tensor needs), mode 1 = zero them again (what the weight gradient needs of its
// dz operand: a halo slot is not a position of this strip).  One thread per (strip edge, image, plane, group, row).
__global__ void __launch_bounds__(256)
k_ps_halo(ps_bf16x8* __restrict__ ps, PsGeo g, PsStrips st, int planes, int mode, int total) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int y = t % g.H;
  int r = t / g.H;
  const int gr = r % g.C8; r /= g.C8;
  const int pl = r % planes; r /= planes;
  const int n = r % st.Nimg, e = r / st.Nimg;               // edge e lies between strips e and e + 1
  const size_t row = (size_t)pl * g.plane + (size_t)(gr * g.HP + y) * g.WP;
  const size_t left = (size_t)(e * st.Nimg + n) * g.img + row, right = left + (size_t)st.Nimg * g.img;
  if (mode == 0) {
    ps[left + st.Ws + 1] = ps[right + 1];
    ps[right] = ps[left + st.Ws];
  } else {
    const ps_bf16x8 z = {};
    ps[left + st.Ws + 1] = z;
    ps[right] = z;
  }
}

__global__ void __launch_bounds__(256)
k_ps_to_f32(const ps_bf16x8* __restrict__ ps, float* __restrict__ x, PsGeo g, PsStrips st, int total) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int xx = t % st.Wf, r = t / st.Wf;
  const int y = r % g.H, r2 = r / g.H;
  const int gr = r2 % g.C8, n = r2 / g.C8;
  const size_t HW = (size_t)g.H * st.Wf;
  const int sidx = xx / st.Ws, xs = xx - sidx * st.Ws;
  const size_t u = (size_t)(sidx * st.Nimg + n) * g.img + (size_t)(gr * g.HP + y) * g.WP + xs + 1;
  const ps_bf16x8 hi = ps[u], lo = ps[u + g.plane];
  float* dst = x + ((size_t)n * g.C + gr * 8) * HW + (size_t)y * st.Wf + xx;
#pragma unroll
  for (int j = 0; j < 8; ++j) dst[j * HW] = (float)hi[j] + (float)lo[j];
}

// backward of the fused pooled-block tail into a PS tensor: dz2 = unpool(dout) * drop_scale * lrelu'(c) from the pooled
// gradient (fp32 NCHW) and the channel-innermost routing bytes (route8 [N][C/8][Hp][Wp][8]) of the forward pass.
// One thread per (image, channel group, window): 8 channels x 4 positions = four hi and four lo units.
template <bool P16>
__global__ void __launch_bounds__(256)
k_pool_route_bwd_ps(const float* __restrict__ dout, const unsigned char* __restrict__ route8, const float* __restrict__ scale,
                    ps_bf16x8* __restrict__ dz, PsGeo g, PsStrips st, int Hp, int Wp, float slope, int total) {
  // Wp: pooled columns of a FULL row (dout is fp32 NCHW of the whole image); the routing bytes and dz are per strip
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int xg = t % Wp, r = t / Wp;
  const int yp = r % Hp, r2 = r / Hp;
  const int gr = r2 % g.C8, n = r2 / g.C8;
  const size_t HWp = (size_t)Hp * Wp;
  const float* src = dout + ((size_t)n * g.C + gr * 8) * HWp + (size_t)yp * Wp + xg;
  const int wsp = st.Ws >> 1, sidx = xg / wsp, xp = xg - sidx * wsp, ns = sidx * st.Nimg + n;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 rb = *reinterpret_cast<const u32x2*>(route8 + (((size_t)(ns * g.C8 + gr) * Hp + yp) * (g.W >> 1) + xp) * 8);
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
    const size_t u = (size_t)ns * g.img + (size_t)(gr * g.HP + 2 * yp + (k >> 1)) * g.WP + 2 * xp + (k & 1) + 1;
    dz[u] = hi;
    if (!P16) dz[u + g.plane] = lo;                        // precision16: the hi plane only
  }
  // strips: this tensor is the dz operand of the weight gradient next, whose halo slots must be zero (fdet_ps.h) -- the edge
  // windows clear them here, which saves the separate fdet_ps_halo_exchange(zero_only) pass over a recycled buffer
  if (st.S > 1) {
    const ps_bf16x8 zero = {};
    const bool left = xp == 0 && sidx > 0, right = xp == wsp - 1 && sidx + 1 < st.S;
    if (left || right) {
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const size_t row = (size_t)ns * g.img + (size_t)(gr * g.HP + 2 * yp + rr) * g.WP;
        if (left) { dz[row] = zero; if (!P16) dz[row + g.plane] = zero; }
        if (right) { dz[row + st.Ws + 1] = zero; if (!P16) dz[row + st.Ws + 1 + g.plane] = zero; }
      }
    }
  }
}

}  // namespace

namespace {
int pool_route_bwd_ps_run(const float* dout_pooled, const unsigned char* route8, const float* drop_scale,
                          void* dz2_ps, int N, int C, int H, int W, float slope, void* stream, bool p16) {
  PsGeo g;
  PsStrips sp;
  FDET_REQUIRE(dout_pooled && route8 && dz2_ps && !(H & 1) && !(W & 1) && ps_geo_strips(N, C, H, W, g, sp),
               "pool_route_bwd_ps: unsupported shape N=%d C=%d H=%d W=%d (even H, W; C %% 8 == 0)", N, C, H, W);
  const long long total = (long long)N * g.C8 * (H / 2) * (W / 2);
  FDET_REQUIRE(total < (1ll << 31), "pool_route_bwd_ps: tensor too large");
  if (p16)
    hipLaunchKernelGGL(k_pool_route_bwd_ps<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dout_pooled, route8, drop_scale, reinterpret_cast<ps_bf16x8*>(dz2_ps), g, sp, H / 2, W / 2, slope, (int)total);
  else
    hipLaunchKernelGGL(k_pool_route_bwd_ps<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dout_pooled, route8, drop_scale, reinterpret_cast<ps_bf16x8*>(dz2_ps), g, sp, H / 2, W / 2, slope, (int)total);
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

// (maps wider than 63 columns: column strips, fdet_ps.h -- every entry point below takes the FULL width)
extern "C" size_t fdet_ps_bytes(int N, int C, int H, int W) {
  PsGeo g;
  PsStrips sp;
  if (!ps_geo_strips(N, C, H, W, g, sp)) return 0;
  return (size_t)(g.N + 2) * g.img * 16;
}

extern "C" size_t fdet_ps_image0_offset(int N, int C, int H, int W) {
  PsGeo g;
  PsStrips sp;
  if (!ps_geo_strips(N, C, H, W, g, sp)) return 0;
  return (size_t)g.img * 16;
}

// number of column strips of a map of width W (1: the plain layout; 0: no layout)
extern "C" int fdet_ps_strips(int W) {
  PsGeo g;
  PsStrips sp;
  return ps_geo_strips(1, 8, 2, W, g, sp) ? sp.S : 0;
}

// strip tensors only (a no-op otherwise): zero_only = 0 copies the neighbour strips' edge columns into the halo slots (after
// a producer that writes real elements only, before a 3x3 conv reads the tensor); zero_only = 1 clears them (before the
// tensor is the dz operand of fdet_conv3x3_wgrad_ps_batched).  hi_only: precision16 tensors (the lo plane is not touched).
extern "C" int fdet_ps_halo_exchange(void* ps, int N, int C, int H, int W, int zero_only, int hi_only, void* stream) {
  PsGeo g;
  PsStrips sp;
  FDET_REQUIRE(ps && ps_geo_strips(N, C, H, W, g, sp), "ps_halo_exchange: unsupported shape N=%d C=%d H=%d W=%d", N, C, H, W);
  if (sp.S == 1) return FDET_OK;
  const int planes = hi_only ? 1 : 2;
  const long long total = (long long)(sp.S - 1) * N * planes * g.C8 * H;
  FDET_REQUIRE(total < (1ll << 31), "ps_halo_exchange: tensor too large");
  hipLaunchKernelGGL(k_ps_halo, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<ps_bf16x8*>(ps), g, sp, planes, zero_only ? 1 : 0, (int)total);
  return check_launch("fdet_ps_halo_exchange");
}

extern "C" int fdet_ps_from_f32(const float* x, void* ps, int N, int C, int H, int W, void* stream) {
  PsGeo g;
  PsStrips sp;
  FDET_REQUIRE(x && ps && ps_geo_strips(N, C, H, W, g, sp), "ps_from_f32: unsupported shape N=%d C=%d H=%d W=%d", N, C, H, W);
  const long long total = (long long)N * g.C8 * H * W;
  FDET_REQUIRE(total < (1ll << 31), "ps_from_f32: tensor too large");
  hipLaunchKernelGGL(k_ps_from_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     reinterpret_cast<ps_bf16x8*>(ps), g, sp, (int)total);
  return check_launch("fdet_ps_from_f32");
}

extern "C" int fdet_ps_to_f32(const void* ps, float* x, int N, int C, int H, int W, void* stream) {
  PsGeo g;
  PsStrips sp;
  FDET_REQUIRE(x && ps && ps_geo_strips(N, C, H, W, g, sp), "ps_to_f32: unsupported shape N=%d C=%d H=%d W=%d", N, C, H, W);
  const long long total = (long long)N * g.C8 * H * W;
  FDET_REQUIRE(total < (1ll << 31), "ps_to_f32: tensor too large");
  hipLaunchKernelGGL(k_ps_to_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const ps_bf16x8*>(ps), x, g, sp, (int)total);
  return check_launch("fdet_ps_to_f32");
}
