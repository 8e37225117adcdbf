// Engine-private "pre-split" (PS) activation format of the bf16x3 conv stack.
//
// A feature map [N,C,H,W] (C % 8 == 0) is kept as bf16 hi | lo planes with 8 channels innermost, the layout the
// MFMA operands have in LDS, so that a consumer stages it with LDS-DMA (no split arithmetic, no LDS stores):
//
//   unit (16 bytes = 8 bf16) index of image n, plane pl (0 = hi, 1 = lo), channel group g = c / 8, row y, slot s
//       n * img + pl * plane + (g * HP + y) * WP + s          plane = (C/8) * HP * WP,  img = 2 * plane
//
//   * slot s = x + 1: slot 0 and slots W+1 .. WP-1 of every row are ZERO; WP = 16 / 32 / 64 >= W + 1.  Slot 0 is the
//     left halo of its row AND (read as "slot WP" of the row above) the right halo of that one, so W = WP - 1 is fine;
//   * rows y = H .. HP-1 are ZERO (HP = H + 1 rounded up to even): one row is both the halo under an image and the
//     halo above the next one, so a band of "virtual rows" v = n * HP + y may run across images, and a row pair
//     (v, v+1) with v even is a row pair (y, y+1) with y even of one image;
//   * the buffer holds one all-zero guard image in front of image 0 and one behind image N-1 (virtual rows -1 and
//     N*HP .. of the first / last band); producers write real elements only, the zeros are written once at
//     allocation (fdet_ps_bytes / fdet_ps_image0_offset describe the allocation).
//
// Same bytes per element as fp32 (2 + 2); value = float(hi) + float(lo), 16 significant bits.
#pragma once
#include "fdet_common.h"

typedef __bf16 ps_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ps_bf16x4 __attribute__((ext_vector_type(4)));

namespace fdet {

struct PsGeo {
  int N, C, H, W, HP, WP, C8;
  int plane, img;          // 16-byte units
};

inline int ps_wp(int W) { return W + 1 <= 16 ? 16 : (W + 1 <= 32 ? 32 : (W + 1 <= 64 ? 64 : 0)); }

inline bool ps_geo(int N, int C, int H, int W, PsGeo& g) {
  g.N = N; g.C = C; g.H = H; g.W = W;
  g.WP = ps_wp(W);
  g.HP = (H + 2) & ~1;
  g.C8 = C / 8;
  if (g.WP == 0 || C % 8 != 0 || N < 1 || H < 1) return false;
  const long long plane = (long long)g.C8 * g.HP * g.WP;
  if ((N + 2) * 2 * plane >= (1ll << 31)) return false;       // 32-bit unit indices
  g.plane = (int)plane;
  g.img = 2 * g.plane;
  return true;
}

// Column strips (round 4): a map wider than 62 columns is kept as S strips of Ws <= 62 columns, every strip a PS "image" of
// its own (64 slots per row; strip-image index = s * Nimg + n, strip-major), so that the 64-slot kernels serve rows of any
// width.  Slot 0 / slot Ws + 1 of an interior strip edge hold the NEIGHBOUR strip's edge column (a real halo, written by
// fdet_ps_from_f32 or fdet_ps_halo_exchange) instead of a zero; the last strip holds Wlast <= Ws columns, the rest stays zero.
struct PsStrips {
  int S, Nimg, Ws, Wf, Wlast;
};

inline bool ps_geo_strips(int N, int C, int H, int W, PsGeo& g, PsStrips& st) {
  st.S = 1; st.Nimg = N; st.Ws = W; st.Wf = W; st.Wlast = W;
  if (W <= 63) return ps_geo(N, C, H, W, g);
  if (W & 1) return false;
  st.S = (W + 61) / 62;
  st.Ws = ((W + st.S - 1) / st.S + 1) & ~1;
  st.Wlast = W - (st.S - 1) * st.Ws;
  if (st.Wlast <= 0 || st.Ws > 62 || (long long)N * st.S >= (1 << 20)) return false;
  if (!ps_geo(N * st.S, C, H, st.Ws, g)) return false;
  return g.WP == 64;
}

}  // namespace fdet

// hi/lo split of four floats into packed bf16 pairs (RNE, lo = bf16(x - float(hi))): hi[0] = {x0,x1}, hi[1] = {x2,x3}
__device__ __forceinline__ void ps_split4(const float (&f)[4], unsigned (&hi)[2], unsigned (&lo)[2]) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const bf16x2_t h = {(__bf16)f[2 * k], (__bf16)f[2 * k + 1]};
    hi[k] = __builtin_bit_cast(unsigned, h);
    const float h0 = __builtin_bit_cast(float, hi[k] << 16), h1 = __builtin_bit_cast(float, hi[k] & 0xffff0000u);
    const bf16x2_t l = {(__bf16)(f[2 * k] - h0), (__bf16)(f[2 * k + 1] - h1)};
    lo[k] = __builtin_bit_cast(unsigned, l);
  }
}
// the hi parts only (precision16: bf16 activations, RNE): hi[0] = {x0,x1}, hi[1] = {x2,x3}
__device__ __forceinline__ void ps_hi4(const float (&f)[4], unsigned (&hi)[2]) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const bf16x2_t h = {(__bf16)f[2 * k], (__bf16)f[2 * k + 1]};
    hi[k] = __builtin_bit_cast(unsigned, h);
  }
}
// value of element k (0 / 1) of a packed hi / lo dword pair
__device__ __forceinline__ float ps_join(unsigned hi, unsigned lo, int k) {
  const float h = __builtin_bit_cast(float, k ? (hi & 0xffff0000u) : (hi << 16));
  const float l = __builtin_bit_cast(float, k ? (lo & 0xffff0000u) : (lo << 16));
  return h + l;
}
