// Weight/bias gradient of the 3x3 convs on the bf16 matrix cores with fp32-level accuracy
// (bf16x3: x = hi + lo, products a_hi*b_lo + a_lo*b_hi + a_hi*b_hi, fp32 accumulation).
//
//   dW[co][ci][ky][kx] = sum_q dz[co][q] * x[ci][q + ky*P + kx - 1]          (P = row pitch)
// The MFMA K index is 16 consecutive POSITIONS, so a lane's fragment is 8 consecutive bf16 of one
// channel row = one aligned 16-byte LDS read -- as long as the start is a multiple of 8.  The row
// pitch P is a multiple of 8, which keeps the ky shifts aligned; the +-1 element kx shifts are
// moved onto the dz operand (u = q + kx - 1  ->  dz[u - kx + 1]) and done in registers:
// kx=1 reads the aligned chunk, kx=0 / kx=2 are a one-element funnel shift (v_alignbyte) of two
// neighbouring chunks.  So per 16-position step a wave issues 12 ds_read_b128, 16 v_alignbyte and
// 27 MFMAs (9 taps x 3 split products).
//
// Tiles (bf16 hi and lo): dz [64 co][QZ], x [32 ci][PX]; a band = R virtual rows (256 positions);
// the fp32 -> (hi,lo) split happens while staging global -> registers -> LDS, the next band is
// prefetched during the MFMAs.  One workgroup (4 waves: co tile x K half) per CU and ci group,
// persistent over bands; K halves are combined in LDS, slabs reduced in fixed order.
#include "fdet_common.h"
#include <algorithm>

using namespace fdet;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NTHR = 256;
constexpr int ZP = 16;        // front pad (elements) of the dz rows
constexpr int XP = 8;         // front pad of the x rows
// prefetch register budget (dz tile, x tile); the scalar-load layout (odd W) gets less because
// every load carries its own address registers
__host__ __device__ constexpr int zreg(int vw) { return vw == 1 ? 32 : 64; }
__host__ __device__ constexpr int xreg(int vw) { return vw == 1 ? 40 : 64; }

constexpr int MAXL = 16;      // layers per batched launch
struct WgX3Args {
  const float* x[MAXL];    // per layer [N,Cin,H,W]
  const float* dz[MAXL];   // per layer [N,Cout,H,W]
  float* ws;               // [L][nslab][9][CoP][CiP]
  float* wsb;              // [L][nslab][CoP]
  int N, Cin, Cout, CoP, CiP, H, W, P, R, VR, QZ, PX, Kext, nbands, L, ncob;
  int NSEG, CW;            // column segments per row (wide images), columns per segment
  unsigned magic_h1;
};

template <int VW> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = f32x2; };
template <> struct Vec<4> { using T = f32x4; };
template <int VW> __device__ __forceinline__ float vget(const typename Vec<VW>::T& v, int k) { return v[k]; }
template <> __device__ __forceinline__ float vget<1>(const float& v, int) { return v; }
template <int VW> __device__ __forceinline__ typename Vec<VW>::T vzero() { typename Vec<VW>::T z = {}; return z; }
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }

__device__ __forceinline__ int fdiv(int v, unsigned magic) { return magic ? (int)__umulhi((unsigned)v, magic) : v; }

// write VW consecutive elements (hi and lo) at element index e of a bf16 row
template <int VW>
__device__ __forceinline__ void put_split(__bf16* hi, __bf16* lo, int e, const typename Vec<VW>::T& v) {
  __bf16 h[VW], l[VW];
#pragma unroll
  for (int k = 0; k < VW; ++k) {
    const float f = vget<VW>(v, k);
    h[k] = (__bf16)f;
    l[k] = (__bf16)(f - (float)h[k]);
  }
  if constexpr (VW == 4) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<bf16x4*>(hi + e) = bf16x4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<bf16x4*>(lo + e) = bf16x4{l[0], l[1], l[2], l[3]};
  } else if constexpr (VW == 2) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<bf16x2*>(hi + e) = bf16x2{h[0], h[1]};
    *reinterpret_cast<bf16x2*>(lo + e) = bf16x2{l[0], l[1]};
  } else {
    hi[e] = h[0];
    lo[e] = l[0];
  }
}

// vector load; PACK rows are only 4-byte aligned (row width not a multiple of 4)
template <int VW, bool UNALIGNED>
__device__ __forceinline__ typename Vec<VW>::T ldvec(const float* p) {
  typename Vec<VW>::T v;
  if (UNALIGNED) __builtin_memcpy(&v, p, 4 * VW);
  else v = *reinterpret_cast<const typename Vec<VW>::T*>(p);
  return v;
}
// PACK: the quad was loaded s elements to the left of its place: out[i] = i + s < 4 ? in[i + s] : 0
template <int VW, bool PACK>
__device__ __forceinline__ typename Vec<VW>::T pk_shift(const typename Vec<VW>::T& v, int s) {
  if constexpr (PACK && VW == 4) {
    f32x4 a = v;
    if (s & 1) a = f32x4{a[1], a[2], a[3], 0.f};
    if (s & 2) a = f32x4{a[2], a[3], 0.f, 0.f};
    return a;
  } else {
    return v;
  }
}

// one-element funnel shift across two 16-byte chunks: out = elements [s .. s+8) of (lo_chunk | hi_chunk),
// s = 1 (SH = 0) or s = 7 (SH = 1)
template <int SH>
__device__ __forceinline__ bf16x8 shift_chunks(const bf16x8& c_lo, const bf16x8& c_hi) {
  const u32x4 a = __builtin_bit_cast(u32x4, c_lo), b = __builtin_bit_cast(u32x4, c_hi);
  u32x4 o;
  if (SH == 0) {            // elements 1..8: dwords (a0>>16|a1<<16), (a1|a2), (a2|a3), (a3|b0)
    o[0] = __builtin_amdgcn_alignbyte(a[1], a[0], 2);
    o[1] = __builtin_amdgcn_alignbyte(a[2], a[1], 2);
    o[2] = __builtin_amdgcn_alignbyte(a[3], a[2], 2);
    o[3] = __builtin_amdgcn_alignbyte(b[0], a[3], 2);
  } else {                  // elements 7..14: dwords (a3>>16|b0<<16), (b0|b1), (b1|b2), (b2|b3)
    o[0] = __builtin_amdgcn_alignbyte(b[0], a[3], 2);
    o[1] = __builtin_amdgcn_alignbyte(b[1], b[0], 2);
    o[2] = __builtin_amdgcn_alignbyte(b[2], b[1], 2);
    o[3] = __builtin_amdgcn_alignbyte(b[3], b[2], 2);
  }
  return __builtin_bit_cast(bf16x8, o);
}

// MTC co tiles per workgroup (64 or 32 output channels), 32 input channels per workgroup.
// SEG: rows wider than 16 vector lanes are cut into column segments of CW <= 56 columns.  The sum
// over (dz position, x position) pairs is partitioned by dz ROW (band) and x COLUMN (segment), so a
// segment stages its own x columns and dz columns [x0-1, x0+CW] -- lanes 14 / 15 of each 16-lane row
// group fetch the two dz halo columns.
// PACK (VW == 4, 4 <= W <= 16): narrow rows are staged FOUR rows per 16-lane group -- lane = (row
// sub-index, column quad) -- with 16-byte loads; the last quad of a row whose width is not a multiple
// of 4 loads the row's last four elements and shifts them into place (branch-free, never reads past
// the row).  One dword-per-lane staging pass per row made the 15x15 layers instruction-issue bound.
template <int MTC, int VW, bool SEG, bool PACK = false>
__global__ void __launch_bounds__(NTHR, 1)
k_wgrad3x3_x3(const WgX3Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MB = MTC * 32;
  constexpr int KS = 4 / MTC;                         // K splits
  constexpr int ZCH = MB / 16;                        // dz channel slots per thread (16 channels per pass)
  constexpr int XCH = 2;
  constexpr int RZ = zreg(VW) / (ZCH * VW);
  constexpr int RX = xreg(VW) / (XCH * VW);
  using VT = typename Vec<VW>::T;
  __bf16* Zh = reinterpret_cast<__bf16*>(smem);       // [MB][QZ]
  __bf16* Zl = Zh + MB * a.QZ;
  __bf16* Xh = Zl + MB * a.QZ;                        // [32][PX]
  __bf16* Xl = Xh + 32 * a.PX;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int cib = blockIdx.y, cob = blockIdx.z % a.ncob, layer = blockIdx.z / a.ncob;
  const float* __restrict__ gx = a.x[layer];
  const float* __restrict__ gdz = a.dz[layer];
  const int m = wid % MTC, kh = wid / MTC;
  const int H1 = a.H + 1, P = a.P, W = a.W, R = a.R;
  const int nks = a.Kext / 16;
  const int ks0 = (nks * kh) / KS, ks1 = (nks * (kh + 1)) / KS;

  {
    f32x4* z = reinterpret_cast<f32x4*>(smem);
    const int n16 = (MB * a.QZ + 32 * a.PX) * 4 / 16;
    for (int t = tid; t < n16; t += NTHR) z[t] = f32x4{0.f, 0.f, 0.f, 0.f};   // pads stay zero
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const bf16x8* zrow_h = reinterpret_cast<const bf16x8*>(Zh + (m * 32 + l31) * a.QZ) + half;
  const bf16x8* zrow_l = reinterpret_cast<const bf16x8*>(Zl + (m * 32 + l31) * a.QZ) + half;
  const bf16x8* xrow_h = reinterpret_cast<const bf16x8*>(Xh + l31 * a.PX) + half;
  const bf16x8* xrow_l = reinterpret_cast<const bf16x8*>(Xl + l31 * a.PX) + half;
  const int p8 = P / 8;                               // row pitch in 16-byte chunks

  // staging geometry: 16 lanes per row (W/VW <= 16, or one segment of <= 14 lanes), 16 channels per pass
  const int xv = tid & 15;
  const int chl = tid >> 4;
  int x0 = 0, wcur = W;                               // current segment
  bool lane_ok = xv < W / VW;
  // PACK geometry: column quad / row sub-index of this lane, load offset and right shift of the quad
  const int xq = tid & 3, rsub = (tid >> 2) & 3;
  const int pk_c0 = min(4 * xq, W - 4), pk_s = 4 * xq - pk_c0;
  if (PACK) lane_ok = 4 * xq < W;
  const int co0 = cob * MB, ci0 = cib * 32;
  const int HW = a.H * W;
  float bpart[ZCH];
#pragma unroll
  for (int c = 0; c < ZCH; ++c) bpart[c] = 0.f;

  VT pz[ZCH][RZ], px[XCH][RX];
  float pzh[SEG ? ZCH : 1][SEG ? RZ : 1];             // dz halo column (lane 14: left, lane 15: right)
#define X3_WG_LOAD(ITEM)                                                                          \
  {                                                                                               \
    int V0;                                                                                       \
    if (SEG) {                                                                                    \
      const int band_ = (ITEM) / a.NSEG, seg_ = (ITEM) - band_ * a.NSEG;                          \
      V0 = band_ * R; x0 = seg_ * a.CW; wcur = min(a.CW, W - x0);                                 \
      lane_ok = xv < wcur / VW;                                                                   \
    } else {                                                                                      \
      V0 = (ITEM) * R;                                                                            \
    }                                                                                             \
    const int hcol_ = xv == 14 ? x0 - 1 : x0 + wcur;                                              \
    const bool hok_ = SEG && xv >= 14 && hcol_ >= 0 && hcol_ < W;                                 \
    _Pragma("unroll") for (int r_ = 0; r_ < RZ; ++r_) {                                           \
      const int rr_ = PACK ? 4 * r_ + rsub : r_;                                                  \
      const int v = (V0) + rr_;                                                                   \
      const int n = fdiv(v, a.magic_h1), yy = v - n * H1 - 1;                                     \
      const bool rok = rr_ < R && v < a.VR && yy >= 0;                                            \
      const float* rowp = gdz + ((size_t)n * a.Cout * a.H + yy) * W;                             \
      _Pragma("unroll") for (int c_ = 0; c_ < ZCH; ++c_) {                                        \
        const int ch = co0 + c_ * 16 + chl;                                                       \
        pz[c_][r_] = (rok && lane_ok && ch < a.Cout)                                              \
                         ? pk_shift<VW, PACK>(ldvec<VW, PACK>(rowp + (size_t)ch * HW + x0 + (PACK ? pk_c0 : xv * VW)), pk_s) : vzero<VW>(); \
        if (SEG) pzh[c_][r_] = (rok && hok_ && ch < a.Cout) ? rowp[(size_t)ch * HW + hcol_] : 0.f; \
      }                                                                                           \
    }                                                                                             \
    _Pragma("unroll") for (int r_ = 0; r_ < RX; ++r_) {                                           \
      const int rr_ = PACK ? 4 * r_ + rsub : r_;                                                  \
      const int v = (V0) - 1 + rr_;                                                               \
      const int n = fdiv(max(v, 0), a.magic_h1), yy = v - n * H1 - 1;                             \
      const bool rok = rr_ < R + 2 && v >= 0 && v < a.VR && yy >= 0;                              \
      const float* rowp = gx + ((size_t)n * a.Cin * a.H + yy) * W;                               \
      _Pragma("unroll") for (int c_ = 0; c_ < XCH; ++c_) {                                        \
        const int ch = ci0 + c_ * 16 + chl;                                                       \
        px[c_][r_] = (rok && lane_ok && ch < a.Cin)                                               \
                         ? pk_shift<VW, PACK>(ldvec<VW, PACK>(rowp + (size_t)ch * HW + x0 + (PACK ? pk_c0 : xv * VW)), pk_s) : vzero<VW>(); \
      }                                                                                           \
    }                                                                                             \
  }
#define X3_WG_STORE()                                                                             \
  {                                                                                               \
    if (SEG ? xv < 14 : lane_ok) {                                                                \
      _Pragma("unroll") for (int r_ = 0; r_ < RZ; ++r_) {                                         \
        const int rr_ = PACK ? 4 * r_ + rsub : r_;                                                \
        if (rr_ < R) {                                                                            \
          _Pragma("unroll") for (int c_ = 0; c_ < ZCH; ++c_) {                                    \
            const int row_ = (c_ * 16 + chl) * a.QZ;                                              \
            put_split<VW>(Zh + row_, Zl + row_, ZP + rr_ * P + (PACK ? 4 * xq : xv * VW), pz[c_][r_]); \
            _Pragma("unroll") for (int k_ = 0; k_ < VW; ++k_) bpart[c_] += vget<VW>(pz[c_][r_], k_); \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
      _Pragma("unroll") for (int r_ = 0; r_ < RX; ++r_) {                                         \
        const int rr_ = PACK ? 4 * r_ + rsub : r_;                                                \
        if (rr_ < R + 2) {                                                                        \
          _Pragma("unroll") for (int c_ = 0; c_ < XCH; ++c_) {                                    \
            const int row_ = (c_ * 16 + chl) * a.PX;                                              \
            put_split<VW>(Xh + row_, Xl + row_, XP + rr_ * P + (PACK ? 4 * xq : xv * VW), px[c_][r_]); \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
    }                                                                                             \
    /* halo columns last: same wave, LDS writes in program order, so they win over the zero fill */ \
    if (SEG && xv >= 14) {                                                                        \
      _Pragma("unroll") for (int r_ = 0; r_ < RZ; ++r_) {                                         \
        if (r_ < R) {                                                                             \
          _Pragma("unroll") for (int c_ = 0; c_ < ZCH; ++c_) {                                    \
            const int row_ = (c_ * 16 + chl) * a.QZ;                                              \
            put_split<1>(Zh + row_, Zl + row_, ZP + r_ * P + (xv == 14 ? -1 : s_wcur), pzh[c_][r_]); \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
    }                                                                                             \
  }

  // work items: bands (x column segments)
  const int nitems = a.nbands * (SEG ? a.NSEG : 1);
  int band = blockIdx.x;
  int s_wcur = W;                    // segment width of the tile held in the prefetch registers
  if (band < nitems) { X3_WG_LOAD(band) s_wcur = wcur; }
  for (; band < nitems; band += gridDim.x) {
    __syncthreads();                 // previous band's MFMAs done (first pass: zero fill done)
    X3_WG_STORE()
    __syncthreads();
    const int nb = band + gridDim.x;
    if (nb < nitems) { X3_WG_LOAD(nb) s_wcur = wcur; }
#pragma unroll 1
    for (int ks = ks0; ks < ks1; ++ks) {
      // dz chunks at elements e, e+8, e+16 (e = 16*ks + 8*half); kx=1 <- [e+8,e+16)
      const bf16x8 z0h = zrow_h[2 * ks], z1h = zrow_h[2 * ks + 1], z2h = zrow_h[2 * ks + 2];
      const bf16x8 z0l = zrow_l[2 * ks], z1l = zrow_l[2 * ks + 1], z2l = zrow_l[2 * ks + 2];
      bf16x8 ah[3], al[3];
      ah[1] = z1h; al[1] = z1l;
      ah[0] = shift_chunks<0>(z1h, z2h); al[0] = shift_chunks<0>(z1l, z2l);     // elements e+9 ..
      ah[2] = shift_chunks<1>(z0h, z1h); al[2] = shift_chunks<1>(z0l, z1l);     // elements e+7 ..
      bf16x8 bh[3], bl[3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) { bh[ky] = xrow_h[2 * ks + ky * p8]; bl[ky] = xrow_l[2 * ks + ky * p8]; }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int t = ky * 3 + kx;
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kx], bl[ky], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kx], bh[ky], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kx], bh[ky], acc[t], 0, 0, 0);
        }
    }
  }

  // ---- combine K splits through LDS, one slab per workgroup; bias partials over the 16 column lanes
  __syncthreads();
  {
    float* red = reinterpret_cast<float*>(smem);          // [MTC][144][64]
#pragma unroll 1
    for (int rnd = 1; rnd < KS; ++rnd) {
      if (kh == rnd) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(m * 144 + t * 16 + r) * 64 + lane] = acc[t][r];
      }
      __syncthreads();
      if (kh == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] += red[(m * 144 + t * 16 + r) * 64 + lane];
      }
      __syncthreads();
    }
  }
  const int s = layer * gridDim.x + blockIdx.x;
  if (kh == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int ci = ci0 + l31;
        a.ws[(((size_t)s * 9 + t) * a.CoP + co) * a.CiP + ci] = acc[t][r];
      }
    }
  }
  if (cib == 0) {
#pragma unroll
    for (int c = 0; c < ZCH; ++c) {
      float v = bpart[c];
      v += __shfl_xor(v, 8, 16); v += __shfl_xor(v, 4, 16); v += __shfl_xor(v, 2, 16); v += __shfl_xor(v, 1, 16);
      if (xv == 0) a.wsb[(size_t)s * a.CoP + co0 + c * 16 + chl] = v;
    }
  }
}

// hi/lo split of two floats with packed ops: 5 VALU per pair (cvt_pk, shift, mask, pk_add with negated
// operand, cvt_pk); hi and lo come back as bf16 pairs in one dword each.  Same values as put_split (RNE).
__device__ __forceinline__ void split_pair(float f0, float f1, unsigned& hi, unsigned& lo) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t h = {(__bf16)f0, (__bf16)f1};
  hi = __builtin_bit_cast(unsigned, h);
  const f32x2 hf = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
  const f32x2 l = f32x2{f0, f1} - hf;
  const bf16x2_t lb = {(__bf16)l[0], (__bf16)l[1]};
  lo = __builtin_bit_cast(unsigned, lb);
}
template <int VW, int DBG = 0>
__device__ __forceinline__ void put_split_pk(__bf16* hi, __bf16* lo, const typename Vec<VW>::T& v) {
  if constexpr (VW == 4) {
    unsigned h0, l0, h1, l1;
    if (DBG & 32) { h0 = __builtin_bit_cast(unsigned, v[0]); h1 = __builtin_bit_cast(unsigned, v[1]); l0 = __builtin_bit_cast(unsigned, v[2]); l1 = __builtin_bit_cast(unsigned, v[3]); }
    else { split_pair(v[0], v[1], h0, l0); split_pair(v[2], v[3], h1, l1); }
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    if (DBG & 16) { asm volatile("" ::"v"(h0), "v"(h1), "v"(l0), "v"(l1)); return; }
    *reinterpret_cast<u32x2_t*>(hi) = u32x2_t{h0, h1};
    *reinterpret_cast<u32x2_t*>(lo) = u32x2_t{l0, l1};
  } else if constexpr (VW == 2) {
    unsigned h0, l0;
    split_pair(v[0], v[1], h0, l0);
    *reinterpret_cast<unsigned*>(hi) = h0;
    *reinterpret_cast<unsigned*>(lo) = l0;
  } else {
    put_split<1>(hi, lo, 0, v);
  }
}

// Staging loads of the pipelined kernel: buffer loads issued from asm, so that (a) the destination
// class is ours -- one of the two in-flight register sets lives in the 112 accumulation registers the
// nine accumulator tiles leave free, the other in arch VGPRs -- and (b) the waits are ours: hipcc
// merged its own counted waits to vmcnt(0) at the loop head (and, out of VGPRs, parked loaded values
// in AGPRs with a copy right behind the load), which serialised memory and matrix phases.  An offset
// past the descriptor's range returns zeros: the predicate of a padded / out-of-batch element is folded
// into the offset, no branch.  Protocol (cdna_hip_programming.md 5.7 form ii): loads "=a"/"=v", then
// one s_waitcnt vmcnt(N) and a "+a"/"+v" pass-through of every destination before the first consumer.
template <int VW, bool AG>
__device__ __forceinline__ void aload(typename Vec<VW>::T& d, unsigned off, const u32x4& rs) {
  if constexpr (VW == 4) {
    if constexpr (AG) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=a"(d) : "v"(off), "s"(rs));
    else asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(d) : "v"(off), "s"(rs));
  } else if constexpr (VW == 2) {
    if constexpr (AG) asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=a"(d) : "v"(off), "s"(rs));
    else asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(d) : "v"(off), "s"(rs));
  } else {
    if constexpr (AG) asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=a"(d) : "v"(off), "s"(rs));
    else asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(d) : "v"(off), "s"(rs));
  }
}
template <bool AG, class T>
__device__ __forceinline__ void apass(T& d) {
  if constexpr (AG) asm volatile("" : "+a"(d)); else asm volatile("" : "+v"(d));
}

// ---------------------------------------------------------------------------------------------
// Pipelined variant (64 output channels per workgroup; rows of four floats or one float per lane, P = 16 VW):
// bands of R * P = 128 positions (nine 16-position MFMA steps), TWO LDS tiles, contiguous bands per workgroup.
// While the MFMAs of band b run on tile b & 1,
//   * the split + LDS writes ("jobs") of band b+1, whose fp32 rows are already in registers, ride one per
//     tap group (three MFMAs) behind sched_barriers, so their VALU issues in the MFMAs' shadow;
//   * the global loads of band b+3 follow four slots behind the job that freed their register: two register
//     sets in flight (one in the accumulation registers the nine accumulators leave free, one in arch VGPRs),
//     waited for with a counted vmcnt one whole band later;
//   * the two x halo rows a band shares with its predecessor are copied LDS -> LDS from the other tile
//     (no global re-read, no split).
// The staging phase of the single-tile kernel (matrix pipe idle: one wave per SIMD, nothing else to fill it)
// is gone; what remains additive is measured in the DBG note below.
// ---------------------------------------------------------------------------------------------
// DBG (development builds, -DFDET_WG_DBG + env FDET_WG_DBG=<sum of bits>; wrong results, timing only): 1 no staging
// jobs, 2 no global loads in the loop, 4 no MFMA steps, 8 every load from image 0 (L2-resident), 16 jobs without
// their LDS writes, 32 jobs without the split arithmetic, 64 non-zero initial tiles.  Measured with them on the
// 60x60 layers (ms per launch incl. reduce): MFMA alone on zero tiles 0.350, on non-zero tiles 0.405 (clock),
// + L2-resident loads +0.00, + split VALU +0.04, + LDS writes +0.04, + real HBM traffic +0.07 -> 0.56.
// LPR = lanes per staged row (16, or 32 for one-float lanes of rows up to 32 columns: the 30x30 layers -- their
// two-float form made hipcc spill accumulators around the 64-bit staging tuples); P = LPR * VW.
// PK4 (rows of P-3..P floats, P = LPR = 16 or 32, VW = 4): the LPR lanes of a channel slot are 4 rows x P/4 float4 quads instead
// of LPR columns -- a row is staged with P/4 16-byte loads (the last quad loaded P - W elements early and shifted into place,
// zeros behind the row's end) instead of W dword loads, 32/P register slots per thread and channel slot (rows rsub + 4 j):
// a quarter of the load / split / LDS-write instructions of the one-float form, which was issue-bound on the 15x15 layers
// (0.344 -> 0.283 ms for the 16 layers).
template <int VW, int DBG = 0, int LPR = 16, int PK4 = 0>
__global__ void __launch_bounds__(NTHR, 1)
k_wgrad3x3_x3_pipe(const WgX3Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CPP = NTHR / LPR;                      // channels per staging pass (16 or 8)
  constexpr int MTC = 2, MB = 64, ZCH = MB / CPP, XCH = 32 / CPP;
  constexpr int RZ = PK4 ? 32 / LPR : 128 / (LPR * VW); // dz row slots per thread and channel slot (= R unless PK4)
  constexpr int RX = RZ + 2;
  using VT = typename Vec<VW>::T;
  // compile-time tile geometry (plan_x3 computes the same numbers; the launcher checks them)
  constexpr int P = PK4 ? LPR : LPR * VW, R = PK4 ? 128 / LPR : RZ, KEXT = 144, QZ = KEXT + 24, PX = KEXT + 2 * P + 8;
  constexpr int QPR = P / 4;                            // PK4: quads per row
  static_assert(R * P == 128 && (QZ / 8) % 2 == 1 && (PX / 8) % 2 == 1, "band = 128 positions, odd 16-byte row pitches");
  constexpr int tile_elems = MB * QZ * 2 + 32 * PX * 2;          // bf16 elements per tile (hi + lo, Z then X)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int cib = blockIdx.y, cob = blockIdx.z % a.ncob, layer = blockIdx.z / a.ncob;
  const float* __restrict__ gx = a.x[layer];
  const float* __restrict__ gdz = a.dz[layer];
  const int m = wid & 1, kh = wid >> 1;
  const int H1 = a.H + 1, W = a.W;
  const int ks0 = kh ? 4 : 0, ks1 = kh ? 9 : 4;                 // nine 16-position steps: 4 + 5
  __bf16* const base = reinterpret_cast<__bf16*>(smem);
  {
    f32x4* z = reinterpret_cast<f32x4*>(smem);
    constexpr int n16 = 2 * tile_elems * 2 / 16;
    for (int t = tid; t < n16; t += NTHR) z[t] = (DBG & 64) ? f32x4{1.37f + t, -0.731f * t, 3.3e-3f * t, 1.1f} : f32x4{0.f, 0.f, 0.f, 0.f};   // pads stay zero
  }
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  constexpr int p8 = P / 8;
  const int xv = tid & (LPR - 1), chl = tid / LPR;
  const int rsub = PK4 ? xv / QPR : 0, qd = xv % QPR;             // PK4: row within a group of four, quad within the row
  const int ecol = PK4 ? qd * 4 : xv * VW;                        // first LDS column of this lane's elements
  const int lcol = PK4 ? min(qd * 4, W - 4) : xv * VW;            // first column it loads (the last quad starts early)
  const int shq = ecol - lcol;                                    // ... and is shifted by this many elements
  const bool lane_ok = PK4 ? true : xv < W / VW;
#define WGP_ROW(r_) (PK4 ? rsub + 4 * (r_) : (r_))
  const int co0 = cob * MB, ci0 = cib * 32;
  const int HW = a.H * W;
  float bpart[ZCH];
#pragma unroll
  for (int c = 0; c < ZCH; ++c) bpart[c] = 0.f;
  const unsigned long long pa_z = (unsigned long long)gdz, pa_x = (unsigned long long)gx;
  const u32x4 rz = {(unsigned)pa_z, (unsigned)(pa_z >> 32), (unsigned)(a.N * a.Cout * HW) * 4u, 0x00020000u};
  const u32x4 rx = {(unsigned)pa_x, (unsigned)(pa_x >> 32), (unsigned)(a.N * a.Cin * HW) * 4u, 0x00020000u};
  unsigned zch_off[ZCH], xch_off[XCH];
  bool zch_ok[ZCH], xch_ok[XCH];
#pragma unroll
  for (int c = 0; c < ZCH; ++c) { zch_off[c] = (unsigned)(co0 + c * CPP + chl) * HW * 4u; zch_ok[c] = co0 + c * CPP + chl < a.Cout; }
#pragma unroll
  for (int c = 0; c < XCH; ++c) { xch_off[c] = (unsigned)(ci0 + c * CPP + chl) * HW * 4u; xch_ok[c] = ci0 + c * CPP + chl < a.Cin; }
  // two register sets, used alternately (0: accumulation registers, 1: arch VGPRs): register l of a set is
  // re-loaded (band + 2) a few slots after job l staged it, so every load has more than one whole band of
  // MFMAs to land
  VT pz0[ZCH][RZ], px0[XCH][RX], pz1[ZCH][RZ], px1[XCH][RX];
  // A workgroup walks CONTIGUOUS bands: the first two x rows of a band are the last two of the band before,
  // already split in the other tile -- they are copied LDS -> LDS (no global re-read, no VALU), so a band
  // loads and splits R new rows of dz and R new rows of x.
  constexpr int NZJ = ZCH * RZ, NXJ = XCH * RZ, NJ = NZJ + NXJ;   // jobs = loads per set
  constexpr int JPS = (NJ + 27) / 28;                         // jobs (and loads) per slot
  constexpr int LDS0 = 4;                                     // load l rides LDS0 slots behind job l
  // load L (0 .. NJ-1) of the band whose first virtual row is V0 into register L of set S
#define WGP_LOAD1(S, L, V0)                                                                       \
  {                                                                                               \
    if ((L) < NZJ) {                                                                              \
      const int c_ = (L) / RZ, r_ = (L) % RZ;                                                 \
      const int v = (V0) + WGP_ROW(r_);                                                           \
      const int n = (int)__umulhi((unsigned)min(v, a.VR), a.magic_h1), yy = v - n * H1 - 1;       \
      const bool ok = v < a.VR && yy >= 0 && lane_ok && zch_ok[c_];                               \
      const unsigned off = ((unsigned)(((DBG & 8) ? 0 : n) * a.Cout * a.H + yy) * W + lcol) * 4u + zch_off[c_];  \
      aload<VW, S == 0>(pz##S[c_][r_], ok ? off : 0x80000000u, rz);                               \
    } else {                                                                                      \
      const int l2_ = (L) - NZJ < NXJ ? (L) - NZJ : 0;                                        \
      const int c_ = l2_ / RZ, r_ = 2 + l2_ % RZ;                                             \
      const int v = (V0) + 1 + WGP_ROW(l2_ % RZ);                                                 \
      const int n = (int)__umulhi((unsigned)min(max(v, 0), a.VR), a.magic_h1), yy = v - n * H1 - 1; \
      const bool ok = v >= 0 && v < a.VR && yy >= 0 && lane_ok && xch_ok[c_];                     \
      const unsigned off = ((unsigned)(((DBG & 8) ? 0 : n) * a.Cin * a.H + yy) * W + lcol) * 4u + xch_off[c_];   \
      aload<VW, S == 0>(px##S[c_][r_], ok ? off : 0x80000000u, rx);                               \
    }                                                                                             \
  }
  // all but the NKEEP youngest loads have landed; every register of set S passes through the wait
#define WGP_WAIT(S, NKEEP)                                                                        \
  {                                                                                               \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NKEEP));                                             \
    _Pragma("unroll") for (int c_ = 0; c_ < ZCH; ++c_)                                            \
      _Pragma("unroll") for (int r_ = 0; r_ < RZ; ++r_) apass<S == 0>(pz##S[c_][r_]);             \
    _Pragma("unroll") for (int c_ = 0; c_ < XCH; ++c_)                                            \
      _Pragma("unroll") for (int r_ = 2; r_ < RX; ++r_) apass<S == 0>(px##S[c_][r_]);             \
  }
  // staging job J (0 .. NJ-1) of the band in set S into tile TB; branch-free: the 16 lanes of a row cover its
  // whole pitch (LPR * VW = P), lanes past the row's width hold zeros (their loads are out of range) and
  // write them into the pad columns, which must be zero anyway
#define WGP_JOB(S, J, TB)                                                                         \
  {                                                                                               \
    __bf16* tb_ = base + (TB) * tile_elems;                                                       \
    if ((J) < NZJ) {                                                                              \
      const int c_ = (J) / RZ, r_ = (J) % RZ;                                                 \
      __bf16* zh_ = tb_ + (c_ * CPP + chl) * QZ + ZP + WGP_ROW(r_) * P + ecol;                    \
      const VT t_ = pk_shift<VW, PK4 != 0>(pz##S[c_][r_], shq);                                    \
      put_split_pk<VW, DBG>(zh_, zh_ + MB * QZ, t_);                                              \
      _Pragma("unroll") for (int k_ = 0; k_ < VW; ++k_) bpart[c_] += vget<VW>(t_, k_);            \
    } else {                                                                                      \
      const int j2_ = (J) - NZJ < NXJ ? (J) - NZJ : 0;                                        \
      const int c_ = j2_ / RZ, r_ = 2 + j2_ % RZ;                                                 \
      __bf16* xh_ = tb_ + 2 * MB * QZ + (c_ * CPP + chl) * PX + XP + (2 + WGP_ROW(j2_ % RZ)) * P + ecol; \
      put_split_pk<VW, DBG>(xh_, xh_ + 32 * PX, pk_shift<VW, PK4 != 0>(px##S[c_][r_], shq));      \
    }                                                                                             \
  }
  // operand fragments of one 16-position MFMA step, two sets (the reads of step s+1 are issued inside step s)
  bf16x8 fz[2][6], fb[2][6];
  // halo copy: 64 x rows (32 channels, hi and lo) x 2P elements = NCP 16-byte chunks per thread
  constexpr int NCP = P / 16, CPR = P / 4;            // chunks per thread, chunks per x row
  bf16x8 cpy[NCP];
  int cp_off[NCP];
#pragma unroll
  for (int i = 0; i < NCP; ++i) { const int q = tid + NTHR * i; cp_off[i] = 2 * MB * QZ + (q / CPR) * PX + XP + (q % CPR) * 8; }
#define WGP_FRAGS(F, TB, KS_)                                                                     \
  {                                                                                               \
    const __bf16* tb_ = base + (TB) * tile_elems;                                                 \
    const bf16x8* zh_ = reinterpret_cast<const bf16x8*>(tb_ + (m * 32 + l31) * QZ) + half;      \
    const bf16x8* zl_ = reinterpret_cast<const bf16x8*>(tb_ + MB * QZ + (m * 32 + l31) * QZ) + half; \
    const bf16x8* xh_ = reinterpret_cast<const bf16x8*>(tb_ + 2 * MB * QZ + l31 * PX) + half;  \
    const bf16x8* xl_ = reinterpret_cast<const bf16x8*>(tb_ + 2 * MB * QZ + 32 * PX + l31 * PX) + half; \
    _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                            \
      fz[F][i_] = zh_[2 * (KS_) + i_]; fz[F][3 + i_] = zl_[2 * (KS_) + i_];                       \
      fb[F][i_] = xh_[2 * (KS_) + i_ * p8]; fb[F][3 + i_] = xl_[2 * (KS_) + i_ * p8];             \
    }                                                                                             \
  }
  // slot U (0 .. 31: eight per MFMA step, behind tap groups 1..8) of a band: its jobs, then its loads
#define WGP_SLOT(S, U, V0N)                                                                       \
  {                                                                                               \
    if ((U) == 0) { _Pragma("unroll") for (int i_ = 0; i_ < NCP; ++i_) cpy[i_] = *reinterpret_cast<const bf16x8*>(base + (1 - (S)) * tile_elems + cp_off[i_] + R * P); } \
    if ((U) == 2) { _Pragma("unroll") for (int i_ = 0; i_ < NCP; ++i_) *reinterpret_cast<bf16x8*>(base + (S) * tile_elems + cp_off[i_]) = cpy[i_]; } \
    _Pragma("unroll") for (int q_ = 0; q_ < JPS; ++q_) {                                          \
      if (!(DBG & 1) && (U) * JPS + q_ < NJ) WGP_JOB(S, ((U) * JPS + q_ < NJ ? (U) * JPS + q_ : 0), S) \
    }                                                                                             \
    _Pragma("unroll") for (int q_ = 0; q_ < JPS; ++q_) {                                          \
      if (!(DBG & 2) && (U) >= LDS0 && ((U) - LDS0) * JPS + q_ < NJ)                              \
        WGP_LOAD1(S, (((U) - LDS0) * JPS + q_ < NJ && (U) >= LDS0 ? ((U) - LDS0) * JPS + q_ : 0), V0N) \
    }                                                                                             \
  }
  // MFMA step ST (0..3: with slots; 4: the odd step of the second K half) of the band on tile PAR, fragment set F
#define WGP_STEP(PAR, S, ST, F, V0N)                                                              \
  {                                                                                               \
    bf16x8 ah[3], al[3];                                                                          \
    ah[1] = fz[F][1]; al[1] = fz[F][4];                                                           \
    ah[0] = shift_chunks<0>(fz[F][1], fz[F][2]); al[0] = shift_chunks<0>(fz[F][4], fz[F][5]);     \
    ah[2] = shift_chunks<1>(fz[F][0], fz[F][1]); al[2] = shift_chunks<1>(fz[F][3], fz[F][4]);     \
    _Pragma("unroll") for (int ky = 0; ky < 3; ++ky)                                              \
      _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                          \
        const int t = ky * 3 + kx;                                                                \
        if (!(DBG & 4)) {                                                                         \
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kx], fb[F][3 + ky], acc[t], 0, 0, 0); \
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kx], fb[F][ky], acc[t], 0, 0, 0);   \
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kx], fb[F][ky], acc[t], 0, 0, 0);   \
        }                                                                                         \
        if (t == 1 && (ST) < 3) WGP_FRAGS((F) ^ 1, PAR, ks0 + (ST) + 1)                           \
        if (t == 1 && (ST) == 3 && ks0 + 4 < ks1) WGP_FRAGS((F) ^ 1, PAR, ks0 + 4)                \
        if ((ST) < 4 && t >= 1) WGP_SLOT(S, (ST) * 8 + t - 1, V0N)                                \
        __builtin_amdgcn_sched_barrier(0);                                                        \
      }                                                                                           \
  }
  const int bpw = (a.nbands + (int)gridDim.x - 1) / (int)gridDim.x;       // contiguous bands per workgroup
  int item = blockIdx.x * bpw;
  const int last = min(item + bpw, a.nbands);
  const int vnone = a.VR + 8;                            // first row of a band that is not this workgroup's: every row fails v < VR
  // the first band's x halo rows (virtual rows V0 - 1, V0) come from global memory
  VT hx[XCH][2];
  {
    const bool b0 = item < last;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int v = item * R - 1 + r;
      const int n = (int)__umulhi((unsigned)min(max(v, 0), a.VR), a.magic_h1), yy = v - n * H1 - 1;
#pragma unroll
      for (int c = 0; c < XCH; ++c) {
        const bool ok = b0 && v >= 0 && v < a.VR && yy >= 0 && lane_ok && xch_ok[c] && (!PK4 || rsub == r);
        aload<VW, false>(hx[c][r], ok ? ((unsigned)(n * a.Cin * a.H + yy) * W + lcol) * 4u + xch_off[c] : 0x80000000u, rx);
      }
    }
#pragma unroll
    for (int l = 0; l < NJ; ++l) WGP_LOAD1(0, l, (b0 ? item * R : vnone))
  }
  WGP_WAIT(0, 0)
#pragma unroll
  for (int c = 0; c < XCH; ++c) { apass<false>(hx[c][0]); apass<false>(hx[c][1]); }
  __syncthreads();                                       // zero fill done
#pragma unroll
  for (int c = 0; c < XCH; ++c)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      __bf16* xh_ = base + 2 * MB * QZ + (c * CPP + chl) * PX + XP + r * P + ecol;
      if (!PK4 || rsub == r) put_split<VW>(xh_, xh_ + 32 * PX, 0, pk_shift<VW, PK4 != 0>(hx[c][r], shq));
    }
#pragma unroll
  for (int j = 0; j < NJ; ++j) WGP_JOB(0, j, 0)
#pragma unroll
  for (int l = 0; l < NJ; ++l) WGP_LOAD1(1, l, (item + 1 < last ? (item + 1) * R : vnone))
#pragma unroll
  for (int l = 0; l < NJ; ++l) WGP_LOAD1(0, l, (item + 2 < last ? (item + 2) * R : vnone))
  __syncthreads();
  // one band on tile PAR; the jobs of the next band (register set S = PAR ^ 1, into tile S) and the loads of
  // the band three ahead ride on MFMA steps 0..3, which every wave has (nine steps, two K halves).
  // In flight at the wait: set S (older), then set PAR.  Bands past the workgroup's last load zeros.
#define WGP_BAND(PAR, S)                                                                          \
  {                                                                                               \
    const int v0n_ = item + 3 < last ? (item + 3) * R : vnone;                                    \
    WGP_FRAGS(0, PAR, ks0)                                                                        \
    WGP_WAIT(S, NJ)                                                                               \
    WGP_STEP(PAR, S, 0, 0, v0n_)                                                                  \
    WGP_STEP(PAR, S, 1, 1, v0n_)                                                                  \
    WGP_STEP(PAR, S, 2, 0, v0n_)                                                                  \
    WGP_STEP(PAR, S, 3, 1, v0n_)                                                                  \
    if (ks0 + 4 < ks1) WGP_STEP(PAR, S, 4, 0, v0n_)                                               \
    /* retire BOTH register sets before the band ends: no asm-load destination is in flight across the loop back */ \
    /* edge, where hipcc bridges the two band bodies with register copies (tools/audit_asm_loads.py: 307 touches */ \
    /* of in-flight destinations before, 0 now).  The loads of band + 3 then have the rest of this band to land  */ \
    /* instead of two bands: measured cost in DESIGN.md 2.2c.                                                      */ \
    WGP_WAIT(PAR, 0)                                                                              \
    WGP_WAIT(S, 0)                                                                                \
    __syncthreads();                                                                              \
    item += 1;                                                                                    \
  }
  while (item < last) {
    WGP_BAND(0, 1)
    if (item >= last) break;
    WGP_BAND(1, 0)
  }
  WGP_WAIT(0, 0)                                         // the last bands' (all-zero) loads still target live registers
  WGP_WAIT(1, 0)
#undef WGP_BAND
#undef WGP_STEP
#undef WGP_SLOT
#undef WGP_FRAGS
#undef WGP_JOB
#undef WGP_WAIT
#undef WGP_LOAD1
#undef WGP_ROW
  // ---- combine the K halves through LDS, one slab per workgroup (as the base kernel)
  __syncthreads();
  {
    float* red = reinterpret_cast<float*>(smem);          // [2][144][64]
    if (kh == 1) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(m * 144 + t * 16 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += red[(m * 144 + t * 16 + r) * 64 + lane];
    }
  }
  const int s = layer * gridDim.x + blockIdx.x;
  if (kh == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        a.ws[(((size_t)s * 9 + t) * a.CoP + co) * a.CiP + ci0 + l31] = acc[t][r];
      }
  }
  if (cib == 0) {
#pragma unroll
    for (int c = 0; c < ZCH; ++c) {
      float v = bpart[c];
      if (LPR == 32) v += __shfl_xor(v, 16, 32);
      v += __shfl_xor(v, 8, 16); v += __shfl_xor(v, 4, 16); v += __shfl_xor(v, 2, 16); v += __shfl_xor(v, 1, 16);
      if (xv == 0) a.wsb[(size_t)s * a.CoP + co0 + c * CPP + chl] = v;
    }
  }
}

// Fixed-order reduction of the slabs (same scheme as fdet_wgrad3x3.hip)
struct WgX3Red { float* dW[MAXL]; float* db[MAXL]; };
__global__ void __launch_bounds__(1024)
k_wgx3_reduce(const float* __restrict__ ws_all, const float* __restrict__ wsb_all, int nslab, int Cout, int Cin,
              int CoP, int CiP, const WgX3Red out) {
  __shared__ float part[1024];
  const float* __restrict__ ws = ws_all + (size_t)blockIdx.z * nslab * 9 * CoP * CiP;
  const float* __restrict__ wsb = wsb_all + (size_t)blockIdx.z * nslab * CoP;
  float* __restrict__ dW = out.dW[blockIdx.z];
  float* __restrict__ db = out.db[blockIdx.z];
  const int tap = blockIdx.x;
  const int e = threadIdx.x & 255, ph = threadIdx.x >> 8;
  const int span = 4 * CiP;
  const int co0 = blockIdx.y * 4;
  for (int e0 = 0; e0 < span; e0 += 256) {
    const int idx = e0 + e;
    float s = 0.f;
    if (idx < span && co0 + idx / CiP < CoP) {
      const float* src = ws + ((size_t)tap * CoP + co0) * CiP + idx;
#pragma unroll 4
      for (int k = ph; k < nslab; k += 4) s += src[(size_t)k * 9 * CoP * CiP];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (ph == 0 && idx < span) {
      const int co = co0 + idx / CiP, ci = idx % CiP;
      if (co < Cout && ci < Cin)
        dW[((size_t)co * Cin + ci) * 9 + tap] = ((part[e] + part[256 + e]) + part[512 + e]) + part[768 + e];
    }
    __syncthreads();
  }
  if (db && tap == 0 && threadIdx.x < 4 * 64) {
    const int c = threadIdx.x >> 6, ln = threadIdx.x & 63;
    float s = 0.f;
    if (co0 + c < Cout)
      for (int k = ln; k < nslab; k += 64) s += wsb[(size_t)k * CoP + co0 + c];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (ln == 0 && co0 + c < Cout) db[co0 + c] = s;
  }
}

unsigned magic_of(int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

struct WgX3Plan { int P, VR, R, QZ, PX, Kext, nbands, nblk, MTC, CoP, CiP, vw, NSEG, CW, pack, pipe, lpr32, pk4; size_t lds, ws_floats; bool ok; };

WgX3Plan plan_x3(int N, int Cin, int Cout, int H, int W, int L = 1) {
  WgX3Plan p{};
  p.vw = (W % 4 == 0) ? 4 : (W % 2 == 0 ? 2 : 1);
  // four narrow rows per 16-lane group: opt-in (FDET_WGRAD_PACK=1) -- measured SLOWER than the
  // dword-per-lane staging on the 15x15 layers (0.64 vs 0.50 ms per 16 layers), kept for tuning
  p.pack = (W >= 4 && W <= 16 && FDET_ENV_ONCE("FDET_WGRAD_PACK") != nullptr) ? 1 : 0;
  if (p.pack) p.vw = 4;
  p.NSEG = 1; p.CW = W;
  if (W / p.vw > 16 && p.vw == 4) { p.CW = 56; p.NSEG = (W + 55) / 56; }     // column segments (14 vector lanes)
  p.P = p.NSEG > 1 ? 64 : (W + 1 + 7) / 8 * 8;
  p.VR = N * (H + 1) + 1;
  p.CoP = (Cout + 31) / 32 * 32;
  p.CiP = (Cin + 31) / 32 * 32;
  p.MTC = (p.CoP % 64 == 0) ? 2 : 1;
  const int wv = p.pack ? 4 : p.CW / p.vw;
  p.ok = wv <= 16 && p.VR < (1 << 20) && (size_t)N * std::max(Cin, Cout) * H * W < ((size_t)1 << 31);
  const int rows_total = p.VR - 1;
  const int zch = p.MTC * 2;
  const int rmul = p.pack ? 4 : 1;
  const int rz = rmul * (zreg(p.vw) / (zch * p.vw)), rx = rmul * (xreg(p.vw) / (2 * p.vw));
  int bestR = 0; double bestC = 1e30;
  for (int r = 1; r <= rz && r + 2 <= rx && r <= rows_total; ++r) {
    const int Q = r * p.P;
    const int Kext = (Q + 9 + 15) / 16 * 16;
    int QZ = Kext + 24; if (((QZ / 8) & 1) == 0) QZ += 8;
    int PX = Kext + 2 * p.P + 8; if (((PX / 8) & 1) == 0) PX += 8;
    const size_t bytes = ((size_t)p.MTC * 32 * QZ + 32 * (size_t)PX) * 4;
    if (bytes > 150 * 1024) break;
    const long nb = (long)((rows_total + r - 1) / r) * p.NSEG;
    const long slots = std::max(1, 256 / ((p.CiP / 32) * (p.CoP / (p.MTC * 32)) * L));   // workgroups along x so that the grid ~ 256
    const long blk = std::max(1L, std::min(nb, slots));
    const double mfma = (double)(((nb + blk - 1) / blk) * blk * Kext) / ((double)rows_total * p.P * p.NSEG);
    const double cost = mfma * (0.7 + 0.3 * (double)(r + 2) / r);
    if (cost <= bestC * 1.0001) { bestC = cost; bestR = r; }
  }
  if (bestR == 0) { p.ok = false; bestR = 1; }
  // pipelined kernel: 64 output channels, bands of exactly 128 positions (R = 128 / P rows = the rows one
  // register set holds), two LDS tiles.  FDET_WGRAD_PIPE=0 keeps the single-tile kernel.
  {
    const char* e = FDET_ENV_ONCE("FDET_WGRAD_PIPE");
    const int rp = 32 / (4 * p.vw);
    const bool lpr32 = p.P == 32 && p.NSEG == 1 && !p.pack && W <= 32;      // one-float lanes, 32 per row (30x30)
    p.pipe = p.ok && p.MTC == 2 && p.NSEG == 1 && !p.pack && ((p.vw != 2 && rp * p.P == 128) || lpr32) && rows_total >= 8 &&
             (size_t)N * std::max(Cin, Cout) * H * W < ((size_t)1 << 29) && !(e && e[0] == '0');   // 32-bit byte offsets
    p.lpr32 = p.pipe && lpr32;
    {
      const char* e4 = FDET_ENV_ONCE("FDET_WGRAD_PK4");
      // rows of 13..16 floats (one-float form, 16 lanes) or 29..32 floats (the 32-lane form): float4 quads instead
      p.pk4 = p.pipe && !(e4 && e4[0] == '0') && ((!p.lpr32 && p.vw == 1 && W >= 13 && W <= 16 && p.P == 16) || (p.lpr32 && W >= 29 && p.P == 32));
    }
    // the one-float 16-lane form (rows of <= 12 floats, or PK4 switched off) is not built as a pipeline any more: its
    // asm-load audit is not clean and those maps are tiny -- they take the staged kernel
    if (p.pipe && p.vw == 1 && !p.lpr32 && !p.pk4) { p.pipe = false; p.lpr32 = false; }
    if (p.pipe) bestR = p.lpr32 ? 4 : rp;      // -> Kext 144, QZ 168, PX 152 + 2 P: the kernel's compile-time geometry
  }
  p.R = bestR;
  const int Q = p.R * p.P;
  p.Kext = (Q + 9 + 15) / 16 * 16;
  p.QZ = p.Kext + 24; if (((p.QZ / 8) & 1) == 0) p.QZ += 8;
  p.PX = p.Kext + 2 * p.P + 8; if (((p.PX / 8) & 1) == 0) p.PX += 8;
  p.nbands = (rows_total + p.R - 1) / p.R;
  const long slots = std::max(1, 256 / ((p.CiP / 32) * (p.CoP / (p.MTC * 32)) * L));
  p.nblk = (int)std::max(1L, std::min((long)p.nbands * p.NSEG, slots));
  p.lds = ((size_t)p.MTC * 32 * p.QZ + 32 * (size_t)p.PX) * 4;
  p.lds = std::max(p.lds, (size_t)p.MTC * 144 * 64 * 4);
  if (p.pipe) p.lds = 2 * (((size_t)64 * p.QZ + 32 * (size_t)p.PX) * 4);
  p.ws_floats = (size_t)L * ((size_t)p.nblk * 9 * p.CoP * p.CiP + (size_t)p.nblk * p.CoP);
  return p;
}

template <int MTC>
int launch_x3(const WgX3Args& a, const WgX3Plan& p, dim3 grid, hipStream_t st) {
  int rc = FDET_OK;
  auto go = [&](auto kern) {
    rc = set_lds_attr((const void*)kern, p.lds, "conv3x3_wgrad_bf16x3");
    if (rc == FDET_OK) hipLaunchKernelGGL(kern, grid, dim3(NTHR), p.lds, st, a);
  };
  if (p.pipe && MTC == 2) {
#ifdef FDET_WG_DBG
    const char* e = FDET_ENV_ONCE("FDET_WG_DBG");
    const int dbg = e ? atoi(e) : 0;
#define WG_DBG_CASE(D) if (dbg == D && !p.lpr32 && p.vw == 4) { go(k_wgrad3x3_x3_pipe<4, D>); return rc; }
    WG_DBG_CASE(1) WG_DBG_CASE(3) WG_DBG_CASE(8) WG_DBG_CASE(9) WG_DBG_CASE(24) WG_DBG_CASE(40) WG_DBG_CASE(73)
#endif
    if (p.lpr32 && p.pk4) go(k_wgrad3x3_x3_pipe<4, 0, 32, 1>);
    else if (p.lpr32) go(k_wgrad3x3_x3_pipe<1, 0, 32>);
    else if (p.pk4) go(k_wgrad3x3_x3_pipe<4, 0, 16, 1>);
    else go(k_wgrad3x3_x3_pipe<4>);                     // p.vw == 4 (plan_x3 routes the one-float 16-lane form to the staged kernel)
  } else if (p.NSEG > 1) go(k_wgrad3x3_x3<MTC, 4, true>);
  else if (p.pack) go(k_wgrad3x3_x3<MTC, 4, false, true>);
  else if (p.vw == 4) go(k_wgrad3x3_x3<MTC, 4, false>);
  else if (p.vw == 2) go(k_wgrad3x3_x3<MTC, 2, false>);
  else go(k_wgrad3x3_x3<MTC, 1, false>);
  return rc;
}

}  // namespace

static int run_wg_x3(const float* const* xs, const float* const* dzs, float* const* dWs, float* const* dbs, int L,
                     void* ws, size_t ws_bytes, int N, int Cin, int Cout, int H, int W, hipStream_t st) {
  const WgX3Plan p = plan_x3(N, Cin, Cout, H, W, L);
  FDET_REQUIRE(p.ok, "conv3x3_wgrad_bf16x3: no tiling for N=%d H=%d W=%d (rows wider than 64 need W %% 4 == 0)", N, H, W);
  if (ws_bytes < p.ws_floats * 4)
    return fail(FDET_EWORKSPACE, "conv3x3_wgrad_bf16x3: workspace %zu < %zu bytes", ws_bytes, p.ws_floats * 4);
  WgX3Args a{};
  WgX3Red r{};
  for (int l = 0; l < L; ++l) { a.x[l] = xs[l]; a.dz[l] = dzs[l]; r.dW[l] = dWs[l]; r.db[l] = dbs[l]; }
  a.ws = (float*)ws; a.wsb = (float*)ws + (size_t)L * p.nblk * 9 * p.CoP * p.CiP;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.CoP = p.CoP; a.CiP = p.CiP; a.H = H; a.W = W; a.P = p.P; a.R = p.R;
  a.VR = p.VR; a.QZ = p.QZ; a.PX = p.PX; a.Kext = p.Kext; a.nbands = p.nbands; a.magic_h1 = magic_of(H + 1);
  a.L = L; a.ncob = p.CoP / (p.MTC * 32); a.NSEG = p.NSEG; a.CW = p.CW;
  dim3 grid(p.nblk, p.CiP / 32, a.ncob * L);
  if (int rc = p.MTC == 2 ? launch_x3<2>(a, p, grid, st) : launch_x3<1>(a, p, grid, st)) return rc;
  if (int rc = check_launch("fdet_conv3x3_wgrad_bf16x3")) return rc;
  hipLaunchKernelGGL(k_wgx3_reduce, dim3(9, (Cout + 3) / 4, L), dim3(1024), 0, st, a.ws, a.wsb, p.nblk, Cout, Cin, p.CoP,
                     p.CiP, r);
  return check_launch("fdet_conv3x3_wgrad_bf16x3(reduce)");
}

extern "C" size_t fdet_conv3x3_wgrad_bf16x3_ws_bytes(int N, int Cin, int Cout, int H, int W) {
  if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  const WgX3Plan p = plan_x3(N, Cin, Cout, H, W);
  return p.ok ? p.ws_floats * 4 : 0;
}

extern "C" size_t fdet_conv3x3_wgrad_bf16x3_batched_ws_bytes(int L, int N, int Cin, int Cout, int H, int W) {
  if (L <= 0 || L > MAXL || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  const WgX3Plan p = plan_x3(N, Cin, Cout, H, W, L);
  return p.ok ? p.ws_floats * 4 : 0;
}

extern "C" int fdet_conv3x3_wgrad_bf16x3(const float* x, const float* dz, float* dW, float* db, void* ws,
                                         size_t ws_bytes, int N, int Cin, int Cout, int H, int W, void* stream) {
  FDET_REQUIRE(x && dz && dW && db && ws, "conv3x3_wgrad_bf16x3: null pointer");
  FDET_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv3x3_wgrad_bf16x3: bad shape");
  return run_wg_x3(&x, &dz, &dW, &db, 1, ws, ws_bytes, N, Cin, Cout, H, W, (hipStream_t)stream);
}

extern "C" int fdet_conv3x3_wgrad_bf16x3_batched(const float* const* h_x, const float* const* h_dz, float* const* h_dW,
                                                 float* const* h_db, int L, void* ws, size_t ws_bytes, int N, int Cin,
                                                 int Cout, int H, int W, void* stream) {
  FDET_REQUIRE(h_x && h_dz && h_dW && h_db && ws && L >= 1 && L <= MAXL, "conv3x3_wgrad_bf16x3_batched: bad arguments (L=%d, max %d)", L, MAXL);
  for (int l = 0; l < L; ++l) FDET_REQUIRE(h_x[l] && h_dz[l] && h_dW[l] && h_db[l], "conv3x3_wgrad_bf16x3_batched: null pointer in layer %d", l);
  FDET_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv3x3_wgrad_bf16x3_batched: bad shape");
  return run_wg_x3(h_x, h_dz, h_dW, h_db, L, ws, ws_bytes, N, Cin, Cout, H, W, (hipStream_t)stream);
}
