// Weight / bias gradient of the 3x3 convs (bf16x3) on PRE-SPLIT operands (fdet_ps.h):
//
//   dW[co][ci][ky][kx] = sum_p dz[co][p] * x[ci][p + (ky-1)*WP + (kx-1)]        db[co] = sum_p dz[co][p]
//
// over the padded slot grid of the PS layout (every halo slot / row is a real zero, so the sum needs no masks and no
// K extension).  Both operands already sit in HBM as bf16 hi | lo units [slot][8 channels]; a line of 64 slots of one
// (plane, channel group) array is ONE 1-KiB LDS-DMA piece.  No staging registers, no split arithmetic, no LDS stores,
// no asm loads with register destinations (the hazard of the register-staged pipeline, fdet_wgrad3x3_x3.hip, cannot
// arise: an LDS-DMA has no register destination and is ordered by the counted vmcnt + barrier below).
//
// The MFMA K index is 16 consecutive SLOTS; the operand fragments (8 consecutive slots of one channel per lane) are read
// from the [slot][8 channel] image with ds_read_b64_tr_b16 (rows = slots, columns = channels; two reads per
// fragment).  A slot shift is a 16-byte address offset, so the kx shifts (on dz) and ky shifts (on x) are immediate
// offsets: no funnel shifts.  Per 16-slot step a wave issues 24 transposed reads and 27 MFMAs (9 taps x 3 split
// products).  One workgroup per CU = 4 waves = (32 of 64 output channels) x (32 of 64 input channels), every wave
// over ALL of K (no K split, nothing to combine in LDS); per-workgroup slabs are reduced in fixed order.
//
// Lines (64 slots) stream through LDS rings: x needs lines L-1, L, L+1 for band L (5-line ring), dz line L (3-line
// ring); the pieces issued at the head of band L (x line L+3, dz line L+2) have two whole bands to land; one barrier
// per band.  Zero rows of the layout are processed like any other (they add zeros).
#include "fdet_conv3x3_x3.h"
#include "fdet_ps.h"
#include "fdet_ldsdma.h"
#include <algorithm>

using namespace fdet;

typedef __attribute__((address_space(3))) void* lds_void_t;
typedef const __attribute__((address_space(1))) void* glb_void_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

namespace {

#ifndef WG_DBG
#define WG_DBG 0           // development builds (timing only): 1 = no DMA inside the band loop, 2 = no MFMAs, 4 = no fragment reads
#endif
constexpr int WG_MAXL = 16;
constexpr int XL = 5, ZL = 3;                 // ring depths (lines)
constexpr int XA = XL * 64 + 4;               // units per x array; == 4 (mod 16): the four arrays a 32-lane half reads fall on disjoint banks
constexpr int ZLS = 66;                       // units per dz ring line: zero guard | 64 | zero guard
constexpr int ZA = 212;                       // units per dz array (>= ZL*ZLS, == 4 mod 16)
constexpr int WG_LDS_UNITS = 16 * XA + 16 * ZA;

struct PsWgArgs {
  const bf16x8* x[WG_MAXL];      // PS image 0, per layer
  const bf16x8* dz[WG_MAXL];
  float* ws;                     // [L][nslab][9][64][64]
  float* wsb;                    // [L][nslab][2][64]
  int plane, img;                // units
  int lpi, real_lpi, nlines;     // lines per image (HP*WP/64), lines holding real rows, N*lpi
  int nslab, lpw;                // workgroups per layer, lines per workgroup
  unsigned magic_lpi;
};

// (a __builtin_bit_cast applied directly to a vector ELEMENT is miscompiled by this hipcc -- every element read element 0;
//  passing the element through a function parameter is fine)
__device__ __forceinline__ bf16x2_t as_bf16x2(unsigned u) { return __builtin_bit_cast(bf16x2_t, u); }
// transposed fragment half: 4 consecutive slots x 1 channel per lane (see the header)
__device__ __forceinline__ u32x2_t tr2(const char* base, int off) {
  return __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(base + off)));
}
// x operand of step S, tap row ky: slot 16S + (ky-1)*WP of the band lies wholly in line L-1, L or L+1 (0 / 1 / 2)
__host__ __device__ constexpr int x_line(int S, int ky, int WP) { return 16 * S + (ky - 1) * WP < 0 ? 0 : (16 * S + (ky - 1) * WP >= 64 ? 2 : 1); }
__host__ __device__ constexpr int x_off(int S, int ky, int WP) {
  return 16 * S + (ky - 1) * WP < 0 ? 16 * S + (ky - 1) * WP + 64 : (16 * S + (ky - 1) * WP >= 64 ? 16 * S + (ky - 1) * WP - 64 : 16 * S + (ky - 1) * WP);
}

// P16 (precision16, see fdet_conv3x3_ps.hip): one MFMA pass on the hi planes; only their pieces are moved.
template <int WP, bool FL1, bool P16 = false>
__global__ void __launch_bounds__(256, 1)
k_wgrad3x3_ps(const PsWgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8* const lds = reinterpret_cast<bf16x8*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = wid & 1, j = wid >> 1;                 // output-channel tile, input-channel tile
  const int layer = blockIdx.y;
  const bf16x8* __restrict__ gx = a.x[layer];
  const bf16x8* __restrict__ gz = a.dz[layer];
  {
    f32x4* z = reinterpret_cast<f32x4*>(smem);
    for (int t = tid; t < WG_LDS_UNITS; t += 256) z[t] = f32x4{0.f, 0.f, 0.f, 0.f};   // the dz guard units stay zero
  }
  // lane geometry of the transposed reads: 16-lane group g = (channel half, k half); lane 4q+p of a group supplies the
  // address of slot q, channels 4p..4p+3 of the group's 16
  const int i16 = lane & 15, g16 = (lane >> 4) & 1, kh = lane >> 5;
  const int q = i16 >> 2, pp = i16 & 3;
  const int GA = 4 * m + 2 * g16 + (pp >> 1), GB = 4 * j + 2 * g16 + (pp >> 1);
  const int lane_x = (GB * XA + 8 * kh + q) * 16 + (pp & 1) * 8;              // + ring slot * 1024 + immediate
  const int lane_z = (16 * XA + GA * ZA + 1 + 8 * kh + q) * 16 + (pp & 1) * 8; // + ring slot * ZLS*16 + immediate
  constexpr int XPL = 8 * XA * 16, ZPL = 8 * ZA * 16;                          // lo-plane byte offsets

  const int l0 = blockIdx.x * a.lpw, l1 = min(l0 + a.lpw, a.nlines);
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  const unsigned bsel0 = j == 0 ? 0x3f803f80u : 0u, bsel1 = j == 1 ? 0x3f803f80u : 0u;   // bf16 (1, 1) on this wave's steps

  // DMA (fdet_ldsdma.h) of global line LG of array index AI (plane * 8 + group) of a tensor (descriptor RS, based at
  // the guard image in front of image 0) into LDS unit DST: the source is wave-uniform + lane * 16.  Lines outside the
  // batch fall into the zero guard images.
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t)smem;
  const unsigned tbytes = (unsigned)((a.nlines / a.lpi) + 2) * (unsigned)a.img * 16u;
  const dma_u32x4 xrs = dma_rsrc(gx - a.img, tbytes), zrs = dma_rsrc(gz - a.img, tbytes);
  const unsigned lane16 = (unsigned)lane * 16u;
#define WG_PIECE(RS, LG, AI, DST)                                                                  \
  {                                                                                                \
    const int lg_ = (LG) + a.lpi;                                                                  \
    const int n_ = (int)__umulhi((unsigned)lg_, a.magic_lpi);                                      \
    const unsigned so_ = (unsigned)(n_ * a.img + ((AI) >> 3) * a.plane + ((AI) & 7) * (a.lpi * 64) + (lg_ - n_ * a.lpi) * 64) * 16u; \
    dma_piece(lds0 + (unsigned)(DST) * 16u, lane16, RS, so_);                                      \
  }
  // wave w moves arrays 4w .. 4w+3 of both tensors: x line LX into ring slot SX, dz line LZ into ring slot SZ
#define WG_ISSUE(LX, SX, LZ, SZ)                                                                   \
  if (!(P16 && wid >= 2)) {                                  /* P16: arrays 8..15 are the lo planes */ \
    _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) {                                             \
      const int ai_ = 4 * wid + k_;                                                                \
      WG_PIECE(xrs, LX, ai_, ai_ * XA + (SX) * 64)                                                  \
      WG_PIECE(zrs, LZ, ai_, 16 * XA + ai_ * ZA + (SZ) * ZLS + 1)                                   \
    }                                                                                              \
  }
  __syncthreads();                                       // zero fill done before the first DMA lands
  // prologue: x lines l0-1 .. l0+2 (ring slots follow the line number mod XL), dz lines l0, l0+1
  int sx = ((l0 - 1) % XL + XL) % XL, sz = l0 % ZL;      // ring slots of x line L-1 and dz line L
#define XS(D) ((sx + (D)) % XL)
  WG_ISSUE(l0 - 1, XS(0), l0, sz)
  WG_ISSUE(l0, XS(1), l0 + 1, (sz + 1) % ZL)
  if (!(P16 && wid >= 2)) {
    _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) {
      const int ai_ = 4 * wid + k_;
      WG_PIECE(xrs, l0 + 1, ai_, ai_ * XA + XS(2) * 64)
      WG_PIECE(xrs, l0 + 2, ai_, ai_ * XA + XS(3) * 64)
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  u32x4_t zf[2][3][2], xf[2][3][2];                     // [set][kx | ky][plane]: 8 slots of this lane's channel
  // fragments of step S of the band: dz slots 16S + 8kh + 4h + q - (kx - 1); x slots of line x_line, offset x_off
#define WG_FRAGS(F, S, ZB, XB0, XB1, XB2)                                                          \
  {                                                                                                \
    _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_)                                               \
      _Pragma("unroll") for (int pl_ = 0; pl_ < (P16 ? 1 : 2); ++pl_)                              \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                         \
          const u32x2_t z_ = (WG_DBG & 4) ? u32x2_t{0x3f803f80u + (unsigned)lane, 0x40004000u} : tr2(ZB, pl_ * ZPL + (16 * (S) + 4 * h_ - (k_ - 1)) * 16); \
          zf[F][k_][pl_][2 * h_] = z_[0]; zf[F][k_][pl_][2 * h_ + 1] = z_[1];                      \
          const char* xb_ = x_line((S), k_, WP) == 0 ? XB0 : (x_line((S), k_, WP) == 1 ? XB1 : XB2); \
          const u32x2_t x_ = (WG_DBG & 4) ? u32x2_t{0x3f803f80u, 0x40004000u + (unsigned)lane} : tr2(xb_, pl_ * XPL + (x_off((S), k_, WP) + 4 * h_) * 16); \
          xf[F][k_][pl_][2 * h_] = x_[0]; xf[F][k_][pl_][2 * h_ + 1] = x_[1];                      \
        }                                                                                          \
  }
  int band = l0;
  if (FL1) {
    // one-band flight: everything issued so far is visible after this barrier; the fragments of a band's first step are
    // read during the last step of the band before (no fragment-latency bubble behind the band barrier)
    __builtin_amdgcn_s_barrier();
    const char* zb_ = smem + lane_z + sz * (ZLS * 16);
    const char* xb0 = smem + lane_x + XS(0) * 1024;
    const char* xb1 = smem + lane_x + XS(1) * 1024;
    const char* xb2 = smem + lane_x + XS(2) * 1024;
    WG_FRAGS(0, 0, zb_, xb0, xb1, xb2)
  }
  // Lines that hold only the zero rows under an image (2 of 62 at 60x60, 1 of 16 at 30x30) contribute nothing: they are
  // walked in a loop of their OWN that rotates the rings and issues the DMA but reads no fragment and runs no MFMA (a
  // branch around the MFMAs inside one loop made hipcc copy the accumulators every band), then the first step's fragments
  // of the next real band are read again.
  int row = l0 - (l0 / a.lpi) * a.lpi;
  while (band < l1) {
  const int n_full = row < a.real_lpi ? min(l1 - band, a.real_lpi - row) : 0;
  row += n_full;
  for (const int bend = band + n_full; band < bend; ++band) {
    // FL1: every piece issued up to the head of the previous band has landed (x <= band+2, dz <= band+1)
    // else: lines issued two bands ago (x band+1, dz band) have landed; the 8 pieces of the previous band may still fly
    if (FL1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ring slots freed by band-1: x line band-2 -> x line band+3 ; dz line band-1 -> dz line band+2
    if (!(WG_DBG & 1)) WG_ISSUE(band + 3, XS(4), band + 2, (sz + 2) % ZL)
    __builtin_amdgcn_sched_barrier(0);
    {                                                    // (zero rows of the layout run too: 2 of HP rows, and a branch
                                                         //  around the MFMAs made hipcc copy the accumulators every band)
      const char* zb_ = smem + lane_z + sz * (ZLS * 16);
      const char* xb0 = smem + lane_x + XS(0) * 1024;
      const char* xb1 = smem + lane_x + XS(1) * 1024;
      const char* xb2 = smem + lane_x + XS(2) * 1024;
      const char* zbn = smem + lane_z + ((sz + 1) % ZL) * (ZLS * 16);   // the next band's lines
      const char* xb3 = smem + lane_x + XS(3) * 1024;
      if (!FL1) {
        WG_FRAGS(0, 0, zb_, xb0, xb1, xb2)
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int F = s & 1;
        if (s < 3) WG_FRAGS(F ^ 1, s + 1, zb_, xb0, xb1, xb2)
        else if (FL1) WG_FRAGS(0, 0, zbn, xb1, xb2, xb3)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int t = ky * 3 + kx;
            const bf16x8 ah = __builtin_bit_cast(bf16x8, zf[F][kx][0]), al = __builtin_bit_cast(bf16x8, zf[F][kx][1]);
            const bf16x8 bh = __builtin_bit_cast(bf16x8, xf[F][ky][0]), bl = __builtin_bit_cast(bf16x8, xf[F][ky][1]);
            if (!(WG_DBG & 2)) {
              if (!P16) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
              }
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
            }
          }
        // bias partial of this lane's channel (the un-shifted dz fragment) by v_dot2 with a ones / zeros pair: the two waves
        // that hold the same dz fragments take alternate steps (branch-free: a branch here would cut the scheduling region)
        {
          const bf16x2_t sel = as_bf16x2((s & 1) ? bsel1 : bsel0);
#pragma unroll
          for (int pl = 0; pl < (P16 ? 1 : 2); ++pl) {
            const u32x4_t zv = zf[F][1][pl];
            bsum = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(zv[0]), sel, bsum, false);
            bsum = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(zv[1]), sel, bsum, false);
            bsum = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(zv[2]), sel, bsum, false);
            bsum = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(zv[3]), sel, bsum, false);
          }
        }
        // the next step's 24 transposed reads ride one per MFMA
        if ((s < 3 || FL1) && !P16) {
#pragma unroll
          for (int i = 0; i < 24; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        } else if (s < 3 || FL1) {                           // P16: 9 MFMAs, 12 transposed reads
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          }
        }
      }
    }
    sx = (sx + 1) % XL;
    sz = (sz + 1) % ZL;
  }
  if (band < l1 && row >= a.real_lpi) {
    const int n_zero = min(l1 - band, a.lpi - row);
    for (const int bend = band + n_zero; band < bend; ++band) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (!(WG_DBG & 1)) WG_ISSUE(band + 3, XS(4), band + 2, (sz + 2) % ZL)
      sx = (sx + 1) % XL;
      sz = (sz + 1) % ZL;
    }
    row += n_zero;
    if (row >= a.lpi) row = 0;
    if (FL1 && band < l1) {                               // the rings hold x <= band+1, dz <= band (the last wait above)
      const char* zb_ = smem + lane_z + sz * (ZLS * 16);
      const char* xb0 = smem + lane_x + XS(0) * 1024;
      const char* xb1 = smem + lane_x + XS(1) * 1024;
      const char* xb2 = smem + lane_x + XS(2) * 1024;
      WG_FRAGS(0, 0, zb_, xb0, xb1, xb2)
    }
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the run-ahead pieces target this workgroup's LDS
#undef WG_FRAGS
#undef XS
#undef WG_ISSUE
#undef WG_PIECE
  // ---- slab of this workgroup: [9][64][64]; bias partials [2][64]
  const int l31 = lane & 31, half = lane >> 5;
  const size_t s = (size_t)layer * a.nslab + blockIdx.x;
  float* __restrict__ wsl = a.ws + s * 9 * 64 * 64;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
      wsl[(t * 64 + co) * 64 + 32 * j + l31] = acc[t][r];
    }
  bsum += __shfl_xor(bsum, 32, 64);                       // the two k halves of a channel
  if (half == 0) a.wsb[(s * 2 + j) * 64 + 32 * m + l31] = bsum;
}

// fixed-order reduction of the slabs: dW [64][64][3][3] (OIHW), db [64].  A block sums 64 consecutive slab elements:
// thread (e, q) adds the slabs k = q, q+4, q+8 ... (eight loads in flight), the four partial sums are combined in LDS in
// the order q = 0..3 -- the same association on every run.
struct PsWgRed { float* dW[WG_MAXL]; float* db[WG_MAXL]; };
__global__ void __launch_bounds__(256)
k_wgrad_ps_reduce(const float* __restrict__ ws, const float* __restrict__ wsb, int nslab, const PsWgRed out) {
  __shared__ float part[256];
  const int layer = blockIdx.y;
  const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
  const bool bias_blk = blockIdx.x == 9 * 64;                   // the last block: the 2 x 64 bias partials per slab
  const size_t stride = bias_blk ? 128 : 9 * 4096;
  const float* __restrict__ src = bias_blk ? wsb + (size_t)layer * nslab * 128 + el
                                           : ws + (size_t)layer * nslab * 9 * 4096 + (size_t)blockIdx.x * 64 + el;
  float s = 0.f;
  int k = q;
  for (; k + 28 < nslab; k += 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + 4 * u) * stride] + (bias_blk ? src[(size_t)(k + 4 * u) * stride + 64] : 0.f);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < nslab; k += 4) s += src[(size_t)k * stride] + (bias_blk ? src[(size_t)k * stride + 64] : 0.f);
  part[threadIdx.x] = s;
  __syncthreads();
  if (q == 0) {
    const float r = ((part[el] + part[64 + el]) + part[128 + el]) + part[192 + el];
    if (bias_blk) {
      out.db[layer][el] = r;
    } else {
      const int e = blockIdx.x * 64 + el;
      const int t = e >> 12, co = (e >> 6) & 63, ci = e & 63;
      out.dW[layer][(co * 64 + ci) * 9 + t] = r;
    }
  }
}

int wg_num_cus() {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return ncu;
}

int wg_plan(int L, int N, int H, int W, PsGeo& g, int& nslab, int& lpw) {
  PsStrips sp;                                           // wide maps: column strips (fdet_ps.h); g.N counts strip-images
  if (!ps_geo_strips(N, 64, H, W, g, sp) || L < 1 || L > WG_MAXL) return 0;
  const int lpi = g.HP * g.WP / 64;
  const long nlines = (long)g.N * lpi;
  if (nlines + 2 * lpi >= (1 << 20) || (size_t)(g.N + 2) * g.img * 16 >= ((size_t)1 << 32)) return 0;   // 32-bit DMA byte offsets
  nslab = std::max(1, std::min((int)nlines, wg_num_cus() / L));
  lpw = (int)((nlines + nslab - 1) / nslab);
  nslab = (int)((nlines + lpw - 1) / lpw);
  return 1;
}

}  // namespace

extern "C" size_t fdet_conv3x3_wgrad_ps_ws_bytes(int L, int N, int C, int H, int W) {
  PsGeo g;
  int nslab = 0, lpw = 0;
  if (C != 64 || !wg_plan(L, N, H, W, g, nslab, lpw)) return 0;
  return (size_t)L * nslab * (9 * 4096 + 128) * sizeof(float);
}

// dW[l] [64,64,3,3], db[l] [64] of L same-shape 64-channel layers from PS tensors x[l], dz[l] (image-0 pointers; host
// arrays of device pointers)
namespace {
int wgrad_ps_run(const void* const* h_x, const void* const* h_dz, float* const* h_dW, float* const* h_db, int L, int N, int C,
                 int H, int W, void* ws, size_t ws_bytes, void* stream, bool p16) {
  PsGeo g;
  int nslab = 0, lpw = 0;
  FDET_REQUIRE(h_x && h_dz && h_dW && h_db && ws, "conv3x3_wgrad_ps: null pointer");
  FDET_REQUIRE(C == 64 && wg_plan(L, N, H, W, g, nslab, lpw), "conv3x3_wgrad_ps: unsupported shape (L=%d N=%d C=%d H=%d W=%d)", L, N, C, H, W);
  const size_t need = (size_t)L * nslab * (9 * 4096 + 128) * sizeof(float);
  FDET_REQUIRE(ws_bytes >= need, "conv3x3_wgrad_ps: workspace of %zu bytes needed, %zu given", need, ws_bytes);
  PsWgArgs a;
  PsWgRed red;
  for (int l = 0; l < L; ++l) {
    FDET_REQUIRE(h_x[l] && h_dz[l] && h_dW[l] && h_db[l], "conv3x3_wgrad_ps: null pointer in layer %d", l);
    a.x[l] = reinterpret_cast<const bf16x8*>(h_x[l]);
    a.dz[l] = reinterpret_cast<const bf16x8*>(h_dz[l]);
    red.dW[l] = h_dW[l];
    red.db[l] = h_db[l];
  }
  a.ws = reinterpret_cast<float*>(ws);
  a.wsb = a.ws + (size_t)L * nslab * 9 * 4096;
  a.plane = g.plane; a.img = g.img;
  a.lpi = g.HP * g.WP / 64;
  a.real_lpi = (H * g.WP + 63) / 64;
  a.nlines = g.N * a.lpi;                                // (strips: the dz halo slots must be ZERO, fdet_ps_halo_exchange(zero_only))
  a.nslab = nslab; a.lpw = lpw;
  a.magic_lpi = magic_of(a.lpi);
  const size_t lds = (size_t)WG_LDS_UNITS * 16;
  hipStream_t st = (hipStream_t)stream;
  static const bool fl1 = [] { const char* e = getenv("FDET_WGPS_FLIGHT"); return !(e && e[0] == '2'); }();   // development: 2 = two-band flight
  auto go = [&](auto kern, bool& done) -> int {
    if (!done) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return fail(FDET_ELAUNCH, "conv3x3_wgrad_ps: cannot reserve %zu bytes of LDS", lds);
      }
      done = true;
    }
    hipLaunchKernelGGL(kern, dim3(nslab, L), dim3(256), lds, st, a);
    return FDET_OK;
  };
  static bool d64a = false, d64b = false, d32a = false, d32b = false, d16a = false, d16b = false, d64p = false, d32p = false, d16p = false;
  int rc0;
  if (p16) {
    if (g.WP == 64) rc0 = go(k_wgrad3x3_ps<64, true, true>, d64p);
    else if (g.WP == 16) rc0 = go(k_wgrad3x3_ps<16, true, true>, d16p);
    else rc0 = go(k_wgrad3x3_ps<32, true, true>, d32p);
  }
  else if (g.WP == 64) rc0 = fl1 ? go(k_wgrad3x3_ps<64, true>, d64a) : go(k_wgrad3x3_ps<64, false>, d64b);
  else if (g.WP == 16) rc0 = fl1 ? go(k_wgrad3x3_ps<16, true>, d16a) : go(k_wgrad3x3_ps<16, false>, d16b);
  else rc0 = fl1 ? go(k_wgrad3x3_ps<32, true>, d32a) : go(k_wgrad3x3_ps<32, false>, d32b);
  if (rc0 != FDET_OK) return rc0;
  int rc = check_launch("fdet_conv3x3_wgrad_ps_batched");
  if (rc != FDET_OK) return rc;
  hipLaunchKernelGGL(k_wgrad_ps_reduce, dim3(9 * 64 + 1, L), dim3(256), 0, st, a.ws, a.wsb, nslab, red);
  return check_launch("fdet_conv3x3_wgrad_ps_batched(reduce)");
}
}  // namespace

extern "C" int fdet_conv3x3_wgrad_ps_batched(const void* const* h_x, const void* const* h_dz, float* const* h_dW,
                                             float* const* h_db, int L, int N, int C, int H, int W, void* ws,
                                             size_t ws_bytes, void* stream) {
  return wgrad_ps_run(h_x, h_dz, h_dW, h_db, L, N, C, H, W, ws, ws_bytes, stream, false);
}

// precision16: one bf16 MFMA pass on the hi planes of x / dz (fp32 accumulate, fp32 dW / db)
extern "C" int fdet_conv3x3_wgrad_ps_batched_p16(const void* const* h_x, const void* const* h_dz, float* const* h_dW,
                                                 float* const* h_db, int L, int N, int C, int H, int W, void* ws,
                                                 size_t ws_bytes, void* stream) {
  return wgrad_ps_run(h_x, h_dz, h_dW, h_db, L, N, C, H, W, ws, ws_bytes, stream, true);
}
