// bf16x3 3x3 conv (forward / data gradient) on PRE-SPLIT activations (fdet_ps.h): the operands already sit in HBM
// as bf16 hi | lo units in the layout the MFMA fragments have in LDS, so staging is LDS-DMA only -- no split
// arithmetic, no LDS stores, no staging registers -- and the epilogue writes the same format for the next consumer.
//
//   * one workgroup (4 waves, one per SIMD, the whole register file) per CU, persistent over tiles of 512 positions
//     x 64 output channels; a wave owns a 2 x 4 grid of 32x32 accumulator tiles (128 positions x 64 channels);
//   * K is walked in chunks of 16 input channels; TWO LDS buffers (weights 36 KB + activations 40.5 KB each): the
//     DMA of chunk s+1 is issued right after the barrier that opens chunk s and has the whole chunk (216 MFMAs per
//     wave) to land; ONE barrier per chunk;
//   * operand fragments are double-buffered in registers: the 12 ds_read_b128 of tap t+1 issue among the MFMAs of tap t;
//   * tiles are bands of R "virtual rows" (fdet_ps.h: v = n*HP + y, zero rows between images), so bands run across
//     images, every image's halo rows are real zero rows, and row pairs stay pool-aligned.
// Arithmetic is that of fdet_conv3x3_x3*.hip (a_hi*b_lo + a_lo*b_hi + a_hi*b_hi, fp32 accumulate, the same K order
// within a chunk), results agree with those kernels to the rounding of the PS format (16 significant bits).
//
// P16 instantiations (round 4, the `precision16` mode = the arithmetic of the reference's Trainer(precision=16),
// train_model.py:50, with bf16 as the 16-bit type): ONE MFMA pass on the hi planes (bf16 activations x bf16 weights, fp32
// accumulate), only the hi planes are moved by the DMA and only the hi plane of the output is written -- a third of the
// MFMAs, half the bytes.  The lo planes of a P16 engine's buffers are never written (they stay zero).
#include "fdet_conv3x3_x3.h"
#include "fdet_ps.h"
#include "fdet_ldsdma.h"
#include <algorithm>

using namespace fdet;

typedef __attribute__((address_space(3))) void* lds_void_t;
typedef const __attribute__((address_space(1))) void* glb_void_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

#ifndef PS_DBG
#define PS_DBG 0          // development builds (-DPS_DBG=n, timing only): 1 = no epilogue, 2 = no MFMAs, 4 = no DMA, 8 = every activation DMA from image 0 (L2-resident), 16 = no weight DMA
#endif
constexpr int PSA = 9 * 2 * 64;      // weight units per plane (hi / lo) and 16-channel chunk: [tap][k half][64 co]
constexpr int PS_ROWI = 10;          // DMA wave-instructions per activation array and chunk: (R + 2) * WP / 64 <= 10

enum { PSE_FWD_FULL = 0, PSE_DGRAD_ACT = 1, PSE_FWD_POOL = 2, PSE_DGRAD_ADDPOOL = 3 };

struct PsConvArgs {
  const bf16x8* x;           // PS input, image 0
  const bf16x8* a_hi;        // weight panels [Cin/16][9][2][64] x 8 bf16
  const bf16x8* a_lo;
  const float* bias;         // forward modes
  bf16x8* y;                 // PS output, image 0
  const bf16x8* aux;         // PSE_DGRAD_ACT: PS activation whose sign selects the LeakyReLU slope; PSE_FWD_POOL: the skip tensor (PS)
  // pooled-block modes (a lane owns one whole 2x2 window, see POOLM in the kernel)
  const float* scale;        // FWD_POOL: [N,64] dropout scale or null
  bf16x8* pool_ps;           // FWD_POOL: pooled output as PS (image 0) or null
  float* pool_f32;           // FWD_POOL: pooled output as fp32 NCHW or null
  unsigned char* route_out;  // FWD_POOL: routing bytes [N][8][Hp][Wp][8] or null (eval)
  const float* dout;         // DGRAD_ADDPOOL: gradient of the pooled block output, fp32 NCHW
  const unsigned char* route_in;
  float* dx_f32;             // DGRAD_ADDPOOL: dx, fp32 NCHW
  int Hp, Wp, HPp, WPp, plane_p, img_p;
  // column strips (fdet_ps.h): N counts strip-images (index = s * Nimg + n); the last strip (index >= last0) holds Wlast
  // columns; the fp32 NCHW tensors of the pooled modes are addressed by image with rows of Wf (pooled: Wpf) columns, strip s
  // starting at column s * xo.  Without strips: Nimg = N, last0 = 0, Wlast = W, Wf = W, Wpf = Wp.
  int Nimg, last0, Wlast, Wf, Wpf, xo;
  unsigned magic_nimg;
  int N, H, W, HP;
  int nch, ntiles;
  int plane_i, img_i, plane_o, img_o;
  unsigned magic_hp;
  float slope;
};

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// timing builds: keeps an accumulator tile (and the MFMAs that feed it) alive without storing it
__device__ __forceinline__ void ps_keep(const f32x16& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::"v"(v));
#endif
}

// one accumulator element -> VGPR (explicit accumulation-register read, see ps_unit)
__device__ __forceinline__ void ps_acc_read(const f32x16& acc, const int r, float& z) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(z) : "a"(acc[r]));
#else
  z = acc[r];
#endif
}

// One epilogue unit: accumulator tile (m, n), group pair gp (channels 32m + 16gp .. + 15 at the lane's position).
// lane = position, registers = channels 32m + 8g + 4half + i; a v_permlane32_swap pair leaves every lane with all 8
// channels of ONE 16-byte unit (lanes 0-31: group 2gp, lanes 32-63: group 2gp+1), so a unit is two 16-byte buffer
// stores (hi, lo), 512 contiguous bytes per wave-instruction.  The validity of the position is folded into the offset
// (out of range = dropped by the hardware): every wave issues exactly two stores per unit, which the counted waits
// of the main loop rely on.
template <int MODE, bool P16>
__device__ __forceinline__ void ps_unit(const f32x16& acc, const int gp, const float (&bza)[4], const float (&bzb)[4],
                                        const u32x2_t (&sg)[2], const float slope, const bool ok, const int ob, const int m,
                                        const int half, const int gstride, const int plane_o_bytes,
                                        const __amdgpu_buffer_rsrc_t yrs) {
  float za[4], zb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // explicit accumulation-register reads: the finished tile stays in AGPRs until the unit that consumes it (left to
    // itself hipcc moves the whole previous tile into 128 VGPRs at the tile boundary and spills around it)
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass silently drops a kernel whose body holds device-only asm constraints)
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(za[i]) : "a"(acc[8 * gp + i]));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(zb[i]) : "a"(acc[8 * gp + 4 + i]));
#else
    za[i] = acc[8 * gp + i]; zb[i] = acc[8 * gp + 4 + i];
#endif
  }
  if (MODE == PSE_FWD_FULL) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float wa = za[i] + bza[i], wb = zb[i] + bzb[i];
      za[i] = fmaxf(wa, wa * slope);                     // == w > 0 ? w : w*slope for 0 <= slope <= 1 (NaN stays NaN)
      zb[i] = fmaxf(wb, wb * slope);
    }
  } else if (MODE == PSE_DGRAD_ACT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // the hi part carries the sign of the saved activation (it is zero only for a zero): its lo plane is not read
      za[i] *= ps_join(sg[0][i >> 1], 0u, i & 1) > 0.f ? 1.f : slope;
      zb[i] *= ps_join(sg[1][i >> 1], 0u, i & 1) > 0.f ? 1.f : slope;
    }
  }
  unsigned ha[2], la[2], hb[2], lb[2];
  if (P16) {
    ps_hi4(za, ha);
    ps_hi4(zb, hb);
  } else {
    ps_split4(za, ha, la);
    ps_split4(zb, hb, lb);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    auto r1 = __builtin_amdgcn_permlane32_swap(ha[k], hb[k], false, false);
    ha[k] = r1[0]; hb[k] = r1[1];
    if (!P16) {
      auto r2 = __builtin_amdgcn_permlane32_swap(la[k], lb[k], false, false);
      la[k] = r2[0]; lb[k] = r2[1];
    }
  }
  const int G = 4 * m + 2 * gp + half;
  const unsigned off = ok ? (unsigned)(ob + G * gstride) * 16u : 0x80000000u;
  __builtin_amdgcn_raw_buffer_store_b128(u32x4{ha[0], ha[1], hb[0], hb[1]}, yrs, off, 0, 0);
  if (!P16) __builtin_amdgcn_raw_buffer_store_b128(u32x4{la[0], la[1], lb[0], lb[1]}, yrs, off, plane_o_bytes, 0);
}

// WOVEN (forward): the epilogue of tile i runs as 16 "units" spread over the first two chunks of tile i+1, one unit
// behind each of taps 0..7, its ~65 VALU instructions and two stores placed two or three per MFMA gap by
// sched_group_barrier; two accumulator sets alternate.  The other modes keep the epilogue at the end of the tile.
template <int MODE, int WP, bool P16>
__global__ void __launch_bounds__(256, 1)
k_conv3x3_ps(const PsConvArgs p) {
  constexpr bool WOVEN = MODE == PSE_FWD_FULL || MODE == PSE_DGRAD_ACT || MODE == PSE_DGRAD_ADDPOOL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8* const lds = reinterpret_cast<bf16x8*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  constexpr int WPL = WP == 64 ? 6 : 5, R = 512 / WP, PT = (R + 2) * WP + 8, NBI = (R + 2) * WP / 64;
  constexpr int buf_units = 2 * PSA + 4 * PT;
  const int HP = p.HP;
  // the wave's four 32-position blocks: n & 1 = row of a pair, n >> 1 = column half (WP 64) / second pair (WP 32)
  int qn[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) qn[n] = WP == 64 ? (2 * wid + (n & 1)) * 64 + (n >> 1) * 32 : (4 * wid + n) * 32;
  // POOLM (pooled-block modes): a lane's four blocks are the four elements of ONE 2x2 pooling window, in scan order
  // n = 2*row + column -- block n is (row 2w' + (n >> 1), column 2*lane + (n & 1)) -- so max / argmax / routing byte and
  // the un-pooling need no cross-lane traffic.  The MFMA does not care which position a column is; the price is a
  // 32-byte lane stride of the activation fragment reads (2-way LDS bank conflict on 8 of a tap's 12 reads).
  constexpr bool POOLM = MODE == PSE_FWD_POOL || MODE == PSE_DGRAD_ADDPOOL;
  if (POOLM) {
#pragma unroll
    for (int n = 0; n < 4; ++n) qn[n] = (WP == 64 ? (2 * wid + (n >> 1)) * 64 : (4 * wid + (n >> 1)) * 32) + (n & 1);
  }
  const int lb = POOLM ? (WP == 64 ? 2 * l31 : 2 * l31 + 32 * (l31 >> 4)) : l31;
  // XCD-aware persistent walk (as fdet_conv3x3_x3_sb.hip): workgroups with equal blockIdx % 8 share an L2 and take
  // one contiguous eighth of the bands, so the halo rows two neighbouring bands share are L2 hits
  int tile = blockIdx.x, tend = p.ntiles, tstep = gridDim.x;
  if ((gridDim.x & 7) == 0) {
    const int grp = blockIdx.x & 7, chunk = (p.ntiles + 7) >> 3;
    tile = grp * chunk + (blockIdx.x >> 3);
    tend = min((grp + 1) * chunk, p.ntiles);
    tstep = gridDim.x >> 3;
    if (tile >= tend) tile = tend = p.ntiles;
  }
  if (tile >= tend) return;

  // per-lane source byte offsets (relative to the guard image in front of image 0, without the array base) of the
  // activation DMA pieces of a tile
  unsigned ro[PS_ROWI];
#define PS_ROWOFF(T)                                                                               \
  {                                                                                                \
    const int v0_ = (T) * R;                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < PS_ROWI; ++i_) {                                       \
      const int u_ = i_ * 64 + lane;                                                               \
      const int vv_ = v0_ - 1 + (u_ >> WPL) + HP;                                                  \
      const int nn_ = (int)__umulhi((unsigned)vv_, p.magic_hp);                                    \
      ro[i_] = (unsigned)(((PS_DBG & 8) ? 1 : nn_) * p.img_i + (vv_ - nn_ * HP) * WP + (u_ & (WP - 1))) * 16u; \
    }                                                                                              \
  }
  // DMA of chunk C of the tile whose offsets are in ro[] into buffer BS (fdet_ldsdma.h): wave w moves weight plane
  // w >> 1, half w & 1 (nine 1-KiB pieces) and activation array w = (plane, k half) (NBI pieces)
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t)smem;
  const dma_u32x4 wrs = dma_rsrc((wid >> 1) ? p.a_lo : p.a_hi, (unsigned)p.nch * PSA * 16u);
  const dma_u32x4 xrs = dma_rsrc(p.x - p.img_i, (unsigned)(p.N + 2) * (unsigned)p.img_i * 16u);
  const unsigned wvoff = (unsigned)((wid & 1) * 576 + lane) * 16u;
#define PS_DMA(C, BS)                                                                              \
  if (!(PS_DBG & 4) && !(P16 && wid >= 2)) {                 /* P16: the lo-plane waves move nothing */ \
    const unsigned db_ = lds0 + (unsigned)(BS) * (buf_units * 16);                                 \
    const unsigned ad_ = db_ + (unsigned)((wid >> 1) * PSA + (wid & 1) * 576) * 16u;               \
    _Pragma("unroll") for (int k_ = 0; k_ < ((PS_DBG & 16) ? 0 : 9); ++k_)                          \
      dma_piece(ad_ + k_ * 1024, wvoff, wrs, (unsigned)(C) * (PSA * 16) + k_ * 1024);              \
    const unsigned bso_ = (unsigned)((wid >> 1) * p.plane_i + (2 * (C) + (wid & 1)) * HP * WP) * 16u; \
    const unsigned bd_ = db_ + (unsigned)(2 * PSA + wid * PT) * 16u;                               \
    _Pragma("unroll") for (int i_ = 0; i_ < NBI; ++i_) dma_piece(bd_ + i_ * 1024, ro[i_], xrs, bso_); \
  }

  f32x16 accA[2][4], accB[WOVEN ? 2 : 1][WOVEN ? 4 : 1];
  if (WOVEN) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[WOVEN ? m : 0][WOVEN ? n : 0][r] = 0.f;   // the first tile's "previous tile": nothing is stored from it
  }

  // this lane's bias values: channel 32m + 8g + 4half + i
  float bz[2][4][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) bz[m][g][i] = (MODE == PSE_FWD_FULL || MODE == PSE_FWD_POOL) ? p.bias[32 * m + 8 * g + 4 * half + i] : 0.f;

  const int a_off = half * 64 + l31;                     // + tap*128 + m*32 ; lo plane: + PSA
  const int b_off = 2 * PSA + half * PT + lb;            // + qn[n] + tap offset ; lo planes: + 2*PT
  const int gstride = HP * WP, plane_o_bytes = p.plane_o * 16;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.N * p.img_o * 16), 0x00020000);

  // output geometry of a tile: unit index of (group 0, this lane's position) per n block, and its validity
  int ob[4];
  bool okn[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) { ob[n] = 0; okn[n] = false; }
#define PS_GEO(T)                                                                                  \
  {                                                                                                \
    _Pragma("unroll") for (int n_ = 0; n_ < 4; ++n_) {                                             \
      const int q_ = qn[n_] + l31;                                                                 \
      const int col_ = q_ & (WP - 1);                                                              \
      const int v_ = (T) * R + (q_ >> WPL);                                                        \
      const int nn_ = (int)__umulhi((unsigned)v_, p.magic_hp), y_ = v_ - nn_ * HP;                 \
      okn[n_] = nn_ < p.N && y_ < p.H && col_ < (nn_ >= p.last0 ? p.Wlast : p.W);                  \
      ob[n_] = nn_ * p.img_o + y_ * WP + col_ + 1;                                                 \
    }                                                                                              \
  }
  // DGRAD_ACT, woven: hi pieces of the saved activation at the PREVIOUS tile's positions (groups 4m+2gp, 4m+2gp+1),
  // loaded when that tile's geometry is known and consumed by its units behind the next tile's first two chunks
  u32x2_t sgp[4][2][2][2];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) sgp[n][m][gp][0] = sgp[n][m][gp][1] = u32x2_t{0u, 0u};
#define PS_SIGNS()                                                                                 \
  {                                                                                                \
    _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                                \
      const u32x2_t* ap = reinterpret_cast<const u32x2_t*>(p.aux + (okn[n] ? ob[n] : 0)) + half;   \
      _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                \
        _Pragma("unroll") for (int gp = 0; gp < 2; ++gp) {                                         \
          const size_t ga = (size_t)(4 * m + 2 * gp) * gstride, gb = ga + (size_t)gstride;         \
          sgp[n][m][gp][0] = ap[ga * 2];                                                           \
          sgp[n][m][gp][1] = ap[gb * 2];                                                           \
        }                                                                                          \
    }                                                                                              \
  }

  int young = 0;                                         // stores of this wave younger than its last DMA: 0, 16 or 32
  int sb = 0;
  // a chunk opens when its DMA (issued a whole chunk ago; only `young` stores are younger) has landed in every wave
#define PS_WAITN(K) else if (young == K) asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory");
#define PS_OPEN()                                                                                  \
  {                                                                                                \
    if (young == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                               \
    PS_WAITN(4) PS_WAITN(8) PS_WAITN(12) PS_WAITN(16) PS_WAITN(32) PS_WAITN(36) PS_WAITN(40) PS_WAITN(44) PS_WAITN(48) PS_WAITN(63) \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
    __builtin_amdgcn_s_barrier();                                                                  \
  }
#define PS_NEXT(C)                                                                                 \
  {                                                                                                \
    if ((C) + 1 < p.nch) {                                                                         \
      PS_DMA((C) + 1, sb ^ 1)                                                                      \
    } else if (tile + tstep < tend) {                                                              \
      PS_ROWOFF(tile + tstep)                                                                      \
      PS_DMA(0, sb ^ 1)                                                                            \
    }                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  }
  bf16x8 ah[2][2], al[2][2], bh[2][4], bl[2][4];
#define PS_FRAGS(F, T)                                                                             \
  {                                                                                                \
    const int to_ = ((T) / 3) * WP + (T) % 3;                                                      \
    _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_) {                                             \
      ah[F][m_] = Aw[(T) * 128 + m_ * 32];                                                         \
      if (!P16) al[F][m_] = Aw[PSA + (T) * 128 + m_ * 32];                                         \
    }                                                                                              \
    _Pragma("unroll") for (int n_ = 0; n_ < 4; ++n_) {                                             \
      bh[F][n_] = Bw[qn[n_] + to_];                                                                \
      if (!P16) bl[F][n_] = Bw[2 * PT + qn[n_] + to_];                                             \
    }                                                                                              \
  }
  // nine taps of one chunk into ACC; FIRST: the tile's first chunk (the first product of every accumulator takes a
  // zero C operand: no clearing pass); JOB 1 / 2: epilogue units of blocks n = 0,1 / 2,3 of PREV behind taps 0..7
#define PS_BODY(ACC, FIRST, JOB, PREV)                                                             \
  {                                                                                                \
    const bf16x8* buf_ = lds + sb * buf_units;                                                     \
    const bf16x8* Aw = buf_ + a_off;                                                               \
    const bf16x8* Bw = buf_ + b_off;                                                               \
    PS_FRAGS(0, 0)                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                \
      const int F = t & 1;                                                                         \
      if (t < 8) PS_FRAGS(F ^ 1, t + 1)                                                            \
      if (!(PS_DBG & 2)) {                                                                         \
        _Pragma("unroll") for (int m = 0; m < 2; ++m)                                              \
          _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                          \
            const f32x16 zero_ = {};                                                               \
            if (P16) {                                                                             \
              ACC[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[F][m], bh[F][n], ((FIRST) && t == 0) ? zero_ : ACC[m][n], 0, 0, 0); \
            } else {                                                                               \
              ACC[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[F][m], bl[F][n], ((FIRST) && t == 0) ? zero_ : ACC[m][n], 0, 0, 0); \
              ACC[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[F][m], bh[F][n], ACC[m][n], 0, 0, 0); \
              ACC[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[F][m], bh[F][n], ACC[m][n], 0, 0, 0); \
            }                                                                                      \
          }                                                                                        \
      }                                                                                            \
      if ((JOB) != 0 && t < 8 && !(PS_DBG & 1)) {                                                  \
        if constexpr (MODE == PSE_DGRAD_ADDPOOL) {                                                 \
          /* two (channel, window) items of the previous tile: dx rows y, y+1 = acc + routed pooled gradient */ \
          _Pragma("unroll") for (int e_ = 0; e_ < 2; ++e_) {                                       \
            const int it_ = ((JOB) - 1) * 16 + 2 * t + e_;                                         \
            const int m_ = it_ >> 4, gp_ = (it_ >> 3) & 1, ab_ = (it_ >> 2) & 1, i_ = it_ & 3;     \
            const int arg_ = (int)((rkp[m_][gp_][ab_] >> (8 * i_ + 4)) & 3u);                      \
            const float gv_ = dgp[m_][gp_][ab_][i_];                                               \
            const int so_ = (32 * m_ + 16 * gp_ + 8 * ab_ + i_) * p.H * p.Wf * 4;                  \
            _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_) {                                     \
              float z0_, z1_;                                                                      \
              ps_acc_read(PREV[WOVEN ? m_ : 0][WOVEN ? 2 * r_ : 0], 8 * gp_ + 4 * ab_ + i_, z0_);  \
              ps_acc_read(PREV[WOVEN ? m_ : 0][WOVEN ? 2 * r_ + 1 : 0], 8 * gp_ + 4 * ab_ + i_, z1_); \
              z0_ += arg_ == 2 * r_ ? gv_ : 0.f;                                                   \
              z1_ += arg_ == 2 * r_ + 1 ? gv_ : 0.f;                                               \
              __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__builtin_bit_cast(unsigned, z0_), __builtin_bit_cast(unsigned, z1_)}, prs, sto, \
                                                    so_ + r_ * p.Wf * 4, 0);                       \
            }                                                                                      \
          }                                                                                        \
        } else {                                                                                   \
          const int n_ = ((JOB) - 1) * 2 + (t >> 2), m_ = (t >> 1) & 1, gp_ = t & 1;               \
          ps_unit<MODE, P16>(PREV[WOVEN ? m_ : 0][WOVEN ? n_ : 0], gp_, bz[m_][2 * gp_], bz[m_][2 * gp_ + 1], sgp[n_][m_][gp_], p.slope, okn[n_], ob[n_], m_, \
                        half, gstride, plane_o_bytes, yrs);                                        \
        }                                                                                          \
      }                                                                                            \
      /* the next tap's twelve fragment reads ride one per MFMA on the first half of this tap, the unit's VALU  */ \
      /* work three per gap, its two stores near the end                                                        */ \
      if (P16) {   /* 8 MFMAs per tap: the six fragment reads first, the unit's VALU work spread over all gaps */ \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                            \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
          if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
          if ((JOB) != 0) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                       \
          if ((JOB) != 0 && (i >= 7 || (MODE == PSE_DGRAD_ADDPOOL && i >= 4))) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0); \
        }                                                                                          \
      } else {                                                                                     \
      _Pragma("unroll") for (int i = 0; i < 12; ++i) {                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
        if ((JOB) != 0) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                         \
      }                                                                                            \
      _Pragma("unroll") for (int i = 0; i < 12; ++i) {                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        if ((JOB) != 0) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                         \
        if ((JOB) != 0 && (i >= 10 || (MODE == PSE_DGRAD_ADDPOOL && i >= 8))) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0); \
      }                                                                                            \
      }                                                                                            \
    }                                                                                              \
    sb ^= 1;                                                                                       \
  }
  // ---- pooled-block epilogues (POOLM mapping): geometry of this lane's window in tile T
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(MODE == PSE_FWD_POOL ? (void*)p.pool_ps : (void*)p.dx_f32, 0,
      MODE == PSE_FWD_POOL ? (int)((size_t)p.N * p.img_p * 16) : (int)((size_t)p.Nimg * 64 * p.H * p.Wf * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(p.pool_f32, 0, (int)((size_t)p.Nimg * 64 * p.Hp * p.Wpf * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(p.route_out, 0, (int)((size_t)p.N * 64 * p.Hp * p.Wp), 0x00020000);
  const int pool_cnt = MODE == PSE_FWD_POOL ? 4 * ((P16 ? 1 : 2) * (p.pool_ps != nullptr) + (p.route_out != nullptr) + 8 * (p.pool_f32 != nullptr)) : 63;
#define PS_WIN(T)                                                                                  \
    const int rb_ = WP == 64 ? 2 * wid : 4 * wid + 2 * (l31 >> 4);                                 \
    const int v_ = (T) * R + rb_;                                                                  \
    const int nn = (int)__umulhi((unsigned)v_, p.magic_hp), y = v_ - nn * HP;                      \
    const int xp = WP == 64 ? l31 : (l31 & 15), yp = y >> 1;                                       \
    const bool okw = nn < p.N && y < p.H && 2 * xp < (nn >= p.last0 ? p.Wlast : p.W);              \
    /* image and pooled column in the fp32 NCHW tensors (strips: nn = s * Nimg + ni) */            \
    const int si = p.last0 == 0 ? 0 : (p.Nimg == 1 ? nn : (int)__umulhi((unsigned)nn, p.magic_nimg)); \
    const int ni = nn - si * p.Nimg, xg = si * (p.xo >> 1) + xp;
  // DGRAD_ADDPOOL, woven: the pooled gradient and routing bytes of the PREVIOUS tile's windows (40 loads, always issued:
  // a lane without a window reads element 0) and the byte offset of its first dx element (out of range without a window)
  float dgp[2][2][2][4];
  unsigned rkp[2][2][2];
  unsigned sto = 0x80000000u;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp)
#pragma unroll
      for (int ab = 0; ab < 2; ++ab) {
        rkp[m][gp][ab] = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) dgp[m][gp][ab][i] = 0.f;
      }
#define PS_POOLPREF(T)                                                                             \
  {                                                                                                \
    PS_WIN(T)                                                                                      \
    const int HWp = p.Hp * p.Wpf, HWr = p.Hp * p.Wp;                                               \
    const int pbase = okw ? (ni * 64 * p.Hp + yp) * p.Wpf + xg : 0;                                \
    const int rbase = okw ? (nn * 8 * p.Hp + yp) * p.Wp + xp : 0;                                  \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                  \
      _Pragma("unroll") for (int gp = 0; gp < 2; ++gp)                                             \
        _Pragma("unroll") for (int ab = 0; ab < 2; ++ab) {                                         \
          const int G = 4 * m + 2 * gp + ab;                                                       \
          _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                          \
            const float v_ = p.dout[pbase + (8 * G + 4 * half + i) * HWp];                         \
            dgp[m][gp][ab][i] = okw ? v_ : 0.f;                                                    \
          }                                                                                        \
          rkp[m][gp][ab] = *reinterpret_cast<const unsigned*>(p.route_in + (size_t)(rbase + G * HWr) * 8 + 4 * half); \
        }                                                                                          \
    sto = okw ? (unsigned)(((ni * 64 + 4 * half) * p.H + y) * p.Wf + 2 * xg) * 4u : 0x80000000u;   \
  }
  // maxpool2x2(lrelu(acc + bias) * scale + skip) -> pooled PS / fp32 NCHW, routing bytes.  Every load first; stores are
  // buffer stores with the validity in the offset (exactly pool_cnt per wave, counted by the next tile's first wait).
#define PS_EPI_FWD_POOL(ACC, T)                                                                    \
  if (!(PS_DBG & 1)) {                                                                             \
    PS_WIN(T)                                                                                      \
    const int sb0 = okw ? nn * p.img_i + y * WP + 2 * xp + 1 : 0;                                  \
    u32x2_t sk[2][2][4][4];                                                                        \
    float scv[2][2][2][4];                                                                         \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                  \
      _Pragma("unroll") for (int gp = 0; gp < 2; ++gp) {                                           \
        const size_t ga = (size_t)(4 * m + 2 * gp) * gstride, gb = ga + (size_t)gstride;           \
        _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                            \
          const u32x2_t* ap = reinterpret_cast<const u32x2_t*>(p.aux + sb0 + (n >> 1) * WP + (n & 1)) + half; \
          sk[m][gp][n][0] = ap[ga * 2];                                                            \
          sk[m][gp][n][1] = P16 ? u32x2_t{0u, 0u} : ap[(ga + p.plane_i) * 2];                      \
          sk[m][gp][n][2] = ap[gb * 2];                                                            \
          sk[m][gp][n][3] = P16 ? u32x2_t{0u, 0u} : ap[(gb + p.plane_i) * 2];                      \
        }                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
          const int ch = 32 * m + 16 * gp + 4 * half + i;                                          \
          scv[m][gp][0][i] = (p.scale && okw) ? p.scale[ni * 64 + ch] : 1.f;                       \
          scv[m][gp][1][i] = (p.scale && okw) ? p.scale[ni * 64 + ch + 8] : 1.f;                   \
        }                                                                                          \
      }                                                                                            \
    /* values first, then ONE section per output whose branch is uniform for the whole kernel: the number of stores */ \
    /* between the DMA and the next wait is pool_cnt on every path (tools/audit_vmcnt.py)                            */ \
    float pvs[2][2][2][4];                                                                         \
    unsigned rts[2][2][2];                                                                         \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                  \
      _Pragma("unroll") for (int gp = 0; gp < 2; ++gp) {                                           \
        unsigned rt[2] = {0u, 0u};                                                                 \
        _Pragma("unroll") for (int ab = 0; ab < 2; ++ab)                                           \
          _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                          \
            float mx = -INFINITY;                                                                  \
            int arg = 0;                                                                           \
            unsigned bits = 0;                                                                     \
            _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                        \
              float z;                                                                             \
              ps_acc_read(ACC[m][n], 8 * gp + 4 * ab + i, z);                                      \
              const float w_ = z + bz[m][2 * gp + ab][i];                                          \
              z = fmaxf(w_, w_ * p.slope);                                                         \
              bits |= (z > 0.f ? 1u : 0u) << n;                                                    \
              const float u_ = z * scv[m][gp][ab][i] + ps_join(sk[m][gp][n][2 * ab][i >> 1], sk[m][gp][n][2 * ab + 1][i >> 1], i & 1); \
              if (u_ > mx || u_ != u_) { mx = u_; arg = n; }   /* first maximum wins, NaN is a maximum (ATen) */ \
            }                                                                                      \
            pvs[m][gp][ab][i] = mx;                                                                \
            rt[ab] |= (bits | ((unsigned)arg << 4)) << (8 * i);                                    \
          }                                                                                        \
        rts[m][gp][0] = rt[0]; rts[m][gp][1] = rt[1];                                              \
      }                                                                                            \
    if (p.pool_f32) {   /* this lane's own 8 channels: 32m + 16gp + 8ab + 4half + i */              \
      _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                \
        _Pragma("unroll") for (int gp = 0; gp < 2; ++gp) {                                         \
          const unsigned off = okw ? (unsigned)(((ni * 64 + 32 * m + 16 * gp + 4 * half) * p.Hp + yp) * p.Wpf + xg) * 4u : 0x80000000u; \
          _Pragma("unroll") for (int jj = 0; jj < 8; ++jj)                                         \
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pvs[m][gp][jj >> 2][jj & 3]), frs, off, \
                                                  (8 * (jj >> 2) + (jj & 3)) * p.Hp * p.Wpf * 4, 0); \
        }                                                                                          \
    }                                                                                              \
    /* half exchange: lanes 0-31 keep group 2gp (channels 0-3 own, 4-7 from the upper half), lanes 32-63 group 2gp+1 */ \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                  \
      _Pragma("unroll") for (int gp = 0; gp < 2; ++gp) {                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
          auto r_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(pvs[m][gp][0][i]), __float_as_uint(pvs[m][gp][1][i]), false, false); \
          const unsigned r0_ = r_[0], r1_ = r_[1];   /* (a bit_cast applied to a vector ELEMENT is miscompiled: copy first) */ \
          pvs[m][gp][0][i] = __uint_as_float(r0_); pvs[m][gp][1][i] = __uint_as_float(r1_);        \
        }                                                                                          \
        { auto r_ = __builtin_amdgcn_permlane32_swap(rts[m][gp][0], rts[m][gp][1], false, false); rts[m][gp][0] = r_[0]; rts[m][gp][1] = r_[1]; } \
      }                                                                                            \
    if (p.pool_ps) {                                                                               \
      _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                \
        _Pragma("unroll") for (int gp = 0; gp < 2; ++gp) {                                         \
          const int G = 4 * m + 2 * gp + half;                                                     \
          unsigned ha[2], la[2], hb[2], lb2[2];                                                    \
          if (P16) { ps_hi4(pvs[m][gp][0], ha); ps_hi4(pvs[m][gp][1], hb); }                       \
          else { ps_split4(pvs[m][gp][0], ha, la); ps_split4(pvs[m][gp][1], hb, lb2); }            \
          const unsigned off = okw ? (unsigned)(nn * p.img_p + (G * p.HPp + yp) * p.WPp + xp + 1) * 16u : 0x80000000u; \
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{ha[0], ha[1], hb[0], hb[1]}, prs, off, 0, 0); \
          if (!P16) __builtin_amdgcn_raw_buffer_store_b128(u32x4{la[0], la[1], lb2[0], lb2[1]}, prs, off, p.plane_p * 16, 0); \
        }                                                                                          \
    }                                                                                              \
    if (p.route_out) {                                                                             \
      _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                \
        _Pragma("unroll") for (int gp = 0; gp < 2; ++gp) {                                         \
          const int G = 4 * m + 2 * gp + half;                                                     \
          const unsigned off = okw ? (unsigned)(((nn * 8 + G) * p.Hp + yp) * p.Wp + xp) * 8u : 0x80000000u; \
          __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{rts[m][gp][0], rts[m][gp][1]}, rrs, off, 0, 0); \
        }                                                                                          \
    }                                                                                              \
  } else {                                                                                         \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                  \
      _Pragma("unroll") for (int n = 0; n < 4; ++n) ps_keep(ACC[m][n]);                            \
  }
  // dx = acc + unpool(dout) through the routing bytes -> fp32 NCHW, one 8-byte store per (channel, row)
#define PS_EPI_ADDPOOL(ACC, T)                                                                     \
  if (!(PS_DBG & 1)) {                                                                             \
    PS_WIN(T)                                                                                      \
    float dg[2][2][2][4];                                                                          \
    unsigned rk[2][2][2];                                                                          \
    const int HWp = p.Hp * p.Wpf;                                                                  \
    const int pbase = okw ? (ni * 64 * p.Hp + yp) * p.Wpf + xg : 0;                                \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                  \
      _Pragma("unroll") for (int gp = 0; gp < 2; ++gp)                                             \
        _Pragma("unroll") for (int ab = 0; ab < 2; ++ab) {                                         \
          const int G = 4 * m + 2 * gp + ab;                                                       \
          _Pragma("unroll") for (int i = 0; i < 4; ++i) dg[m][gp][ab][i] = okw ? p.dout[pbase + (8 * G + 4 * half + i) * HWp] : 0.f; \
          rk[m][gp][ab] = okw ? *reinterpret_cast<const unsigned*>(p.route_in + (size_t)(((nn * 8 + G) * p.Hp + yp) * p.Wp + xp) * 8 + 4 * half) : 0u; \
        }                                                                                          \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                  \
      _Pragma("unroll") for (int gp = 0; gp < 2; ++gp)                                             \
        _Pragma("unroll") for (int ab = 0; ab < 2; ++ab)                                           \
          _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                          \
            const int ch = 32 * m + 16 * gp + 8 * ab + 4 * half + i;                               \
            const int arg = (int)((rk[m][gp][ab] >> (8 * i + 4)) & 3u);                            \
            const float gv = dg[m][gp][ab][i];                                                     \
            _Pragma("unroll") for (int r = 0; r < 2; ++r) {                                        \
              float z0, z1;                                                                        \
              ps_acc_read(ACC[m][2 * r], 8 * gp + 4 * ab + i, z0);                                 \
              ps_acc_read(ACC[m][2 * r + 1], 8 * gp + 4 * ab + i, z1);                             \
              z0 += arg == 2 * r ? gv : 0.f;                                                       \
              z1 += arg == 2 * r + 1 ? gv : 0.f;                                                   \
              const unsigned off = okw ? (unsigned)(((ni * 64 + ch) * p.H + y + r) * p.Wf + 2 * xg) * 4u : 0x80000000u; \
              __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__builtin_bit_cast(unsigned, z0), __builtin_bit_cast(unsigned, z1)}, prs, off, 0, 0); \
            }                                                                                      \
          }                                                                                        \
  } else {                                                                                         \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                  \
      _Pragma("unroll") for (int n = 0; n < 4; ++n) ps_keep(ACC[m][n]);                            \
  }
  // the whole epilogue of ACC (geometry in ob / okn) at once: every load precedes the first store
#define PS_EPILOGUE(ACC)                                                                           \
  if constexpr (MODE == PSE_FWD_POOL) { PS_EPI_FWD_POOL(ACC, etile) }                              \
  else if constexpr (MODE == PSE_DGRAD_ADDPOOL) { PS_EPI_ADDPOOL(ACC, etile) }                     \
  else if (!(PS_DBG & 1)) {                                                                        \
    if (MODE == PSE_DGRAD_ACT) PS_SIGNS()                                                          \
    _Pragma("unroll") for (int n = 0; n < 4; ++n)                                                  \
      _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                \
        _Pragma("unroll") for (int gp = 0; gp < 2; ++gp)                                           \
          ps_unit<MODE, P16>(ACC[m][n], gp, bz[m][2 * gp], bz[m][2 * gp + 1], sgp[n][m][gp], p.slope,  \
                        okn[n], ob[n], m, half, gstride, plane_o_bytes, yrs);                      \
  } else {                                                                                         \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                  \
      _Pragma("unroll") for (int n = 0; n < 4; ++n) ps_keep(ACC[m][n]);                            \
  }
  // one tile into ACC
#define PS_TILE(ACC, PREV)                                                                         \
  {                                                                                                \
    PS_OPEN()                                                                                      \
    PS_NEXT(0)                                                                                     \
    PS_BODY(ACC, 1, (WOVEN ? 1 : 0), PREV)                                                         \
    young = (WOVEN && !(PS_DBG & 1)) ? (MODE == PSE_DGRAD_ADDPOOL ? 32 : (P16 ? 8 : 16)) : 0;      \
    PS_OPEN()                                                                                      \
    PS_NEXT(1)                                                                                     \
    PS_BODY(ACC, 0, (WOVEN ? 2 : 0), PREV)                                                         \
    for (int c = 2; c < p.nch; ++c) {                                                              \
      PS_OPEN()                                                                                    \
      PS_NEXT(c)                                                                                   \
      PS_BODY(ACC, 0, 0, PREV)                                                                     \
      young = 0;                                                                                   \
    }                                                                                              \
    PS_GEO(tile)                                                                                   \
    etile = tile;                                                                                  \
    if (WOVEN && MODE == PSE_DGRAD_ACT && !(PS_DBG & 1)) {                                         \
      PS_SIGNS()                                                                                   \
      young += 32;                                                                                 \
    }                                                                                              \
    if constexpr (MODE == PSE_DGRAD_ADDPOOL) {                                                     \
      if (!(PS_DBG & 1)) {                                                                         \
        PS_POOLPREF(tile)                                                                          \
        young = min(young + 40, 63);                                                               \
      }                                                                                            \
    }                                                                                              \
    if (!WOVEN) {                                                                                  \
      PS_EPILOGUE(ACC)                                                                             \
      young = (PS_DBG & 1) ? 0 : (POOLM ? pool_cnt : 32);                                          \
    }                                                                                              \
  }

  int etile = tile;                                      // the tile whose epilogue PS_EPILOGUE runs (woven: the loop has moved on)
  PS_ROWOFF(tile)
  PS_DMA(0, 0)
  if (WOVEN) {
    for (;;) {
      PS_TILE(accA, accB)
      tile += tstep;
      if (tile >= tend) { PS_EPILOGUE(accA) break; }
      PS_TILE(accB, accA)
      tile += tstep;
      if (tile >= tend) { PS_EPILOGUE(accB) break; }
    }
  } else {
    for (; tile < tend; tile += tstep) PS_TILE(accA, accA)
  }
#undef PS_TILE
#undef PS_EPILOGUE
#undef PS_EPI_ADDPOOL
#undef PS_EPI_FWD_POOL
#undef PS_POOLPREF
#undef PS_WIN
#undef PS_WAITN
#undef PS_BODY
#undef PS_FRAGS
#undef PS_NEXT
#undef PS_OPEN
#undef PS_GEO
#undef PS_SIGNS
#undef PS_DMA
#undef PS_ROWOFF
}

int ps_num_cus() {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return ncu;
}

// One translation unit per (mode, row width, precision): the woven kernels take a minute or more each to compile, so the
// Makefile builds this source 17 times -- -DPS_TU=<8*P16 + 2*mode + (WP == 64)> emits ONE kernel instantiation behind
// fdet_ps_launch_<n>(), and the plain build keeps the host logic and the C-ABI.
#ifndef PS_TU
#define PS_TU (-1)
#endif
#define PS_CAT_(a, b) a##b
#define PS_CAT(a, b) PS_CAT_(a, b)
#define PS_TU_NAME PS_CAT(fdet_ps_launch_, PS_TU)
template <int MODE, int WP, bool P16>
int launch_ps(const PsConvArgs& p, size_t lds, int grid, hipStream_t st) {
  static bool attr_set = false;
  constexpr int lds_max = 2 * (2 * PSA + 4 * 648) * 16;  // the 64-slot geometry
  if (!attr_set) {
    if ((int)lds > lds_max || hipFuncSetAttribute((const void*)k_conv3x3_ps<MODE, WP, P16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess) {
      (void)hipGetLastError();
      return fail(FDET_ELAUNCH, "conv3x3_ps: cannot reserve %zu bytes of LDS", lds);
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((k_conv3x3_ps<MODE, WP, P16>), dim3(grid), dim3(256), lds, st, p);
  return check_launch("fdet_conv3x3_ps");
}

}  // namespace
#define PS_DECL_LAUNCH(N) extern "C" int fdet_ps_launch_##N(const void* args, size_t lds, int grid, void* stream);
PS_DECL_LAUNCH(0) PS_DECL_LAUNCH(1) PS_DECL_LAUNCH(2) PS_DECL_LAUNCH(3) PS_DECL_LAUNCH(4) PS_DECL_LAUNCH(5) PS_DECL_LAUNCH(6) PS_DECL_LAUNCH(7)
PS_DECL_LAUNCH(8) PS_DECL_LAUNCH(9) PS_DECL_LAUNCH(10) PS_DECL_LAUNCH(11) PS_DECL_LAUNCH(12) PS_DECL_LAUNCH(13) PS_DECL_LAUNCH(14) PS_DECL_LAUNCH(15)
#if PS_TU >= 0
extern "C" int PS_TU_NAME(const void* args, size_t lds, int grid, void* stream) {
  return launch_ps<(PS_TU & 7) / 2, (PS_TU & 1) ? 64 : 32, (PS_TU >= 8)>(*reinterpret_cast<const PsConvArgs*>(args), lds, grid, (hipStream_t)stream);
}
#endif
namespace {
#if PS_TU == -1
typedef int (*ps_launch_fn)(const void*, size_t, int, void*);
const ps_launch_fn PS_LAUNCHERS[16] = {fdet_ps_launch_0, fdet_ps_launch_1, fdet_ps_launch_2, fdet_ps_launch_3,
                                       fdet_ps_launch_4, fdet_ps_launch_5, fdet_ps_launch_6, fdet_ps_launch_7,
                                       fdet_ps_launch_8, fdet_ps_launch_9, fdet_ps_launch_10, fdet_ps_launch_11,
                                       fdet_ps_launch_12, fdet_ps_launch_13, fdet_ps_launch_14, fdet_ps_launch_15};
struct PsPoolIO {
  const float* scale = nullptr; void* pool_ps = nullptr; float* pool_f32 = nullptr; unsigned char* route_out = nullptr;
  const float* dout = nullptr; const unsigned char* route_in = nullptr; float* dx_f32 = nullptr;
};

int run_ps(int mode, const void* x, const void* wpk, const float* bias, const void* aux, void* y, const PsPoolIO& io, int N,
           int Cin, int Cout, int H, int W, float slope, hipStream_t st, bool p16 = false) {
  PsGeo gi, go, gp;
  PsStrips sp, spo;
  const bool pooled = mode == PSE_FWD_POOL || mode == PSE_DGRAD_ADDPOOL;
  FDET_REQUIRE(x && wpk && (y || pooled), "conv3x3_ps: null pointer");
  FDET_REQUIRE(Cout == 64 && Cin % 16 == 0 && Cin >= 32, "conv3x3_ps: Cout must be 64 and Cin a multiple of 16 (Cin=%d Cout=%d)", Cin, Cout);
  // maps wider than 62 columns run as column strips (fdet_ps.h): N, W below are those of the strip-images
  FDET_REQUIRE(ps_geo_strips(N, Cin, H, W, gi, sp) && ps_geo_strips(N, Cout, H, W, go, spo) && gi.WP >= 32 && sp.Ws + 2 <= gi.WP,
               "conv3x3_ps: unsupported map %dx%d (W + 2 <= 32 or 64 slots, or an even width in strips)", H, W);
  FDET_REQUIRE((size_t)(gi.N + 2) * gi.img * 16 < ((size_t)1 << 32) && (size_t)(go.N + 2) * go.img * 16 < ((size_t)1 << 32),
               "conv3x3_ps: tensor too large for the 32-bit byte offsets of the DMA descriptors (N=%d H=%d W=%d)", N, H, W);
  const int Nimg = N, Wfull = W;
  N = gi.N; W = sp.Ws;
  FDET_REQUIRE(slope >= 0.f && slope <= 1.f, "conv3x3_ps: slope must be in [0, 1]");
  PsConvArgs p;
  p.x = reinterpret_cast<const bf16x8*>(x);
  const size_t units = (size_t)(Cin / 16) * PSA;
  p.a_hi = reinterpret_cast<const bf16x8*>(wpk);
  p.a_lo = p.a_hi + units;
  p.bias = bias;
  p.y = reinterpret_cast<bf16x8*>(y);
  p.aux = reinterpret_cast<const bf16x8*>(aux);
  p.scale = io.scale; p.pool_ps = reinterpret_cast<bf16x8*>(io.pool_ps); p.pool_f32 = io.pool_f32; p.route_out = io.route_out;
  p.dout = io.dout; p.route_in = io.route_in; p.dx_f32 = io.dx_f32;
  p.Hp = H / 2; p.Wp = W / 2; p.HPp = p.WPp = p.plane_p = p.img_p = 0;
  p.Nimg = Nimg; p.last0 = sp.S > 1 ? (sp.S - 1) * Nimg : 0; p.Wlast = sp.Wlast; p.Wf = Wfull; p.Wpf = Wfull / 2; p.xo = sp.Ws;
  p.magic_nimg = Nimg > 1 ? magic_of(Nimg) : 0u;
  if (pooled) {
    FDET_REQUIRE(!(H & 1) && !(W & 1) && Cin == 64, "conv3x3_ps (pooled block): even map and 64 channels required (H=%d W=%d Cin=%d)", H, W, Cin);
    FDET_REQUIRE((size_t)Nimg * 64 * H * Wfull * 4 < ((size_t)1 << 31), "conv3x3_ps (pooled block): tensor too large for 32-bit offsets");
    FDET_REQUIRE(sp.S == 1 || !io.pool_ps, "conv3x3_ps_fwd_pool: a strip map writes its pooled output as fp32 NCHW (pool_f32)");
    if (mode == PSE_FWD_POOL && io.pool_ps) {
      FDET_REQUIRE(ps_geo(N, 64, H / 2, W / 2, gp), "conv3x3_ps_fwd_pool: the pooled map %dx%d has no PS layout", H / 2, W / 2);
      p.HPp = gp.HP; p.WPp = gp.WP; p.plane_p = gp.plane; p.img_p = gp.img;
    }
  }
  p.N = N; p.H = H; p.W = W; p.HP = gi.HP;
  p.nch = Cin / 16;
  const int R = 512 / gi.WP, PT = (R + 2) * gi.WP + 8;
  const long vr = (long)N * gi.HP;
  FDET_REQUIRE(vr + gi.HP < (1 << 20), "conv3x3_ps: too many rows");
  p.ntiles = (int)((vr + R - 1) / R);
  p.plane_i = gi.plane; p.img_i = gi.img; p.plane_o = go.plane; p.img_o = go.img;
  p.magic_hp = magic_of(gi.HP);
  p.slope = slope;
  const size_t lds = (size_t)2 * (2 * PSA + 4 * PT) * 16;
  const int grid = std::min(p.ntiles, ps_num_cus());
  const bool w64 = gi.WP == 64;
  switch (mode) {
    case PSE_FWD_FULL:
      FDET_REQUIRE(bias, "conv3x3_ps_fwd: bias is required");
      break;
    case PSE_DGRAD_ACT:
      FDET_REQUIRE(aux, "conv3x3_ps_dgrad_act: the activation is required");
      break;
    case PSE_FWD_POOL:
      FDET_REQUIRE(bias && aux && (io.pool_ps || io.pool_f32), "conv3x3_ps_fwd_pool: bias, skip and an output are required");
      break;
    default:
      FDET_REQUIRE(io.dout && io.route_in && io.dx_f32, "conv3x3_ps_dgrad_unpool: dout, route and dx are required");
      break;
  }
  return PS_LAUNCHERS[(p16 ? 8 : 0) + 2 * mode + (w64 ? 1 : 0)](&p, lds, grid, (void*)st);
}
#endif  // PS_TU == -1

}  // namespace

#if PS_TU == -1
// y_ps = LeakyReLU(conv3x3(x_ps, W) + bias); x_ps / y_ps: PS tensors (image 0), wpk: forward panels of
// fdet_pack_conv3x3_weights_bf16x3
extern "C" int fdet_conv3x3_ps_fwd(const void* x_ps, const void* wpk, const float* bias, void* y_ps, int N, int Cin,
                                   int Cout, int H, int W, float slope, void* stream) {
  return run_ps(PSE_FWD_FULL, x_ps, wpk, bias, nullptr, y_ps, PsPoolIO{}, N, Cin, Cout, H, W, slope, (hipStream_t)stream);
}

// dx_ps = conv3x3^T(dz_ps, W) * LeakyReLU'(act_ps); wpk: backward panels
extern "C" int fdet_conv3x3_ps_dgrad_act(const void* dz_ps, const void* wpk, const void* act_ps, void* dx_ps, int N,
                                         int Cin, int Cout, int H, int W, float slope, void* stream) {
  return run_ps(PSE_DGRAD_ACT, dz_ps, wpk, nullptr, act_ps, dx_ps, PsPoolIO{}, N, Cout, Cin, H, W, slope, (hipStream_t)stream);
}

// Pooled residual block on PS tensors (models/PoolResnet.py:36-42 and its autograd), the tail fused as in
// fdet_conv3x3_fwd_pool_bf16x3 / fdet_conv3x3_dgrad_unpool_bf16x3:
//   forward : pooled = maxpool2x2(LeakyReLU(conv(x_ps) + bias) * drop_scale + skip_ps) -> pool_ps (PS) and / or pool_f32
//             (fp32 NCHW); route8 [N][8][H/2][W/2][8] bytes (channel-innermost; NULL in eval): bits 0-3 = (c > 0) of the
//             window's elements in scan order, bits 4-5 = index of the maximum (first maximum wins, NaN is a maximum)
//   backward: dx (fp32 NCHW) = conv^T(dz_ps) + unpool(dout_pooled) through route8
extern "C" int fdet_conv3x3_ps_fwd_pool(const void* x_ps, const void* wpk, const float* bias, const void* skip_ps,
                                        const float* drop_scale, void* pool_ps, float* pool_f32, unsigned char* route8,
                                        int N, int Cin, int Cout, int H, int W, float slope, void* stream) {
  PsPoolIO io;
  io.scale = drop_scale; io.pool_ps = pool_ps; io.pool_f32 = pool_f32; io.route_out = route8;
  return run_ps(PSE_FWD_POOL, x_ps, wpk, bias, skip_ps, nullptr, io, N, Cin, Cout, H, W, slope, (hipStream_t)stream);
}

extern "C" int fdet_conv3x3_ps_dgrad_unpool(const void* dz_ps, const void* wpk, const float* dout_pooled,
                                            const unsigned char* route8, float* dx, int N, int Cin, int Cout, int H, int W,
                                            float slope, void* stream) {
  PsPoolIO io;
  io.dout = dout_pooled; io.route_in = route8; io.dx_f32 = dx;
  return run_ps(PSE_DGRAD_ADDPOOL, dz_ps, wpk, nullptr, nullptr, nullptr, io, N, Cout, Cin, H, W, slope, (hipStream_t)stream);
}

// ---- precision16 (round 4): the same four operations with ONE bf16 MFMA pass on the hi planes (bf16 activations and
// weights, fp32 accumulate and epilogue arithmetic), hi planes only moved and written.  The arithmetic of the reference's
// Trainer(precision=16) (train_model.py:50) with bf16 as the 16-bit type.
extern "C" int fdet_conv3x3_ps_fwd_p16(const void* x_ps, const void* wpk, const float* bias, void* y_ps, int N, int Cin,
                                       int Cout, int H, int W, float slope, void* stream) {
  return run_ps(PSE_FWD_FULL, x_ps, wpk, bias, nullptr, y_ps, PsPoolIO{}, N, Cin, Cout, H, W, slope, (hipStream_t)stream, true);
}
extern "C" int fdet_conv3x3_ps_dgrad_act_p16(const void* dz_ps, const void* wpk, const void* act_ps, void* dx_ps, int N,
                                             int Cin, int Cout, int H, int W, float slope, void* stream) {
  return run_ps(PSE_DGRAD_ACT, dz_ps, wpk, nullptr, act_ps, dx_ps, PsPoolIO{}, N, Cout, Cin, H, W, slope, (hipStream_t)stream, true);
}
extern "C" int fdet_conv3x3_ps_fwd_pool_p16(const void* x_ps, const void* wpk, const float* bias, const void* skip_ps,
                                            const float* drop_scale, void* pool_ps, float* pool_f32, unsigned char* route8,
                                            int N, int Cin, int Cout, int H, int W, float slope, void* stream) {
  PsPoolIO io;
  io.scale = drop_scale; io.pool_ps = pool_ps; io.pool_f32 = pool_f32; io.route_out = route8;
  return run_ps(PSE_FWD_POOL, x_ps, wpk, bias, skip_ps, nullptr, io, N, Cin, Cout, H, W, slope, (hipStream_t)stream, true);
}
extern "C" int fdet_conv3x3_ps_dgrad_unpool_p16(const void* dz_ps, const void* wpk, const float* dout_pooled,
                                                const unsigned char* route8, float* dx, int N, int Cin, int Cout, int H, int W,
                                                float slope, void* stream) {
  PsPoolIO io;
  io.dout = dout_pooled; io.route_in = route8; io.dx_f32 = dx;
  return run_ps(PSE_DGRAD_ADDPOOL, dz_ps, wpk, nullptr, nullptr, nullptr, io, N, Cout, Cin, H, W, slope, (hipStream_t)stream, true);
}

#endif  // PS_TU == -1
