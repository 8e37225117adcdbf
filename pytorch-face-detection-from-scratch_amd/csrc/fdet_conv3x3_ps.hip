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
#include "fdet_conv3x3_x3.h"
#include "fdet_ps.h"
#include <algorithm>

using namespace fdet;

typedef __attribute__((address_space(3))) void* lds_void_t;
typedef const __attribute__((address_space(1))) void* glb_void_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

#ifndef PS_DBG
#define PS_DBG 0          // development builds (-DPS_DBG=n, timing only): 1 = no epilogue, 2 = no MFMAs, 4 = no DMA
#endif
constexpr int PSA = 9 * 2 * 64;      // weight units per plane (hi / lo) and 16-channel chunk: [tap][k half][64 co]
constexpr int PS_ROWI = 10;          // DMA wave-instructions per activation array and chunk: (R + 2) * WP / 64 <= 10

enum { PSE_FWD_FULL = 0, PSE_DGRAD_ACT = 1 };

struct PsConvArgs {
  const bf16x8* x;           // PS input, image 0
  const bf16x8* a_hi;        // weight panels [Cin/16][9][2][64] x 8 bf16
  const bf16x8* a_lo;
  const float* bias;         // forward modes
  bf16x8* y;                 // PS output, image 0
  const bf16x8* aux;         // PSE_DGRAD_ACT: PS activation whose sign selects the LeakyReLU slope
  int N, H, W, HP;
  int nch, ntiles;
  int plane_i, img_i, plane_o, img_o;
  unsigned magic_hp;
  float slope;
};

template <int MODE, int WP>
__global__ void __launch_bounds__(256, 1)
k_conv3x3_ps(const PsConvArgs p) {
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

  // per-lane source offsets (units, without the array base) of the activation DMA instructions of a tile
  int ro[PS_ROWI];
#define PS_ROWOFF(T)                                                                               \
  {                                                                                                \
    const int v0_ = (T) * R;                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < PS_ROWI; ++i_) {                                       \
      const int u_ = i_ * 64 + lane;                                                               \
      const int vv_ = v0_ - 1 + (u_ >> WPL) + HP;                                                \
      const int nn_ = (int)__umulhi((unsigned)vv_, p.magic_hp);                                    \
      ro[i_] = (nn_ - 1) * p.img_i + (vv_ - nn_ * HP) * WP + (u_ & (WP - 1));                      \
    }                                                                                              \
  }
  // DMA of chunk C of the tile whose offsets are in ro[] into buffer BS: wave w moves weight plane w >> 1, half
  // w & 1 (nine 1-KiB pieces) and activation array w = (plane, k half) (NBI pieces)
#define PS_DMA(C, BS)                                                                              \
  if (!(PS_DBG & 4)) {                                                                              \
    bf16x8* db_ = lds + (BS) * buf_units;                                                          \
    const bf16x8* as_ = ((wid >> 1) ? p.a_lo : p.a_hi) + (size_t)(C) * PSA + (wid & 1) * 576 + lane; \
    bf16x8* ad_ = db_ + (wid >> 1) * PSA + (wid & 1) * 576;                                        \
    _Pragma("unroll") for (int k_ = 0; k_ < 9; ++k_)                                               \
      __builtin_amdgcn_global_load_lds((glb_void_t)(as_ + k_ * 64), (lds_void_t)(ad_ + k_ * 64), 16, 0, 0); \
    const bf16x8* bs_ = p.x + (size_t)(wid >> 1) * p.plane_i + (size_t)(2 * (C) + (wid & 1)) * HP * WP; \
    bf16x8* bd_ = db_ + 2 * PSA + wid * PT;                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < PS_ROWI; ++i_)                                         \
      if (i_ < NBI) __builtin_amdgcn_global_load_lds((glb_void_t)(bs_ + ro[i_]), (lds_void_t)(bd_ + i_ * 64), 16, 0, 0); \
  }

  f32x16 acc[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // this lane's bias values: channel 32m + 8g + 4half + i
  float bz[2][4][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) bz[m][g][i] = (MODE == PSE_FWD_FULL) ? p.bias[32 * m + 8 * g + 4 * half + i] : 0.f;

  const int a_off = half * 64 + l31;                     // + tap*128 + m*32 ; lo plane: + PSA
  const int b_off = 2 * PSA + half * PT + l31;           // + qn[n] + tap offset ; lo planes: + 2*PT

  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.N * p.img_o * 16), 0x00020000);
  bool stored = false;                                   // the previous tile's 32 epilogue stores may still be in flight
  PS_ROWOFF(tile)
  PS_DMA(0, 0)
  int sb = 0;
  for (; tile < tend; tile += tstep) {
    const int v0 = tile * R;
    for (int c = 0; c < p.nch; ++c) {
      // chunk c of this tile has landed (it was issued a whole chunk ago; the epilogue's stores are younger)
      if (c == 0 && stored) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // ... in every wave, and the other buffer is free
      if (c + 1 < p.nch) {
        PS_DMA(c + 1, sb ^ 1)
      } else if (tile + tstep < tend) {
        PS_ROWOFF(tile + tstep)
        PS_DMA(0, sb ^ 1)
      }
      const bf16x8* buf = lds + sb * buf_units;
      const bf16x8* Aw = buf + a_off;
      const bf16x8* Bw = buf + b_off;
      bf16x8 ah[2][2], al[2][2], bh[2][4], bl[2][4];
#define PS_FRAGS(F, T)                                                                             \
  {                                                                                                \
    const int to_ = ((T) / 3) * WP + (T) % 3;                                                      \
    _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_) {                                             \
      ah[F][m_] = Aw[(T) * 128 + m_ * 32];                                                         \
      al[F][m_] = Aw[PSA + (T) * 128 + m_ * 32];                                                   \
    }                                                                                              \
    _Pragma("unroll") for (int n_ = 0; n_ < 4; ++n_) {                                             \
      bh[F][n_] = Bw[qn[n_] + to_];                                                                \
      bl[F][n_] = Bw[2 * PT + qn[n_] + to_];                                                       \
    }                                                                                              \
  }
      __builtin_amdgcn_sched_barrier(0);
      PS_FRAGS(0, 0)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int F = t & 1;
        if (t < 8) PS_FRAGS(F ^ 1, t + 1)
        if (!(PS_DBG & 2)) {
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[F][m], bl[F][n], acc[m][n], 0, 0, 0);
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[F][m], bh[F][n], acc[m][n], 0, 0, 0);
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[F][m], bh[F][n], acc[m][n], 0, 0, 0);
            }
        }
        // the next tap's twelve fragment reads ride one per MFMA on the first half of this tap
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
      }
#undef PS_FRAGS
      sb ^= 1;
    }
    // ---- epilogue: lane = position, registers = channels 32m + 8g + 4half + i.  A v_permlane32_swap pair makes
    // a lane hold all 8 channels of one unit: 16-byte stores, 512 contiguous bytes per wave-instruction.  Stores are
    // buffer stores with the validity folded into the offset (out of range = dropped): every wave issues exactly 32,
    // which is what the counted wait at the head of the next tile relies on.  Every load precedes the first store.
    if (!(PS_DBG & 1)) {
      int ob[4];
      bool okn[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int q = qn[n] + l31;
        const int col = q & (WP - 1);
        const int v = v0 + (q >> WPL);
        const int nn = (int)__umulhi((unsigned)v, p.magic_hp), y = v - nn * HP;
        okn[n] = nn < p.N && y < p.H && col < p.W;
        ob[n] = nn * p.img_o + y * WP + col + 1;                 // + G*HP*WP (+ plane_o)
      }
      typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
      u32x2_t sg[MODE == PSE_DGRAD_ACT ? 4 : 1][2][2][4];
      if (MODE == PSE_DGRAD_ACT) {
        // the saved activation: this lane's 4 channels (8 bytes) of groups 4m+2gp and 4m+2gp+1, hi and lo
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const u32x2_t* ap = reinterpret_cast<const u32x2_t*>(p.aux + (okn[n] ? ob[n] : 0)) + half;
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
              const size_t ga = (size_t)(4 * m + 2 * gp) * HP * WP, gb = ga + (size_t)HP * WP;
              sg[n][m][gp][0] = ap[ga * 2];
              sg[n][m][gp][1] = ap[(ga + p.plane_o) * 2];
              sg[n][m][gp][2] = ap[gb * 2];
              sg[n][m][gp][3] = ap[(gb + p.plane_o) * 2];
            }
        }
      }
#pragma unroll
      for (int n = 0; n < 4; ++n) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            float za[4], zb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { za[i] = acc[m][n][8 * gp + i]; zb[i] = acc[m][n][8 * gp + 4 + i]; }
            if (MODE == PSE_FWD_FULL) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float wa = za[i] + bz[m][2 * gp][i], wb = zb[i] + bz[m][2 * gp + 1][i];
                za[i] = fmaxf(wa, wa * p.slope);
                zb[i] = fmaxf(wb, wb * p.slope);
              }
            } else if (MODE == PSE_DGRAD_ACT) {
              const int sn = MODE == PSE_DGRAD_ACT ? n : 0;
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                za[i] *= ps_join(sg[sn][m][gp][0][i >> 1], sg[sn][m][gp][1][i >> 1], i & 1) > 0.f ? 1.f : p.slope;
                zb[i] *= ps_join(sg[sn][m][gp][2][i >> 1], sg[sn][m][gp][3][i >> 1], i & 1) > 0.f ? 1.f : p.slope;
              }
            }
            unsigned ha[2], la[2], hb[2], lb[2];
            ps_split4(za, ha, la);
            ps_split4(zb, hb, lb);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              auto r1 = __builtin_amdgcn_permlane32_swap(ha[k], hb[k], false, false);
              ha[k] = r1[0]; hb[k] = r1[1];
              auto r2 = __builtin_amdgcn_permlane32_swap(la[k], lb[k], false, false);
              la[k] = r2[0]; lb[k] = r2[1];
            }
            const int G = 4 * m + 2 * gp + half;
            const unsigned off = okn[n] ? (unsigned)(ob[n] + G * HP * WP) * 16u : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{ha[0], ha[1], hb[0], hb[1]}, yrs, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{la[0], la[1], lb[0], lb[1]}, yrs, off, p.plane_o * 16, 0);
          }
      }
      stored = true;
    }
    if (PS_DBG & 1) {                                    // timing builds: keep the accumulators (and their MFMAs) alive
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) asm volatile("" ::"v"(acc[m][n]));
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  }
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

template <int MODE, int WP>
int launch_ps(const PsConvArgs& p, size_t lds, int grid, hipStream_t st) {
  static bool attr_set = false;
  constexpr int lds_max = 2 * (2 * PSA + 4 * 648) * 16;  // the 64-slot geometry
  if (!attr_set) {
    if ((int)lds > lds_max || hipFuncSetAttribute((const void*)k_conv3x3_ps<MODE, WP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess) {
      (void)hipGetLastError();
      return fail(FDET_ELAUNCH, "conv3x3_ps: cannot reserve %zu bytes of LDS", lds);
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((k_conv3x3_ps<MODE, WP>), dim3(grid), dim3(256), lds, st, p);
  return check_launch("fdet_conv3x3_ps");
}

int run_ps(int mode, const void* x, const void* wpk, const float* bias, const void* aux, void* y, int N, int Cin, int Cout,
           int H, int W, float slope, hipStream_t st) {
  PsGeo gi, go;
  FDET_REQUIRE(x && wpk && y, "conv3x3_ps: null pointer");
  FDET_REQUIRE(Cout == 64 && Cin % 16 == 0 && Cin >= 16, "conv3x3_ps: Cout must be 64 and Cin a multiple of 16 (Cin=%d Cout=%d)", Cin, Cout);
  FDET_REQUIRE(ps_geo(N, Cin, H, W, gi) && ps_geo(N, Cout, H, W, go) && gi.WP >= 32, "conv3x3_ps: unsupported map %dx%d", H, W);
  FDET_REQUIRE(slope >= 0.f && slope <= 1.f, "conv3x3_ps: slope must be in [0, 1]");
  PsConvArgs p;
  p.x = reinterpret_cast<const bf16x8*>(x);
  const size_t units = (size_t)(Cin / 16) * PSA;
  p.a_hi = reinterpret_cast<const bf16x8*>(wpk);
  p.a_lo = p.a_hi + units;
  p.bias = bias;
  p.y = reinterpret_cast<bf16x8*>(y);
  p.aux = reinterpret_cast<const bf16x8*>(aux);
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
  if (mode == PSE_FWD_FULL) {
    FDET_REQUIRE(bias, "conv3x3_ps_fwd: bias is required");
    return gi.WP == 64 ? launch_ps<PSE_FWD_FULL, 64>(p, lds, grid, st) : launch_ps<PSE_FWD_FULL, 32>(p, lds, grid, st);
  }
  FDET_REQUIRE(aux, "conv3x3_ps_dgrad: the activation is required");
  return gi.WP == 64 ? launch_ps<PSE_DGRAD_ACT, 64>(p, lds, grid, st) : launch_ps<PSE_DGRAD_ACT, 32>(p, lds, grid, st);
}

}  // namespace

// y_ps = LeakyReLU(conv3x3(x_ps, W) + bias); x_ps / y_ps: PS tensors (image 0), wpk: forward panels of
// fdet_pack_conv3x3_weights_bf16x3
extern "C" int fdet_conv3x3_ps_fwd(const void* x_ps, const void* wpk, const float* bias, void* y_ps, int N, int Cin,
                                   int Cout, int H, int W, float slope, void* stream) {
  return run_ps(PSE_FWD_FULL, x_ps, wpk, bias, nullptr, y_ps, N, Cin, Cout, H, W, slope, (hipStream_t)stream);
}

// dx_ps = conv3x3^T(dz_ps, W) * LeakyReLU'(act_ps); wpk: backward panels
extern "C" int fdet_conv3x3_ps_dgrad_act(const void* dz_ps, const void* wpk, const void* act_ps, void* dx_ps, int N,
                                         int Cin, int Cout, int H, int W, float slope, void* stream) {
  return run_ps(PSE_DGRAD_ACT, dz_ps, wpk, nullptr, act_ps, dx_ps, N, Cout, Cin, H, W, slope, (hipStream_t)stream);
}
