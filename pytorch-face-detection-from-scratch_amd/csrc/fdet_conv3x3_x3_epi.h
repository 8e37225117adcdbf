// Tile epilogue shared by the ping-pong kernel (fdet_conv3x3_x3_pp.hip) and the aligned-band variant of the small-tile
// kernel (fdet_conv3x3_x3_sb.hip): image-aligned bands, a wave owns two 32-position blocks n = 0,1 (row-pair mapping: the
// same 32 columns of two adjacent rows, so whole 2x2 pooling windows sit in one lane after the quad transpose).
// ARGS must provide: ConvArgs c; PoolArgs q; unsigned magic_wp.
#pragma once
#include "fdet_conv3x3_x3.h"

namespace {

// 4x4 dword transpose across each quad of lanes: register i of lane j <-> register j of lane i, for TWO register
// quads at once.  Two exchange steps (lane ^ 1 on registers (0,1),(2,3); lane ^ 2 on (0,2),(1,3)); every new value is
// ONE v_cndmask_b32_dpp (D = vcc ? own : other lane's), two instructions per exchanged pair where round 1's builtin
// form (select, DPP move, two selects) cost four.  Written as asm because hipcc turns `cond ? own : dpp(other)` into a
// DPP move under a narrowed EXEC, which reads zeros from the disabled source lanes.
//   mb0 / mb1: wave masks of (lane & 1) / (lane & 2); nb0 / nb1 their complements.
// Hazards inside the string: a VGPR written by a VALU needs two wait states before a DPP read (s_nop 1 between the
// steps; within a step every DPP source is an input).  The inputs are MFMA results of an earlier phase (a barrier and
// hundreds of cycles away), so no XDL-write hazard reaches this code.
__device__ __forceinline__ void quad_transpose8(const float (&v)[8], float (&c)[8], unsigned long long mb0,
                                                unsigned long long nb0, unsigned long long mb1, unsigned long long nb1) {
  float a0, a1, a2, a3, a4, a5, a6, a7;
  asm volatile(
      "s_mov_b64 vcc, %[mb0]\n\t"
      "v_cndmask_b32_dpp %[a1], %[v0], %[v1], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %[a3], %[v2], %[v3], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %[a5], %[v4], %[v5], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %[a7], %[v6], %[v7], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_mov_b64 vcc, %[nb0]\n\t"
      "v_cndmask_b32_dpp %[a0], %[v1], %[v0], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %[a2], %[v3], %[v2], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %[a4], %[v5], %[v4], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %[a6], %[v7], %[v6], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_mov_b64 vcc, %[mb1]\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_dpp %[c2], %[a0], %[a2], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %[c3], %[a1], %[a3], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %[c6], %[a4], %[a6], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %[c7], %[a5], %[a7], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_mov_b64 vcc, %[nb1]\n\t"
      "v_cndmask_b32_dpp %[c0], %[a2], %[a0], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %[c1], %[a3], %[a1], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %[c4], %[a6], %[a4], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %[c5], %[a7], %[a5], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1"
      : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [a4] "=&v"(a4), [a5] "=&v"(a5), [a6] "=&v"(a6),
        [a7] "=&v"(a7), [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [c2] "=&v"(c[2]), [c3] "=&v"(c[3]), [c4] "=&v"(c[4]),
        [c5] "=&v"(c[5]), [c6] "=&v"(c[6]), [c7] "=&v"(c[7])
      : [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]),
        [v7] "v"(v[7]), [mb0] "s"(mb0), [nb0] "s"(nb0), [mb1] "s"(mb1), [nb1] "s"(nb1)
      : "vcc");
}

// first maximum in window scan order wins, NaN is a maximum (ATen max_pool2d; fdet_tail.hip)
__device__ __forceinline__ void upd(float v, int k, float& m, int& arg) {
  if (v > m || v != v) { m = v; arg = k; }
}

#define PP_VEC_LD(DST, SRC, NV)                                                                    \
  { if ((NV) == 4) __builtin_memcpy(&(DST), (SRC), 16); else if ((NV) == 3) __builtin_memcpy(&(DST), (SRC), 12); \
    else if ((NV) == 2) __builtin_memcpy(&(DST), (SRC), 8); else if ((NV) == 1) __builtin_memcpy(&(DST), (SRC), 4); }

// Epilogue of one tile, in two parts so that its global loads travel under the LDS writes of the next chunk and
// under the register transposes.  The MFMA leaves lane = position, 4 registers = 4 consecutive channels; a 4x4 dword
// transpose across each quad of lanes gives lane = channel, 4 registers = 4 consecutive columns of one row:
// 16-byte global accesses over 128-byte runs.  n = 0,1 are the wave's two 32-position blocks (rowpair: the same
// 32 columns of two adjacent rows).
struct EpiGeo {
  int nv[2], idx0[2];      // valid columns (0..4) and element index of (channel cob0+4half+j, row, column) per n
  int npair, pidx0;        // pooled modes: valid windows (0..2) and pooled element index
};

template <int MODE, class ARGS>
__device__ __forceinline__ EpiGeo epi_geometry(const ARGS& p, int img, int y0, int qwave, int nstride, int cob0, int l31, int half) {
  const ConvArgs& a = p.c;
  constexpr bool POOLM = MODE == EPI_FWD_POOL || MODE == EPI_DGRAD_ADDPOOL;
  const int HW = a.H * a.W, j = l31 & 3;
  EpiGeo e;
  int yrow0 = 0, ox0 = 0;
  bool ok0 = false;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int q4 = qwave + n * nstride + (l31 & ~3);
    const int tr = fdiv(q4, p.magic_wp), ox = q4 - tr * a.WP;
    const int y = y0 + tr;
    const bool ok = tr < a.R && y < a.H && ox < a.W;
    e.nv[n] = ok ? min(4, a.W - ox) : 0;
    e.idx0[n] = (ok ? ((img * a.Cout) * a.H + y) * a.W + ox : 0) + (cob0 + 4 * half + j) * HW;   // + (32m + 8g)*HW
    if (n == 0) { yrow0 = y; ox0 = ox; ok0 = ok; }
  }
  // pooled geometry (rowpair tiles only: n = 0/1 are rows y, y+1 with y even, same columns)
  const int Hp = a.H >> 1, Wp = a.W >> 1;
  e.npair = (POOLM && ok0) ? min(2, (a.W - ox0) >> 1) : 0;
  e.pidx0 = POOLM ? ((ok0 ? ((img * a.Cout) * Hp + (yrow0 >> 1)) * Wp + (ox0 >> 1) : 0) + (cob0 + 4 * half + j) * Hp * Wp) : 0;
  return e;
}

// part 1: every global load of the epilogue (nothing consumes them here)
template <int MT, int MODE, class ARGS>
__device__ __forceinline__ void epi_loads(const ARGS& p, const EpiGeo& e, f32x4 (&u)[MT][2][4], float (&dg)[MT][4][2],
                                          unsigned (&mk)[MT][4], float (&bz)[MT][4], float (&sc)[MT][4], int img, int cob0,
                                          int l31, int half) {
  const ConvArgs& a = p.c;
  {
    // bias / dropout scale of this lane's channels: requested HERE, ahead of every store of the epilogue -- a load issued
    // after stores is only usable once those stores have been acknowledged (vmcnt counts in order): 2-3 k cycles each
    constexpr bool FWD_ = MODE == EPI_FWD_FULL || MODE == EPI_FWD_BOTH || MODE == EPI_FWD_OUT || MODE == EPI_FWD_POOL;
    const int j_ = l31 & 3;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = cob0 + 32 * m + 8 * g + 4 * half + j_;
        bz[m][g] = FWD_ ? a.bias[ch] : 0.f;
        sc[m][g] = ((MODE == EPI_FWD_BOTH || MODE == EPI_FWD_POOL) && a.scale) ? a.scale[img * a.Cout + ch] : 1.f;
      }
  }
  constexpr bool HAS_LD = MODE == EPI_FWD_BOTH || MODE == EPI_FWD_OUT || MODE == EPI_DGRAD_ACT || MODE == EPI_DGRAD_ADD || MODE == EPI_FWD_POOL;
  const float* __restrict__ src = MODE == EPI_DGRAD_ACT ? a.act : a.skip;
  const int HW = a.H * a.W;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) u[m][n][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (HAS_LD) {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      // the branch on the valid count sits outside the load loops (a branch per load serialises them)
      if (e.nv[n] == 4) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int g = 0; g < 4; ++g) __builtin_memcpy(&u[m][n][g], src + e.idx0[n] + (32 * m + 8 * g) * HW, 16);
      } else if (e.nv[n] > 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int g = 0; g < 4; ++g) PP_VEC_LD(u[m][n][g], src + e.idx0[n] + (32 * m + 8 * g) * HW, e.nv[n])
      }
    }
  }
  if (MODE == EPI_DGRAD_ADDPOOL) {
    const float* __restrict__ g_din = p.q.pool_din;
    const unsigned char* __restrict__ g_mk = p.q.mask_in;
    const int HWp = (a.H >> 1) * (a.W >> 1);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int pi = e.pidx0 + (32 * m + 8 * g) * HWp;
        dg[m][g][0] = dg[m][g][1] = 0.f;
        mk[m][g] = 0u;
        if (e.npair == 2) { __builtin_memcpy(dg[m][g], g_din + pi, 8); mk[m][g] = (unsigned)g_mk[pi] | ((unsigned)g_mk[pi + 1] << 8); }
        else if (e.npair == 1) { dg[m][g][0] = g_din[pi]; mk[m][g] = g_mk[pi]; }
      }
  }
}

// part 2: transposes, arithmetic, stores
template <int MT, int MODE, class ARGS>
__device__ __forceinline__ void epi_finish(const ARGS& p, const EpiGeo& e, f32x16 (&acc)[MT][2], f32x4 (&u)[MT][2][4],
                                           float (&dg)[MT][4][2], unsigned (&mk)[MT][4], const float (&bzm)[MT][4],
                                           const float (&scm)[MT][4], int img, int cob0, int l31, int half) {
  const ConvArgs& a = p.c;
  float* __restrict__ g_full = a.y_full;
  float* __restrict__ g_out = a.y_out;
  const int HW = a.H * a.W, HWp = (a.H >> 1) * (a.W >> 1);
  const unsigned long long mb0 = 0xAAAAAAAAAAAAAAAAull, nb0 = ~mb0, mb1 = 0xCCCCCCCCCCCCCCCCull, nb1 = ~mb1;   // lane & 1, lane & 2
  constexpr bool FWD = MODE == EPI_FWD_FULL || MODE == EPI_FWD_BOTH || MODE == EPI_FWD_OUT || MODE == EPI_FWD_POOL;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    float bz[4], sc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { bz[g] = bzm[m][g]; sc[g] = scm[m][g]; }
    f32x4 t[2][4];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        float vi[8], co[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) vi[r] = acc[m][n][4 * g + r];
        quad_transpose8(vi, co, mb0, nb0, mb1, nb1);
        t[n][g] = f32x4{co[0], co[1], co[2], co[3]};
        t[n][g + 1] = f32x4{co[4], co[5], co[6], co[7]};
      }
    if (MODE == EPI_DGRAD_ADDPOOL) {
      // unpool(dout): window (row pair, column pair pc) sends its gradient to element arg = 2*row + col
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
          const int arg = (mk[m][g] >> (8 * pc + 4)) & 3;
          const float gv = dg[m][g][pc];
          u[m][0][g][2 * pc] = arg == 0 ? gv : 0.f;
          u[m][0][g][2 * pc + 1] = arg == 1 ? gv : 0.f;
          u[m][1][g][2 * pc] = arg == 2 ? gv : 0.f;
          u[m][1][g][2 * pc + 1] = arg == 3 ? gv : 0.f;
        }
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float z = t[n][g][i];
          if (FWD) {
            const float w_ = z + bz[g];
            z = fmaxf(w_, w_ * a.slope);              // == w > 0 ? w : w*slope for 0 <= slope <= 1 (NaN stays NaN)
            if (MODE == EPI_FWD_BOTH || MODE == EPI_FWD_POOL) u[m][n][g][i] = z * sc[g] + u[m][n][g][i];
            if (MODE == EPI_FWD_OUT) u[m][n][g][i] = z + u[m][n][g][i];
          } else if (MODE == EPI_DGRAD_ACT) {
            z *= (u[m][n][g][i] > 0.f) ? 1.f : a.slope;
          } else {
            z += u[m][n][g][i];
          }
          t[n][g][i] = z;
        }
    if (MODE == EPI_FWD_POOL) {
      float* __restrict__ g_pool = p.q.pool_out;
      unsigned char* __restrict__ g_mk = p.q.mask_out;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float po[2];
        unsigned char pm[2];
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
          float mx = -INFINITY;
          int arg = 0;
          upd(u[m][0][g][2 * pc], 0, mx, arg);
          upd(u[m][0][g][2 * pc + 1], 1, mx, arg);
          upd(u[m][1][g][2 * pc], 2, mx, arg);
          upd(u[m][1][g][2 * pc + 1], 3, mx, arg);
          po[pc] = mx;
          pm[pc] = (unsigned char)((t[0][g][2 * pc] > 0.f ? 1 : 0) | (t[0][g][2 * pc + 1] > 0.f ? 2 : 0) |
                                   (t[1][g][2 * pc] > 0.f ? 4 : 0) | (t[1][g][2 * pc + 1] > 0.f ? 8 : 0) | (arg << 4));
        }
        const int pi = e.pidx0 + (32 * m + 8 * g) * HWp;
        if (e.npair == 2) {
          __builtin_memcpy(g_pool + pi, po, 8);
          if (g_mk) { g_mk[pi] = pm[0]; g_mk[pi + 1] = pm[1]; }
        } else if (e.npair == 1) {
          g_pool[pi] = po[0];
          if (g_mk) g_mk[pi] = pm[0];
        }
      }
    } else {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
#define PP_ST(BYTES)                                                                               \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                  \
    const int idx_ = e.idx0[n] + (32 * m + 8 * g) * HW;                                            \
    if (MODE != EPI_FWD_OUT) __builtin_memcpy(g_full + idx_, &t[n][g], BYTES);                     \
    if (MODE == EPI_FWD_BOTH || MODE == EPI_FWD_OUT) __builtin_memcpy(g_out + idx_, &u[m][n][g], BYTES); \
  }
        if (e.nv[n] == 4) { PP_ST(16) } else if (e.nv[n] == 3) { PP_ST(12) } else if (e.nv[n] == 2) { PP_ST(8) } else if (e.nv[n] == 1) { PP_ST(4) }
#undef PP_ST
      }
    }
  }
}


}  // namespace
