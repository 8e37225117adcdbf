// bf16x3 3x3 conv forward / data gradient, "ping-pong" kernel for rows of <= 64 columns (the PoolResnet
// resolutions 60x60 / 30x30 / 15x15).  Same arithmetic, weight panels and LDS slot layout as the other
// bf16x3 conv kernels (fdet_conv3x3_x3_kernel.inc); what differs is who does what when:
//
//   * ONE workgroup of 8 waves per CU = two groups of 4 waves (one wave of each group per SIMD).  Each group
//     owns a tile (256 padded positions x MB output channels), its accumulators and its activation buffer.
//   * Time is cut into phases separated by one workgroup barrier.  In every phase one group is in its MFMA
//     segment (one 16-channel chunk: 9 taps x 12 MFMAs per wave, fragments of tap t+1 fetched under the MFMAs
//     of tap t) and has the matrix pipe of its SIMD to itself, while the other group is in its memory
//     segment: global loads of its next chunk, fp32 -> (hi,lo) split, LDS writes, and -- when its tile is
//     complete -- the whole epilogue.  The barrier swaps the roles.  The round-1 kernel (fdet_conv3x3_x3_sb.hip)
//     relied on two independent workgroups per CU drifting into such an alternation; they do not (both sit in the
//     same segment most of the time: matrix pipe 37-47 % busy), here the barrier enforces it.
//   * The pre-split weight panels of a chunk are the same for both groups: they travel L2 -> LDS by LDS-DMA
//     (global_load_lds, no registers, no ds_write) into a two-slot ring shared by the groups, issued by group 0
//     one phase ahead; a chunk is fetched once per TWO tiles.
//   * Tiles are image-aligned bands (rows y0 .. y0+R-1 of ONE image, halo rows outside the image are zeros):
//     no separator rows, and with R even a wave that owns two adjacent rows x 32 columns holds whole 2x2
//     pooling windows, so the pooled blocks' tails are epilogue modes:
//       EPI_FWD_POOL      out = maxpool2x2(lrelu(conv+bias)*scale + skip), plus one routing byte per window
//                         (bits 0-3: lrelu'(c) > 0 of the four elements, bits 4-5: argmax in ATen scan order)
//                         -- c itself is never written (models/PoolResnet.py:37-42)
//       EPI_DGRAD_ADDPOOL dx = conv^T(dz) + unpool(dout) read through the routing bytes
//     which removes k_tail_fwd / k_tail_bwd's full-resolution round trips (fdet_pool_route_bwd below writes dz2).
#include "fdet_conv3x3_x3.h"
#include <algorithm>
#include <cstdint>

using namespace fdet;

namespace {

constexpr int GTHR = 256;     // threads per group
constexpr int PTHR = 512;     // threads per workgroup
__host__ __device__ constexpr int nbs_pp(int vw) { return vw == 4 ? 1 : (vw == 2 ? 2 : 3); }

struct PpArgs {
  ConvArgs c;                // x, bias, epilogue pointers, geometry (WP, R, mode ...)
  PoolArgs q;
  const bf16x8* a_hi;        // [Cin/16][9][2][CoP] x 8 bf16
  const bf16x8* a_lo;
  int PT;                    // units per activation array ((R+2)*WP + pad)
  int p_in;                  // (R+2)*(W/VW) staging items per k-half
  unsigned magic_w;          // W/VW
  unsigned magic_wp;
  int ncob;                  // output-channel blocks
  int bpi;                   // bands per image
  unsigned magic_bpi;
  int ntiles_mb;             // N * bpi
  int rowpair;               // 1: a wave owns 2 rows x 32 columns (WP in {32,64}); 0: 64 consecutive padded positions
};

template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// 4x4 dword transpose across each quad of lanes: register i of lane j <-> register j of lane i
__device__ __forceinline__ f32x4 quad_transpose4(float v0, float v1, float v2, float v3, bool b0, bool b1) {
  float v[4] = {v0, v1, v2, v3};
#pragma unroll
  for (int k = 0; k < 4; k += 2) {
    const float lo = v[k], hi = v[k + 1];
    const float recv = dpp_quad<0xB1>(b0 ? lo : hi);            // quad_perm [1,0,3,2]
    v[k] = b0 ? recv : lo;
    v[k + 1] = b0 ? hi : recv;
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float lo = v[k], hi = v[k + 2];
    const float recv = dpp_quad<0x4E>(b1 ? lo : hi);            // quad_perm [2,3,0,1]
    v[k] = b1 ? recv : lo;
    v[k + 2] = b1 ? hi : recv;
  }
  return f32x4{v[0], v[1], v[2], v[3]};
}

// first maximum in window scan order wins, NaN is a maximum (ATen max_pool2d; fdet_tail.hip)
__device__ __forceinline__ void upd(float v, int k, float& m, int& arg) {
  if (v > m || v != v) { m = v; arg = k; }
}

#define PP_VEC_LD(DST, SRC, NV)                                                                    \
  { if ((NV) == 4) __builtin_memcpy(&(DST), (SRC), 16); else if ((NV) == 3) __builtin_memcpy(&(DST), (SRC), 12); \
    else if ((NV) == 2) __builtin_memcpy(&(DST), (SRC), 8); else if ((NV) == 1) __builtin_memcpy(&(DST), (SRC), 4); }

// Epilogue of one tile.  The MFMA leaves lane = position, 4 registers = 4 consecutive channels; a 4x4 dword
// transpose across each quad of lanes gives lane = channel, 4 registers = 4 consecutive columns of one row:
// 16-byte global accesses over 128-byte runs.  n = 0,1 are the wave's two 32-position blocks (rowpair: the same
// 32 columns of two adjacent rows).
template <int MT, int MODE>
__device__ __forceinline__ void epilogue_pp(const PpArgs& p, f32x16 (&acc)[MT][2], int img, int y0, int qwave,
                                            int nstride, int cob0, int l31, int half) {
  const ConvArgs& a = p.c;
  const float* __restrict__ g_bias = a.bias;
  const float* __restrict__ g_scale = a.scale;
  float* __restrict__ g_full = a.y_full;
  float* __restrict__ g_out = a.y_out;
  const int HW = a.H * a.W;
  const bool b0 = l31 & 1, b1 = l31 & 2;
  const int j = l31 & 3;
  constexpr bool FWD = MODE == EPI_FWD_FULL || MODE == EPI_FWD_BOTH || MODE == EPI_FWD_OUT || MODE == EPI_FWD_POOL;
  constexpr bool POOLM = MODE == EPI_FWD_POOL || MODE == EPI_DGRAD_ADDPOOL;
  constexpr bool HAS_LD = MODE == EPI_FWD_BOTH || MODE == EPI_FWD_OUT || MODE == EPI_DGRAD_ACT || MODE == EPI_DGRAD_ADD || MODE == EPI_FWD_POOL;
  const float* __restrict__ src = MODE == EPI_DGRAD_ACT ? a.act : a.skip;
  int nv[2], idx0[2], yrow0 = 0, ox0 = 0;
  bool ok0 = false;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int q4 = qwave + n * nstride + (l31 & ~3);
    const int tr = fdiv(q4, p.magic_wp), ox = q4 - tr * a.WP;
    const int y = y0 + tr;
    const bool ok = tr < a.R && y < a.H && ox < a.W;
    nv[n] = ok ? min(4, a.W - ox) : 0;
    idx0[n] = (ok ? ((img * a.Cout) * a.H + y) * a.W + ox : 0) + (cob0 + 4 * half + j) * HW;   // + (32m + 8g)*HW
    if (n == 0) { yrow0 = y; ox0 = ox; ok0 = ok; }
  }
  // pooled geometry (rowpair tiles only: n = 0/1 are rows y, y+1 with y even, same columns)
  const int Hp = a.H >> 1, Wp = a.W >> 1;
  const int npair = (POOLM && ok0) ? min(2, (a.W - ox0) >> 1) : 0;
  const int pidx0 = POOLM ? ((ok0 ? ((img * a.Cout) * Hp + (yrow0 >> 1)) * Wp + (ox0 >> 1) : 0) + (cob0 + 4 * half + j) * Hp * Wp) : 0;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    float bz[4], sc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = cob0 + 32 * m + 8 * g + 4 * half + j;
      bz[g] = FWD ? g_bias[ch] : 0.f;
      sc[g] = ((MODE == EPI_FWD_BOTH || MODE == EPI_FWD_POOL) && g_scale) ? g_scale[img * a.Cout + ch] : 1.f;
    }
    f32x4 t[2][4], u[2][4];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) u[n][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (HAS_LD) {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        // the branch on the valid count sits outside the load loop (a branch per load serialises them)
        if (nv[n] == 4) {
#pragma unroll
          for (int g = 0; g < 4; ++g) __builtin_memcpy(&u[n][g], src + idx0[n] + (32 * m + 8 * g) * HW, 16);
        } else if (nv[n] > 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g) PP_VEC_LD(u[n][g], src + idx0[n] + (32 * m + 8 * g) * HW, nv[n])
        }
      }
    }
    if (MODE == EPI_DGRAD_ADDPOOL) {
      // unpool(dout): window (row pair, column pair pc) sends its gradient to element arg = 2*row + col
      const float* __restrict__ g_din = p.q.pool_din;
      const unsigned char* __restrict__ g_mk = p.q.mask_in;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int pi = pidx0 + (32 * m + 8 * g) * Hp * Wp;
        float dg[2] = {0.f, 0.f};
        unsigned mk[2] = {0u, 0u};
        if (npair == 2) { __builtin_memcpy(dg, g_din + pi, 8); mk[0] = g_mk[pi]; mk[1] = g_mk[pi + 1]; }
        else if (npair == 1) { dg[0] = g_din[pi]; mk[0] = g_mk[pi]; }
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
          const int arg = (mk[pc] >> 4) & 3;
          u[0][g][2 * pc] = arg == 0 ? dg[pc] : 0.f;
          u[0][g][2 * pc + 1] = arg == 1 ? dg[pc] : 0.f;
          u[1][g][2 * pc] = arg == 2 ? dg[pc] : 0.f;
          u[1][g][2 * pc + 1] = arg == 3 ? dg[pc] : 0.f;
        }
      }
    }
    // (memory requests first, the register transposes of the accumulators travel under their latency)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        t[n][g] = quad_transpose4(acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3], b0, b1);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float z = t[n][g][i];
          if (FWD) {
            const float w_ = z + bz[g];
            z = w_ > 0.f ? w_ : w_ * a.slope;
            if (MODE == EPI_FWD_BOTH || MODE == EPI_FWD_POOL) u[n][g][i] = z * sc[g] + u[n][g][i];
            if (MODE == EPI_FWD_OUT) u[n][g][i] = z + u[n][g][i];
          } else if (MODE == EPI_DGRAD_ACT) {
            z *= (u[n][g][i] > 0.f) ? 1.f : a.slope;
          } else {
            z += u[n][g][i];
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
          upd(u[0][g][2 * pc], 0, mx, arg);
          upd(u[0][g][2 * pc + 1], 1, mx, arg);
          upd(u[1][g][2 * pc], 2, mx, arg);
          upd(u[1][g][2 * pc + 1], 3, mx, arg);
          po[pc] = mx;
          pm[pc] = (unsigned char)((t[0][g][2 * pc] > 0.f ? 1 : 0) | (t[0][g][2 * pc + 1] > 0.f ? 2 : 0) |
                                   (t[1][g][2 * pc] > 0.f ? 4 : 0) | (t[1][g][2 * pc + 1] > 0.f ? 8 : 0) | (arg << 4));
        }
        const int pi = pidx0 + (32 * m + 8 * g) * Hp * Wp;
        if (npair == 2) {
          __builtin_memcpy(g_pool + pi, po, 8);
          if (g_mk) { g_mk[pi] = pm[0]; g_mk[pi + 1] = pm[1]; }
        } else if (npair == 1) {
          g_pool[pi] = po[0];
          if (g_mk) g_mk[pi] = pm[0];
        }
      }
    } else {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
#define PP_ST(BYTES)                                                                               \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                  \
    const int idx_ = idx0[n] + (32 * m + 8 * g) * HW;                                              \
    if (MODE != EPI_FWD_OUT) __builtin_memcpy(g_full + idx_, &t[n][g], BYTES);                     \
    if (MODE == EPI_FWD_BOTH || MODE == EPI_FWD_OUT) __builtin_memcpy(g_out + idx_, &u[n][g], BYTES); \
  }
        if (nv[n] == 4) { PP_ST(16) } else if (nv[n] == 3) { PP_ST(12) } else if (nv[n] == 2) { PP_ST(8) } else if (nv[n] == 1) { PP_ST(4) }
#undef PP_ST
      }
    }
  }
}

typedef __attribute__((address_space(3))) void* lds_void_t;
typedef const __attribute__((address_space(1))) void* glb_void_t;

template <int MT, int VW, int MODE>
__global__ void __launch_bounds__(PTHR, 2)
k_conv3x3_x3_pp(const PpArgs p) {
  constexpr int NT = 2;
  constexpr int NBS = nbs_pp(VW);
  using VT = typename Vec<VW>::T;
  const ConvArgs& a = p.c;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MB = MT * 32;
  constexpr int A_UNITS = 9 * 2 * MB;                 // 16-byte units per weight array (hi or lo) per chunk
  constexpr int W_PIECES = 2 * A_UNITS / 64;          // 1-KiB LDS-DMA pieces per chunk (hi + lo)
  const int PT = p.PT, WP = a.WP;
  bf16x8* lds = reinterpret_cast<bf16x8*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction; tell the compiler
  const int grp = wid >> 2, w = wid & 3, gt = tid & (GTHR - 1);
  const int l31 = lane & 31, half = lane >> 5;
  bf16x8* Bbuf = lds + 4 * A_UNITS + grp * 4 * PT;    // this group's activation arrays: hi {k-half 0,1}, lo {k-half 0,1}
  const size_t HW = (size_t)a.H * a.W;
  const int nch = a.Cin / 16;

  // ---- this workgroup's tiles: one output-channel block, a contiguous range of (image, band) tiles.
  // blockIdx round-robins over the 8 XCDs: workgroups with equal blockIdx % 8 take neighbouring ranges, so the halo
  // rows a band shares with its neighbours and the weight panels are served by that XCD's L2.
  const int G = gridDim.x;
  int rank = blockIdx.x;
  if ((G & 7) == 0) rank = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int wg_per_mb = G / p.ncob;
  const int mb = rank / wg_per_mb, r_in = rank - mb * wg_per_mb;
  const int t_lo = (int)((long long)p.ntiles_mb * r_in / wg_per_mb);
  const int t_hi = (int)((long long)p.ntiles_mb * (r_in + 1) / wg_per_mb);
  const int nt_all = t_hi - t_lo;
  if (nt_all <= 0) return;                            // whole workgroup (uniform)
  const int n_g = (nt_all + 1 - grp) >> 1;            // group 0 takes tiles t_lo, t_lo+2, ..; group 1 the odd ones
  const int K = n_g * nch;                            // this group's MFMA segments
  const int K0 = ((nt_all + 1) >> 1) * nch;           // group 0's (>= group 1's)

  {  // zero the activation buffers once: halo / pad slots are never written again
    f32x4* z = reinterpret_cast<f32x4*>(lds + 4 * A_UNITS);
    for (int t = tid; t < 8 * PT; t += PTHR) z[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- activation staging geometry (per thread of the group): slot -> (k-half, tile row, column)
  int b_tr[NBS], b_xo[NBS], b_dst[NBS], b_src[NBS];   // b_tr < 0: no item; b_src < 0: row outside the image (zeros)
#pragma unroll
  for (int s = 0; s < NBS; ++s) {
    const int it = s * GTHR + gt;
    b_tr[s] = -1; b_xo[s] = 0; b_dst[s] = 0; b_src[s] = -1;
    if (it < 2 * p.p_in) {
      const int h = it & 1;                           // k-half fastest: 2-way instead of 4-way LDS store conflicts
      const int pp = it >> 1;
      const int tr = fdiv(pp, p.magic_w), ix = (pp - tr * (a.W / VW)) * VW;
      b_tr[s] = tr;
      b_xo[s] = 8 * h * a.H * a.W + ix;
      b_dst[s] = h * PT + tr * WP + 1 + ix;           // unit index inside the hi array pair; lo = +2*PT
    }
  }
  VT pb[NBS][8];
  int cur_img = 0, cur_y0 = 0;                        // tile being accumulated (epilogue coordinates)
  int st_img = 0, st_y0 = 0;                          // tile being staged

  // sources of tile number TI (of this group) for the staging slots
#define PP_TILE_SRC(TI)                                                                            \
  {                                                                                                \
    const int t_ = t_lo + 2 * (TI) + grp;                                                          \
    st_img = fdiv(t_, p.magic_bpi);                                                                \
    st_y0 = (t_ - st_img * p.bpi) * a.R;                                                           \
    _Pragma("unroll") for (int s_ = 0; s_ < NBS; ++s_) {                                           \
      const int y_ = st_y0 - 1 + b_tr[s_];                                                         \
      const bool ok_ = b_tr[s_] >= 0 && y_ >= 0 && y_ < a.H;                                       \
      b_src[s_] = ok_ ? (st_img * a.Cin * a.H + y_) * a.W + b_xo[s_] : -1;                         \
    }                                                                                              \
  }
#define PP_ISSUE_B(C16)                                                                            \
  {                                                                                                \
    const float* xs_ = a.x + (size_t)(C16) * CK16 * HW;                                            \
    _Pragma("unroll") for (int s_ = 0; s_ < NBS; ++s_) {                                           \
      const float* q_ = xs_ + max(b_src[s_], 0);                                                   \
      _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) pb[s_][j_] = *reinterpret_cast<const VT*>(q_ + j_ * HW); \
    }                                                                                              \
  }
#define PP_WRITE_B()                                                                               \
  {                                                                                                \
    _Pragma("unroll") for (int s_ = 0; s_ < NBS; ++s_) {                                           \
      if (b_tr[s_] >= 0) {                                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < VW; ++i_) {                                        \
          float f_[8];                                                                             \
          _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) f_[j_] = b_src[s_] >= 0 ? vget<VW>(pb[s_][j_], i_) : 0.f; \
          bf16x8 hi_, lo_;                                                                         \
          split8(f_, hi_, lo_);                                                                    \
          Bbuf[b_dst[s_] + i_] = hi_;                                                              \
          Bbuf[b_dst[s_] + i_ + 2 * PT] = lo_;                                                     \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
  }
  // weight panels of segment KS (chunk KS % nch) -> ring slot KS & 1, by LDS-DMA: 1 KiB per wave-instruction,
  // LDS destination = wave-uniform base + lane * 16 (linear), the per-lane SOURCE address carries the panel layout
  const int a_chunk_units = 9 * 2 * a.CoP;
#define PP_DMA_W(KS)                                                                               \
  {                                                                                                \
    const int c16_ = (KS) % nch;                                                                   \
    bf16x8* slot_ = lds + ((KS) & 1) * 2 * A_UNITS;                                                \
    _Pragma("unroll") for (int i_ = 0; i_ < (W_PIECES + 3) / 4; ++i_) {                            \
      const int piece_ = i_ * 4 + w;                                                               \
      if (piece_ < W_PIECES) {                                                                     \
        const int u_ = piece_ * 64 + lane;                                                         \
        const int lo_ = u_ >= A_UNITS ? 1 : 0;                                                     \
        const int r_ = u_ - lo_ * A_UNITS;                                                         \
        const int th_ = r_ / MB, co_ = r_ - th_ * MB;                                              \
        const bf16x8* src_ = (lo_ ? p.a_lo : p.a_hi) + (size_t)c16_ * a_chunk_units + th_ * a.CoP + mb * MB + co_; \
        __builtin_amdgcn_global_load_lds((glb_void_t)src_, (lds_void_t)(slot_ + piece_ * 64), 16, 0, 0); \
      }                                                                                            \
    }                                                                                              \
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  int tapoff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) tapoff[t] = (t / 3) * WP + (t % 3);

  const int wpr = WP >> 5;                            // waves per row (rowpair mode)
  const int qwave = p.rowpair ? ((w / wpr) * 2 * WP + (w - (w / wpr) * wpr) * 32) : w * 64;
  const int nstride = p.rowpair ? WP : 32;
  const int a_off = half * MB + l31;                  // + tap*2*MB + m*32 ; lo: + A_UNITS
  const int b_off = half * PT + qwave + l31;          // + tapoff + n*nstride ; lo: + 2*PT

  // ---- prologue: both groups stage their first chunk, group 0 fetches the first weight panel
  __syncthreads();                                    // zero fill complete
  if (K > 0) {
    PP_TILE_SRC(0)
    cur_img = st_img; cur_y0 = st_y0;
    PP_ISSUE_B(0)
  }
  if (grp == 0) PP_DMA_W(0)
  if (K > 0) PP_WRITE_B()
  __syncthreads();

  const int nphase = 2 * K0 + 1;
  for (int ph = 0; ph < nphase; ++ph) {
    const int d = ph - grp;
    if (d >= 0 && (d & 1) == 0) {
      // =========================== MFMA segment: chunk ks % nch of this group's current tile
      const int ks = d >> 1;
      if (ks < K) {
        const bf16x8* Aw = lds + (ks & 1) * 2 * A_UNITS + a_off;
        const bf16x8* Bw = Bbuf + b_off;
        bf16x8 ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) { ah[0][m] = Aw[m * 32]; al[0][m] = Aw[A_UNITS + m * 32]; }
#pragma unroll
        for (int n = 0; n < NT; ++n) { bh[0][n] = Bw[tapoff[0] + n * nstride]; bl[0][n] = Bw[2 * PT + tapoff[0] + n * nstride]; }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int cur = t & 1, nxt = cur ^ 1;
          if (t + 1 < 9) {                            // fragments of tap t+1 travel under the MFMAs of tap t
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              ah[nxt][m] = Aw[(t + 1) * 2 * MB + m * 32];
              al[nxt][m] = Aw[A_UNITS + (t + 1) * 2 * MB + m * 32];
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              bh[nxt][n] = Bw[tapoff[t + 1] + n * nstride];
              bl[nxt][n] = Bw[2 * PT + tapoff[t + 1] + n * nstride];
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][m], bl[cur][n], acc[m][n], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cur][m], bh[cur][n], acc[m][n], 0, 0, 0);
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][m], bh[cur][n], acc[m][n], 0, 0, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else if (d >= 0) {
      // =========================== memory segment: everything this group needs before its segment kn
      const int kn = (d + 1) >> 1;
      const int cn = kn % nch;
      if (kn < K) {
        if (cn == 0) PP_TILE_SRC(kn / nch)
        PP_ISSUE_B(cn)
      }
      if (grp == 0 && kn < K0) PP_DMA_W(kn)           // also serves group 1's segment kn, one phase later
      if (cn == 0 && kn >= nch && kn <= K) {          // the tile finished by segment kn-1
        epilogue_pp<MT, MODE>(p, acc, cur_img, cur_y0, qwave, nstride, mb * MB, l31, half);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
      }
      if (kn < K) {
        if (cn == 0) { cur_img = st_img; cur_y0 = st_y0; }
        PP_WRITE_B()
      }
    }
    __syncthreads();
  }
}

template <int MT>
int launch_pp(const PpArgs& p, int VW, size_t lds, dim3 grid, hipStream_t st) {
  int rc = FDET_OK;
  auto go = [&](auto kern) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      rc = fail(FDET_ELAUNCH, "conv3x3_bf16x3(pp): cannot reserve %zu bytes of LDS", lds);
      return;
    }
    hipLaunchKernelGGL(kern, grid, dim3(PTHR), lds, st, p);
  };
#define PP_MODES(V_)                                                                               \
  switch (p.c.mode) {                                                                              \
    case EPI_FWD_FULL: go(k_conv3x3_x3_pp<MT, V_, EPI_FWD_FULL>); break;                           \
    case EPI_FWD_BOTH: go(k_conv3x3_x3_pp<MT, V_, EPI_FWD_BOTH>); break;                           \
    case EPI_FWD_OUT: go(k_conv3x3_x3_pp<MT, V_, EPI_FWD_OUT>); break;                             \
    case EPI_DGRAD_ACT: go(k_conv3x3_x3_pp<MT, V_, EPI_DGRAD_ACT>); break;                         \
    case EPI_DGRAD_ADD: go(k_conv3x3_x3_pp<MT, V_, EPI_DGRAD_ADD>); break;                         \
    case EPI_FWD_POOL: go(k_conv3x3_x3_pp<MT, V_, EPI_FWD_POOL>); break;                           \
    case EPI_DGRAD_ADDPOOL: go(k_conv3x3_x3_pp<MT, V_, EPI_DGRAD_ADDPOOL>); break;                 \
    default: return fail(FDET_EINVAL, "conv3x3_bf16x3(pp): internal: mode %d", p.c.mode);          \
  }
  if (VW == 4) { PP_MODES(4) } else if (VW == 2) { PP_MODES(2) } else { PP_MODES(1) }
#undef PP_MODES
  if (rc != FDET_OK) return rc;
  return check_launch("fdet_conv3x3_bf16x3(pp)");
}

int pp_num_cus() {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return ncu;
}

}  // namespace

// Returns 1 when this kernel has no tiling / epilogue for the request (the caller then uses the round-1 kernels),
// else the launch status.  `a` arrives with N/Cin/Cout/H/W, pointers, dgrad, slope set; `q` all null for plain modes.
int fdet_x3_pp_run(ConvArgs a, PoolArgs q, hipStream_t st) {
  static const bool disabled = [] { const char* e = getenv("FDET_CONV_PP"); return e && e[0] == '0'; }();
  const bool pooled = q.pool_out || q.pool_din;
  if (disabled && !pooled) return 1;
  if (a.W > 63 || a.Cout % 32 != 0 || a.Cin % 16 != 0) return pooled ? fail(FDET_EINVAL, "conv3x3 pooled epilogue: needs W <= 63, Cout %% 32 == 0 (W=%d Cout=%d)", a.W, a.Cout) : 1;
  a.WP = (a.W + 1 + 3) / 4 * 4;           // pitch % 4 == 0: a quad of positions never straddles two rows
  if (pooled) a.WP = a.W <= 31 ? 32 : 64; // whole 2x2 windows per wave need the row-pair mapping (pitch 32 or 64)
  auto aligned = [](const void* ptr, size_t b) { return ((uintptr_t)ptr % b) == 0; };
  const int VW = (a.W % 4 == 0 && aligned(a.x, 16)) ? 4 : ((a.W % 2 == 0 && aligned(a.x, 8)) ? 2 : 1);
  if ((size_t)a.N * std::max(a.Cin, a.Cout) * a.H * a.W >= (size_t)1 << 31) return pooled ? fail(FDET_EINVAL, "conv3x3 pooled epilogue: tensor too large for 32-bit indexing") : 1;
  a.CoP = (a.Cout + 31) / 32 * 32;
  a.mode = -1;
  if (q.pool_out) {
    if (!a.dgrad && a.bias && a.skip && !a.y_full && !a.y_out) a.mode = EPI_FWD_POOL;
  } else if (q.pool_din) {
    if (a.dgrad && q.mask_in && !a.act && !a.skip) a.mode = EPI_DGRAD_ADDPOOL;
  } else if (!a.dgrad && a.bias) {
    if (a.y_full && !a.y_out) a.mode = EPI_FWD_FULL;
    else if (a.y_full && a.y_out && a.skip && a.scale) a.mode = EPI_FWD_BOTH;
    else if (!a.y_full && a.y_out && a.skip && !a.scale) a.mode = EPI_FWD_OUT;
  } else if (a.dgrad) {
    if (a.act && !a.skip) a.mode = EPI_DGRAD_ACT;
    else if (!a.act && a.skip) a.mode = EPI_DGRAD_ADD;
  }
  if (a.mode < 0) return pooled ? fail(FDET_EINVAL, "conv3x3 pooled epilogue: inconsistent pointer set") : 1;
  const int MT = (a.CoP % 64 == 0) ? 2 : 1;
  const int cap = 256;
  int R = cap / a.WP;
  const bool rowpair = (a.WP == 32 || a.WP == 64);
  if (pooled && (!rowpair || (a.H & 1) || (a.W & 1))) return fail(FDET_EINVAL, "conv3x3 pooled epilogue: needs even H and W <= 62 (H=%d W=%d)", a.H, a.W);
  if (R > a.H) R = rowpair ? ((a.H + 1) & ~1) : a.H;
  if (2 * (R + 2) * (a.W / VW) > nbs_pp(VW) * GTHR) return pooled ? fail(FDET_EINVAL, "conv3x3 pooled epilogue: staging slots") : 1;
  a.R = R;
  PpArgs p;
  p.q = q;
  p.PT = cap + 2 * a.WP + 3;
  p.p_in = (a.R + 2) * (a.W / VW);
  p.magic_w = magic_of(a.W / VW);
  p.magic_wp = magic_of(a.WP);
  p.bpi = (a.H + a.R - 1) / a.R;
  p.magic_bpi = magic_of(p.bpi);
  p.ntiles_mb = a.N * p.bpi;
  if (p.ntiles_mb >= (1 << 20)) return pooled ? fail(FDET_EINVAL, "conv3x3 pooled epilogue: too many tiles") : 1;
  p.rowpair = rowpair ? 1 : 0;
  p.ncob = a.CoP / (MT * 32);
  const size_t units = (size_t)(a.Cin / 16) * 9 * 2 * a.CoP;       // per hi / lo half
  p.a_hi = reinterpret_cast<const bf16x8*>(a.wpk);
  p.a_lo = p.a_hi + units;
  a.VR = 0; a.nbands = p.bpi; a.magic_h1 = 0; a.stagger = 0;
  p.c = a;
  const size_t lds = (size_t)(4 * 9 * 2 * MT * 32 + 8 * p.PT) * 16;
  if (lds > 160 * 1024) return pooled ? fail(FDET_EINVAL, "conv3x3 pooled epilogue: LDS") : 1;
  // one workgroup per CU, each with >= 2 tiles where the problem has them; a multiple of ncob (and of 8 when possible)
  const int ncu = pp_num_cus();
  int per_mb = std::min(ncu / p.ncob, (p.ntiles_mb + 1) / 2);
  if (per_mb < 1) per_mb = 1;
  int g = per_mb * p.ncob;
  if (g > 8) g = g / 8 * 8;
  if (g % p.ncob) g = g / p.ncob * p.ncob;
  if (g < p.ncob) g = p.ncob;
  dim3 grid(g, 1);
  return MT == 2 ? launch_pp<2>(p, VW, lds, grid, st) : launch_pp<1>(p, VW, lds, grid, st);
}

// ---- pooled residual block, backward of the tail: dz2 = unpool(dout) * drop_scale * lrelu'(c), everything read
// from the pooled gradient and the routing bytes written by EPI_FWD_POOL (c and the block input are not needed).
namespace {
__global__ void __launch_bounds__(256)
k_pool_route_bwd(const float* __restrict__ dout, const unsigned char* __restrict__ mask, const float* __restrict__ scale,
                 float* __restrict__ dz2, int NF, int Hp, int Wp, float slope) {
  const int W = 2 * Wp;
  const size_t total = (size_t)NF * Hp * (Wp >> 1);             // one thread per TWO windows (16-byte row stores)
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int oxp = (int)(t % (Wp >> 1));
    const size_t r = t / (Wp >> 1);
    const int oy = (int)(r % Hp);
    const size_t nf = r / Hp;
    const float sc = scale ? scale[nf] : 1.f;
    const size_t pi = (nf * Hp + oy) * Wp + 2 * oxp;
    const float g0 = dout[pi], g1 = dout[pi + 1];
    const unsigned m0 = mask[pi], m1 = mask[pi + 1];
    float row0[4], row1[4];
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      const unsigned mk = pc ? m1 : m0;
      const float g = (pc ? g1 : g0) * sc;
      const int arg = (mk >> 4) & 3;
      row0[2 * pc] = arg == 0 ? g * ((mk & 1) ? 1.f : slope) : 0.f;
      row0[2 * pc + 1] = arg == 1 ? g * ((mk & 2) ? 1.f : slope) : 0.f;
      row1[2 * pc] = arg == 2 ? g * ((mk & 4) ? 1.f : slope) : 0.f;
      row1[2 * pc + 1] = arg == 3 ? g * ((mk & 8) ? 1.f : slope) : 0.f;
    }
    float* o = dz2 + (nf * (2 * Hp) + 2 * (size_t)oy) * W + 4 * (size_t)oxp;
    __builtin_memcpy(o, row0, 16);
    __builtin_memcpy(o + W, row1, 16);
  }
}
// odd pooled width: the last window of a row has no partner
__global__ void __launch_bounds__(256)
k_pool_route_bwd_last(const float* __restrict__ dout, const unsigned char* __restrict__ mask, const float* __restrict__ scale,
                      float* __restrict__ dz2, int NF, int Hp, int Wp, float slope) {
  const int W = 2 * Wp;
  const size_t total = (size_t)NF * Hp;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int oy = (int)(t % Hp);
    const size_t nf = t / Hp;
    const float sc = scale ? scale[nf] : 1.f;
    const size_t pi = (nf * Hp + oy) * Wp + (Wp - 1);
    const unsigned mk = mask[pi];
    const float g = dout[pi] * sc;
    const int arg = (mk >> 4) & 3;
    float* o = dz2 + (nf * (2 * Hp) + 2 * (size_t)oy) * W + 2 * (size_t)(Wp - 1);
    o[0] = arg == 0 ? g * ((mk & 1) ? 1.f : slope) : 0.f;
    o[1] = arg == 1 ? g * ((mk & 2) ? 1.f : slope) : 0.f;
    o[W] = arg == 2 ? g * ((mk & 4) ? 1.f : slope) : 0.f;
    o[W + 1] = arg == 3 ? g * ((mk & 8) ? 1.f : slope) : 0.f;
  }
}
}  // namespace

extern "C" int fdet_pool_route_bwd(const float* dout, const unsigned char* mask, const float* drop_scale, float* dz2,
                                   int N, int F, int H, int W, float slope, void* stream) {
  FDET_REQUIRE(dout && mask && dz2 && N > 0 && F > 0 && H >= 2 && W >= 2 && !(H & 1) && !(W & 1),
               "pool_route_bwd: bad arguments (even H, W >= 2 required; H=%d W=%d)", H, W);
  const int Hp = H / 2, Wp = W / 2;
  const size_t total = (size_t)N * F * Hp * (Wp >> 1);
  if (total) {
    size_t blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_pool_route_bwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dout, mask, drop_scale,
                       dz2, N * F, Hp, Wp, slope);
  }
  if (Wp & 1) {
    size_t blocks = ((size_t)N * F * Hp + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_pool_route_bwd_last, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dout, mask,
                       drop_scale, dz2, N * F, Hp, Wp, slope);
  }
  return check_launch("fdet_pool_route_bwd");
}
