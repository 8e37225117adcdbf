// bf16x3 3x3 conv forward / data gradient, small-tile variant: one band of <= 256 positions per
// workgroup, ONE LDS buffer (<= 80 KB) and <= 256 registers, so two workgroups share a CU and one's
// staging / epilogue runs beside the other's MFMAs.  Fastest for rows of <= 64 columns (the
// PoolResnet resolutions); anything it has no tiling for goes to the general persistent kernel
// (fdet_conv3x3_x3_kernel.inc).  Same arithmetic, LDS layout and epilogue fusion modes as described there.
//
// Tried and dropped (round 1): the woven one-workgroup-per-CU pipeline that pays for the weight gradient and
// the stem (two LDS chunk buffers, one staging job / asm buffer load per group of four MFMAs, counted
// vmcnt).  Bit-identical results, no gain at 60x60 (fwd 0.31 vs 0.25 ms per launch in tools/probe/conv_dbg.py
// terms): ablation gave MFMA loop alone 0.18, + staging jobs and loads +0.085, + epilogue +0.055 -- every
// part ADDS even when woven (the LDS store path and the wave's single issue stream are shared with the
// fragment reads), and only a second co-resident workgroup overlaps the epilogue.
#include "fdet_conv3x3_x3.h"
#include "fdet_conv3x3_x3_epi.h"
#include <algorithm>
#include <cstdint>

using namespace fdet;

namespace {

// B staging slots per thread (8 loads of VW floats each): 2*(R+2)*(W/VW) <= nbs*NTHR is checked on the host
__host__ __device__ constexpr int nbs_sb(int nt, int vw) { return vw == 4 ? 1 : (vw == 2 ? 2 : 3); }

struct X3SbArgs {
  int ntiles, ncob;            // persistent: tiles = nbands * ncob, walked with stride gridDim.x
  ConvArgs c;                // x, bias, epilogue pointers, geometry (WP, R, VR, nbands, mode ...)
  const bf16x8* a_hi;        // [Cin/16][9][2][CoP] x 8 bf16
  const bf16x8* a_lo;
  int PT;                    // positions per B array (cap + 2*WP + 3)
  int p_in;                  // (R+2)*W staged positions per k-half
  unsigned magic_w;
  // aligned-band variant (AL): tiles are (image, band of R rows), a wave owns two rows x 32 columns
  PoolArgs q;
  unsigned magic_wp;
  int bpi;                   // bands per image
  unsigned magic_bpi;
};

// Single LDS buffer (<= 80 KB, <= 256 registers): two workgroups share a CU, and one's staging /
// epilogue runs beside the other's MFMAs.
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// 4x4 dword transpose across each quad of lanes: register i of lane j <-> register j of lane i
__device__ __forceinline__ void quad_transpose4(float (&v)[4], bool b0, bool b1) {
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
}

// Epilogue with 16-byte global accesses.  The MFMA leaves lane = position, 4 registers = 4
// consecutive channels; a 4x4 dword transpose across each quad of lanes gives lane = channel,
// 4 registers = 4 consecutive positions of one row (WP % 4 == 0, so a quad never straddles rows):
// 16 memory instructions per tensor instead of 64, each covering 128-byte runs.  Loads of all
// tiles first, then arithmetic, then stores; the branch on the valid count sits outside the loops.
template <int MT, int NT, int MODE>
__device__ __forceinline__ void epilogue_sb(const ConvArgs& a, f32x16 (&acc)[MT][NT], int v0, int qwave, int cob0,
                                            int l31, int half) {
  const float* __restrict__ g_bias = a.bias;
  const float* __restrict__ g_skip = a.skip;
  const float* __restrict__ g_scale = a.scale;
  const float* __restrict__ g_act = a.act;
  float* __restrict__ g_full = a.y_full;
  float* __restrict__ g_out = a.y_out;
  const int H1 = a.H + 1, HW = a.H * a.W;
  const bool b0 = l31 & 1, b1 = l31 & 2;
  const int j = l31 & 3;
  constexpr bool FWD = MODE == EPI_FWD_FULL || MODE == EPI_FWD_BOTH || MODE == EPI_FWD_OUT;
  constexpr bool HAS_LD = MODE != EPI_FWD_FULL;
  const float* __restrict__ src = MODE == EPI_DGRAD_ACT ? g_act : g_skip;
  int nv[NT], idx0[NT], img[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int q4 = qwave + n * 32 + (l31 & ~3);
    const int tr = q4 / a.WP, ox = q4 - tr * a.WP;
    const int v = v0 + tr;
    const int im = fdiv(v, a.magic_h1), oy = v - im * H1 - 1;
    const bool ok = tr < a.R && v < a.VR && oy >= 0 && ox < a.W;
    nv[n] = ok ? min(4, a.W - ox) : 0;
    img[n] = ok ? im : 0;
    idx0[n] = (ok ? ((im * a.Cout) * a.H + oy) * a.W + ox : 0) + (cob0 + 4 * half + j) * HW;   // + (32m + 8g)*HW
  }
  float bz[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) bz[m][g] = FWD ? g_bias[cob0 + 32 * m + 8 * g + 4 * half + j] : 0.f;
  // one 32-position block (n) at a time: transposed accumulators t[m][g][i] = channel
  // cob0 + 32m + 8g + 4half + j at position q4 + i; loads of the block, arithmetic, stores
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    f32x4 t[MT][4], u[MT][4];
    float sc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v_[4] = {acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
        quad_transpose4(v_, b0, b1);
        t[m][g] = f32x4{v_[0], v_[1], v_[2], v_[3]};
        u[m][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        sc[m][g] = MODE == EPI_FWD_BOTH ? g_scale[img[n] * a.Cout + cob0 + 32 * m + 8 * g + 4 * half + j] : 1.f;
      }
#define SB_LD(BYTES)                                                                               \
  _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int g = 0; g < 4; ++g)     \
    __builtin_memcpy(&u[m][g], src + idx0[n] + (32 * m + 8 * g) * HW, BYTES);
    if (HAS_LD) {
      if (nv[n] == 4) { SB_LD(16) } else if (nv[n] == 3) { SB_LD(12) } else if (nv[n] == 2) { SB_LD(8) } else if (nv[n] == 1) { SB_LD(4) }
    }
#undef SB_LD
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float z = t[m][g][i];
          if (FWD) {
            const float w_ = z + bz[m][g];
            z = w_ > 0.f ? w_ : w_ * a.slope;
            if (MODE == EPI_FWD_BOTH) u[m][g][i] = z * sc[m][g] + u[m][g][i];
            if (MODE == EPI_FWD_OUT) u[m][g][i] = z + u[m][g][i];
          } else if (MODE == EPI_DGRAD_ACT) {
            z *= (u[m][g][i] > 0.f) ? 1.f : a.slope;
          } else {
            z += u[m][g][i];
          }
          t[m][g][i] = z;
        }
#define SB_ST(BYTES)                                                                               \
  _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int g = 0; g < 4; ++g) {   \
    const int idx_ = idx0[n] + (32 * m + 8 * g) * HW;                                              \
    if (MODE != EPI_FWD_OUT) __builtin_memcpy(g_full + idx_, &t[m][g], BYTES);                     \
    if (MODE == EPI_FWD_BOTH || MODE == EPI_FWD_OUT) __builtin_memcpy(g_out + idx_, &u[m][g], BYTES); \
  }
    if (nv[n] == 4) { SB_ST(16) } else if (nv[n] == 3) { SB_ST(12) } else if (nv[n] == 2) { SB_ST(8) } else if (nv[n] == 1) { SB_ST(4) }
#undef SB_ST
  }
}

// AL (aligned bands): tiles are bands of R rows of ONE image (halo rows outside the image are zeros, no separator rows),
// a wave owns two adjacent rows x 32 columns (NT == 2), and the epilogue is the shared one of fdet_conv3x3_x3_epi.h,
// which holds whole 2x2 pooling windows per lane: the pooled-block modes EPI_FWD_POOL / EPI_DGRAD_ADDPOOL.
template <int MT, int NT, int VW, int MODE, bool AL = false>
__global__ void __launch_bounds__(NTHR, 2)
k_conv3x3_x3_sb(const X3SbArgs p) {
  constexpr int NBS = nbs_sb(NT, VW);
  using VT = typename Vec<VW>::T;
  const ConvArgs& a = p.c;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MB = MT * 32;
  constexpr int A_UNITS = 9 * 2 * MB;                 // 16-byte units per A array per chunk
  constexpr int NA = (2 * A_UNITS + NTHR - 1) / NTHR; // hi and lo
  const int PT = p.PT, WP = a.WP;
  const int buf_units = 2 * A_UNITS + 4 * PT;         // 16-byte units per buffer
  bf16x8* lds = reinterpret_cast<bf16x8*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  // XCD-aware persistent walk: blockIdx round-robins over the 8 XCDs (private L2 each), so workgroups with the same
  // blockIdx % 8 take one contiguous eighth of the tiles, neighbours in it at the same time -- the two halo rows a
  // band shares with the next one are then re-read from that XCD's L2 instead of HBM.
  int tile = blockIdx.x, tend = p.ntiles, tstep = gridDim.x;
  if (gridDim.x % 8 == 0) {
    const int grp = blockIdx.x & 7, chunk = (p.ntiles + 7) >> 3;
    tile = grp * chunk + (blockIdx.x >> 3);
    tend = min((grp + 1) * chunk, p.ntiles);
    tstep = gridDim.x >> 3;
    if (tile >= tend) tile = tend = p.ntiles;          // no tile for this workgroup
  }
  int mb = tile % p.ncob;
  int v0 = (tile / p.ncob) * a.R;                     // AL: (image, first row) of the band, see X3_TILE_POS
  int t_img = 0;
#define X3_TILE_POS(T)                                                                             \
  {                                                                                                \
    mb = (T) % p.ncob;                                                                             \
    const int bt_ = (T) / p.ncob;                                                                  \
    if (AL) { t_img = fdiv(bt_, p.magic_bpi); v0 = (bt_ - t_img * p.bpi) * a.R; }                  \
    else v0 = bt_ * a.R;                                                                           \
  }
  if (tile < tend) X3_TILE_POS(tile)
  const int H1 = a.H + 1;
  const size_t HW = (size_t)a.H * a.W;

  {  // zero both buffers: halo positions of the B arrays are never written again
    f32x4* z = reinterpret_cast<f32x4*>(smem);
    for (int t = tid; t < buf_units; t += NTHR) z[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- B staging geometry: slot -> (k-half, tile row, column); the source depends on the tile
  int b_tr[NBS], b_xo[NBS], b_dst[NBS], b_src[NBS];   // b_tr < 0: no item; b_src < 0: row outside the batch (zeros)
#pragma unroll
  for (int s = 0; s < NBS; ++s) {
    const int it = s * NTHR + tid;
    b_tr[s] = -1; b_xo[s] = 0; b_dst[s] = 0; b_src[s] = -1;
    if (it < 2 * p.p_in) {
      // k-half fastest: neighbouring lanes write LDS units 3 (mod 8) apart instead of every lane 4 apart --
      // the 16-byte B stores of a wave fall on twice as many banks (4-way -> 2-way conflicts)
      const int h = it & 1;
      const int pp = it >> 1;
      const int tr = fdiv(pp, p.magic_w), ix = (pp - tr * (a.W / VW)) * VW;
      b_tr[s] = tr;
      b_xo[s] = 8 * h * a.H * a.W + ix;
      b_dst[s] = h * PT + tr * WP + 1 + ix;           // unit index inside the hi array pair; lo = +2*PT
    }
  }
#define X3_TILE_SRC(V0)                                                                            \
  {                                                                                                \
    _Pragma("unroll") for (int s_ = 0; s_ < NBS; ++s_) {                                           \
      const int v_ = (V0) - 1 + b_tr[s_];                                                          \
      if (AL) {                                                                                    \
        const bool ok_ = b_tr[s_] >= 0 && v_ >= 0 && v_ < a.H;                                     \
        b_src[s_] = ok_ ? (t_img * a.Cin * a.H + v_) * a.W + b_xo[s_] : -1;                        \
      } else {                                                                                     \
        const int n_ = fdiv(max(v_, 0), a.magic_h1), yy_ = v_ - n_ * H1 - 1;                       \
        const bool ok_ = b_tr[s_] >= 0 && v_ >= 0 && v_ < a.VR && yy_ >= 0;                        \
        b_src[s_] = ok_ ? (n_ * a.Cin * a.H + yy_) * a.W + b_xo[s_] : -1;                          \
      }                                                                                            \
    }                                                                                              \
  }
  X3_TILE_SRC(v0)
  const int a_chunk_units = 9 * 2 * a.CoP;            // units per chunk per array in HBM
  bf16x8 pa[NA];
  VT pb[NBS][8];
#define X3_ISSUE_LOADS(C16)                                                                        \
  {                                                                                                \
    _Pragma("unroll") for (int s_ = 0; s_ < NA; ++s_) {                                            \
      const int u_ = min(tid + s_ * NTHR, 2 * A_UNITS - 1);                                        \
      const int lo_ = u_ >= A_UNITS ? 1 : 0;                                                       \
      const int r_ = u_ - lo_ * A_UNITS;                                                           \
      const int th_ = r_ / MB, co_ = r_ - th_ * MB;                                                \
      const bf16x8* src_ = (lo_ ? p.a_lo : p.a_hi) + (size_t)(C16) * a_chunk_units + th_ * a.CoP + mb * MB + co_; \
      pa[s_] = *src_;                                                                              \
    }                                                                                              \
    const float* xs_ = a.x + (size_t)(C16) * CK16 * HW;                                            \
    _Pragma("unroll") for (int s_ = 0; s_ < NBS; ++s_) {                                           \
      const float* q_ = xs_ + max(b_src[s_], 0);                                                   \
      _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) pb[s_][j_] = *reinterpret_cast<const VT*>(q_ + j_ * HW); \
    }                                                                                              \
  }
#define X3_WRITE_LDS(BUF)                                                                          \
  {                                                                                                \
    bf16x8* buf_ = (BUF);                                                                          \
    _Pragma("unroll") for (int s_ = 0; s_ < NA; ++s_) {                                            \
      const int u_ = tid + s_ * NTHR;                                                              \
      if (u_ < 2 * A_UNITS) buf_[u_] = pa[s_];                                                     \
    }                                                                                              \
    bf16x8* B_ = buf_ + 2 * A_UNITS;                                                               \
    _Pragma("unroll") for (int s_ = 0; s_ < NBS; ++s_) {                                           \
      if (b_tr[s_] >= 0) {                                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < VW; ++i_) {                                        \
          float f_[8];                                                                             \
          _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) f_[j_] = b_src[s_] >= 0 ? vget<VW>(pb[s_][j_], i_) : 0.f; \
          bf16x8 hi_, lo_;                                                                         \
          split8(f_, hi_, lo_);                                                                    \
          B_[b_dst[s_] + i_] = hi_;                                                                \
          B_[b_dst[s_] + i_ + 2 * PT] = lo_;                                                       \
        }                                                                                          \
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

  const int wpr = WP >> 5;                                     // AL: waves per row pair (WP is 32 or 64)
  const int qwave = AL ? ((wid / wpr) * 2 * WP + (wid - (wid / wpr) * wpr) * 32) : wid * NT * 32;
  const int nstride = AL ? WP : 32;                            // AL: n = 0,1 are the same columns of two adjacent rows
  const int a_off = half * MB + l31;                           // + tap*2*MB + m*32 ; lo: + A_UNITS
  const int b_off = 2 * A_UNITS + half * PT + qwave + l31;     // + tapoff + n*nstride ; lo: + 2*PT

  X3_ISSUE_LOADS(0)
  __syncthreads();                       // zero fill complete

  const int nch = a.Cin / CK16;
  bool first = true;
  for (; tile < tend; tile += tstep) {
  const int cur_v0 = v0, cur_mb = mb, cur_img = t_img;    // the tile whose chunks are consumed below
  for (int c = 0; c < nch; ++c) {
    const bf16x8* buf = lds;
    if (!first) __syncthreads();         // every wave is done reading the previous chunk
    first = false;
    X3_WRITE_LDS(lds)
    __syncthreads();
    if (c + 1 < nch) {
      X3_ISSUE_LOADS(c + 1)
    } else if (tile + tstep < tend) {
      // the next tile's first chunk travels during this tile's last MFMA block and its epilogue
      const int nt_ = tile + tstep;
      X3_TILE_POS(nt_)
      X3_TILE_SRC(v0)
      X3_ISSUE_LOADS(0)
    }
    const bf16x8* Aw = buf + a_off;
    const bf16x8* Bw = buf + b_off;
    // one fragment set: while this wave waits for its LDS reads the partner wave of the other
    // workgroup owns the matrix pipe
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) ah[m] = Aw[t * 2 * MB + m * 32];
#pragma unroll
      for (int n = 0; n < NT; ++n) bl[n] = Bw[2 * PT + tapoff[t] + n * nstride];
#pragma unroll
      for (int m = 0; m < MT; ++m) al[m] = Aw[A_UNITS + t * 2 * MB + m * 32];
#pragma unroll
      for (int n = 0; n < NT; ++n) bh[n] = Bw[tapoff[t] + n * nstride];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: fast modes through the quad transpose (16-byte accesses), GENERIC through the
  //      channel-per-register epilogue shared with the fp32 kernel
  if (AL) {
    if constexpr (NT == 2) {
      const EpiGeo eg = epi_geometry<MODE>(p, cur_img, cur_v0, qwave, nstride, cur_mb * MB, l31, half);
      f32x4 u[MT][2][4];
      float dg[MT][4][2], bzm[MT][4], scm[MT][4];
      unsigned mk[MT][4];
      epi_loads<MT, MODE>(p, eg, u, dg, mk, bzm, scm, cur_img, cur_mb * MB, l31, half);
      epi_finish<MT, MODE>(p, eg, acc, u, dg, mk, bzm, scm, cur_img, cur_mb * MB, l31, half);
    }
  } else if (MODE != EPI_GENERIC) {
    epilogue_sb<MT, NT, MODE>(a, acc, cur_v0, qwave, cur_mb * MB, l31, half);
  } else {
    const int qlimit = a.R * WP;
    bool okn[NT];
    size_t basen[NT];
    int imgn[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int q = qwave + n * 32 + l31;
      const int tr = q / WP, ox = q - tr * WP;
      const int v = cur_v0 + tr;
      const int img = v / H1, oy = v - img * H1 - 1;
      okn[n] = (q < qlimit) && (ox < a.W) && (v < a.VR) && (oy >= 0);
      basen[n] = ((size_t)img * a.Cout * a.H + oy) * a.W + ox;
      imgn[n] = img;
    }
    epilogue<MT, NT, EPI_GENERIC>(a, acc, okn, basen, imgn, cur_mb * MB + 4 * half, HW);
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  }                                      // tile loop
}

template <int MT, int NT>
int launch_sb(const X3SbArgs& p, int VW, size_t lds, dim3 grid, hipStream_t st) {
  int rc = FDET_OK;
  auto go = [&](auto kern) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      rc = fail(FDET_ELAUNCH, "conv3x3_bf16x3(sb): cannot reserve %zu bytes of LDS", lds);
      return;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NTHR), lds, st, p);
  };
#define SB_MODES(V_)                                                                               \
  switch (p.c.mode) {                                                                              \
    case EPI_FWD_FULL: go(k_conv3x3_x3_sb<MT, NT, V_, EPI_FWD_FULL>); break;                       \
    case EPI_FWD_BOTH: go(k_conv3x3_x3_sb<MT, NT, V_, EPI_FWD_BOTH>); break;                       \
    case EPI_FWD_OUT: go(k_conv3x3_x3_sb<MT, NT, V_, EPI_FWD_OUT>); break;                         \
    case EPI_DGRAD_ACT: go(k_conv3x3_x3_sb<MT, NT, V_, EPI_DGRAD_ACT>); break;                     \
    case EPI_DGRAD_ADD: go(k_conv3x3_x3_sb<MT, NT, V_, EPI_DGRAD_ADD>); break;                     \
    default: go(k_conv3x3_x3_sb<MT, NT, V_, EPI_GENERIC>); break;                                  \
  }
  if (VW == 4) { SB_MODES(4) } else if (VW == 2) { SB_MODES(2) } else { SB_MODES(1) }
#undef SB_MODES
  if (rc != FDET_OK) return rc;
  return check_launch("fdet_conv3x3_bf16x3(sb)");
}

// aligned-band variant: the pooled-block modes
template <int MT>
int launch_sb_pool(const X3SbArgs& p, int VW, size_t lds, dim3 grid, hipStream_t st) {
  int rc = FDET_OK;
  auto go = [&](auto kern) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      rc = fail(FDET_ELAUNCH, "conv3x3_bf16x3(sb, pooled): cannot reserve %zu bytes of LDS", lds);
      return;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NTHR), lds, st, p);
  };
#define SB_PMODES(V_)                                                                              \
  switch (p.c.mode) {                                                                              \
    case EPI_FWD_POOL: go(k_conv3x3_x3_sb<MT, 2, V_, EPI_FWD_POOL, true>); break;                  \
    case EPI_DGRAD_ADDPOOL: go(k_conv3x3_x3_sb<MT, 2, V_, EPI_DGRAD_ADDPOOL, true>); break;        \
    case EPI_FWD_FULL: go(k_conv3x3_x3_sb<MT, 2, V_, EPI_FWD_FULL, true>); break;                  \
    case EPI_FWD_BOTH: go(k_conv3x3_x3_sb<MT, 2, V_, EPI_FWD_BOTH, true>); break;                  \
    case EPI_FWD_OUT: go(k_conv3x3_x3_sb<MT, 2, V_, EPI_FWD_OUT, true>); break;                    \
    case EPI_DGRAD_ACT: go(k_conv3x3_x3_sb<MT, 2, V_, EPI_DGRAD_ACT, true>); break;                \
    default: go(k_conv3x3_x3_sb<MT, 2, V_, EPI_DGRAD_ADD, true>); break;                           \
  }
  if (VW == 4) { SB_PMODES(4) } else if (VW == 2) { SB_PMODES(2) } else { SB_PMODES(1) }
#undef SB_PMODES
  if (rc != FDET_OK) return rc;
  return check_launch("fdet_conv3x3_bf16x3(sb, pooled)");
}

int sb_num_cus() {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return ncu;
}

}  // namespace

// Returns 1 when this kernel family has no tiling for the shape (the caller then uses the general
// persistent kernel), else the launch status.  `a` arrives with N/Cin/Cout/H/W, pointers, dgrad, slope set.
int fdet_x3_sb_run(ConvArgs a, hipStream_t st) {
  a.WP = (a.W + 1 + 3) / 4 * 4;           // pitch % 4 == 0: a quad of positions never straddles two rows
  auto aligned = [](const void* q, size_t b) { return ((uintptr_t)q % b) == 0; };
  const int VW = (a.W % 4 == 0 && aligned(a.x, 16)) ? 4 : ((a.W % 2 == 0 && aligned(a.x, 8)) ? 2 : 1);
  a.VR = a.N * (a.H + 1) + 1;
  if (a.VR >= (1 << 20) || (size_t)a.N * std::max(a.Cin, a.Cout) * a.H * a.W >= (size_t)1 << 31) return 1;
  a.CoP = (a.Cout + 31) / 32 * 32;
  a.mode = EPI_GENERIC;
  if (a.Cout % 32 == 0) {
    if (!a.dgrad && a.bias) {
      if (a.y_full && !a.y_out) a.mode = EPI_FWD_FULL;
      else if (a.y_full && a.y_out && a.skip && a.scale) a.mode = EPI_FWD_BOTH;
      else if (!a.y_full && a.y_out && a.skip && !a.scale) a.mode = EPI_FWD_OUT;
    } else if (a.dgrad) {
      if (a.act && !a.skip) a.mode = EPI_DGRAD_ACT;
      else if (!a.act && a.skip) a.mode = EPI_DGRAD_ADD;
    }
  }
  const int rows_total = a.VR - 1;
  int bestNT = 0, bestMT = 0, bestR = 0; double bestT = 0;
  static const struct Force { int mt = 0, nt = 0, grid = -1; Force() {            // development knobs, read once
    if (const char* e = getenv("FDET_SB_TILE")) sscanf(e, "%d,%d", &mt, &nt);
    if (const char* e = getenv("FDET_SB_GRID")) grid = atoi(e);
  } } force;
  const int forceMT = force.mt, forceNT = force.nt;
  for (int MT = (a.CoP % 64 == 0) ? 2 : 1; MT >= 1; --MT)
    for (int NT = 2; NT >= 1; NT >>= 1) {
      if (forceMT && (MT != forceMT || NT != forceNT)) continue;
      const int cap = 4 * NT * 32;
      if (a.WP > cap) continue;
      int R = cap / a.WP;
      if (R > rows_total) R = rows_total;
      if (2 * (R + 2) * (a.W / VW) > nbs_sb(NT, VW) * NTHR) continue;
      const int PT = cap + 2 * a.WP + 3;
      const size_t lds = (size_t)(2 * 9 * 2 * MT * 32 + 4 * PT) * 16;
      if (lds > 80 * 1024) continue;
      const long nb = (rows_total + R - 1) / R;
      const long wgs = nb * (a.CoP / (MT * 32));
      const long rounds = (wgs + 511) / 512;
      const double t = rounds * (2.0 * MT * NT + 0.6) * (1.0 + 0.25 * 2.0 / R);
      if (bestNT == 0 || t < bestT) { bestNT = NT; bestMT = MT; bestR = R; bestT = t; }
    }
  if (bestNT == 0) return 1;
  const int NT = bestNT, MT = bestMT;
  const int cap = 4 * NT * 32;
  a.R = bestR;
  a.nbands = (rows_total + a.R - 1) / a.R;
  a.magic_h1 = magic_of(a.H + 1);
  X3SbArgs p;
  p.PT = cap + 2 * a.WP + 3;
  p.p_in = (a.R + 2) * (a.W / VW);
  p.magic_w = magic_of(a.W / VW);
  const size_t units = (size_t)(a.Cin / 16) * 9 * 2 * a.CoP;       // per hi / lo half
  p.a_hi = reinterpret_cast<const bf16x8*>(a.wpk);
  p.a_lo = p.a_hi + units;
  p.c = a;
  const size_t lds = (size_t)(2 * 9 * 2 * MT * 32 + 4 * p.PT) * 16;
  p.ncob = a.CoP / (MT * 32);
  p.ntiles = a.nbands * p.ncob;
  // persistent: two workgroups per CU walk the tiles (the zero fill and the first-chunk latency are
  // paid once per workgroup, later tiles prefetch their first chunk under the previous tile's tail)
  int gsz = 2 * sb_num_cus();
  if (force.grid >= 0) gsz = force.grid > 0 ? force.grid : p.ntiles;
  dim3 grid(p.ntiles < gsz ? p.ntiles : gsz, 1);
  p.c.stagger = 0;
  if (MT == 2 && NT == 2) return launch_sb<2, 2>(p, VW, lds, grid, st);
  if (MT == 2 && NT == 1) return launch_sb<2, 1>(p, VW, lds, grid, st);
  if (MT == 1 && NT == 2) return launch_sb<1, 2>(p, VW, lds, grid, st);
  return launch_sb<1, 1>(p, VW, lds, grid, st);
}

// Pooled-block modes on the aligned-band variant (two workgroups per CU).  Returns 1 when it has no tiling.
int fdet_x3_sb_pool_run(ConvArgs a, PoolArgs q, hipStream_t st) {
  const bool pooled_ = q.pool_out || q.pool_din;
  if (!(a.slope >= 0.f && a.slope <= 1.f) || a.W > 63 || a.Cout % 32 != 0 || a.Cin % 16 != 0) return 1;
  if (pooled_ && (a.W > 62 || (a.W & 1) || (a.H & 1))) return 1;
  if (!pooled_ && a.W < 17) return 1;     // narrow plain maps: the 64-positions-per-wave mapping wastes fewer lanes
  a.WP = a.W <= 31 ? 32 : 64;             // a wave owns two rows x 32 columns: row pitch 32 or 64
  auto aligned = [](const void* ptr, size_t b) { return ((uintptr_t)ptr % b) == 0; };
  const int VW = (a.W % 4 == 0 && aligned(a.x, 16)) ? 4 : (aligned(a.x, 8) ? 2 : 1);
  if ((size_t)a.N * std::max(a.Cin, a.Cout) * a.H * a.W >= (size_t)1 << 31) return 1;
  a.CoP = a.Cout;
  const bool pooled = q.pool_out || q.pool_din;
  a.mode = -1;
  if (pooled) {
    if (q.pool_out && !a.dgrad && a.bias && a.skip && !a.y_full && !a.y_out) a.mode = EPI_FWD_POOL;
    else if (q.pool_din && q.mask_in && a.dgrad && !a.act && !a.skip && a.y_full) a.mode = EPI_DGRAD_ADDPOOL;
  } else if (!a.dgrad && a.bias) {
    if (a.y_full && !a.y_out) a.mode = EPI_FWD_FULL;
    else if (a.y_full && a.y_out && a.skip && a.scale) a.mode = EPI_FWD_BOTH;
    else if (!a.y_full && a.y_out && a.skip && !a.scale) a.mode = EPI_FWD_OUT;
  } else if (a.dgrad) {
    if (a.act && !a.skip) a.mode = EPI_DGRAD_ACT;
    else if (!a.act && a.skip) a.mode = EPI_DGRAD_ADD;
  }
  if (a.mode < 0) return 1;
  const int MT = (a.CoP % 64 == 0) ? 2 : 1, cap = 256;
  int R = cap / a.WP;
  if (R > a.H) R = (a.H + 1) & ~1;        // row pairs
  if (2 * (R + 2) * (a.W / VW) > nbs_sb(2, VW) * NTHR) return 1;
  a.R = R;
  X3SbArgs p;
  p.q = q;
  p.PT = cap + 2 * a.WP + 3;
  p.p_in = (a.R + 2) * (a.W / VW);
  p.magic_w = magic_of(a.W / VW);
  p.magic_wp = magic_of(a.WP);
  p.bpi = (a.H + a.R - 1) / a.R;
  p.magic_bpi = magic_of(p.bpi);
  const long nbt = (long)a.N * p.bpi;
  if (nbt >= (1 << 20)) return 1;
  const size_t units = (size_t)(a.Cin / 16) * 9 * 2 * a.CoP;
  p.a_hi = reinterpret_cast<const bf16x8*>(a.wpk);
  p.a_lo = p.a_hi + units;
  a.VR = 0; a.nbands = (int)nbt; a.magic_h1 = 0; a.stagger = 0;
  p.c = a;
  const size_t lds = (size_t)(2 * 9 * 2 * MT * 32 + 4 * p.PT) * 16;
  if (lds > 80 * 1024) return 1;
  p.ncob = a.CoP / (MT * 32);
  p.ntiles = (int)nbt * p.ncob;
  const int gsz = 2 * sb_num_cus();
  dim3 grid(p.ntiles < gsz ? p.ntiles : gsz, 1);
  return MT == 2 ? launch_sb_pool<2>(p, VW, lds, grid, st) : launch_sb_pool<1>(p, VW, lds, grid, st);
}
