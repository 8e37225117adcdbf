// PoolResnet stem forward (Conv2d(3,64,10,stride 8,pad 2), bf16x3 or one-pass bf16) with a pre-split (PS) output, round-4
// form: the fp32 input rows arrive in LDS by LDS-DMA (fdet_ldsdma.h) -- no staging registers, no asm loads with register
// destinations, no LDS stores -- and a wave turns the 8 consecutive pixels of its MFMA B fragment into bf16 hi | lo parts in
// registers right before the MFMAs that consume them.
//
//   * MFMA K = one (ci, ky) input row x 16 taps (t = kx + 2; taps outside [2, 12) carry zero weights), as fdet_stem_x3.hip: a
//     lane's B fragment for output column ox and k half h is the 8 pixels 8 ox + 8 h + j of a row stored with a 4-pixel
//     zero pad in front: 32 contiguous, 32-byte aligned bytes;
//   * one workgroup (4 waves) walks contiguous output rows; wave = (position tile nt, K half ks): BOTH 32-channel output tiles
//     for 32 columns over 15 of the 30 input rows, so every fragment is split once per workgroup (the (m, nt) mapping of
//     the register-staged kernel made the two channel-half waves split the same pixels) and the split's 24 VALU sit beside
//     6 MFMAs: the pipe, not the issue port, bounds a row (2880 cycles).  The two K halves swap one accumulator tile through
//     LDS and each finishes one 32-channel tile (the PS epilogue of fdet_stem_x3.hip);
//   * two fp32 row tiles (2 x 30 rows x 2080 B): the 60 one-KiB pieces of row r+1 are issued right after the barrier that
//     opens row r and have the whole row to land; the wait that opens row r+1 is `vmcnt(stores of row r's epilogue)`.
#include "fdet_common.h"
#include "fdet_ps.h"
#include "fdet_ldsdma.h"

using namespace fdet;
typedef float sd_f32x16 __attribute__((ext_vector_type(16)));
typedef float sd_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 sd_bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* sd_lds_void_t;

namespace {

constexpr int SD_KS = 10, SD_ST = 8, SD_PD = 2, SD_CIN = 3;
constexpr int SD_NROW = SD_CIN * SD_KS;        // 30 (ci, ky) rows
constexpr int SD_RLF = 520;                    // floats per LDS row: 4 zero | 480 pixels | zeros up to 8 * 63 + 15
constexpr int SD_TILE = SD_NROW * SD_RLF * 4;  // bytes
constexpr int SD_XCH = 4 * 16 * 64 * 4;        // accumulator exchange: [wave][16 registers][64 lanes]
constexpr int SD_LDS = 2 * SD_TILE + SD_XCH + 256;

struct StemDmaArgs {
  const float* x; const float* w; const float* bias; void* y;
  int N, H, W, Ho, Wo, nrows;
  int ps_hp, ps_wp, ps_plane, ps_img;
};

template <bool P16>
__global__ void __launch_bounds__(256, 1)
k_stem_fwd_dma(const StemDmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int nt = wid & 1, ks = wid >> 1;
  float* const sbias = reinterpret_cast<float*>(smem + 2 * SD_TILE + SD_XCH);
  {
    sd_f32x4* z = reinterpret_cast<sd_f32x4*>(smem);
    for (int t = tid; t < (2 * SD_TILE) / 16; t += 256) z[t] = sd_f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < 64) sbias[tid] = a.bias[tid];
  }
  // A fragments of this wave's 15 input rows, both channel tiles: element j <-> tap t = 8 half + j, kx = t - 2
  sd_bf16x8 ah[15][2], al[15][2];
#pragma unroll
  for (int i = 0; i < 15; ++i)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kx = 8 * half + j - 2, rr = 15 * ks + i, co = 32 * m + l31;
        const float f = (kx >= 0 && kx < SD_KS) ? a.w[((size_t)co * SD_NROW + rr) * SD_KS + kx] : 0.f;
        const __bf16 h = (__bf16)f;
        ah[i][m][j] = h;
        if (!P16) al[i][m][j] = (__bf16)(f - (float)h);
      }
  const unsigned lds0 = (unsigned)(size_t)(sd_lds_void_t)smem;
  const dma_u32x4 xrs = dma_rsrc(a.x, (unsigned)((size_t)a.N * SD_CIN * a.H * a.W * 4));
  const unsigned lane16 = (unsigned)lane * 16u;
  // the 15 pieces of this wave for output row R (R >= last: nothing valid, the pieces bring zeros) into tile TB
  const int bpw = (a.nrows + (int)gridDim.x - 1) / (int)gridDim.x;
  int row = blockIdx.x * bpw;
  const int last = min(row + bpw, a.nrows);
  if (row >= last) return;
#define SD_ISSUE(R, TB)                                                                            \
  {                                                                                                \
    const int n_ = (R) / a.Ho, oy_ = (R) - n_ * a.Ho;                                              \
    _Pragma("unroll") for (int k_ = 0; k_ < 15; ++k_) {                                            \
      const int q_ = wid * 15 + k_;                                                                \
      const int rr_ = q_ >> 1, part_ = q_ & 1;                                                     \
      const int ci_ = rr_ / SD_KS, ky_ = rr_ - ci_ * SD_KS;                                        \
      const int iy_ = oy_ * SD_ST - SD_PD + ky_;                                                   \
      const bool ok_ = (R) < last && iy_ >= 0 && iy_ < a.H && (part_ * 256 + lane * 4) < a.W;      \
      const unsigned so_ = (unsigned)(((n_ * SD_CIN + ci_) * a.H + (iy_ < 0 ? 0 : iy_)) * a.W + part_ * 256) * 4u; \
      dma_piece(lds0 + (unsigned)(TB) * SD_TILE + (unsigned)(rr_ * SD_RLF + 4 + part_ * 256) * 4u, ok_ ? lane16 : 0x80000000u, xrs, \
                (R) < last ? so_ : 0u);                                                            \
    }                                                                                              \
  }
  __syncthreads();                                         // zero fill done before the first piece lands
  SD_ISSUE(row, 0)
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.N * a.ps_img * 16, 0x00020000);
  float* const xch = reinterpret_cast<float*>(smem + 2 * SD_TILE);
  const int ox = nt * 32 + l31;
  int tb = 0;
  bool first = true;
  for (; row < last; ++row, tb ^= 1) {
    // this row's pieces (issued a whole row ago) have landed in every wave: only the previous epilogue's stores are younger
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (P16) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    first = false;
    __builtin_amdgcn_s_barrier();
    SD_ISSUE(row + 1, tb ^ 1)
    __builtin_amdgcn_sched_barrier(0);
    const int n = row / a.Ho, oy = row - n * a.Ho;
    const float* T = reinterpret_cast<const float*>(smem + tb * SD_TILE) + (15 * ks) * SD_RLF + 8 * ox + 8 * half;
    sd_f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    sd_f32x4 f0[2], f1[2];
    f0[0] = *reinterpret_cast<const sd_f32x4*>(T);
    f1[0] = *reinterpret_cast<const sd_f32x4*>(T + 4);
#pragma unroll
    for (int i = 0; i < 15; ++i) {
      const int cur = i & 1, nxt = cur ^ 1;
      if (i + 1 < 15) {
        f0[nxt] = *reinterpret_cast<const sd_f32x4*>(T + (i + 1) * SD_RLF);
        f1[nxt] = *reinterpret_cast<const sd_f32x4*>(T + (i + 1) * SD_RLF + 4);
      }
      sd_bf16x8 bh, bl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = j < 4 ? f0[cur][j] : f1[cur][j - 4];
        const __bf16 h = (__bf16)v;
        bh[j] = h;
        if (!P16) bl[j] = (__bf16)(v - (float)h);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (!P16) {
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i][m], bl, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i][m], bh, acc[m], 0, 0, 0);
        }
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i][m], bh, acc[m], 0, 0, 0);
      }
    }
    // the two K halves of a position tile swap one accumulator tile: wave ks finishes channel tile m = ks.  (A register array
    // indexed by the run-time ks would make hipcc fall back to s_set_gpr_idx moves: a uniform branch picks the tiles instead.)
    sd_f32x16 keep, give;
    if (ks == 0) { keep = acc[0]; give = acc[1]; } else { keep = acc[1]; give = acc[0]; }
    {
      float* mine = xch + (size_t)wid * 16 * 64;
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[r * 64 + lane] = give[r];
    }
    // (not __syncthreads(): its vmcnt(0) would wait for the pieces of the NEXT row that were just issued)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
      const float* theirs = xch + (size_t)(wid ^ 2) * 16 * 64;
#pragma unroll
      for (int r = 0; r < 16; ++r) keep[r] += theirs[r * 64 + lane];
    }
    const int m = ks;
    // PS epilogue (fdet_stem_x3.hip): lane = position, registers = channels 32m + 8g + 4half + i; a v_permlane32_swap pair
    // leaves a lane with the 8 channels of one unit
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      float za[4], zb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        za[i] = keep[8 * gp + i] + sbias[m * 32 + 16 * gp + 4 * half + i];
        zb[i] = keep[8 * gp + 4 + i] + sbias[m * 32 + 16 * gp + 8 + 4 * half + i];
      }
      unsigned ha[2], la[2], hb[2], lb[2];
      if (P16) { ps_hi4(za, ha); ps_hi4(zb, hb); } else { ps_split4(za, ha, la); ps_split4(zb, hb, lb); }
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
      const unsigned off = ox < a.Wo ? (unsigned)(n * a.ps_img + (G * a.ps_hp + oy) * a.ps_wp + ox + 1) * 16u : 0x80000000u;
      typedef unsigned sd_u32x4 __attribute__((ext_vector_type(4)));
      __builtin_amdgcn_raw_buffer_store_b128(sd_u32x4{ha[0], ha[1], hb[0], hb[1]}, ry, off, 0, 0);
      if (!P16) __builtin_amdgcn_raw_buffer_store_b128(sd_u32x4{la[0], la[1], lb[0], lb[1]}, ry, off, a.ps_plane * 16, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the run-ahead pieces target this workgroup's LDS
#undef SD_ISSUE
}

}  // namespace

namespace fdet {
// -> FDET_OK, or 1 when this geometry is not served (the caller keeps the register-staged kernel)
int stem_dma_fwd_ps(const float* x, const float* w, const float* bias, void* y_ps, int N, int F, int H, int W, hipStream_t st, bool p16) {
  StemDmaArgs a{};
  a.x = x; a.w = w; a.bias = bias; a.y = y_ps; a.N = N; a.H = H; a.W = W;
  a.Ho = (H + 4 - 10) / 8 + 1; a.Wo = (W + 4 - 10) / 8 + 1; a.nrows = N * a.Ho;
  PsGeo g;
  if (F != 64 || W > 480 || W % 4 || a.Wo > 62 || !ps_geo(N, F, a.Ho, a.Wo, g) || (size_t)N * SD_CIN * H * W >= ((size_t)1 << 30)) return 1;
  a.ps_hp = g.HP; a.ps_wp = g.WP; a.ps_plane = g.plane; a.ps_img = g.img;
  const int nblk = a.nrows < 256 ? a.nrows : 256;
  const void* kern = p16 ? (const void*)k_stem_fwd_dma<true> : (const void*)k_stem_fwd_dma<false>;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, SD_LDS) != hipSuccess) {
    (void)hipGetLastError();
    return fail(FDET_ELAUNCH, "stem_fwd_ps (dma): cannot reserve %d bytes of LDS", SD_LDS);
  }
  if (p16) hipLaunchKernelGGL(k_stem_fwd_dma<true>, dim3(nblk), dim3(256), SD_LDS, st, a);
  else hipLaunchKernelGGL(k_stem_fwd_dma<false>, dim3(nblk), dim3(256), SD_LDS, st, a);
  return check_launch("fdet_stem_fwd_ps(dma)");
}
}  // namespace fdet
