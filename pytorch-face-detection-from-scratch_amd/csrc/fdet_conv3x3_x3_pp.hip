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
#include "fdet_conv3x3_x3_epi.h"
#include <algorithm>
#include <cstdint>

using namespace fdet;

namespace {

// Issue arbitration between the two waves of a SIMD (MI355X_MICROARCH.md, "Two waves per SIMD"): a wave whose next
// instruction is an MFMA that waits for the busy matrix pipe keeps winning the vector issue port by priority / age, and
// its partner gets about one VALU instruction per MFMA.  The MFMA stream therefore yields explicitly: after every MFMA
// it sleeps FDET_PP_NOP cycles (s_nop: not a candidate for issue), which hands the partner group's split / epilogue
// VALU ~4 issue slots per MFMA, and runs at raised priority so that it gets the port back the moment the pipe frees.
#ifndef FDET_PP_NOP
#define FDET_PP_NOP 0
#endif
#ifndef FDET_PP_ABL
#define FDET_PP_ABL 0      // diagnostic builds only: 1 = empty memory segments, 2 = no B prefetch in the MFMA segment, 4 = no MFMAs, 8 = no LDS writes / weight DMA
#endif
#ifndef FDET_PP_PRIO
#define FDET_PP_PRIO 0
#endif
#define PP_STR2(x) #x
#define PP_STR(x) PP_STR2(x)
#if FDET_PP_NOP > 0
#define PP_YIELD() { asm volatile("s_nop " PP_STR(FDET_PP_NOP) " - 1"); __builtin_amdgcn_sched_barrier(0); }
#else
#define PP_YIELD()
#endif
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
  long long* stamps;         // diagnostic build (-DFDET_PP_STAMPS, tools/probe) only; null otherwise
};

#ifdef FDET_PP_STAMPS
// per workgroup (first 32 only) and wave: [0] = HW_ID, [1] = phases, then per phase {start, end of work} in shader clocks
#define PP_STAMP(SLOT) { __builtin_amdgcn_sched_barrier(0); if (p.stamps && blockIdx.x < 32 && lane == 0 && (SLOT) < 250) p.stamps[((size_t)blockIdx.x * 8 + wid) * 256 + (SLOT)] = (long long)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define PP_STAMP(SLOT) {}
#endif

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

  if (a.stagger > 0) {
    // De-synchronise the CUs.  Every workgroup runs the same phase sequence, so without this all 256 CUs issue their
    // epilogue store bursts (and their chunk loads) in the same few microseconds and HBM alternates between saturated
    // and idle: measured 8 k cycles for an epilogue segment that takes 3 k on an otherwise quiet chip.  Workgroup i
    // starts (i / 8) % 8 eighths of a tile period late (the 8 workgroups of a "column" sit on the 8 XCDs).
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    const long long wait = (long long)a.stagger * ((blockIdx.x >> 3) & 7);
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
  }
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

  // ---- prologue: both groups stage their first chunk; the first weight panel arrives (every wave waits for its own
  // LDS-DMA pieces, the barrier covers the others)
  __syncthreads();                                    // zero fill complete
  if (grp == 1) PP_DMA_W(0)
  if (K > 0) {
    PP_TILE_SRC(0)
    cur_img = st_img; cur_y0 = st_y0;
    PP_ISSUE_B(0)
    PP_WRITE_B()
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // Phase ph: group (ph & 1) is in its MFMA segment, the other one in its memory segment.  Group g's segment number
  // ks (chunk ks % nch of its tile ks / nch) runs in phase 2*ks + g; its activations were loaded into registers
  // during segment ks-1 (one phase of MFMAs hides the HBM latency) and written to LDS in the memory segment between.
  // Weight panel of segment s (both groups: phases 2s and 2s+1) lives in ring slot s & 1, free from the end of phase
  // 2s-3.  GROUP 1 requests it at the start of its memory segment in phase 2s-2 and waits for it at the END of its MFMA
  // segment in phase 2s-1 -- a whole phase later, so the wait (vmcnt counts in order: it also covers the epilogue stores
  // issued after the request) costs nothing, and the barrier that ends phase 2s-1 publishes the slot.
  const int nphase = 2 * K0 + 1;
#ifdef FDET_PP_STAMPS
  if (p.stamps && blockIdx.x < 32 && lane == 0) {
    p.stamps[((size_t)blockIdx.x * 8 + wid) * 256 + 0] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    p.stamps[((size_t)blockIdx.x * 8 + wid) * 256 + 1] = nphase;
  }
#endif
  for (int ph = 0; ph < nphase; ++ph) {
    PP_STAMP(2 + 2 * ph)
    const int d = ph - grp;
    if (d >= 0 && (d & 1) == 0) {
      // =========================== MFMA segment
      const int ks = d >> 1;
      if (ks < K) {
        __builtin_amdgcn_s_setprio(FDET_PP_PRIO);     // this wave's MFMAs before the partner group's VALU / LDS-store stream
        const int kn = ks + 1;                        // the loads of the next segment travel under this one
        if (kn < K && !(FDET_PP_ABL & 2)) {
          const int cn = kn % nch;
          if (cn == 0) PP_TILE_SRC(kn / nch)
          PP_ISSUE_B(cn)
        }
        const bf16x8* Aw = lds + (ks & 1) * 2 * A_UNITS + a_off;
        const bf16x8* Bw = Bbuf + b_off;
        bf16x8 ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) { ah[0][m] = Aw[m * 32]; al[0][m] = Aw[A_UNITS + m * 32]; }
#pragma unroll
        for (int n = 0; n < NT; ++n) { bh[0][n] = Bw[tapoff[0] + n * nstride]; bl[0][n] = Bw[2 * PT + tapoff[0] + n * nstride]; }
#pragma unroll
        for (int t = 0; t < ((FDET_PP_ABL & 4) ? 0 : 9); ++t) {
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
            {
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][m], bl[cur][n], acc[m][n], 0, 0, 0);
              PP_YIELD()
            }
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cur][m], bh[cur][n], acc[m][n], 0, 0, 0);
              PP_YIELD()
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][m], bh[cur][n], acc[m][n], 0, 0, 0);
              PP_YIELD()
            }
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
      }
      if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the panel requested one phase ago (see above)
    } else if (d < 0) {
      if (1 < K0 && !(FDET_PP_ABL & 8)) PP_DMA_W(1)   // group 1, phase 0: panel of segment 1 (waited for at the end of phase 1)
    } else if (!(FDET_PP_ABL & 1)) {
      // =========================== memory segment: everything this group needs before its segment kn
      const int kn = (d + 1) >> 1;
      const int cn = kn % nch;
      const bool tile_done = cn == 0 && kn >= nch && kn <= K;      // segment kn-1 completed a tile
      if (tile_done) {
        const EpiGeo eg = epi_geometry<MODE>(p, cur_img, cur_y0, qwave, nstride, mb * MB, l31, half);
        f32x4 u[MT][2][4];
        float dg[MT][4][2];
        unsigned mk[MT][4];
        float bzm[MT][4], scm[MT][4];
        if (kn == 2 * nch) PP_STAMP(200)
        epi_loads<MT, MODE>(p, eg, u, dg, mk, bzm, scm, cur_img, mb * MB, l31, half);   // issue; the LDS writes and the transposes hide their latency
        if (kn == 2 * nch) PP_STAMP(201)
        if (kn < K && !(FDET_PP_ABL & 8)) PP_WRITE_B()
        if (kn == 2 * nch) PP_STAMP(202)
        epi_finish<MT, MODE>(p, eg, acc, u, dg, mk, bzm, scm, cur_img, mb * MB, l31, half);
        if (kn == 2 * nch) PP_STAMP(203)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        cur_img = st_img; cur_y0 = st_y0;
      } else if (kn < K && !(FDET_PP_ABL & 8)) {
        PP_WRITE_B()
      }
      // LAST in the segment: while an LDS-DMA is in flight hipcc turns every wait for an ordinary load into vmcnt(0), which
      // would expose the DMA's ~2 us flight inside this segment (measured: +4.4 k cycles)
      if (grp == 1 && kn + 1 < K0 && !(FDET_PP_ABL & 8)) PP_DMA_W(kn + 1)   // panel of segment kn+1 (phases 2kn+2, 2kn+3)
    }
    PP_STAMP(3 + 2 * ph)
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

long long* g_pp_stamps = nullptr;    // set only by the diagnostic build's fdet_pp_probe_set

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
  // Plain epilogue modes: measured on MI355X (r02, 256 x 64 x 60x60) this kernel ties or trails the round-1 small-tile kernel
  // (fwd 0.229 vs 0.208 ms, dgrad 0.280 vs 0.237 ms): its whole epilogue (128 KB of HBM traffic per tile) falls into ONE
  // phase, ~10 k cycles at a CU's share of HBM bandwidth against a 3.5 k cycle MFMA segment, while two free-running
  // workgroups per CU spread it.  It is therefore opt-in for them (FDET_CONV_PP=1) and the default only for the pooled modes,
  // which only it implements.
  static const bool enabled = [] { const char* e = getenv("FDET_CONV_PP"); return e && e[0] == '1'; }();
  const bool pooled = q.pool_out || q.pool_din;
  if (!enabled && !pooled) return 1;
  if (!(a.slope >= 0.f && a.slope <= 1.f)) return pooled ? fail(FDET_EINVAL, "conv3x3 pooled epilogue: LeakyReLU slope must be in [0,1] (slope=%f)", (double)a.slope) : 1;
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
  a.VR = 0; a.nbands = p.bpi; a.magic_h1 = 0;
  // start stagger: one eighth of a tile-pair period (2 * Cin/16 phases of ~3.7 k cycles) per step; only worth it when
  // the run is long enough to amortise the late starters (FDET_PP_STAGGER=<cycles per step> overrides, 0 disables)
  static const int stagger_env = [] { const char* e = getenv("FDET_PP_STAGGER"); return e ? atoi(e) : -1; }();
  a.stagger = stagger_env >= 0 ? stagger_env : ((p.ntiles_mb * p.ncob >= 8 * pp_num_cus()) ? (2 * (a.Cin / 16) * 3700) / 8 : 0);
  p.c = a;
  p.stamps = g_pp_stamps;
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

#ifdef FDET_PP_STAMPS
extern "C" int fdet_pp_probe_set(long long* buf) { g_pp_stamps = buf; return 0; }
extern "C" int fdet_pp_probe_conv(const float* x, const void* wpk, const float* bias, float* y, const float* act, int N, int C,
                                  int H, int W, int dgrad, void* stream) {
  ConvArgs a{};
  a.x = x; a.wpk = (const float*)wpk; a.bias = dgrad ? nullptr : bias; a.y_full = y; a.act = dgrad ? act : nullptr;
  a.N = N; a.Cin = C; a.Cout = C; a.H = H; a.W = W; a.dgrad = dgrad; a.slope = 0.2f;
  return fdet_x3_pp_run(a, PoolArgs{nullptr, nullptr, nullptr, nullptr}, (hipStream_t)stream);
}
#endif
