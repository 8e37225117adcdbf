// 3x3 stride-1 pad-1 convolutions of the residual blocks as fp32 implicit GEMM on
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD = the fp32 roof).
//
// Data layout trick ("flattened padded rows"): the batch is one tall virtual image -- image n
// occupies virtual rows n*(H+1)+1 .. n*(H+1)+H, row n*(H+1) is a shared zero row -- and every
// row carries ONE zero column (WP = W+1) which is at once the right halo of row r and the left
// halo of row r+1.  An output position q = row*WP + col then reads tap (ky,kx) at LDS offset
// q + ky*WP + kx: a constant per tap, so the 32 lanes of an MFMA B operand are 32 consecutive
// dwords (conflict-free ds_read_b32) for any image width.  Garbage positions (zero column,
// zero rows) are computed and dropped; cost (W+1)/W * (H+1)/H.
//
// GEMM mapping (forward):  D[co][q] += A[co][k] * B[k][q],  k = (ci, tap)
//   A = packed weights  wpk[k][co]     (8-channel chunks, LDS)
//   B = input band      x[ci][q+off]   (band of R virtual rows + halo rows, LDS)
// MFMA 32x32x2: lanes 0-31 carry k even (channel 2c), lanes 32-63 k odd (channel 2c+1).
// The data-gradient conv is the same kernel on dz with flipped/transposed packed weights.
//
// Workgroup = 4 waves (one per SIMD), each wave owns MT x NT 32x32 tiles; LDS is double
// buffered and the next chunk is prefetched global->registers while the current chunk's
// MFMAs run (one barrier per chunk).  Two workgroups share a CU so that one's prologue /
// epilogue (not MFMA work) overlaps the other's MFMA loop.
#include "fdet_conv_common.h"

using namespace fdet;


namespace {

template <int MT, int NT, int VW>
__global__ void __launch_bounds__(NTHR, 2)
k_conv3x3(const ConvArgs a) {
#ifdef FDET_CONV_DBG
  const int DBG = a.dbg;                      // ablation builds only: skips loads / LDS writes / barriers (WRONG results)
#else
  constexpr int DBG = 0;                      // the shipped library cannot be talked into skipping its barriers
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MB = MT * 32;                 // output channels per block
  constexpr int A_ITEMS = CK * 9 * (MB / 4);  // float4 items of one A chunk
  constexpr int NA = (A_ITEMS + NTHR - 1) / NTHR;
  constexpr int NBMAX = nbmax(VW);
  using VT = typename Vec<VW>::T;
  const int CS = a.CS, WP = a.WP;
  const int bufsz = CK * 9 * MB + CK * CS;    // floats per buffer (A then B)
  float* lds = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int band = blockIdx.x, mb = blockIdx.y;
  const int v0 = band * a.R;
  const int H1 = a.H + 1;

  // Two workgroups share a CU and would otherwise run their prologue / MFMA loop / epilogue in
  // lockstep; delaying the one in the odd wave slot lets one's epilogue overlap the other's MFMAs.
  if (a.stagger > 0) {
    const unsigned hwid = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID bits [3:0] = wave slot
    if (hwid & 1u)
      for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  // B halos stay zero for all chunks (staging never touches them)
  for (int t = tid; t < 2 * bufsz; t += NTHR) lds[t] = 0.f;

  // ---- per-thread staging geometry (chunk invariant)
  const int rows = a.R + 2;
  const int wv = a.W / VW;
  const int lpr = 1 << a.lpr_log2;
  const int xv = tid & (lpr - 1);
  const int pair0 = tid >> a.lpr_log2, pair_step = NTHR >> a.lpr_log2;
  const int npairs = CK * rows;
  int b_src[NBMAX], b_dst[NBMAX];             // src: offset inside a channel-chunk of x; dst: inside B tile; -1 = skip
#pragma unroll
  for (int s = 0; s < NBMAX; ++s) {
    const int pr = pair0 + s * pair_step;
    b_src[s] = -1; b_dst[s] = 0;
    if (pr < npairs && xv < wv) {
      const int ci = fdiv(pr, a.magic_rows), tr = pr - ci * rows;
      const int v = v0 - 1 + tr;
      if (v >= 0 && v < a.VR) {
        const int n = fdiv(v, a.magic_h1), yy = v - n * H1 - 1;
        if (yy >= 0) {
          b_src[s] = ((n * a.Cin + ci) * a.H + yy) * a.W + xv * VW;
          b_dst[s] = ci * CS + tr * WP + 1 + xv * VW;
        }
      }
    }
  }
  const size_t chunk_stride = (size_t)CK * a.H * a.W;
  const float* asrc0 = a.wpk + mb * MB;

  f32x4 pa[NA];
  VT pb[NBMAX];
#define FDET_ISSUE_LOADS(C0)                                                                         \
  {                                                                                                  \
    const float* asrc = asrc0 + (size_t)(C0) * 9 * a.CoP;                                            \
    _Pragma("unroll") for (int s_ = 0; s_ < NA; ++s_) {                                              \
      const int t_ = min(tid + s_ * NTHR, A_ITEMS - 1);                                              \
      const int row_ = t_ / (MB / 4), c4_ = t_ - row_ * (MB / 4);                                    \
      pa[s_] = *reinterpret_cast<const f32x4*>(asrc + (size_t)row_ * a.CoP + c4_ * 4);              \
    }                                                                                                \
    const float* xsrc = a.x + (size_t)((C0) / CK) * chunk_stride;                                    \
    _Pragma("unroll") for (int s_ = 0; s_ < NBMAX; ++s_)                                             \
      pb[s_] = *reinterpret_cast<const VT*>(xsrc + max(b_src[s_], 0)); /* dummy load when skipped */ \
  }
#define FDET_WRITE_LDS(BUF)                                                                          \
  {                                                                                                  \
    float* buf_ = (BUF);                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < NA; ++s_) {                                              \
      const int t_ = tid + s_ * NTHR;                                                                \
      if (t_ < A_ITEMS) *reinterpret_cast<f32x4*>(buf_ + t_ * 4) = pa[s_];                           \
    }                                                                                                \
    float* B_ = buf_ + CK * 9 * MB;                                                                  \
    _Pragma("unroll") for (int s_ = 0; s_ < NBMAX; ++s_) {                                           \
      if (b_src[s_] >= 0) {                                                                          \
        _Pragma("unroll") for (int k_ = 0; k_ < VW; ++k_) B_[b_dst[s_] + k_] = vget<VW>(pb[s_], k_); \
      }                                                                                              \
    }                                                                                                \
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

  const int qwave = wid * NT * 32;
  const int a_off = half * 9 * MB + l31;
  const int b_off = CK * 9 * MB + half * CS + qwave + l31;

  FDET_ISSUE_LOADS(0)
  __syncthreads();                       // zero fill complete
  FDET_WRITE_LDS(lds)
  __syncthreads();

  const int nch = a.Cin / CK;
  for (int c = 0; c < nch; ++c) {
    const float* buf = lds + (c & 1) * bufsz;
    if (c + 1 < nch && !(DBG & 2)) FDET_ISSUE_LOADS((c + 1) * CK)
    const float* Aw = buf + a_off;
    const float* Bw = buf + b_off;
    // software-pipelined operand fetch: the LDS reads of k-pair kk+1 are issued before the
    // MFMAs of k-pair kk (pinned with sched_barrier), so their latency hides under 512+ cycles
    // of MFMA issue instead of stalling every group.
    float av[2][MT], bv[2][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) av[0][m] = Aw[m * 32];
#pragma unroll
    for (int n = 0; n < NT; ++n) bv[0][n] = Bw[tapoff[0] + n * 32];
#pragma unroll
    for (int kk = 0; kk < (CK / 2) * 9; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < (CK / 2) * 9) {
        const int cp = (kk + 1) / 9, t = (kk + 1) % 9;
#pragma unroll
        for (int m = 0; m < MT; ++m) av[nxt][m] = Aw[(cp * 2 * 9 + t) * MB + m * 32];
#pragma unroll
        for (int n = 0; n < NT; ++n) bv[nxt][n] = Bw[cp * 2 * CS + tapoff[t] + n * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][m], bv[cur][n], acc[m][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (c + 1 < nch && !(DBG & 4)) FDET_WRITE_LDS(lds + ((c + 1) & 1) * bufsz)
    if (!(DBG & 8)) __syncthreads();
  }

  // ---- epilogue: one uniform switch on the fusion mode, then straight-line code per 32x32 tile:
  // all loads of a tile (skip / scale / act / add) are issued before its stores.
  const int qlimit = a.R * WP;
  const size_t HW = (size_t)a.H * a.W;
  bool okn[NT];
  size_t basen[NT];
  int imgn[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int q = qwave + n * 32 + l31;
    const int tr = q / WP, ox = q - tr * WP;
    const int v = v0 + tr;
    const int img = v / H1, oy = v - img * H1 - 1;
    okn[n] = (q < qlimit) && (ox < a.W) && (v < a.VR) && (oy >= 0) && !((DBG & 1) && acc[0][n][0] != 12345.f);
    basen[n] = ((size_t)img * a.Cout * a.H + oy) * a.W + ox;
    imgn[n] = img;
  }
  const int cob0 = mb * MB + 4 * half;
  switch (a.mode) {
    case EPI_FWD_FULL: epilogue<MT, NT, EPI_FWD_FULL>(a, acc, okn, basen, imgn, cob0, HW); break;
    case EPI_FWD_BOTH: epilogue<MT, NT, EPI_FWD_BOTH>(a, acc, okn, basen, imgn, cob0, HW); break;
    case EPI_FWD_OUT: epilogue<MT, NT, EPI_FWD_OUT>(a, acc, okn, basen, imgn, cob0, HW); break;
    case EPI_DGRAD_ACT: epilogue<MT, NT, EPI_DGRAD_ACT>(a, acc, okn, basen, imgn, cob0, HW); break;
    case EPI_DGRAD_ADD: epilogue<MT, NT, EPI_DGRAD_ADD>(a, acc, okn, basen, imgn, cob0, HW); break;
    default: epilogue<MT, NT, EPI_GENERIC>(a, acc, okn, basen, imgn, cob0, HW); break;
  }
}

// ---------------------------------------------------------------------------------------
// pack weights
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pack3x3(const float* __restrict__ w, int Cout, int Cin, int CoP, int CiP, float* __restrict__ fwd,
          float* __restrict__ bwd) {
  const int nf = Cin * 9 * CoP, nb = Cout * 9 * CiP;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (fwd && t < nf) {
    const int k = t / CoP, co = t - k * CoP;
    const int ci = k / 9, tap = k - ci * 9;
    fwd[t] = (co < Cout) ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.f;
  }
  if (bwd && t < nb) {
    const int k = t / CiP, ci = t - k * CiP;
    const int co = k / 9, tap = k - co * 9;
    bwd[t] = (ci < Cin) ? w[((size_t)co * Cin + ci) * 9 + (8 - tap)] : 0.f;   // flipped tap
  }
}


template <int MT, int NT>
int launch_conv_vw(const ConvArgs& a, int vw, size_t lds, dim3 grid, hipStream_t st) {
  auto set = [&](const void* f) -> int { return lds > 64 * 1024 ? set_lds_attr(f, lds, "conv3x3") : FDET_OK; };

  if (vw == 4) { if (int rc = set((const void*)k_conv3x3<MT, NT, 4>)) return rc; hipLaunchKernelGGL((k_conv3x3<MT, NT, 4>), grid, dim3(NTHR), lds, st, a); }
  else if (vw == 2) { if (int rc = set((const void*)k_conv3x3<MT, NT, 2>)) return rc; hipLaunchKernelGGL((k_conv3x3<MT, NT, 2>), grid, dim3(NTHR), lds, st, a); }
  else { if (int rc = set((const void*)k_conv3x3<MT, NT, 1>)) return rc; hipLaunchKernelGGL((k_conv3x3<MT, NT, 1>), grid, dim3(NTHR), lds, st, a); }
  return check_launch("fdet_conv3x3");
}

int run_conv(ConvArgs a, hipStream_t st) {
  a.WP = a.W + 1;
  a.VR = a.N * (a.H + 1) + 1;
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
  if (a.VR >= (1 << 20)) return fail(FDET_EINVAL, "conv3x3: N*(H+1)=%d virtual rows exceed the index range", a.VR);
  a.CoP = (a.Cout + 31) / 32 * 32;
  const int vw = (a.W % 4 == 0) ? 4 : (a.W % 2 == 0 ? 2 : 1);
  const int wv = a.W / vw;
  int lpr_log2 = 0;
  while ((1 << lpr_log2) < wv) ++lpr_log2;
  if ((1 << lpr_log2) > NTHR) return fail(FDET_EINVAL, "conv3x3: W=%d too wide", a.W);
  // choose (MT, NT): every SIMD hosts waves of MT*NT tile-jobs; time ~ ceil(waves/1024) * MT*NT.
  // Ties -> larger tiles (more operand reuse per LDS byte).
  const int rows_total = a.VR - 1;             // the last virtual row is a zero row
  int bestNT = 0, bestMT = 0, bestR = 0; long bestT = 0;
  int forceMT = 0, forceNT = 0;                 // development override: FDET_CONV_TILE="MT,NT"
  if (const char* e = getenv("FDET_CONV_TILE")) sscanf(e, "%d,%d", &forceMT, &forceNT);
#ifdef FDET_CONV_DBG
  if (const char* e = getenv("FDET_CONV_DBG")) a.dbg = atoi(e);   // ablation builds only (-DFDET_CONV_DBG)
#endif
  if (const char* e = getenv("FDET_CONV_STAGGER")) a.stagger = atoi(e);
  for (int MT = (a.CoP % 64 == 0) ? 2 : 1; MT >= 1; --MT)
  for (int NT = 4; NT >= 1; NT >>= 1) {
    if (forceMT && (MT != forceMT || NT != forceNT)) continue;
    const int cap = 4 * NT * 32;
    if (a.WP > cap) continue;
    int R = cap / a.WP;
    if (R > rows_total) R = rows_total;
    const int pairs = CK * (R + 2);
    const int slots = (pairs + (NTHR >> lpr_log2) - 1) / (NTHR >> lpr_log2);
    if (slots > nbmax(vw)) continue;
    const size_t lds = (size_t)2 * (CK * 9 * MT * 32 + CK * (cap + 2 * a.WP + 3)) * 4;
    if (lds > 160 * 1024) continue;
    const long nb = (rows_total + R - 1) / R;
    const long waves = nb * (a.CoP / (MT * 32)) * 4;
    // + a fixed per-wave overhead (prologue, A-panel staging, epilogue) worth ~1.5 tile-jobs
    const long t = ((waves + 1023) / 1024) * (2 * MT * NT + 3);
    if (bestNT == 0 || t < bestT) { bestNT = NT; bestMT = MT; bestR = R; bestT = t; }
  }
  if (bestNT == 0) return fail(FDET_EINVAL, "conv3x3: no tiling for W=%d H=%d", a.W, a.H);
  const int NT = bestNT, MT = bestMT;
  const int cap = 4 * NT * 32;
  a.R = bestR;
  a.nbands = (rows_total + a.R - 1) / a.R;
  a.CS = cap + 2 * a.WP + 2;
  if ((a.CS & 1) == 0) a.CS += 1;
  a.lpr_log2 = lpr_log2;
  a.magic_h1 = magic_of(a.H + 1);
  a.magic_rows = magic_of(a.R + 2);
  const size_t lds = (size_t)2 * (CK * 9 * MT * 32 + CK * a.CS) * 4;
  dim3 grid(a.nbands, a.CoP / (MT * 32));
  if (MT == 2) {
    if (NT == 4) return launch_conv_vw<2, 4>(a, vw, lds, grid, st);
    if (NT == 2) return launch_conv_vw<2, 2>(a, vw, lds, grid, st);
    return launch_conv_vw<2, 1>(a, vw, lds, grid, st);
  }
  if (NT == 4) return launch_conv_vw<1, 4>(a, vw, lds, grid, st);
  if (NT == 2) return launch_conv_vw<1, 2>(a, vw, lds, grid, st);
  return launch_conv_vw<1, 1>(a, vw, lds, grid, st);
}

}  // namespace

extern "C" int fdet_pack_conv3x3_weights(const float* w, int Cout, int Cin, float* wpk_fwd, float* wpk_bwd,
                                         void* stream) {
  FDET_REQUIRE(w && Cout > 0 && Cin > 0 && (wpk_fwd || wpk_bwd), "pack_conv3x3_weights: bad arguments");
  const int CoP = (Cout + 31) / 32 * 32, CiP = (Cin + 31) / 32 * 32;
  const int n = max(Cin * 9 * CoP, Cout * 9 * CiP);
  hipLaunchKernelGGL(k_pack3x3, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, CoP, CiP,
                     wpk_fwd, wpk_bwd);
  return check_launch("fdet_pack_conv3x3_weights");
}

extern "C" int fdet_conv3x3_fwd(const float* x, const float* wpk, const float* bias, float* y_full,
                                const float* skip, const float* drop_scale, float* y_out, int N, int Cin,
                                int Cout, int H, int W, int pool, float slope, void* stream) {
  FDET_REQUIRE(x && wpk && (y_full || y_out), "conv3x3_fwd: null pointer");
  FDET_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % CK == 0,
               "conv3x3_fwd: unsupported shape N=%d Cin=%d Cout=%d H=%d W=%d (Cin must be a multiple of %d)", N,
               Cin, Cout, H, W, CK);
  FDET_REQUIRE(pool == 1, "conv3x3_fwd: pooled tails go through fdet_block_tail_fwd (pool=%d)", pool);
  ConvArgs a{};
  a.x = x; a.wpk = wpk; a.bias = bias; a.y_full = y_full; a.skip = skip; a.scale = drop_scale; a.y_out = y_out;
  a.act = nullptr; a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.dgrad = 0; a.slope = slope;
  return run_conv(a, (hipStream_t)stream);
}

extern "C" int fdet_conv3x3_dgrad(const float* dz, const float* wpk, const float* act, const float* add,
                                  float* dx, int N, int Cin, int Cout, int H, int W, float slope, void* stream) {
  FDET_REQUIRE(dz && wpk && dx, "conv3x3_dgrad: null pointer");
  FDET_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cout % CK == 0,
               "conv3x3_dgrad: unsupported shape N=%d Cin=%d Cout=%d H=%d W=%d", N, Cin, Cout, H, W);
  ConvArgs a{};
  // the data gradient is a forward conv over dz: "input channels" = Cout, "output channels" = Cin
  a.x = dz; a.wpk = wpk; a.bias = nullptr; a.y_full = dx; a.skip = add; a.scale = nullptr; a.y_out = nullptr;
  a.act = act; a.N = N; a.Cin = Cout; a.Cout = Cin; a.H = H; a.W = W; a.dgrad = 1; a.slope = slope;
  return run_conv(a, (hipStream_t)stream);
}
