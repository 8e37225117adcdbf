// 3x3 convolutions (forward and data gradient) on the bf16 matrix cores with fp32-level accuracy:
// every fp32 operand x is split into two bf16 values  x = hi + lo  (hi = bf16(x), lo = bf16(x-hi))
// and a product is evaluated as  a_hi*b_lo + a_lo*b_hi + a_hi*b_hi  with fp32 accumulation
// ("bf16x3").  The dropped a_lo*b_lo term is 2^-16 relative, so results stay within ~1e-5 of the
// fp32 FMA chain (tests: 1e-4) while v_mfma_f32_32x32x16_bf16 delivers 16x the K per clock of the
// f32 MFMA: three passes are ~5x faster, which moves the 3x3 convs from MFMA-bound to HBM-bound.
//
// This file: weight-panel packing, tile selection and the C-ABI entry points.  The kernels live in
// fdet_conv3x3_x3_kernel.inc, compiled once per epilogue mode (fdet_conv3x3_x3_m*.hip).
#include "fdet_conv3x3_x3.h"
#include <algorithm>

namespace {

long long* g_probe_stamps = nullptr;
constexpr int PACK_MAXL = 32;
struct PackBatch { const float* w[PACK_MAXL]; bf16x8* fwd[PACK_MAXL]; bf16x8* bwd[PACK_MAXL]; };   // set only by the diagnostic build's fdet_x3_probe_set

// ---------------------------------------------------------------------------------------
// weight panels: split fp32 OIHW weights into bf16 hi/lo, K-major per 16-channel chunk
//   fwd:  unit ((c16*9 + tap)*2 + h)*CoP + co  holds W[co][16*c16 + 8h + j][tap],      j = 0..7
//   bwd:  unit ((o16*9 + tap)*2 + h)*CiP + ci  holds W[16*o16 + 8h + j][ci][8 - tap]
// Each panel is [hi units | lo units]; both halves together are exactly as large as the fp32
// panel of fdet_pack_conv3x3_weights, so callers size one buffer for either precision.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pack3x3_x3(const float* __restrict__ w, int Cout, int Cin, int CoP, int CiP, bf16x8* __restrict__ fwd,
             bf16x8* __restrict__ bwd) {
  const int nf = (Cin / 16) * 9 * 2 * CoP, nb = (Cout / 16) * 9 * 2 * CiP;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (fwd && t < nf) {
    const int co = t % CoP, r = t / CoP;
    const int h = r & 1, r2 = r >> 1, tap = r2 % 9, c16 = r2 / 9;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (co < Cout) ? w[((size_t)co * Cin + c16 * 16 + 8 * h + j) * 9 + tap] : 0.f;
    bf16x8 hi, lo;
    split8(f, hi, lo);
    fwd[t] = hi;
    fwd[nf + t] = lo;
  }
  if (bwd && t < nb) {
    const int ci = t % CiP, r = t / CiP;
    const int h = r & 1, r2 = r >> 1, tap = r2 % 9, o16 = r2 / 9;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (ci < Cin) ? w[((size_t)(o16 * 16 + 8 * h + j) * Cin + ci) * 9 + (8 - tap)] : 0.f;
    bf16x8 hi, lo;
    split8(f, hi, lo);
    bwd[t] = hi;
    bwd[nb + t] = lo;
  }
}

int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

int run_x3(ConvArgs a, hipStream_t st) {
  // rows of up to 64 columns: the small-tile kernel (two workgroups per CU) is the faster one
  // (FDET_CONV_KERNEL=general forces the persistent kernel)
  {
    static const char kernel_choice = [] { const char* e = getenv("FDET_CONV_KERNEL"); return e ? e[0] : '\0'; }();
    if (a.W <= 64 && kernel_choice != 'g') {
      // ping-pong kernel first (one 8-wave workgroup per CU, barrier-enforced MFMA / memory alternation), then the
      // round-1 small-tile kernel (FDET_CONV_KERNEL=s), then the general persistent kernel (=g)
      if (kernel_choice != 's') {
        const int rc = fdet_x3_pp_run(a, PoolArgs{nullptr, nullptr, nullptr, nullptr}, st);
        if (rc != 1) return rc;
      }
      // aligned-band variant of the small-tile kernel (no separator rows, shared epilogue with two-instruction quad
      // exchanges): measured -6 % / -7 % on the 60x60 forward / data-gradient launches, a wash at 30x30 (bands of 8 rows
      // cover 32 of 30): default for rows of 33..63 columns; FDET_SB_AL=0 / 1 forces it off / on for every width
      static const int sb_al = [] { const char* e = getenv("FDET_SB_AL"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
      if (sb_al == 1 || (sb_al < 0 && a.W >= 33)) {
        const int rc = fdet_x3_sb_pool_run(a, PoolArgs{nullptr, nullptr, nullptr, nullptr}, st);
        if (rc != 1) return rc;
      }
      const int rc = fdet_x3_sb_run(a, st);
      if (rc != 1) return rc;
    }
  }
  a.VR = a.N * (a.H + 1) + 1;
  if (a.VR >= (1 << 20)) return fail(FDET_EINVAL, "conv3x3_bf16x3: N*(H+1)=%d virtual rows exceed the index range", a.VR);
  if ((size_t)a.N * (size_t)max(a.Cin, a.Cout) * a.H * a.W >= (size_t)1 << 31)
    return fail(FDET_EINVAL, "conv3x3_bf16x3: tensor too large for 32-bit offsets");
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
  } else if (a.Cout % 16 == 0 && a.dgrad) {                // the data-gradient epilogues guard the upper half tile themselves
    if (a.act && !a.skip) a.mode = EPI_DGRAD_ACT;
    else if (!a.act && a.skip) a.mode = EPI_DGRAD_ADD;
  }
  // vector width of the global accesses: rows of W floats must keep VW-float alignment
  auto aligned = [](const void* q, size_t b) { return q == nullptr || ((uintptr_t)q % b) == 0; };
  int VW = 1;
  if (a.W % 4 == 0 && aligned(a.x, 16) && aligned(a.y_full, 16) && aligned(a.y_out, 16) && aligned(a.skip, 16) &&
      aligned(a.act, 16)) VW = 4;
  else if (a.W % 2 == 0 && aligned(a.x, 8)) VW = 2;
  const int rows_total = a.VR - 1;
  const int ncu = num_cus();
  const int MT = (a.CoP % 64 == 0) ? 2 : 1;
  const int ncob = a.CoP / (MT * 32);
  // Tile choice: (NW waves, NT tiles per wave): positions per workgroup = 32*NW*NT; column segmentation.
  // measured on MI355X (64 channels, 60x60 / 30x30 / 15x15): microseconds per 128 positions of a tile
  static const int cfgs2[][2] = {{8, 1}, {4, 4}, {8, 2}};     // must match fdet_conv3x3_x3_configs.h
  static const double kcfg2[] = {8.1, 8.5, 10.8};
  static const int cfgs1[][2] = {{8, 1}, {8, 2}};
  // 32-channel tiles only reach this kernel on rows wider than 64 columns (the SSD trunk's 240 / 120-column levels), where the
  // step is HBM-bound and the two halo rows of a band are the cost: {8, 2} (two / four rows per band instead of one / two)
  // measured 15-25 % faster there (round 4: 0.44 -> 0.37 ms forward at 240 columns, 0.23 -> 0.18 at 120)
  static const double kcfg1[] = {8.1, 6.5};
  const int (*cfgs)[2] = MT == 2 ? cfgs2 : cfgs1;
  const double* kcfg = MT == 2 ? kcfg2 : kcfg1;
  const int ncfg = MT == 2 ? 3 : 2;
  int bestNW = 0, bestNT = 0, bestR = 0, bestCW = 0, bestNSEG = 0, bestWP = 0; double bestT = 0;
  int forceNW = 0, forceNT = 0;
  if (const char* e = getenv("FDET_CONV_TILE")) sscanf(e, "%d,%d", &forceNW, &forceNT);
  for (int ci = 0; ci < ncfg; ++ci) {
    const int NW = cfgs[ci][0], NT = cfgs[ci][1], nthr = NW * 64;
    if (forceNT && (NT != forceNT || NW != forceNW)) continue;
    const int cap = NW * NT * 32;
    // widest segment whose double-buffered tile fits LDS
    for (int NSEG = 1; NSEG <= 64; ++NSEG) {
      int CW = (a.W + NSEG - 1) / NSEG;
      if (NSEG > 1) CW = (CW + 3) / 4 * 4;
      if (NSEG > 1 && (NSEG - 1) * CW >= a.W) continue;
      const int WP = ((NSEG > 1 ? CW + 2 : a.W + 1) + 3) / 4 * 4;
      if (WP > cap) continue;
      int R = cap / WP;
      if (R > rows_total) R = rows_total;
      const int PT = cap + 2 * WP + 3;
      const size_t lds = ((size_t)2 * (2 * 9 * 2 * MT * 32 + 4 * PT) + nthr) * 16;
      if (lds > 160 * 1024) continue;
      if (NSEG > 1 && VW != 4) continue;                              // segmented rows: W % 4 == 0 only
      if (2 * (R + 2) * (CW / VW) > nbs_of(NW, NT, VW) * nthr) continue;
      if (NSEG > 1 && 4 * (R + 2) > nthr) continue;
      const long nb = (rows_total + R - 1) / R;
      const long tiles = nb * NSEG * ncob;
      const long rounds = (tiles + ncu - 1) / ncu;
      // per tile: MFMA work ~ cap, plus an un-overlapped share; halo rows/columns are re-read
      const double t = rounds * (cap / 128.0) * kcfg[ci] * (1.0 + 0.1 * (2.0 / R + (NSEG > 1 ? 2.0 / CW : 0.0)));
      if (bestNT == 0 || t < bestT) { bestNW = NW; bestNT = NT; bestR = R; bestCW = CW; bestNSEG = NSEG; bestWP = WP; bestT = t; }
      break;                                                          // first NSEG that fits is the widest
    }
  }
  if (bestNT == 0) return fail(FDET_EINVAL, "conv3x3_bf16x3: no tiling for W=%d H=%d", a.W, a.H);
  const int NT = bestNT, NW = bestNW;
  const int cap = NW * NT * 32;
  a.R = bestR;
  a.WP = bestWP;
  a.nbands = (rows_total + a.R - 1) / a.R;
  a.magic_h1 = magic_of(a.H + 1);
  X3Args p;
  p.PT = cap + 2 * a.WP + 3;
  p.CW = bestCW;
  p.NSEG = bestNSEG;
  p.ncob = ncob;
  p.ntiles = a.nbands * bestNSEG * ncob;
  p.WV = bestCW / VW;
  p.items_half = (a.R + 2) * p.WV;
  p.magic_wv = magic_of(p.WV);
  p.magic_wp = magic_of(a.WP);
  const size_t units = (size_t)(a.Cin / 16) * 9 * 2 * a.CoP;       // per hi / lo half
  p.a_hi = reinterpret_cast<const bf16x8*>(a.wpk);
  p.a_lo = p.a_hi + units;
  p.c = a;
  const size_t lds = ((size_t)2 * (2 * 9 * 2 * MT * 32 + 4 * p.PT) + NW * 64) * 16;
  const int grid = p.ntiles < ncu ? p.ntiles : ncu;
  p.c.stagger = 0;
  if (NW == 4 && p.ntiles >= 4 * grid) p.c.stagger = cap / 128 * 9000;   // ~ one tile time, in shader clocks
  if (const char* e = getenv("FDET_CONV_STAGGER")) p.c.stagger = atoi(e);
  const bool seg = bestNSEG > 1;
  p.stamps = g_probe_stamps;
  switch (a.mode) {
    case EPI_FWD_FULL: return fdet_x3_launch_m1(p, MT, NW, NT, VW, seg, lds, grid, st);
    case EPI_FWD_BOTH: return fdet_x3_launch_m2(p, MT, NW, NT, VW, seg, lds, grid, st);
    case EPI_FWD_OUT: return fdet_x3_launch_m3(p, MT, NW, NT, VW, seg, lds, grid, st);
    case EPI_DGRAD_ACT: return fdet_x3_launch_m4(p, MT, NW, NT, VW, seg, lds, grid, st);
    case EPI_DGRAD_ADD: return fdet_x3_launch_m5(p, MT, NW, NT, VW, seg, lds, grid, st);
    default: return fdet_x3_launch_m0(p, MT, NW, NT, VW, seg, lds, grid, st);
  }
}

}  // namespace

#ifdef FDET_X3_STAMPS
extern "C" int fdet_x3_probe_set(long long* buf) { g_probe_stamps = buf; return 0; }
#endif

extern "C" int fdet_pack_conv3x3_weights_bf16x3(const float* w, int Cout, int Cin, void* wpk_fwd, void* wpk_bwd,
                                                void* stream) {
  FDET_REQUIRE(w && Cout > 0 && Cin > 0 && (wpk_fwd || wpk_bwd), "pack_conv3x3_weights_bf16x3: bad arguments");
  FDET_REQUIRE(Cin % 16 == 0 && Cout % 16 == 0, "pack_conv3x3_weights_bf16x3: channel counts must be multiples of 16 (Cin=%d Cout=%d)", Cin, Cout);
  const int CoP = (Cout + 31) / 32 * 32, CiP = (Cin + 31) / 32 * 32;
  const int n = max((Cin / 16) * 9 * 2 * CoP, (Cout / 16) * 9 * 2 * CiP);
  hipLaunchKernelGGL(k_pack3x3_x3, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, CoP, CiP,
                     (bf16x8*)wpk_fwd, (bf16x8*)wpk_bwd);
  return check_launch("fdet_pack_conv3x3_weights_bf16x3");
}

__global__ void __launch_bounds__(256)
k_pack3x3_x3_batched(const PackBatch b, int Cout, int Cin, int CoP, int CiP) {
  const float* __restrict__ w = b.w[blockIdx.y];
  bf16x8* __restrict__ fwd = b.fwd[blockIdx.y];
  bf16x8* __restrict__ bwd = b.bwd[blockIdx.y];
  const int nf = (Cin / 16) * 9 * 2 * CoP, nb = (Cout / 16) * 9 * 2 * CiP;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (fwd && t < nf) {
    const int co = t % CoP, r = t / CoP;
    const int h = r & 1, r2 = r >> 1, tap = r2 % 9, c16 = r2 / 9;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (co < Cout) ? w[((size_t)co * Cin + c16 * 16 + 8 * h + j) * 9 + tap] : 0.f;
    bf16x8 hi, lo;
    split8(f, hi, lo);
    fwd[t] = hi;
    fwd[nf + t] = lo;
  }
  if (bwd && t < nb) {
    const int ci = t % CiP, r = t / CiP;
    const int h = r & 1, r2 = r >> 1, tap = r2 % 9, o16 = r2 / 9;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (ci < Cin) ? w[((size_t)(o16 * 16 + 8 * h + j) * Cin + ci) * 9 + (8 - tap)] : 0.f;
    bf16x8 hi, lo;
    split8(f, hi, lo);
    bwd[t] = hi;
    bwd[nb + t] = lo;
  }
}

extern "C" int fdet_pack_conv3x3_weights_bf16x3_batched(const float* const* h_w, int L, int Cout, int Cin,
                                                        void* const* h_wpk_fwd, void* const* h_wpk_bwd, void* stream) {
  FDET_REQUIRE(h_w && L >= 1 && Cout > 0 && Cin > 0 && (h_wpk_fwd || h_wpk_bwd), "pack_conv3x3_weights_bf16x3_batched: bad arguments");
  FDET_REQUIRE(Cin % 16 == 0 && Cout % 16 == 0, "pack_conv3x3_weights_bf16x3_batched: channel counts must be multiples of 16 (Cin=%d Cout=%d)", Cin, Cout);
  const int CoP = (Cout + 31) / 32 * 32, CiP = (Cin + 31) / 32 * 32;
  const int n = max((Cin / 16) * 9 * 2 * CoP, (Cout / 16) * 9 * 2 * CiP);
  for (int l0 = 0; l0 < L; l0 += PACK_MAXL) {
    const int nl = min(PACK_MAXL, L - l0);
    PackBatch b{};
    for (int l = 0; l < nl; ++l) {
      FDET_REQUIRE(h_w[l0 + l], "pack_conv3x3_weights_bf16x3_batched: null weight pointer (layer %d)", l0 + l);
      b.w[l] = h_w[l0 + l];
      b.fwd[l] = h_wpk_fwd ? (bf16x8*)h_wpk_fwd[l0 + l] : nullptr;
      b.bwd[l] = h_wpk_bwd ? (bf16x8*)h_wpk_bwd[l0 + l] : nullptr;
    }
    hipLaunchKernelGGL(k_pack3x3_x3_batched, dim3((n + 255) / 256, nl), dim3(256), 0, (hipStream_t)stream, b, Cout, Cin, CoP, CiP);
    if (int rc = check_launch("fdet_pack_conv3x3_weights_bf16x3_batched")) return rc;
  }
  return FDET_OK;
}

extern "C" int fdet_conv3x3_fwd_bf16x3(const float* x, const void* wpk, const float* bias, float* y_full,
                                       const float* skip, const float* drop_scale, float* y_out, int N, int Cin,
                                       int Cout, int H, int W, int pool, float slope, void* stream) {
  FDET_REQUIRE(x && wpk && (y_full || y_out), "conv3x3_fwd_bf16x3: null pointer");
  FDET_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 16 == 0,
               "conv3x3_fwd_bf16x3: unsupported shape N=%d Cin=%d Cout=%d H=%d W=%d (channels must be multiples of 16)",
               N, Cin, Cout, H, W);
  FDET_REQUIRE(pool == 1, "conv3x3_fwd_bf16x3: pooled tails go through fdet_block_tail_fwd (pool=%d)", pool);
  ConvArgs a{};
  a.x = x; a.wpk = (const float*)wpk; a.bias = bias; a.y_full = y_full; a.skip = skip; a.scale = drop_scale;
  a.y_out = y_out; a.act = nullptr; a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.dgrad = 0; a.slope = slope;
  return run_x3(a, (hipStream_t)stream);
}

// pooled-block modes: aligned-band small-tile kernel (two workgroups per CU), or the ping-pong kernel (FDET_POOL_KERNEL=pp)
static int run_x3_pooled(const ConvArgs& a, const PoolArgs& q, hipStream_t st) {
  static const bool use_pp = [] { const char* e = getenv("FDET_POOL_KERNEL"); return e && e[0] == 'p'; }();
  if (!use_pp) {
    const int rc = fdet_x3_sb_pool_run(a, q, st);
    if (rc != 1) return rc;
  }
  return fdet_x3_pp_run(a, q, st);
}

// 1 when the fused pooled-block kernels (fdet_conv3x3_fwd_pool_bf16x3 / fdet_conv3x3_dgrad_unpool_bf16x3) have a tiling
// for the shape -- the same conditions fdet_x3_sb_pool_run checks before it launches (even map of <= 62 columns, channel
// multiples, 32-bit element offsets, < 2^20 (image, band) tiles) -- so a caller can choose the separate conv + tail
// kernels up front instead of catching FDET_EINVAL.
extern "C" int fdet_conv3x3_pool_fusion_ok(int N, int Cin, int Cout, int H, int W) {
  if (N < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || W > 62 || Cin % 16 != 0 || Cout % 32 != 0 || Cin % 32 != 0) return 0;
  if ((size_t)N * (size_t)std::max(Cin, Cout) * H * W >= ((size_t)1 << 31)) return 0;
  const int WP = W <= 31 ? 32 : 64, R = std::min(256 / WP, (H + 1) & ~1);
  const long bands = (long)N * ((H + R - 1) / R);
  return bands < (1 << 20) ? 1 : 0;
}

extern "C" int fdet_conv3x3_fwd_pool_bf16x3(const float* x, const void* wpk, const float* bias, const float* skip,
                                            const float* drop_scale, float* out_pooled, unsigned char* route, int N,
                                            int Cin, int Cout, int H, int W, float slope, void* stream) {
  FDET_REQUIRE(x && wpk && bias && skip && out_pooled, "conv3x3_fwd_pool_bf16x3: null pointer");
  FDET_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 32 == 0 && !(H & 1) && !(W & 1),
               "conv3x3_fwd_pool_bf16x3: unsupported shape N=%d Cin=%d Cout=%d H=%d W=%d (even maps, Cin %% 16 == 0, Cout %% 32 == 0)",
               N, Cin, Cout, H, W);
  ConvArgs a{};
  a.x = x; a.wpk = (const float*)wpk; a.bias = bias; a.y_full = nullptr; a.skip = skip; a.scale = drop_scale;
  a.y_out = nullptr; a.act = nullptr; a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.dgrad = 0; a.slope = slope;
  const int rc = run_x3_pooled(a, PoolArgs{out_pooled, route, nullptr, nullptr}, (hipStream_t)stream);
  return rc == 1 ? fail(FDET_EINVAL, "conv3x3_fwd_pool_bf16x3: no tiling for H=%d W=%d", H, W) : rc;
}

extern "C" int fdet_conv3x3_dgrad_unpool_bf16x3(const float* dz, const void* wpk, const float* dout_pooled,
                                                const unsigned char* route, float* dx, int N, int Cin, int Cout,
                                                int H, int W, float slope, void* stream) {
  FDET_REQUIRE(dz && wpk && dout_pooled && route && dx, "conv3x3_dgrad_unpool_bf16x3: null pointer");
  FDET_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 16 == 0 && !(H & 1) && !(W & 1),
               "conv3x3_dgrad_unpool_bf16x3: unsupported shape N=%d Cin=%d Cout=%d H=%d W=%d", N, Cin, Cout, H, W);
  ConvArgs a{};
  a.x = dz; a.wpk = (const float*)wpk; a.bias = nullptr; a.y_full = dx; a.skip = nullptr; a.scale = nullptr; a.y_out = nullptr;
  a.act = nullptr; a.N = N; a.Cin = Cout; a.Cout = Cin; a.H = H; a.W = W; a.dgrad = 1; a.slope = slope;
  const int rc = run_x3_pooled(a, PoolArgs{nullptr, nullptr, dout_pooled, route}, (hipStream_t)stream);
  return rc == 1 ? fail(FDET_EINVAL, "conv3x3_dgrad_unpool_bf16x3: no tiling for H=%d W=%d", H, W) : rc;
}

extern "C" int fdet_conv3x3_dgrad_bf16x3(const float* dz, const void* wpk, const float* act, const float* add,
                                         float* dx, int N, int Cin, int Cout, int H, int W, float slope, void* stream) {
  FDET_REQUIRE(dz && wpk && dx, "conv3x3_dgrad_bf16x3: null pointer");
  FDET_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 16 == 0,
               "conv3x3_dgrad_bf16x3: unsupported shape N=%d Cin=%d Cout=%d H=%d W=%d", N, Cin, Cout, H, W);
  ConvArgs a{};
  a.x = dz; a.wpk = (const float*)wpk; a.bias = nullptr; a.y_full = dx; a.skip = add; a.scale = nullptr; a.y_out = nullptr;
  a.act = act; a.N = N; a.Cin = Cout; a.Cout = Cin; a.H = H; a.W = W; a.dgrad = 1; a.slope = slope;
  return run_x3(a, (hipStream_t)stream);
}
