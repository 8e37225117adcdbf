// 3x3 convolutions (forward and data gradient) on the bf16 matrix cores with fp32-level accuracy:
// every fp32 operand x is split into two bf16 values  x = hi + lo  (hi = bf16(x), lo = bf16(x-hi))
// and a product is evaluated as  a_hi*b_lo + a_lo*b_hi + a_hi*b_hi  with fp32 accumulation
// ("bf16x3").  The dropped a_lo*b_lo term is 2^-16 relative, so results stay within ~1e-5 of the
// fp32 FMA chain (tests: 1e-4) while v_mfma_f32_32x32x16_bf16 delivers 16x the K per clock of the
// f32 MFMA: three passes are ~5x faster, which moves the 3x3 convs from MFMA-bound to HBM-bound.
//
// Same "flattened padded rows" implicit GEMM as fdet_conv3x3.hip, but the MFMA K index is 16
// input channels at ONE tap, so LDS tiles are channel-innermost:
//   B: four arrays  {hi,lo} x {k-half h}  of  [position][8 x bf16]   (16 B per position)
//   A: {hi,lo} x [tap][h][co][8 x bf16]                              (pre-split panels in HBM)
// A lane's fragment is one aligned 16-byte ds_read_b128; consecutive lanes read consecutive
// 16-byte slots (conflict-free); a tap is still a constant position offset.
// The fp32 -> (hi,lo) split of activations happens while staging global -> registers -> LDS.
// One workgroup (4 waves, one per SIMD, 512-register budget) per CU; double-buffered LDS.
#include "fdet_conv_common.h"

using namespace fdet;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int CK16 = 16;     // input channels per chunk (= MFMA K)
constexpr int NBS = 5;       // B staging slots per thread (8 fp32 loads each)

struct X3Args {
  ConvArgs c;                // x, bias, epilogue pointers, geometry (WP, R, VR, nbands, mode ...)
  const bf16x8* a_hi;        // [Cin/16][9][2][CoP] x 8 bf16
  const bf16x8* a_lo;
  int PT;                    // positions per B array (cap + 2*WP + 3)
  int p_in;                  // (R+2)*W staged positions per k-half
  unsigned magic_w;
};

__device__ __forceinline__ void split8(const float (&f)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)f[j];
    hi[j] = h;
    lo[j] = (__bf16)(f[j] - (float)h);
  }
}

template <int MT, int NT>
__global__ void __launch_bounds__(NTHR, 1)
k_conv3x3_x3(const X3Args p) {
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
  const int band = blockIdx.x, mb = blockIdx.y;
  const int v0 = band * a.R;
  const int H1 = a.H + 1;
  const size_t HW = (size_t)a.H * a.W;

  {  // zero both buffers: halo positions of the B arrays are never written again
    f32x4* z = reinterpret_cast<f32x4*>(smem);
    for (int t = tid; t < 2 * buf_units; t += NTHR) z[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- B staging geometry (chunk invariant): slot -> (k-half, tile row, column)
  int b_src[NBS], b_dst[NBS];                         // src: element offset for channel 8h of chunk 0 (-1: skip)
#pragma unroll
  for (int s = 0; s < NBS; ++s) {
    const int it = s * NTHR + tid;
    b_src[s] = -1; b_dst[s] = 0;
    if (it < 2 * p.p_in) {
      const int h = it >= p.p_in ? 1 : 0;
      const int pp = it - h * p.p_in;
      const int tr = fdiv(pp, p.magic_w), ix = pp - tr * a.W;
      const int v = v0 - 1 + tr;
      if (v >= 0 && v < a.VR) {
        const int n = fdiv(v, a.magic_h1), yy = v - n * H1 - 1;
        if (yy >= 0) {
          b_src[s] = ((n * a.Cin + 8 * h) * a.H + yy) * a.W + ix;
          b_dst[s] = h * PT + tr * WP + 1 + ix;       // unit index inside the hi array pair; lo = +2*PT
        }
      }
    }
  }
  const int a_chunk_units = 9 * 2 * a.CoP;            // units per chunk per array in HBM
  bf16x8 pa[NA];
  float pb[NBS][8];
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
      _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) pb[s_][j_] = q_[j_ * HW];                   \
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
      if (b_src[s_] >= 0) {                                                                        \
        bf16x8 hi_, lo_;                                                                           \
        split8(pb[s_], hi_, lo_);                                                                  \
        B_[b_dst[s_]] = hi_;                                                                       \
        B_[b_dst[s_] + 2 * PT] = lo_;                                                              \
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

  const int qwave = wid * NT * 32;
  const int a_off = half * MB + l31;                           // + tap*2*MB + m*32 ; lo: + A_UNITS
  const int b_off = 2 * A_UNITS + half * PT + qwave + l31;     // + tapoff + n*32   ; lo: + 2*PT

  X3_ISSUE_LOADS(0)
  __syncthreads();                       // zero fill complete
  X3_WRITE_LDS(lds)
  __syncthreads();

  const int nch = a.Cin / CK16;
  for (int c = 0; c < nch; ++c) {
    const bf16x8* buf = lds + (c & 1) * buf_units;
    if (c + 1 < nch) X3_ISSUE_LOADS(c + 1)
    const bf16x8* Aw = buf + a_off;
    const bf16x8* Bw = buf + b_off;
    // fragments of tap t+1 are fetched before the MFMAs of tap t
    bf16x8 ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { ah[0][m] = Aw[m * 32]; al[0][m] = Aw[A_UNITS + m * 32]; }
#pragma unroll
    for (int n = 0; n < NT; ++n) { bh[0][n] = Bw[tapoff[0] + n * 32]; bl[0][n] = Bw[2 * PT + tapoff[0] + n * 32]; }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int cur = t & 1, nxt = cur ^ 1;
      if (t + 1 < 9) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          ah[nxt][m] = Aw[(t + 1) * 2 * MB + m * 32];
          al[nxt][m] = Aw[A_UNITS + (t + 1) * 2 * MB + m * 32];
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          bh[nxt][n] = Bw[tapoff[t + 1] + n * 32];
          bl[nxt][n] = Bw[2 * PT + tapoff[t + 1] + n * 32];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][m], bl[cur][n], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cur][m], bh[cur][n], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][m], bh[cur][n], acc[m][n], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (c + 1 < nch) X3_WRITE_LDS(lds + ((c + 1) & 1) * buf_units)
    __syncthreads();
  }

  // ---- epilogue (shared with the fp32 kernel: same accumulator layout)
  const int qlimit = a.R * WP;
  bool okn[NT];
  size_t basen[NT];
  int imgn[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int q = qwave + n * 32 + l31;
    const int tr = q / WP, ox = q - tr * WP;
    const int v = v0 + tr;
    const int img = v / H1, oy = v - img * H1 - 1;
    okn[n] = (q < qlimit) && (ox < a.W) && (v < a.VR) && (oy >= 0);
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

template <int MT, int NT>
int launch_x3(const X3Args& p, size_t lds, dim3 grid, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)k_conv3x3_x3<MT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_conv3x3_x3<MT, NT>), grid, dim3(NTHR), lds, st, p);
  return check_launch("fdet_conv3x3_bf16x3");
}

int run_x3(ConvArgs a, hipStream_t st) {
  a.WP = a.W + 1;
  a.VR = a.N * (a.H + 1) + 1;
  if (a.VR >= (1 << 20)) return fail(FDET_EINVAL, "conv3x3_bf16x3: N*(H+1)=%d virtual rows exceed the index range", a.VR);
  if ((size_t)a.N * a.Cin * a.H * a.W >= (size_t)1 << 31) return fail(FDET_EINVAL, "conv3x3_bf16x3: tensor too large for 32-bit offsets");
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
  int bestNT = 0, bestMT = 0, bestR = 0; long bestT = 0;
  int forceMT = 0, forceNT = 0;
  if (const char* e = getenv("FDET_CONV_TILE")) sscanf(e, "%d,%d", &forceMT, &forceNT);
  for (int MT = (a.CoP % 64 == 0) ? 2 : 1; MT >= 1; --MT)
    for (int NT = 4; NT >= 1; NT >>= 1) {
      if (forceMT && (MT != forceMT || NT != forceNT)) continue;
      const int cap = 4 * NT * 32;
      if (a.WP > cap) continue;
      int R = cap / a.WP;
      if (R > rows_total) R = rows_total;
      if (2 * (R + 2) * a.W > NBS * NTHR) continue;
      const int PT = cap + 2 * a.WP + 3;
      const size_t lds = (size_t)2 * (2 * 9 * 2 * MT * 32 + 4 * PT) * 16;
      if (lds > 160 * 1024) continue;
      const long nb = (rows_total + R - 1) / R;
      const long waves = nb * (a.CoP / (MT * 32)) * 4;
      // one workgroup per CU: rounds of 1024 waves; fixed per-workgroup cost ~ 2 tile-jobs
      const long t = ((waves + 1023) / 1024) * (MT * NT + 2);
      if (bestNT == 0 || t < bestT) { bestNT = NT; bestMT = MT; bestR = R; bestT = t; }
    }
  if (bestNT == 0) return fail(FDET_EINVAL, "conv3x3_bf16x3: no tiling for W=%d H=%d", a.W, a.H);
  const int NT = bestNT, MT = bestMT;
  const int cap = 4 * NT * 32;
  a.R = bestR;
  a.nbands = (rows_total + a.R - 1) / a.R;
  a.magic_h1 = magic_of(a.H + 1);
  X3Args p;
  p.PT = cap + 2 * a.WP + 3;
  p.p_in = (a.R + 2) * a.W;
  p.magic_w = magic_of(a.W);
  const size_t units = (size_t)(a.Cin / 16) * 9 * 2 * a.CoP;       // per hi / lo half
  p.a_hi = reinterpret_cast<const bf16x8*>(a.wpk);
  p.a_lo = p.a_hi + units;
  p.c = a;
  const size_t lds = (size_t)2 * (2 * 9 * 2 * MT * 32 + 4 * p.PT) * 16;
  dim3 grid(a.nbands, a.CoP / (MT * 32));
  if (MT == 2) {
    if (NT == 4) return launch_x3<2, 4>(p, lds, grid, st);
    if (NT == 2) return launch_x3<2, 2>(p, lds, grid, st);
    return launch_x3<2, 1>(p, lds, grid, st);
  }
  if (NT == 4) return launch_x3<1, 4>(p, lds, grid, st);
  if (NT == 2) return launch_x3<1, 2>(p, lds, grid, st);
  return launch_x3<1, 1>(p, lds, grid, st);
}

}  // namespace

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
