// PoolResnet stem (Conv2d(3,F,10,stride 8,pad 2), models/PoolResnet.py:70-76) on fp32 MFMA:
// forward and weight gradient.  One band = one output row (n, oy): the 10 input rows it needs
// (3 channels x 10 rows x W) are staged in LDS DE-INTERLEAVED by column phase
//     ix + 2 = 8*bx + phase   ->   X[ci][row][phase][bx]
// so that tap (ky,kx) of output column ox sits at   base(ci,ky,kx) + ox   : 32 consecutive
// dwords for the 32 lanes of an MFMA operand, and every tap is an immediate offset.
//
// forward : D[co][ox]  += A[co][k] * B[k][ox]     k = (ci,ky,kx) = 300, two ky per MFMA
//           A (weights) lives in 150 VGPRs per wave for the whole kernel (persistent grid).
// wgrad   : dW[co][k]  += dy[co][ox] * B[ox][k]   K = ox pairs, N = k (10 tiles of 32)
#include "fdet_common.h"

using namespace fdet;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KS = 10, ST = 8, PD = 2, CIN = 3;
constexpr int KK = CIN * KS * KS;          // 300
constexpr int BXS = 66;                    // bx 0..64 used (+1 pad)
constexpr int RS = ST * BXS;               // LDS stride between input rows
constexpr int CIS = KS * RS;               // LDS stride between channels
constexpr int XT = CIN * CIS;              // floats of the x tile (15840 = 61.9 KB)
constexpr int NSLOT = 15;                  // float4 staging slots per thread: 30 rows x 128 lanes / 256

struct StemArgs {
  const float* x;      // [N,3,H,W]
  const float* w;      // fwd: [F,3,10,10]
  const float* bias;   // fwd
  float* y;            // fwd: [N,F,Ho,Wo]
  const float* dy;     // wgrad: [N,F,Ho,Wo]
  float* ws;           // wgrad: [nblk][FP][320]
  float* wsb;          // wgrad: [nblk][FP]
  int N, F, H, W, Ho, Wo, nrows;
};

// prefetch one band of x into registers: slot s covers (row = (s*256+tid)>>7, j = &127), rows = ci*10+ky
#define STEM_LOAD_X(ROW_N, ROW_OY)                                                              \
  {                                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < NSLOT; ++s_) {                                      \
      const int it = s_ * 256 + tid;                                                            \
      const int rr = it >> 7, j = it & 127;                                                     \
      const int ci = rr / KS, ky = rr - ci * KS;                                                \
      const int iy = (ROW_OY) * ST - PD + ky;                                                   \
      const bool ok = j < jmax && iy >= 0 && iy < a.H;                                          \
      px[s_] = ok ? *reinterpret_cast<const f32x4*>(a.x + (((size_t)(ROW_N) * CIN + ci) * a.H + iy) * a.W + j * 4) \
                  : f32x4{0.f, 0.f, 0.f, 0.f};                                                  \
    }                                                                                           \
  }
// scatter the prefetched band into the de-interleaved tile (ix' = 4j+2+c)
#define STEM_STORE_X()                                                                          \
  {                                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < NSLOT; ++s_) {                                      \
      const int it = s_ * 256 + tid;                                                            \
      const int rr = it >> 7, j = it & 127;                                                     \
      if (j < jmax) {                                                                           \
        float* rowp = X + (rr / KS) * CIS + (rr % KS) * RS;                                     \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) {                                      \
          const int ixp = 4 * j + PD + c_;                                                      \
          rowp[(ixp & 7) * BXS + (ixp >> 3)] = px[s_][c_];                                      \
        }                                                                                       \
      }                                                                                         \
    }                                                                                           \
  }

// one workgroup (4 waves, one per SIMD) per CU: each wave may use the full 512-register file
__global__ void __launch_bounds__(256, 1)
k_stem_fwd_mfma(const StemArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* X = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int m = wid & 1, nt = wid >> 1;
  const int cob = blockIdx.y;
  const int co = cob * 64 + m * 32 + l31;
  const int jmax = a.W / 4;

  for (int t = tid; t < XT; t += 256) X[t] = 0.f;

  // A fragments: av[p], p = (ci*5 + kyp)*10 + kx, lane half selects ky = 2*kyp + half
  float av[KK / 2];
#pragma unroll
  for (int p = 0; p < KK / 2; ++p) {
    const int ci = p / 50, r = p - ci * 50, kyp = r / 10, kx = r - kyp * 10;
    av[p] = (co < a.F) ? a.w[((size_t)co * CIN + ci) * (KS * KS) + (2 * kyp + half) * KS + kx] : 0.f;
  }
  const float* Bl = X + half * RS + nt * 32 + l31;

  f32x4 px[NSLOT];
  int row = blockIdx.x;
  if (row < a.nrows) { const int n = row / a.Ho, oy = row - n * a.Ho; STEM_LOAD_X(n, oy) }
  for (; row < a.nrows; row += gridDim.x) {
    const int n = row / a.Ho, oy = row - n * a.Ho;
    __syncthreads();
    STEM_STORE_X()
    __syncthreads();
    const int nrow = row + gridDim.x;
    if (nrow < a.nrows) { const int n2 = nrow / a.Ho, oy2 = nrow - n2 * a.Ho; STEM_LOAD_X(n2, oy2) }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int p = 0; p < KK / 2; ++p) {
      const int ci = p / 50, r = p - ci * 50, kyp = r / 10, kx = r - kyp * 10;
      const float bv = Bl[ci * CIS + (2 * kyp) * RS + (kx & 7) * BXS + (kx >> 3)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p], bv, acc, 0, 0, 0);
    }
    const int ox = nt * 32 + l31;
    if (ox < a.Wo) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c2 = cob * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (c2 < a.F) a.y[(((size_t)n * a.F + c2) * a.Ho + oy) * a.Wo + ox] = acc[r] + a.bias[c2];
      }
    }
  }
}

constexpr int DYS = 67;                    // dy tile row stride (odd: conflict-free column reads), 64 + pad
constexpr int DSLOT = 4;                   // float4 slots per thread for the dy row: 64 co x 16 lanes / 256

__global__ void __launch_bounds__(256, 1)
k_stem_wgrad_mfma(const StemArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* X = reinterpret_cast<float*>(smem);           // de-interleaved x tile
  float* D = X + XT;                                   // [64][DYS] dy tile, zero beyond Wo
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int m = wid & 1, ng = wid >> 1;                // co tile, k-tile group (5 tiles each)
  const int cob = blockIdx.y;
  const int jmax = a.W / 4, dmax = a.Wo / 4;

  for (int t = tid; t < XT + 64 * DYS; t += 256) X[t] = 0.f;

  // per-lane LDS base of B for each of this wave's 5 k tiles
  int kbase[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int k = (ng * 5 + t) * 32 + l31;
    const int kc = min(k, KK - 1);
    const int ci = kc / 100, r = kc - ci * 100, ky = r / 10, kx = r - ky * 10;
    kbase[t] = ci * CIS + ky * RS + (kx & 7) * BXS + (kx >> 3) + half;   // k >= 300: garbage columns, dropped
  }
  const float* Al = D + (m * 32 + l31) * DYS + half;

  f32x16 acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  f32x4 px[NSLOT], pd[DSLOT];
#define STEM_LOAD_D(ROW_N, ROW_OY)                                                              \
  {                                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < DSLOT; ++s_) {                                      \
      const int it = s_ * 256 + tid;                                                            \
      const int c = it >> 4, j = it & 15;                                                       \
      const int cg = cob * 64 + c;                                                              \
      pd[s_] = (j < dmax && cg < a.F)                                                           \
                   ? *reinterpret_cast<const f32x4*>(a.dy + (((size_t)(ROW_N) * a.F + cg) * a.Ho + (ROW_OY)) * a.Wo + j * 4) \
                   : f32x4{0.f, 0.f, 0.f, 0.f};                                                 \
    }                                                                                           \
  }
  int row = blockIdx.x;
  if (row < a.nrows) { const int n = row / a.Ho, oy = row - n * a.Ho; STEM_LOAD_X(n, oy) STEM_LOAD_D(n, oy) }
  const int npair = (a.Wo + 1) / 2;
  for (; row < a.nrows; row += gridDim.x) {
    __syncthreads();
    STEM_STORE_X()
#pragma unroll
    for (int s_ = 0; s_ < DSLOT; ++s_) {
      const int it = s_ * 256 + tid;
      const int c = it >> 4, j = it & 15;
      if (j < dmax) {
#pragma unroll
        for (int c_ = 0; c_ < 4; ++c_) D[c * DYS + j * 4 + c_] = pd[s_][c_];
      }
    }
    __syncthreads();
    const int nrow = row + gridDim.x;
    if (nrow < a.nrows) { const int n2 = nrow / a.Ho, oy2 = nrow - n2 * a.Ho; STEM_LOAD_X(n2, oy2) STEM_LOAD_D(n2, oy2) }
    for (int i = 0; i < npair; ++i) {
      const float avv = Al[2 * i];
      bsum += avv;
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        const float bv = X[kbase[t] + 2 * i];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(avv, bv, acc[t], 0, 0, 0);
      }
    }
  }
  // slab: ws[blk][co][k]  (co within this 64-channel block row; FP columns)
  const int FP = gridDim.y * 64;
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int k = (ng * 5 + t) * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c2 = cob * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      a.ws[((size_t)blockIdx.x * FP + c2) * 320 + k] = acc[t][r];
    }
  }
  bsum += __shfl_xor(bsum, 32, 64);
  if (ng == 0 && half == 0) a.wsb[(size_t)blockIdx.x * FP + cob * 64 + m * 32 + l31] = bsum;
}

// dW[f][k] = sum_b ws[b][f][k], db[f] = sum_b wsb[b][f]  (fixed order; 4 phases per output)
__global__ void __launch_bounds__(256)
k_stem_mfma_reduce(const float* __restrict__ ws, const float* __restrict__ wsb, int nblk, int F, int FP,
                   float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float part[256];
  const int f = blockIdx.y;
  const int kq = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + kq;
  float s = 0.f;
  if (k < 320)
    for (int b = ph; b < nblk; b += 4) s += ws[((size_t)b * FP + f) * 320 + k];
  part[threadIdx.x] = s;
  __syncthreads();
  if (ph == 0 && k < KK) dW[(size_t)f * KK + k] = ((part[kq] + part[64 + kq]) + part[128 + kq]) + part[192 + kq];
  __syncthreads();
  if (blockIdx.x == 0) {
    float t = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 256) t += wsb[(size_t)b * FP + f];
    part[threadIdx.x] = t;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) db[f] = part[0];
  }
}

}  // namespace

namespace fdet {

bool stem_mfma_ok(int Cin, int F, int H, int W, int k, int stride, int pad) {
  if (!(Cin == 3 && k == 10 && stride == 8 && pad == 2)) return false;
  const int Wo = (W + 2 * pad - k) / stride + 1;
  return W % 4 == 0 && W <= 512 && Wo % 4 == 0 && Wo <= 64 && F >= 1;
}

size_t stem_mfma_ws_floats(int N, int F, int H, int W) {
  const int Ho = (H + 4 - 10) / 8 + 1;
  const int FP = (F + 63) / 64 * 64;
  const int nrows = N * Ho;
  const int nblk = nrows < 256 ? nrows : 256;
  return (size_t)nblk * FP * 320 + (size_t)nblk * FP;
}

int stem_mfma_fwd(const float* x, const float* w, const float* bias, float* y, int N, int F, int H, int W,
                  hipStream_t st) {
  StemArgs a{};
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.N = N; a.F = F; a.H = H; a.W = W;
  a.Ho = (H + 4 - 10) / 8 + 1; a.Wo = (W + 4 - 10) / 8 + 1; a.nrows = N * a.Ho;
  const int FP = (F + 63) / 64 * 64;
  const int nblk = a.nrows < 256 ? a.nrows : 256;
  const size_t lds = (size_t)XT * 4;
  { if (int rc_ = set_lds_attr((const void*)k_stem_fwd_mfma, (size_t)(lds), __func__)) return rc_; }
  hipLaunchKernelGGL(k_stem_fwd_mfma, dim3(nblk, FP / 64), dim3(256), lds, st, a);
  return check_launch("fdet_stem_fwd(mfma)");
}

int stem_mfma_wgrad(const float* x, const float* dy, float* dW, float* db, float* ws, int N, int F, int H, int W,
                    hipStream_t st) {
  StemArgs a{};
  a.x = x; a.dy = dy; a.N = N; a.F = F; a.H = H; a.W = W;
  a.Ho = (H + 4 - 10) / 8 + 1; a.Wo = (W + 4 - 10) / 8 + 1; a.nrows = N * a.Ho;
  const int FP = (F + 63) / 64 * 64;
  const int nblk = a.nrows < 256 ? a.nrows : 256;
  a.ws = ws; a.wsb = ws + (size_t)nblk * FP * 320;
  const size_t lds = (size_t)(XT + 64 * DYS) * 4;
  { if (int rc_ = set_lds_attr((const void*)k_stem_wgrad_mfma, (size_t)(lds), __func__)) return rc_; }
  hipLaunchKernelGGL(k_stem_wgrad_mfma, dim3(nblk, FP / 64), dim3(256), lds, st, a);
  if (int rc = check_launch("fdet_stem_wgrad(mfma)")) return rc;
  hipLaunchKernelGGL(k_stem_mfma_reduce, dim3(5, F), dim3(256), 0, st, a.ws, a.wsb, nblk, F, FP, dW, db);
  return check_launch("fdet_stem_wgrad(mfma reduce)");
}

}  // namespace fdet
