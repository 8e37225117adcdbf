// Stem convolution (3 -> F, no activation), forward and weight gradient.
// models/PoolResnet.py:70-76,98 (k10 s8 p2) and models/Resnet.py:64-70 (k3 s2 p1).
// The input needs no gradient, so there is no data-gradient kernel.
//
// v1: fp32 VALU kernels with compile-time (k, stride, pad).  One workgroup per output row
// (n, oy); the k input rows it needs are staged in LDS de-interleaved by column phase
// (ix+pad = stride*bx + phase) so that the 64 lanes of a wave (consecutive ox) read
// consecutive dwords for every tap.
#include "fdet_common.h"
#include <algorithm>
#include <cstdlib>

using namespace fdet;

namespace fdet {   // fdet_stem_mfma.hip
bool stem_mfma_ok(int Cin, int F, int H, int W, int k, int stride, int pad);
size_t stem_mfma_ws_floats(int N, int F, int H, int W);
int stem_mfma_fwd(const float* x, const float* w, const float* bias, float* y, int N, int F, int H, int W, hipStream_t st);
int stem_mfma_wgrad(const float* x, const float* dy, float* dW, float* db, float* ws, int N, int F, int H, int W, hipStream_t st);
int stem_x3_fwd(const float* x, const float* w, const float* bias, float* y, int N, int F, int H, int W, hipStream_t st);   // fdet_stem_x3.hip
int stem_x3_wgrad(const float* x, const float* dy, float* dW, float* db, float* ws, int N, int F, int H, int W, hipStream_t st);
}

namespace {

template <int KS, int ST, int PD, int CIN>
struct StemGeo {
  static constexpr int KK = CIN * KS * KS;
  static constexpr int XTRA = (KS + ST - 1) / ST;     // extra bx reached by the widest tap
};

// ---- forward: lanes = ox, each wave owns groups of 16 output channels (weights are wave-uniform
// -> scalar loads from the [k][FP] packed copy)
template <int KS, int ST, int PD, int CIN>
__global__ void __launch_bounds__(256)
k_stem_fwd(const float* __restrict__ x, const float* __restrict__ wpk /*[KK][FP]*/, const float* __restrict__ bias,
           float* __restrict__ y, int F, int FP, int H, int W, int Ho, int Wo, int BXS) {
  using G = StemGeo<KS, ST, PD, CIN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* X = reinterpret_cast<float*>(smem);          // [CIN][KS][ST][BXS]
  const int n = blockIdx.x / Ho, oy = blockIdx.x - n * Ho;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // stage: zero, then scatter valid pixels
  for (int t = tid; t < CIN * KS * ST * BXS; t += 256) X[t] = 0.f;
  __syncthreads();
  for (int t = tid; t < CIN * KS * W; t += 256) {
    const int ci = t / (KS * W), r = t - ci * (KS * W);
    const int ky = r / W, ix = r - ky * W;
    const int iy = oy * ST - PD + ky;
    if (iy < 0 || iy >= H) continue;
    const int ixp = ix + PD;
    X[((ci * KS + ky) * ST + (ixp % ST)) * BXS + ixp / ST] = x[(((size_t)n * CIN + ci) * H + iy) * W + ix];
  }
  __syncthreads();
  const int FG = FP / 16;
  for (int ox0 = 0; ox0 < Wo; ox0 += 64) {
    const int ox = ox0 + lane;
    for (int fg = wid; fg < FG; fg += 4) {
      float acc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll 1
      for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll 1
        for (int ky = 0; ky < KS; ++ky) {
          const float* xr = X + ((ci * KS + ky) * ST) * BXS + ox;
          const float* wr = wpk + (size_t)((ci * KS + ky) * KS) * FP + fg * 16;
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const float xv = xr[(kx % ST) * BXS + kx / ST];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = fmaf(xv, wr[kx * FP + j], acc[j]);
          }
        }
      }
      if (ox < Wo) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int f = fg * 16 + j;
          if (f < F) y[(((size_t)n * F + f) * Ho + oy) * Wo + ox] = acc[j] + bias[f];
        }
      }
    }
  }
}

// ---- weight gradient: lanes = f (64 per pass), each wave owns KK/4 taps kept in registers;
// x values are wave-uniform LDS broadcasts, dy values come from an LDS tile [f][ox].
template <int KS, int ST, int PD, int CIN>
__global__ void __launch_bounds__(256)
k_stem_wgrad(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ ws /*[nblk][KKP][FP]*/,
             float* __restrict__ wsb /*[nblk][FP]*/, int N, int F, int FP, int H, int W, int Ho, int Wo, int XS,
             int DS) {
  using G = StemGeo<KS, ST, PD, CIN>;
  constexpr int KPW = (G::KK + 3) / 4;               // taps per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* X = reinterpret_cast<float*>(smem);          // [CIN][KS][XS]   row-major padded rows (ix+PD)
  float* D = X + CIN * KS * XS;                       // [64][DS]        dy tile, DS odd
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fb = blockIdx.y;                          // 64-channel block
  const int f = fb * 64 + lane;
  float acc[KPW];
#pragma unroll
  for (int j = 0; j < KPW; ++j) acc[j] = 0.f;
  float bsum = 0.f;
  const int nrows = N * Ho;
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int n = row / Ho, oy = row - n * Ho;
    __syncthreads();
    for (int t = tid; t < CIN * KS * XS; t += 256) {
      const int ci = t / (KS * XS), r = t - ci * (KS * XS);
      const int ky = r / XS, ixp = r - ky * XS;
      const int iy = oy * ST - PD + ky, ix = ixp - PD;
      float v = 0.f;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)n * CIN + ci) * H + iy) * W + ix];
      X[t] = v;
    }
    for (int t = tid; t < 64 * Wo; t += 256) {
      const int fl = t / Wo, ox = t - fl * Wo;
      const int ff = fb * 64 + fl;
      D[fl * DS + ox] = (ff < F) ? dy[(((size_t)n * F + ff) * Ho + oy) * Wo + ox] : 0.f;
    }
    __syncthreads();
    // tap offsets are wave-uniform and row-invariant: keep the divisions out of the ox loop
    int xoff[KPW];
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
      const int k = min(wid * KPW + j, G::KK - 1);
      const int ci = k / (KS * KS), r = k - ci * (KS * KS);
      const int ky = r / KS, kx = r - ky * KS;
      xoff[j] = (ci * KS + ky) * XS + kx;
    }
    const float* Dl = D + lane * DS;
#pragma unroll 4
    for (int ox = 0; ox < Wo; ++ox) {
      const float dv = Dl[ox];
      if (wid == 0) bsum += dv;
#pragma unroll
      for (int j = 0; j < KPW; ++j) {
        const float xv = (wid * KPW + j < G::KK) ? X[xoff[j] + ox * ST] : 0.f;
        acc[j] = fmaf(dv, xv, acc[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < KPW; ++j) {
    const int k = wid * KPW + j;
    if (k < G::KK) ws[((size_t)blockIdx.x * G::KK + k) * FP + f] = acc[j];
  }
  if (wid == 0) wsb[(size_t)blockIdx.x * FP + f] = bsum;
}

// dW[f][k] = sum_b ws[b][k][f] ; db[f] = sum_b wsb[b][f]   (fixed order)
__global__ void __launch_bounds__(256)
k_stem_reduce(const float* __restrict__ ws, const float* __restrict__ wsb, int nblk, int KK, int F, int FP,
              float* __restrict__ dW, float* __restrict__ db) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < KK * FP) {
    const int k = t / FP, f = t - k * FP;
    if (f < F) {
      float s = 0.f;
      for (int b = 0; b < nblk; ++b) s += ws[((size_t)b * KK + k) * FP + f];
      dW[(size_t)f * KK + k] = s;
    }
  } else if (t < KK * FP + FP) {
    const int f = t - KK * FP;
    if (f < F) {
      float s = 0.f;
      for (int b = 0; b < nblk; ++b) s += wsb[(size_t)b * FP + f];
      db[f] = s;
    }
  }
}

__global__ void __launch_bounds__(256)
k_stem_pack(const float* __restrict__ w, int F, int FP, int KK, float* __restrict__ wpk) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < KK * FP) {
    const int k = t / FP, f = t - k * FP;
    wpk[t] = (f < F) ? w[(size_t)f * KK + k] : 0.f;
  }
}

struct StemPlan { int Ho, Wo, FP, KK, nblk, BXS, XS, DS; size_t lds_fwd, lds_wg, ws_floats, pack_floats; };

bool stem_plan(int N, int Cin, int F, int H, int W, int k, int stride, int pad, StemPlan& p) {
  if (Cin != 3) return false;
  if (!((k == 10 && stride == 8 && pad == 2) || (k == 3 && stride == 2 && pad == 1))) return false;
  p.Ho = (H + 2 * pad - k) / stride + 1;
  p.Wo = (W + 2 * pad - k) / stride + 1;
  p.FP = (F + 63) / 64 * 64;
  p.KK = Cin * k * k;
  const int wo64 = (p.Wo + 63) / 64 * 64;
  p.BXS = wo64 + (k + stride - 1) / stride + 1;
  p.XS = W + 2 * pad + stride + 4;
  p.DS = p.Wo | 1;
  const int nrows = N * p.Ho;
  p.nblk = nrows < 512 ? nrows : 512;
  p.lds_fwd = (size_t)Cin * k * stride * p.BXS * 4;
  p.lds_wg = ((size_t)Cin * k * p.XS + 64 * (size_t)p.DS) * 4;
  p.pack_floats = (size_t)p.KK * p.FP;
  p.ws_floats = p.pack_floats + (size_t)p.nblk * p.KK * p.FP + (size_t)p.nblk * p.FP;
  return p.lds_fwd <= 160 * 1024 && p.lds_wg <= 160 * 1024;
}

}  // namespace

extern "C" size_t fdet_stem_ws_bytes(int N, int Cin, int F, int H, int W, int k, int stride, int pad) {
  StemPlan p;
  if (!stem_plan(N, Cin, F, H, W, k, stride, pad, p)) return 0;
  size_t fl = p.ws_floats;
  if (stem_mfma_ok(Cin, F, H, W, k, stride, pad)) fl = std::max(fl, stem_mfma_ws_floats(N, F, H, W));
  return fl * 4;
}

extern "C" int fdet_stem_fwd(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes,
                             int N, int Cin, int F, int H, int W, int k, int stride, int pad, void* stream) {
  FDET_REQUIRE(x && w && bias && y && ws && N > 0 && F > 0, "stem_fwd: bad arguments");
  StemPlan p;
  FDET_REQUIRE(stem_plan(N, Cin, F, H, W, k, stride, pad, p),
               "stem_fwd: unsupported stem Cin=%d k=%d stride=%d pad=%d W=%d (built: 3ch k10s8p2, k3s2p1)", Cin, k,
               stride, pad, W);
  hipStream_t st = (hipStream_t)stream;
  if (stem_mfma_ok(Cin, F, H, W, k, stride, pad) && !getenv("FDET_STEM_VALU")) return stem_mfma_fwd(x, w, bias, y, N, F, H, W, st);
  if (ws_bytes < p.pack_floats * 4) return fail(FDET_EWORKSPACE, "stem_fwd: workspace %zu < %zu", ws_bytes, p.pack_floats * 4);
  float* wpk = (float*)ws;
  hipLaunchKernelGGL(k_stem_pack, dim3((p.KK * p.FP + 255) / 256), dim3(256), 0, st, w, F, p.FP, p.KK, wpk);
  dim3 grid(N * p.Ho);
  if (k == 10) {
    if (p.lds_fwd > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_stem_fwd<10, 8, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_fwd);
    hipLaunchKernelGGL((k_stem_fwd<10, 8, 2, 3>), grid, dim3(256), p.lds_fwd, st, x, wpk, bias, y, F, p.FP, H, W, p.Ho, p.Wo, p.BXS);
  } else {
    if (p.lds_fwd > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_stem_fwd<3, 2, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_fwd);
    hipLaunchKernelGGL((k_stem_fwd<3, 2, 1, 3>), grid, dim3(256), p.lds_fwd, st, x, wpk, bias, y, F, p.FP, H, W, p.Ho, p.Wo, p.BXS);
  }
  return check_launch("fdet_stem_fwd");
}

extern "C" int fdet_stem_fwd_bf16x3(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes,
                                    int N, int Cin, int F, int H, int W, int k, int stride, int pad, void* stream) {
  (void)ws; (void)ws_bytes;
  FDET_REQUIRE(x && w && bias && y && N > 0 && F > 0, "stem_fwd_bf16x3: bad arguments");
  FDET_REQUIRE(stem_mfma_ok(Cin, F, H, W, k, stride, pad),
               "stem_fwd_bf16x3: only the PoolResnet stem (3ch k10 s8 p2, W%%4==0, W<=512) is built; got Cin=%d k=%d s=%d p=%d W=%d",
               Cin, k, stride, pad, W);
  return stem_x3_fwd(x, w, bias, y, N, F, H, W, (hipStream_t)stream);
}

extern "C" int fdet_stem_wgrad_bf16x3(const float* x, const float* dy, float* dW, float* db, void* ws, size_t ws_bytes,
                                      int N, int Cin, int F, int H, int W, int k, int stride, int pad, void* stream) {
  FDET_REQUIRE(x && dy && dW && db && ws && N > 0 && F > 0, "stem_wgrad_bf16x3: bad arguments");
  FDET_REQUIRE(stem_mfma_ok(Cin, F, H, W, k, stride, pad) && W % 16 == 0,
               "stem_wgrad_bf16x3: only the PoolResnet stem (3ch k10 s8 p2, W%%16==0, W<=512) is built; got Cin=%d k=%d s=%d p=%d W=%d",
               Cin, k, stride, pad, W);
  if (ws_bytes < stem_mfma_ws_floats(N, F, H, W) * 4) return fail(FDET_EWORKSPACE, "stem_wgrad_bf16x3: workspace too small");
  return stem_x3_wgrad(x, dy, dW, db, (float*)ws, N, F, H, W, (hipStream_t)stream);
}

extern "C" int fdet_stem_wgrad(const float* x, const float* dy, float* dW, float* db, void* ws, size_t ws_bytes,
                               int N, int Cin, int F, int H, int W, int k, int stride, int pad, void* stream) {
  FDET_REQUIRE(x && dy && dW && db && ws && N > 0 && F > 0, "stem_wgrad: bad arguments");
  StemPlan p;
  FDET_REQUIRE(stem_plan(N, Cin, F, H, W, k, stride, pad, p), "stem_wgrad: unsupported stem k=%d stride=%d pad=%d", k, stride, pad);
  hipStream_t st = (hipStream_t)stream;
  if (stem_mfma_ok(Cin, F, H, W, k, stride, pad) && !getenv("FDET_STEM_VALU")) {
    if (ws_bytes < stem_mfma_ws_floats(N, F, H, W) * 4) return fail(FDET_EWORKSPACE, "stem_wgrad: workspace too small");
    return stem_mfma_wgrad(x, dy, dW, db, (float*)ws, N, F, H, W, st);
  }
  if (ws_bytes < p.ws_floats * 4) return fail(FDET_EWORKSPACE, "stem_wgrad: workspace %zu < %zu", ws_bytes, p.ws_floats * 4);
  float* wsW = (float*)ws + p.pack_floats;
  float* wsb = wsW + (size_t)p.nblk * p.KK * p.FP;
  dim3 grid(p.nblk, p.FP / 64);
  if (k == 10) {
    if (p.lds_wg > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_stem_wgrad<10, 8, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_wg);
    hipLaunchKernelGGL((k_stem_wgrad<10, 8, 2, 3>), grid, dim3(256), p.lds_wg, st, x, dy, wsW, wsb, N, F, p.FP, H, W, p.Ho, p.Wo, p.XS, p.DS);
  } else {
    if (p.lds_wg > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_stem_wgrad<3, 2, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_wg);
    hipLaunchKernelGGL((k_stem_wgrad<3, 2, 1, 3>), grid, dim3(256), p.lds_wg, st, x, dy, wsW, wsb, N, F, p.FP, H, W, p.Ho, p.Wo, p.XS, p.DS);
  }
  if (int rc = check_launch("fdet_stem_wgrad")) return rc;
  hipLaunchKernelGGL(k_stem_reduce, dim3((p.KK * p.FP + p.FP + 255) / 256), dim3(256), 0, st, wsW, wsb, p.nblk, p.KK, F,
                     p.FP, dW, db);
  return check_launch("fdet_stem_wgrad(reduce)");
}
