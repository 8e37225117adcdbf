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
#include <cstdint>

using namespace fdet;

namespace fdet {   // fdet_stem_mfma.hip
bool stem_mfma_ok(int Cin, int F, int H, int W, int k, int stride, int pad);
size_t stem_mfma_ws_floats(int N, int F, int H, int W);
int stem_mfma_fwd(const float* x, const float* w, const float* bias, float* y, int N, int F, int H, int W, hipStream_t st);
int stem_mfma_wgrad(const float* x, const float* dy, float* dW, float* db, float* ws, int N, int F, int H, int W, hipStream_t st);
int stem_x3_fwd(const float* x, const float* w, const float* bias, float* y, int N, int F, int H, int W, hipStream_t st);   // fdet_stem_x3.hip
int stem_x3_fwd_ps(const void* x, const float* w, const float* bias, void* y_ps, int N, int F, int H, int W, hipStream_t st, bool p16, bool u8 = false);
int stem_dma_fwd_ps(const float* x, const float* w, const float* bias, void* y_ps, int N, int F, int H, int W, hipStream_t st, bool p16);
int stem_x3_wgrad(const float* x, const float* dy, float* dW, float* db, float* ws, int N, int F, int H, int W, hipStream_t st, bool p16);
// fdet_stem_k3.hip: the Resnet stem (k3 s2 p1) on the matrix cores / with a PS (column-strip) output
bool stem3_wgrad_ok(int Cin, int F, int H, int W, int k, int stride, int pad);
size_t stem3_wgrad_ws_floats(int N, int F, int H, int W);
int stem3_wgrad(const float* x, const float* dz, float* dW, float* db, float* ws, size_t ws_floats, int N, int F, int H, int W, hipStream_t st);
bool stem3_fwd_ps_ok(int Cin, int F, int H, int W, int k, int stride, int pad);
int stem3_fwd_ps(const float* x, const float* w, const float* bias, void* y_ps, int N, int F, int H, int W, hipStream_t st, bool p16);
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

// Resnet stem (k3 s2 p1) weight gradient, VALU but scalar-fed: a lane owns one output channel and ALL 27 taps, the
// eight waves of a workgroup split an output row into chunks of 8 columns.  The x values a chunk needs (3 rows x 17
// floats per input channel) are the same for every lane: they are read with wave-uniform addresses (scalar loads
// from the constant cache, no LDS traffic, no staging), so the inner loop is 216 FMAs with a scalar operand per 8
// LDS reads of dy.  dy rows are staged channel-major in LDS with an odd pitch (lane = channel reads without bank
// conflicts).  The generic kernel above spends its time on LDS broadcasts of x and on integer divisions in its
// staging loops (3.9 ms of the 16.6 ms Resnet-640 step); this one takes 1.0 ms.  H, W even; Wo % 4 == 0.
__global__ void __launch_bounds__(512)
k_stem_wgrad_k3(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ ws /*[nblk][27][FP]*/,
                float* __restrict__ wsb /*[nblk][FP]*/, int N, int F, int FP, int H, int W, int Ho, int Wo, int nseg) {
  // work item = (output row, segment of 64 columns): the eight waves take one 8-column chunk each; the small dy tile
  // (16.6 KB) lets four workgroups share a CU, whose 32 waves hide the scalar-load latency
  constexpr int SEG = 64, DS = SEG + 1;
  __shared__ float D[64 * DS];                          // [64][DS] dy tile; later [8][7][64] wave partials, four rounds
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fb = blockIdx.y;
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
  float bsum = 0.f;
  const int nitems = N * Ho * nseg;
  const int ipw = (nitems + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it0 = blockIdx.x * ipw, it1 = min(it0 + ipw, nitems);
  const bool w16 = (W % 16 == 0) && ((reinterpret_cast<unsigned long long>(x) & 63) == 0);
  for (int item = it0; item < it1; ++item) {
    const int row = item / nseg, seg = item - row * nseg;
    const int n = row / Ho, oy = row - n * Ho;
    const int oxs = seg * SEG;
    __syncthreads();                                    // previous item consumed
    for (int t = tid; t < 64 * (SEG / 4); t += 512) {
      const int fl = t >> 4, j = t & 15;
      const int ff = fb * 64 + fl, ox = oxs + 4 * j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ff < F && ox < Wo) v = *reinterpret_cast<const float4*>(dy + (((size_t)n * F + ff) * Ho + oy) * Wo + ox);   // Wo % 4 == 0
      float* d = D + fl * DS + 4 * j;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const int ox0 = oxs + wid * 8;
    if (ox0 < Wo) {
      const float* Dl = D + lane * DS + wid * 8;
      float dv[8];
#pragma unroll
      for (int o = 0; o < 8; ++o) { dv[o] = Dl[o]; bsum += dv[o]; }            // zeros beyond Wo
      const bool full = ox0 + 8 <= Wo;                  // wave-uniform
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int iy = 2 * oy + ky - 1;
          if (iy < 0) continue;                         // top padding row (iy <= H-1 always: H even)
          const float* xr = x + (((size_t)n * 3 + ci) * H + iy) * W;          // wave-uniform row base
          float xv[17];                                  // ix = 2*ox0 - 1 + j
          if (full) {
            const int b0 = 2 * ox0;                      // multiple of 16 floats: with W % 16 == 0 a 64-byte aligned run
            xv[0] = ox0 > 0 ? xr[b0 - 1] : 0.f;
            if (w16) {
              const float* xa = static_cast<const float*>(__builtin_assume_aligned(xr + b0, 64));
#pragma unroll
              for (int j = 0; j < 16; ++j) xv[1 + j] = xa[j];                 // one s_load_dwordx16
            } else {
#pragma unroll
              for (int j = 1; j < 17; ++j) xv[j] = xr[b0 - 1 + j];
            }
          } else {
#pragma unroll
            for (int j = 0; j < 17; ++j) {
              const int ix = 2 * ox0 - 1 + j;
              xv[j] = (ix >= 0 && ix < W) ? xr[ix] : 0.f;
            }
          }
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int o = 0; o < 8; ++o) acc[ci * 9 + ky * 3 + kx] = fmaf(dv[o], xv[2 * o + kx], acc[ci * 9 + ky * 3 + kx]);
        }
      }
    }
  }
  // combine the eight waves through LDS (the dy tile is free), seven values at a time; one slab per workgroup
  float* R = D;                                         // [8][7][64]
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 7; ++k) R[(wid * 7 + k) * 64 + lane] = kb * 7 + k < 27 ? acc[kb * 7 + k < 27 ? kb * 7 + k : 0] : bsum;
    __syncthreads();
    if (tid < 7 * 64) {
      const int k = tid >> 6, l = tid & 63;
      float s_ = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s_ += R[(w * 7 + k) * 64 + l];
      const int kk = kb * 7 + k;
      if (kk < 27) ws[((size_t)blockIdx.x * 27 + kk) * FP + fb * 64 + l] = s_;
      else wsb[(size_t)blockIdx.x * FP + fb * 64 + l] = s_;
    }
  }
}

// dW[f][k] = sum_b ws[b][k][f] ; db[f] = sum_b wsb[b][f] in a fixed order: 256 threads = 16 consecutive outputs x 16 slab
// lanes; lane g sums the slabs g, g+16, ... (four independent partial sums, so an output's loads are in flight together
// instead of one dependent chain of `nblk` loads per thread), then the 16 lanes are combined in lane order
__global__ void __launch_bounds__(256)
k_stem_reduce(const float* __restrict__ ws, const float* __restrict__ wsb, int nblk, int KK, int F, int FP,
              float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float part[16][17];
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int t = blockIdx.x * 16 + o;
  const bool isw = t < KK * FP, isb = !isw && t < KK * FP + FP;
  const int f = isw ? t % FP : t - KK * FP;
  float s = 0.f;
  if ((isw || isb) && f < F) {
    const float* __restrict__ p = isw ? ws + t : wsb + f;
    const size_t st = isw ? (size_t)KK * FP : (size_t)FP;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = g;
    for (; b + 48 < nblk; b += 64) {
      s0 += p[(size_t)b * st]; s1 += p[(size_t)(b + 16) * st]; s2 += p[(size_t)(b + 32) * st]; s3 += p[(size_t)(b + 48) * st];
    }
    for (; b < nblk; b += 16) s0 += p[(size_t)b * st];
    s = (s0 + s1) + (s2 + s3);
  }
  part[g][o] = s;
  __syncthreads();
  if (g == 0 && (isw || isb) && f < F) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) r += part[k][o];
    if (isw) dW[(size_t)f * KK + t / FP] = r;
    else db[f] = r;
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
  const int maxblk = k == 3 ? 1024 : 512;               // slabs in the workspace (k3: four workgroups per CU)
  p.nblk = nrows < maxblk ? nrows : maxblk;
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
  if (stem3_wgrad_ok(Cin, F, H, W, k, stride, pad)) fl = std::max(fl, stem3_wgrad_ws_floats(N, F, H, W));
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
  if (stem_mfma_ok(Cin, F, H, W, k, stride, pad) && !FDET_ENV_ONCE("FDET_STEM_VALU")) return stem_mfma_fwd(x, w, bias, y, N, F, H, W, st);
  if (ws_bytes < p.pack_floats * 4) return fail(FDET_EWORKSPACE, "stem_fwd: workspace %zu < %zu", ws_bytes, p.pack_floats * 4);
  float* wpk = (float*)ws;
  hipLaunchKernelGGL(k_stem_pack, dim3((p.KK * p.FP + 255) / 256), dim3(256), 0, st, w, F, p.FP, p.KK, wpk);
  dim3 grid(N * p.Ho);
  if (k == 10) {
    if (p.lds_fwd > 64 * 1024) { if (int rc_ = set_lds_attr((const void*)k_stem_fwd<10, 8, 2, 3>, (size_t)p.lds_fwd, __func__)) return rc_; }
    hipLaunchKernelGGL((k_stem_fwd<10, 8, 2, 3>), grid, dim3(256), p.lds_fwd, st, x, wpk, bias, y, F, p.FP, H, W, p.Ho, p.Wo, p.BXS);
  } else {
    if (p.lds_fwd > 64 * 1024) { if (int rc_ = set_lds_attr((const void*)k_stem_fwd<3, 2, 1, 3>, (size_t)p.lds_fwd, __func__)) return rc_; }
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

// the PoolResnet stem with a pre-split (PS) output: y_ps = image-0 pointer of a PS tensor (N, 64, Ho, Wo)
extern "C" int fdet_stem_fwd_ps(const float* x, const float* w, const float* bias, void* y_ps, int N, int Cin, int F, int H,
                                int W, int k, int stride, int pad, void* stream) {
  FDET_REQUIRE(x && w && bias && y_ps && N > 0, "stem_fwd_ps: bad arguments");
  if (stem3_fwd_ps_ok(Cin, F, H, W, k, stride, pad)) return stem3_fwd_ps(x, w, bias, y_ps, N, F, H, W, (hipStream_t)stream, false);
  FDET_REQUIRE(F == 64, "stem_fwd_ps: F must be 64");
  FDET_REQUIRE(stem_mfma_ok(Cin, F, H, W, k, stride, pad),
               "stem_fwd_ps: only the PoolResnet stem (3ch k10 s8 p2, W%%4==0, W<=512) is built; got Cin=%d k=%d s=%d p=%d W=%d",
               Cin, k, stride, pad, W);
  { const char* e = FDET_ENV_ONCE("FDET_STEM_DMA");        // the LDS-DMA form (fdet_stem_dma.hip)
    if (e && e[0] == '1') { const int rc = stem_dma_fwd_ps(x, w, bias, y_ps, N, F, H, W, (hipStream_t)stream, false); if (rc != 1) return rc; } }
  return stem_x3_fwd_ps(x, w, bias, y_ps, N, F, H, W, (hipStream_t)stream, false);
}

// precision16: one MFMA pass on bf16(x) x bf16(w), hi plane of the PS output only (see fdet_conv3x3_ps_fwd_p16)
extern "C" int fdet_stem_fwd_ps_p16(const float* x, const float* w, const float* bias, void* y_ps, int N, int Cin, int F, int H,
                                    int W, int k, int stride, int pad, void* stream) {
  FDET_REQUIRE(x && w && bias && y_ps && N > 0, "stem_fwd_ps_p16: bad arguments");
  if (stem3_fwd_ps_ok(Cin, F, H, W, k, stride, pad)) return stem3_fwd_ps(x, w, bias, y_ps, N, F, H, W, (hipStream_t)stream, true);
  FDET_REQUIRE(F == 64, "stem_fwd_ps_p16: F must be 64");
  FDET_REQUIRE(stem_mfma_ok(Cin, F, H, W, k, stride, pad),
               "stem_fwd_ps_p16: only the PoolResnet stem (3ch k10 s8 p2, W%%4==0, W<=512) is built; got Cin=%d k=%d s=%d p=%d W=%d",
               Cin, k, stride, pad, W);
  { const char* e = FDET_ENV_ONCE("FDET_STEM_DMA");
    if (e && e[0] == '1') { const int rc = stem_dma_fwd_ps(x, w, bias, y_ps, N, F, H, W, (hipStream_t)stream, true); if (rc != 1) return rc; } }
  return stem_x3_fwd_ps(x, w, bias, y_ps, N, F, H, W, (hipStream_t)stream, true);
}

// Inference: the same stem on the uint8 FRAMES themselves -- `x / 255.0` of models/PoolResnet.py:95 (BaseModel.py:65) fused into
// the staging of the stem (results identical to fdet_u8_to_f32_norm followed by fdet_stem_fwd_ps; the fp32 image is never
// written).  frames: [N][3][H][W] uint8, W % 4 == 0.  precision16: one MFMA pass, hi plane only.
extern "C" int fdet_stem_fwd_ps_u8(const unsigned char* frames, const float* w, const float* bias, void* y_ps, int N, int Cin, int F,
                                   int H, int W, int k, int stride, int pad, int precision16, void* stream) {
  FDET_REQUIRE(frames && w && bias && y_ps && N > 0 && F == 64, "stem_fwd_ps_u8: bad arguments (F must be 64)");
  FDET_REQUIRE(stem_mfma_ok(Cin, F, H, W, k, stride, pad) && ((uintptr_t)frames % 4) == 0,
               "stem_fwd_ps_u8: only the PoolResnet stem (3ch k10 s8 p2, W%%4==0, W<=512) on 4-byte aligned frames; got Cin=%d k=%d s=%d p=%d W=%d",
               Cin, k, stride, pad, W);
  return stem_x3_fwd_ps(frames, w, bias, y_ps, N, F, H, W, (hipStream_t)stream, precision16 != 0, true);
}

extern "C" int fdet_stem_wgrad_bf16x3(const float* x, const float* dy, float* dW, float* db, void* ws, size_t ws_bytes,
                                      int N, int Cin, int F, int H, int W, int k, int stride, int pad, void* stream) {
  FDET_REQUIRE(x && dy && dW && db && ws && N > 0 && F > 0, "stem_wgrad_bf16x3: bad arguments");
  if (stem3_wgrad_ok(Cin, F, H, W, k, stride, pad))       // the Resnet stem (k3 s2 p1), fdet_stem_k3.hip
    return stem3_wgrad(x, dy, dW, db, (float*)ws, ws_bytes / 4, N, F, H, W, (hipStream_t)stream);
  FDET_REQUIRE(stem_mfma_ok(Cin, F, H, W, k, stride, pad) && W % 16 == 0,
               "stem_wgrad_bf16x3: only the PoolResnet stem (3ch k10 s8 p2, W%%16==0, W<=512) is built; got Cin=%d k=%d s=%d p=%d W=%d",
               Cin, k, stride, pad, W);
  if (ws_bytes < stem_mfma_ws_floats(N, F, H, W) * 4) return fail(FDET_EWORKSPACE, "stem_wgrad_bf16x3: workspace too small");
  return stem_x3_wgrad(x, dy, dW, db, (float*)ws, N, F, H, W, (hipStream_t)stream, false);
}

// precision16: one MFMA pass on bf16(dy) x bf16(x), fp32 accumulation; same workspace as fdet_stem_wgrad_bf16x3
extern "C" int fdet_stem_wgrad_bf16(const float* x, const float* dy, float* dW, float* db, void* ws, size_t ws_bytes,
                                    int N, int Cin, int F, int H, int W, int k, int stride, int pad, void* stream) {
  FDET_REQUIRE(x && dy && dW && db && ws && N > 0 && F > 0, "stem_wgrad_bf16: bad arguments");
  FDET_REQUIRE(stem_mfma_ok(Cin, F, H, W, k, stride, pad) && W % 16 == 0,
               "stem_wgrad_bf16: only the PoolResnet stem (3ch k10 s8 p2, W%%16==0, W<=512) is built; got Cin=%d k=%d s=%d p=%d W=%d",
               Cin, k, stride, pad, W);
  if (ws_bytes < stem_mfma_ws_floats(N, F, H, W) * 4) return fail(FDET_EWORKSPACE, "stem_wgrad_bf16: workspace too small");
  return stem_x3_wgrad(x, dy, dW, db, (float*)ws, N, F, H, W, (hipStream_t)stream, true);
}

extern "C" int fdet_stem_wgrad(const float* x, const float* dy, float* dW, float* db, void* ws, size_t ws_bytes,
                               int N, int Cin, int F, int H, int W, int k, int stride, int pad, void* stream) {
  FDET_REQUIRE(x && dy && dW && db && ws && N > 0 && F > 0, "stem_wgrad: bad arguments");
  StemPlan p;
  FDET_REQUIRE(stem_plan(N, Cin, F, H, W, k, stride, pad, p), "stem_wgrad: unsupported stem k=%d stride=%d pad=%d", k, stride, pad);
  hipStream_t st = (hipStream_t)stream;
  if (stem_mfma_ok(Cin, F, H, W, k, stride, pad) && !FDET_ENV_ONCE("FDET_STEM_VALU")) {
    if (ws_bytes < stem_mfma_ws_floats(N, F, H, W) * 4) return fail(FDET_EWORKSPACE, "stem_wgrad: workspace too small");
    return stem_mfma_wgrad(x, dy, dW, db, (float*)ws, N, F, H, W, st);
  }
  if (ws_bytes < p.ws_floats * 4) return fail(FDET_EWORKSPACE, "stem_wgrad: workspace %zu < %zu", ws_bytes, p.ws_floats * 4);
  float* wsW = (float*)ws + p.pack_floats;
  float* wsb = wsW + (size_t)p.nblk * p.KK * p.FP;
  dim3 grid(p.nblk, p.FP / 64);
  if (k == 10) {
    if (p.lds_wg > 64 * 1024) { if (int rc_ = set_lds_attr((const void*)k_stem_wgrad<10, 8, 2, 3>, (size_t)p.lds_wg, __func__)) return rc_; }
    hipLaunchKernelGGL((k_stem_wgrad<10, 8, 2, 3>), grid, dim3(256), p.lds_wg, st, x, dy, wsW, wsb, N, F, p.FP, H, W, p.Ho, p.Wo, p.XS, p.DS);
  } else if (H % 2 == 0 && W % 2 == 0 && p.Wo % 4 == 0 && !FDET_ENV_ONCE("FDET_STEM_K3_GENERIC")) {
    // scalar-fed kernel: items = (row, 64-column segment), four 8-wave workgroups per CU, one slab per workgroup
    const int nseg = (p.Wo + 63) / 64;
    const int nitems = N * p.Ho * nseg;
    int ncu = 256;
    { int dev = 0, v = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v; }
    const int nb = std::min(std::min(nitems, 4 * ncu), p.nblk);            // the workspace holds p.nblk slabs
    hipLaunchKernelGGL(k_stem_wgrad_k3, dim3(nb, p.FP / 64), dim3(512), 0, st, x, dy, wsW, wsb, N, F, p.FP, H, W, p.Ho, p.Wo, nseg);
    if (int rc = check_launch("fdet_stem_wgrad(k3)")) return rc;
    hipLaunchKernelGGL(k_stem_reduce, dim3((p.KK * p.FP + p.FP + 15) / 16), dim3(256), 0, st, wsW, wsb, nb, p.KK, F, p.FP, dW, db);
    return check_launch("fdet_stem_wgrad(reduce)");
  } else {
    if (p.lds_wg > 64 * 1024) { if (int rc_ = set_lds_attr((const void*)k_stem_wgrad<3, 2, 1, 3>, (size_t)p.lds_wg, __func__)) return rc_; }
    hipLaunchKernelGGL((k_stem_wgrad<3, 2, 1, 3>), grid, dim3(256), p.lds_wg, st, x, dy, wsW, wsb, N, F, p.FP, H, W, p.Ho, p.Wo, p.XS, p.DS);
  }
  if (int rc = check_launch("fdet_stem_wgrad")) return rc;
  hipLaunchKernelGGL(k_stem_reduce, dim3((p.KK * p.FP + p.FP + 15) / 16), dim3(256), 0, st, wsW, wsb, p.nblk, p.KK, F,
                     p.FP, dW, db);
  return check_launch("fdet_stem_wgrad(reduce)");
}
