// Detection head: Dropout2d(0.5) -> Conv2d(F,5,k,pad) -> Sigmoid, forward and backward.
// models/PoolResnet.py:83-90,100-102 (k6 p0), models/Resnet.py:77-84,94-96 (k3 p1).
// 1.15 MMAC per image (0.2 % of the step's FLOPs): fp32 VALU, one workgroup per image.
// The image's activation (x dropout scale) sits zero-padded in LDS so no tap needs a bounds
// check; lanes run over positions, weights are wave-uniform (scalar loads).
#include "fdet_common.h"

using namespace fdet;

namespace {

struct HeadGeo { int F, H, W, k, pad, So, Wo, HP, WP, FC; };

__device__ __forceinline__ float sigmoidf(float z) { return 1.f / (1.f + expf(-z)); }

// stage channels [f0, f0+FC) of image n, scaled, into the zero-padded tile X[fl][HP*WP]
__device__ __forceinline__ void stage_x(float* X, const float* __restrict__ x, const float* __restrict__ scale,
                                        int n, int f0, const HeadGeo& g) {
  const int HW = g.H * g.W, PW = g.HP * g.WP;
  for (int t = threadIdx.x; t < g.FC * HW; t += blockDim.x) {
    const int fl = t / HW, p = t - fl * HW;
    const int f = f0 + fl;
    float v = 0.f;
    if (f < g.F) { v = x[((size_t)n * g.F + f) * HW + p]; if (scale) v *= scale[(size_t)n * g.F + f]; }
    const int iy = p / g.W, ix = p - iy * g.W;
    X[fl * PW + (iy + g.pad) * g.WP + ix + g.pad] = v;
  }
}

// y[n,o,sy,sx] = sigmoid(b[o] + sum_{f,ky,kx} w[o,f,ky,kx] * xs[n,f,sy+ky-p,sx+kx-p])
// 16 waves: position group (64 positions) x channel split; partials combined through LDS.
// K is a template parameter so the tap loops unroll (independent LDS reads and scalar weight
// loads in flight instead of one dependent chain per tap).
constexpr int HT = 1024;     // threads per workgroup (16 waves)
template <int K>
__global__ void __launch_bounds__(HT)
k_head_fwd(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ w,
           const float* __restrict__ bias, float* __restrict__ y, HeadGeo g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int PW = g.HP * g.WP, P = g.So * g.Wo;
  float* X = reinterpret_cast<float*>(smem);            // [FC][PW]
  float* part = X + g.FC * PW;                          // [16][5][64] partial sums per wave
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int npg = (P + 63) / 64;                        // position groups
  const int pgr = min(npg, 16);                         // position groups per round
  const int fsplit = 16 / pgr;                          // channel splits (waves beyond pgr*fsplit idle)
  for (int t = threadIdx.x; t < g.FC * PW; t += HT) X[t] = 0.f;
  constexpr int kk = K * K;
  for (int pg0 = 0; pg0 < npg; pg0 += pgr) {
    const int pg = pg0 + wid % pgr, fs = wid / pgr;
    const int pos = pg * 64 + lane;
    const bool active = pg < npg && fs < fsplit;
    const int pc = min(pos, P - 1);
    const int sy = pc / g.Wo, sx = pc - sy * g.Wo;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int f0 = 0; f0 < g.F; f0 += g.FC) {
      __syncthreads();
      stage_x(X, x, scale, n, f0, g);
      __syncthreads();
      if (active) {
        const int fcn = min(g.FC, g.F - f0);
        const int fper = (fcn + fsplit - 1) / fsplit;
        const int fa = fs * fper, fb = min(fcn, fa + fper);
        for (int fl = fa; fl < fb; ++fl) {
          const float* xr = X + fl * PW + sy * g.WP + sx;
          const float* wr = w + (size_t)(f0 + fl) * kk;                // + o*F*kk, wave-uniform
#pragma unroll
          for (int ky = 0; ky < K; ++ky) {
            float xv[K];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) xv[kx] = xr[ky * g.WP + kx];
#pragma unroll
            for (int o = 0; o < 5; ++o)
#pragma unroll
              for (int kx = 0; kx < K; ++kx) acc[o] = fmaf(xv[kx], wr[(size_t)o * g.F * kk + ky * K + kx], acc[o]);
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < 5; ++o) part[(wid * 5 + o) * 64 + lane] = active ? acc[o] : 0.f;
    __syncthreads();
    // combine the channel splits of this round's position groups
    const int npos = min(P - pg0 * 64, pgr * 64);
    for (int t = threadIdx.x; t < 5 * npos; t += HT) {
      const int o = t / npos, pl = t - o * npos;
      const int pgl = pl >> 6, ln = pl & 63;
      float z = bias[o];
      for (int s2 = 0; s2 < fsplit; ++s2) z += part[((s2 * pgr + pgl) * 5 + o) * 64 + ln];
      y[((size_t)n * 5 + o) * P + pg0 * 64 + pl] = sigmoidf(z);
    }
  }
}

// Backward, one workgroup per image.
//   dzh = dy * y * (1-y)                          (sigmoid')
//   dx[f,iy,ix] = scale[f] * sum_{o,ky,kx} dzh[o, iy-ky+p, ix-kx+p] * w[o,f,ky,kx]
//   dW partial [o,f,ky,kx] = sum_pos dzh[o,pos] * xs[f, pos+k-p] ;  db partial [o] = sum dzh
template <int K>
__global__ void __launch_bounds__(HT)
k_head_bwd(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ wT /*[5][kk][F]*/,
           const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
           float* __restrict__ wsW, float* __restrict__ wsb, HeadGeo g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int PW = g.HP * g.WP, P = g.So * g.Wo, HW = g.H * g.W;
  constexpr int kk = K * K, km1 = K - 1;
  const int DH = g.So + 2 * km1, DW = g.Wo + 2 * km1;   // dzh tile padded by k-1 on every side
  float* X = reinterpret_cast<float*>(smem);            // [FC][PW] padded, scaled activation
  float* D = X + ((g.FC * PW + 3) & ~3);                // [5][DH*DW] zero-padded dzh
  float* Dc = D + ((5 * DH * DW + 3) & ~3);             // [P][8] position-major copy for the dW pass (16 B aligned)
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int t = threadIdx.x; t < ((g.FC * PW + 3) & ~3) + ((5 * DH * DW + 3) & ~3) + P * 8; t += HT) X[t] = 0.f;
  __syncthreads();
  for (int t = threadIdx.x; t < 5 * P; t += HT) {
    const int o = t / P, p = t - o * P;
    const float yv = y[(size_t)n * 5 * P + t];
    const float d = dy[(size_t)n * 5 * P + t] * (yv * (1.f - yv));
    const int sy = p / g.Wo, sx = p - sy * g.Wo;
    D[o * DH * DW + (sy + km1) * DW + sx + km1] = d;
    Dc[p * 8 + o] = d;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += Dc[p * 8 + threadIdx.x];
    wsb[(size_t)n * 8 + threadIdx.x] = s;
  }
  for (int f0 = 0; f0 < g.F; f0 += g.FC) {
    const int fcn = min(g.FC, g.F - f0);
    __syncthreads();
    stage_x(X, x, scale, n, f0, g);
    __syncthreads();
    // ---- dx: waves = (position group) x (16-channel group); weights are wave-uniform scalars
    {
      const int npg = (HW + 63) / 64, ncg = (fcn + 15) / 16;
      for (int job = wid; job < npg * ncg; job += HT / 64) {
        const int pgp = job % npg, cgp = job / npg;
        const int p = pgp * 64 + lane;
        const int pc = min(p, HW - 1);
        const int iy = pc / g.W, ix = pc - iy * g.W;
        const float* dr = D + (iy + g.pad + km1) * DW + ix + g.pad + km1;
        const int fl0 = cgp * 16;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        for (int o = 0; o < 5; ++o) {
#pragma unroll
          for (int ky = 0; ky < K; ++ky) {
            float dv[K];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) dv[kx] = dr[o * DH * DW - ky * DW - kx];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
              const float* wr = wT + ((size_t)o * kk + ky * K + kx) * g.F + f0 + fl0;   // wave-uniform
#pragma unroll
              for (int j = 0; j < 16; ++j) acc[j] = fmaf(dv[kx], wr[j], acc[j]);
            }
          }
        }
        if (p < HW) {
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int f = f0 + fl0 + j;
            if (fl0 + j < fcn) dx[((size_t)n * g.F + f) * HW + p] = scale ? acc[j] * scale[(size_t)n * g.F + f] : acc[j];
          }
        }
      }
    }
    // ---- partial dW: lanes over channels (odd channel stride: conflict-free), (channel group, tap) jobs over waves
    {
      const int ncg = (fcn + 63) / 64;
      for (int job = wid; job < ncg * kk; job += HT / 64) {
        const int tap = job % kk, fl = (job / kk) * 64 + lane;
        const int ky = tap / K, kx = tap - ky * K;
        const float* xr = X + min(fl, g.FC - 1) * PW;
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int sy = 0; sy < g.So; ++sy) {
          const float* xrow = xr + (sy + ky) * g.WP + kx;
          const float* drow = Dc + sy * g.Wo * 8;
#pragma unroll 4
          for (int sx = 0; sx < g.Wo; ++sx) {
            const float xv = xrow[sx];
            const float4 d4 = *reinterpret_cast<const float4*>(drow + sx * 8);     // broadcast
            const float d5 = drow[sx * 8 + 4];
            acc[0] = fmaf(d4.x, xv, acc[0]); acc[1] = fmaf(d4.y, xv, acc[1]); acc[2] = fmaf(d4.z, xv, acc[2]);
            acc[3] = fmaf(d4.w, xv, acc[3]); acc[4] = fmaf(d5, xv, acc[4]);
          }
        }
        if (fl < fcn) {
#pragma unroll
          for (int o = 0; o < 5; ++o)
            wsW[(size_t)n * 5 * g.F * kk + ((size_t)o * g.F + f0 + fl) * kk + tap] = acc[o];
        }
      }
    }
  }
}

__global__ void __launch_bounds__(1024)
k_head_reduce(const float* __restrict__ wsW, const float* __restrict__ wsb, int N, int nW, float* __restrict__ dW,
              float* __restrict__ db) {
  // 64 outputs x 16 image phases per workgroup (the per-image slabs are 46 KB apart: the sum is a chain of
  // dependent-latency loads, so more phases = fewer round trips), fixed-order combine
  __shared__ float part[1024];
  const int q = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + q;
  float s = 0.f;
  if (t < nW) {
    for (int n = ph; n < N; n += 16) s += wsW[(size_t)n * nW + t];
  } else if (t < nW + 5) {
    for (int n = ph; n < N; n += 16) s += wsb[(size_t)n * 8 + (t - nW)];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if (ph == 0) {
    float tot = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) tot += part[p * 64 + q];
    if (t < nW) dW[t] = tot;
    else if (t < nW + 5) db[t - nW] = tot;
  }
}

__global__ void __launch_bounds__(256)
k_head_pack(const float* __restrict__ w, int F, int kk, float* __restrict__ wT) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < 5 * kk * F) {
    const int o = t / (kk * F), r = t - o * (kk * F), tap = r / F, f = r - tap * F;
    wT[t] = w[((size_t)o * F + f) * kk + tap];
  }
}

bool head_geo(int F, int H, int W, int k, int pad, HeadGeo& g, size_t extra_floats_per_fc0) {
  g.F = F; g.H = H; g.W = W; g.k = k; g.pad = pad;
  g.So = H + 2 * pad - k + 1; g.Wo = W + 2 * pad - k + 1;
  g.HP = H + 2 * pad; g.WP = W + 2 * pad;
  if (((g.HP * g.WP) & 1) == 0) g.WP += 1;               // odd channel stride: conflict-free lane-per-channel reads
  if (g.So <= 0 || g.Wo <= 0) return false;
  const size_t per = (size_t)g.HP * g.WP * 4;
  const size_t budget = 150 * 1024 - extra_floats_per_fc0 * 4;
  int fc = (int)(budget / per);
  if (fc >= F) fc = F; else fc = fc / 16 * 16;
  g.FC = fc;
  return fc >= 16 || fc == F;
}

}  // namespace

extern "C" int fdet_head_fwd(const float* x, const float* drop_scale, const float* w, const float* bias, float* y,
                             int N, int F, int H, int W, int k, int pad, void* stream) {
  FDET_REQUIRE(x && w && bias && y && N > 0 && F > 0 && H > 0 && W > 0 && k > 0 && pad >= 0, "head_fwd: bad arguments");
  HeadGeo g;
  const int So = H + 2 * pad - k + 1, Wo = W + 2 * pad - k + 1;
  FDET_REQUIRE(So > 0 && Wo > 0 && So * Wo <= 4096, "head_fwd: unsupported output size %dx%d", So, Wo);
  FDET_REQUIRE(k == 6 || k == 3, "head_fwd: kernel size %d not built (6 and 3 are)", k);
  const size_t part = (size_t)16 * 5 * 64;
  FDET_REQUIRE(head_geo(F, H, W, k, pad, g, part), "head_fwd: activation %dx%d too large for LDS staging", H, W);
  const size_t lds = ((size_t)g.FC * g.HP * g.WP + part) * 4;
  if (k == 6) {
    if (lds > 64 * 1024) { if (int rc_ = set_lds_attr((const void*)k_head_fwd<6>, (size_t)(lds), __func__)) return rc_; }
    hipLaunchKernelGGL(k_head_fwd<6>, dim3(N), dim3(HT), lds, (hipStream_t)stream, x, drop_scale, w, bias, y, g);
  } else {
    if (lds > 64 * 1024) { if (int rc_ = set_lds_attr((const void*)k_head_fwd<3>, (size_t)(lds), __func__)) return rc_; }
    hipLaunchKernelGGL(k_head_fwd<3>, dim3(N), dim3(HT), lds, (hipStream_t)stream, x, drop_scale, w, bias, y, g);
  }
  return check_launch("fdet_head_fwd");
}

extern "C" size_t fdet_head_bwd_ws_bytes(int N, int F, int H, int W, int k, int pad) {
  (void)H; (void)W; (void)pad;
  return ((size_t)N * 5 * F * k * k + (size_t)N * 8 + (size_t)5 * k * k * F + 64) * 4;   // +64: 16-wide weight reads
}

extern "C" int fdet_head_bwd(const float* x, const float* drop_scale, const float* w, const float* y,
                             const float* dy, float* dx, float* dW, float* db, void* ws, size_t ws_bytes, int N,
                             int F, int H, int W, int k, int pad, void* stream) {
  FDET_REQUIRE(x && w && y && dy && dx && dW && db && ws && N > 0 && F > 0, "head_bwd: bad arguments");
  const int So = H + 2 * pad - k + 1, Wo = W + 2 * pad - k + 1;
  FDET_REQUIRE(So > 0 && Wo > 0 && So * Wo <= 4096, "head_bwd: unsupported output size %dx%d", So, Wo);
  FDET_REQUIRE(F % 16 == 0 || F < 16, "head_bwd: F=%d must be a multiple of 16 (or < 16)", F);
  const size_t need = fdet_head_bwd_ws_bytes(N, F, H, W, k, pad);
  if (ws_bytes < need) return fail(FDET_EWORKSPACE, "head_bwd: workspace %zu < %zu bytes", ws_bytes, need);
  HeadGeo g;
  const int km1 = k - 1;
  const size_t extra = (((size_t)5 * (So + 2 * km1) * (Wo + 2 * km1) + 3) & ~(size_t)3) + (size_t)So * Wo * 8;
  FDET_REQUIRE(head_geo(F, H, W, k, pad, g, extra), "head_bwd: activation %dx%d too large for LDS staging", H, W);
  const size_t lds = ((((size_t)g.FC * g.HP * g.WP + 3) & ~(size_t)3) + extra) * 4;
  FDET_REQUIRE(k == 6 || k == 3, "head_bwd: kernel size %d not built (6 and 3 are)", k);
  const int kk = k * k;
  float* wsW = (float*)ws;
  float* wsb = wsW + (size_t)N * 5 * F * kk;
  float* wT = wsb + (size_t)N * 8;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_head_pack, dim3((5 * kk * F + 255) / 256), dim3(256), 0, st, w, F, kk, wT);
  if (k == 6) {
    if (lds > 64 * 1024) { if (int rc_ = set_lds_attr((const void*)k_head_bwd<6>, (size_t)(lds), __func__)) return rc_; }
    hipLaunchKernelGGL(k_head_bwd<6>, dim3(N), dim3(HT), lds, st, x, drop_scale, wT, y, dy, dx, wsW, wsb, g);
  } else {
    if (lds > 64 * 1024) { if (int rc_ = set_lds_attr((const void*)k_head_bwd<3>, (size_t)(lds), __func__)) return rc_; }
    hipLaunchKernelGGL(k_head_bwd<3>, dim3(N), dim3(HT), lds, st, x, drop_scale, wT, y, dy, dx, wsW, wsb, g);
  }
  if (int rc = check_launch("fdet_head_bwd")) return rc;
  const int nW = 5 * F * kk;
  hipLaunchKernelGGL(k_head_reduce, dim3((nW + 5 + 63) / 64), dim3(1024), 0, st, wsW, wsb, N, nW, dW, db);
  return check_launch("fdet_head_bwd(reduce)");
}
