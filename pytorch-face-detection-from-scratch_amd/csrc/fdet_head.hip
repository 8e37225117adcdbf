// Detection head: Dropout2d(0.5) -> Conv2d(F,5,k,pad) -> Sigmoid, forward and backward.
// models/PoolResnet.py:83-90,100-102 (k6 p0), models/Resnet.py:77-84,94-96 (k3 p1).
// 1.15 MMAC per image (0.2 % of the step): plain VALU, one workgroup per image, the image's
// activation staged in LDS in 32-channel chunks.
#include "fdet_common.h"

using namespace fdet;

namespace {

constexpr int FC = 32;   // channels per LDS chunk

// y[n,o,sy,sx] = sigmoid(b[o] + sum_{f,ky,kx} w[o,f,ky,kx] * x[n,f,sy+ky-p,sx+kx-p] * scale[n,f])
__global__ void __launch_bounds__(256)
k_head_fwd(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ w,
           const float* __restrict__ bias, float* __restrict__ y, int F, int H, int W, int k, int pad, int So,
           int Wo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* X = reinterpret_cast<float*>(smem);          // [FC][H*W]
  const int n = blockIdx.x, HW = H * W, nout = 5 * So * Wo;
  constexpr int MAXO = 8;                              // outputs per thread (5*S*S <= 2048)
  float acc[MAXO];
#pragma unroll
  for (int j = 0; j < MAXO; ++j) acc[j] = 0.f;
  for (int f0 = 0; f0 < F; f0 += FC) {
    __syncthreads();
    for (int t = threadIdx.x; t < FC * HW; t += 256) {
      const int f = f0 + t / HW;
      float v = 0.f;
      if (f < F) { v = x[((size_t)n * F + f) * HW + (t % HW)]; if (scale) v *= scale[(size_t)n * F + f]; }
      X[t] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MAXO; ++j) {
      const int t = threadIdx.x + j * 256;
      if (t >= nout) break;
      const int o = t / (So * Wo), pos = t - o * (So * Wo);
      const int sy = pos / Wo, sx = pos - sy * Wo;
      float s = acc[j];
      for (int fl = 0; fl < FC && f0 + fl < F; ++fl) {
        const float* wr = w + (((size_t)o * F + f0 + fl) * k) * k;
        const float* xr = X + fl * HW;
        for (int ky = 0; ky < k; ++ky) {
          const int iy = sy + ky - pad;
          if (iy < 0 || iy >= H) continue;
          for (int kx = 0; kx < k; ++kx) {
            const int ix = sx + kx - pad;
            if (ix < 0 || ix >= W) continue;
            s = fmaf(wr[ky * k + kx], xr[iy * W + ix], s);
          }
        }
      }
      acc[j] = s;
    }
  }
#pragma unroll
  for (int j = 0; j < MAXO; ++j) {
    const int t = threadIdx.x + j * 256;
    if (t >= nout) break;
    const int o = t / (So * Wo);
    const float z = acc[j] + bias[o];
    y[(size_t)n * nout + t] = 1.f / (1.f + expf(-z));
  }
}

// per image: dzh = dy*y*(1-y);  dx[n,f,iy,ix] = scale * sum_{o,ky,kx} dzh[o,iy-ky+p,ix-kx+p] w[o,f,ky,kx]
// partial dW[n][o,f,ky,kx] = sum_pos dzh[o,pos] * x[n,f,pos+k-p]*scale ; partial db[n][o] = sum dzh
__global__ void __launch_bounds__(256)
k_head_bwd(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ w,
           const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
           float* __restrict__ wsW, float* __restrict__ wsb, int F, int H, int W, int k, int pad, int So, int Wo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = blockIdx.x, HW = H * W, P = So * Wo, nout = 5 * P;
  float* D = reinterpret_cast<float*>(smem);           // [5][P]  dzh
  float* X = D + ((nout + 3) & ~3);                    // [FC][HW]
  for (int t = threadIdx.x; t < nout; t += 256) {
    const float yv = y[(size_t)n * nout + t];
    D[t] = dy[(size_t)n * nout + t] * (yv * (1.f - yv));
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += D[threadIdx.x * P + p];
    wsb[(size_t)n * 8 + threadIdx.x] = s;
  }
  const int kk = k * k;
  for (int f0 = 0; f0 < F; f0 += FC) {
    __syncthreads();
    for (int t = threadIdx.x; t < FC * HW; t += 256) {
      const int f = f0 + t / HW;
      float v = 0.f;
      if (f < F) { v = x[((size_t)n * F + f) * HW + (t % HW)]; if (scale) v *= scale[(size_t)n * F + f]; }
      X[t] = v;
    }
    __syncthreads();
    // dx for this channel chunk
    for (int t = threadIdx.x; t < FC * HW; t += 256) {
      const int fl = t / HW, f = f0 + fl;
      if (f >= F) break;
      const int pos = t - fl * HW, iy = pos / W, ix = pos - iy * W;
      float s = 0.f;
      for (int o = 0; o < 5; ++o) {
        const float* wr = w + (((size_t)o * F + f) * k) * k;
        for (int ky = 0; ky < k; ++ky) {
          const int sy = iy - ky + pad;
          if (sy < 0 || sy >= So) continue;
          for (int kx = 0; kx < k; ++kx) {
            const int sx = ix - kx + pad;
            if (sx < 0 || sx >= Wo) continue;
            s = fmaf(D[o * P + sy * Wo + sx], wr[ky * k + kx], s);
          }
        }
      }
      dx[((size_t)n * F + f) * HW + pos] = scale ? s * scale[(size_t)n * F + f] : s;
    }
    // partial dW for this channel chunk: outputs (o, fl, ky, kx)
    for (int t = threadIdx.x; t < 5 * FC * kk; t += 256) {
      const int o = t / (FC * kk), r = t - o * (FC * kk);
      const int fl = r / kk, tap = r - fl * kk, ky = tap / k, kx = tap - ky * k;
      const int f = f0 + fl;
      if (f >= F) continue;
      float s = 0.f;
      for (int sy = 0; sy < So; ++sy) {
        const int iy = sy + ky - pad;
        if (iy < 0 || iy >= H) continue;
        for (int sx = 0; sx < Wo; ++sx) {
          const int ix = sx + kx - pad;
          if (ix < 0 || ix >= W) continue;
          s = fmaf(D[o * P + sy * Wo + sx], X[fl * HW + iy * W + ix], s);
        }
      }
      wsW[(size_t)n * 5 * F * kk + ((size_t)o * F + f) * kk + tap] = s;
    }
  }
}

__global__ void __launch_bounds__(256)
k_head_reduce(const float* __restrict__ wsW, const float* __restrict__ wsb, int N, int nW, float* __restrict__ dW,
              float* __restrict__ db) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < nW) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += wsW[(size_t)n * nW + t];
    dW[t] = s;
  } else if (t < nW + 5) {
    const int o = t - nW;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += wsb[(size_t)n * 8 + o];
    db[o] = s;
  }
}

}  // namespace

static size_t head_lds_fwd(int H, int W) { return (size_t)FC * H * W * 4; }
static size_t head_lds_bwd(int H, int W, int So, int Wo) { return ((size_t)((5 * So * Wo + 3) & ~3) + (size_t)FC * H * W) * 4; }

extern "C" int fdet_head_fwd(const float* x, const float* drop_scale, const float* w, const float* bias, float* y,
                             int N, int F, int H, int W, int k, int pad, void* stream) {
  FDET_REQUIRE(x && w && bias && y && N > 0 && F > 0 && H > 0 && W > 0 && k > 0 && pad >= 0, "head_fwd: bad arguments");
  const int So = H + 2 * pad - k + 1, Wo = W + 2 * pad - k + 1;
  FDET_REQUIRE(So > 0 && Wo > 0 && 5 * So * Wo <= 2048, "head_fwd: unsupported output size %dx%d", So, Wo);
  const size_t lds = head_lds_fwd(H, W);
  FDET_REQUIRE(lds <= 160 * 1024, "head_fwd: activation %dx%d too large for LDS staging", H, W);
  if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k_head_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_head_fwd, dim3(N), dim3(256), lds, (hipStream_t)stream, x, drop_scale, w, bias, y, F, H, W, k,
                     pad, So, Wo);
  return check_launch("fdet_head_fwd");
}

extern "C" size_t fdet_head_bwd_ws_bytes(int N, int F, int H, int W, int k, int pad) {
  (void)H; (void)W; (void)pad;
  return ((size_t)N * 5 * F * k * k + (size_t)N * 8) * 4;
}

extern "C" int fdet_head_bwd(const float* x, const float* drop_scale, const float* w, const float* y,
                             const float* dy, float* dx, float* dW, float* db, void* ws, size_t ws_bytes, int N,
                             int F, int H, int W, int k, int pad, void* stream) {
  FDET_REQUIRE(x && w && y && dy && dx && dW && db && ws && N > 0 && F > 0, "head_bwd: bad arguments");
  const int So = H + 2 * pad - k + 1, Wo = W + 2 * pad - k + 1;
  FDET_REQUIRE(So > 0 && Wo > 0 && 5 * So * Wo <= 2048, "head_bwd: unsupported output size %dx%d", So, Wo);
  const size_t need = fdet_head_bwd_ws_bytes(N, F, H, W, k, pad);
  if (ws_bytes < need) return fail(FDET_EWORKSPACE, "head_bwd: workspace %zu < %zu bytes", ws_bytes, need);
  const size_t lds = head_lds_bwd(H, W, So, Wo);
  FDET_REQUIRE(lds <= 160 * 1024, "head_bwd: activation %dx%d too large for LDS staging", H, W);
  if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k_head_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  float* wsW = (float*)ws;
  float* wsb = wsW + (size_t)N * 5 * F * k * k;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_head_bwd, dim3(N), dim3(256), lds, st, x, drop_scale, w, y, dy, dx, wsW, wsb, F, H, W, k, pad,
                     So, Wo);
  if (int rc = check_launch("fdet_head_bwd")) return rc;
  const int nW = 5 * F * k * k;
  hipLaunchKernelGGL(k_head_reduce, dim3((nW + 5 + 255) / 256), dim3(256), 0, st, wsW, wsb, N, nW, dW, db);
  return check_launch("fdet_head_bwd(reduce)");
}
