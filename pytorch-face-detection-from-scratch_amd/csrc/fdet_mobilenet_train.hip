// MobileNetV3-small backbone, TRAINING pieces (round 4): what `ModelMeta.training_step` needs from
// `MobilenetV3Backbone` (models/MobilenetV3Backbone.py:49-60 = timm tf_mobilenetv3_small_100 features + Conv2d(576,5,3,p1) +
// sigmoid) beyond the inference engine of fdet_mobilenet.hip: BatchNorm with BATCH statistics (and its running-statistics
// update), depthwise convs, the Conv2dSame stem, SqueezeExcite -- each with its backward.  The 1x1 convs reuse the pointwise
// GEMM kernels (fdet_pointwise_x3.hip), the head fdet_head_fwd / fdet_head_bwd.
//
// Layout: fp32 NCHW (the reference's), one tensor in, one out per kernel; correctness first -- these are plain
// one-thread-per-element VALU kernels with fixed-order (deterministic) reductions, not roofline work: BASELINE config 5 is an
// inference run.  PARITY UNPINNED like the inference backbone (timm is absent: no reference output exists); the tests check
// every piece and the whole train step against torch autograd on the CPU oracle (oracle/mobilenet_oracle.py).
#include "fdet_common.h"

using namespace fdet;

namespace {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_HSWISH = 2 };

__device__ __forceinline__ float mbt_act(float u, int act) {
  if (act == ACT_RELU) return u > 0.f ? u : 0.f;
  if (act == ACT_HSWISH) return u * fminf(fmaxf(u + 3.f, 0.f), 6.f) / 6.f;
  return u;
}
// d act / d u (ATen: hardswish_backward = 0 below -3, u/3 + 0.5 up to 3, 1 above)
__device__ __forceinline__ float mbt_dact(float u, int act) {
  if (act == ACT_RELU) return u > 0.f ? 1.f : 0.f;
  if (act == ACT_HSWISH) return u < -3.f ? 0.f : (u <= 3.f ? u / 3.f + 0.5f : 1.f);
  return 1.f;
}

// TF "SAME" padding of timm's Conv2dSame / pad_same: total = max((ceil(i/s) - 1) * s + k - i, 0), the smaller half in front
__host__ __device__ inline int same_pad_front(int i, int k, int s) {
  const int o = (i + s - 1) / s;
  const int tot = (o - 1) * s + k - i;
  return tot > 0 ? tot / 2 : 0;
}

// block-wide sum of doubles (blockDim.x <= 1024, a multiple of 64); result valid in thread 0
__device__ double block_sum(double v, double* sh) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

// ---------------------------------------------------------------------------------------------------------------------
// stem: Conv2dSame(3, 16, 3, stride 2), no bias
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_mbt_stem_fwd(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ z, int N, int H, int W, int Ho, int Wo,
               int pt, int pl) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)N * Ho * Wo;
  if (t >= total) return;
  const int ox = (int)(t % Wo), oy = (int)((t / Wo) % Ho), n = (int)(t / ((long long)Wo * Ho));
  float acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.f;
  for (int ci = 0; ci < 3; ++ci)
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - pt + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - pl + kx;
        if (ix < 0 || ix >= W) continue;
        const float v = x[(((size_t)n * 3 + ci) * H + iy) * W + ix];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = fmaf(v, w[((c * 3 + ci) * 3 + ky) * 3 + kx], acc[c]);
      }
    }
#pragma unroll
  for (int c = 0; c < 16; ++c) z[(((size_t)n * 16 + c) * Ho + oy) * Wo + ox] = acc[c];
}

// dW[c][ci][ky][kx] = sum_{n,oy,ox} dz[n][c][oy][ox] * x[n][ci][2oy - pt + ky][2ox - pl + kx].  One block per (channel, slice of
// the (n, oy, ox) space): a thread reads dz once per position and keeps all 27 taps in registers; block partials in fp64, the
// slices are added in fixed order by k_mbt_taps_finish.
constexpr int TAP_SPLIT = 64;
__global__ void __launch_bounds__(256)
k_mbt_stem_wgrad(const float* __restrict__ x, const float* __restrict__ dz, double* __restrict__ part, int N, int H, int W, int Ho,
                 int Wo, int pt, int pl) {
  __shared__ double sh[16];
  const int c = blockIdx.x, j = blockIdx.y;
  const long long total = (long long)N * Ho * Wo;
  const long long lo = total * j / TAP_SPLIT, hi = total * (j + 1) / TAP_SPLIT;
  float acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = 0.f;
  for (long long t = lo + threadIdx.x; t < hi; t += 256) {
    const int ox = (int)(t % Wo), oy = (int)((t / Wo) % Ho), n = (int)(t / ((long long)Wo * Ho));
    const float g = dz[(((size_t)n * 16 + c) * Ho + oy) * Wo + ox];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - pt + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = ox * 2 - pl + kx;
          const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
          const float v = ok ? x[(((size_t)n * 3 + ci) * H + iy) * W + ix] : 0.f;
          acc[(ci * 3 + ky) * 3 + kx] = fmaf(g, v, acc[(ci * 3 + ky) * 3 + kx]);
        }
      }
  }
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    const double tot = block_sum((double)acc[t], sh);
    if (threadIdx.x == 0) part[((size_t)c * TAP_SPLIT + j) * 27 + t] = tot;
  }
}

// out[c * ntap + t] = sum_j part[(c * TAP_SPLIT + j) * ntap + t]   (fixed order)
__global__ void __launch_bounds__(256)
k_mbt_taps_finish(const double* __restrict__ part, float* __restrict__ out, int C, int ntap) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * ntap) return;
  const int c = i / ntap, t = i % ntap;
  double s = 0.0;
  for (int j = 0; j < TAP_SPLIT; ++j) s += part[((size_t)c * TAP_SPLIT + j) * ntap + t];
  out[i] = (float)s;
}

// ---------------------------------------------------------------------------------------------------------------------
// depthwise conv k x k, stride 1 (pad k/2) or stride 2 (TF SAME), no bias: forward, data gradient, weight gradient
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_mbt_dw_fwd(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ z, int N, int C, int H, int W, int Ho,
             int Wo, int k, int s, int pt, int pl) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)N * C * Ho * Wo;
  if (t >= total) return;
  const int ox = (int)(t % Wo), oy = (int)((t / Wo) % Ho);
  const long long nc = t / ((long long)Wo * Ho);
  const int c = (int)(nc % C);
  const float* xp = x + (size_t)nc * H * W;
  const float* wp = w + (size_t)c * k * k;
  float acc = 0.f;
  for (int ky = 0; ky < k; ++ky) {
    const int iy = oy * s - pt + ky;
    if (iy < 0 || iy >= H) continue;
    for (int kx = 0; kx < k; ++kx) {
      const int ix = ox * s - pl + kx;
      if (ix < 0 || ix >= W) continue;
      acc = fmaf(xp[(size_t)iy * W + ix], wp[ky * k + kx], acc);
    }
  }
  z[t] = acc;
}

// dx[n][c][iy][ix] = sum_{ky,kx : (iy + pt - ky) % s == 0 ...} dz[n][c][(iy + pt - ky)/s][(ix + pl - kx)/s] * w[c][ky][kx]
__global__ void __launch_bounds__(256)
k_mbt_dw_bwd_data(const float* __restrict__ dz, const float* __restrict__ w, float* __restrict__ dx, int N, int C, int H, int W,
                  int Ho, int Wo, int k, int s, int pt, int pl) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)N * C * H * W;
  if (t >= total) return;
  const int ix = (int)(t % W), iy = (int)((t / W) % H);
  const long long nc = t / ((long long)W * H);
  const int c = (int)(nc % C);
  const float* zp = dz + (size_t)nc * Ho * Wo;
  const float* wp = w + (size_t)c * k * k;
  float acc = 0.f;
  for (int ky = 0; ky < k; ++ky) {
    const int a = iy + pt - ky;
    if (a < 0 || a % s) continue;
    const int oy = a / s;
    if (oy >= Ho) continue;
    for (int kx = 0; kx < k; ++kx) {
      const int b = ix + pl - kx;
      if (b < 0 || b % s) continue;
      const int ox = b / s;
      if (ox >= Wo) continue;
      acc = fmaf(zp[(size_t)oy * Wo + ox], wp[ky * k + kx], acc);
    }
  }
  dx[t] = acc;
}

// dW[c][ky][kx] = sum_{n,oy,ox} dz * x: one block per (channel, slice); a thread keeps the k*k taps of its positions
template <int K>
__global__ void __launch_bounds__(256)
k_mbt_dw_bwd_weight(const float* __restrict__ x, const float* __restrict__ dz, double* __restrict__ part, int N, int C, int H, int W,
                    int Ho, int Wo, int s, int pt, int pl) {
  __shared__ double sh[16];
  constexpr int kk = K * K;
  const int c = blockIdx.x, j = blockIdx.y;
  const long long per = (long long)Ho * Wo, total = (long long)N * per;
  const long long lo = total * j / TAP_SPLIT, hi = total * (j + 1) / TAP_SPLIT;
  float acc[kk];
#pragma unroll
  for (int t = 0; t < kk; ++t) acc[t] = 0.f;
  for (long long t = lo + threadIdx.x; t < hi; t += 256) {
    const int n = (int)(t / per);
    const int r = (int)(t % per), oy = r / Wo, ox = r % Wo;
    const float g = dz[((size_t)n * C + c) * per + r];
    const float* xp = x + ((size_t)n * C + c) * H * W;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int iy = oy * s - pt + ky;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int ix = ox * s - pl + kx;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        acc[ky * K + kx] = fmaf(g, ok ? xp[(size_t)iy * W + ix] : 0.f, acc[ky * K + kx]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < kk; ++t) {
    const double tot = block_sum((double)acc[t], sh);
    if (threadIdx.x == 0) part[((size_t)c * TAP_SPLIT + j) * kk + t] = tot;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// BatchNorm2d, training mode (batch statistics) + activation (+ residual), forward and backward.  z: [N][C][P]
// ---------------------------------------------------------------------------------------------------------------------
constexpr int BN_SPLIT = 32;

// partial sums of one channel over a slice of the (n, p) space: ws[(c * BN_SPLIT + j) * 2 + {0,1}] = sum, sum of squares
__global__ void __launch_bounds__(1024)
k_mbt_bn_partial(const float* __restrict__ z, double* __restrict__ ws, int N, int C, int P) {
  __shared__ double sh[16];
  const int c = blockIdx.x, j = blockIdx.y;
  const long long total = (long long)N * P;
  const long long lo = total * j / BN_SPLIT, hi = total * (j + 1) / BN_SPLIT;
  double s = 0.0, q = 0.0;
  for (long long t = lo + threadIdx.x; t < hi; t += 1024) {
    const int n = (int)(t / P), p = (int)(t % P);
    const double v = (double)z[((size_t)n * C + c) * P + p];
    s += v; q += v * v;
  }
  const double ts = block_sum(s, sh);
  const double tq = block_sum(q, sh);
  if (threadIdx.x == 0) { ws[((size_t)c * BN_SPLIT + j) * 2] = ts; ws[((size_t)c * BN_SPLIT + j) * 2 + 1] = tq; }
}

// mean / invstd of the batch (biased variance, as F.batch_norm normalises) and the running statistics (unbiased variance,
// running = (1 - momentum) * running + momentum * batch), nn.BatchNorm2d semantics
__global__ void __launch_bounds__(64)
k_mbt_bn_finish(const double* __restrict__ ws, float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ run_mean,
                float* __restrict__ run_var, int C, long long M, float momentum, float eps) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int j = 0; j < BN_SPLIT; ++j) { s += ws[((size_t)c * BN_SPLIT + j) * 2]; q += ws[((size_t)c * BN_SPLIT + j) * 2 + 1]; }
  const double m = s / (double)M;
  double var = q / (double)M - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)m;
  if (run_var) {
    const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
  }
}

// y = act(gamma * (z - mean) * invstd + beta) (+ residual)
__global__ void __launch_bounds__(256)
k_mbt_bn_apply(const float* __restrict__ z, const float* __restrict__ gamma, const float* __restrict__ beta,
               const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ residual,
               float* __restrict__ y, int C, int P, long long total, int act) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int c = (int)((t / P) % C);
  const float u = gamma[c] * ((z[t] - mean[c]) * invstd[c]) + beta[c];
  float v = mbt_act(u, act);
  if (residual) v += residual[t];
  y[t] = v;
}

// backward, pass 1: g = dy * act'(u); per channel S1 = sum g, S2 = sum g * xhat (partials per slice)
__global__ void __launch_bounds__(1024)
k_mbt_bn_bwd_partial(const float* __restrict__ z, const float* __restrict__ dy, const float* __restrict__ gamma,
                     const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ invstd,
                     double* __restrict__ ws, int N, int C, int P, int act) {
  __shared__ double sh[16];
  const int c = blockIdx.x, j = blockIdx.y;
  const long long total = (long long)N * P;
  const long long lo = total * j / BN_SPLIT, hi = total * (j + 1) / BN_SPLIT;
  const float ga = gamma[c], be = beta[c], mu = mean[c], is = invstd[c];
  double s1 = 0.0, s2 = 0.0;
  for (long long t = lo + threadIdx.x; t < hi; t += 1024) {
    const int n = (int)(t / P), p = (int)(t % P);
    const size_t e = ((size_t)n * C + c) * P + p;
    const float xh = (z[e] - mu) * is;
    const float g = dy[e] * mbt_dact(ga * xh + be, act);
    s1 += (double)g; s2 += (double)g * (double)xh;
  }
  const double t1 = block_sum(s1, sh);
  const double t2 = block_sum(s2, sh);
  if (threadIdx.x == 0) { ws[((size_t)c * BN_SPLIT + j) * 2] = t1; ws[((size_t)c * BN_SPLIT + j) * 2 + 1] = t2; }
}

__global__ void __launch_bounds__(64)
k_mbt_bn_bwd_finish(const double* __restrict__ ws, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ s12,
                    int C) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int j = 0; j < BN_SPLIT; ++j) { s1 += ws[((size_t)c * BN_SPLIT + j) * 2]; s2 += ws[((size_t)c * BN_SPLIT + j) * 2 + 1]; }
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
  s12[2 * c] = (float)s1; s12[2 * c + 1] = (float)s2;
}

// pass 2: dz = gamma * invstd * (g - S1 / M - xhat * S2 / M)
__global__ void __launch_bounds__(256)
k_mbt_bn_bwd_apply(const float* __restrict__ z, const float* __restrict__ dy, const float* __restrict__ gamma,
                   const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ invstd,
                   const float* __restrict__ s12, float* __restrict__ dz, int C, int P, long long total, float inv_m, int act) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int c = (int)((t / P) % C);
  const float xh = (z[t] - mean[c]) * invstd[c];
  const float g = dy[t] * mbt_dact(gamma[c] * xh + beta[c], act);
  dz[t] = gamma[c] * invstd[c] * (g - s12[2 * c] * inv_m - xh * (s12[2 * c + 1] * inv_m));
}

// ---------------------------------------------------------------------------------------------------------------------
// SqueezeExcite (timm): y = x * hardsigmoid(W2 relu(W1 mean_hw(x) + b1) + b2); one block per image
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float hsig(float v) { return fminf(fmaxf(v + 3.f, 0.f), 6.f) / 6.f; }

// mean over H*W of every (n, c) plane: one wave per plane (fixed order)
__global__ void __launch_bounds__(256)
k_mbt_se_pool(const float* __restrict__ x, float* __restrict__ pooled, int NC, int P) {
  const int lane = threadIdx.x & 63;
  const int pl = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pl >= NC) return;
  const float* xp = x + (size_t)pl * P;
  float s = 0.f;
  for (int p = lane; p < P; p += 64) s += xp[p];
  s = wave_sum(s);
  if (lane == 0) pooled[pl] = s / (float)P;
}

// the two tiny FC layers of one image: hidden = W1 pooled + b1 (kept pre-ReLU), pre = W2 relu(hidden) + b2
__global__ void __launch_bounds__(256)
k_mbt_se_fc(const float* __restrict__ pooled, const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
            const float* __restrict__ b2, float* __restrict__ hidden, float* __restrict__ pre, int C, int R) {
  extern __shared__ float smf[];                         // pooled[C] | hidden[R]
  float* sp = smf; float* shd = smf + C;
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) sp[c] = pooled[(size_t)n * C + c];
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += 256) {
    float a = b1[r];
    for (int c = 0; c < C; ++c) a = fmaf(w1[(size_t)r * C + c], sp[c], a);
    shd[r] = a > 0.f ? a : 0.f; hidden[(size_t)n * R + r] = a;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = b2[c];
    for (int r = 0; r < R; ++r) a = fmaf(w2[(size_t)c * R + r], shd[r], a);
    pre[(size_t)n * C + c] = a;
  }
}

__global__ void __launch_bounds__(256)
k_mbt_se_scale(const float* __restrict__ x, const float* __restrict__ pre, float* __restrict__ y, int P, long long total) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  y[t] = x[t] * hsig(pre[t / P]);
}

// backward 1: dpre[n][c] = hardsigmoid'(pre) * sum_p dy * x  (one wave per plane)
__global__ void __launch_bounds__(256)
k_mbt_se_bwd_gate(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ dpre,
                  int NC, int P) {
  const int lane = threadIdx.x & 63;
  const int pl = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pl >= NC) return;
  const float* xp = x + (size_t)pl * P;
  const float* gp = dy + (size_t)pl * P;
  float s = 0.f;
  for (int p = lane; p < P; p += 64) s = fmaf(gp[p], xp[p], s);
  s = wave_sum(s);
  if (lane == 0) { const float a = pre[pl]; dpre[pl] = (a > -3.f && a < 3.f) ? s * (1.f / 6.f) : 0.f; }
}

// backward 2 (per image): dhid = relu'(hidden) * W2^T dpre ; dpool = W1^T dhid / P
__global__ void __launch_bounds__(256)
k_mbt_se_bwd_fc(const float* __restrict__ dpre, const float* __restrict__ hidden, const float* __restrict__ w1,
                const float* __restrict__ w2, float* __restrict__ dhid, float* __restrict__ dpool, int C, int R, int P) {
  extern __shared__ float smf[];                         // dpre[C] | dhid[R]
  float* sdpre = smf; float* sdh = smf + C;
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) sdpre[c] = dpre[(size_t)n * C + c];
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += 256) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a = fmaf(w2[(size_t)c * R + r], sdpre[c], a);
    const float d = hidden[(size_t)n * R + r] > 0.f ? a : 0.f;
    sdh[r] = d; dhid[(size_t)n * R + r] = d;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (int r = 0; r < R; ++r) a = fmaf(w1[(size_t)r * C + c], sdh[r], a);
    dpool[(size_t)n * C + c] = a / (float)P;
  }
}

// backward 3: dx = dy * hardsigmoid(pre) + dpool
__global__ void __launch_bounds__(256)
k_mbt_se_bwd_apply(const float* __restrict__ dy, const float* __restrict__ pre, const float* __restrict__ dpool, float* __restrict__ dx,
                   int P, long long total) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const long long pl = t / P;
  dx[t] = dy[t] * hsig(pre[pl]) + dpool[pl];
}

// dW2[c][r] = sum_n dpre[n][c] * relu(hidden[n][r]); db2[c] = sum_n dpre; dW1[r][c] = sum_n dhid[n][r] * pooled[n][c]; db1[r] = sum_n dhid
__global__ void __launch_bounds__(256)
k_mbt_se_wreduce(const float* __restrict__ dpre, const float* __restrict__ dhid, const float* __restrict__ pooled,
                 const float* __restrict__ hidden, float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                 float* __restrict__ db2, int N, int C, int R) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int nW = C * R;
  if (t < nW) {                                          // dW2[c][r]
    const int c = t / R, r = t % R;
    float s = 0.f;
    for (int n = 0; n < N; ++n) { const float h = hidden[(size_t)n * R + r]; s = fmaf(dpre[(size_t)n * C + c], h > 0.f ? h : 0.f, s); }
    dw2[t] = s;
  } else if (t < 2 * nW) {                               // dW1[r][c]
    const int u = t - nW, r = u / C, c = u % C;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s = fmaf(dhid[(size_t)n * R + r], pooled[(size_t)n * C + c], s);
    dw1[u] = s;
  } else if (t < 2 * nW + C) {
    const int c = t - 2 * nW;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dpre[(size_t)n * C + c];
    db2[c] = s;
  } else if (t < 2 * nW + C + R) {
    const int r = t - 2 * nW - C;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dhid[(size_t)n * R + r];
    db1[r] = s;
  }
}

inline unsigned nblk(long long total, int per) { return (unsigned)((total + per - 1) / per); }

}  // namespace

extern "C" int fdet_mbt_stem_fwd(const float* x, const float* w, float* z, int N, int H, int W, void* stream) {
  FDET_REQUIRE(x && w && z && N > 0 && H > 0 && W > 0, "mbt_stem_fwd: bad arguments");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long long total = (long long)N * Ho * Wo;
  FDET_REQUIRE(total < (1ll << 31), "mbt_stem_fwd: tensor too large");
  hipLaunchKernelGGL(k_mbt_stem_fwd, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, x, w, z, N, H, W, Ho, Wo,
                     same_pad_front(H, 3, 2), same_pad_front(W, 3, 2));
  return check_launch("fdet_mbt_stem_fwd");
}

extern "C" size_t fdet_mbt_taps_ws_bytes(int C, int k) { return (size_t)C * TAP_SPLIT * (k == 0 ? 27 : k * k) * sizeof(double); }

// ws: fdet_mbt_taps_ws_bytes(16, 0) bytes
extern "C" int fdet_mbt_stem_wgrad(const float* x, const float* dz, float* dW, void* ws, size_t ws_bytes, int N, int H, int W,
                                   void* stream) {
  FDET_REQUIRE(x && dz && dW && ws && N > 0, "mbt_stem_wgrad: bad arguments");
  FDET_REQUIRE(ws_bytes >= fdet_mbt_taps_ws_bytes(16, 0), "mbt_stem_wgrad: workspace too small");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  hipStream_t st = (hipStream_t)stream;
  double* part = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(k_mbt_stem_wgrad, dim3(16, TAP_SPLIT), dim3(256), 0, st, x, dz, part, N, H, W, Ho, Wo,
                     same_pad_front(H, 3, 2), same_pad_front(W, 3, 2));
  hipLaunchKernelGGL(k_mbt_taps_finish, dim3((16 * 27 + 255) / 256), dim3(256), 0, st, part, dW, 16, 27);
  return check_launch("fdet_mbt_stem_wgrad");
}

namespace {
bool dw_geo(int H, int W, int k, int s, int& Ho, int& Wo, int& pt, int& pl) {
  if ((k != 3 && k != 5) || (s != 1 && s != 2) || H < 1 || W < 1) return false;
  if (s == 1) { Ho = H; Wo = W; pt = pl = k / 2; }
  else { Ho = (H + 1) / 2; Wo = (W + 1) / 2; pt = same_pad_front(H, k, 2); pl = same_pad_front(W, k, 2); }
  return true;
}
}  // namespace

extern "C" int fdet_mbt_dw_fwd(const float* x, const float* w, float* z, int N, int C, int H, int W, int k, int s, void* stream) {
  int Ho, Wo, pt, pl;
  FDET_REQUIRE(x && w && z && N > 0 && C > 0 && dw_geo(H, W, k, s, Ho, Wo, pt, pl), "mbt_dw_fwd: bad arguments (k 3|5, stride 1|2)");
  const long long total = (long long)N * C * Ho * Wo;
  hipLaunchKernelGGL(k_mbt_dw_fwd, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, x, w, z, N, C, H, W, Ho, Wo, k, s, pt, pl);
  return check_launch("fdet_mbt_dw_fwd");
}

// ws: fdet_mbt_taps_ws_bytes(C, k) bytes
extern "C" int fdet_mbt_dw_bwd(const float* x, const float* dz, const float* w, float* dx, float* dW, void* ws, size_t ws_bytes, int N,
                               int C, int H, int W, int k, int s, void* stream) {
  int Ho, Wo, pt, pl;
  FDET_REQUIRE(x && dz && w && dx && dW && ws && N > 0 && C > 0 && dw_geo(H, W, k, s, Ho, Wo, pt, pl), "mbt_dw_bwd: bad arguments");
  FDET_REQUIRE(ws_bytes >= fdet_mbt_taps_ws_bytes(C, k), "mbt_dw_bwd: workspace too small");
  const long long total = (long long)N * C * H * W;
  hipStream_t st = (hipStream_t)stream;
  double* part = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(k_mbt_dw_bwd_data, dim3(nblk(total, 256)), dim3(256), 0, st, dz, w, dx, N, C, H, W, Ho, Wo, k, s, pt, pl);
  if (k == 3) hipLaunchKernelGGL(k_mbt_dw_bwd_weight<3>, dim3(C, TAP_SPLIT), dim3(256), 0, st, x, dz, part, N, C, H, W, Ho, Wo, s, pt, pl);
  else hipLaunchKernelGGL(k_mbt_dw_bwd_weight<5>, dim3(C, TAP_SPLIT), dim3(256), 0, st, x, dz, part, N, C, H, W, Ho, Wo, s, pt, pl);
  hipLaunchKernelGGL(k_mbt_taps_finish, dim3((C * k * k + 255) / 256), dim3(256), 0, st, part, dW, C, k * k);
  return check_launch("fdet_mbt_dw_bwd");
}

extern "C" size_t fdet_mbt_bn_ws_bytes(int C) { return (size_t)C * BN_SPLIT * 2 * sizeof(double) + (size_t)C * 2 * sizeof(float); }

// y = act(BatchNorm_train(z)) (+ residual): batch statistics over (N, P), running statistics updated in place (momentum as
// nn.BatchNorm2d: running = (1 - momentum) * running + momentum * batch, unbiased variance), save_mean / save_invstd for backward
extern "C" int fdet_mbt_bn_fwd(const float* z, const float* gamma, const float* beta, float* running_mean, float* running_var,
                               float momentum, float eps, float* save_mean, float* save_invstd, const float* residual, float* y,
                               void* ws, size_t ws_bytes, int N, int C, int P, int act, void* stream) {
  FDET_REQUIRE(z && gamma && beta && save_mean && save_invstd && y && ws && N > 0 && C > 0 && P > 0 && act >= 0 && act <= 2,
               "mbt_bn_fwd: bad arguments");
  FDET_REQUIRE(ws_bytes >= fdet_mbt_bn_ws_bytes(C), "mbt_bn_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  double* wsd = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(k_mbt_bn_partial, dim3(C, BN_SPLIT), dim3(1024), 0, st, z, wsd, N, C, P);
  hipLaunchKernelGGL(k_mbt_bn_finish, dim3((C + 63) / 64), dim3(64), 0, st, wsd, save_mean, save_invstd, running_mean, running_var, C,
                     (long long)N * P, momentum, eps);
  const long long total = (long long)N * C * P;
  hipLaunchKernelGGL(k_mbt_bn_apply, dim3(nblk(total, 256)), dim3(256), 0, st, z, gamma, beta, save_mean, save_invstd, residual, y, C, P,
                     total, act);
  return check_launch("fdet_mbt_bn_fwd");
}

// dy = gradient w.r.t. y (the residual branch's gradient is dy itself: the caller adds it where the skip came from)
extern "C" int fdet_mbt_bn_bwd(const float* z, const float* dy, const float* gamma, const float* beta, const float* save_mean,
                               const float* save_invstd, float* dz, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, int N,
                               int C, int P, int act, void* stream) {
  FDET_REQUIRE(z && dy && gamma && beta && save_mean && save_invstd && dz && dgamma && dbeta && ws && N > 0 && C > 0 && P > 0,
               "mbt_bn_bwd: bad arguments");
  FDET_REQUIRE(ws_bytes >= fdet_mbt_bn_ws_bytes(C), "mbt_bn_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  double* wsd = reinterpret_cast<double*>(ws);
  float* s12 = reinterpret_cast<float*>(wsd + (size_t)C * BN_SPLIT * 2);
  hipLaunchKernelGGL(k_mbt_bn_bwd_partial, dim3(C, BN_SPLIT), dim3(1024), 0, st, z, dy, gamma, beta, save_mean, save_invstd, wsd, N, C, P, act);
  hipLaunchKernelGGL(k_mbt_bn_bwd_finish, dim3((C + 63) / 64), dim3(64), 0, st, wsd, dgamma, dbeta, s12, C);
  const long long total = (long long)N * C * P;
  hipLaunchKernelGGL(k_mbt_bn_bwd_apply, dim3(nblk(total, 256)), dim3(256), 0, st, z, dy, gamma, beta, save_mean, save_invstd, s12, dz, C, P,
                     total, 1.f / (float)((long long)N * P), act);
  return check_launch("fdet_mbt_bn_bwd");
}

// SqueezeExcite forward: pooled [N,C], hidden [N,R] (pre-ReLU), pre [N,C] (pre-hardsigmoid) are kept for backward
extern "C" int fdet_mbt_se_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* pooled,
                               float* hidden, float* pre, float* y, int N, int C, int R, int P, void* stream) {
  FDET_REQUIRE(x && w1 && b1 && w2 && b2 && pooled && hidden && pre && y && N > 0 && C > 0 && R > 0 && P > 0, "mbt_se_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_mbt_se_pool, dim3((N * C + 3) / 4), dim3(256), 0, st, x, pooled, N * C, P);
  hipLaunchKernelGGL(k_mbt_se_fc, dim3(N), dim3(256), (size_t)(C + R) * 4, st, pooled, w1, b1, w2, b2, hidden, pre, C, R);
  const long long total = (long long)N * C * P;
  hipLaunchKernelGGL(k_mbt_se_scale, dim3(nblk(total, 256)), dim3(256), 0, st, x, pre, y, P, total);
  return check_launch("fdet_mbt_se_fwd");
}

// ws: (2*N*C + N*R) floats
extern "C" int fdet_mbt_se_bwd(const float* x, const float* dy, const float* pooled, const float* hidden, const float* pre,
                               const float* w1, const float* w2, float* dx, float* dw1, float* db1, float* dw2, float* db2, void* ws,
                               size_t ws_bytes, int N, int C, int R, int P, void* stream) {
  FDET_REQUIRE(x && dy && pooled && hidden && pre && w1 && w2 && dx && dw1 && db1 && dw2 && db2 && ws && N > 0, "mbt_se_bwd: bad arguments");
  FDET_REQUIRE(ws_bytes >= ((size_t)2 * N * C + (size_t)N * R) * 4, "mbt_se_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* dpre = reinterpret_cast<float*>(ws);
  float* dhid = dpre + (size_t)N * C;
  float* dpool = dhid + (size_t)N * R;
  hipLaunchKernelGGL(k_mbt_se_bwd_gate, dim3((N * C + 3) / 4), dim3(256), 0, st, x, dy, pre, dpre, N * C, P);
  hipLaunchKernelGGL(k_mbt_se_bwd_fc, dim3(N), dim3(256), (size_t)(C + R) * 4, st, dpre, hidden, w1, w2, dhid, dpool, C, R, P);
  { const long long total = (long long)N * C * P;
    hipLaunchKernelGGL(k_mbt_se_bwd_apply, dim3(nblk(total, 256)), dim3(256), 0, st, dy, pre, dpool, dx, P, total); }
  const int tot = 2 * C * R + C + R;
  hipLaunchKernelGGL(k_mbt_se_wreduce, dim3((tot + 255) / 256), dim3(256), 0, st, dpre, dhid, pooled, hidden, dw1, db1, dw2, db2, N, C, R);
  return check_launch("fdet_mbt_se_bwd");
}
