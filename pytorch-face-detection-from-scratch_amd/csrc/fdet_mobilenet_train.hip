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
// (round 4b: one plane (n, c) per blockIdx.x, 256 positions of it per blockIdx.y; k and the stride are template parameters, the
//  row of a position comes from a multiply-high with ceil(2^32 / width) -- exact for positions < 2^20 and widths < 2^12 --
//  so no kernel of this family divides per element any more)
__device__ __forceinline__ int mbt_div(int v, unsigned magic) { return magic ? (int)__umulhi((unsigned)v, magic) : v; }

template <int K, int S>
__global__ void __launch_bounds__(256)
k_mbt_dw_fwd(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ z, int C, int H, int W, int Ho, int Wo,
             int pt, int pl, unsigned magic_wo) {
  const int nc = blockIdx.x, r = blockIdx.y * 256 + threadIdx.x;
  if (r >= Ho * Wo) return;
  const int oy = mbt_div(r, magic_wo), ox = r - oy * Wo;
  const int c = nc % C;
  const float* xp = x + (size_t)nc * H * W;
  const float* wp = w + (size_t)c * K * K;
  float acc = 0.f;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int iy = oy * S - pt + ky;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int ix = ox * S - pl + kx;
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
      acc = fmaf(ok ? xp[(size_t)iy * W + ix] : 0.f, wp[ky * K + kx], acc);
    }
  }
  z[(size_t)nc * Ho * Wo + r] = acc;
}

// dx[n][c][iy][ix] = sum_{ky,kx : (iy + pt - ky) % s == 0 ...} dz[n][c][(iy + pt - ky)/s][(ix + pl - kx)/s] * w[c][ky][kx]
template <int K, int S>
__global__ void __launch_bounds__(256)
k_mbt_dw_bwd_data(const float* __restrict__ dz, const float* __restrict__ w, float* __restrict__ dx, int C, int H, int W, int Ho,
                  int Wo, int pt, int pl, unsigned magic_w) {
  const int nc = blockIdx.x, r = blockIdx.y * 256 + threadIdx.x;
  if (r >= H * W) return;
  const int iy = mbt_div(r, magic_w), ix = r - iy * W;
  const int c = nc % C;
  const float* zp = dz + (size_t)nc * Ho * Wo;
  const float* wp = w + (size_t)c * K * K;
  float acc = 0.f;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int a = iy + pt - ky;
    const int oy = S == 1 ? a : a >> 1;
    const bool oky = a >= 0 && (S == 1 || !(a & 1)) && oy < Ho;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int b = ix + pl - kx;
      const int ox = S == 1 ? b : b >> 1;
      const bool ok = oky && b >= 0 && (S == 1 || !(b & 1)) && ox < Wo;
      acc = fmaf(ok ? zp[(size_t)oy * Wo + ox] : 0.f, wp[ky * K + kx], acc);
    }
  }
  dx[(size_t)nc * H * W + r] = acc;
}

// dW[c][ky][kx] = sum_{n,oy,ox} dz * x.  The (image, 2048-position chunk) pairs of a channel are dealt to the TAP_SPLIT slices in
// turn; a thread keeps the k*k taps of its positions in registers; one LDS pass combines the waves.
template <int K, int S>
__global__ void __launch_bounds__(256)
k_mbt_dw_bwd_weight(const float* __restrict__ x, const float* __restrict__ dz, double* __restrict__ part, int N, int C, int H, int W,
                    int Ho, int Wo, int pt, int pl, unsigned magic_wo) {
  constexpr int kk = K * K, CH = 2048;
  __shared__ float sh[4][kk];
  const int c = blockIdx.x, j = blockIdx.y;
  const int per = Ho * Wo, cpp = (per + CH - 1) / CH, nchunk = N * cpp;
  float acc[kk];
#pragma unroll
  for (int t = 0; t < kk; ++t) acc[t] = 0.f;
  for (int q = j; q < nchunk; q += TAP_SPLIT) {
    const int n = q / cpp, r0 = (q - n * cpp) * CH, r1 = min(per, r0 + CH);
    const float* zp = dz + ((size_t)n * C + c) * per;
    const float* xp = x + ((size_t)n * C + c) * H * W;
    for (int r = r0 + threadIdx.x; r < r1; r += 256) {
      const int oy = mbt_div(r, magic_wo), ox = r - oy * Wo;
      const float g = zp[r];
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * S - pt + ky;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int ix = ox * S - pl + kx;
          const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
          acc[ky * K + kx] = fmaf(g, ok ? xp[(size_t)iy * W + ix] : 0.f, acc[ky * K + kx]);
        }
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < kk; ++t) {
    float v = acc[t];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) sh[wv][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < kk) {
    const int t = threadIdx.x;
    part[((size_t)c * TAP_SPLIT + j) * kk + t] = ((double)sh[0][t] + (double)sh[1][t]) + ((double)sh[2][t] + (double)sh[3][t]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// BatchNorm2d, training mode (batch statistics) + activation (+ residual), forward and backward.  z: [N][C][P]
// ---------------------------------------------------------------------------------------------------------------------
constexpr int BN_SPLIT = 32;

// partial sums of one channel: the (image, 2048-position chunk) pairs are dealt to the BN_SPLIT slices in turn (no per-element
// division); ws[(c * BN_SPLIT + j) * 2 + {0,1}] = sum, sum of squares
constexpr int BN_CH = 2048;
__global__ void __launch_bounds__(256)
k_mbt_bn_partial(const float* __restrict__ z, double* __restrict__ ws, int N, int C, int P) {
  __shared__ double sh[16];
  const int c = blockIdx.x, j = blockIdx.y;
  const int cpp = (P + BN_CH - 1) / BN_CH, nchunk = N * cpp;
  double s = 0.0, q = 0.0;
  for (int k = j; k < nchunk; k += BN_SPLIT) {
    const int n = k / cpp, p0 = (k - n * cpp) * BN_CH, p1 = min(P, p0 + BN_CH);
    const float* zp = z + ((size_t)n * C + c) * P;
    float s4 = 0.f, q4 = 0.f;                              // <= 8 elements per thread and chunk in fp32, chunks added in fp64
    for (int p = p0 + threadIdx.x; p < p1; p += 256) { const float v = zp[p]; s4 += v; q4 = fmaf(v, v, q4); }
    s += (double)s4; q += (double)q4;
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

// y = act(gamma * (z - mean) * invstd + beta) (+ residual); blockIdx.x = plane (n, c), blockIdx.y = 256-position piece
__global__ void __launch_bounds__(256)
k_mbt_bn_apply(const float* __restrict__ z, const float* __restrict__ gamma, const float* __restrict__ beta,
               const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ residual,
               float* __restrict__ y, int C, int P, int act) {
  const int p = blockIdx.y * 256 + threadIdx.x;
  if (p >= P) return;
  const int c = blockIdx.x % C;
  const size_t t = (size_t)blockIdx.x * P + p;
  const float u = gamma[c] * ((z[t] - mean[c]) * invstd[c]) + beta[c];
  float v = mbt_act(u, act);
  if (residual) v += residual[t];
  y[t] = v;
}

// backward, pass 1: g = dy * act'(u); per channel S1 = sum g, S2 = sum g * xhat (partials per slice, chunks as k_mbt_bn_partial)
__global__ void __launch_bounds__(256)
k_mbt_bn_bwd_partial(const float* __restrict__ z, const float* __restrict__ dy, const float* __restrict__ gamma,
                     const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ invstd,
                     double* __restrict__ ws, int N, int C, int P, int act) {
  __shared__ double sh[16];
  const int c = blockIdx.x, j = blockIdx.y;
  const int cpp = (P + BN_CH - 1) / BN_CH, nchunk = N * cpp;
  const float ga = gamma[c], be = beta[c], mu = mean[c], is = invstd[c];
  double s1 = 0.0, s2 = 0.0;
  for (int k = j; k < nchunk; k += BN_SPLIT) {
    const int n = k / cpp, p0 = (k - n * cpp) * BN_CH, p1 = min(P, p0 + BN_CH);
    const size_t base = ((size_t)n * C + c) * P;
    float a1 = 0.f, a2 = 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
      const float xh = (z[base + p] - mu) * is;
      const float g = dy[base + p] * mbt_dact(ga * xh + be, act);
      a1 += g; a2 = fmaf(g, xh, a2);
    }
    s1 += (double)a1; s2 += (double)a2;
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

// pass 2: dz = gamma * invstd * (g - S1 / M - xhat * S2 / M); blockIdx.x = plane (n, c)
__global__ void __launch_bounds__(256)
k_mbt_bn_bwd_apply(const float* __restrict__ z, const float* __restrict__ dy, const float* __restrict__ gamma,
                   const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ invstd,
                   const float* __restrict__ s12, float* __restrict__ dz, int C, int P, float inv_m, int act) {
  const int p = blockIdx.y * 256 + threadIdx.x;
  if (p >= P) return;
  const int c = blockIdx.x % C;
  const size_t t = (size_t)blockIdx.x * P + p;
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
k_mbt_se_scale(const float* __restrict__ x, const float* __restrict__ pre, float* __restrict__ y, int P) {
  const int p = blockIdx.y * 256 + threadIdx.x;            // blockIdx.x = plane (n, c)
  if (p >= P) return;
  const size_t t = (size_t)blockIdx.x * P + p;
  y[t] = x[t] * hsig(pre[blockIdx.x]);
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

// backward 3: dx = dy * hardsigmoid(pre) + dpool; blockIdx.x = plane (n, c)
__global__ void __launch_bounds__(256)
k_mbt_se_bwd_apply(const float* __restrict__ dy, const float* __restrict__ pre, const float* __restrict__ dpool, float* __restrict__ dx,
                   int P) {
  const int p = blockIdx.y * 256 + threadIdx.x;
  if (p >= P) return;
  const size_t t = (size_t)blockIdx.x * P + p;
  dx[t] = dy[t] * hsig(pre[blockIdx.x]) + dpool[blockIdx.x];
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
inline unsigned mbt_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }   // see mbt_div

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
  FDET_REQUIRE((long long)H * W < (1 << 20) && W < 4096, "mbt_dw_fwd: plane too large");
  const dim3 grid((unsigned)(N * C), (unsigned)((Ho * Wo + 255) / 256));
  const unsigned mg = mbt_magic(Wo);
  hipStream_t st = (hipStream_t)stream;
#define MBT_DW_FWD(K_, S_) hipLaunchKernelGGL((k_mbt_dw_fwd<K_, S_>), grid, dim3(256), 0, st, x, w, z, C, H, W, Ho, Wo, pt, pl, mg)
  if (k == 3 && s == 1) MBT_DW_FWD(3, 1); else if (k == 3) MBT_DW_FWD(3, 2); else if (s == 1) MBT_DW_FWD(5, 1); else MBT_DW_FWD(5, 2);
#undef MBT_DW_FWD
  return check_launch("fdet_mbt_dw_fwd");
}

// ws: fdet_mbt_taps_ws_bytes(C, k) bytes
extern "C" int fdet_mbt_dw_bwd(const float* x, const float* dz, const float* w, float* dx, float* dW, void* ws, size_t ws_bytes, int N,
                               int C, int H, int W, int k, int s, void* stream) {
  int Ho, Wo, pt, pl;
  FDET_REQUIRE(x && dz && w && dx && dW && ws && N > 0 && C > 0 && dw_geo(H, W, k, s, Ho, Wo, pt, pl), "mbt_dw_bwd: bad arguments");
  FDET_REQUIRE(ws_bytes >= fdet_mbt_taps_ws_bytes(C, k), "mbt_dw_bwd: workspace too small");
  FDET_REQUIRE((long long)H * W < (1 << 20) && W < 4096, "mbt_dw_bwd: plane too large");
  hipStream_t st = (hipStream_t)stream;
  double* part = reinterpret_cast<double*>(ws);
  const dim3 gd((unsigned)(N * C), (unsigned)((H * W + 255) / 256)), gw((unsigned)C, TAP_SPLIT);
  const unsigned mw = mbt_magic(W), mo = mbt_magic(Wo);
#define MBT_DW_BWD(K_, S_)                                                                                                          \
  {                                                                                                                                  \
    hipLaunchKernelGGL((k_mbt_dw_bwd_data<K_, S_>), gd, dim3(256), 0, st, dz, w, dx, C, H, W, Ho, Wo, pt, pl, mw);                    \
    hipLaunchKernelGGL((k_mbt_dw_bwd_weight<K_, S_>), gw, dim3(256), 0, st, x, dz, part, N, C, H, W, Ho, Wo, pt, pl, mo);             \
  }
  if (k == 3 && s == 1) MBT_DW_BWD(3, 1) else if (k == 3) MBT_DW_BWD(3, 2) else if (s == 1) MBT_DW_BWD(5, 1) else MBT_DW_BWD(5, 2)
#undef MBT_DW_BWD
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
  hipLaunchKernelGGL(k_mbt_bn_partial, dim3(C, BN_SPLIT), dim3(256), 0, st, z, wsd, N, C, P);
  hipLaunchKernelGGL(k_mbt_bn_finish, dim3((C + 63) / 64), dim3(64), 0, st, wsd, save_mean, save_invstd, running_mean, running_var, C,
                     (long long)N * P, momentum, eps);
  hipLaunchKernelGGL(k_mbt_bn_apply, dim3((unsigned)(N * C), (unsigned)((P + 255) / 256)), dim3(256), 0, st, z, gamma, beta, save_mean,
                     save_invstd, residual, y, C, P, act);
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
  hipLaunchKernelGGL(k_mbt_bn_bwd_partial, dim3(C, BN_SPLIT), dim3(256), 0, st, z, dy, gamma, beta, save_mean, save_invstd, wsd, N, C, P, act);
  hipLaunchKernelGGL(k_mbt_bn_bwd_finish, dim3((C + 63) / 64), dim3(64), 0, st, wsd, dgamma, dbeta, s12, C);
  hipLaunchKernelGGL(k_mbt_bn_bwd_apply, dim3((unsigned)(N * C), (unsigned)((P + 255) / 256)), dim3(256), 0, st, z, dy, gamma, beta,
                     save_mean, save_invstd, s12, dz, C, P, 1.f / (float)((long long)N * P), act);
  return check_launch("fdet_mbt_bn_bwd");
}

// SqueezeExcite forward: pooled [N,C], hidden [N,R] (pre-ReLU), pre [N,C] (pre-hardsigmoid) are kept for backward
extern "C" int fdet_mbt_se_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* pooled,
                               float* hidden, float* pre, float* y, int N, int C, int R, int P, void* stream) {
  FDET_REQUIRE(x && w1 && b1 && w2 && b2 && pooled && hidden && pre && y && N > 0 && C > 0 && R > 0 && P > 0, "mbt_se_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_mbt_se_pool, dim3((N * C + 3) / 4), dim3(256), 0, st, x, pooled, N * C, P);
  hipLaunchKernelGGL(k_mbt_se_fc, dim3(N), dim3(256), (size_t)(C + R) * 4, st, pooled, w1, b1, w2, b2, hidden, pre, C, R);
  hipLaunchKernelGGL(k_mbt_se_scale, dim3((unsigned)(N * C), (unsigned)((P + 255) / 256)), dim3(256), 0, st, x, pre, y, P);
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
  hipLaunchKernelGGL(k_mbt_se_bwd_apply, dim3((unsigned)(N * C), (unsigned)((P + 255) / 256)), dim3(256), 0, st, dy, pre, dpool, dx, P);
  const int tot = 2 * C * R + C + R;
  hipLaunchKernelGGL(k_mbt_se_wreduce, dim3((tot + 255) / 256), dim3(256), 0, st, dpre, dhid, pooled, hidden, dw1, db1, dw2, db2, N, C, R);
  return check_launch("fdet_mbt_se_bwd");
}
