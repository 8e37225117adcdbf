// MobileNetV3-small backbone (BASELINE.json config 5; models/MobilenetV3Backbone.py:11-60 = timm tf_mobilenetv3_small_100
// without its classifier + Conv2d(576,5,3,p1) + sigmoid), inference, bf16.  Engine-private layout: NHWC bf16 (channels
// innermost, every channel count a multiple of 8), BatchNorm folded into the conv weights on the host.  The model is
// memory-bound (depthwise convs and thin 1x1 GEMMs: 2-30 FLOP/B), so every kernel is built around 16-byte channel-vector
// accesses, reads each activation once and writes each once:
//
//   k_mb_stem      Conv2dSame(3,16,3,s2) + BN + Hardswish, NCHW f32 / u8 (the /255 fused) -> NHWC bf16
//   k_mb_dw<K>     depthwise KxK (stride 1: pad K/2; stride 2: TF "SAME", the smaller half of the padding in front) + BN + act;
//                  optionally the SqueezeExcite global-average-pool numerators (per image and channel) as a by-product
//   k_mb_se        SqueezeExcite gate: mean -> FC reduce + ReLU -> FC expand -> Hardsigmoid, one workgroup per image
//   k_mb_pw<MT>    pointwise conv as a bf16 MFMA GEMM (v_mfma_f32_32x32x16_bf16, fp32 accumulate): A = weights [Cout][Cin]
//                  and B = activations [position][Cin] are both K-contiguous, so every fragment is one 16-byte global load
//                  (no LDS, no transposes); the SE gate is applied to the B fragment, bias + activation + residual in the
//                  epilogue
//   k_mb_head      Conv2d(576,5,3,p1) + sigmoid -> NCHW f32 maps for the decode / NMS kernels of the YOLO path
#include "fdet_conv3x3_x3.h"
#include <algorithm>
#include <cstdint>

using namespace fdet;

namespace {

typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef unsigned short u16;
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef u16 u16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(u16 v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }   // RNE, NaN-safe (v_cvt_pk_bf16_f32)
__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == 1) return v > 0.f ? v : 0.f;                                           // ReLU
  if (act == 2) return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);           // Hardswish
  return v;
}

// ------------------------------------------------------------------------------------------------ stem
template <typename TIN>
__global__ void __launch_bounds__(256)
k_mb_stem(const TIN* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, u16* __restrict__ y,
          int N, int H, int W, int Ho, int Wo, float in_scale) {
  __shared__ float ws[16 * 27 + 16];
  for (int t = threadIdx.x; t < 16 * 27; t += 256) ws[t] = w[t];
  if (threadIdx.x < 16) ws[16 * 27 + threadIdx.x] = bias[threadIdx.x];
  __syncthreads();
  const size_t total = (size_t)N * Ho * Wo;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int ox = (int)(t % Wo);
    const size_t r = t / Wo;
    const int oy = (int)(r % Ho), n = (int)(r / Ho);
    float v[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int iy = 2 * oy + ky, ix = 2 * ox + kx;            // TF SAME, even input: nothing on the top/left, one on the bottom/right
          v[(c * 3 + ky) * 3 + kx] = (iy < H && ix < W) ? (float)x[(((size_t)n * 3 + c) * H + iy) * W + ix] * in_scale : 0.f;
        }
    u16 o[16];
#pragma unroll
    for (int co = 0; co < 16; ++co) {
      float s = ws[16 * 27 + co];
#pragma unroll
      for (int k = 0; k < 27; ++k) s = fmaf(v[k], ws[co * 27 + k], s);
      o[co] = f2bf(act_apply(s, 2));
    }
    __builtin_memcpy(y + t * 16, o, 32);
  }
}

// ------------------------------------------------------------------------------------------------ depthwise
struct DwArgs {
  const u16* x; const float* w; const float* bias; u16* y; float* pool;   // w: [K*K][C] folded; pool: [N][C] sums or null
  int N, H, W, C, Ho, Wo, stride, pad, act;
};

template <int K>
__global__ void __launch_bounds__(256)
k_mb_dw(const DwArgs a) {
  // one thread = one output position x 8 channels; threads of a workgroup: CG = C/8 channel groups fastest, then positions of
  // ONE image (blockIdx.y = image), so the pooled sums of a workgroup go to one row of `pool`
  extern __shared__ __attribute__((aligned(16))) float wl[];      // [K*K][C] weights, then [C] bias
  const int C = a.C, CG = C >> 3;
  for (int t = threadIdx.x; t < K * K * C; t += 256) wl[t] = a.w[t];
  for (int t = threadIdx.x; t < C; t += 256) wl[K * K * C + t] = a.bias[t];
  __syncthreads();
  const int n = blockIdx.y;
  const int per_wg = 256 / CG;                                    // positions per workgroup pass (CG <= 72 -> >= 3)
  const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
  const int npos = a.Ho * a.Wo;
  float psum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (pl < per_wg) {
    for (int p = blockIdx.x * per_wg + pl; p < npos; p += gridDim.x * per_wg) {
      const int oy = p / a.Wo, ox = p - oy * a.Wo;
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = wl[K * K * C + cg * 8 + j];
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * a.stride - a.pad + ky;
        if (iy < 0 || iy >= a.H) continue;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int ix = ox * a.stride - a.pad + kx;
          if (ix < 0 || ix >= a.W) continue;
          const u16x8 v = *reinterpret_cast<const u16x8*>(a.x + (((size_t)n * a.H + iy) * a.W + ix) * C + cg * 8);
          const float* wt = wl + (ky * K + kx) * C + cg * 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = fmaf(bf2f(v[j]), wt[j], acc[j]);
        }
      }
      u16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float r = act_apply(acc[j], a.act);
        o[j] = f2bf(r);
        psum[j] += bf2f(o[j]);                                    // the pool sees what the next layer reads
      }
      *reinterpret_cast<u16x8*>(a.y + ((size_t)n * npos + p) * C + cg * 8) = o;
    }
  }
  if (a.pool) {
    // SqueezeExcite numerators: fixed-order sum over this workgroup's positions per channel (LDS), then ONE float atomic
    // per (workgroup, channel) -- the only order-dependent step (differences ~1 ulp of an fp32 mean that is then read
    // through a hardsigmoid and a bf16 product)
    float* part = wl + K * K * C + C;                             // [256][8]
#pragma unroll
    for (int j = 0; j < 8; ++j) part[threadIdx.x * 8 + j] = (pl < per_wg) ? psum[j] : 0.f;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      const int g = c >> 3, j = c & 7;
      float s = 0.f;
      for (int q = 0; q < per_wg; ++q) s += part[(q * CG + g) * 8 + j];
      atomicAdd(a.pool + (size_t)n * C + c, s);
    }
  }
}

// ------------------------------------------------------------------------------------------------ SqueezeExcite gate
__global__ void __launch_bounds__(256)
k_mb_se(const float* __restrict__ pool, float inv_hw, const float* __restrict__ w1, const float* __restrict__ b1,
        const float* __restrict__ w2, const float* __restrict__ b2, int C, int R, float* __restrict__ gate) {
  __shared__ float m[576], h[160];
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) m[c] = pool[(size_t)n * C + c] * inv_hw;
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += 256) {
    float s = b1[r];
    for (int c = 0; c < C; ++c) s = fmaf(w1[(size_t)r * C + c], m[c], s);
    h[r] = s > 0.f ? s : 0.f;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = b2[c];
    for (int r = 0; r < R; ++r) s = fmaf(w2[(size_t)c * R + r], h[r], s);
    gate[(size_t)n * C + c] = fminf(fmaxf(s + 3.f, 0.f), 6.f) * (1.f / 6.f);       // Hardsigmoid
  }
}

// ------------------------------------------------------------------------------------------------ pointwise GEMM
struct PwmArgs {
  const u16* x;        // [N][P][Cin] bf16
  const u16* w;        // [CoP][CiP] bf16, zero padded (CoP % 32 == 0, CiP % 16 == 0), BN scale folded
  const float* bias;   // [CoP]
  const float* gate;   // [N][Cin] or null (SqueezeExcite)
  const u16* res;      // [N][P][Cout] or null
  u16* y;              // [N][P][Cout]
  int N, P, Cin, CiP, Cout, tiles_per_img, act;
};

template <int MT>
__global__ void __launch_bounds__(256)
k_mb_pw(const PwmArgs a) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int n = blockIdx.x / a.tiles_per_img;
  const int p = (blockIdx.x - n * a.tiles_per_img) * 128 + wid * 32 + l31;      // this lane's position (B column)
  const int cob = blockIdx.y * (MT * 32);
  const bool pok = p < a.P;
  const u16* __restrict__ xr = a.x + ((size_t)n * a.P + (pok ? p : 0)) * a.Cin;
  const float* __restrict__ gr = a.gate ? a.gate + (size_t)n * a.Cin : nullptr;
  f32x16v acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  for (int k0 = 0; k0 < a.CiP; k0 += 16) {
    const int k = k0 + 8 * half;
    u16x8 b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pok && k < a.Cin) b = *reinterpret_cast<const u16x8*>(xr + k);
    if (gr && k < a.Cin) {
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = f2bf(bf2f(b[j]) * gr[k + j]);
    }
    const bf16x8 bf = __builtin_bit_cast(bf16x8, b);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(a.w + (size_t)(cob + m * 32 + l31) * a.CiP + k);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[m], 0, 0, 0);
    }
  }
  if (!pok) return;
  // acc[m][r]: column = this lane's position, row (output channel) = cob + 32m + (r&3) + 8(r>>2) + 4half
  u16* __restrict__ yr = a.y + ((size_t)n * a.P + p) * a.Cout;
  const u16* __restrict__ rr = a.res ? a.res + ((size_t)n * a.P + p) * a.Cout : nullptr;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = cob + 32 * m + 8 * g + 4 * half;
      if (co >= a.Cout) continue;                                 // Cout % 8 == 0: a group of 4 is all in or all out
      u16x4 rv = {0, 0, 0, 0};
      if (rr) rv = *reinterpret_cast<const u16x4*>(rr + co);
      u16x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = act_apply(acc[m][4 * g + i] + a.bias[co + i], a.act);
        if (rr) v += bf2f(rv[i]);
        o[i] = f2bf(v);
      }
      *reinterpret_cast<u16x4*>(yr + co) = o;
    }
}

// ------------------------------------------------------------------------------------------------ head
// Conv2d(C,5,3,p1) + sigmoid on the S x S feature map: one workgroup per (image, output position); threads split the
// 9*C products, fixed-order LDS reduction.  Output NCHW f32 (what fdet_reduce_bounding_boxes reads).
__global__ void __launch_bounds__(256)
k_mb_head(const u16* __restrict__ f, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
          int N, int S, int C) {
  __shared__ float part[5][256];
  const int pos = blockIdx.x % (S * S), n = blockIdx.x / (S * S);
  const int oy = pos / S, ox = pos - oy * S;
  float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int t = threadIdx.x; t < 9 * C; t += 256) {
    const int tap = t / C, c = t - tap * C;
    const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
    if (iy < 0 || iy >= S || ix < 0 || ix >= S) continue;
    const float v = bf2f(f[(((size_t)n * S + iy) * S + ix) * C + c]);
#pragma unroll
    for (int o = 0; o < 5; ++o) s[o] = fmaf(v, w[((size_t)o * 9 + tap) * C + c], s[o]);   // w: [5][9][C]
  }
#pragma unroll
  for (int o = 0; o < 5; ++o) part[o][threadIdx.x] = s[o];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
#pragma unroll
      for (int o = 0; o < 5; ++o) part[o][threadIdx.x] += part[o][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x < 5) {
    const float z = part[threadIdx.x][0] + bias[threadIdx.x];
    y[(((size_t)n * 5 + threadIdx.x) * S + oy) * S + ox] = 1.f / (1.f + expf(-z));
  }
}

}  // namespace

extern "C" int fdet_mb_stem(const void* x, int x_is_u8, const float* w, const float* bias, void* y, int N, int H, int W,
                            void* stream) {
  FDET_REQUIRE(x && w && bias && y && N > 0 && H >= 2 && W >= 2 && !(H & 1) && !(W & 1), "mb_stem: bad arguments (even H, W)");
  const int Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)N * Ho * Wo;
  size_t blocks = (total + 255) / 256; if (blocks > 16384) blocks = 16384;
  if (x_is_u8) hipLaunchKernelGGL(k_mb_stem<unsigned char>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)x, w, bias, (u16*)y, N, H, W, Ho, Wo, 1.0f / 255.0f);
  else hipLaunchKernelGGL(k_mb_stem<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, w, bias, (u16*)y, N, H, W, Ho, Wo, 1.0f);
  return check_launch("fdet_mb_stem");
}

extern "C" int fdet_mb_depthwise(const void* x, const float* w, const float* bias, void* y, float* pool, int N, int H, int W,
                                 int C, int K, int stride, int act, void* stream) {
  FDET_REQUIRE(x && w && bias && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C <= 576 && (K == 3 || K == 5) &&
               (stride == 1 || stride == 2) && act >= 0 && act <= 2, "mb_depthwise: bad arguments (C %% 8 == 0, C <= 576, K 3|5, stride 1|2)");
  DwArgs a;
  a.x = (const u16*)x; a.w = w; a.bias = bias; a.y = (u16*)y; a.pool = pool;
  a.N = N; a.H = H; a.W = W; a.C = C; a.stride = stride; a.act = act;
  a.Ho = (H + stride - 1) / stride; a.Wo = (W + stride - 1) / stride;
  // stride 1: symmetric K/2.  stride 2: TF "SAME": total = max((Ho-1)*2 + K - H, 0), the smaller half in front
  a.pad = stride == 1 ? K / 2 : std::max((a.Ho - 1) * 2 + K - H, 0) / 2;
  if (stride == 2) FDET_REQUIRE(H == W, "mb_depthwise: stride-2 layers need square maps (one pad value for both axes)");
  if (pool) (void)hipMemsetAsync(pool, 0, (size_t)N * C * sizeof(float), (hipStream_t)stream);
  const int CG = C / 8, per_wg = 256 / CG;
  const int npos = a.Ho * a.Wo;
  int bx = (npos + per_wg - 1) / per_wg;
  const int cap = std::max(1, 4096 / N);
  if (bx > cap) bx = cap;
  const size_t lds = ((size_t)K * K * C + C + 256 * 8) * sizeof(float);
  dim3 grid(bx, N);
  if (K == 3) {
    FDET_REQUIRE(hipFuncSetAttribute((const void*)k_mb_dw<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess, "mb_depthwise: LDS");
    hipLaunchKernelGGL(k_mb_dw<3>, grid, dim3(256), lds, (hipStream_t)stream, a);
  } else {
    FDET_REQUIRE(hipFuncSetAttribute((const void*)k_mb_dw<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess, "mb_depthwise: LDS");
    hipLaunchKernelGGL(k_mb_dw<5>, grid, dim3(256), lds, (hipStream_t)stream, a);
  }
  return check_launch("fdet_mb_depthwise");
}

extern "C" int fdet_mb_se_gate(const float* pool, int HW, const float* w1, const float* b1, const float* w2, const float* b2,
                               int N, int C, int R, float* gate, void* stream) {
  FDET_REQUIRE(pool && w1 && b1 && w2 && b2 && gate && N > 0 && C > 0 && C <= 576 && R > 0 && R <= 160 && HW > 0, "mb_se_gate: bad arguments");
  hipLaunchKernelGGL(k_mb_se, dim3(N), dim3(256), 0, (hipStream_t)stream, pool, 1.0f / (float)HW, w1, b1, w2, b2, C, R, gate);
  return check_launch("fdet_mb_se_gate");
}

extern "C" int fdet_mb_pointwise(const void* x, const void* w, const float* bias, const float* gate, const void* res, void* y,
                                 int N, int P, int Cin, int Cout, int act, void* stream) {
  FDET_REQUIRE(x && w && bias && y && N > 0 && P > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0 && act >= 0 && act <= 2,
               "mb_pointwise: bad arguments (channel counts must be multiples of 8)");
  PwmArgs a;
  a.x = (const u16*)x; a.w = (const u16*)w; a.bias = bias; a.gate = gate; a.res = (const u16*)res; a.y = (u16*)y;
  a.N = N; a.P = P; a.Cin = Cin; a.CiP = (Cin + 15) / 16 * 16; a.Cout = Cout; a.act = act;
  a.tiles_per_img = (P + 127) / 128;
  const int CoT = (Cout + 31) / 32;                               // 32-channel row tiles of the (zero padded) weight panel
  const int MT = CoT >= 6 && CoT % 6 == 0 ? 6 : (CoT % 4 == 0 ? 4 : (CoT % 3 == 0 ? 3 : (CoT % 2 == 0 ? 2 : 1)));
  dim3 grid((unsigned)((size_t)N * a.tiles_per_img), CoT / MT);
  switch (MT) {
    case 6: hipLaunchKernelGGL(k_mb_pw<6>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
    case 4: hipLaunchKernelGGL(k_mb_pw<4>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
    case 3: hipLaunchKernelGGL(k_mb_pw<3>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
    case 2: hipLaunchKernelGGL(k_mb_pw<2>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
    default: hipLaunchKernelGGL(k_mb_pw<1>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
  }
  return check_launch("fdet_mb_pointwise");
}

extern "C" int fdet_mb_head(const void* f, const float* w, const float* bias, float* y, int N, int S, int C, void* stream) {
  FDET_REQUIRE(f && w && bias && y && N > 0 && S > 0 && C > 0, "mb_head: bad arguments");
  hipLaunchKernelGGL(k_mb_head, dim3((unsigned)((size_t)N * S * S)), dim3(256), 0, (hipStream_t)stream, (const u16*)f, w, bias, y, N, S, C);
  return check_launch("fdet_mb_head");
}
