// MobileNetV3-small backbone (BASELINE.json config 5; models/MobilenetV3Backbone.py:11-60 = timm tf_mobilenetv3_small_100
// without its classifier + Conv2d(576,5,3,p1) + sigmoid), inference, bf16.  Engine-private layout: NHWC bf16 (channels
// innermost, every channel count a multiple of 8), BatchNorm folded into the conv weights on the host.  The model is
// memory-bound (depthwise convs and thin 1x1 GEMMs: 2-30 FLOP/B), so every kernel is built around 16-byte channel-vector
// accesses, reads each activation once and writes each once:
//
//   k_mb_stem      Conv2dSame(3,16,3,s2) + BN + Hardswish, NCHW f32 / u8 (the /255 fused) -> NHWC bf16.  A workgroup stages the
//                  5 input rows of 2 output rows in LDS with coalesced row loads (the stride-2 taps then come from LDS);
//                  the 432 weights are wave-uniform scalar loads
//   k_mb_dw<K,S,XS> depthwise KxK (stride 1: pad K/2; stride 2: TF "SAME", the smaller half of the padding in front) + BN + act.
//                  A thread owns 8 channels x a strip of XS output columns: the (XS-1)S+K input columns of a row are loaded
//                  once (branch-free: clamped address + select) and reused by the XS outputs.  Optionally the SqueezeExcite
//                  global-average-pool numerators (per image and channel) as a by-product
//   k_mb_se        SqueezeExcite gate: mean -> FC reduce + ReLU -> FC expand -> Hardsigmoid, one workgroup per image
//   k_mb_pw<MT,WL> pointwise conv as a bf16 MFMA GEMM (v_mfma_f32_32x32x16_bf16, fp32 accumulate): A = weights [Cout][Cin]
//                  and B = activations [position][Cin] are both K-contiguous, so a fragment is one 16-byte load: B straight
//                  from global, A from the workgroup's weight panel in LDS (WL; panels above 64 KB stay in L2).  The SE gate
//                  is applied to the B fragment; bias + activation + residual in the epilogue, which pairs the half-waves'
//                  4-channel groups with v_permlane32_swap so that every store is 16 bytes
//   k_mb_head_*    Conv2d(576,5,3,p1) + sigmoid: per-position MFMA GEMM to the 45 (tap, channel) products (weights split into
//                  bf16 hi + lo: fp32-grade products; the features are read once), then a 9-neighbour gather + bias + sigmoid
//                  -> NCHW f32 maps for the decode / NMS kernels
#include "fdet_conv3x3_x3.h"
#include <algorithm>
#include <cstdint>
#include <cstdlib>

using namespace fdet;

namespace {

typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
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

template <int ACT> __device__ __forceinline__ float act_c(float v) {
  if (ACT == 1) return v > 0.f ? v : 0.f;
  if (ACT == 2) return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
  return v;
}
// bias + activation applied to a whole accumulator set with the activation a compile-time constant: a run-time `act` inside
// the element loops cost two scalar compares and branches PER ELEMENT (the pointwise kernels were instruction-issue
// bound: 4 waves per SIMD each active 24 % of their cycles)
template <int ACT, int PT, int MT>
__device__ __forceinline__ void pw_bias_act(f32x16v (&acc)[PT][MT], const float* __restrict__ bias_cob, int half) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b = *reinterpret_cast<const float4*>(bias_cob + 32 * m + 8 * g + 4 * half);   // bias padded to CoP: always readable
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        acc[j][m][4 * g + 0] = act_c<ACT>(acc[j][m][4 * g + 0] + b.x);
        acc[j][m][4 * g + 1] = act_c<ACT>(acc[j][m][4 * g + 1] + b.y);
        acc[j][m][4 * g + 2] = act_c<ACT>(acc[j][m][4 * g + 2] + b.z);
        acc[j][m][4 * g + 3] = act_c<ACT>(acc[j][m][4 * g + 3] + b.w);
      }
    }
}
template <int ACT, int XS>
__device__ __forceinline__ void dw_act(f32x2v (&acc)[XS][4]) {
#pragma unroll
  for (int q = 0; q < XS; ++q)
#pragma unroll
    for (int h = 0; h < 4; ++h) { acc[q][h].x = act_c<ACT>(acc[q][h].x); acc[q][h].y = act_c<ACT>(acc[q][h].y); }
}

// ------------------------------------------------------------------------------------------------ stem
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const unsigned char* p) {
  const unsigned v = *reinterpret_cast<const unsigned*>(p);
  return float4{(float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u), (float)(v >> 24)};
}

// grid (ceil(Wo/256), ceil(Ho/2), N); thread = one output column, two output rows
template <typename TIN>
__global__ void __launch_bounds__(256)
k_mb_stem(const TIN* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, u16* __restrict__ y,
          int H, int W, int Ho, int Wo, float in_div) {
  constexpr int LW = 516;                                         // 2*256 + 1 input columns of a 256-column tile, padded
  __shared__ __attribute__((aligned(16))) float in[3][5][LW];
  const int n = blockIdx.z, oy0 = blockIdx.y * 2, ox0 = blockIdx.x * 256;
  if ((W & 3) == 0) {
    // 15 rows x 129 four-pixel groups, all of a thread's (up to 8) loads issued before the first LDS write: one memory
    // latency per workgroup instead of one per row
    constexpr int NV = LW / 4;
    float4 val[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx = threadIdx.x + 256 * j, r = idx / NV, c4 = idx - r * NV;
      const int c = r / 5, iy = 2 * oy0 + (r - c * 5), col = 2 * ox0 + 4 * c4;
      const bool ok = idx < 15 * NV && iy < H && col < W;          // W % 4 == 0: a group is all in or all out
      val[j] = load4(x + (((size_t)n * 3 + (ok ? c : 0)) * H + (ok ? iy : 0)) * W + (ok ? col : 0));
      if (!ok) val[j] = float4{0.f, 0.f, 0.f, 0.f};               // TF SAME: nothing in front, zeros behind the last row/column
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx = threadIdx.x + 256 * j, r = idx / NV, c4 = idx - r * NV;
      if (idx < 15 * NV)
        *reinterpret_cast<float4*>(&in[0][0][0] + r * LW + 4 * c4) =
            float4{val[j].x / in_div, val[j].y / in_div, val[j].z / in_div, val[j].w / in_div};      // IEEE division: uint8 frames give exactly the `x / 255.0` of the reference
    }
  } else {
    const int ncol = min(2 * (Wo - ox0) + 1, 513);
    for (int r = 0; r < 15; ++r) {
      const int c = r / 5, iy = 2 * oy0 + (r - c * 5);
      const TIN* __restrict__ src = x + (((size_t)n * 3 + c) * H + (iy < H ? iy : 0)) * W + 2 * ox0;
      for (int t = threadIdx.x; t < ncol; t += 256)
        in[c][r - c * 5][t] = (iy < H && 2 * ox0 + t < W) ? (float)src[t] / in_div : 0.f;
    }
  }
  __syncthreads();
  const int ox = ox0 + threadIdx.x;
  if (ox >= Wo) return;
  // the thread's two output rows ride in the two halves of packed fp32 FMAs (v_pk_fma_f32, the weight a broadcast SGPR)
  f32x2v v[27];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        v[(c * 3 + ky) * 3 + kx].x = in[c][ky][2 * threadIdx.x + kx];
        v[(c * 3 + ky) * 3 + kx].y = in[c][2 + ky][2 * threadIdx.x + kx];
      }
  u16 o[2][16];
#pragma unroll
  for (int co = 0; co < 16; ++co) {
    const float b = bias[co];                                     // uniform addresses: scalar loads
    f32x2v s = {b, b};
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const float wk = w[co * 27 + k];
      s = __builtin_elementwise_fma(v[k], f32x2v{wk, wk}, s);
    }
    o[0][co] = f2bf(act_apply(s.x, 2));
    o[1][co] = f2bf(act_apply(s.y, 2));
  }
  __builtin_memcpy(y + (((size_t)n * Ho + oy0) * Wo + ox) * 16, o[0], 32);
  if (oy0 + 1 < Ho) __builtin_memcpy(y + (((size_t)n * Ho + oy0 + 1) * Wo + ox) * 16, o[1], 32);
}

// ------------------------------------------------------------------------------------------------ depthwise
struct DwArgs {
  const u16* x; const float* w; const float* bias; u16* y; float* pool;   // w: [K*K][C] folded; pool: [N][C] sums or null
  int N, H, W, C, Ho, Wo, pad, act, strips;                               // strips = ceil(Wo / XS) per output row
  int CW, IMG;                                                            // channels per workgroup (blockIdx.z chunk), images per workgroup
  int pool_slots;                                                         // = gridDim.x: partial-sum rows per image in `pool`
  unsigned x_bytes;                                                       // size of x (range of the buffer loads)
};

// one input row of a strip through range-checked buffer loads: a tap outside the image gets an offset past the buffer and
// the hardware returns zeros -- one add and one select per load instead of a clamped 64-bit address plus four selects on
// the loaded dwords (these kernels are instruction-issue bound)
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
template <int NC>
__device__ __forceinline__ void dw_load_row(u16x8 (&v)[NC], __amdgpu_buffer_rsrc_t rs, unsigned img_off, int iy, int H, unsigned row_bytes,
                                            const unsigned (&coff)[NC], const bool (&cok)[NC]) {
  const bool rok = iy >= 0 && iy < H;
  const unsigned roff = img_off + (unsigned)(rok ? iy : 0) * row_bytes;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const unsigned off = (rok && cok[c]) ? roff + coff[c] : 0xFFFFFFF0u;
    v[c] = __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
  }
}

template <int K, int S, int XS>
__global__ void __launch_bounds__(256)
k_mb_dw(const DwArgs a) {
  // one thread = 8 channels x XS adjacent output columns.  A workgroup owns a chunk of CW channels (blockIdx.z) whose
  // K*K*CW weights sit in LDS, and walks IMG images (blockIdx.y) with them; its threads: CGW = CW/8 channel groups
  // fastest, then strips of the image, so the pooled sums of one image's walk go to one row of `pool`
  constexpr int NC = (XS - 1) * S + K;                            // input columns a strip reads per row
  // LDS: [K*K + 1][2][CGW][4] weights and (last row) bias, then the pool scratch.  A thread's 8 channels are two float4 in
  // two planes (channels 0-3 / 4-7 of every group): the 16 lanes the LDS serves together read 16 CONSECUTIVE 16-byte
  // chunks -- with the natural [tap][channel] order lanes cg and cg+8 are 256 B apart, on the same banks (PMC: 75 % of
  // the LDS cycles were bank conflicts)
  extern __shared__ __attribute__((aligned(16))) float wl[];
  // the two-plane order pays for the 5x5 layers (0.150 -> 0.130 ms at 30x30x240); the 3x3 stride-2 layers measured SLOWER
  // with it (0.157 -> 0.205 ms) and keep the natural order (plane offset 4 floats, group pitch 8)
  constexpr bool PL = K == 5;
  const int C = a.C, CW = a.CW, CGW = CW >> 3, c0 = blockIdx.z * CW;
  const int hoff = PL ? (CW >> 1) : 4, gp = PL ? 4 : 8;           // offset of channels 4-7, pitch of a channel group
  for (int t = threadIdx.x; t < K * K * (CW >> 2); t += 256) {
    const int tap = t / (CW >> 2), c4 = t - tap * (CW >> 2);
    *reinterpret_cast<float4*>(wl + tap * CW + (c4 & 1) * hoff + (c4 >> 1) * gp) = *reinterpret_cast<const float4*>(a.w + (size_t)tap * C + c0 + c4 * 4);
  }
  for (int t = threadIdx.x; t < CW; t += 256) wl[K * K * CW + ((t >> 2) & 1) * hoff + (t >> 3) * gp + (t & 3)] = a.bias[c0 + t];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(a.x), 0, (unsigned)a.x_bytes, 0x00020000);
  const unsigned row_bytes = (unsigned)a.W * C * 2u;
  const int per_wg = 256 / CGW;                                   // strips per workgroup pass (CGW <= 72 -> >= 3)
  const int cg = threadIdx.x % CGW, pl = threadIdx.x / CGW;
  const int nitems = a.Ho * a.strips;
  float* part = wl + K * K * CW + CW;                             // [256][8]
  for (int im = 0; im < a.IMG; ++im) {
    const int n = blockIdx.y * a.IMG + im;
    if (n >= a.N) break;                                          // uniform over the workgroup
    const unsigned img_off = ((unsigned)n * a.H * a.W * C + c0 + cg * 8) * 2u;      // byte offsets: the tensor is below 4 GB (host check)
    float psum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (pl < per_wg) {
      for (int it = blockIdx.x * per_wg + pl; it < nitems; it += gridDim.x * per_wg) {
        const int oy = it / a.strips, ox0 = (it - oy * a.strips) * XS;
        unsigned coff[NC];
        bool cok[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int ix = ox0 * S - a.pad + c;
          cok[c] = ix >= 0 && ix < a.W;
          coff[c] = (unsigned)(cok[c] ? ix : 0) * (unsigned)C * 2u;
        }
        // channel pairs ride in packed fp32 FMAs (v_pk_fma_f32): acc[q][h] = channels 2h, 2h+1 of output column q
        f32x2v acc[XS][4];
        {
          const float4 b0 = *reinterpret_cast<const float4*>(wl + K * K * CW + cg * gp);
          const float4 b1 = *reinterpret_cast<const float4*>(wl + K * K * CW + hoff + cg * gp);
#pragma unroll
          for (int q = 0; q < XS; ++q) {
            acc[q][0] = f32x2v{b0.x, b0.y}; acc[q][1] = f32x2v{b0.z, b0.w};
            acc[q][2] = f32x2v{b1.x, b1.y}; acc[q][3] = f32x2v{b1.z, b1.w};
          }
        }
#pragma unroll 1                                                  // one row of taps at a time: ~120 VGPRs, 4 waves per SIMD hide the loads
        for (int ky = 0; ky < K; ++ky) {
          u16x8 v[NC];
          dw_load_row<NC>(v, rs, img_off, oy * S - a.pad + ky, a.H, row_bytes, coff, cok);
          f32x2v xv[NC][4];
#pragma unroll
          for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int h = 0; h < 4; ++h) xv[c][h] = f32x2v{bf2f(v[c][2 * h]), bf2f(v[c][2 * h + 1])};
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            const float4 w0 = *reinterpret_cast<const float4*>(wl + (ky * K + kx) * CW + cg * gp);
            const float4 w1 = *reinterpret_cast<const float4*>(wl + (ky * K + kx) * CW + hoff + cg * gp);
            const f32x2v wv[4] = {f32x2v{w0.x, w0.y}, f32x2v{w0.z, w0.w}, f32x2v{w1.x, w1.y}, f32x2v{w1.z, w1.w}};
#pragma unroll
            for (int q = 0; q < XS; ++q)
#pragma unroll
              for (int h = 0; h < 4; ++h) acc[q][h] = __builtin_elementwise_fma(xv[q * S + kx][h], wv[h], acc[q][h]);
          }
        }
        if (a.act == 2) dw_act<2, XS>(acc); else if (a.act == 1) dw_act<1, XS>(acc);
#pragma unroll
        for (int q = 0; q < XS; ++q) {
          if (ox0 + q >= a.Wo) break;
          u16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            o[j] = f2bf(acc[q][j >> 1][j & 1]);
            psum[j] += bf2f(o[j]);                                // the pool sees what the next layer reads
          }
          *reinterpret_cast<u16x8*>(a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox0 + q) * C + c0 + cg * 8) = o;
        }
      }
    }
    if (a.pool) {
      // SqueezeExcite numerators: fixed-order sum over this workgroup's threads per channel (LDS), written to this
      // workgroup's OWN row of `pool` ([image][gridDim.x][C]); k_mb_se adds the rows in order -- no atomics, so the forward
      // is bit-reproducible (float atomics made run-to-run differences at the bf16-rounding level: a flipped rounding
      // moved a sigmoid output by up to 1e-2)
#pragma unroll
      for (int j = 0; j < 8; ++j) part[threadIdx.x * 8 + j] = (pl < per_wg) ? psum[j] : 0.f;
      __syncthreads();
      for (int c = threadIdx.x; c < CW; c += 256) {
        const int g = c >> 3, j = c & 7;
        float s = 0.f;
        for (int q = 0; q < per_wg; ++q) s += part[(q * CGW + g) * 8 + j];
        a.pool[((size_t)n * a.pool_slots + blockIdx.x) * C + c0 + c] = s;
      }
      __syncthreads();                                            // `part` is rewritten for the next image
    }
  }
}

// ------------------------------------------------------------------------------------------------ SqueezeExcite gate
__global__ void __launch_bounds__(256)
k_mb_se(const float* __restrict__ pool, int slots, float inv_hw, const float* __restrict__ w1, const float* __restrict__ b1,
        const float* __restrict__ w2, const float* __restrict__ b2, int C, int R, float* __restrict__ gate) {
  __shared__ float m[576], h[160];
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int q = 0; q < slots; ++q) s += pool[((size_t)n * slots + q) * C + c];      // the depthwise workgroups' partial sums, in order
    m[c] = s * inv_hw;
  }
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
  unsigned cpr_magic;  // ceil(2^32 / (CiP / 8)): t / (CiP/8) = umulhi(t, magic) for the panel's piece indices
};

template <int MT, bool WL, bool ST, int KU, int PT>
__global__ void __launch_bounds__(256)
k_mb_pw(const PwmArgs a) {
  // WL: [MT*32][CiP + 8] weight panel (row stride = odd number of 16-byte units); ST: then 4 x [32][MT*32 + 4] output staging.
  // A wave owns PT x 32 positions (PT = 2 for narrow outputs: twice the operand bytes in flight per wave -- these layers
  // are bound by memory latency x occupancy, PMC: waves parked 60-77 % of their cycles)
  extern __shared__ __attribute__((aligned(16))) u16 wpan[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int n = blockIdx.x / a.tiles_per_img;
  const int pw0 = (blockIdx.x - n * a.tiles_per_img) * (128 * PT) + wid * (32 * PT);      // first position of this wave
  const int cob = blockIdx.y * (MT * 32);
  const int LS = a.CiP + 8;
  if (WL) {
    // the weight panel: four 16-byte pieces per thread in flight at a time (a load -> wait -> LDS write loop serialised
    // one L2 round trip per piece: up to 27 per thread for the 288-channel panels)
    const int cpr = a.CiP >> 3, total = MT * 32 * cpr;            // 16-byte chunks per weight row
    for (int t0 = threadIdx.x; t0 < total; t0 += 1024) {
      u16x8 v[4];
      int dst[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = min(t0 + 256 * u, total - 1);
        const int r = (int)__umulhi((unsigned)t, a.cpr_magic), c = t - r * cpr;
        v[u] = *reinterpret_cast<const u16x8*>(a.w + (size_t)(cob + r) * a.CiP + c * 8);
        dst[u] = r * LS + c * 8;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (t0 + 256 * u < total) *reinterpret_cast<u16x8*>(wpan + dst[u]) = v[u];
    }
    __syncthreads();
  }
  bool pok[PT];
  const u16* __restrict__ xr[PT];
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    const int p = pw0 + 32 * j + l31;                             // this lane's position (B column) in sub-tile j
    pok[j] = p < a.P;
    xr[j] = a.x + ((size_t)n * a.P + (pok[j] ? p : 0)) * a.Cin;
  }
  const float* __restrict__ gr = a.gate ? a.gate + (size_t)n * a.Cin : nullptr;
  f32x16v acc[PT][MT];
  const f32x16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < a.CiP; k0 += 16 * KU) {
    // KU k-steps of operand loads issued together (KU = 4 for Cin >= 128: one memory latency per 64 input channels)
    u16x8 b[PT][KU];
    float4 g0[KU], g1[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int k = k0 + 16 * u + 8 * half;
      const bool kok = k < a.Cin;                                 // Cin % 8 == 0: a group of 8 is all in or all out
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        b[j][u] = *reinterpret_cast<const u16x8*>(xr[j] + (kok ? k : 0));
        if (!(pok[j] && kok)) b[j][u] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
      if (gr) {
        g0[u] = *reinterpret_cast<const float4*>(gr + (kok ? k : 0));
        g1[u] = *reinterpret_cast<const float4*>(gr + (kok ? k : 0) + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (KU > 1 && k0 + 16 * u >= a.CiP) break;                  // uniform
      const int k = k0 + 16 * u + 8 * half;
      bf16x8 bf[PT];
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        if (gr) {
          const float gv[8] = {g0[u].x, g0[u].y, g0[u].z, g0[u].w, g1[u].x, g1[u].y, g1[u].z, g1[u].w};
#pragma unroll
          for (int i = 0; i < 8; ++i) b[j][u][i] = f2bf(bf2f(b[j][u][i]) * gv[i]);
        }
        bf[j] = __builtin_bit_cast(bf16x8, b[j][u]);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const bf16x8 af = WL ? *reinterpret_cast<const bf16x8*>(wpan + (m * 32 + l31) * LS + k)
                             : *reinterpret_cast<const bf16x8*>(a.w + (size_t)(cob + m * 32 + l31) * a.CiP + k);
        // the very first k-step takes a constant-zero C operand: no accumulator initialisation instructions
#pragma unroll
        for (int j = 0; j < PT; ++j)
          acc[j][m] = (k0 == 0 && u == 0) ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[j], zero16, 0, 0, 0)
                                          : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[j], acc[j][m], 0, 0, 0);
      }
    }
  }
  // acc[j][m][r]: column = this lane's position of sub-tile j, row (output channel) = cob + 32m + (r&3) + 8(r>>2) + 4half
  if (a.act == 2) pw_bias_act<2, PT, MT>(acc, a.bias + cob, half);
  else if (a.act == 1) pw_bias_act<1, PT, MT>(acc, a.bias + cob, half);
  else pw_bias_act<0, PT, MT>(acc, a.bias + cob, half);
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    const int p = pw0 + 32 * j + l31;
    const size_t row = ((size_t)n * a.P + (pok[j] ? p : 0)) * a.Cout;
    if (ST) {
      // the wave's 32 positions x MT*32 channels go through LDS so that the stores walk memory contiguously (a position's
      // channels are contiguous, and so are consecutive positions when the workgroup covers all of Cout).  Row pitch
      // MT*32 + 4 bf16 = an odd number of 8-byte units: the 8-byte writes of 32 lanes fall on 64 distinct banks
      constexpr int SS = MT * 32 + 4;
      u16* stg = wpan + (WL ? MT * 32 * LS : 0) + wid * 32 * SS;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = cob + 32 * m + 8 * g + 4 * half;
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[j][m][4 * g + i];
          if (a.res && pok[j] && ch < a.Cout) {
            const u16x4 rv = *reinterpret_cast<const u16x4*>(a.res + row + ch);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += bf2f(rv[i]);
          }
          uint2 o;
          o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
          o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
          *reinterpret_cast<uint2*>(stg + l31 * SS + 32 * m + 8 * g + 4 * half) = o;
        }
      __syncthreads();
      const int cw = min(MT * 32, a.Cout - cob), cpr = cw >> 3;   // 16-byte chunks per position
      const int p0 = pw0 + 32 * j;
      u16* __restrict__ yb = a.y + ((size_t)n * a.P + p0) * a.Cout + cob;
      const unsigned om = 0xFFFFFFFFu / (unsigned)cpr + 1u;       // idx / cpr = umulhi(idx, om) for idx < 32 * cpr (wave-uniform: scalar)
      for (int idx = lane; idx < 32 * cpr; idx += 64) {
        const int r = (int)__umulhi((unsigned)idx, om), c = idx - r * cpr;
        if (p0 + r < a.P) {
          const uint2 lo = *reinterpret_cast<const uint2*>(stg + r * SS + c * 8), hi = *reinterpret_cast<const uint2*>(stg + r * SS + c * 8 + 4);
          *reinterpret_cast<uint4*>(yb + (size_t)r * a.Cout + c * 8) = uint4{lo.x, lo.y, hi.x, hi.y};
        }
      }
      if (j + 1 < PT) __syncthreads();                            // the staging rows are rewritten by the next sub-tile
      continue;
    }
    // direct stores: the half-waves exchange 4-channel groups (v_permlane32_swap) so that the lower half owns channels
    // 16q..16q+7 and the upper half 16q+8..16q+15 of the position: one 16-byte store each
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int co = cob + 32 * m + 16 * q + 8 * half;          // first of this lane's 8 channels after the exchange
        unsigned pk[2][2];                                        // [group 2q, 2q+1][two packed bf16 pairs] before the exchange
        float v[2][4];
#pragma unroll
        for (int gq = 0; gq < 2; ++gq)
#pragma unroll
          for (int i = 0; i < 4; ++i) v[gq][i] = acc[j][m][4 * (2 * q + gq) + i];
        if (a.res) {
          // the residual is added BEFORE the exchange, in this lane's own (pre-exchange) channel groups
#pragma unroll
          for (int gq = 0; gq < 2; ++gq) {
            const int ch = cob + 32 * m + 8 * (2 * q + gq) + 4 * half;
            if (pok[j] && ch < a.Cout) {
              const u16x4 rv = *reinterpret_cast<const u16x4*>(a.res + row + ch);
#pragma unroll
              for (int i = 0; i < 4; ++i) v[gq][i] += bf2f(rv[i]);
            }
          }
        }
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
          pk[gq][0] = (unsigned)f2bf(v[gq][0]) | ((unsigned)f2bf(v[gq][1]) << 16);
          pk[gq][1] = (unsigned)f2bf(v[gq][2]) | ((unsigned)f2bf(v[gq][3]) << 16);
        }
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
        if (pok[j] && co < a.Cout) {                              // Cout % 8 == 0: 8 channels all in or all out
          uint4 o; o.x = s0[0]; o.y = s1[0]; o.z = s0[1]; o.w = s1[1];
          *reinterpret_cast<uint4*>(a.y + row + co) = o;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------ head
// Conv2d(C,5,3,p1) + sigmoid on the S x S map.  The conv is linear, so it is computed as (1) a per-position GEMM
// g[q][tap*5 + ch] = W[ch][tap] . f[q] -- the features are read ONCE, 45 outputs per position in two 32-row tiles, weights
// bf16 hi + lo (two MFMAs per tile and step: the products carry ~16 mantissa bits of the fp32 weight) -- and (2) a gather
// y[ch][pos] = sigmoid(bias + sum over the 9 taps of g[pos + tap offset][tap*5 + ch]), zeros outside the map.
// w: [2][64][C] bf16 (hi, lo; rows 45..63 zero), g: [total][48] f32.
__global__ void __launch_bounds__(256)
k_mb_head_gemm(const u16* __restrict__ f, const u16* __restrict__ w, float* __restrict__ g, int total, int C) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int q = blockIdx.x * 128 + wid * 32 + l31;
  const bool qok = q < total;
  const u16* __restrict__ fr = f + (size_t)(qok ? q : 0) * C + 8 * half;
  const u16* __restrict__ w0 = w + (size_t)l31 * C + 8 * half;               // tile 0 row; tile 1 = +32 rows; lo = +64 rows
  f32x16v acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  for (int k0 = 0; k0 < C; k0 += 32) {
    u16x8 b[2], ah[2][2], al[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = min(k0 + 16 * u, C - 16);                     // C % 16 == 0; a step past the end is loaded but not used
      b[u] = *reinterpret_cast<const u16x8*>(fr + k);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        ah[u][m] = *reinterpret_cast<const u16x8*>(w0 + (size_t)(32 * m) * C + k);
        al[u][m] = *reinterpret_cast<const u16x8*>(w0 + (size_t)(64 + 32 * m) * C + k);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (k0 + 16 * u >= C) break;
      if (!qok) b[u] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
      const bf16x8 bf = __builtin_bit_cast(bf16x8, b[u]);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[u][m]), bf, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[u][m]), bf, acc[m], 0, 0, 0);
      }
    }
  }
  if (!qok) return;
  // acc[m][4g + i]: row 32m + 8g + 4half + i of this lane's position; rows < 48 are stored (45..47 are zeros)
  float* __restrict__ gq = g + (size_t)q * 48;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      const int row = 32 * m + 8 * gi + 4 * half;
      if (row < 48) *reinterpret_cast<float4*>(gq + row) = float4{acc[m][4 * gi], acc[m][4 * gi + 1], acc[m][4 * gi + 2], acc[m][4 * gi + 3]};
    }
}

__global__ void __launch_bounds__(256)
k_mb_head_gather(const float* __restrict__ g, const float* __restrict__ bias, float* __restrict__ y, int total, int S) {
  const int t = blockIdx.x * 256 + threadIdx.x;                   // one thread per (position, channel)
  if (t >= total * 5) return;
  const int q = t / 5, ch = t - q * 5;
  const int n = q / (S * S), pos = q - n * S * S, oy = pos / S, ox = pos - oy * S;
  float z = 0.f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {                             // fixed tap order
    const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
    if (iy >= 0 && iy < S && ix >= 0 && ix < S) z += g[((size_t)(n * S + iy) * S + ix) * 48 + tap * 5 + ch];
  }
  y[((size_t)n * 5 + ch) * S * S + pos] = 1.f / (1.f + expf(-(z + bias[ch])));
}

}  // namespace

extern "C" int fdet_mb_stem(const void* x, int x_is_u8, const float* w, const float* bias, void* y, int N, int H, int W,
                            void* stream) {
  FDET_REQUIRE(x && w && bias && y && N > 0 && H >= 2 && W >= 2 && !(H & 1) && !(W & 1), "mb_stem: bad arguments (even H, W)");
  FDET_REQUIRE(N <= 65535 && H / 4 + 1 <= 65535, "mb_stem: batch or height beyond the launch grid");
  const int Ho = H / 2, Wo = W / 2;
  dim3 grid((Wo + 255) / 256, (Ho + 1) / 2, N);
  if (x_is_u8) hipLaunchKernelGGL(k_mb_stem<unsigned char>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)x, w, bias, (u16*)y, H, W, Ho, Wo, 255.0f);
  else hipLaunchKernelGGL(k_mb_stem<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, w, bias, (u16*)y, H, W, Ho, Wo, 1.0f);
  return check_launch("fdet_mb_stem");
}

// launch geometry of the depthwise kernel (also what sizes the SqueezeExcite partial-sum rows)
struct DwPlan { int strips, nchunk, CW, bx, IMG; };
static DwPlan dw_plan(int N, int C, int Ho, int Wo, int XS) {
  DwPlan p;
  p.strips = (Wo + XS - 1) / XS;
  // channel chunks of at most 192 (28 KB of LDS with 5x5 taps: 5 workgroups per CU); the chunk must be a multiple of 8
  p.nchunk = (C + 191) / 192;
  while (p.nchunk < C / 8 && C % (8 * p.nchunk)) ++p.nchunk;
  p.CW = C / p.nchunk;
  const int CGW = p.CW / 8, per_wg = 256 / CGW;
  const int passes = (Ho * p.strips + per_wg - 1) / per_wg;       // workgroup passes over one image's strips
  // workgroup columns per image: a function of the IMAGE geometry only (about 16), so that the order in which an
  // image's SqueezeExcite sums are added does not depend on the batch size; beyond ~8192 workgroups per launch a
  // workgroup takes several images with one fill of its weights
  const int ppi = std::max(1, (passes + 8) / 16);                 // passes per workgroup: ~16 columns, evenly loaded
  p.bx = (passes + ppi - 1) / ppi;
  const long long total = (long long)N * p.nchunk * p.bx;
  p.IMG = (int)std::min<long long>(std::max<long long>(1, total / 8192), 64);
  return p;
}
static int dw_xs(int K, int stride) { return stride == 1 ? 4 : 2; }

template <int K, int S, int XS>
static int launch_dw(DwArgs a, hipStream_t st) {
  const DwPlan p = dw_plan(a.N, a.C, a.Ho, a.Wo, XS);
  a.strips = p.strips; a.CW = p.CW; a.IMG = p.IMG; a.pool_slots = p.bx;
  const size_t lds = ((size_t)K * K * a.CW + a.CW + 256 * 8) * sizeof(float);
  static bool attr_done = false;                                  // per instantiation
  if (!attr_done) {
    FDET_REQUIRE(hipFuncSetAttribute((const void*)k_mb_dw<K, S, XS>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess, "mb_depthwise: LDS");
    attr_done = true;
  }
  FDET_REQUIRE(lds <= 80 * 1024, "mb_depthwise: channel chunk does not fit LDS");
  hipLaunchKernelGGL((k_mb_dw<K, S, XS>), dim3(p.bx, (a.N + a.IMG - 1) / a.IMG, p.nchunk), dim3(256), lds, st, a);
  return check_launch("fdet_mb_depthwise");
}

extern "C" int fdet_mb_depthwise_pool_slots(int N, int H, int W, int C, int K, int stride) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || (K != 3 && K != 5) || (stride != 1 && stride != 2)) return 0;
  return dw_plan(N, C, (H + stride - 1) / stride, (W + stride - 1) / stride, dw_xs(K, stride)).bx;
}

extern "C" int fdet_mb_depthwise(const void* x, const float* w, const float* bias, void* y, float* pool, int N, int H, int W,
                                 int C, int K, int stride, int act, void* stream) {
  FDET_REQUIRE(x && w && bias && y && N > 0 && N <= 65535 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C <= 576 && (K == 3 || K == 5) &&
               (stride == 1 || stride == 2) && act >= 0 && act <= 2, "mb_depthwise: bad arguments (C %% 8 == 0, C <= 576, K 3|5, stride 1|2)");
  DwArgs a;
  a.x = (const u16*)x; a.w = w; a.bias = bias; a.y = (u16*)y; a.pool = pool;
  a.N = N; a.H = H; a.W = W; a.C = C; a.act = act;
  FDET_REQUIRE((size_t)N * H * W * C * 2 < 0xFFFF0000ull, "mb_depthwise: input tensor beyond the 4 GB range of 32-bit buffer offsets");
  a.x_bytes = (unsigned)((size_t)N * H * W * C * 2);
  a.Ho = (H + stride - 1) / stride; a.Wo = (W + stride - 1) / stride;
  // stride 1: symmetric K/2.  stride 2: TF "SAME": total = max((Ho-1)*2 + K - H, 0), the smaller half in front
  a.pad = stride == 1 ? K / 2 : std::max((a.Ho - 1) * 2 + K - H, 0) / 2;
  if (stride == 2) FDET_REQUIRE(H == W, "mb_depthwise: stride-2 layers need square maps (one pad value for both axes)");
  hipStream_t st = (hipStream_t)stream;
  if (K == 3 && stride == 1) return launch_dw<3, 1, 4>(a, st);
  if (K == 3) return launch_dw<3, 2, 2>(a, st);
  if (stride == 1) return launch_dw<5, 1, 4>(a, st);
  return launch_dw<5, 2, 2>(a, st);
}

extern "C" int fdet_mb_se_gate(const float* pool, int slots, int HW, const float* w1, const float* b1, const float* w2,
                               const float* b2, int N, int C, int R, float* gate, void* stream) {
  FDET_REQUIRE(pool && w1 && b1 && w2 && b2 && gate && N > 0 && C > 0 && C <= 576 && R > 0 && R <= 160 && HW > 0 && slots > 0, "mb_se_gate: bad arguments");
  hipLaunchKernelGGL(k_mb_se, dim3(N), dim3(256), 0, (hipStream_t)stream, pool, slots, 1.0f / (float)HW, w1, b1, w2, b2, C, R, gate);
  return check_launch("fdet_mb_se_gate");
}

template <int MT, bool WL, bool ST, int KU, int PT>
static int launch_pw4(PwmArgs a, size_t lds, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done && lds) {
    FDET_REQUIRE(hipFuncSetAttribute((const void*)k_mb_pw<MT, WL, ST, KU, PT>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) == hipSuccess, "mb_pointwise: LDS");
    attr_done = true;
  }
  a.tiles_per_img = (a.P + 128 * PT - 1) / (128 * PT);
  FDET_REQUIRE((size_t)a.N * a.tiles_per_img < 0x7fffffffu, "mb_pointwise: too many position tiles");
  dim3 grid((unsigned)((size_t)a.N * a.tiles_per_img), ((a.Cout + 31) / 32) / MT);
  hipLaunchKernelGGL((k_mb_pw<MT, WL, ST, KU, PT>), grid, dim3(256), lds, st, a);
  return check_launch("fdet_mb_pointwise");
}

template <int MT, bool WL, bool ST>
static int launch_pw2(const PwmArgs& a, size_t lds, hipStream_t st) {
  // two 32-position sub-tiles per wave while the accumulators stay within ~100 registers (MT <= 3)
  static const int pt_env = std::getenv("FDET_MB_PW_PT") ? std::atoi(std::getenv("FDET_MB_PW_PT")) : 0;
  constexpr bool can2 = MT <= 3;
  const bool pt2 = can2 && pt_env == 2;                           // measured slower (212 VGPRs: half the waves): opt-in only
  if constexpr (can2) {
    if (pt2) return a.CiP >= 128 ? launch_pw4<MT, WL, ST, 4, 2>(a, lds, st) : launch_pw4<MT, WL, ST, 1, 2>(a, lds, st);
  }
  return a.CiP >= 128 ? launch_pw4<MT, WL, ST, 4, 1>(a, lds, st) : launch_pw4<MT, WL, ST, 1, 1>(a, lds, st);
}

template <int MT>
static int launch_pw(const PwmArgs& a, hipStream_t st) {
  // LDS budget 72 KB (2 workgroups per CU): the weight panel first (above 64 KB it stays in L2), then the output staging
  static const int no_stage = std::getenv("FDET_MB_PW_STAGE") ? !std::atoi(std::getenv("FDET_MB_PW_STAGE")) : 0;
  const size_t panel = (size_t)MT * 32 * (a.CiP + 8) * sizeof(u16);
  const size_t stage = (size_t)4 * 32 * (MT * 32 + 4) * sizeof(u16);
  const bool wl = panel <= 64 * 1024;
  const bool stg = !no_stage && (wl ? panel : 0) + stage <= 72 * 1024;
  if (wl && stg) return launch_pw2<MT, true, true>(a, panel + stage, st);
  if (wl) return launch_pw2<MT, true, false>(a, panel, st);
  if (stg) return launch_pw2<MT, false, true>(a, stage, st);
  return launch_pw2<MT, false, false>(a, 0, st);
}

extern "C" int fdet_mb_pointwise(const void* x, const void* w, const float* bias, const float* gate, const void* res, void* y,
                                 int N, int P, int Cin, int Cout, int act, void* stream) {
  FDET_REQUIRE(x && w && bias && y && N > 0 && P > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0 && act >= 0 && act <= 2,
               "mb_pointwise: bad arguments (channel counts must be multiples of 8)");
  PwmArgs a;
  a.x = (const u16*)x; a.w = (const u16*)w; a.bias = bias; a.gate = gate; a.res = (const u16*)res; a.y = (u16*)y;
  a.N = N; a.P = P; a.Cin = Cin; a.CiP = (Cin + 15) / 16 * 16; a.Cout = Cout; a.act = act;
  a.cpr_magic = (unsigned)(0xFFFFFFFFull / (unsigned)(a.CiP / 8)) + 1u;
  const int CoT = (Cout + 31) / 32;                               // 32-channel row tiles of the (zero padded) weight panel
  // Row tiles per workgroup (the rest of Cout goes to blockIdx.y, re-reading the input from L2).  Measured over this
  // network's layers: 2 tiles (a position's 64 channels = one 128-byte line per workgroup) beat 4-6 tiles by 20-33 % on
  // the wide outputs (240, 576 channels: more waves in flight per CU matter more than reading the input once), 3 tiles
  // when the count is a multiple of 3 but odd (72, 88, 96, 288 channels), and single tiles for the rest -- never for
  // output-dominated layers with an even count: 64-byte pieces of a position written by different workgroups cost 2.3x
  static const int mt_force = std::getenv("FDET_MB_PW_MT") ? std::atoi(std::getenv("FDET_MB_PW_MT")) : 0;
  int MT = CoT % 2 == 0 ? 2 : (CoT % 3 == 0 ? 3 : 1);
  if (mt_force >= 1 && mt_force <= 3 && CoT % mt_force == 0) MT = mt_force;
  hipStream_t st = (hipStream_t)stream;
  switch (MT) {
    case 3: return launch_pw<3>(a, st);
    case 2: return launch_pw<2>(a, st);
    default: return launch_pw<1>(a, st);
  }
}

extern "C" size_t fdet_mb_head_ws_bytes(int N, int S) { return N > 0 && S > 0 ? (size_t)N * S * S * 48 * sizeof(float) : 0; }

extern "C" int fdet_mb_head(const void* f, const void* w, const float* bias, float* y, void* ws, size_t ws_bytes, int N, int S,
                            int C, void* stream) {
  FDET_REQUIRE(f && w && bias && y && ws && N > 0 && S > 0 && C > 0 && C % 16 == 0 && (size_t)N * S * S * 5 < 0x7fffffffu,
               "mb_head: bad arguments (C %% 16 == 0)");
  FDET_REQUIRE(ws_bytes >= fdet_mb_head_ws_bytes(N, S), "mb_head: workspace smaller than fdet_mb_head_ws_bytes()");
  const int total = N * S * S;
  hipLaunchKernelGGL(k_mb_head_gemm, dim3((total + 127) / 128), dim3(256), 0, (hipStream_t)stream, (const u16*)f, (const u16*)w, (float*)ws, total, C);
  FDET_REQUIRE(check_launch("fdet_mb_head") == 0, "mb_head: gemm launch failed");
  hipLaunchKernelGGL(k_mb_head_gather, dim3((total * 5 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)ws, bias, y, total, S);
  return check_launch("fdet_mb_head");
}
