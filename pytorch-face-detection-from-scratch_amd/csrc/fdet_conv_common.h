// Shared pieces of the 3x3 conv kernels (fp32-MFMA and bf16x3-MFMA variants): argument block,
// epilogue fusion modes and the mode-specialised epilogue.
#pragma once
#include "fdet_common.h"
#include <cstdlib>

using namespace fdet;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace fdet {
// (named namespace: the bf16x3 kernels pass this block between translation units)
struct ConvArgs {
  const float* x;        // [N,Cin,H,W]
  const float* wpk;      // [Cin*9][CoP]
  const float* bias;     // [Cout] or null
  float* y_full;         // fwd: lrelu(conv+bias)        | dgrad: dx
  const float* skip;     // fwd: residual input or null  | dgrad: `add` or null
  const float* scale;    // fwd: [N,Cout] dropout scale or null
  float* y_out;          // fwd: z*scale + skip or null
  const float* act;      // dgrad: lrelu' source or null
  int N, Cin, Cout, CoP, H, W, WP, R, VR, CS, nbands, dgrad;
  int mode;              // EPI_* fusion mode of the epilogue
  int dbg;               // development ablation flags (FDET_CONV_DBG), 0 in production
  int stagger;           // start delay (x 64*127 clocks) of every second co-resident workgroup
  int lpr_log2;          // log2(lanes per staged row), lanes >= W/VW
  unsigned magic_h1;     // ceil(2^32/(H+1))
  unsigned magic_rows;   // ceil(2^32/(R+2))
  float slope;
};
}  // namespace fdet

namespace {

constexpr int CK = 8;        // input channels per LDS chunk
constexpr int NTHR = 256;    // 4 waves
// B prefetch slots per thread (vector items of VW floats)
__host__ __device__ constexpr int nbmax(int vw) { return vw == 4 ? 6 : (vw == 2 ? 8 : 10); }


typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// native vector types (HIP's float4 struct defeats scalar replacement of the prefetch arrays)
template <int VW> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = f32x2; };
template <> struct Vec<4> { using T = f32x4; };
template <int VW> __device__ __forceinline__ float vget(const typename Vec<VW>::T& v, int k) { return v[k]; }
template <> __device__ __forceinline__ float vget<1>(const float& v, int) { return v; }

// floor(v/d) for 0 <= v < 2^20, d < 2^12 with magic = ceil(2^32/d); magic == 0 encodes d == 1
__device__ __forceinline__ int fdiv(int v, unsigned magic) { return magic ? (int)__umulhi((unsigned)v, magic) : v; }

// Fusion modes of the epilogue (host-selected, wave-uniform).  The fast modes assume the channel
// count is a multiple of 32 and the listed pointers are non-null; everything else is GENERIC.
enum { EPI_GENERIC = 0,
       EPI_FWD_FULL,    // y_full = lrelu(acc + bias)
       EPI_FWD_BOTH,    // y_full = z ; y_out = z*scale + skip        (training, un-pooled block tail)
       EPI_FWD_OUT,     // y_out = z + skip                          (eval, un-pooled block tail)
       EPI_DGRAD_ACT,   // dx = acc * lrelu'(act)
       EPI_DGRAD_ADD,   // dx = acc + add
       EPI_FWD_POOL,    // pool_out = maxpool2x2(z*scale + skip) + routing bytes   (pooled block tail; ping-pong kernel only)
       EPI_DGRAD_ADDPOOL }; // dx = acc + unpool(dout) through the routing bytes  (ping-pong kernel only)

template <int MT, int NT, int MODE>
__device__ __forceinline__ void epilogue(const ConvArgs& a, f32x16 (&acc)[MT][NT], const bool (&okn)[NT],
                                         const size_t (&basen)[NT], const int (&imgn)[NT], int cob0, size_t HW) {
  const float* __restrict__ g_bias = a.bias;
  const float* __restrict__ g_skip = a.skip;
  const float* __restrict__ g_scale = a.scale;
  const float* __restrict__ g_act = a.act;
  float* __restrict__ g_full = a.y_full;
  float* __restrict__ g_out = a.y_out;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int cobase = cob0 + m * 32;                    // + (r&3) + 8*(r>>2)
    float bz[16];
    if (MODE == EPI_FWD_FULL || MODE == EPI_FWD_BOTH || MODE == EPI_FWD_OUT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) bz[r] = g_bias[cobase + (r & 3) + 8 * (r >> 2)];
    } else if (MODE == EPI_GENERIC) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cobase + (r & 3) + 8 * (r >> 2);
        bz[r] = (g_bias && co < a.Cout) ? g_bias[co] : 0.f;
      }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      if (!okn[n]) continue;
      const size_t idx0 = basen[n] + (size_t)cobase * HW;
      float t0[16], t1[16];
      if (MODE == EPI_FWD_FULL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float z = acc[m][n][r] + bz[r];
          g_full[idx0 + ((r & 3) + 8 * (r >> 2)) * HW] = z > 0.f ? z : z * a.slope;
        }
      } else if (MODE == EPI_FWD_BOTH || MODE == EPI_FWD_OUT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cr = (r & 3) + 8 * (r >> 2);
          t0[r] = g_skip[idx0 + cr * HW];
          if (MODE == EPI_FWD_BOTH) t1[r] = g_scale[(size_t)imgn[n] * a.Cout + cobase + cr];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cr = (r & 3) + 8 * (r >> 2);
          float z = acc[m][n][r] + bz[r];
          z = z > 0.f ? z : z * a.slope;
          if (MODE == EPI_FWD_BOTH) { g_full[idx0 + cr * HW] = z; g_out[idx0 + cr * HW] = z * t1[r] + t0[r]; }
          else g_out[idx0 + cr * HW] = z + t0[r];
        }
      } else if (MODE == EPI_DGRAD_ACT || MODE == EPI_DGRAD_ADD) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          t0[r] = (MODE == EPI_DGRAD_ACT ? g_act : g_skip)[idx0 + ((r & 3) + 8 * (r >> 2)) * HW];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float z = acc[m][n][r];
          if (MODE == EPI_DGRAD_ACT) z *= (t0[r] > 0.f) ? 1.f : a.slope; else z += t0[r];
          g_full[idx0 + ((r & 3) + 8 * (r >> 2)) * HW] = z;
        }
      } else if (!a.dgrad) {                              // GENERIC forward
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cr = (r & 3) + 8 * (r >> 2);
          const bool cok = cobase + cr < a.Cout;
          t0[r] = (g_out && g_skip && cok) ? g_skip[idx0 + cr * HW] : 0.f;
          t1[r] = (g_out && g_scale && cok) ? g_scale[(size_t)imgn[n] * a.Cout + cobase + cr] : 1.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cr = (r & 3) + 8 * (r >> 2);
          if (cobase + cr >= a.Cout) continue;
          float z = acc[m][n][r] + bz[r];
          z = z > 0.f ? z : z * a.slope;
          if (g_full) g_full[idx0 + cr * HW] = z;
          if (g_out) g_out[idx0 + cr * HW] = z * t1[r] + t0[r];
        }
      } else {                                            // GENERIC data gradient
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cr = (r & 3) + 8 * (r >> 2);
          const bool cok = cobase + cr < a.Cout;
          t0[r] = (g_act && cok) ? g_act[idx0 + cr * HW] : 1.f;
          t1[r] = (g_skip && cok) ? g_skip[idx0 + cr * HW] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cr = (r & 3) + 8 * (r >> 2);
          if (cobase + cr >= a.Cout) continue;
          float z = acc[m][n][r];
          if (g_act) z *= (t0[r] > 0.f) ? 1.f : a.slope;
          z += t1[r];
          g_full[idx0 + cr * HW] = z;
        }
      }
    }
  }
}


inline unsigned magic_of(int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

}  // namespace
