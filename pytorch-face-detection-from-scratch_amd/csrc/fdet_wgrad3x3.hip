// Weight/bias gradient of the 3x3 convs as fp32 implicit GEMM on v_mfma_f32_32x32x2_f32.
//
//   dW[co][ci][tap] = sum_q dz[co][q] * x[ci][q + off(tap)]       (q over all positions)
// GEMM: M = co (A = dz, one LDS row per channel), N = 32 input channels at ONE tap (B = x,
// one LDS row per channel, constant tap offset), K = positions, two per MFMA (lanes 0-31 take
// q, lanes 32-63 take q+1).  Same flattened-padded-row layout as fdet_conv3x3.hip; garbage
// positions contribute nothing because the dz tile is zero there.
// Each workgroup walks bands grid-stride, keeps its 9 tap tiles per wave in registers
// (144 accumulator VGPRs) and finally writes one slab; a second kernel sums the slabs in
// fixed order (deterministic, no float atomics).
#include "fdet_common.h"

using namespace fdet;
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct WgArgs {
  const float* x;    // [N,Cin,H,W]
  const float* dz;   // [N,Cout,H,W]
  float* ws;         // [nslab][9][CoP][CiP]
  float* wsb;        // [nslab][CoP]
  int N, Cin, Cout, CoP, CiP, H, W, WP, R, VR, CSZ, CSX, nbands;
};

template <int MT>
__global__ void __launch_bounds__(256, 2)
k_wgrad3x3(const WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MB = MT * 32;
  constexpr int KS = 4 / MT;                    // K-splits (waves sharing one co tile)
  float* Z = reinterpret_cast<float*>(smem);    // [MB][CSZ]
  float* X = Z + MB * a.CSZ;                    // [32][CSX]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int cig = blockIdx.y, cob = blockIdx.z;
  const int m = wid % MT, kh = wid / MT;
  const int H1 = a.H + 1, WP = a.WP, W = a.W;
  const int Q = a.R * WP;
  int Lk = (Q + KS - 1) / KS; Lk = (Lk + 1) & ~1;
  const int qs = kh * Lk, qe = min(Q, qs + Lk);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  int tapoff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) tapoff[t] = (t / 3) * WP + (t % 3);

  const float* zr = Z + (m * 32 + l31) * a.CSZ + half;
  const float* xr = X + l31 * a.CSX + half;
  const int zrows = a.R, xrows = a.R + 2;

  for (int band = blockIdx.x; band < a.nbands; band += gridDim.x) {
    const int v0 = band * a.R;
    __syncthreads();
    // ---- stage dz tile: MB channels x R rows x WP (zero column, zero rows) + tail pad
    for (int t = tid; t < MB * a.CSZ; t += 256) {
      const int c = t / a.CSZ, p = t - c * a.CSZ;
      const int tr = p / WP, ox = p - tr * WP;
      const int v = v0 + tr;
      const int n = v / H1, yy = v - n * H1 - 1;
      const int co = cob * MB + c;
      float val = 0.f;
      if (tr < zrows && ox < W && v < a.VR && yy >= 0 && co < a.Cout)
        val = a.dz[(((size_t)n * a.Cout + co) * a.H + yy) * W + ox];
      Z[t] = val;
    }
    // ---- stage x tile: 32 channels x (R+2) rows; position p = tr*WP + 1 + ix, p=tr*WP is the zero column
    for (int t = tid; t < 32 * a.CSX; t += 256) {
      const int c = t / a.CSX, p = t - c * a.CSX;
      const int tr = p / WP, ix = p - tr * WP - 1;
      const int v = v0 - 1 + tr;
      const int ci = cig * 32 + c;
      float val = 0.f;
      if (tr < xrows && ix >= 0 && v >= 0 && v < a.VR && ci < a.Cin) {
        const int n = v / H1, yy = v - n * H1 - 1;
        if (yy >= 0) val = a.x[(((size_t)n * a.Cin + ci) * a.H + yy) * W + ix];
      }
      X[t] = val;
    }
    __syncthreads();
    for (int q = qs; q < qe; q += 2) {
      const float av = zr[q];
      bsum += av;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float bv = xr[q + tapoff[t]];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
      }
    }
  }

  // ---- write slab: ws[s][tap][co][ci], wsb[s][co]
  const int s = blockIdx.x * KS + kh;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cob * MB + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int ci = cig * 32 + l31;
      a.ws[(((size_t)s * 9 + t) * a.CoP + co) * a.CiP + ci] = acc[t][r];
    }
  }
  bsum += __shfl_xor(bsum, 32, 64);
  if (cig == 0 && half == 0) a.wsb[(size_t)s * a.CoP + cob * MB + m * 32 + l31] = bsum;
}

// Fixed-order reduction of the slabs.  One workgroup per (tap, co): 64-thread rows of ci,
// four slab phases, then an LDS combine in phase order.
__global__ void __launch_bounds__(256)
k_wgrad3x3_reduce(const float* __restrict__ ws, const float* __restrict__ wsb, int nslab, int Cout, int Cin,
                  int CoP, int CiP, float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float part[256];
  const int tap = blockIdx.x, co = blockIdx.y;
  const int cil = threadIdx.x & 63, ph = threadIdx.x >> 6;
  for (int c0 = 0; c0 < CiP; c0 += 64) {
    const int ci = c0 + cil;
    float s = 0.f;
    if (ci < CiP)
      for (int k = ph; k < nslab; k += 4) s += ws[(((size_t)k * 9 + tap) * CoP + co) * CiP + ci];
    part[threadIdx.x] = s;
    __syncthreads();
    if (ph == 0 && ci < Cin) {
      const float tot = ((part[cil] + part[64 + cil]) + part[128 + cil]) + part[192 + cil];
      dW[((size_t)co * Cin + ci) * 9 + tap] = tot;
    }
    __syncthreads();
  }
  if (db && tap == 0) {
    float s = 0.f;
    for (int k = threadIdx.x; k < nslab; k += 256) s += wsb[(size_t)k * CoP + co];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) db[co] = part[0];
  }
}

struct WgPlan { int WP, VR, R, CSZ, CSX, nbands, nblk, MT, CoP, CiP, nslab; size_t lds, ws_floats; };

WgPlan plan_wgrad(int N, int Cin, int Cout, int H, int W) {
  WgPlan p;
  p.WP = W + 1;
  p.VR = N * (H + 1) + 1;
  p.CoP = (Cout + 31) / 32 * 32;
  p.CiP = (Cin + 31) / 32 * 32;
  p.MT = (p.CoP % 64 == 0) ? 2 : 1;
  // band rows: keep LDS <= ~64 KB so two workgroups share a CU
  const int MB = p.MT * 32;
  int R = 1;
  for (int r = 1; r <= 64; ++r) {
    const size_t bytes = ((size_t)MB * (r * p.WP + 3) + 32 * ((size_t)(r + 2) * p.WP + 4)) * 4;
    if (bytes <= 64 * 1024 && (long)r <= p.VR) R = r; else break;
  }
  p.R = R;
  p.CSZ = R * p.WP + 2; if ((p.CSZ & 1) == 0) p.CSZ += 1;
  p.CSX = (R + 2) * p.WP + 3; if ((p.CSX & 1) == 0) p.CSX += 1;
  p.nbands = (p.VR + R - 1) / R;
  p.nblk = p.nbands < 256 ? p.nbands : 256;
  p.nslab = p.nblk * (4 / p.MT);
  p.lds = ((size_t)MB * p.CSZ + 32 * (size_t)p.CSX) * 4;
  p.ws_floats = (size_t)p.nslab * 9 * p.CoP * p.CiP + (size_t)p.nslab * p.CoP;
  return p;
}

}  // namespace

extern "C" size_t fdet_conv3x3_wgrad_ws_bytes(int N, int Cin, int Cout, int H, int W) {
  if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  return plan_wgrad(N, Cin, Cout, H, W).ws_floats * 4;
}

extern "C" int fdet_conv3x3_wgrad(const float* x, const float* dz, float* dW, float* db, void* ws,
                                  size_t ws_bytes, int N, int Cin, int Cout, int H, int W, void* stream) {
  FDET_REQUIRE(x && dz && dW && ws, "conv3x3_wgrad: null pointer");
  FDET_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv3x3_wgrad: bad shape");
  const WgPlan p = plan_wgrad(N, Cin, Cout, H, W);
  FDET_REQUIRE(p.lds <= 160 * 1024, "conv3x3_wgrad: W=%d too wide for the row-band LDS tiling (needs %zu B of LDS)", W, p.lds);
  if (ws_bytes < p.ws_floats * 4)
    return fail(FDET_EWORKSPACE, "conv3x3_wgrad: workspace %zu < %zu bytes", ws_bytes, p.ws_floats * 4);
  WgArgs a;
  a.x = x; a.dz = dz; a.ws = (float*)ws; a.wsb = (float*)ws + (size_t)p.nslab * 9 * p.CoP * p.CiP;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.CoP = p.CoP; a.CiP = p.CiP; a.H = H; a.W = W; a.WP = p.WP; a.R = p.R;
  a.VR = p.VR; a.CSZ = p.CSZ; a.CSX = p.CSX; a.nbands = p.nbands;
  dim3 grid(p.nblk, p.CiP / 32, p.CoP / (p.MT * 32));
  hipStream_t st = (hipStream_t)stream;
  if (p.MT == 2) {
    if (p.lds > 64 * 1024) hipFuncSetAttribute((const void*)k_wgrad3x3<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
    hipLaunchKernelGGL(k_wgrad3x3<2>, grid, dim3(256), p.lds, st, a);
  } else {
    if (p.lds > 64 * 1024) hipFuncSetAttribute((const void*)k_wgrad3x3<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
    hipLaunchKernelGGL(k_wgrad3x3<1>, grid, dim3(256), p.lds, st, a);
  }
  if (int rc = check_launch("fdet_conv3x3_wgrad")) return rc;
  hipLaunchKernelGGL(k_wgrad3x3_reduce, dim3(9, Cout), dim3(256), 0, st, a.ws, a.wsb, p.nslab, Cout, Cin, p.CoP,
                     p.CiP, dW, db);
  return check_launch("fdet_conv3x3_wgrad(reduce)");
}
