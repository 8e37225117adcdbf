// Weight/bias gradient of the 3x3 convs as fp32 implicit GEMM on v_mfma_f32_32x32x2_f32.
//
//   dW[co][ci][tap] = sum_q dz[co][q] * x[ci][q + off(tap)]       (q over all positions)
// GEMM: M = co (A = dz, one LDS row per channel), N = 32 input channels at ONE tap (B = x,
// one LDS row per channel, constant tap offset), K = positions, two per MFMA (lanes 0-31 take
// q, lanes 32-63 take q+1).  Same flattened-padded-row layout as fdet_conv3x3.hip; garbage
// positions contribute nothing because the dz tile is zero there.
//
// Workgroup = 8 waves.  Wave roles: (co tile m, ci group cg, K split kh); every wave keeps its
// 9 tap tiles (144 accumulator VGPRs) for the whole kernel while the workgroup walks bands of
// R virtual rows grid-stride.  The next band is prefetched global->registers during the MFMAs.
// At the end each (workgroup, kh) writes one slab; a second kernel sums the slabs in fixed
// order (deterministic, no float atomics).
#include "fdet_common.h"
#include <algorithm>

using namespace fdet;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NTHR = 512;
// prefetch register budget (dz tile, x tile); scalar-load layouts (odd W) get less, their
// per-load addressing costs more registers
__host__ __device__ constexpr int zregs(int vw) { return vw == 1 ? 8 : 16; }
__host__ __device__ constexpr int xregs(int vw) { return vw == 1 ? 16 : 32; }

struct WgArgs {
  const float* x;    // [N,Cin,H,W]
  const float* dz;   // [N,Cout,H,W]
  float* ws;         // [nslab][9][CoP][CiP]
  float* wsb;        // [nslab][CoP]
  int N, Cin, Cout, CoP, CiP, H, W, WP, R, VR, CSZ, CSX, nbands;
  unsigned magic_h1;
};

template <int VW> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = f32x2; };
template <> struct Vec<4> { using T = f32x4; };
template <int VW> __device__ __forceinline__ float vget(const typename Vec<VW>::T& v, int k) { return v[k]; }
template <> __device__ __forceinline__ float vget<1>(const float& v, int) { return v; }
template <int VW> __device__ __forceinline__ typename Vec<VW>::T vzero() { typename Vec<VW>::T z = {}; return z; }
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }

// floor(v/d) for 0 <= v < 2^20, d < 2^12 with magic = ceil(2^32/d); magic == 0 encodes d == 1
__device__ __forceinline__ int fdiv(int v, unsigned magic) { return magic ? (int)__umulhi((unsigned)v, magic) : v; }

// MTC co tiles x CG ci groups handled per workgroup; KS = 8/(MTC*CG) K splits.
// Staging: 16 lanes per tile row (W/VW <= 16); a thread owns fixed (channel, column-vector)
// slots and walks the band's rows, so all row arithmetic (virtual row -> image, y) is
// wave-uniform scalar work and the per-lane part is one add.
template <int MTC, int CG, int VW>
__global__ void __launch_bounds__(NTHR, 2)
k_wgrad3x3(const WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = 8 / (MTC * CG);
  constexpr int LPR = 16;                             // lanes per staged row
  constexpr int ZCH = MTC * 32 * LPR / NTHR;          // dz channel slots per thread (1 or 2)
  constexpr int XCH = CG * 32 * LPR / NTHR;           // x channel slots per thread
  constexpr int RZ = zregs(VW) / (ZCH * VW);              // max band rows held in registers
  constexpr int RX = xregs(VW) / (XCH * VW);
  using VT = typename Vec<VW>::T;
  float* Z = reinterpret_cast<float*>(smem);    // [MTC*32][CSZ]
  float* X = Z + MTC * 32 * a.CSZ;              // [CG*32][CSX]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int cib = blockIdx.y, cob = blockIdx.z;
  const int m = wid % MTC, cg = (wid / MTC) % CG, kh = wid / (MTC * CG);
  const int H1 = a.H + 1, WP = a.WP, W = a.W, R = a.R;
  const int Q = R * WP;
  int Lk = (Q + KS - 1) / KS; Lk = (Lk + 1) & ~1;
  const int qs = kh * Lk, qe = min(Q, qs + Lk);

  for (int t = tid; t < MTC * 32 * a.CSZ + CG * 32 * a.CSX; t += NTHR) Z[t] = 0.f;   // pads stay zero

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
  const float* xr = X + (cg * 32 + l31) * a.CSX + half;

  // staging geometry
  const int wv = W / VW;
  const int xv = tid & (LPR - 1);
  const int chl = tid >> 4;                           // 0..31: channel within a 32-channel slot
  const bool lane_ok = xv < wv;
  const int co0 = cob * MTC * 32, ci0 = cib * CG * 32;
  const int HW = a.H * W;

  VT pz[ZCH][RZ], px[XCH][RX];
#define FDET_WG_LOAD(V0)                                                                          \
  {                                                                                               \
    _Pragma("unroll") for (int r_ = 0; r_ < RZ; ++r_) {                                           \
      const int v = (V0) + r_;                          /* wave-uniform */                        \
      const int n = fdiv(v, a.magic_h1), yy = v - n * H1 - 1;                                     \
      const bool rok = r_ < R && v < a.VR && yy >= 0;                                             \
      const float* rowp = a.dz + ((size_t)n * a.Cout * a.H + yy) * W;                             \
      _Pragma("unroll") for (int c_ = 0; c_ < ZCH; ++c_) {                                        \
        const int ch = co0 + c_ * 32 + chl;                                                       \
        pz[c_][r_] = (rok && lane_ok && ch < a.Cout)                                              \
                         ? *reinterpret_cast<const VT*>(rowp + (size_t)ch * HW + xv * VW) : vzero<VW>(); \
      }                                                                                           \
    }                                                                                             \
    _Pragma("unroll") for (int r_ = 0; r_ < RX; ++r_) {                                           \
      const int v = (V0) - 1 + r_;                                                                \
      const int n = fdiv(max(v, 0), a.magic_h1), yy = v - n * H1 - 1;                             \
      const bool rok = r_ < R + 2 && v >= 0 && v < a.VR && yy >= 0;                               \
      const float* rowp = a.x + ((size_t)n * a.Cin * a.H + yy) * W;                               \
      _Pragma("unroll") for (int c_ = 0; c_ < XCH; ++c_) {                                        \
        const int ch = ci0 + c_ * 32 + chl;                                                       \
        px[c_][r_] = (rok && lane_ok && ch < a.Cin)                                               \
                         ? *reinterpret_cast<const VT*>(rowp + (size_t)ch * HW + xv * VW) : vzero<VW>(); \
      }                                                                                           \
    }                                                                                             \
  }
#define FDET_WG_STORE()                                                                           \
  {                                                                                               \
    if (lane_ok) {                                                                                \
      _Pragma("unroll") for (int r_ = 0; r_ < RZ; ++r_) {                                         \
        if (r_ < R) {                                                                             \
          _Pragma("unroll") for (int c_ = 0; c_ < ZCH; ++c_) {                                    \
            float* d_ = Z + (c_ * 32 + chl) * a.CSZ + r_ * WP + xv * VW;                          \
            _Pragma("unroll") for (int k_ = 0; k_ < VW; ++k_) d_[k_] = vget<VW>(pz[c_][r_], k_);  \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
      _Pragma("unroll") for (int r_ = 0; r_ < RX; ++r_) {                                         \
        if (r_ < R + 2) {                                                                         \
          _Pragma("unroll") for (int c_ = 0; c_ < XCH; ++c_) {                                    \
            float* d_ = X + (c_ * 32 + chl) * a.CSX + r_ * WP + 1 + xv * VW;                      \
            _Pragma("unroll") for (int k_ = 0; k_ < VW; ++k_) d_[k_] = vget<VW>(px[c_][r_], k_);  \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
    }                                                                                             \
  }

  int band = blockIdx.x;
  if (band < a.nbands) FDET_WG_LOAD(band * R)
  for (; band < a.nbands; band += gridDim.x) {
    __syncthreads();                 // previous band's MFMAs done (first pass: zero fill done)
    FDET_WG_STORE()
    __syncthreads();
    const int nb = band + gridDim.x;
    if (nb < a.nbands) FDET_WG_LOAD(nb * R)
    // software-pipelined: operands of position pair q+2 are fetched before the MFMAs of q
    if (qs < qe) {
      float av0 = zr[qs], bv0[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) bv0[t] = xr[qs + tapoff[t]];
      for (int q = qs; q < qe; q += 4) {
        float av1 = 0.f, bv1[9];
        const int q1 = q + 2;                       // reads past qe stay inside the (zero padded) tiles
        av1 = zr[q1];
#pragma unroll
        for (int t = 0; t < 9; ++t) bv1[t] = xr[q1 + tapoff[t]];
        __builtin_amdgcn_sched_barrier(0);
        bsum += av0;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv0[t], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (q1 < qe) {
          const int q2 = q + 4;
          av0 = zr[q2];
#pragma unroll
          for (int t = 0; t < 9; ++t) bv0[t] = xr[q2 + tapoff[t]];
          __builtin_amdgcn_sched_barrier(0);
          bsum += av1;
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv1[t], acc[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }

  // ---- combine the K splits of this workgroup through LDS (staging tiles are dead now), then
  //      the kh == 0 waves write ONE slab per workgroup: ws[blk][tap][co][ci], wsb[blk][co]
  bsum += __shfl_xor(bsum, 32, 64);
  {
    float* red = reinterpret_cast<float*>(smem);          // [MTC*CG][9*16][64] + [MTC*CG][64]
    const int role = wid % (MTC * CG);
#pragma unroll 1
    for (int rnd = 1; rnd < KS; ++rnd) {
      __syncthreads();
      if (kh == rnd) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(role * 144 + t * 16 + r) * 64 + lane] = acc[t][r];
        red[MTC * CG * 144 * 64 + role * 64 + lane] = bsum;
      }
      __syncthreads();
      if (kh == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] += red[(role * 144 + t * 16 + r) * 64 + lane];
        bsum += red[MTC * CG * 144 * 64 + role * 64 + lane];
      }
    }
  }
  if (kh == 0) {
    const int s = blockIdx.x;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int ci = ci0 + cg * 32 + l31;
        a.ws[(((size_t)s * 9 + t) * a.CoP + co) * a.CiP + ci] = acc[t][r];
      }
    }
    if (cib == 0 && cg == 0 && half == 0) a.wsb[(size_t)s * a.CoP + co0 + m * 32 + l31] = bsum;
  }
}

struct WgArgsG {
  const float* x;    // [N,Cin,H,W]
  const float* dz;   // [N,Cout,H,W]
  float* ws;         // [nslab][9][CoP][CiP]
  float* wsb;        // [nslab][CoP]
  int N, Cin, Cout, CoP, CiP, H, W, WP, R, VR, CSZ, CSX, nbands;
};

// Generic fallback (any width that fits LDS): 4 waves, scalar staging, no prefetch.
template <int MT>
__global__ void __launch_bounds__(256, 2)
k_wgrad3x3_generic(const WgArgsG a) {
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
    // software-pipelined: operands of position pair q+2 are fetched before the MFMAs of q
    if (qs < qe) {
      float av0 = zr[qs], bv0[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) bv0[t] = xr[qs + tapoff[t]];
      for (int q = qs; q < qe; q += 4) {
        float av1 = 0.f, bv1[9];
        const int q1 = q + 2;                       // reads past qe stay inside the (zero padded) tiles
        av1 = zr[q1];
#pragma unroll
        for (int t = 0; t < 9; ++t) bv1[t] = xr[q1 + tapoff[t]];
        __builtin_amdgcn_sched_barrier(0);
        bsum += av0;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv0[t], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (q1 < qe) {
          const int q2 = q + 4;
          av0 = zr[q2];
#pragma unroll
          for (int t = 0; t < 9; ++t) bv0[t] = xr[q2 + tapoff[t]];
          __builtin_amdgcn_sched_barrier(0);
          bsum += av1;
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv1[t], acc[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
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


// Fixed-order reduction of the slabs.  One workgroup (1024 threads) per (tap, 4 output
// channels): 256 consecutive floats of every slab, four slab phases, LDS combine in phase order.
__global__ void __launch_bounds__(1024)
k_wgrad3x3_reduce(const float* __restrict__ ws, const float* __restrict__ wsb, int nslab, int Cout, int Cin,
                  int CoP, int CiP, float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float part[1024];
  const int tap = blockIdx.x;
  const int e = threadIdx.x & 255, ph = threadIdx.x >> 8;
  const int span = 4 * CiP;                               // floats per (slab, tap, 4 channels)
  const int co0 = blockIdx.y * 4;
  for (int e0 = 0; e0 < span; e0 += 256) {
    const int idx = e0 + e;
    float s = 0.f;
    if (idx < span && co0 + idx / CiP < CoP) {
      const float* src = ws + ((size_t)tap * CoP + co0) * CiP + idx;
#pragma unroll 4
      for (int k = ph; k < nslab; k += 4) s += src[(size_t)k * 9 * CoP * CiP];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (ph == 0 && idx < span) {
      const int co = co0 + idx / CiP, ci = idx % CiP;
      if (co < Cout && ci < Cin)
        dW[((size_t)co * Cin + ci) * 9 + tap] = ((part[e] + part[256 + e]) + part[512 + e]) + part[768 + e];
    }
    __syncthreads();
  }
  if (db && tap == 0 && threadIdx.x < 4 * 64) {
    // 4 channels x 64 slab lanes
    const int c = threadIdx.x >> 6, ln = threadIdx.x & 63;
    float s = 0.f;
    if (co0 + c < Cout)
      for (int k = ln; k < nslab; k += 64) s += wsb[(size_t)k * CoP + co0 + c];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (ln == 0 && co0 + c < Cout) db[co0 + c] = s;
  }
}

unsigned magic_of(int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

struct WgPlan { int WP, VR, R, CSZ, CSX, nbands, nblk, MTC, CG, KS, CoP, CiP, nslab, vw; size_t lds, ws_floats; bool ok; };

// fast path: 16 lanes per staged row, band rows limited by the prefetch register budget
WgPlan plan_wgrad(int N, int Cin, int Cout, int H, int W) {
  WgPlan p{};
  p.WP = W + 1;
  p.VR = N * (H + 1) + 1;
  p.CoP = (Cout + 31) / 32 * 32;
  p.CiP = (Cin + 31) / 32 * 32;
  p.MTC = (p.CoP % 64 == 0) ? 2 : 1;
  p.CG = (p.CiP % 64 == 0) ? 2 : 1;
  p.KS = 8 / (p.MTC * p.CG);
  p.vw = (W % 4 == 0) ? 4 : (W % 2 == 0 ? 2 : 1);
  const int wv = W / p.vw;
  p.ok = wv <= 16 && p.VR < (1 << 20);
  const int rows_total = p.VR - 1;
  const int rz = zregs(p.vw) / (p.MTC * p.vw), rx = xregs(p.vw) / (p.CG * p.vw);
  int bestR = 0; double bestC = 1e30;
  for (int r = 1; r <= rz && r + 2 <= rx && r <= rows_total; ++r) {
    const size_t bytes = ((size_t)p.MTC * 32 * (r * p.WP + 9) + (size_t)p.CG * 32 * ((r + 2) * p.WP + 11)) * 4;
    if (bytes > 150 * 1024) break;
    const long nb = (rows_total + r - 1) / r;
    const long blk = nb < 256 ? nb : 256;
    const double mfma = (double)(((nb + blk - 1) / blk) * blk * r) / rows_total;   // MFMA time incl. tail waste
    const double cost = mfma * (0.8 + 0.2 * (double)(r + 2) / r);                    // + staging share (halo rows)
    if (cost <= bestC * 1.0001) { bestC = cost; bestR = r; }
  }
  if (bestR == 0) { p.ok = false; bestR = 1; }
  p.R = bestR;
  p.CSZ = p.R * p.WP + 8; if ((p.CSZ & 1) == 0) p.CSZ += 1;
  p.CSX = (p.R + 2) * p.WP + 10; if ((p.CSX & 1) == 0) p.CSX += 1;
  p.nbands = (rows_total + p.R - 1) / p.R;
  p.nblk = p.nbands < 256 ? p.nbands : 256;
  p.nslab = p.nblk;                                   // K splits are combined inside the workgroup
  p.lds = ((size_t)p.MTC * 32 * p.CSZ + (size_t)p.CG * 32 * p.CSX) * 4;
  if (p.KS > 1) p.lds = std::max(p.lds, (size_t)p.MTC * p.CG * (144 + 1) * 64 * 4);
  p.ws_floats = (size_t)p.nslab * 9 * p.CoP * p.CiP + (size_t)p.nslab * p.CoP;
  return p;
}

struct WgPlanG { int WP, VR, R, CSZ, CSX, nbands, nblk, MT, CoP, CiP, nslab; size_t lds, ws_floats; };

WgPlanG plan_wgrad_generic(int N, int Cin, int Cout, int H, int W) {
  WgPlanG p;
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


template <int MTC, int CG>
int launch_wg(const WgArgs& a, const WgPlan& p, dim3 grid, hipStream_t st) {
  int rc = FDET_OK;
  auto go = [&](auto kern) {
    if (p.lds > 64 * 1024) rc = set_lds_attr((const void*)kern, p.lds, "conv3x3_wgrad");
    if (rc == FDET_OK) hipLaunchKernelGGL(kern, grid, dim3(NTHR), p.lds, st, a);
  };
  if (p.vw == 4) go(k_wgrad3x3<MTC, CG, 4>);
  else if (p.vw == 2) go(k_wgrad3x3<MTC, CG, 2>);
  else go(k_wgrad3x3<MTC, CG, 1>);
  return rc;
}

}  // namespace

extern "C" size_t fdet_conv3x3_wgrad_ws_bytes(int N, int Cin, int Cout, int H, int W) {
  if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  const WgPlan p = plan_wgrad(N, Cin, Cout, H, W);
  if (p.ok) return p.ws_floats * 4;
  return plan_wgrad_generic(N, Cin, Cout, H, W).ws_floats * 4;
}

extern "C" int fdet_conv3x3_wgrad(const float* x, const float* dz, float* dW, float* db, void* ws,
                                  size_t ws_bytes, int N, int Cin, int Cout, int H, int W, void* stream) {
  FDET_REQUIRE(x && dz && dW && ws, "conv3x3_wgrad: null pointer");
  FDET_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv3x3_wgrad: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const WgPlan p = plan_wgrad(N, Cin, Cout, H, W);
  if (p.ok) {
    if (ws_bytes < p.ws_floats * 4)
      return fail(FDET_EWORKSPACE, "conv3x3_wgrad: workspace %zu < %zu bytes", ws_bytes, p.ws_floats * 4);
    WgArgs a;
    a.x = x; a.dz = dz; a.ws = (float*)ws; a.wsb = (float*)ws + (size_t)p.nslab * 9 * p.CoP * p.CiP;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.CoP = p.CoP; a.CiP = p.CiP; a.H = H; a.W = W; a.WP = p.WP; a.R = p.R;
    a.VR = p.VR; a.CSZ = p.CSZ; a.CSX = p.CSX; a.nbands = p.nbands;
    a.magic_h1 = magic_of(H + 1);
    dim3 grid(p.nblk, p.CiP / (p.CG * 32), p.CoP / (p.MTC * 32));
    int lrc;
    if (p.MTC == 2 && p.CG == 2) lrc = launch_wg<2, 2>(a, p, grid, st);
    else if (p.MTC == 2) lrc = launch_wg<2, 1>(a, p, grid, st);
    else if (p.CG == 2) lrc = launch_wg<1, 2>(a, p, grid, st);
    else lrc = launch_wg<1, 1>(a, p, grid, st);
    if (lrc != FDET_OK) return lrc;
    if (int rc = check_launch("fdet_conv3x3_wgrad")) return rc;
    hipLaunchKernelGGL(k_wgrad3x3_reduce, dim3(9, (Cout + 3) / 4), dim3(1024), 0, st, a.ws, a.wsb, p.nslab, Cout, Cin, p.CoP,
                       p.CiP, dW, db);
    return check_launch("fdet_conv3x3_wgrad(reduce)");
  }
  // generic fallback for wide rows
  const WgPlanG g = plan_wgrad_generic(N, Cin, Cout, H, W);
  FDET_REQUIRE(g.lds <= 160 * 1024, "conv3x3_wgrad: W=%d too wide for the row-band LDS tiling (needs %zu B of LDS)", W, g.lds);
  if (ws_bytes < g.ws_floats * 4)
    return fail(FDET_EWORKSPACE, "conv3x3_wgrad: workspace %zu < %zu bytes", ws_bytes, g.ws_floats * 4);
  WgArgsG a;
  a.x = x; a.dz = dz; a.ws = (float*)ws; a.wsb = (float*)ws + (size_t)g.nslab * 9 * g.CoP * g.CiP;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.CoP = g.CoP; a.CiP = g.CiP; a.H = H; a.W = W; a.WP = g.WP; a.R = g.R;
  a.VR = g.VR; a.CSZ = g.CSZ; a.CSX = g.CSX; a.nbands = g.nbands;
  dim3 grid(g.nblk, g.CiP / 32, g.CoP / (g.MT * 32));
  if (g.MT == 2) {
    if (g.lds > 64 * 1024) { if (int rc_ = set_lds_attr((const void*)k_wgrad3x3_generic<2>, (size_t)(g.lds), __func__)) return rc_; }
    hipLaunchKernelGGL(k_wgrad3x3_generic<2>, grid, dim3(256), g.lds, st, a);
  } else {
    if (g.lds > 64 * 1024) { if (int rc_ = set_lds_attr((const void*)k_wgrad3x3_generic<1>, (size_t)(g.lds), __func__)) return rc_; }
    hipLaunchKernelGGL(k_wgrad3x3_generic<1>, grid, dim3(256), g.lds, st, a);
  }
  if (int rc = check_launch("fdet_conv3x3_wgrad(generic)")) return rc;
  hipLaunchKernelGGL(k_wgrad3x3_reduce, dim3(9, (Cout + 3) / 4), dim3(1024), 0, st, a.ws, a.wsb, g.nslab, Cout, Cin, g.CoP,
                     g.CiP, dW, db);
  return check_launch("fdet_conv3x3_wgrad(reduce)");
}
