// 3x3 stride-1 pad-1 convolutions of the residual blocks as fp32 implicit GEMM on
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD = the fp32 roof).
//
// Data layout trick ("flattened padded rows"): the batch is one tall virtual image -- image n
// occupies virtual rows n*(H+1)+1 .. n*(H+1)+H, row n*(H+1) is a shared zero row -- and every
// row carries ONE zero column (WP = W+1) which is at once the right halo of row r and the left
// halo of row r+1.  An output position q = row*WP + col then reads tap (ky,kx) at LDS offset
// q + ky*WP + kx: a constant per tap, so the 32 lanes of an MFMA B operand are 32 consecutive
// dwords (conflict-free ds_read_b32) for any image width.  Garbage positions (zero column,
// zero rows) are computed and dropped; cost (W+1)/W * (H+1)/H.
//
// GEMM mapping (forward):  D[co][q] += A[co][k] * B[k][q],  k = (ci, tap)
//   A = packed weights  wpk[k][co]     (staged per 8-channel chunk into LDS)
//   B = input band      x[ci][q+off]   (band of R virtual rows + halo rows in LDS)
// MFMA 32x32x2: lanes 0-31 carry k even (channel 2c), lanes 32-63 k odd (channel 2c+1).
// The data-gradient conv is the same kernel on dz with flipped/transposed packed weights.
#include "fdet_common.h"

using namespace fdet;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int CK = 8;   // input channels per LDS chunk

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
  float slope;
};

template <int VW> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<4> { using T = float4; };

template <int MT, int NT, int VW>
__global__ void __launch_bounds__(256, (MT * NT >= 8) ? 2 : ((MT * NT >= 4) ? 3 : 4))
k_conv3x3(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MB = MT * 32;                 // output channels per block
  float* A_lds = reinterpret_cast<float*>(smem);                 // [CK*9][MB]
  float* B_lds = A_lds + CK * 9 * MB;                            // [CK][CS]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int band = blockIdx.x, mb = blockIdx.y;
  const int v0 = band * a.R;
  const int H1 = a.H + 1, WP = a.WP, CS = a.CS;

  for (int t = tid; t < CK * CS; t += 256) B_lds[t] = 0.f;       // halos stay zero for all chunks

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  int tapoff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) tapoff[t] = (t / 3) * WP + (t % 3);

  const int qwave = wid * NT * 32;
  const float* Aw = A_lds + half * 9 * MB + l31;
  const float* Bw = B_lds + half * CS + qwave + l31;

  const int rows = a.R + 2;
  const int wv = a.W / VW;
  const int b_items = CK * rows * wv;
  using VT = typename Vec<VW>::T;

  for (int c0 = 0; c0 < a.Cin; c0 += CK) {
    __syncthreads();                                             // previous chunk consumed (and zero fill done)
    // ---- stage A: rows (c0*9 .. (c0+CK)*9) x MB columns of wpk
    {
      const float* src = a.wpk + (size_t)c0 * 9 * a.CoP + mb * MB;
      for (int t = tid; t < CK * 9 * (MB / 4); t += 256) {
        const int row = t / (MB / 4), c4 = t - row * (MB / 4);
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)row * a.CoP + c4 * 4);
        *reinterpret_cast<float4*>(A_lds + row * MB + c4 * 4) = v;
      }
    }
    // ---- stage B: CK channels x (R+2) virtual rows x W columns
    for (int t = tid; t < b_items; t += 256) {
      const int ci = t / (rows * wv);
      const int rem = t - ci * (rows * wv);
      const int tr = rem / wv, xv = rem - tr * wv;
      const int v = v0 - 1 + tr;
      if (v < 0 || v >= a.VR) continue;
      const int n = v / H1, yy = v - n * H1 - 1;
      if (yy < 0) continue;
      const VT val = *reinterpret_cast<const VT*>(
          a.x + (((size_t)n * a.Cin + c0 + ci) * a.H + yy) * a.W + xv * VW);
      float* dst = B_lds + ci * CS + tr * WP + 1 + xv * VW;
      const float* vs = reinterpret_cast<const float*>(&val);
#pragma unroll
      for (int k = 0; k < VW; ++k) dst[k] = vs[k];
    }
    __syncthreads();
    // ---- MFMA: CK/2 channel pairs x 9 taps
#pragma unroll
    for (int cp = 0; cp < CK / 2; ++cp) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        float av[MT], bv[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = Aw[(cp * 2 * 9 + t) * MB + m * 32];
#pragma unroll
        for (int n = 0; n < NT; ++n) bv[n] = Bw[cp * 2 * CS + tapoff[t] + n * 32];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
      }
    }
  }

  // ---- epilogue
  const int qlimit = a.R * WP;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int q = qwave + n * 32 + l31;
    const int tr = q / WP, ox = q - tr * WP;
    const int v = v0 + tr;
    const int img = v / H1, oy = v - img * H1 - 1;
    const bool ok = (q < qlimit) && (ox < a.W) && (v < a.VR) && (oy >= 0);
    if (!ok) continue;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = mb * MB + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co >= a.Cout) continue;
        const size_t idx = (((size_t)img * a.Cout + co) * a.H + oy) * a.W + ox;
        float z = acc[m][n][r];
        if (!a.dgrad) {
          if (a.bias) z += a.bias[co];
          z = z > 0.f ? z : z * a.slope;
          if (a.y_full) a.y_full[idx] = z;
          if (a.y_out) {
            float e = z;
            if (a.scale) e *= a.scale[(size_t)img * a.Cout + co];
            if (a.skip) e += a.skip[idx];
            a.y_out[idx] = e;
          }
        } else {
          if (a.act) z *= (a.act[idx] > 0.f) ? 1.f : a.slope;
          if (a.skip) z += a.skip[idx];
          a.y_full[idx] = z;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// pack weights
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pack3x3(const float* __restrict__ w, int Cout, int Cin, int CoP, int CiP, float* __restrict__ fwd,
          float* __restrict__ bwd) {
  const int nf = Cin * 9 * CoP, nb = Cout * 9 * CiP;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (fwd && t < nf) {
    const int k = t / CoP, co = t - k * CoP;
    const int ci = k / 9, tap = k - ci * 9;
    fwd[t] = (co < Cout) ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.f;
  }
  if (bwd && t < nb) {
    const int k = t / CiP, ci = t - k * CiP;
    const int co = k / 9, tap = k - co * 9;
    bwd[t] = (ci < Cin) ? w[((size_t)co * Cin + ci) * 9 + (8 - tap)] : 0.f;   // flipped tap
  }
}

template <int MT, int NT>
int launch_conv_vw(const ConvArgs& a, int vw, size_t lds, dim3 grid, hipStream_t st) {
  auto set = [&](const void* f) {
    if (lds > 64 * 1024) hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  };
  if (vw == 4) { set((const void*)k_conv3x3<MT, NT, 4>); hipLaunchKernelGGL((k_conv3x3<MT, NT, 4>), grid, dim3(256), lds, st, a); }
  else if (vw == 2) { set((const void*)k_conv3x3<MT, NT, 2>); hipLaunchKernelGGL((k_conv3x3<MT, NT, 2>), grid, dim3(256), lds, st, a); }
  else { set((const void*)k_conv3x3<MT, NT, 1>); hipLaunchKernelGGL((k_conv3x3<MT, NT, 1>), grid, dim3(256), lds, st, a); }
  return check_launch("fdet_conv3x3");
}

int run_conv(ConvArgs a, hipStream_t st) {
  a.WP = a.W + 1;
  a.VR = a.N * (a.H + 1) + 1;
  a.CoP = (a.Cout + 31) / 32 * 32;
  const int MT = (a.CoP % 64 == 0) ? 2 : 1;
  // choose NT (N tiles per wave) so that the launch has enough workgroups for 256 CUs
  const long total_q = (long)a.VR * a.WP;
  int NT = 4;
  while (NT > 1 && total_q / (4L * NT * 32) < 1024) NT >>= 1;
  const int cap = 4 * NT * 32;
  if (a.WP > cap) return fail(FDET_EINVAL, "conv3x3: W=%d too wide for the band tiling", a.W);
  a.R = cap / a.WP;
  if (a.R > a.VR) a.R = a.VR;
  a.nbands = (a.VR + a.R - 1) / a.R;
  a.CS = cap + 2 * a.WP + 2;
  if ((a.CS & 1) == 0) a.CS += 1;
  const int vw = (a.W % 4 == 0) ? 4 : (a.W % 2 == 0 ? 2 : 1);
  const size_t lds = (size_t)(CK * 9 * MT * 32 + CK * a.CS) * 4;
  dim3 grid(a.nbands, a.CoP / (MT * 32));
  if (MT == 2) {
    if (NT == 4) return launch_conv_vw<2, 4>(a, vw, lds, grid, st);
    if (NT == 2) return launch_conv_vw<2, 2>(a, vw, lds, grid, st);
    return launch_conv_vw<2, 1>(a, vw, lds, grid, st);
  }
  if (NT == 4) return launch_conv_vw<1, 4>(a, vw, lds, grid, st);
  if (NT == 2) return launch_conv_vw<1, 2>(a, vw, lds, grid, st);
  return launch_conv_vw<1, 1>(a, vw, lds, grid, st);
}

}  // namespace

extern "C" int fdet_pack_conv3x3_weights(const float* w, int Cout, int Cin, float* wpk_fwd, float* wpk_bwd,
                                         void* stream) {
  FDET_REQUIRE(w && Cout > 0 && Cin > 0 && (wpk_fwd || wpk_bwd), "pack_conv3x3_weights: bad arguments");
  const int CoP = (Cout + 31) / 32 * 32, CiP = (Cin + 31) / 32 * 32;
  const int n = max(Cin * 9 * CoP, Cout * 9 * CiP);
  hipLaunchKernelGGL(k_pack3x3, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, CoP, CiP,
                     wpk_fwd, wpk_bwd);
  return check_launch("fdet_pack_conv3x3_weights");
}

extern "C" int fdet_conv3x3_fwd(const float* x, const float* wpk, const float* bias, float* y_full,
                                const float* skip, const float* drop_scale, float* y_out, int N, int Cin,
                                int Cout, int H, int W, int pool, float slope, void* stream) {
  FDET_REQUIRE(x && wpk && (y_full || y_out), "conv3x3_fwd: null pointer");
  FDET_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % CK == 0,
               "conv3x3_fwd: unsupported shape N=%d Cin=%d Cout=%d H=%d W=%d (Cin must be a multiple of %d)", N,
               Cin, Cout, H, W, CK);
  FDET_REQUIRE(pool == 1, "conv3x3_fwd: pooled tails go through fdet_block_tail_fwd (pool=%d)", pool);
  ConvArgs a{};
  a.x = x; a.wpk = wpk; a.bias = bias; a.y_full = y_full; a.skip = skip; a.scale = drop_scale; a.y_out = y_out;
  a.act = nullptr; a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.dgrad = 0; a.slope = slope;
  return run_conv(a, (hipStream_t)stream);
}

extern "C" int fdet_conv3x3_dgrad(const float* dz, const float* wpk, const float* act, const float* add,
                                  float* dx, int N, int Cin, int Cout, int H, int W, float slope, void* stream) {
  FDET_REQUIRE(dz && wpk && dx, "conv3x3_dgrad: null pointer");
  FDET_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cout % CK == 0,
               "conv3x3_dgrad: unsupported shape N=%d Cin=%d Cout=%d H=%d W=%d", N, Cin, Cout, H, W);
  ConvArgs a{};
  // the data gradient is a forward conv over dz: "input channels" = Cout, "output channels" = Cin
  a.x = dz; a.wpk = wpk; a.bias = nullptr; a.y_full = dx; a.skip = add; a.scale = nullptr; a.y_out = nullptr;
  a.act = act; a.N = N; a.Cin = Cout; a.Cout = Cin; a.H = H; a.W = W; a.dgrad = 1; a.slope = slope;
  return run_conv(a, (hipStream_t)stream);
}
