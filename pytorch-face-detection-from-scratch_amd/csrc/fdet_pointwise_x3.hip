// Pointwise (1x1) convolutions / per-position Linear layers as the dense GEMMs they are, on the bf16 matrix cores with
// the bf16x3 split (fp32-level accuracy, see fdet_conv3x3_x3.hip).  Replaces, in the SSD path,
//   SeparableResidualBlock.pointwise_conv_skip = nn.Conv2d(Cin, Cout, 1)   (models/SSD.py:24-30)
//   extracting_layers[i] = nn.Linear(C, 5) applied per position            (models/SSD.py:183-185, :228-240)
// which round 1 ran through the 3x3 kernels with centre-tap weights (8 of 9 MFMAs multiplied zeros, halo rows staged for
// nothing).  NCHW fp32 tensors; a position index p runs over the H*W plane of one image, so there is no halo and no row
// structure: Y[n][co][p] = act(sum_ci W[co][ci] X[n][ci][p] + b[co]) (+ add[n][co][p]).
//
//   forward / data gradient  k_pw_x3<MT,VW>: one workgroup = 256 positions of one image x MB = 32*MT output channels.
//       K loop over 16-channel chunks: the chunk's activations go global -> registers -> (hi,lo) split -> LDS slots
//       [k-half][position] of 8 bf16 (16 KB, several workgroups per CU hide the latency: the op is HBM-bound, 4 B in and
//       4 B out per position and channel); the weight fragments come straight from the pre-split panel in L2 (one 16-byte
//       load per lane and fragment, no staging).  The data gradient is the same kernel on the transposed panel.
//   weight gradient  k_pw_wgrad_x3: dW[co][ci] = sum_{n,p} dz[n][co][p] x[n][ci][p] is a GEMM whose K runs over positions,
//       which are CONTIGUOUS in both operands: every lane loads its 8 positions of its row straight into the fragment
//       (no LDS at all), splits, and feeds the MFMA.  Workgroups cut the (n, p) range into slabs; a second kernel adds the
//       slab partials in a fixed order (deterministic), and reduces the bias gradient.
#include "fdet_conv3x3_x3.h"
#include <algorithm>
#include <cstdint>

using namespace fdet;

namespace {

constexpr int PW_THR = 256;
constexpr int PW_POS = 256;          // positions per workgroup

struct PwArgs {
  const float* x;          // [N,Cin,P]
  const bf16x8* a_hi;      // [Cin/16][2][CoP] units of 8 bf16 (hi); lo follows at +units
  const bf16x8* a_lo;
  const float* bias;       // [Cout] or null
  const float* add;        // [N,Cout,P] or null
  float* y;                // [N,Cout,P]
  int N, Cin, Cout, CoP, P, tiles_per_img;
  float slope;             // LeakyReLU slope of the epilogue (1 = identity)
};

// weight panels of a pointwise layer: fwd unit (c16*2 + h)*CoP + co holds W[co][16 c16 + 8h + j]; bwd unit
// (o16*2 + h)*CiP + ci holds W[16 o16 + 8h + j][ci]  (the K-major A operand of Y = W X resp. dX = W^T dZ)
__global__ void __launch_bounds__(256)
k_pack_pw_x3(const float* __restrict__ w, int Cout, int Cin, int CoP, int CiP, int CinP16, int CoutP16,
             bf16x8* __restrict__ fwd, bf16x8* __restrict__ bwd) {
  const int nf = (CinP16 / 16) * 2 * CoP, nb = (CoutP16 / 16) * 2 * CiP;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (fwd && t < nf) {
    const int co = t % CoP, r = t / CoP, h = r & 1, c16 = r >> 1;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = c16 * 16 + 8 * h + j;
      f[j] = (co < Cout && ci < Cin) ? w[(size_t)co * Cin + ci] : 0.f;
    }
    bf16x8 hi, lo;
    split8(f, hi, lo);
    fwd[t] = hi; fwd[nf + t] = lo;
  }
  if (bwd && t < nb) {
    const int ci = t % CiP, r = t / CiP, h = r & 1, o16 = r >> 1;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int co = o16 * 16 + 8 * h + j;
      f[j] = (ci < Cin && co < Cout) ? w[(size_t)co * Cin + ci] : 0.f;
    }
    bf16x8 hi, lo;
    split8(f, hi, lo);
    bwd[t] = hi; bwd[nb + t] = lo;
  }
}

template <int MT, int VW>
__global__ void __launch_bounds__(PW_THR)
k_pw_x3(const PwArgs a) {
  constexpr int NT = 2, MB = MT * 32;
  constexpr int PT = PW_POS + 4;                       // units per LDS array (pad: the two k-halves on different banks)
  __shared__ __attribute__((aligned(16))) bf16x8 lds[4 * PT];   // hi {h0,h1}, lo {h0,h1}
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int img = blockIdx.x / a.tiles_per_img;
  const int p0 = (blockIdx.x - img * a.tiles_per_img) * PW_POS;
  const int cob0 = blockIdx.y * MB;
  const size_t P = (size_t)a.P;
  const float* __restrict__ xi = a.x + (size_t)img * a.Cin * P;
  const int nch = (a.Cin + 15) / 16;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // staging slots: (k-half h, VW consecutive positions); 2 * 256 / VW items over 256 threads
  constexpr int NSL = (2 * PW_POS / VW + PW_THR - 1) / PW_THR;
  using VT = typename Vec<VW>::T;
  const int qwave = wid * 64;
  const int b_off = half * PT + qwave + l31;
  for (int c = 0; c < nch; ++c) {
    VT pb[NSL][8];
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const int it = s * PW_THR + tid;
      const int h = it & 1, pp = (it >> 1) * VW;
      const bool in_p = it < 2 * PW_POS / VW && p0 + pp < a.P;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ch = c * 16 + 8 * h + j;
        VT v{};
        if (in_p && ch < a.Cin) {
          const float* q = xi + (size_t)ch * P + p0 + pp;
          if (VW == 1 || p0 + pp + VW <= a.P) v = *reinterpret_cast<const VT*>(q);
          else { float t_[4] = {0.f, 0.f, 0.f, 0.f}; for (int i = 0; i < VW && p0 + pp + i < a.P; ++i) t_[i] = q[i]; __builtin_memcpy(&v, t_, sizeof(VT)); }
        }
        pb[s][j] = v;
      }
    }
    if (c) __syncthreads();                            // every wave is done reading the previous chunk
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const int it = s * PW_THR + tid;
      if (it < 2 * PW_POS / VW) {
        const int h = it & 1, pp = (it >> 1) * VW;
#pragma unroll
        for (int i = 0; i < VW; ++i) {
          float f[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = vget<VW>(pb[s][j], i);
          bf16x8 hi, lo;
          split8(f, hi, lo);
          lds[h * PT + pp + i] = hi;
          lds[(2 + h) * PT + pp + i] = lo;
        }
      }
    }
    __syncthreads();
    const size_t wbase = (size_t)(c * 2 + half) * a.CoP + cob0 + l31;
    bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { ah[m] = a.a_hi[wbase + m * 32]; al[m] = a.a_lo[wbase + m * 32]; }
#pragma unroll
    for (int n = 0; n < NT; ++n) { bh[n] = lds[b_off + n * 32]; bl[n] = lds[2 * PT + b_off + n * 32]; }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
      }
  }
  // ---- epilogue: lane = position (q = qwave + 32 n + l31), register r = channel cob0 + 32 m + (r&3) + 8 (r>>2) + 4 half.
  // A lane's 32 positions-neighbours cover 128 contiguous bytes of one channel row: dword accesses, coalesced per row.
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int p = p0 + qwave + n * 32 + l31;
    if (p >= a.P) continue;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = cob0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (ch >= a.Cout) continue;
        const size_t idx = ((size_t)img * a.Cout + ch) * P + p;
        float z = acc[m][n][r] + (a.bias ? a.bias[ch] : 0.f);
        z = z > 0.f ? z : z * a.slope;
        if (a.add) z += a.add[idx];
        a.y[idx] = z;
      }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct PwWgArgs {
  const float* x;          // [N,Cin,P]
  const float* dz;         // [N,Cout,P]
  float* ws;               // [nslab][CoT*32][CiT*32] partial dW
  int N, Cin, Cout, P, CoT, CiT, nslab, steps_per_img, steps_total, steps_per_slab, vec_ok;
};

// 8 consecutive positions of row `row` (null: zeros) starting at p, zeros past P; VEC: 16-byte aligned rows
template <bool VEC>
__device__ __forceinline__ void pw_load8(const float* __restrict__ row, int p, int P, float (&f)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = 0.f;
  if (!row || p >= P) return;
  if (VEC && p + 8 <= P) {
    __builtin_memcpy(&f[0], row + p, 16);
    __builtin_memcpy(&f[4], row + p + 4, 16);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) if (p + j < P) f[j] = row[p + j];
  }
}

// one wave = one 32x32 block of dW over a slab of positions; workgroup = 4 waves = 4 consecutive (co-tile, ci-tile) blocks
// (round 4: the bias gradient rides along -- the waves of input-channel tile 0 sum their dz fragments before splitting them,
//  one partial per (slab, output channel); the separate pass over dz, k_pw_bias_part, is gone)
template <bool VEC>
__global__ void __launch_bounds__(256)
k_pw_wgrad_x3(const PwWgArgs a, float* __restrict__ bias_part /*[nslab][CoT*32] or null*/) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int blk = blockIdx.x * 4 + wid;                 // (co tile, ci tile)
  const int slab = blockIdx.y;
  if (blk >= a.CoT * a.CiT) return;
  const int cot = blk / a.CiT, cit = blk - cot * a.CiT;
  const int co = cot * 32 + l31, ci = cit * 32 + l31;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  const bool want_b = bias_part != nullptr && cit == 0;    // wave-uniform
  const int s0 = slab * a.steps_per_slab, s1 = min(s0 + a.steps_per_slab, a.steps_total);
  for (int s = s0; s < s1; ++s) {
    const int n = s / a.steps_per_img, p = (s - n * a.steps_per_img) * 16 + 8 * half;
    float fa[8], fb[8];
    pw_load8<VEC>(co < a.Cout ? a.dz + ((size_t)n * a.Cout + co) * a.P : nullptr, p, a.P, fa);
    pw_load8<VEC>(ci < a.Cin ? a.x + ((size_t)n * a.Cin + ci) * a.P : nullptr, p, a.P, fb);
    if (want_b) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bsum += fa[j];
    }
    bf16x8 ah, al, bh, bl;
    split8(fa, ah, al);
    split8(fb, bh, bl);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  }
  // acc[r]: row (co) = (r&3) + 8 (r>>2) + 4 half, column (ci) = l31
  const int ldw = a.CiT * 32;
  float* __restrict__ w = a.ws + ((size_t)slab * a.CoT * 32 + cot * 32) * ldw + cit * 32 + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) w[(size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * ldw] = acc[r];
  if (want_b) {
    bsum += __shfl_xor(bsum, 32, 64);                      // the two k halves of a channel
    if (half == 0) bias_part[(size_t)slab * a.CoT * 32 + cot * 32 + l31] = bsum;
  }
}

// db[c] = sum over the slabs of bias_part[slab][c]: one workgroup per channel, thread t adds the slabs t, t + 256, ...; the 256
// partial sums are combined by a fixed tree in LDS (the same association on every run)
__global__ void __launch_bounds__(256)
k_pw_bias_slabs(const float* __restrict__ part, int nslab, int ld, int C, float* __restrict__ db) {
  __shared__ float sh[256];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int b = threadIdx.x; b < nslab; b += 256) s += part[(size_t)b * ld + c];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) db[c] = sh[0];
}

// dW[co][ci] = sum over the slabs, in a fixed order: 256 threads = 16 consecutive outputs x 16 slab lanes; lane g sums the
// slabs g, g+16, ... (four independent partial sums: the loads of an output are in flight together instead of one
// dependent chain of `nslab` loads per thread), then the 16 lanes are combined in lane order
__global__ void __launch_bounds__(256)
k_pw_wgrad_reduce(const float* __restrict__ ws, int nslab, int CoP32, int CiP32, int Cout, int Cin, float* __restrict__ dW) {
  __shared__ float part[16][17];
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int t = blockIdx.x * 16 + o;
  float s = 0.f;
  if (t < Cout * Cin) {
    const int co = t / Cin, ci = t - co * Cin;
    const float* __restrict__ p = ws + (size_t)co * CiP32 + ci;
    const size_t st = (size_t)CoP32 * CiP32;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = g;
    for (; b + 48 < nslab; b += 64) {
      s0 += p[(size_t)b * st]; s1 += p[(size_t)(b + 16) * st]; s2 += p[(size_t)(b + 32) * st]; s3 += p[(size_t)(b + 48) * st];
    }
    for (; b < nslab; b += 16) s0 += p[(size_t)b * st];
    s = (s0 + s1) + (s2 + s3);
  }
  part[g][o] = s;
  __syncthreads();
  if (g == 0 && t < Cout * Cin) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) r += part[k][o];
    dW[t] = r;
  }
}

int pw_num_cus() {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return ncu;
}

int run_pw(const float* x, const void* wpk, const float* bias, const float* add, float* y, int N, int Cin, int Cout, int P,
           float slope, hipStream_t st, const char* what) {
  PwArgs a;
  a.x = x; a.bias = bias; a.add = add; a.y = y;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.P = P; a.slope = slope;
  a.CoP = (Cout + 31) / 32 * 32;
  a.tiles_per_img = (P + PW_POS - 1) / PW_POS;
  const size_t units = (size_t)((Cin + 15) / 16) * 2 * a.CoP;
  a.a_hi = reinterpret_cast<const bf16x8*>(wpk);
  a.a_lo = a.a_hi + units;
  const int MT = (a.CoP % 64 == 0) ? 2 : 1;
  dim3 grid((unsigned)((size_t)N * a.tiles_per_img), a.CoP / (32 * MT));
  const bool v4 = (P % 4 == 0) && ((uintptr_t)x % 16 == 0);
  if (MT == 2) { if (v4) hipLaunchKernelGGL((k_pw_x3<2, 4>), grid, dim3(PW_THR), 0, st, a); else hipLaunchKernelGGL((k_pw_x3<2, 1>), grid, dim3(PW_THR), 0, st, a); }
  else { if (v4) hipLaunchKernelGGL((k_pw_x3<1, 4>), grid, dim3(PW_THR), 0, st, a); else hipLaunchKernelGGL((k_pw_x3<1, 1>), grid, dim3(PW_THR), 0, st, a); }
  return check_launch(what);
}

}  // namespace

extern "C" size_t fdet_pointwise_packed_bytes(int Cout, int Cin) {
  // the larger of the forward and the backward panel (hi + lo, 16-byte units), so one size serves both
  const size_t f = (size_t)((Cin + 15) / 16) * 2 * ((Cout + 31) / 32 * 32), b = (size_t)((Cout + 15) / 16) * 2 * ((Cin + 31) / 32 * 32);
  return std::max(f, b) * 2 * 16;
}

extern "C" int fdet_pack_pointwise_weights_bf16x3(const float* w, int Cout, int Cin, void* wpk_fwd, void* wpk_bwd, void* stream) {
  FDET_REQUIRE(w && Cout > 0 && Cin > 0 && (wpk_fwd || wpk_bwd), "pack_pointwise_weights_bf16x3: bad arguments");
  const int CoP = (Cout + 31) / 32 * 32, CiP = (Cin + 31) / 32 * 32, CinP16 = (Cin + 15) / 16 * 16, CoutP16 = (Cout + 15) / 16 * 16;
  const int n = std::max((CinP16 / 16) * 2 * CoP, (CoutP16 / 16) * 2 * CiP);
  hipLaunchKernelGGL(k_pack_pw_x3, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, CoP, CiP, CinP16,
                     CoutP16, (bf16x8*)wpk_fwd, (bf16x8*)wpk_bwd);
  return check_launch("fdet_pack_pointwise_weights_bf16x3");
}

extern "C" int fdet_pointwise_fwd_bf16x3(const float* x, const void* wpk_fwd, const float* bias, float* y, int N, int Cin,
                                         int Cout, int P, float slope, void* stream) {
  FDET_REQUIRE(x && wpk_fwd && y && N > 0 && Cin > 0 && Cout > 0 && P > 0, "pointwise_fwd_bf16x3: bad arguments");
  FDET_REQUIRE((size_t)N * std::max(Cin, Cout) * P < ((size_t)1 << 40), "pointwise_fwd_bf16x3: tensor too large");
  return run_pw(x, wpk_fwd, bias, nullptr, y, N, Cin, Cout, P, slope, (hipStream_t)stream, "fdet_pointwise_fwd_bf16x3");
}

extern "C" int fdet_pointwise_dgrad_bf16x3(const float* dz, const void* wpk_bwd, const float* add, float* dx, int N, int Cin,
                                           int Cout, int P, void* stream) {
  FDET_REQUIRE(dz && wpk_bwd && dx && N > 0 && Cin > 0 && Cout > 0 && P > 0, "pointwise_dgrad_bf16x3: bad arguments");
  // dX = W^T dZ: the same GEMM with the roles of the channel counts exchanged
  return run_pw(dz, wpk_bwd, nullptr, add, dx, N, Cout, Cin, P, 1.0f, (hipStream_t)stream, "fdet_pointwise_dgrad_bf16x3");
}

extern "C" size_t fdet_pointwise_wgrad_ws_bytes(int N, int Cin, int Cout, int P) {
  if (N <= 0 || Cin <= 0 || Cout <= 0 || P <= 0) return 0;
  const int CoT = (Cout + 31) / 32, CiT = (Cin + 31) / 32;
  const long steps = (long)N * ((P + 15) / 16);
  const long blocks = ((long)CoT * CiT + 3) / 4;
  long nslab = std::max<long>(1, std::min<long>(steps, (8L * 256 + blocks - 1) / blocks));   // ~8 workgroups per CU
  return ((size_t)nslab * CoT * 32 * CiT * 32 + (size_t)nslab * CoT * 32) * sizeof(float);
}

extern "C" int fdet_pointwise_wgrad_bf16x3(const float* x, const float* dz, float* dW, float* db, void* ws, size_t ws_bytes,
                                           int N, int Cin, int Cout, int P, void* stream) {
  FDET_REQUIRE(x && dz && dW && ws && N > 0 && Cin > 0 && Cout > 0 && P > 0, "pointwise_wgrad_bf16x3: bad arguments");
  FDET_REQUIRE(ws_bytes >= fdet_pointwise_wgrad_ws_bytes(N, Cin, Cout, P), "pointwise_wgrad_bf16x3: workspace too small (%zu bytes)", ws_bytes);
  PwWgArgs a;
  a.x = x; a.dz = dz; a.ws = (float*)ws; a.N = N; a.Cin = Cin; a.Cout = Cout; a.P = P;
  a.CoT = (Cout + 31) / 32; a.CiT = (Cin + 31) / 32;
  a.steps_per_img = (P + 15) / 16;
  a.steps_total = N * a.steps_per_img;
  const long blocks = ((long)a.CoT * a.CiT + 3) / 4;
  a.nslab = (int)std::max<long>(1, std::min<long>(a.steps_total, (8L * 256 + blocks - 1) / blocks));
  a.steps_per_slab = (a.steps_total + a.nslab - 1) / a.nslab;
  a.nslab = (a.steps_total + a.steps_per_slab - 1) / a.steps_per_slab;
  const bool vec = (P % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)dz % 16 == 0);
  a.vec_ok = vec;
  dim3 grid((unsigned)blocks, (unsigned)a.nslab);
  float* bpart = db ? (float*)ws + (size_t)a.nslab * a.CoT * 32 * a.CiT * 32 : nullptr;     // [nslab][CoT*32]
  if (vec) hipLaunchKernelGGL(k_pw_wgrad_x3<true>, grid, dim3(256), 0, (hipStream_t)stream, a, bpart);
  else hipLaunchKernelGGL(k_pw_wgrad_x3<false>, grid, dim3(256), 0, (hipStream_t)stream, a, bpart);
  hipLaunchKernelGGL(k_pw_wgrad_reduce, dim3((Cout * Cin + 15) / 16), dim3(256), 0, (hipStream_t)stream, (const float*)ws,
                     a.nslab, a.CoT * 32, a.CiT * 32, Cout, Cin, dW);
  if (db) hipLaunchKernelGGL(k_pw_bias_slabs, dim3(Cout), dim3(256), 0, (hipStream_t)stream, (const float*)bpart, a.nslab, a.CoT * 32, Cout, db);
  return check_launch("fdet_pointwise_wgrad_bf16x3");
}
