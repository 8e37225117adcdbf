// Resnet stem (models/Resnet.py:64-70: Conv2d(3, F, 3, stride 2, padding 1), no activation) on the matrix cores, round 4:
//
//   k_stem3_wgrad_x3   dW[c][ci][ky][kx] = sum_{n,oy,ox} dz[n][c][oy][ox] * x[n][ci][2oy+ky-1][2ox+kx-1],  db[c] = sum dz
//                      as a [64 channels] x [27 taps + a ones column] GEMM whose K index is 16 consecutive output columns
//                      of one row (bf16x3: a_hi*b_lo + a_lo*b_hi + a_hi*b_hi, fp32 accumulate -- the arithmetic of every
//                      other weight gradient of the stack).  A fragment = 8 consecutive dz floats of one channel straight
//                      from HBM (fp32 NCHW is K-contiguous per channel), B fragment = 8 stride-2 floats of an input row
//                      staged in LDS; both are split in registers.  HBM-paced: 1.0 GB per launch at 640^2 bs 32.
//   k_stem3_fwd_ps     the forward with a pre-split (PS, fdet_ps.h) output written directly in the column-strip layout of the
//                      first block (halo slots included): fp32 VALU, lane = output column, weights wave-uniform.
//
// The scalar-fed VALU weight gradient this replaces (k_stem_wgrad_k3, fdet_stem.hip) took 0.79 ms at 640^2 bs 32; it remains
// for shapes this kernel does not cover (Wo % 16 != 0, Wo > 320) and for the exact-fp32 path.
#include "fdet_conv3x3_x3.h"
#include "fdet_ps.h"
#include <algorithm>

using namespace fdet;

namespace {

constexpr int S3_KPW = 5;                  // k-steps (16 output columns) per wave and row: Wo <= 4 * 5 * 16 = 320

struct Stem3WgArgs {
  const float* x;          // [N][3][H][W]
  const float* dz;         // [N][F][Ho][Wo]
  float* ws;               // [grid][64][32] slabs
  int N, F, H, W, Ho, Wo, nrows, pitch, ksteps;
};

__device__ __forceinline__ void s3_split8(const float (&f)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)f[j];
    hi[j] = h;
    lo[j] = (__bf16)(f[j] - (float)h);
  }
}

// One workgroup (4 waves) walks output rows; per row the nine input rows (3 channels x 3 ky) sit in LDS as
// [row][4 zero floats | W floats], pitch == 4 (mod 64) floats so that the taps of a fragment read fall on distinct banks.
// blockIdx.y = 64-channel block of F.
__global__ void __launch_bounds__(256)
k_stem3_wgrad_x3(const Stem3WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* X = reinterpret_cast<float*>(smem);            // [9][pitch]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int fb = blockIdx.y;
  // B operand of this lane: tap t = l31 (ci, ky, kx), t == 27: the ones column (bias), t > 27: zeros
  const int t = l31;
  const int trow = t < 27 ? t / 3 : 0, tkx = t < 27 ? t % 3 : 0;
  const float* xl = X + trow * a.pitch + tkx + 3 + 16 * half;        // + 2 * (16 * ks + j): ix + 4 = 2 ox + kx + 3
  // A operand: channel 32 m + l31 of this 64-block, columns 16 ks + 8 half + 0..7
  const int c0 = fb * 64 + l31, c1 = c0 + 32;
  const bool okc0 = c0 < a.F, okc1 = c1 < a.F;
  const size_t plane = (size_t)a.Ho * a.Wo;
  f32x16 acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  const int W4 = a.W >> 2;
  for (int row = blockIdx.x; row < a.nrows; row += gridDim.x) {
    const int n = row / a.Ho, oy = row - n * a.Ho;
    // this row's dz fragments first (HBM latency under the staging of the x rows)
    float4 av[S3_KPW][2][2];
    const float* z0 = a.dz + ((size_t)n * a.F + c0) * plane + (size_t)oy * a.Wo + 8 * half;
    const float* z1 = z0 + 32 * plane;
#pragma unroll
    for (int s = 0; s < S3_KPW; ++s) {
      const int ks = wid + 4 * s;
      const bool ok = ks < a.ksteps;
      const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
      av[s][0][0] = (ok && okc0) ? *reinterpret_cast<const float4*>(z0 + 16 * ks) : zero4;
      av[s][0][1] = (ok && okc0) ? *reinterpret_cast<const float4*>(z0 + 16 * ks + 4) : zero4;
      av[s][1][0] = (ok && okc1) ? *reinterpret_cast<const float4*>(z1 + 16 * ks) : zero4;
      av[s][1][1] = (ok && okc1) ? *reinterpret_cast<const float4*>(z1 + 16 * ks + 4) : zero4;
    }
    __syncthreads();                                     // the previous row's fragment reads are done
    for (int u = tid; u < 9 * (W4 + 1); u += 256) {
      const int r = u / (W4 + 1), q = u - r * (W4 + 1);
      const int ci = r / 3, ky = r - 3 * ci;
      const int iy = 2 * oy + ky - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q > 0 && iy >= 0 && iy < a.H) v = *reinterpret_cast<const float4*>(a.x + (((size_t)n * 3 + ci) * a.H + iy) * a.W + 4 * (q - 1));
      *reinterpret_cast<float4*>(X + r * a.pitch + 4 * q) = v;       // q == 0: the left padding (ix = -4 .. -1)
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < S3_KPW; ++s) {
      const int ks = wid + 4 * s;
      if (ks < a.ksteps) {                               // wave-uniform
        float bf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bf[j] = xl[2 * (16 * ks + j)];
        if (t >= 27) {
#pragma unroll
          for (int j = 0; j < 8; ++j) bf[j] = t == 27 ? 1.f : 0.f;
        }
        bf16x8 bh, bl, ah, al;
        s3_split8(bf, bh, bl);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const float af[8] = {av[s][m][0].x, av[s][m][0].y, av[s][m][0].z, av[s][m][0].w,
                               av[s][m][1].x, av[s][m][1].y, av[s][m][1].z, av[s][m][1].w};
          s3_split8(af, ah, al);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[m], 0, 0, 0);
        }
      }
    }
  }
  // the four waves' tiles are added in wave order through LDS; one [64][32] slab per workgroup
  __syncthreads();
  float* R = X;                                          // [4][64][32] floats = 32 KB (the launch reserves at least that)
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
      R[(wid * 64 + co) * 32 + l31] = acc[m][r];
    }
  __syncthreads();
  float* slab = a.ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2048;
  for (int e = tid; e < 2048; e += 256) slab[e] = ((R[e] + R[2048 + e]) + R[4096 + e]) + R[6144 + e];
}

// dW [F][27], db [F] = fixed-order sums of the slabs: one thread per (channel, column) and slab group, combined in LDS
__global__ void __launch_bounds__(256)
k_stem3_reduce(const float* __restrict__ ws, int nslab, int F, float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float part[8][33];
  const int col = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x;                              // channel
  const float* p = ws + ((size_t)(c >> 6) * nslab) * 2048 + (size_t)(c & 63) * 32 + col;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = g;
  for (; b + 24 < nslab; b += 32) {
    s0 += p[(size_t)b * 2048]; s1 += p[(size_t)(b + 8) * 2048]; s2 += p[(size_t)(b + 16) * 2048]; s3 += p[(size_t)(b + 24) * 2048];
  }
  for (; b < nslab; b += 8) s0 += p[(size_t)b * 2048];
  part[g][col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) r += part[k][col];
    if (col < 27) dW[(size_t)c * 27 + col] = r;
    else if (col == 27) db[c] = r;
  }
}

// ---- forward with a PS output (column strips, halo slots included).  One workgroup per output row; lane = output column;
// a thread keeps its 27 input values in registers and walks the channel groups (weights: wave-uniform reads of w[F][27]).
template <bool P16>
__global__ void __launch_bounds__(256)
k_stem3_fwd_ps(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, ps_bf16x8* __restrict__ y,
               PsGeo g, PsStrips st, int F, int H, int W, int Ho, int Wo) {
  const int n = blockIdx.x / Ho, oy = blockIdx.x - n * Ho;
  for (int ox = threadIdx.x; ox < Wo; ox += 256) {
    float xv[27];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy + ky - 1;
        const bool oky = iy >= 0 && iy < H;
        const float* xr = x + (((size_t)n * 3 + ci) * H + (oky ? iy : 0)) * W + 2 * ox;
        const float2 v = *reinterpret_cast<const float2*>(xr);           // W even: 8-byte aligned
        const float vm = ox > 0 ? xr[-1] : 0.f;
        xv[(ci * 3 + ky) * 3 + 0] = oky ? vm : 0.f;
        xv[(ci * 3 + ky) * 3 + 1] = oky ? v.x : 0.f;
        xv[(ci * 3 + ky) * 3 + 2] = oky ? v.y : 0.f;
      }
    const int sidx = ox / st.Ws, xs = ox - sidx * st.Ws;
    const size_t u0 = (size_t)(sidx * st.Nimg + n) * g.img + (size_t)oy * g.WP + xs + 1;
    for (int gr = 0; gr < (F >> 3); ++gr) {
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = bias[gr * 8 + j];
#pragma unroll
      for (int k = 0; k < 27; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv[k], w[(gr * 8 + j) * 27 + k], acc[j]);
      ps_bf16x8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)acc[j];
        hi[j] = h;
        lo[j] = (__bf16)(acc[j] - (float)h);
      }
      const size_t u = u0 + (size_t)gr * g.HP * g.WP;
      y[u] = hi;
      if (!P16) y[u + g.plane] = lo;
      if (xs == 0 && sidx > 0) {                            // right halo of the strip to the left
        const size_t v = u - (size_t)st.Nimg * g.img + st.Ws;
        y[v] = hi;
        if (!P16) y[v + g.plane] = lo;
      }
      if (xs == st.Ws - 1 && sidx + 1 < st.S) {             // left halo of the strip to the right
        const size_t v = u + (size_t)st.Nimg * g.img - st.Ws;
        y[v] = hi;
        if (!P16) y[v + g.plane] = lo;
      }
    }
  }
}

int s3_num_cus() {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return ncu;
}

}  // namespace

namespace fdet {

bool stem3_wgrad_ok(int Cin, int F, int H, int W, int k, int stride, int pad) {
  if (!(Cin == 3 && k == 3 && stride == 2 && pad == 1 && F > 0 && F % 8 == 0)) return false;
  const int Wo = (W + 2 - 3) / 2 + 1;
  return H % 2 == 0 && W % 4 == 0 && Wo % 16 == 0 && Wo <= 64 * S3_KPW;
}

// slabs the workspace must hold: [F/64 blocks][grid][64][32] floats
size_t stem3_wgrad_ws_floats(int N, int F, int H, int W) {
  (void)N; (void)H; (void)W;
  return (size_t)((F + 63) / 64) * (size_t)(3 * s3_num_cus()) * 2048;
}

int stem3_wgrad(const float* x, const float* dz, float* dW, float* db, float* ws, size_t ws_floats, int N, int F, int H, int W,
                hipStream_t st) {
  Stem3WgArgs a;
  a.x = x; a.dz = dz; a.ws = ws; a.N = N; a.F = F; a.H = H; a.W = W;
  a.Ho = H / 2; a.Wo = W / 2;
  a.nrows = N * a.Ho;
  a.pitch = ((W + 4 + 59) / 64) * 64 + 4;                 // >= W + 4, == 4 (mod 64)
  a.ksteps = a.Wo / 16;
  const int fblk = (F + 63) / 64;
  int grid = std::min(a.nrows, 3 * s3_num_cus());
  grid = (int)std::min<size_t>((size_t)grid, ws_floats / ((size_t)fblk * 2048));
  if (grid < 1) return fail(FDET_EWORKSPACE, "stem_wgrad_bf16x3 (k3): workspace too small");
  const size_t lds = std::max<size_t>((size_t)9 * a.pitch * 4, (size_t)4 * 2048 * 4);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k_stem3_wgrad_x3, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) {
      (void)hipGetLastError();
      return fail(FDET_ELAUNCH, "stem_wgrad_bf16x3 (k3): cannot reserve LDS");
    }
    attr = true;
  }
  if (lds > 64 * 1024) return fail(FDET_EINVAL, "stem_wgrad_bf16x3 (k3): rows of %d columns do not fit the LDS plan", W);
  hipLaunchKernelGGL(k_stem3_wgrad_x3, dim3(grid, fblk), dim3(256), lds, st, a);
  if (int rc = check_launch("fdet_stem_wgrad_bf16x3(k3)")) return rc;
  hipLaunchKernelGGL(k_stem3_reduce, dim3(F), dim3(256), 0, st, ws, grid, F, dW, db);
  return check_launch("fdet_stem_wgrad_bf16x3(k3 reduce)");
}

bool stem3_fwd_ps_ok(int Cin, int F, int H, int W, int k, int stride, int pad) {
  return Cin == 3 && k == 3 && stride == 2 && pad == 1 && F % 8 == 0 && H % 2 == 0 && W % 2 == 0;
}

int stem3_fwd_ps(const float* x, const float* w, const float* bias, void* y_ps, int N, int F, int H, int W, hipStream_t st, bool p16) {
  PsGeo g;
  PsStrips sp;
  const int Ho = H / 2, Wo = W / 2;
  if (!ps_geo_strips(N, F, Ho, Wo, g, sp)) return fail(FDET_EINVAL, "stem_fwd_ps (k3): the %dx%d output has no PS layout", Ho, Wo);
  if (p16)
    hipLaunchKernelGGL(k_stem3_fwd_ps<true>, dim3(N * Ho), dim3(256), 0, st, x, w, bias, reinterpret_cast<ps_bf16x8*>(y_ps), g, sp, F, H, W, Ho, Wo);
  else
    hipLaunchKernelGGL(k_stem3_fwd_ps<false>, dim3(N * Ho), dim3(256), 0, st, x, w, bias, reinterpret_cast<ps_bf16x8*>(y_ps), g, sp, F, H, W, Ho, Wo);
  return check_launch("fdet_stem_fwd_ps(k3)");
}

}  // namespace fdet
