// Training head in ONE kernel per step: Dropout2d(0.5) -> Conv2d(64,5,6) -> Sigmoid -> yolo_loss (+ its gradient) -> the
// head's data gradient and per-image weight-gradient slabs.  models/PoolResnet.py:83-90,100-102, losses/YoloLoss.py:4-44,
// models/ModelMeta.py:173-176.  Replaces the launch sequence k_head_fwd, k_yolo_loss, k_sum_fixed, k_head_pack, k_head_bwd
// (0.12 ms of fp32 VALU work per step at 256 images, 5 % of its roof) by matrix-core work on one staged tile:
//
//   * one workgroup (8 waves) per image; the image's activation x * dropout scale is read from HBM ONCE and kept in LDS
//     as bf16 hi | lo twice: position-major units of 8 channels (Xb: the B operand of the forward GEMM, a tap = an index
//     offset, pitch 16) and channel-major rows (XT: the B operand of the weight-gradient GEMM, K = positions);
//   * arithmetic is the conv stack's bf16x3 (a_hi*b_lo + a_lo*b_hi + a_hi*b_hi, fp32 accumulate) on
//     v_mfma_f32_16x16x32_bf16: the head has 5 output channels, which fill 5 of 16 rows (a 32-row tile would waste 27);
//       forward  z[o][q]      = sum_{tap, c}   w[o][c][tap] * xs[c][q + tap]          M = o, N = 16 positions, K = 32 channels
//       dx       dx[c][q]     = sum_{tap, o}   w[o][c][tap] * dz[o][q - tap]          M = 16 channels, N = positions, K = 4 taps x 8 o-slots
//       dW       dW[o][c][tap]= sum_q          dz[o][q - kx] * xs[c][q + 16 ky]       M = o, N = 16 channels, K = 32 positions
//     (the kx shift of a tap sits on the tiny dz operand -- six shifted copies -- so that every 16-byte fragment read
//     of the big operand is aligned);
//   * the loss and its gradient are computed by wave 0 with the instruction sequence of k_yolo_loss (fdet_detect.hip) on
//     the sigmoid outputs in LDS: loss_per_image and loss_sum are bit-identical to the unfused path; the last workgroup
//     to finish adds the per-image losses in k_sum_fixed's order (ticket counter in the workspace, reset by its user);
//   * per-image slabs [tap][5][64] + fixed-order reduce (k_head_fused_reduce): deterministic, batch-size independent.
#include "fdet_common.h"
#include <cfloat>

using namespace fdet;

typedef __bf16 hf_bf16x8 __attribute__((ext_vector_type(8)));
typedef float hf_f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int HF_T = 512;            // threads (8 waves, two per SIMD)
constexpr int HF_F = 64;             // channels
constexpr int HF_K = 6;              // kernel size (pad 0)
constexpr int HF_KK = HF_K * HF_K;
constexpr int HF_XPT = 248;          // Xb: positions per (plane, channel group); q = iy * 16 + ix
constexpr int HF_XTP = 248;          // XT: bf16 elements per (plane, channel) row (496 B: conflict-free 16-lane reads)
constexpr int HF_DZOFF = (HF_K - 1) * 16 + (HF_K - 1);
constexpr int HF_DZP = 328;          // DZ units per plane: index HF_DZOFF + q - tapoff, q < 240
constexpr int HF_DQ = 160;           // DZT entries per (kx, plane, o): q' = sy * 16 + sx + kx
constexpr int HF_OFF_XB = 0;
constexpr int HF_OFF_XT = HF_OFF_XB + 2 * 8 * HF_XPT * 16;
constexpr int HF_OFF_DZ = HF_OFF_XT + 2 * HF_F * HF_XTP * 2;
constexpr int HF_OFF_DZT = HF_OFF_DZ + 2 * HF_DZP * 16;
constexpr int HF_OFF_PART = HF_OFF_DZT + HF_K * 2 * 5 * HF_DQ * 2;
constexpr int HF_OFF_SC = HF_OFF_PART + 2 * 5 * 160 * 4;
constexpr int HF_OFF_MISC = HF_OFF_SC + HF_F * 4;
constexpr int HF_LDS = HF_OFF_MISC + 64;
static_assert(HF_LDS <= 160 * 1024, "fused head: LDS budget");

struct HeadFusedArgs {
  const float* x;            // [N,64,H,W] f32
  const float* scale;        // [N,64] dropout scale or null
  const hf_bf16x8* wf;       // forward A fragments  [plane][tap][c32][kg][8 o-slots]   (k_head_fused_pack)
  const hf_bf16x8* wd;       // dx A fragments       [plane][tap][64 c]  x 8 o-slots
  const float* bias;         // [5]
  const float* gt;           // [N,5,So,Wo] targets
  float* y;                  // [N,5,So,Wo] sigmoid outputs
  float* lpi;                // [N] loss per image
  float* lsum;               // [1] batch sum (or null)
  unsigned* counter;         // ticket counter (zero between launches)
  float* dx;                 // [N,64,H,W]
  float* wsW;                // [N][36][5][64] weight-gradient slabs
  float* wsb;                // [N][8] bias-gradient slabs
  int N, H, W, So, Wo;
};

__device__ __forceinline__ float hf_sigmoid(float z) { return 1.f / (1.f + expf(-z)); }

__device__ __forceinline__ void hf_split8(const float (&f)[8], hf_bf16x8& hi, hf_bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)f[j];
    hi[j] = h;
    lo[j] = (__bf16)(f[j] - (float)h);
  }
}

#define HF_MFMA3(ACC, AH, AL, BH, BL)                                              \
  {                                                                                \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AH, BL, ACC, 0, 0, 0);           \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AL, BH, ACC, 0, 0, 0);           \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AH, BH, ACC, 0, 0, 0);           \
  }

__global__ void __launch_bounds__(HF_T, 1)
k_head_fused(const HeadFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  hf_bf16x8* const Xb = reinterpret_cast<hf_bf16x8*>(smem + HF_OFF_XB);       // [plane][cg][HF_XPT]
  __bf16* const XT = reinterpret_cast<__bf16*>(smem + HF_OFF_XT);              // [plane][c][HF_XTP]
  hf_bf16x8* const DZ = reinterpret_cast<hf_bf16x8*>(smem + HF_OFF_DZ);       // [plane][HF_DZP], unit = 8 o-slots
  __bf16* const DZT = reinterpret_cast<__bf16*>(smem + HF_OFF_DZT);            // [kx][plane][o 5][HF_DQ]
  float* const part = reinterpret_cast<float*>(smem + HF_OFF_PART);            // [2][5][160] forward partial sums
  float* const sc = reinterpret_cast<float*>(smem + HF_OFF_SC);                // [64]
  int* const misc = reinterpret_cast<int*>(smem + HF_OFF_MISC);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kg = lane >> 4;
  const int n = blockIdx.x;
  const int H = a.H, W = a.W, HW = H * W, So = a.So, Wo = a.Wo, P = So * Wo;

  // ---- stage: item = (channel group, position): 8 channels of one position -> one unit per plane (Xb) + 8 scalars (XT).
  // A thread owns up to 4 items (8 * 225 / 512); all of its 32 global loads are issued before the first is used.
  {
    constexpr int IT = 4;
    float v[IT][8];
    int qi[IT], cgi[IT];
#pragma unroll
    for (int r = 0; r < IT; ++r) {
      const int t = tid + r * HF_T;
      const bool ok = t < 8 * HW;
      const int tc = ok ? t : 0;
      const int cg = tc / HW, p = tc - cg * HW;
      const int iy = p / W, ix = p - iy * W;
      cgi[r] = cg; qi[r] = ok ? iy * 16 + ix : -1;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[r][j] = a.x[((size_t)n * HF_F + 8 * cg + j) * HW + p];
    }
    // zero fill (pads of every operand image) and the image's dropout scales, under the latency of those loads
    {
      hf_f32x4* z = reinterpret_cast<hf_f32x4*>(smem);
      for (int t = tid; t < HF_OFF_PART / 16; t += HF_T) z[t] = hf_f32x4{0.f, 0.f, 0.f, 0.f};
      if (tid < HF_F) sc[tid] = a.scale ? a.scale[(size_t)n * HF_F + tid] : 1.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < IT; ++r) {
      if (qi[r] >= 0) {
        const int cg = cgi[r], q = qi[r];
        float w8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w8[j] = v[r][j] * sc[8 * cg + j];
        hf_bf16x8 hi, lo;
        hf_split8(w8, hi, lo);
        Xb[(0 * 8 + cg) * HF_XPT + q] = hi;
        Xb[(1 * 8 + cg) * HF_XPT + q] = lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          XT[(0 * HF_F + 8 * cg + j) * HF_XTP + q] = hi[j];
          XT[(1 * HF_F + 8 * cg + j) * HF_XTP + q] = lo[j];
        }
      }
    }
  }
  __syncthreads();

  // ---- forward: wave = (K half, tile group); tile = one output row (16 positions, Wo of them real)
  {
    const int kh = wid & 1, tg = wid >> 1;
    const int t0 = tg < 2 ? 3 * tg : 6 + 2 * (tg - 2), nt = tg < 2 ? 3 : 2;
    hf_f32x4 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = hf_f32x4{0.f, 0.f, 0.f, 0.f};
    const int orow = m < 7 ? m : 7;                        // rows 5..15 read a zero slot of the pack
    // the A fragments come from L2 (74 KB pack, the same for every workgroup): a ring of three taps, loaded two taps ahead
    hf_bf16x8 ra[3][2][2];                                 // [ring][c32][plane]
#define HF_LOAD_WF(R, TAP)                                                                        \
    _Pragma("unroll") for (int c32_ = 0; c32_ < 2; ++c32_)                                         \
      _Pragma("unroll") for (int pl_ = 0; pl_ < 2; ++pl_)                                          \
        ra[R][c32_][pl_] = a.wf[(((pl_ * HF_KK + (TAP)) * 2 + c32_) * 4 + kg) * 8 + orow];
    const int tap0 = 18 * kh;
    HF_LOAD_WF(0, tap0)
    HF_LOAD_WF(1, tap0 + 1)
#pragma unroll
    for (int tt = 0; tt < 18; ++tt) {
      const int tap = tap0 + tt;
      if (tt + 2 < 18) { HF_LOAD_WF((tt + 2) % 3, tap + 2) }
      const int ky = tap / HF_K, kx = tap - ky * HF_K, toff = ky * 16 + kx;
#pragma unroll
      for (int c32 = 0; c32 < 2; ++c32) {
        const hf_bf16x8 ah = ra[tt % 3][c32][0], al = ra[tt % 3][c32][1];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          if (i < nt) {
            const int q = (t0 + i) * 16 + m + toff;
            const hf_bf16x8 bh = Xb[(0 * 8 + 4 * c32 + kg) * HF_XPT + q];
            const hf_bf16x8 bl = Xb[(1 * 8 + 4 * c32 + kg) * HF_XPT + q];
            HF_MFMA3(acc[i], ah, al, bh, bl)
          }
        }
      }
    }
#undef HF_LOAD_WF
    // C: column = lane & 15 (position), row = 4 * (lane >> 4) + register (output channel)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i < nt) {
        const int q = (t0 + i) * 16 + m;
        if (kg == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) part[(kh * 5 + r) * 160 + q] = acc[i][r];
        } else if (kg == 1) {
          part[(kh * 5 + 4) * 160 + q] = acc[i][0];
        }
      }
    }
  }
  __syncthreads();
  // ---- bias + sigmoid (5 * P <= 512: one value per thread), y to HBM and compact [5][P] into LDS for the loss
  float yv = 0.f;
  if (tid < 5 * P) {
    const int o = tid / P, c = tid - o * P;
    const int sy = c / Wo, sx = c - sy * Wo, q = sy * 16 + sx;
    float z = a.bias[o];
    z += part[(0 * 5 + o) * 160 + q];
    z += part[(1 * 5 + o) * 160 + q];
    yv = hf_sigmoid(z);
    a.y[(size_t)n * 5 * P + tid] = yv;
  }
  __syncthreads();
  float* const Yc = part;                                  // [5][P]
  if (tid < 5 * P) Yc[tid] = yv;
  __syncthreads();

  // ---- yolo_loss forward + backward by wave 0 (the arithmetic of k_yolo_loss, fdet_detect.hip), dz = dL/dy * y (1 - y)
  if (wid == 0) {
    const int C = P, S = So;
    const float* p = Yc;
    const float* g = a.gt + (size_t)n * 5 * C;
    float ns = 0.f;
    for (int t = lane; t < 5 * C; t += 64) { const float v = p[t]; ns += (v == v) ? v : 0.f; }
    ns = wave_sum_all(ns);
    const bool fix = (ns != 0.f);
    const float inv_s = (float)(1.0 / (double)S);
    float acc = 0.f;
    float dbs[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = lane; c < C; c += 64) {
      float pv[5], fin[5], raw[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        float v = p[k * C + c];
        raw[k] = v;
        fin[k] = 1.f;
        if (fix) {
          fin[k] = (isfinite(v)) ? 1.f : 0.f;
          if (v != v) v = 0.1f;
          else if (isinf(v)) v = (v > 0.f) ? FLT_MAX : -FLT_MAX;
        }
        pv[k] = v;
      }
      const float g0 = g[c], g1 = g[C + c], g2 = g[2 * C + c], g3 = g[3 * C + c], g4 = g[4 * C + c];
      const float obj = g0, noobj = 1.f - g0;
      const float dxx = g1 - pv[2], dyy = g2 - pv[1];
      const float sg3 = sqrtf(g3), sp3 = sqrtf(pv[3]), sg4 = sqrtf(g4), sp4 = sqrtf(pv[4]);
      const float dw = sg3 - sp3, dh = sg4 - sp4;
      const float cw = 3.f * obj;
      const float xy = cw * (dxx * dxx + dyy * dyy);
      const float wh = cw * (dw * dw + dh * dh);
      const float wconf = obj + noobj * inv_s;
      const float dc = g0 - pv[0];
      const float conf = wconf * (dc * dc);
      acc += xy + wh + conf;
      float d[5];
      d[0] = wconf * (2.f * dc) * -1.f;
      d[1] = cw * (2.f * dyy) * -1.f;
      d[2] = cw * (2.f * dxx) * -1.f;
      d[3] = (cw * (2.f * dw) * -1.f) * (0.5f * (1.f / sp3));
      d[4] = (cw * (2.f * dh) * -1.f) * (0.5f * (1.f / sp4));
      float dz[8];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        float v = d[k];
        if (fix) v = v * fin[k];
        dz[k] = v * (raw[k] * (1.f - raw[k]));             // k_head_bwd: dy * (y * (1 - y))
        dbs[k] += dz[k];
      }
      dz[5] = dz[6] = dz[7] = 0.f;
      hf_bf16x8 hi, lo;
      hf_split8(dz, hi, lo);
      const int sy = c / Wo, sx = c - sy * Wo, q = sy * 16 + sx;
      DZ[0 * HF_DZP + HF_DZOFF + q] = hi;
      DZ[1 * HF_DZP + HF_DZOFF + q] = lo;
#pragma unroll
      for (int kx = 0; kx < HF_K; ++kx)
#pragma unroll
        for (int o = 0; o < 5; ++o) {
          DZT[((kx * 2 + 0) * 5 + o) * HF_DQ + q + kx] = hi[o];
          DZT[((kx * 2 + 1) * 5 + o) * HF_DQ + q + kx] = lo[o];
        }
    }
    acc = wave_sum(acc);
#pragma unroll
    for (int k = 0; k < 5; ++k) dbs[k] = wave_sum(dbs[k]);
    if (lane == 0) {
      a.lpi[n] = acc;
#pragma unroll
      for (int k = 0; k < 5; ++k) a.wsb[(size_t)n * 8 + k] = dbs[k];
      int last = 0;
      if (a.lsum) {
        __threadfence();
        last = atomicAdd(a.counter, 1u) == (unsigned)(a.N - 1);
      }
      misc[0] = last;
    }
  }
  __syncthreads();

  // ---- backward, part 1: dx.  wave = (16-channel tile, half of the rows); K step = 4 taps x 8 o-slots
  {
    const int mt = wid & 3, nh = wid >> 2;
    const int r0 = nh * 8, nr = nh ? 7 : 8;
    hf_f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = hf_f32x4{0.f, 0.f, 0.f, 0.f};
    hf_bf16x8 rd[3][2];                                    // [ring][plane]: A fragments two K steps ahead (L2)
#define HF_LOAD_WD(R, KS)                                                                         \
    _Pragma("unroll") for (int pl_ = 0; pl_ < 2; ++pl_)                                            \
      rd[R][pl_] = a.wd[(pl_ * HF_KK + 4 * (KS) + kg) * HF_F + 16 * mt + m];
    HF_LOAD_WD(0, 0)
    HF_LOAD_WD(1, 1)
#pragma unroll
    for (int ks = 0; ks < HF_KK / 4; ++ks) {
      if (ks + 2 < HF_KK / 4) { HF_LOAD_WD((ks + 2) % 3, ks + 2) }
      const int tap = 4 * ks + kg;
      const int ky = tap / HF_K, kx = tap - ky * HF_K, toff = ky * 16 + kx;
      const hf_bf16x8 ah = rd[ks % 3][0], al = rd[ks % 3][1];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < nr) {
          const int u = HF_DZOFF + (r0 + i) * 16 + m - toff;
          const hf_bf16x8 bh = DZ[0 * HF_DZP + u];
          const hf_bf16x8 bl = DZ[1 * HF_DZP + u];
          HF_MFMA3(acc[i], ah, al, bh, bl)
        }
      }
    }
#undef HF_LOAD_WD
    // C: column = position ix = lane & 15, row = channel 16 mt + 4 kg + register
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int iy = r0 + i;
      if (i < nr && iy < H && m < W) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * mt + 4 * kg + r;
          const float v = acc[i][r];
          a.dx[(((size_t)n * HF_F + f) * H + iy) * W + m] = a.scale ? v * sc[f] : v;
        }
      }
    }
  }
  // ---- backward, part 2: weight-gradient slab of this image.  wave = (16-channel tile, three kx); K = 32 positions
  {
    const int nt = wid & 3, kx0 = (wid >> 2) * 3;
    const int orow = m < 4 ? m : 4;
    float* const slab = a.wsW + (size_t)n * HF_KK * 5 * HF_F;
    for (int kx = kx0; kx < kx0 + 3; ++kx) {
      hf_bf16x8 ah[5], al[5];
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        const hf_bf16x8 zero = {};
        const hf_bf16x8 h = *reinterpret_cast<const hf_bf16x8*>(DZT + ((kx * 2 + 0) * 5 + orow) * HF_DQ + 32 * ks + 8 * kg);
        const hf_bf16x8 l = *reinterpret_cast<const hf_bf16x8*>(DZT + ((kx * 2 + 1) * 5 + orow) * HF_DQ + 32 * ks + 8 * kg);
        ah[ks] = m < 5 ? h : zero;
        al[ks] = m < 5 ? l : zero;
      }
      hf_f32x4 acc[HF_K];
#pragma unroll
      for (int ky = 0; ky < HF_K; ++ky) acc[ky] = hf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 5; ++ks)
#pragma unroll
        for (int ky = 0; ky < HF_K; ++ky) {
          const hf_bf16x8 bh = *reinterpret_cast<const hf_bf16x8*>(XT + (0 * HF_F + 16 * nt + m) * HF_XTP + 32 * ks + 8 * kg + 16 * ky);
          const hf_bf16x8 bl = *reinterpret_cast<const hf_bf16x8*>(XT + (1 * HF_F + 16 * nt + m) * HF_XTP + 32 * ks + 8 * kg + 16 * ky);
          HF_MFMA3(acc[ky], ah[ks], al[ks], bh, bl)
        }
      // C: column = channel 16 nt + (lane & 15), row = o = 4 kg + register
#pragma unroll
      for (int ky = 0; ky < HF_K; ++ky) {
        float* const s = slab + (size_t)((ky * HF_K + kx) * 5) * HF_F + 16 * nt + m;
        if (kg == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) s[r * HF_F] = acc[ky][r];
        } else if (kg == 1) {
          s[4 * HF_F] = acc[ky][0];
        }
      }
    }
  }
  // ---- batch loss: the last workgroup to finish adds the per-image losses in k_sum_fixed's order
  if (misc[0]) {
    __syncthreads();                                       // (uniform: misc[0] was written before the barrier above)
    float* const red = part;
    if (tid < 256) {
      float s = 0.f;
      for (int i = tid; i < a.N; i += 256) s += __hip_atomic_load(a.lpi + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      red[tid] = s;
    }
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (tid < w) red[tid] += red[tid + w];
      __syncthreads();
    }
    if (tid == 0) {
      a.lsum[0] = red[0] * 1.0f;
      *a.counter = 0u;                                     // ready for the next launch
    }
  }
}

// A-operand fragment packs of the fused kernel (once per optimisation step):
//   wf[plane][tap][c32][kg][slot]  x 8 bf16 = w[o = slot][32 c32 + 8 kg + j][tap], slots 5..7 zero
//   wd[plane][tap][c]              x 8 bf16 = w[o = j][c][tap], j = 5..7 zero
__global__ void __launch_bounds__(256)
k_head_fused_pack(const float* __restrict__ w, hf_bf16x8* __restrict__ wf, hf_bf16x8* __restrict__ wd) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int nwf = HF_KK * 2 * 4 * 8, nwd = HF_KK * HF_F;
  if (t < nwf) {
    const int slot = t & 7, kgp = (t >> 3) & 3, c32 = (t >> 5) & 1, tap = t >> 6;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = slot < 5 ? w[((size_t)slot * HF_F + 32 * c32 + 8 * kgp + j) * HF_KK + tap] : 0.f;
    hf_bf16x8 hi, lo;
    hf_split8(v, hi, lo);
    wf[t] = hi;
    wf[nwf + t] = lo;
  } else if (t < nwf + nwd) {
    const int u = t - nwf, c = u & 63, tap = u >> 6;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = j < 5 ? w[((size_t)j * HF_F + c) * HF_KK + tap] : 0.f;
    hf_bf16x8 hi, lo;
    hf_split8(v, hi, lo);
    wd[u] = hi;
    wd[nwd + u] = lo;
  }
}

// dW[o][c][tap] = sum_n slab[n][tap][o][c], db[o] = sum_n wsb[n][o]: 64 slab entries x 16 image phases per workgroup,
// fixed-order combine (as k_head_reduce)
__global__ void __launch_bounds__(1024)
k_head_fused_reduce(const float* __restrict__ wsW, const float* __restrict__ wsb, int N, float* __restrict__ dW,
                    float* __restrict__ db) {
  __shared__ float part[1024];
  const int q = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int nS = HF_KK * 5 * HF_F;
  const int t = blockIdx.x * 64 + q;                       // slab entry (tap, o, c), or nS + o for the bias
  float s = 0.f;
  if (t < nS) {
    for (int n = ph; n < N; n += 16) s += wsW[(size_t)n * nS + t];
  } else if (t < nS + 5) {
    for (int n = ph; n < N; n += 16) s += wsb[(size_t)n * 8 + (t - nS)];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if (ph == 0) {
    float tot = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) tot += part[p * 64 + q];
    if (t < nS) {
      const int c = t & 63, r = t >> 6, o = r % 5, tap = r / 5;
      dW[((size_t)o * HF_F + c) * HF_KK + tap] = tot;
    } else if (t < nS + 5) {
      db[t - nS] = tot;
    }
  }
}

constexpr size_t hf_pack_units() { return (size_t)2 * HF_KK * 2 * 4 * 8 + (size_t)2 * HF_KK * HF_F; }

}  // namespace

extern "C" int fdet_head_loss_fused_supported(int F, int H, int W, int k, int pad) {
  const int So = H - k + 1, Wo = W - k + 1;
  return F == HF_F && k == HF_K && pad == 0 && H >= k && W >= k && H <= 15 && W <= 15 && So <= 10 && 5 * So * Wo <= HF_T;
}

extern "C" size_t fdet_head_loss_fused_ws_bytes(int N, int F, int H, int W, int k, int pad) {
  if (!fdet_head_loss_fused_supported(F, H, W, k, pad) || N < 1) return 0;
  return hf_pack_units() * 16 + ((size_t)N * HF_KK * 5 * HF_F + (size_t)N * 8) * 4 + 64;
}

// One launch sequence (pack, fused kernel, slab reduce) for
//   y = sigmoid(conv(x * drop_scale, w) + bias);  loss_per_image[n] = yolo_loss(y[n], gt[n]);  loss_sum = sum_n
//   dx = d loss_sum / d x,  dW = d loss_sum / d w,  db = d loss_sum / d bias
// ws: fdet_head_loss_fused_ws_bytes() bytes whose LAST 64 bytes hold the ticket counter: they must be zero at the first
// call (the kernel leaves them zero) and the workspace must not be shared by launches that may overlap.
extern "C" int fdet_head_loss_fused(const float* x, const float* drop_scale, const float* w, const float* bias,
                                    const float* gt, float* y, float* loss_per_image, float* loss_sum, float* dx,
                                    float* dW, float* db, void* ws, size_t ws_bytes, int N, int F, int H, int W, int k,
                                    int pad, void* stream) {
  FDET_REQUIRE(x && w && bias && gt && y && loss_per_image && dx && dW && db && ws && N > 0, "head_loss_fused: bad arguments");
  FDET_REQUIRE(fdet_head_loss_fused_supported(F, H, W, k, pad),
               "head_loss_fused: needs F=64, k=6, pad=0 and a map of at most 15x15 (got F=%d k=%d pad=%d %dx%d)", F, k, pad, H, W);
  const size_t need = fdet_head_loss_fused_ws_bytes(N, F, H, W, k, pad);
  if (ws_bytes < need) return fail(FDET_EWORKSPACE, "head_loss_fused: workspace %zu < %zu bytes", ws_bytes, need);
  FDET_REQUIRE(((size_t)ws & 15) == 0, "head_loss_fused: workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  HeadFusedArgs a;
  hf_bf16x8* pk = reinterpret_cast<hf_bf16x8*>(ws);
  hf_bf16x8* wf = pk;
  hf_bf16x8* wd = pk + (size_t)2 * HF_KK * 2 * 4 * 8;
  float* wsW = reinterpret_cast<float*>(pk + hf_pack_units());
  float* wsb = wsW + (size_t)N * HF_KK * 5 * HF_F;
  unsigned* counter = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + need - 64);
  a.x = x; a.scale = drop_scale; a.wf = wf; a.wd = wd; a.bias = bias; a.gt = gt; a.y = y; a.lpi = loss_per_image;
  a.lsum = loss_sum; a.counter = counter; a.dx = dx; a.wsW = wsW; a.wsb = wsb;
  a.N = N; a.H = H; a.W = W; a.So = H - k + 1; a.Wo = W - k + 1;
  const int npk = HF_KK * 2 * 4 * 8 + HF_KK * HF_F;
  hipLaunchKernelGGL(k_head_fused_pack, dim3((npk + 255) / 256), dim3(256), 0, st, w, wf, wd);
  static bool attr_set = false;
  if (!attr_set) {
    if (int rc_ = set_lds_attr((const void*)k_head_fused, (size_t)HF_LDS, __func__)) return rc_;
    attr_set = true;
  }
  hipLaunchKernelGGL(k_head_fused, dim3(N), dim3(HF_T), HF_LDS, st, a);
  if (int rc = check_launch("fdet_head_loss_fused")) return rc;
  const int nS = HF_KK * 5 * HF_F;
  hipLaunchKernelGGL(k_head_fused_reduce, dim3((nS + 5 + 63) / 64), dim3(1024), 0, st, wsW, wsb, N, dW, db);
  return check_launch("fdet_head_loss_fused(reduce)");
}
