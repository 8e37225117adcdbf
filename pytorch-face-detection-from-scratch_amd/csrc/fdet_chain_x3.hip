// Residual-block CHAIN at one resolution with the running activation resident on the CU
// (bf16x3 arithmetic: x = hi + lo in bf16, a_hi*b_lo + a_lo*b_hi + a_hi*b_hi, fp32 accumulate).
//
// At the last resolution of PoolResnet (15x15, blocks 2..9 = 16 convs) a conv layer is ~7 us of
// matrix work but ~22 us as its own launch (prologue, four staging rounds, epilogue, tail of the
// grid).  Here ONE workgroup owns ONE image for the whole chain: a 64-channel 15x15 map is 57.6 KB of
// fp32 = 70 KB as channel-innermost bf16 hi/lo slots in LDS -- it stays there, the layer's output is
// written back over it, and HBM only sees what backward needs (a, c, block outputs).  Only the
// weight panels (147 KB per layer, L2-resident, shared by every workgroup) stream through a
// double-buffered 2 x 36.8 KB LDS ring, one 16-channel chunk ahead, across layer boundaries.
//
//   forward  (models/PoolResnet.py:33-43, pool == 1):  a = lrelu(conv1(h)); c = lrelu(conv2(a));
//            h <- c*drop_scale + h
//   backward (its autograd, data path only):           dz2 = dout*drop_scale*lrelu'(c);
//            dz1 = conv2^T(dz2)*lrelu'(a);  dout <- conv1^T(dz1) + dout
//            (dz2 / dz1 are stored: they are the operands of the weight gradients)
//
// GEMM orientation: M = output channels (A operand = weights), N = 32 positions per wave
// (B operand = activations), so a lane holds ONE position and 16 channels per 32-channel tile:
// exactly the (position, 8-channel slot) pieces the next layer's B operand is made of -- the
// epilogue writes them back into LDS with 8-byte stores, no cross-lane traffic.
// 8 waves (two per SIMD), each 32 positions x 64 channels; the skip connection lives in registers.
#include "fdet_common.h"
#include <cstdint>

using namespace fdet;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MAXL = 32;          // conv layers per launch (16 blocks)
constexpr int NTHR = 512;         // 8 waves
constexpr int FCH = 64;           // channels
constexpr int A_UNITS = 9 * 2 * FCH;   // 16-byte units per weight array (hi or lo) per 16-channel chunk
constexpr int NWLD = (2 * A_UNITS + NTHR - 1) / NTHR;   // weight staging units per thread

struct ChainArgs {
  const float* in;               // fwd: x ; bwd: dout            [N,64,H,W]
  const bf16x8* w[MAXL];         // per executed layer: bf16x3 panel (hi units; lo = + 4*9*2*64)
  const float* bias[MAXL];       // fwd: bias of the layer
  const float* sc[MAXL];         // fwd odd layers: dropout scale of the block ; bwd odd layers: scale of the NEXT block to run
  const float* ld[MAXL];         // bwd even layers: a_k (lrelu') ; bwd odd layers: c of the next block to run
  float* st[MAXL];               // fwd: a (even) / c (odd) ; bwd even: dz1
  float* st2[MAXL];              // fwd odd: block output ; bwd odd: dz2 of the next block to run, or dx after the last
  const float* pre_ld;           // bwd: c of the first block to run
  const float* pre_sc;           // bwd: its dropout scale
  float* pre_st;                 // bwd: its dz2
  int nlayers, N, H, W, WP, PT, bwd;
  float slope;
};

__device__ __forceinline__ void split4(const float (&f)[4], bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 h = (__bf16)f[j];
    hi[j] = h;
    lo[j] = (__bf16)(f[j] - (float)h);
  }
}

template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// 4x4 dword transpose across each quad of lanes: register i of lane j <-> register j of lane i
__device__ __forceinline__ void quad_transpose4(float (&v)[4], bool b0, bool b1) {
#pragma unroll
  for (int k = 0; k < 4; k += 2) {
    const float lo = v[k], hi = v[k + 1];
    const float recv = dpp_quad<0xB1>(b0 ? lo : hi);            // quad_perm [1,0,3,2]
    v[k] = b0 ? recv : lo;
    v[k + 1] = b0 ? hi : recv;
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float lo = v[k], hi = v[k + 2];
    const float recv = dpp_quad<0x4E>(b1 ? lo : hi);            // quad_perm [2,3,0,1]
    v[k] = b1 ? recv : lo;
    v[k + 2] = b1 ? hi : recv;
  }
}

__global__ void __launch_bounds__(NTHR, 1)
k_block_chain_x3(const ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8* X = reinterpret_cast<bf16x8*>(smem);                 // [c16 4][hl 2][kh 2][PT] units
  const int PT = a.PT, WP = a.WP;
  bf16x8* Wb = X + 16 * PT;                                     // 2 buffers x [hi A_UNITS | lo A_UNITS]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int n = blockIdx.x;
  const int HW = a.H * a.W;
  const int q = wid * 32 + l31;                                 // position in the padded row space
  const int pr = q / WP, pc = q - pr * WP;
  const bool valid = pr < a.H && pc < a.W;
  const int ebase = (n * FCH) * HW + (valid ? pr * a.W + pc : 0);   // + ch*HW
  const int xslot = q + WP + 1;                                 // slot of this position in the X arrays
  // transposed arrangement for global I/O: this lane's 4 consecutive positions and channel offset
  const int q4 = q & ~3, pr4 = q4 / WP, pc4 = q4 - pr4 * WP;
  const int nv4 = pr4 < a.H ? max(0, min(4, a.W - pc4)) : 0;
  const bool qb0 = l31 & 1, qb1 = l31 & 2;
  const int ebase4 = (n * FCH + 4 * half + (l31 & 3)) * HW + (nv4 > 0 ? pr4 * a.W + pc4 : 0);   // + (32m + 8g)*HW

  {  // zero X (halo rows / pad columns stay zero for the whole chain)
    f32x4* z = reinterpret_cast<f32x4*>(smem);
    for (int t = tid; t < 16 * PT; t += NTHR) z[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- weight staging: thread -> units of the [hi | lo] chunk image
  bf16x8 pw[NWLD];
  const int a_layer_units = 4 * A_UNITS;                        // hi (or lo) units per layer
  // source offsets (tile invariant): unit u of the [hi | lo] chunk image <- panel unit
  int wsrc[NWLD];
#pragma unroll
  for (int s_ = 0; s_ < NWLD; ++s_) {
    const int u_ = min(tid + s_ * NTHR, 2 * A_UNITS - 1);
    const int lo_ = u_ >= A_UNITS ? 1 : 0;
    wsrc[s_] = lo_ * a_layer_units + (u_ - lo_ * A_UNITS);
  }
#define CH_ISSUE_W1(BASE, S) { pw[S] = (BASE)[wsrc[S]]; }
#define CH_WRITE_W1(BUF, S)                                                                        \
  {                                                                                                \
    const int u_ = tid + (S) * NTHR;                                                               \
    if (u_ < 2 * A_UNITS) (Wb + (BUF) * 2 * A_UNITS)[u_] = pw[S];                                  \
  }
#define CH_ISSUE_W(L, C16)                                                                         \
  {                                                                                                \
    const bf16x8* base_ = a.w[L] + (C16) * A_UNITS;                                                \
    _Pragma("unroll") for (int s_ = 0; s_ < NWLD; ++s_) CH_ISSUE_W1(base_, s_)                     \
  }
#define CH_WRITE_W(BUF)                                                                            \
  {                                                                                                \
    _Pragma("unroll") for (int s_ = 0; s_ < NWLD; ++s_) CH_WRITE_W1(BUF, s_)                       \
  }
  // value of (m, r16) is channel 32m + 8*(r16>>2) + 4*half + (r16&3) at position q
#define CH_OF(M, R) (32 * (M) + 8 * ((R) >> 2) + 4 * half + ((R) & 3))
  // write a full 64-channel register tile into the X arrays (bf16 hi/lo split), valid positions only
#define CH_WRITE_X(V)                                                                              \
  {                                                                                                \
    if (valid) {                                                                                   \
      _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                             \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                         \
          const float f_[4] = {V[m_][4 * g_], V[m_][4 * g_ + 1], V[m_][4 * g_ + 2], V[m_][4 * g_ + 3]}; \
          bf16x4 hi_, lo_;                                                                         \
          split4(f_, hi_, lo_);                                                                    \
          const int c16_ = 2 * m_ + (g_ >> 1), kh_ = g_ & 1;                                       \
          bf16x4* uh_ = reinterpret_cast<bf16x4*>(X + ((c16_ * 2 + 0) * 2 + kh_) * PT + xslot);    \
          bf16x4* ul_ = reinterpret_cast<bf16x4*>(X + ((c16_ * 2 + 1) * 2 + kh_) * PT + xslot);    \
          uh_[half] = hi_;                                                                         \
          ul_[half] = lo_;                                                                         \
        }                                                                                          \
    }                                                                                              \
  }
  // Global tile I/O, 16 bytes per lane: a 4x4 dword transpose across each quad of lanes turns
  // (lane = position, 4 registers = 4 consecutive channels) into (lane = channel, 4 registers = 4
  // consecutive positions of one row), so a tile moves in 8 instructions instead of 32 -- the
  // vector-memory instruction rate of the CU, not bytes, limits a burst of dword accesses.
  // After the transpose lane j of a quad owns channel 32m + 8g + 4*half + j at positions q4 .. q4+3.
#define CH_LD_(BYTES)                                                                              \
  _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                                 \
    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_)                                               \
      __builtin_memcpy(&t_[m_][g_], p_ + ebase4 + (32 * m_ + 8 * g_) * HW, BYTES);
  // the branch on the valid count sits OUTSIDE the tile loops: eight independent loads per arm
  // (a branch per load makes the compiler serialise every load behind an s_waitcnt)
#define CH_LOAD_TILE(DST, PTR)                                                                     \
  {                                                                                                \
    const float* __restrict__ p_ = (PTR);                                                          \
    f32x4 t_[2][4];                                                                                \
    _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                               \
      _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) t_[m_][g_] = f32x4{0.f, 0.f, 0.f, 0.f};     \
    if (nv4 == 4) { CH_LD_(16) } else if (nv4 == 3) { CH_LD_(12) } else if (nv4 == 2) { CH_LD_(8) } else if (nv4 == 1) { CH_LD_(4) } \
    _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                               \
      _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                           \
        float v_[4] = {t_[m_][g_][0], t_[m_][g_][1], t_[m_][g_][2], t_[m_][g_][3]};                \
        quad_transpose4(v_, qb0, qb1);                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) DST[m_][4 * g_ + i_] = v_[i_];            \
      }                                                                                            \
  }
#define CH_STORE_TILE(PTR, V)                                                                      \
  {                                                                                                \
    float* __restrict__ p_ = (PTR);                                                                \
    f32x4 t_[2][4];                                                                                \
    _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                               \
      _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                           \
        float v_[4] = {V[m_][4 * g_], V[m_][4 * g_ + 1], V[m_][4 * g_ + 2], V[m_][4 * g_ + 3]};    \
        quad_transpose4(v_, qb0, qb1);                                                             \
        t_[m_][g_] = f32x4{v_[0], v_[1], v_[2], v_[3]};                                            \
      }                                                                                            \
    if (nv4 == 4) { CH_ST_(16) } else if (nv4 == 3) { CH_ST_(12) } else if (nv4 == 2) { CH_ST_(8) } else if (nv4 == 1) { CH_ST_(4) } \
  }
#define CH_ST_(BYTES)                                                                              \
  _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                                 \
    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_)                                               \
      __builtin_memcpy(p_ + ebase4 + (32 * m_ + 8 * g_) * HW, &t_[m_][g_], BYTES);

  // per-(image, channel) dropout scales in tile layout (1 when the pointer is null)
#define CH_SCALE_TILE(DST, PTR)                                                                    \
  {                                                                                                \
    const float* __restrict__ p_ = (PTR);                                                          \
    if (p_) {                                                                                      \
      _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                             \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) DST[m_][r_] = p_[n * FCH + CH_OF(m_, r_)]; \
    } else {                                                                                       \
      _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                             \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) DST[m_][r_] = 1.f;                       \
    }                                                                                              \
  }
  f32x16 Hreg[2];                                               // skip connection / running gradient
  f32x16 aux[2];                                                // prefetched lrelu' operand (bwd)
  CH_ISSUE_W(0, 0)
  CH_LOAD_TILE(Hreg, a.in)
  if (a.bwd) CH_LOAD_TILE(aux, a.pre_ld)
  __syncthreads();                                              // zero fill done
  CH_WRITE_W(0)
  if (!a.bwd) {
    CH_WRITE_X(Hreg)
  } else {
    // dz2 of the first block to run = dout * scale * lrelu'(c)
    f32x16 t[2];
    CH_SCALE_TILE(t, a.pre_sc)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) t[m][r] = Hreg[m][r] * t[m][r] * (aux[m][r] > 0.f ? 1.f : a.slope);
    CH_STORE_TILE(a.pre_st, t)
    CH_WRITE_X(t)
  }
  __syncthreads();

  int tapoff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) tapoff[t] = (t / 3) * WP + (t % 3);
  const int w_off = half * FCH + l31;                           // + tap*2*64 + m*32 ; lo: + A_UNITS
  const int x_off = half * PT + wid * 32 + l31;                 // + (c16*2 + hl)*2*PT + tapoff

  int stage = 0;
  for (int L = 0; L < a.nlayers; ++L) {
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const bool odd = L & 1;
    // the lrelu' operand of this layer's epilogue travels while the MFMAs run
    if (a.bwd && a.ld[L]) CH_LOAD_TILE(aux, a.ld[L])
    for (int c = 0; c < 4; ++c, ++stage) {
      const bool has_next = c < 3 || L + 1 < a.nlayers;
      // next weight chunk (possibly the next layer's first): one load per tap, one LDS write per
      // later tap -- a burst at the chunk boundary would stall all eight waves on the memory pipe
      const bf16x8* wnext = has_next ? (c < 3 ? a.w[L] + (c + 1) * A_UNITS : a.w[L + 1]) : a.w[L];
      const bf16x8* Ww = Wb + (stage & 1) * 2 * A_UNITS + w_off;
      const bf16x8* Xh = X + (c * 2 + 0) * 2 * PT + x_off;
      const bf16x8* Xl = X + (c * 2 + 1) * 2 * PT + x_off;
      bf16x8 wh[2][2], wl[2][2], xh[2], xl[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) { wh[0][m] = Ww[m * 32]; wl[0][m] = Ww[A_UNITS + m * 32]; }
      xh[0] = Xh[tapoff[0]]; xl[0] = Xl[tapoff[0]];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        if (t + 1 < 9) {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            wh[nxt][m] = Ww[(t + 1) * 2 * FCH + m * 32];
            wl[nxt][m] = Ww[A_UNITS + (t + 1) * 2 * FCH + m * 32];
          }
          xh[nxt] = Xh[tapoff[t + 1]];
          xl[nxt] = Xl[tapoff[t + 1]];
        }
        if (t < NWLD) CH_ISSUE_W1(wnext, t)
        __builtin_amdgcn_sched_barrier(0);                      // keep the fragment reads one tap ahead of their MFMAs
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cur][m], xl[cur], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[cur][m], xh[cur], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cur][m], xh[cur], acc[m], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t >= 9 - NWLD && has_next) CH_WRITE_W1((stage & 1) ^ 1, t - (9 - NWLD))
      }
      __syncthreads();                                          // chunk consumed by every wave; next one visible
    }

    // ---- epilogue: every wave is past its last read of X, so X can be overwritten in place
    if (!a.bwd) {
      const float* __restrict__ bias = a.bias[L];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float t = acc[m][r] + bias[CH_OF(m, r)];
          acc[m][r] = t > 0.f ? t : t * a.slope;
        }
      if (a.st[L]) CH_STORE_TILE(a.st[L], acc)                  // a (even) / c (odd), saved for backward
      if (odd) {
        f32x16 s2[2];
        CH_SCALE_TILE(s2, a.sc[L])
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) Hreg[m][r] = acc[m][r] * s2[m][r] + Hreg[m][r];
        if (a.st2[L]) CH_STORE_TILE(a.st2[L], Hreg)             // block output
        if (L + 1 < a.nlayers) CH_WRITE_X(Hreg)
      } else {
        CH_WRITE_X(acc)
      }
    } else {
      if (!odd) {                                               // conv2^T: dz1 = acc * lrelu'(a)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][r] *= (aux[m][r] > 0.f ? 1.f : a.slope);
        CH_STORE_TILE(a.st[L], acc)
        CH_WRITE_X(acc)
      } else {                                                  // conv1^T: dx = acc + dout
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) Hreg[m][r] += acc[m][r];
        if (L + 1 < a.nlayers) {                                // dz2 of the next block to run
          CH_SCALE_TILE(acc, a.sc[L])
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = Hreg[m][r] * acc[m][r] * (aux[m][r] > 0.f ? 1.f : a.slope);
          CH_STORE_TILE(a.st2[L], acc)
          CH_WRITE_X(acc)
        } else {
          CH_STORE_TILE(a.st2[L], Hreg)                         // gradient w.r.t. the chain input
        }
      }
    }
    __syncthreads();                                            // new X visible
  }
}

int chain_geometry(int F, int H, int W, int& WP, int& PT, size_t& lds) {
  if (F != FCH || H <= 0 || W <= 0) return 0;
  WP = (W + 1 + 3) / 4 * 4;
  if (H * WP > 8 * 32) return 0;                                // 8 waves x 32 positions
  PT = (H + 2) * WP + 3;
  if (PT < 8 * 32 + 2 * WP + 3) PT = 8 * 32 + 2 * WP + 3;       // garbage positions of the last wave read in range
  lds = ((size_t)16 * PT + 2 * 2 * A_UNITS) * 16;
  return lds <= 160 * 1024;
}

int launch_chain(ChainArgs& a, hipStream_t st) {
  size_t lds = 0;
  if (!chain_geometry(FCH, a.H, a.W, a.WP, a.PT, lds))
    return fail(FDET_EINVAL, "block_chain_bf16x3: unsupported map %dx%d (needs 64 channels and H*roundup4(W+1) <= 256)", a.H, a.W);
  if ((size_t)a.N * FCH * a.H * a.W >= ((size_t)1 << 31)) return fail(FDET_EINVAL, "block_chain_bf16x3: tensor too large");
  { if (int rc_ = set_lds_attr((const void*)k_block_chain_x3, (size_t)(lds), __func__)) return rc_; }
  hipLaunchKernelGGL(k_block_chain_x3, dim3(a.N), dim3(NTHR), lds, st, a);
  return check_launch("fdet_block_chain_bf16x3");
}

}  // namespace

extern "C" int fdet_block_chain_supported(int F, int H, int W) {
  int WP, PT; size_t lds;
  return chain_geometry(F, H, W, WP, PT, lds);
}

extern "C" int fdet_block_chain_fwd_bf16x3(const float* x, const void* const* h_wpk1, const float* const* h_b1,
                                           const void* const* h_wpk2, const float* const* h_b2,
                                           const float* const* h_scale, float* const* h_a, float* const* h_c,
                                           float* const* h_out, int nblocks, int N, int F, int H, int W, float slope,
                                           void* stream) {
  FDET_REQUIRE(x && h_wpk1 && h_b1 && h_wpk2 && h_b2 && h_out, "block_chain_fwd_bf16x3: null pointer");
  FDET_REQUIRE(nblocks >= 1 && 2 * nblocks <= MAXL && N > 0, "block_chain_fwd_bf16x3: 1..%d blocks (got %d), N=%d", MAXL / 2, nblocks, N);
  FDET_REQUIRE(F == FCH, "block_chain_fwd_bf16x3: 64 channels only (got %d)", F);
  ChainArgs a{};
  a.in = x; a.nlayers = 2 * nblocks; a.N = N; a.H = H; a.W = W; a.bwd = 0; a.slope = slope;
  for (int k = 0; k < nblocks; ++k) {
    FDET_REQUIRE(h_wpk1[k] && h_wpk2[k] && h_b1[k] && h_b2[k], "block_chain_fwd_bf16x3: null weights in block %d", k);
    a.w[2 * k] = (const bf16x8*)h_wpk1[k]; a.w[2 * k + 1] = (const bf16x8*)h_wpk2[k];
    a.bias[2 * k] = h_b1[k]; a.bias[2 * k + 1] = h_b2[k];
    a.sc[2 * k + 1] = h_scale ? h_scale[k] : nullptr;
    a.st[2 * k] = h_a ? h_a[k] : nullptr;
    a.st[2 * k + 1] = h_c ? h_c[k] : nullptr;
    a.st2[2 * k + 1] = h_out[k];
  }
  FDET_REQUIRE(h_out[nblocks - 1], "block_chain_fwd_bf16x3: the last block's output pointer is required");
  return launch_chain(a, (hipStream_t)stream);
}

extern "C" int fdet_block_chain_bwd_bf16x3(const float* dout, const void* const* h_wpk1b, const void* const* h_wpk2b,
                                           const float* const* h_scale, const float* const* h_a,
                                           const float* const* h_c, float* const* h_dz1, float* const* h_dz2,
                                           float* dx, int nblocks, int N, int F, int H, int W, float slope,
                                           void* stream) {
  FDET_REQUIRE(dout && h_wpk1b && h_wpk2b && h_a && h_c && h_dz1 && h_dz2 && dx, "block_chain_bwd_bf16x3: null pointer");
  FDET_REQUIRE(nblocks >= 1 && 2 * nblocks <= MAXL && N > 0, "block_chain_bwd_bf16x3: 1..%d blocks (got %d), N=%d", MAXL / 2, nblocks, N);
  FDET_REQUIRE(F == FCH, "block_chain_bwd_bf16x3: 64 channels only (got %d)", F);
  ChainArgs a{};
  a.in = dout; a.nlayers = 2 * nblocks; a.N = N; a.H = H; a.W = W; a.bwd = 1; a.slope = slope;
  // executed order: blocks nblocks-1 .. 0; layer 2j = conv2^T, 2j+1 = conv1^T of block k = nblocks-1-j
  for (int j = 0; j < nblocks; ++j) {
    const int k = nblocks - 1 - j;
    FDET_REQUIRE(h_wpk1b[k] && h_wpk2b[k] && h_a[k] && h_c[k] && h_dz1[k] && h_dz2[k], "block_chain_bwd_bf16x3: null pointer in block %d", k);
    a.w[2 * j] = (const bf16x8*)h_wpk2b[k]; a.w[2 * j + 1] = (const bf16x8*)h_wpk1b[k];
    a.ld[2 * j] = h_a[k];
    a.st[2 * j] = h_dz1[k];
    if (k > 0) {
      a.ld[2 * j + 1] = h_c[k - 1];
      a.sc[2 * j + 1] = h_scale ? h_scale[k - 1] : nullptr;
      a.st2[2 * j + 1] = h_dz2[k - 1];
    } else {
      a.st2[2 * j + 1] = dx;
    }
  }
  a.pre_ld = h_c[nblocks - 1];
  a.pre_sc = h_scale ? h_scale[nblocks - 1] : nullptr;
  a.pre_st = h_dz2[nblocks - 1];
  return launch_chain(a, (hipStream_t)stream);
}
