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
#include "fdet_ldsdma.h"
#include "fdet_ps.h"
#include <cstdint>

using namespace fdet;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MAXL = 32;          // conv layers per launch (16 blocks)
constexpr int NTHR = 256;         // 4 waves, one per SIMD, each 64 positions (two 32-position tiles) x 64 channels
constexpr int NT = 2;             // position tiles per wave
constexpr int FCH = 64;           // channels
constexpr int A_UNITS = 9 * 2 * FCH;   // 16-byte units per weight array (hi or lo) per 16-channel chunk

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
  int nlayers, N, H, W, WP, PT, bwd, stagger;
  // PS I/O (fdet_ps.h): bit 0 = the tensors kept per layer (st / st2 / ld / pre_*) are PS image-0 pointers, except the
  // LAST st2 (chain output / input gradient), which stays fp32 NCHW; bit 1 = the forward input `in` is PS
  int psio, ps_hpwp, ps_wp, ps_plane, ps_img;                   // 16-byte units
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

typedef __attribute__((address_space(3))) void* lds_void_t;
#ifndef CH_DBG
#define CH_DBG 0            // timing ablations (wrong results): 1 no epilogue, 2 no deferred jobs, 4 no MFMAs
#endif
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// (an element of an ext-vector is copied to a scalar before any bit cast: see fdet_wgrad3x3_ps.hip)
__device__ __forceinline__ unsigned ch_elem(const u32x2& v, int k) { return k ? v.y : v.x; }

// PS = the per-block tensors are PS (ChainArgs::psio bit 0): that instantiation carries no fp32 tile of the kept
// activation, and spends the registers on the NEXT layer's bias (the accumulators start from it) and this layer's
// dropout scales, both loaded a layer ahead of their use
template <bool PS>
__global__ void __launch_bounds__(NTHR, 1)
k_block_chain_x3(const ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8* X = reinterpret_cast<bf16x8*>(smem);                 // [c16 4][hl 2][kh 2][PT] units
  const int PT = a.PT, WP = a.WP;
  bf16x8* Wb = X + 16 * PT;                                     // 2 buffers x [hi A_UNITS | lo A_UNITS]
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int n = blockIdx.x;
  const int HW = a.H * a.W;
  const bool qb0 = l31 & 1, qb1 = l31 & 2;
  // per position tile: this lane's position in the padded row space, its slot in the X arrays, and the transposed
  // arrangement for global I/O (4 consecutive positions of one row, one channel per lane of a quad)
  bool valid[NT];
  int xslot[NT], nv4[NT], ebase4[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int q = wid * (32 * NT) + j * 32 + l31;
    const int pr = q / WP, pc = q - pr * WP;
    valid[j] = pr < a.H && pc < a.W;
    xslot[j] = q + WP + 1;
    const int q4 = q & ~3, pr4 = q4 / WP, pc4 = q4 - pr4 * WP;
    nv4[j] = pr4 < a.H ? max(0, min(4, a.W - pc4)) : 0;
    ebase4[j] = (n * FCH + 4 * half + (l31 & 3)) * HW + (nv4[j] > 0 ? pr4 * a.W + pc4 : 0);   // + (32m + 8g)*HW
  }

  // PS addressing: byte offset of this lane's 8-byte half of the hi unit of (image n, group 0, its position); the unit of
  // channel group G = 4m + g is + G * ps_hpwp units, the lo plane + ps_plane units
  unsigned psb[NT];
  int ps_ops = 0;                                               // 8-byte accesses per plane of a tile issued by this wave
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int q = wid * (32 * NT) + j * 32 + l31;
    const int pr = q / WP, pc = q - pr * WP;
    psb[j] = (unsigned)(n * a.ps_img + (valid[j] ? pr * a.ps_wp + pc + 1 : 0)) * 16u + 8u * half;
    ps_ops += __builtin_amdgcn_ballot_w64(valid[j]) != 0 ? 8 : 0;
  }
  const unsigned ps_g = (unsigned)a.ps_hpwp * 16u, ps_lo = (unsigned)a.ps_plane * 16u;

  // ---- weight ring by LDS-DMA (fdet_ldsdma.h): the [hi | lo] image of one 16-channel chunk is 36 one-KiB pieces;
  // wave w moves plane w >> 1, half w & 1 (nine pieces).  Issued at the top of a chunk for the NEXT chunk (possibly the
  // next layer's first) into the buffer every wave left at the barrier before; `s_waitcnt vmcnt(0)` + barrier at the end.
  const unsigned lds_w = (unsigned)(size_t)(lds_void_t)smem + (unsigned)(16 * PT) * 16u;
  const unsigned wvoff = (unsigned)lane * 16u;
  const unsigned wpiece = (unsigned)((wid >> 1) * A_UNITS + (wid & 1) * 576) * 16u;            // within the LDS chunk image
  const unsigned wsrc = (unsigned)((wid >> 1) * 4 * A_UNITS + (wid & 1) * 576) * 16u;          // within the layer panel
#define CH_DMA_W(L, C16, BUF)                                                                      \
  {                                                                                                \
    const dma_u32x4 rs_ = dma_rsrc(a.w[L], 8u * A_UNITS * 16u);                                    \
    _Pragma("unroll") for (int k_ = 0; k_ < 9; ++k_)                                               \
      dma_piece(lds_w + (unsigned)(BUF) * (2 * A_UNITS * 16) + wpiece + k_ * 1024, wvoff, rs_,     \
                wsrc + (unsigned)(C16) * (A_UNITS * 16) + k_ * 1024);                              \
  }
  CH_DMA_W(0, 0, 0)
  // Workgroups would otherwise run in lockstep and send their tile stores to HBM in the same microsecond, layer after
  // layer: start them `stagger` x 64 cycles x (0..7) apart so that one workgroup's store burst meets the others' MFMAs.
  for (int i = ((blockIdx.x >> 3) & 7) * a.stagger; i > 0; --i) __builtin_amdgcn_s_sleep(1);

  {  // zero X (halo rows / pad columns stay zero for the whole chain)
    f32x4* z = reinterpret_cast<f32x4*>(smem);
    for (int t = tid; t < 16 * PT; t += NTHR) z[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // value of (m, r16) is channel 32m + 8*(r16>>2) + 4*half + (r16&3) at the lane's position
#define CH_OF(M, R) (32 * (M) + 8 * ((R) >> 2) + 4 * half + ((R) & 3))
  // write a full 64-channel register tile into the X arrays (bf16 hi/lo split), valid positions only
#define CH_WRITE_X(V)                                                                              \
  {                                                                                                \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                              \
      if (valid[j_]) {                                                                             \
        _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                           \
          _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                       \
            const float f_[4] = {V[j_][m_][4 * g_], V[j_][m_][4 * g_ + 1], V[j_][m_][4 * g_ + 2], V[j_][m_][4 * g_ + 3]}; \
            bf16x4 hi_, lo_;                                                                       \
            split4(f_, hi_, lo_);                                                                  \
            const int c16_ = 2 * m_ + (g_ >> 1), kh_ = g_ & 1;                                     \
            bf16x4* uh_ = reinterpret_cast<bf16x4*>(X + ((c16_ * 2 + 0) * 2 + kh_) * PT + xslot[j_]); \
            bf16x4* ul_ = reinterpret_cast<bf16x4*>(X + ((c16_ * 2 + 1) * 2 + kh_) * PT + xslot[j_]); \
            uh_[half] = hi_;                                                                       \
            ul_[half] = lo_;                                                                       \
          }                                                                                        \
      }                                                                                            \
  }
  // the same, and the hi / lo pieces also go to the PS tensor PTR (if not null): a lane's four channels are one half of
  // a 16-byte unit, 32 lanes = 32 consecutive units -- 512 contiguous bytes per instruction, no transposes
#define CH_WRITE_X_PS(V, PTR)                                                                      \
  {                                                                                                \
    char* __restrict__ pp_ = reinterpret_cast<char*>(PTR);                                         \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                              \
      if (valid[j_]) {                                                                             \
        _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                           \
          _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                       \
            const float f_[4] = {V[j_][m_][4 * g_], V[j_][m_][4 * g_ + 1], V[j_][m_][4 * g_ + 2], V[j_][m_][4 * g_ + 3]}; \
            bf16x4 hi_, lo_;                                                                       \
            split4(f_, hi_, lo_);                                                                  \
            const int c16_ = 2 * m_ + (g_ >> 1), kh_ = g_ & 1;                                     \
            bf16x4* uh_ = reinterpret_cast<bf16x4*>(X + ((c16_ * 2 + 0) * 2 + kh_) * PT + xslot[j_]); \
            bf16x4* ul_ = reinterpret_cast<bf16x4*>(X + ((c16_ * 2 + 1) * 2 + kh_) * PT + xslot[j_]); \
            uh_[half] = hi_;                                                                       \
            ul_[half] = lo_;                                                                       \
            if (pp_) {                                                                             \
              *reinterpret_cast<bf16x4*>(pp_ + psb[j_] + (4 * m_ + g_) * ps_g) = hi_;              \
              *reinterpret_cast<bf16x4*>(pp_ + psb[j_] + (4 * m_ + g_) * ps_g + ps_lo) = lo_;      \
            }                                                                                      \
          }                                                                                        \
      }                                                                                            \
  }
  // hi plane only (a tensor of which backward needs the signs, nothing else)
#define CH_STORE_HI(V, PTR)                                                                        \
  {                                                                                                \
    char* __restrict__ pp_ = reinterpret_cast<char*>(PTR);                                         \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                              \
      if (valid[j_]) {                                                                             \
        _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                           \
          _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                       \
            const bf16x4 hi_ = {(__bf16)V[j_][m_][4 * g_], (__bf16)V[j_][m_][4 * g_ + 1], (__bf16)V[j_][m_][4 * g_ + 2], (__bf16)V[j_][m_][4 * g_ + 3]}; \
            *reinterpret_cast<bf16x4*>(pp_ + psb[j_] + (4 * m_ + g_) * ps_g) = hi_;                \
          }                                                                                        \
      }                                                                                            \
  }
  // raw hi pieces of a PS tensor (two dwords per channel quad); CH_POS reads the sign of one value out of them
  // (float(hi) > 0  <=>  value > 0: the hi part carries the sign, and is zero only for a zero)
#define CH_LOAD_HI(DST, PTR)                                                                       \
  {                                                                                                \
    const char* __restrict__ pp_ = reinterpret_cast<const char*>(PTR);                             \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                              \
      _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                             \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_)                                           \
          DST[j_][m_][g_] = *reinterpret_cast<const u32x2*>(pp_ + psb[j_] + (4 * m_ + g_) * ps_g); \
  }
#define CH_POS(B, J, M, R) (__builtin_bit_cast(float, ((R) & 1) ? (ch_elem(B[J][M][(R) >> 2], ((R) >> 1) & 1) & 0xffff0000u) : (ch_elem(B[J][M][(R) >> 2], ((R) >> 1) & 1) << 16)) > 0.f)
  // both planes, joined to fp32 (forward input in PS)
#define CH_LOAD_PS(DST, PTR)                                                                       \
  {                                                                                                \
    const char* __restrict__ pp_ = reinterpret_cast<const char*>(PTR);                             \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                              \
      _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                             \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                         \
          const u32x2 h_ = *reinterpret_cast<const u32x2*>(pp_ + psb[j_] + (4 * m_ + g_) * ps_g);  \
          const u32x2 l_ = *reinterpret_cast<const u32x2*>(pp_ + psb[j_] + (4 * m_ + g_) * ps_g + ps_lo); \
          _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                         \
            DST[j_][m_][4 * g_ + i_] = valid[j_] ? ps_join(ch_elem(h_, i_ >> 1), ch_elem(l_, i_ >> 1), i_ & 1) : 0.f; \
        }                                                                                          \
  }
  // Global tile I/O, 16 bytes per lane: a 4x4 dword transpose across each quad of lanes turns
  // (lane = position, 4 registers = 4 consecutive channels) into (lane = channel, 4 registers = 4
  // consecutive positions of one row), so a tile moves in 8 instructions instead of 32 -- the
  // vector-memory instruction rate of the CU, not bytes, limits a burst of dword accesses.
  // After the transpose lane j of a quad owns channel 32m + 8g + 4*half + j at positions q4 .. q4+3.
#define CH_LD_(J, BYTES)                                                                           \
  _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                                 \
    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_)                                               \
      __builtin_memcpy(&t_[J][m_][g_], p_ + ebase4[J] + (32 * m_ + 8 * g_) * HW, BYTES);
  // the branch on the valid count sits OUTSIDE the tile loops: eight independent loads per arm
  // (a branch per load makes the compiler serialise every load behind an s_waitcnt)
#define CH_LOAD_TILE(DST, PTR)                                                                     \
  {                                                                                                \
    const float* __restrict__ p_ = (PTR);                                                          \
    f32x4 t_[NT][2][4];                                                                            \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                              \
      _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                             \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) t_[j_][m_][g_] = f32x4{0.f, 0.f, 0.f, 0.f}; \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) {                                            \
      if (nv4[j_] == 4) { CH_LD_(j_, 16) } else if (nv4[j_] == 3) { CH_LD_(j_, 12) } else if (nv4[j_] == 2) { CH_LD_(j_, 8) } else if (nv4[j_] == 1) { CH_LD_(j_, 4) } \
    }                                                                                              \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                              \
      _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                             \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                         \
          float v_[4] = {t_[j_][m_][g_][0], t_[j_][m_][g_][1], t_[j_][m_][g_][2], t_[j_][m_][g_][3]}; \
          quad_transpose4(v_, qb0, qb1);                                                           \
          _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) DST[j_][m_][4 * g_ + i_] = v_[i_];      \
        }                                                                                          \
  }
#define CH_ST_(J, BYTES)                                                                           \
  _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                                 \
    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_)                                               \
      __builtin_memcpy(p_ + ebase4[J] + (32 * m_ + 8 * g_) * HW, &t_[m_][g_], BYTES);
#define CH_STORE_TILE(PTR, V)                                                                      \
  {                                                                                                \
    float* __restrict__ p_ = (PTR);                                                                \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) {                                            \
      f32x4 t_[2][4];                                                                              \
      _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                             \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                         \
          float v_[4] = {V[j_][m_][4 * g_], V[j_][m_][4 * g_ + 1], V[j_][m_][4 * g_ + 2], V[j_][m_][4 * g_ + 3]}; \
          quad_transpose4(v_, qb0, qb1);                                                           \
          t_[m_][g_] = f32x4{v_[0], v_[1], v_[2], v_[3]};                                          \
        }                                                                                          \
      if (nv4[j_] == 4) { CH_ST_(j_, 16) } else if (nv4[j_] == 3) { CH_ST_(j_, 12) } else if (nv4[j_] == 2) { CH_ST_(j_, 8) } else if (nv4[j_] == 1) { CH_ST_(j_, 4) } \
    }                                                                                              \
  }

  // per-(image, channel) dropout scales in tile layout (1 when the pointer is null); the same for both position tiles
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
  // the same from a per-channel vector, four consecutive channels per load
#define CH_VEC_TILE(DST, PTR)                                                                      \
  {                                                                                                \
    const float* __restrict__ p_ = (PTR);                                                          \
    _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                               \
      _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                           \
        const f32x4 v_ = *reinterpret_cast<const f32x4*>(p_ + 32 * m_ + 8 * g_ + 4 * half);       \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) DST[m_][4 * g_ + i_] = v_[i_];            \
      }                                                                                            \
  }
#define CH_FOR_ALL _Pragma("unroll") for (int j = 0; j < NT; ++j) _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int r = 0; r < 16; ++r)
  f32x16 Hreg[NT][2];                                           // skip connection / running gradient
  f32x16 aux[NT][2];                                            // prefetched lrelu' operand (bwd)
  u32x2 auxb[NT][2][4];                                         // the same as raw PS hi pieces
  constexpr bool ps = PS;
  if (a.psio & 2) CH_LOAD_PS(Hreg, a.in) else CH_LOAD_TILE(Hreg, a.in)
  if (a.bwd) { if (ps) CH_LOAD_HI(auxb, a.pre_ld) else CH_LOAD_TILE(aux, a.pre_ld) }
  __syncthreads();                                              // zero fill done
  if (!a.bwd) {
    CH_WRITE_X(Hreg)
  } else {
    // dz2 of the first block to run = dout * scale * lrelu'(c)
    f32x16 t[NT][2], s2[2];
    CH_SCALE_TILE(s2, a.pre_sc)
    if (ps) {
      CH_FOR_ALL t[j][m][r] = Hreg[j][m][r] * s2[m][r] * (CH_POS(auxb, j, m, r) ? 1.f : a.slope);
      CH_WRITE_X_PS(t, a.pre_st)
    } else {
      CH_FOR_ALL t[j][m][r] = Hreg[j][m][r] * s2[m][r] * (aux[j][m][r] > 0.f ? 1.f : a.slope);
      CH_STORE_TILE(a.pre_st, t)
      CH_WRITE_X(t)
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // first weight chunk landed
  __syncthreads();
  CH_DMA_W(0, 1, 1)

  // Vector-memory instructions of one tile load / store issued by THIS wave: eight per arm of the valid-count branch
  // that has at least one lane (a lower bound is all the counted wait below needs).
  int tile_ops = 0;
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int k = 1; k <= 4; ++k) tile_ops += __builtin_amdgcn_ballot_w64(nv4[j] == k) != 0 ? 8 : 0;
  // vmcnt(K), K a runtime multiple of 8 (the immediate has 6 bits): wait until at most K vector-memory operations
  // are outstanding.  Memory operations retire in order, so with K <= the number issued AFTER a DMA, that DMA has landed.
#define CH_WAIT_YOUNG(K)                                                                           \
  switch (min((K), 56) >> 3) {                                                                     \
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;                                \
    case 1: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;                                \
    case 2: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;                               \
    case 3: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;                               \
    case 4: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;                               \
    case 5: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;                               \
    case 6: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;                               \
    default: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;                              \
  }
  int young = 0;                                                // tile loads / stores issued after the DMA of chunk 1

  int tapoff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) tapoff[t] = (t / 3) * WP + (t % 3);
  const int w_off = half * FCH + l31;                           // + tap*2*64 + m*32 ; lo: + A_UNITS
  const int x_off = half * PT + wid * (32 * NT) + l31;          // + j*32 + (c16*2 + hl)*2*PT + tapoff

  f32x16 bnext[2], s2p[2];                                      // PS: bias of the layer about to run; scales of this layer
  if (ps && !a.bwd) CH_VEC_TILE(bnext, a.bias[0])
  int stage = 0;
  for (int L = 0; L < a.nlayers; ++L) {
    f32x16 acc[NT][2];
    const bool odd = L & 1;
    if (ps && !a.bwd) {
      CH_FOR_ALL acc[j][m][r] = bnext[m][r];
      if (L + 1 < a.nlayers) CH_VEC_TILE(bnext, a.bias[L + 1])
    } else if (!a.bwd) {                                        // fp32 flavour: the same sum order (bias first), loaded here
      f32x16 b0[2];
      CH_VEC_TILE(b0, a.bias[L])
      CH_FOR_ALL acc[j][m][r] = b0[m][r];
    } else {
      CH_FOR_ALL acc[j][m][r] = 0.f;
    }
    if (ps && odd) {
      if (a.sc[L]) { CH_VEC_TILE(s2p, a.sc[L] + n * FCH) } else { _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int r = 0; r < 16; ++r) s2p[m][r] = 1.f; }
    }
    // the lrelu' operand of this layer's epilogue travels while the MFMAs run
    if (a.bwd && a.ld[L]) {
      if (ps) { CH_LOAD_HI(auxb, a.ld[L]) young += ps_ops; } else { CH_LOAD_TILE(aux, a.ld[L]) young += tile_ops; }
    }
    for (int c = 0; c < 4; ++c, ++stage) {
      // weight ring: chunk 1 of a layer is issued BEFORE the previous layer's epilogue (below), chunks 2, 3 and the
      // next layer's chunk 0 at the top of the chunk before them
      if (c == 1 || c == 2) CH_DMA_W(L, c + 1, (stage & 1) ^ 1)
      else if (c == 3 && L + 1 < a.nlayers) CH_DMA_W(L + 1, 0, (stage & 1) ^ 1)
      const bf16x8* Ww = Wb + (stage & 1) * 2 * A_UNITS + w_off;
      const bf16x8* Xh = X + (c * 2 + 0) * 2 * PT + x_off;
      const bf16x8* Xl = X + (c * 2 + 1) * 2 * PT + x_off;
      bf16x8 wh[2][2], wl[2][2], xh[2][NT], xl[2][NT];
#pragma unroll
      for (int m = 0; m < 2; ++m) { wh[0][m] = Ww[m * 32]; wl[0][m] = Ww[A_UNITS + m * 32]; }
#pragma unroll
      for (int j = 0; j < NT; ++j) { xh[0][j] = Xh[tapoff[0] + j * 32]; xl[0][j] = Xl[tapoff[0] + j * 32]; }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        if (t + 1 < 9) {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            wh[nxt][m] = Ww[(t + 1) * 2 * FCH + m * 32];
            wl[nxt][m] = Ww[A_UNITS + (t + 1) * 2 * FCH + m * 32];
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            xh[nxt][j] = Xh[tapoff[t + 1] + j * 32];
            xl[nxt][j] = Xl[tapoff[t + 1] + j * 32];
          }
        }
        __builtin_amdgcn_sched_barrier(0);                      // keep the fragment reads one tap ahead of their MFMAs
        // product-major: consecutive MFMAs go to different accumulators
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[j][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cur][m], xl[cur][j], acc[j][m], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[j][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[cur][m], xh[cur][j], acc[j][m], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[j][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cur][m], xh[cur][j], acc[j][m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // This wave's pieces of the next chunk have landed.  After chunk 0 that is chunk 1, issued before the previous
      // epilogue: only the operations OLDER than the epilogue's tile stores (and this layer's tile loads) are waited
      // for, so the stores drain into HBM behind the MFMAs instead of stalling every wave at the first chunk boundary.
      if (c == 0) { CH_WAIT_YOUNG(young) } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                          // chunk consumed by every wave; next one visible
    }
    young = 0;
    if (L + 1 < a.nlayers) CH_DMA_W(L + 1, 1, (stage & 1) ^ 1)  // buffer of chunk 3 is free: next layer's chunk 1

    // ---- epilogue: every wave is past its last read of X, so X can be overwritten in place
    const bool last = L + 1 == a.nlayers;
    if (!a.bwd) {
      CH_FOR_ALL { const float t = acc[j][m][r]; acc[j][m][r] = t > 0.f ? t : t * a.slope; }   // (the sums started from the bias)
      if (!odd) {                                               // a: next layer's input, kept for backward
        if (ps) { CH_WRITE_X_PS(acc, a.st[L]) young += a.st[L] ? 2 * ps_ops : 0; }
        else { if (a.st[L]) { CH_STORE_TILE(a.st[L], acc) young += tile_ops; } CH_WRITE_X(acc) }
      } else {                                                  // c (kept: fp32, or only its hi part = its signs), block output
        if (a.st[L]) { if (ps) { CH_STORE_HI(acc, a.st[L]) young += ps_ops; } else { CH_STORE_TILE(a.st[L], acc) young += tile_ops; } }
        f32x16 s2[2];
        if (ps) { s2[0] = s2p[0]; s2[1] = s2p[1]; } else CH_SCALE_TILE(s2, a.sc[L])
        CH_FOR_ALL Hreg[j][m][r] = acc[j][m][r] * s2[m][r] + Hreg[j][m][r];
        if (ps && !last) { CH_WRITE_X_PS(Hreg, a.st2[L]) young += a.st2[L] ? 2 * ps_ops : 0; }
        else {
          if (a.st2[L]) { CH_STORE_TILE(a.st2[L], Hreg) young += tile_ops; }
          if (!last) CH_WRITE_X(Hreg)
        }
      }
    } else {
      if (!odd) {                                               // conv2^T: dz1 = acc * lrelu'(a)
        if (ps) {
          CH_FOR_ALL acc[j][m][r] *= (CH_POS(auxb, j, m, r) ? 1.f : a.slope);
          CH_WRITE_X_PS(acc, a.st[L])
          young += 2 * ps_ops;
        } else {
          CH_FOR_ALL acc[j][m][r] *= (aux[j][m][r] > 0.f ? 1.f : a.slope);
          CH_STORE_TILE(a.st[L], acc)
          young += tile_ops;
          CH_WRITE_X(acc)
        }
      } else {                                                  // conv1^T: dx = acc + dout
        CH_FOR_ALL Hreg[j][m][r] += acc[j][m][r];
        if (!last) {                                            // dz2 of the next block to run
          f32x16 s2[2];
          if (ps) { s2[0] = s2p[0]; s2[1] = s2p[1]; } else CH_SCALE_TILE(s2, a.sc[L])
          if (ps) {
            CH_FOR_ALL acc[j][m][r] = Hreg[j][m][r] * s2[m][r] * (CH_POS(auxb, j, m, r) ? 1.f : a.slope);
            CH_WRITE_X_PS(acc, a.st2[L])
            young += 2 * ps_ops;
          } else {
            CH_FOR_ALL acc[j][m][r] = Hreg[j][m][r] * s2[m][r] * (aux[j][m][r] > 0.f ? 1.f : a.slope);
            CH_STORE_TILE(a.st2[L], acc)
            young += tile_ops;
            CH_WRITE_X(acc)
          }
        } else {
          CH_STORE_TILE(a.st2[L], Hreg)                         // gradient w.r.t. the chain input
        }
      }
    }
    __syncthreads();                                            // new X visible
  }
}

// ---- PS flavour, epilogue pipelined across the layer boundary --------------------------------------------------
// Layer L+1's chunk c only reads the 16-channel group c of layer L's output, and there is a workgroup barrier at the end
// of every chunk anyway.  So layer L's epilogue does its arithmetic, writes group 0 (X in LDS + the PS tensor in HBM) and
// passes the barrier; groups 1, 2, 3 wait in 48 registers and are split / written as four "jobs" per chunk (taps 1, 3,
// 5, 7) of layer L+1's chunks 0, 1, 2, behind its MFMAs.  Everything a job does is branch-free so that it can sit in one
// scheduling region with the MFMAs: lanes of pad positions write LDS to a per-lane dummy slot and HBM through a buffer
// descriptor at an out-of-range offset (dropped); a tensor that is not kept has an EMPTY descriptor.  Store counts are
// therefore exact -- 12 per piece -- and the waits for the weight ring are counted (vmcnt 12 / 24), not drains.
// P16 (precision16, see fdet_conv3x3_ps.hip): one MFMA pass on the hi planes; the lo planes are neither moved, kept in
// LDS nor written: a job is one LDS store and one buffer store, SJ = 4 stores per hand-over group instead of 8.
template <bool BWD, bool P16 = false>
__global__ void __launch_bounds__(NTHR, 1)
k_block_chain_ps(const ChainArgs a) {
  constexpr int SJ = P16 ? 4 : 8;                               // buffer stores of one group of jobs (NT tiles x 2 halves x planes)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8* X = reinterpret_cast<bf16x8*>(smem);                 // [c16 4][hl 2][kh 2][PT] units
  const int PT = a.PT, WP = a.WP;
  bf16x8* Wb = X + 16 * PT;                                     // 2 buffers x [hi A_UNITS | lo A_UNITS]
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int n = blockIdx.x;
  const int HW = a.H * a.W;
  const bool qb0 = l31 & 1, qb1 = l31 & 2;
  const unsigned dmy = (unsigned)(16 * PT + 4 * A_UNITS) * 16u + (unsigned)tid * 8u;   // this lane's dummy LDS slot
  float* const Lb = reinterpret_cast<float*>(smem + (size_t)(16 * PT + 4 * A_UNITS) * 16 + NTHR * 8);   // [layer][64] biases (fwd)
  float* const Ls = Lb + MAXL * FCH;                           // [layer / 2][64] dropout scales of this image (odd layers)
  bool valid[NT];
  int xslot[NT], nv4[NT], ebase4[NT];
  unsigned psb[NT], xw[NT], pso[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int q = wid * (32 * NT) + j * 32 + l31;
    const int pr = q / WP, pc = q - pr * WP;
    valid[j] = pr < a.H && pc < a.W;
    xslot[j] = q + WP + 1;
    const int q4 = q & ~3, pr4 = q4 / WP, pc4 = q4 - pr4 * WP;
    nv4[j] = pr4 < a.H ? max(0, min(4, a.W - pc4)) : 0;
    ebase4[j] = (n * FCH + 4 * half + (l31 & 3)) * HW + (nv4[j] > 0 ? pr4 * a.W + pc4 : 0);
    psb[j] = (unsigned)(n * a.ps_img + (valid[j] ? pr * a.ps_wp + pc + 1 : 0)) * 16u + 8u * half;
    xw[j] = (unsigned)xslot[j] * 16u + 8u * half;
    pso[j] = valid[j] ? psb[j] : 0x80000000u;
  }
  const unsigned ps_g = (unsigned)a.ps_hpwp * 16u, ps_lo = (unsigned)a.ps_plane * 16u;
  const int ps_bytes = (int)((unsigned)a.N * (unsigned)a.ps_img * 16u);

  const unsigned lds_w = (unsigned)(size_t)(lds_void_t)smem + (unsigned)(16 * PT) * 16u;
  const unsigned wvoff = (unsigned)lane * 16u;
  const unsigned wpiece = (unsigned)((wid >> 1) * A_UNITS + (wid & 1) * 576) * 16u;
  const unsigned wsrc = (unsigned)((wid >> 1) * 4 * A_UNITS + (wid & 1) * 576) * 16u;
  if (!(P16 && wid >= 2)) CH_DMA_W(0, 0, 0)
  {
    f32x4* z = reinterpret_cast<f32x4*>(smem);
    for (int t = tid; t < 16 * PT; t += NTHR) z[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = tid; t < a.nlayers * FCH; t += NTHR) {
      const int l_ = t >> 6, ch_ = t & 63;
      if (!BWD) Lb[t] = a.bias[l_][ch_];
      if (l_ & 1) Ls[(l_ >> 1) * FCH + ch_] = a.sc[l_] ? a.sc[l_][n * FCH + ch_] : 1.f;
    }
  }
  // a per-channel vector of the LDS tables in tile layout
#define CHP_LDS_TILE(DST, TAB)                                                                     \
  {                                                                                                \
    _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                               \
      _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                           \
        const f32x4 v_ = *reinterpret_cast<const f32x4*>((TAB) + 32 * m_ + 8 * g_ + 4 * half);    \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) DST[m_][4 * g_ + i_] = v_[i_];            \
      }                                                                                            \
  }

  // one job: four values (position tile J, 16-channel group P, k half HH) -> hi / lo split -> X, PS tensor, c's hi plane
#define CHP_SUB(F0, F1, F2, F3, J, P, HH)                                                     \
  {                                                                                                \
    const float f_[4] = {F0, F1, F2, F3};                                                          \
    unsigned hi_[2], lo_[2];                                                                       \
    if (P16) ps_hi4(f_, hi_); else ps_split4(f_, hi_, lo_);                                        \
    const unsigned ah_ = (unsigned)((((P) * 2 + 0) * 2 + (HH)) * PT) * 16u, al_ = (unsigned)((((P) * 2 + 1) * 2 + (HH)) * PT) * 16u; \
    *reinterpret_cast<u32x2*>(smem + (valid[J] ? ah_ + xw[J] : dmy)) = u32x2{hi_[0], hi_[1]};      \
    if (!P16) *reinterpret_cast<u32x2*>(smem + (valid[J] ? al_ + xw[J] : dmy)) = u32x2{lo_[0], lo_[1]}; \
    const unsigned go_ = pso[J] + (unsigned)(4 * ((P) >> 1) + 2 * ((P) & 1) + (HH)) * ps_g;        \
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{hi_[0], hi_[1]}, tgt_rs, go_, 0, 0);               \
    if (!P16) __builtin_amdgcn_raw_buffer_store_b64(u32x2{lo_[0], lo_[1]}, tgt_rs, go_, ps_lo, 0); \
  }
  // group 0 of a finished tile now; groups 1..3 into the pending registers
#define CHP_HAND_OVER(V)                                                                           \
  {                                                                                                \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                              \
      _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                             \
        CHP_SUB(V[j_][0][4 * h_], V[j_][0][4 * h_ + 1], V[j_][0][4 * h_ + 2], V[j_][0][4 * h_ + 3], j_, 0, h_) \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                              \
      _Pragma("unroll") for (int p_ = 1; p_ < 4; ++p_)                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) pend[j_][p_ - 1][i_] = V[j_][p_ >> 1][8 * (p_ & 1) + i_]; \
  }
  __amdgpu_buffer_rsrc_t tgt_rs = __builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0x00020000);
#define CHP_TARGET(PTR) __builtin_amdgcn_make_buffer_rsrc((void*)(PTR), 0, (PTR) ? ps_bytes : 0, 0x00020000)
  float pend[NT][3][8];
  int young = 0;                                                // stores issued after the DMA of the next layer's chunk 1

  f32x16 Hreg[NT][2];                                           // skip connection / running gradient
  u32x2 auxb[NT][2][4];                                         // raw PS hi pieces of the lrelu' operand (bwd)
  if (a.psio & 2) CH_LOAD_PS(Hreg, a.in) else CH_LOAD_TILE(Hreg, a.in)
  if (BWD) CH_LOAD_HI(auxb, a.pre_ld)
  __syncthreads();                                              // zero fill done
  if (!BWD) {
    CHP_HAND_OVER(Hreg)
  } else {                                                      // dz2 of the first block to run = dout * scale * lrelu'(c)
    f32x16 t[NT][2], s2[2];
    CH_SCALE_TILE(s2, a.pre_sc)
    CH_FOR_ALL t[j][m][r] = Hreg[j][m][r] * s2[m][r] * (CH_POS(auxb, j, m, r) ? 1.f : a.slope);
    tgt_rs = CHP_TARGET(a.pre_st);
    CHP_HAND_OVER(t)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // first weight chunk landed
  __syncthreads();
  if (!(P16 && wid >= 2)) CH_DMA_W(0, 1, 1)

  int tapoff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) tapoff[t] = (t / 3) * WP + (t % 3);
  const int w_off = half * FCH + l31;
  const int x_off = half * PT + wid * (32 * NT) + l31;

  for (int L = 0; L < a.nlayers; ++L) {
    f32x16 acc[NT][2];
    const bool odd = L & 1, last = L + 1 == a.nlayers;
    if (!BWD) {                                                 // the sums start from the bias
      f32x16 b0[2];
      CHP_LDS_TILE(b0, Lb + L * FCH)
      CH_FOR_ALL acc[j][m][r] = b0[m][r];
    } else {
      CH_FOR_ALL acc[j][m][r] = 0.f;
    }
    if (BWD && a.ld[L]) CH_LOAD_HI(auxb, a.ld[L])               // the lrelu' operand of this layer's epilogue
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // weight ring (buffer c & 1 holds chunk c): chunk 1 of a layer was issued before the previous epilogue
      if (!(P16 && wid >= 2)) {                                 // P16: the lo-plane waves move nothing
        if (c == 1 || c == 2) CH_DMA_W(L, c + 1, (c & 1) ^ 1)
        else if (c == 3 && !last) CH_DMA_W(L + 1, 0, 0)
      }
      const bf16x8* Ww = Wb + (c & 1) * 2 * A_UNITS + w_off;
      const bf16x8* Xh = X + (c * 2 + 0) * 2 * PT + x_off;
      const bf16x8* Xl = X + (c * 2 + 1) * 2 * PT + x_off;
      bf16x8 wh[2][2], wl[2][2], xh[2][NT], xl[2][NT];
#pragma unroll
      for (int m = 0; m < 2; ++m) { wh[0][m] = Ww[m * 32]; if (!P16) wl[0][m] = Ww[A_UNITS + m * 32]; }
#pragma unroll
      for (int j = 0; j < NT; ++j) { xh[0][j] = Xh[tapoff[0] + j * 32]; if (!P16) xl[0][j] = Xl[tapoff[0] + j * 32]; }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        if (t + 1 < 9) {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            wh[nxt][m] = Ww[(t + 1) * 2 * FCH + m * 32];
            if (!P16) wl[nxt][m] = Ww[A_UNITS + (t + 1) * 2 * FCH + m * 32];
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            xh[nxt][j] = Xh[tapoff[t + 1] + j * 32];
            if (!P16) xl[nxt][j] = Xl[tapoff[t + 1] + j * 32];
          }
        }
        __builtin_amdgcn_sched_barrier(0);                      // keep the fragment reads one tap ahead of their MFMAs
        if (CH_DBG & 4) { acc[0][0][0] += (float)wh[cur][0][0] + (float)xh[cur][0][0] + (float)wh[cur][1][0] + (float)xh[cur][1][0]; } else {
        if (!P16) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[j][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cur][m], xl[cur][j], acc[j][m], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[j][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[cur][m], xh[cur][j], acc[j][m], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[j][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cur][m], xh[cur][j], acc[j][m], 0, 0, 0);
        }
        if (!(CH_DBG & 2) && c < 3 && (t & 1)) {                // a pending job of the previous layer's group c + 1
          constexpr int dummy_ = 0; (void)dummy_;
          const int jj = (t - 1) >> 2, hh = ((t - 1) >> 1) & 1;
          CHP_SUB(pend[jj][c][4 * hh], pend[jj][c][4 * hh + 1], pend[jj][c][4 * hh + 2], pend[jj][c][4 * hh + 3], jj, c + 1, hh)
          if (!P16) {
#pragma unroll
          for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            if (i == 8 || i == 9) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            if (i >= 10) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
          }
          } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            if (i == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            if (i == 3) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
          }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // this wave's pieces of the next chunk have landed: everything issued after them is the 8 job stores of this chunk
      // (chunks 0..2) and, after chunk 0, the `young` stores of the previous epilogue (8 of group 0, 16 more for c's hi plane)
      if (c == 0) {
        if (young == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SJ) : "memory");
        else if (young == SJ) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SJ) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SJ + 16) : "memory");
      }
      else if (c < 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SJ) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                          // chunk consumed by every wave; next one visible
    }
    if (!last && !(P16 && wid >= 2)) CH_DMA_W(L + 1, 1, 1)      // buffer of chunk 3 is free: next layer's chunk 1

    // ---- epilogue arithmetic for the whole tile (in place), then group 0 now and groups 1..3 handed to the next layer
    young = SJ;
    if (CH_DBG & 1) {
    } else if (!BWD) {
      CH_FOR_ALL { const float t = acc[j][m][r]; acc[j][m][r] = t > 0.f ? t : t * a.slope; }
      if (!odd) {
        tgt_rs = CHP_TARGET(a.st[L]);
        CHP_HAND_OVER(acc)
      } else {
        // c: only its hi plane (its signs) is kept -- stored here for the whole tile, 16 exact-count stores
        const __amdgpu_buffer_rsrc_t crs = CHP_TARGET(a.st[L]);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
              const bf16x2_t h0 = {(__bf16)acc[j][m][4 * g], (__bf16)acc[j][m][4 * g + 1]}, h1 = {(__bf16)acc[j][m][4 * g + 2], (__bf16)acc[j][m][4 * g + 3]};
              __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)}, crs,
                                                    pso[j] + (unsigned)(4 * m + g) * ps_g, 0, 0);
            }
        f32x16 s2[2];
        CHP_LDS_TILE(s2, Ls + (L >> 1) * FCH)
        CH_FOR_ALL Hreg[j][m][r] = acc[j][m][r] * s2[m][r] + Hreg[j][m][r];
        if (last) {
          CH_STORE_TILE(a.st2[L], Hreg)                         // chain output, fp32 NCHW
        } else {
          tgt_rs = CHP_TARGET(a.st2[L]);
          CHP_HAND_OVER(Hreg)
          young = SJ + 16;
        }
      }
    } else {
      if (!odd) {                                               // conv2^T: dz1 = acc * lrelu'(a)
        CH_FOR_ALL acc[j][m][r] *= (CH_POS(auxb, j, m, r) ? 1.f : a.slope);
        tgt_rs = CHP_TARGET(a.st[L]);
        CHP_HAND_OVER(acc)
      } else {                                                  // conv1^T: dx = acc + dout
        CH_FOR_ALL Hreg[j][m][r] += acc[j][m][r];
        if (last) {
          CH_STORE_TILE(a.st2[L], Hreg)                         // gradient w.r.t. the chain input, fp32 NCHW
        } else {                                                // dz2 of the next block to run
          f32x16 s2[2];
          CHP_LDS_TILE(s2, Ls + (L >> 1) * FCH)
          CH_FOR_ALL acc[j][m][r] = Hreg[j][m][r] * s2[m][r] * (CH_POS(auxb, j, m, r) ? 1.f : a.slope);
          tgt_rs = CHP_TARGET(a.st2[L]);
          CHP_HAND_OVER(acc)
        }
      }
    }
    __syncthreads();                                            // group 0 of the new X visible
  }
}


int chain_geometry(int F, int H, int W, int& WP, int& PT, size_t& lds) {
  if (F != FCH || H <= 0 || W <= 0) return 0;
  WP = (W + 1 + 3) / 4 * 4;
  if (H * WP > 4 * NT * 32) return 0;                           // 4 waves x 64 positions
  PT = (H + 2) * WP + 3;
  if (PT < 4 * NT * 32 + 2 * WP + 3) PT = 4 * NT * 32 + 2 * WP + 3;   // garbage positions of the last wave read in range
  lds = ((size_t)16 * PT + 2 * 2 * A_UNITS) * 16 + (size_t)NTHR * 8 + (size_t)(MAXL + MAXL / 2) * FCH * 4;   // + one 8-byte dummy slot per lane, bias and scale tables (PS flavour)
  return lds <= 160 * 1024;
}

int launch_chain(ChainArgs& a, hipStream_t st) {
  size_t lds = 0;
  if (!chain_geometry(FCH, a.H, a.W, a.WP, a.PT, lds))
    return fail(FDET_EINVAL, "block_chain_bf16x3: unsupported map %dx%d (needs 64 channels and H*roundup4(W+1) <= 256)", a.H, a.W);
  if ((size_t)a.N * FCH * a.H * a.W >= ((size_t)1 << 31)) return fail(FDET_EINVAL, "block_chain_bf16x3: tensor too large");
  { const char* e_ = FDET_ENV_ONCE("FDET_CHAIN_STAGGER"); a.stagger = e_ ? atoi(e_) : 0; }
  if ((a.psio & 1) && (a.psio & 4) && a.bwd) {             // psio bit 2: precision16
    { if (int rc_ = set_lds_attr((const void*)k_block_chain_ps<true, true>, (size_t)(lds), __func__)) return rc_; }
    hipLaunchKernelGGL((k_block_chain_ps<true, true>), dim3(a.N), dim3(NTHR), lds, st, a);
  } else if ((a.psio & 1) && (a.psio & 4)) {
    { if (int rc_ = set_lds_attr((const void*)k_block_chain_ps<false, true>, (size_t)(lds), __func__)) return rc_; }
    hipLaunchKernelGGL((k_block_chain_ps<false, true>), dim3(a.N), dim3(NTHR), lds, st, a);
  } else if ((a.psio & 1) && a.bwd) {
    { if (int rc_ = set_lds_attr((const void*)k_block_chain_ps<true>, (size_t)(lds), __func__)) return rc_; }
    hipLaunchKernelGGL(k_block_chain_ps<true>, dim3(a.N), dim3(NTHR), lds, st, a);
  } else if (a.psio & 1) {
    { if (int rc_ = set_lds_attr((const void*)k_block_chain_ps<false>, (size_t)(lds), __func__)) return rc_; }
    hipLaunchKernelGGL(k_block_chain_ps<false>, dim3(a.N), dim3(NTHR), lds, st, a);
  } else {
    { if (int rc_ = set_lds_attr((const void*)k_block_chain_x3<false>, (size_t)(lds), __func__)) return rc_; }
    hipLaunchKernelGGL(k_block_chain_x3<false>, dim3(a.N), dim3(NTHR), lds, st, a);
  }
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

// ---- PS flavour (fdet_ps.h): the tensors kept per block are PS image-0 pointers -- a_k and the block outputs with both
// planes (operands of the weight gradients), c_k with its hi plane only (backward needs its signs); dz1_k / dz2_k are
// written as PS.  The chain output (last block) and the input gradient stay fp32 NCHW; the forward input is PS when
// x_is_ps.  h_out_ps has nblocks - 1 entries (blocks 0 .. nblocks-2); any of h_a_ps / h_c_ps / h_out_ps may be null
// (inference: nothing kept).
namespace {
int chain_ps_geo(ChainArgs& a, int N, int H, int W) {
  PsGeo g;
  if (!ps_geo(N, FCH, H, W, g) || (unsigned long long)(N + 1) * g.img * 16ull >= (1ull << 31)) return 0;
  a.ps_hpwp = g.HP * g.WP; a.ps_wp = g.WP; a.ps_plane = g.plane; a.ps_img = g.img;
  return 1;
}
}  // namespace

namespace {
int chain_fwd_ps_run(const void* x, int x_is_ps, const void* const* h_wpk1, const float* const* h_b1,
                     const void* const* h_wpk2, const float* const* h_b2, const float* const* h_scale,
                     void* const* h_a_ps, void* const* h_c_ps, void* const* h_out_ps, float* out_last,
                     int nblocks, int N, int F, int H, int W, float slope, void* stream, bool p16) {
  FDET_REQUIRE(x && h_wpk1 && h_b1 && h_wpk2 && h_b2 && out_last, "block_chain_fwd_ps: null pointer");
  FDET_REQUIRE(nblocks >= 1 && 2 * nblocks <= MAXL && N > 0, "block_chain_fwd_ps: 1..%d blocks (got %d), N=%d", MAXL / 2, nblocks, N);
  FDET_REQUIRE(F == FCH, "block_chain_fwd_ps: 64 channels only (got %d)", F);
  ChainArgs a{};
  FDET_REQUIRE(chain_ps_geo(a, N, H, W), "block_chain_fwd_ps: no PS layout for N=%d %dx%d", N, H, W);
  a.in = reinterpret_cast<const float*>(x); a.nlayers = 2 * nblocks; a.N = N; a.H = H; a.W = W; a.bwd = 0; a.slope = slope;
  a.psio = 1 | (x_is_ps ? 2 : 0) | (p16 ? 4 : 0);
  for (int k = 0; k < nblocks; ++k) {
    FDET_REQUIRE(h_wpk1[k] && h_wpk2[k] && h_b1[k] && h_b2[k], "block_chain_fwd_ps: null weights in block %d", k);
    a.w[2 * k] = (const bf16x8*)h_wpk1[k]; a.w[2 * k + 1] = (const bf16x8*)h_wpk2[k];
    a.bias[2 * k] = h_b1[k]; a.bias[2 * k + 1] = h_b2[k];
    a.sc[2 * k + 1] = h_scale ? h_scale[k] : nullptr;
    a.st[2 * k] = h_a_ps ? reinterpret_cast<float*>(h_a_ps[k]) : nullptr;
    a.st[2 * k + 1] = h_c_ps ? reinterpret_cast<float*>(h_c_ps[k]) : nullptr;
    a.st2[2 * k + 1] = k + 1 < nblocks ? (h_out_ps ? reinterpret_cast<float*>(h_out_ps[k]) : nullptr) : out_last;
  }
  return launch_chain(a, (hipStream_t)stream);
}

int chain_bwd_ps_run(const float* dout, const void* const* h_wpk1b, const void* const* h_wpk2b,
                     const float* const* h_scale, const void* const* h_a_ps, const void* const* h_c_ps,
                     void* const* h_dz1_ps, void* const* h_dz2_ps, float* dx, int nblocks, int N, int F,
                     int H, int W, float slope, void* stream, bool p16) {
  FDET_REQUIRE(dout && h_wpk1b && h_wpk2b && h_a_ps && h_c_ps && h_dz1_ps && h_dz2_ps && dx, "block_chain_bwd_ps: null pointer");
  FDET_REQUIRE(nblocks >= 1 && 2 * nblocks <= MAXL && N > 0, "block_chain_bwd_ps: 1..%d blocks (got %d), N=%d", MAXL / 2, nblocks, N);
  FDET_REQUIRE(F == FCH, "block_chain_bwd_ps: 64 channels only (got %d)", F);
  ChainArgs a{};
  FDET_REQUIRE(chain_ps_geo(a, N, H, W), "block_chain_bwd_ps: no PS layout for N=%d %dx%d", N, H, W);
  a.in = dout; a.nlayers = 2 * nblocks; a.N = N; a.H = H; a.W = W; a.bwd = 1; a.slope = slope; a.psio = 1 | (p16 ? 4 : 0);
  for (int j = 0; j < nblocks; ++j) {
    const int k = nblocks - 1 - j;
    FDET_REQUIRE(h_wpk1b[k] && h_wpk2b[k] && h_a_ps[k] && h_c_ps[k] && h_dz1_ps[k] && h_dz2_ps[k], "block_chain_bwd_ps: null pointer in block %d", k);
    a.w[2 * j] = (const bf16x8*)h_wpk2b[k]; a.w[2 * j + 1] = (const bf16x8*)h_wpk1b[k];
    a.ld[2 * j] = reinterpret_cast<const float*>(h_a_ps[k]);
    a.st[2 * j] = reinterpret_cast<float*>(h_dz1_ps[k]);
    if (k > 0) {
      a.ld[2 * j + 1] = reinterpret_cast<const float*>(h_c_ps[k - 1]);
      a.sc[2 * j + 1] = h_scale ? h_scale[k - 1] : nullptr;
      a.st2[2 * j + 1] = reinterpret_cast<float*>(h_dz2_ps[k - 1]);
    } else {
      a.st2[2 * j + 1] = dx;
    }
  }
  a.pre_ld = reinterpret_cast<const float*>(h_c_ps[nblocks - 1]);
  a.pre_sc = h_scale ? h_scale[nblocks - 1] : nullptr;
  a.pre_st = reinterpret_cast<float*>(h_dz2_ps[nblocks - 1]);
  return launch_chain(a, (hipStream_t)stream);
}
}  // namespace

extern "C" int fdet_block_chain_fwd_ps(const void* x, int x_is_ps, const void* const* h_wpk1, const float* const* h_b1,
                                       const void* const* h_wpk2, const float* const* h_b2, const float* const* h_scale,
                                       void* const* h_a_ps, void* const* h_c_ps, void* const* h_out_ps, float* out_last,
                                       int nblocks, int N, int F, int H, int W, float slope, void* stream) {
  return chain_fwd_ps_run(x, x_is_ps, h_wpk1, h_b1, h_wpk2, h_b2, h_scale, h_a_ps, h_c_ps, h_out_ps, out_last, nblocks, N, F, H, W,
                          slope, stream, false);
}
extern "C" int fdet_block_chain_bwd_ps(const float* dout, const void* const* h_wpk1b, const void* const* h_wpk2b,
                                       const float* const* h_scale, const void* const* h_a_ps, const void* const* h_c_ps,
                                       void* const* h_dz1_ps, void* const* h_dz2_ps, float* dx, int nblocks, int N, int F,
                                       int H, int W, float slope, void* stream) {
  return chain_bwd_ps_run(dout, h_wpk1b, h_wpk2b, h_scale, h_a_ps, h_c_ps, h_dz1_ps, h_dz2_ps, dx, nblocks, N, F, H, W, slope,
                          stream, false);
}
// precision16 (one bf16 MFMA pass on the hi planes; the tensors kept per block receive their hi plane only)
extern "C" int fdet_block_chain_fwd_ps_p16(const void* x, int x_is_ps, const void* const* h_wpk1, const float* const* h_b1,
                                           const void* const* h_wpk2, const float* const* h_b2, const float* const* h_scale,
                                           void* const* h_a_ps, void* const* h_c_ps, void* const* h_out_ps, float* out_last,
                                           int nblocks, int N, int F, int H, int W, float slope, void* stream) {
  return chain_fwd_ps_run(x, x_is_ps, h_wpk1, h_b1, h_wpk2, h_b2, h_scale, h_a_ps, h_c_ps, h_out_ps, out_last, nblocks, N, F, H, W,
                          slope, stream, true);
}
extern "C" int fdet_block_chain_bwd_ps_p16(const float* dout, const void* const* h_wpk1b, const void* const* h_wpk2b,
                                           const float* const* h_scale, const void* const* h_a_ps, const void* const* h_c_ps,
                                           void* const* h_dz1_ps, void* const* h_dz2_ps, float* dx, int nblocks, int N, int F,
                                           int H, int W, float slope, void* stream) {
  return chain_bwd_ps_run(dout, h_wpk1b, h_wpk2b, h_scale, h_a_ps, h_c_ps, h_dz1_ps, h_dz2_ps, dx, nblocks, N, F, H, W, slope,
                          stream, true);
}
