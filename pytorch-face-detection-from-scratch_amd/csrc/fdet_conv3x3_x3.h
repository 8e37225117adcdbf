// Shared declarations of the bf16x3 conv kernels (fdet_conv3x3_x3*.hip).
#pragma once
#include "fdet_conv_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace fdet {
struct X3Args {
  ConvArgs c;                // x, bias, epilogue pointers, geometry (WP, R, VR, nbands, mode ...)
  const bf16x8* a_hi;        // [Cin/16][9][2][CoP] x 8 bf16
  const bf16x8* a_lo;
  int PT;                    // positions per activation array (cap + 2*WP + 3)
  int CW, NSEG;              // columns per segment, segments per row
  int ncob, ntiles;          // output-channel blocks, tiles = nbands*NSEG*ncob
  int items_half;            // (R+2)*(CW/VW) staging items per k-half
  int WV;                    // CW / VW
  unsigned magic_wv, magic_wp;
  long long* stamps;         // diagnostic builds (-DFDET_X3_STAMPS) only; null otherwise
};
}  // namespace fdet

using namespace fdet;

namespace {

constexpr int CK16 = 16;     // input channels per chunk (= MFMA K)
constexpr int TW0 = 3;       // first tap whose MFMAs share the pipe with the next chunk's LDS writes

// activation staging slots per thread: one slot = VW positions x 8 channels (8 loads of VW floats).
// Sized for segments of up to ~92 columns; 2*(R+2)*(CW/VW) <= nbs*threads is verified on the host.
__host__ __device__ constexpr int nbs_of(int nw, int nt, int vw) {
  return (2 * (nw * nt * 32 + 184) / vw + nw * 64 - 1) / (nw * 64);
}


__device__ __forceinline__ void split8(const float (&f)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)f[j];
    hi[j] = h;
    lo[j] = (__bf16)(f[j] - (float)h);
  }
}


}  // namespace

namespace fdet {
// pooled-block fusion of the ping-pong kernel (fdet_conv3x3_x3_pp.hip); all null = plain epilogue modes
struct PoolArgs {
  float* pool_out;                  // EPI_FWD_POOL: [N,Cout,H/2,W/2] = maxpool2x2(lrelu(conv+bias)*scale + skip)
  unsigned char* mask_out;          // EPI_FWD_POOL (training): [N,Cout,H/2,W/2] routing bytes, may be null
  const float* pool_din;            // EPI_DGRAD_ADDPOOL: gradient of the pooled block output [N,Cout,H/2,W/2]
  const unsigned char* mask_in;     // EPI_DGRAD_ADDPOOL: routing bytes of the forward pass
};
}  // namespace fdet

// ping-pong variant for rows of <= 63 columns (fdet_conv3x3_x3_pp.hip); returns 1 when it has no tiling
int fdet_x3_pp_run(fdet::ConvArgs a, fdet::PoolArgs q, hipStream_t st);

// small-tile single-buffer variant (fdet_conv3x3_x3_sb.hip); returns 1 when it has no tiling
int fdet_x3_sb_run(fdet::ConvArgs a, hipStream_t st);
// ... its aligned-band variant with the pooled-block epilogues
int fdet_x3_sb_pool_run(fdet::ConvArgs a, fdet::PoolArgs q, hipStream_t st);

// one translation unit per epilogue mode (fdet_conv3x3_x3_m<MODE>.hip): picks the kernel
// instantiation for (MT, NW, NT, VW, seg) and launches it
#define X3_DECL_LAUNCH(M_) int fdet_x3_launch_m##M_(const X3Args& p, int MT, int NW, int NT, int VW, bool seg, size_t lds, int grid, hipStream_t st);
X3_DECL_LAUNCH(0) X3_DECL_LAUNCH(1) X3_DECL_LAUNCH(2) X3_DECL_LAUNCH(3) X3_DECL_LAUNCH(4) X3_DECL_LAUNCH(5)
#undef X3_DECL_LAUNCH
