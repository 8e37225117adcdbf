// Input preprocessing on the device: bilinear Resize fused with the /255 normalisation.
// Replaces `self.resize(x) / 255.0` of models/PoolResnet.py:91,95 (models/Resnet.py likewise) and
// `Resize(...)(x); x / 255.0` of models/BaseModel.py:64-65, i.e. torchvision 0.11.2
// transforms.Resize on tensors: uint8 -> float32, F.interpolate(mode="bilinear",
// align_corners=False, no antialias), round-half-even, cast back to uint8 -- then the model
// divides by 255.  Feeding uint8 frames and resizing on the device keeps the host->device copy at
// 1 byte per pixel.
//
// Index arithmetic follows ATen's upsample_bilinear2d (area_pixel_compute_source_index with
// align_corners=False, guard_index_and_lambda): fp32 throughout,
//   src = ratio*(o + 0.5) - 0.5 (one fused multiply-add, as the CPU build contracts it), clamped at 0,
//   i0 = min(int(src), in-1), i1 = i0 + (i0 < in-1), l1 = clamp(src - i0, 0, 1), l0 = 1 - l1,
//   out = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11);  same-size dimensions are the identity.
// HBM-bound and tiny next to the conv stack: one thread per output pixel, lanes along W.
#include "fdet_common.h"
#include <cstdint>

using namespace fdet;

namespace {

struct AxisMap { int i0, i1; float l0, l1; };

__device__ __forceinline__ AxisMap axis_map(int o, int in, int out, float ratio) {
  AxisMap m;
  if (in == out) { m.i0 = o; m.i1 = o; m.l0 = 1.f; m.l1 = 0.f; return m; }
  float src = fmaf(ratio, (float)o + 0.5f, -0.5f);
  src = src < 0.f ? 0.f : src;
  m.i0 = min((int)src, in - 1);
  m.i1 = m.i0 + (m.i0 < in - 1 ? 1 : 0);
  m.l1 = fminf(fmaxf(src - (float)m.i0, 0.f), 1.f);
  m.l0 = 1.f - m.l1;
  return m;
}

template <typename T, bool ROUND_U8>
__global__ void __launch_bounds__(256)
k_resize_bilinear_norm(const T* __restrict__ src, float* __restrict__ dst, int planes, int Hs, int Ws, int Hd, int Wd,
                       float ratio_h, float ratio_w, float divisor) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x;
  const int oy = blockIdx.y;
  if (ox >= Wd) return;
  const AxisMap mh = axis_map(oy, Hs, Hd, ratio_h);
  const AxisMap mw = axis_map(ox, Ws, Wd, ratio_w);
  for (int pl = blockIdx.z; pl < planes; pl += gridDim.z) {
    const T* s = src + (size_t)pl * Hs * Ws;
    const float v00 = (float)s[(size_t)mh.i0 * Ws + mw.i0], v01 = (float)s[(size_t)mh.i0 * Ws + mw.i1];
    const float v10 = (float)s[(size_t)mh.i1 * Ws + mw.i0], v11 = (float)s[(size_t)mh.i1 * Ws + mw.i1];
    const float top = mw.l0 * v00 + mw.l1 * v01;
    const float bot = mw.l0 * v10 + mw.l1 * v11;
    float v = mh.l0 * top + mh.l1 * bot;
    if (ROUND_U8) v = fminf(fmaxf(rintf(v), 0.f), 255.f);      // round-half-even, then the uint8 cast
    dst[((size_t)pl * Hd + oy) * Wd + ox] = v / divisor;
  }
}

template <typename T, bool ROUND_U8>
int run_resize(const T* src, float* dst, int N, int C, int Hs, int Ws, int Hd, int Wd, float divisor, hipStream_t st) {
  const int planes = N * C;
  dim3 grid((Wd + 255) / 256, Hd, planes < 64 ? planes : 64);
  hipLaunchKernelGGL((k_resize_bilinear_norm<T, ROUND_U8>), grid, dim3(256), 0, st, src, dst, planes, Hs, Ws, Hd, Wd,
                     (float)Hs / (float)Hd, (float)Ws / (float)Wd, divisor);
  return check_launch("fdet_resize_bilinear");
}

}  // namespace

extern "C" int fdet_resize_bilinear_u8_norm(const uint8_t* src, float* dst, int N, int C, int Hs, int Ws, int Hd,
                                            int Wd, void* stream) {
  FDET_REQUIRE(src && dst, "resize_bilinear_u8_norm: null pointer");
  FDET_REQUIRE(N > 0 && C > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && Hd <= 65535,
               "resize_bilinear_u8_norm: bad shape N=%d C=%d %dx%d -> %dx%d", N, C, Hs, Ws, Hd, Wd);
  return run_resize<uint8_t, true>(src, dst, N, C, Hs, Ws, Hd, Wd, 255.0f, (hipStream_t)stream);
}

extern "C" int fdet_resize_bilinear_f32_norm(const float* src, float* dst, int N, int C, int Hs, int Ws, int Hd,
                                             int Wd, float divisor, void* stream) {
  FDET_REQUIRE(src && dst, "resize_bilinear_f32_norm: null pointer");
  FDET_REQUIRE(N > 0 && C > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && Hd <= 65535 && divisor != 0.f,
               "resize_bilinear_f32_norm: bad shape N=%d C=%d %dx%d -> %dx%d", N, C, Hs, Ws, Hd, Wd);
  return run_resize<float, false>(src, dst, N, C, Hs, Ws, Hd, Wd, divisor, (hipStream_t)stream);
}
