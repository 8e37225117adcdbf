// Shared host-side helpers for libfdet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../include/fdet.h"

namespace fdet {

inline char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(FDET_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return FDET_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) with its result checked (a kernel that needs more than 64 KB of LDS
// fails at launch with an unhelpful error otherwise)
inline int set_lds_attr(const void* kern, size_t lds, const char* what) {
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    (void)hipGetLastError();
    return fail(FDET_ELAUNCH, "%s: cannot reserve %zu bytes of LDS", what, lds);
  }
  return FDET_OK;
}
// environment switch read once per call site (launch paths are hot on the launch-bound demo path)
#define FDET_ENV_ONCE(NAME) ([]() -> const char* { static const char* v_ = getenv(NAME); return v_; }())

constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;   // valid in lane 0
}
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace fdet

#define FDET_REQUIRE(cond, ...) \
  do { if (!(cond)) return fdet::fail(FDET_EINVAL, __VA_ARGS__); } while (0)
