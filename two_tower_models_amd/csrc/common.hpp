// Shared helpers for the gfx950 kernels of libtt_hotpath.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tt_hotpath.h"

namespace tt {

void set_error(const char* fmt, ...);

inline int fail_arg(const char* what) {
  set_error("bad argument: %s", what);
  return TT_E_BADARG;
}

// Every launch is followed by this: launch-configuration errors surface
// immediately as the (positive) hipError_t; execution errors surface at the
// caller's next synchronisation, as HIP defines.
inline int check_launch(const char* kernel) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", kernel, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

inline hipStream_t S(tt_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// RAII timing scope around one kernel launch (no-op unless tt_profile_enable(1))
struct ProfScope {
  ProfScope(const char* name, hipStream_t st);
  ~ProfScope();
  hipStream_t st_;
  int slot_, idx_;
};

constexpr int WAVE = 64;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// carve 256-byte aligned pieces out of a caller-provided workspace
struct Carver {
  char* base;
  int64_t off = 0;
  explicit Carver(void* p) : base(reinterpret_cast<char*>(p)) {}
  template <typename T>
  T* take(int64_t count) {
    T* p = reinterpret_cast<T*>(base + off);
    off += round_up(count * (int64_t)sizeof(T), 256);
    return p;
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace tt
