// Does the achievable fp32 MFMA rate depend on the operand DATA (power / clocks)?  Same register-only
// loop as mfma_peak.hip, operands either constant small integers or full-entropy floats that change
// every iteration.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_data.hip -o /tmp/mpd && /tmp/mpd
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool RANDOM>
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c)
    for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  float a[8], b[8];
  for (int k = 0; k < 8; ++k) {
    s = s * 1664525u + 1013904223u;
    a[k] = RANDOM ? __uint_as_float(0x3f000000u | (s >> 9)) - 0.75f : (float)(threadIdx.x & 3);
    s = s * 1664525u + 1013904223u;
    b[k] = RANDOM ? __uint_as_float(0x3f000000u | (s >> 9)) - 0.75f : 1.0f;
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + c) & 7], acc[c], 0, 0, 0);
  }
  float t = 0.f;
  for (int c = 0; c < 4; ++c)
    for (int e = 0; e < 16; ++e) t += acc[c][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <bool RANDOM>
void run(const char* name, int ms_target) {
  float* out;
  const int blocks = 512;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  spin<RANDOM><<<blocks, 256>>>(out, 100);
  hipDeviceSynchronize();
  for (int iters : {2000, 20000, 200000}) {
    hipEventRecord(e0);
    spin<RANDOM><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 32 * (2.0 * 32 * 32 * 2);
    printf("%s: %8.2f ms  %.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
  }
  hipFree(out);
}

int main() {
  run<false>("constant operands", 0);
  run<true>("random operands  ", 0);
  return 0;
}
