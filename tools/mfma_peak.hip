// Achievable MFMA issue rate on this GPU: register-only loops of independent accumulator chains.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, bool BF16>
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int e = 0; e < 16; ++e) acc[c][e] = (float)(threadIdx.x + c);
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 7); b[e] = (__bf16)1.0f; }
  const float fa = (float)(threadIdx.x & 3), fb = 1.0f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (BF16) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
      else acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[c], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int e = 0; e < 16; ++e) s += acc[c][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS, bool BF16>
void run(const char* name, int wg_per_cu) {
  float* out;
  const int blocks = 256 * wg_per_cu, iters = 20000;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  spin<CHAINS, BF16><<<blocks, 256>>>(out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  spin<CHAINS, BF16><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 4 * iters * CHAINS * (BF16 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2);
  printf("%s chains=%d waves/SIMD=%d: %.1f TFLOP/s\n", name, CHAINS, wg_per_cu, flop / ms / 1e9);
  hipFree(out);
}

int main() {
  run<1, true>("bf16 32x32x16", 1); run<2, true>("bf16 32x32x16", 1); run<4, true>("bf16 32x32x16", 1);
  run<2, true>("bf16 32x32x16", 2); run<4, true>("bf16 32x32x16", 2);
  run<1, false>("f32 32x32x2", 1); run<2, false>("f32 32x32x2", 1); run<4, false>("f32 32x32x2", 2);
  return 0;
}
