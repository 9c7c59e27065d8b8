// Do VALU instructions overlap MFMAs on gfx950?  Register-only loops: per iteration 4 independent bf16
// 32x32x16 MFMAs (128 matrix-pipe cycles per SIMD) plus NV VALU instructions of the kind the MIPS epilogue
// uses (compare / med3 / select chains), either from the SAME wave (intra) or from a second wave on the same
// SIMD that does only VALU (inter: 2 workgroups per CU, even ones MFMA, odd ones VALU).  If the pipes co-execute,
// time stays flat until the VALU cycles exceed the MFMA cycles; if they serialise, it grows linearly from NV = 0.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_coexec.hip -o /tmp/mfma_valu_coexec && /tmp/mfma_valu_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV>
__device__ __forceinline__ void valu_block(float (&m1)[4], float (&m2)[4], int (&arg)[4], const float (&x)[4]) {
  // NV instructions in 4 independent chains of (cmp, med3, select, select); inline assembly so that none is folded away
#pragma unroll
  for (int k = 0; k < NV / 4; ++k) {
    const int c = k & 3;
    asm volatile(
        "v_cmp_gt_f32 vcc, %3, %0\n\t"
        "v_med3_f32 %1, %0, %1, %3\n\t"
        "v_cndmask_b32 %0, %0, %3, vcc\n\t"
        "v_cndmask_b32 %2, %2, 7, vcc"
        : "+v"(m1[c]), "+v"(m2[c]), "+v"(arg[c])
        : "v"(x[c])
        : "vcc");
  }
}

// MODE 0: every wave does MFMA + NV VALU.  MODE 1: even workgroups MFMA only, odd workgroups NV VALU only.
// MODE 2: like 0, but the accumulators live in AGPRs (inline-assembly MFMAs with "a" operands)
template <int NV, int MODE>
__global__ __launch_bounds__(256, 2) void spin(float* out, int iters) {
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c)
    for (int e = 0; e < 16; ++e) acc[c][e] = (float)(threadIdx.x + c);
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 7); b[e] = (__bf16)1.0f; }
  float m1[4], m2[4], x[4];
  int arg[4];
  for (int c = 0; c < 4; ++c) { m1[c] = -1e30f; m2[c] = -1e30f; arg[c] = 0; x[c] = (float)(threadIdx.x * (c + 1)); }
  const bool do_mfma = MODE == 0 || (blockIdx.x & 1) == 0;
  const bool do_valu = MODE != 1 || (blockIdx.x & 1) == 1;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 2) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[c]) : "v"(a), "v"(b));
    } else if (do_mfma) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    if (do_valu) {
      valu_block<NV>(m1, m2, arg, x);
#pragma unroll
      for (int c = 0; c < 4; ++c) x[c] = __int_as_float(__float_as_int(x[c]) ^ (i & 0x7fff));  // keep x changing (1 VALU each)
    }
  }
  float s = 0.f;
  for (int c = 0; c < 4; ++c) {
    for (int e = 0; e < 16; ++e) s += acc[c][e];
    s += m1[c] + m2[c] + (float)arg[c];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int MODE>
void run(int wg_per_cu) {
  float* out;
  const int blocks = 256 * wg_per_cu, iters = 20000;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  spin<NV, MODE><<<blocks, 256>>>(out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  spin<NV, MODE><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // cycles per iteration per SIMD at 2.4 GHz
  printf("mode=%s NV=%3d VALU/iter, %d wg/CU: %7.3f ms = %6.0f cycles/iter (MFMA alone: 128 x waves sharing the SIMD)\n",
         MODE == 0 ? "intra" : MODE == 2 ? "agpr " : "inter", NV + 4, wg_per_cu, ms, ms * 1e-3 * 2.4e9 / iters);
  hipFree(out);
}

int main() {
  run<0, 0>(1); run<16, 0>(1); run<32, 0>(1); run<64, 0>(1); run<128, 0>(1);
  run<0, 0>(2); run<16, 0>(2); run<32, 0>(2); run<64, 0>(2); run<128, 0>(2);
  run<0, 2>(1); run<16, 2>(1); run<32, 2>(1); run<64, 2>(1); run<128, 2>(1);
  run<0, 2>(2); run<32, 2>(2); run<64, 2>(2); run<128, 2>(2);
  return 0;
}
