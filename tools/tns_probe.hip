// Probe: how many fp32 MFMAs per global load does a "global -> registers -> MFMA" stream need on gfx950?
// Every variant runs the same number of v_mfma_f32_32x32x2_f32 per wave; operands come from a small
// L1/L2-resident buffer (the memory side is out of the picture) through different load mixes.
//   hipcc --offload-arch=gfx950 -O3 tools/tns_probe.hip -o /tmp/tns_probe && /tmp/tns_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// AW, BW: floats per lane per k-step from A / B (AW x BW tiles per wave).  SRC: 0 = none (registers only),
// 1 = global loads, 2 = LDS reads.  WPS = waves per SIMD (threads = 256 * WPS).
template <int AW, int BW, int U, int WPS, int SRC>
__global__ __launch_bounds__(256 * WPS, WPS) void probe(const float* __restrict__ A, const float* __restrict__ B, float* out,
                                                        int lda, int ldb, int chunks) {
  __shared__ float lds[SRC == 2 ? 2 * U * 2 * 640 : 1];
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const unsigned va = h * lda + AW * r, vb = h * ldb + BW * r;
  if (SRC == 2) {
    for (int i = threadIdx.x; i < 2 * U * 2 * 640; i += blockDim.x) lds[i] = A[i % 4096];
    __syncthreads();
  }
  f32x16 acc[AW][BW];
  for (int i = 0; i < AW; ++i) for (int j = 0; j < BW; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float a0[U][AW], b0[U][BW], a1[U][AW], b1[U][BW];
  auto ld = [&](float (&a)[AW], float (&b)[BW], const float* pa, const float* pb, int s) {
    if constexpr (SRC == 2) {
      const float* la = lds + s * 1280 + h * 640 + AW * r; const float* lb = la + 512 / 4 * 0 + 128 * (AW == 4 ? 1 : 1);
      if constexpr (AW == 4) { f32x4 v = *(const f32x4*)la; a[0]=v.x;a[1]=v.y;a[2]=v.z;a[3]=v.w; }
      else if constexpr (AW == 2) { f32x2 v = *(const f32x2*)la; a[0]=v.x;a[1]=v.y; } else a[0] = *la;
      const float* lbb = lds + s * 1280 + h * 640 + 512 + BW * r; (void)lb;
      if constexpr (BW == 4) { f32x4 v = *(const f32x4*)lbb; b[0]=v.x;b[1]=v.y;b[2]=v.z;b[3]=v.w; }
      else if constexpr (BW == 2) { f32x2 v = *(const f32x2*)lbb; b[0]=v.x;b[1]=v.y; } else b[0] = *lbb;
    } else {
      if constexpr (AW == 4) { f32x4 v = *(const f32x4*)pa; a[0]=v.x;a[1]=v.y;a[2]=v.z;a[3]=v.w; }
      else if constexpr (AW == 2) { f32x2 v = *(const f32x2*)pa; a[0]=v.x;a[1]=v.y; } else a[0] = *pa;
      if constexpr (BW == 4) { f32x4 v = *(const f32x4*)pb; b[0]=v.x;b[1]=v.y;b[2]=v.z;b[3]=v.w; }
      else if constexpr (BW == 2) { f32x2 v = *(const f32x2*)pb; b[0]=v.x;b[1]=v.y; } else b[0] = *pb;
    }
  };
  auto mm = [&](const float (&a)[AW], const float (&b)[BW]) {
#pragma unroll
    for (int i = 0; i < AW; ++i)
#pragma unroll
      for (int j = 0; j < BW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
  };
  const float* ca = A; const float* cb = B;
#pragma unroll
  for (int s = 0; s < U; ++s) ld(a0[s], b0[s], ca + (unsigned)(2 * s * lda) + va, cb + (unsigned)(2 * s * ldb) + vb, s);
  for (int c = 0; c < chunks; c += 2) {
#pragma unroll
    for (int s = 0; s < U; ++s) {
      if (SRC) ld(a1[s], b1[s], ca + (unsigned)(2 * s * lda) + va, cb + (unsigned)(2 * s * ldb) + vb, s + U);
      mm(a0[s], b0[s]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < U; ++s) {
      if (SRC) ld(a0[s], b0[s], ca + (unsigned)(2 * s * lda) + va, cb + (unsigned)(2 * s * ldb) + vb, s);
      mm(SRC ? a1[s] : a0[s], SRC ? b1[s] : b0[s]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float t = 0.f;
  for (int i = 0; i < AW; ++i) for (int j = 0; j < BW; ++j) for (int e = 0; e < 16; ++e) t += acc[i][j][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int AW, int BW, int U, int WPS, int SRC>
static void run(const char* name, const float* A, const float* B, float* out) {
  // equal MFMA count per SIMD: WPS waves x chunks x U x AW*BW
  const int mf_per_simd = 4800 * 4;  // 19 200 MFMAs per SIMD
  int chunks = mf_per_simd / (WPS * U * AW * BW);
  chunks &= ~1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) probe<AW, BW, U, WPS, SRC><<<256, 256 * WPS>>>(A, B, out, 512, 128, chunks);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) probe<AW, BW, U, WPS, SRC><<<256, 256 * WPS>>>(A, B, out, 512, 128, chunks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double mf = (double)chunks * U * AW * BW * WPS * 4 * 256;  // MFMAs on the chip
  printf("%-44s %7.1f us  %6.1f TFLOP/s  (%d MFMAs per wave, %.2f loads per MFMA)\n", name, ms * 1e3, mf * 4096 / ms / 1e9,
         chunks * U * AW * BW, SRC ? 2.0 / (AW * BW) : 0.0);
}

int main() {
  float *A, *B, *out;
  hipMalloc(&A, 1 << 20); hipMalloc(&B, 1 << 20); hipMalloc(&out, 1 << 22);
  std::vector<float> h(1 << 18, 0.001f);
  hipMemcpy(A, h.data(), 1 << 20, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), 1 << 20, hipMemcpyHostToDevice);
  run<4, 1, 6, 3, 0>("registers only, 4 tiles, 3 waves/SIMD", A, B, out);
  run<4, 1, 6, 3, 1>("global x4 + x1 per 4 MFMAs, 3 waves/SIMD", A, B, out);
  run<4, 1, 6, 3, 2>("LDS b128 + b32 per 4 MFMAs, 3 waves/SIMD", A, B, out);
  run<2, 2, 6, 3, 1>("global x2 + x2 per 4 MFMAs, 3 waves/SIMD", A, B, out);
  run<4, 2, 8, 2, 0>("registers only, 8 tiles, 2 waves/SIMD", A, B, out);
  run<4, 2, 8, 2, 1>("global x4 + x2 per 8 MFMAs, 2 waves/SIMD", A, B, out);
  run<4, 2, 8, 2, 2>("LDS b128 + b64 per 8 MFMAs, 2 waves/SIMD", A, B, out);
  run<4, 4, 6, 1, 0>("registers only, 16 tiles, 1 wave/SIMD", A, B, out);
  run<4, 4, 6, 1, 1>("global x4 + x4 per 16 MFMAs, 1 wave/SIMD", A, B, out);
  run<4, 4, 6, 1, 2>("LDS b128 + b128 per 16 MFMAs, 1 wave/SIMD", A, B, out);
  run<2, 1, 8, 3, 1>("global x2 + x1 per 2 MFMAs, 3 waves/SIMD", A, B, out);
  run<1, 1, 8, 3, 1>("global x1 + x1 per 1 MFMA, 3 waves/SIMD", A, B, out);
  return 0;
}
