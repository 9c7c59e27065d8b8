// Measurement aid, not on the hot path: what the matrix pipe of THIS box sustains on random operands, measured in the
// same process as the kernels priced against it (bench.py `roofline.sustained_peak`).  The spec figures (2.5 PFLOP/s
// bf16, 157.3 TFLOP/s fp32) assume the boost clock; under a full-chip MFMA load with random mantissas the chip clocks to
// its power budget and a register-only loop -- no LDS, no HBM, no epilogue -- reaches 0.58-0.66 of the bf16 figure on
// this class of box (profiles/r04_mips_bf16_ceiling.txt).  Four independent accumulator chains per wave, two waves per
// SIMD: the issue rate is the pipe's, not a dependency chain's.
#include "common.hpp"

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t probe_hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <bool BF16>
__global__ __launch_bounds__(256) void mfma_probe_kernel(float* __restrict__ sink, int iters) {
  constexpr int CHAINS = 4;
  f32x16 acc[CHAINS];
  const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  // random operands: random sign, random mantissa (7 bits bf16 / 23 bits fp32) AND a random exponent out of 2^-7 .. 2^0 --
  // what normally distributed activations look like to the multipliers (with one fixed exponent the loop draws less
  // power and clocks ~25 % higher than any real kernel can)
  bf16x8 a, b;
  uint16_t* ar = reinterpret_cast<uint16_t*>(&a);
  uint16_t* br = reinterpret_cast<uint16_t*>(&b);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const uint32_t h = probe_hash(tid * 16 + e), g = probe_hash(tid * 16 + 8 + e);
    ar[e] = (uint16_t)((h & 0x807fu) | ((120u + ((h >> 8) & 7u)) << 7));
    br[e] = (uint16_t)((g & 0x807fu) | ((120u + ((g >> 8) & 7u)) << 7));
  }
  const uint32_t ha = probe_hash(tid), hb = probe_hash(tid + 77);
  const float fa = __uint_as_float((ha & 0x807fffffu) | ((120u + ((ha >> 24) & 7u)) << 23));
  const float fb = __uint_as_float((hb & 0x807fffffu) | ((120u + ((hb >> 24) & 7u)) << 23));
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if constexpr (BF16) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
      else acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[c], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[c][e];
  sink[tid] = s;
}

}  // namespace tt

using namespace tt;

extern "C" int64_t tt_mfma_probe_flops(int dtype, int32_t iters) {
  // 512 workgroups (2 per CU = 2 waves per SIMD) x 4 waves x iters x 4 chains x one MFMA
  const double per = dtype == TT_BF16 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
  return (int64_t)(512.0 * 4 * iters * 4 * per);
}

extern "C" int tt_mfma_probe(int dtype, int32_t iters, float* sink, int64_t sink_floats, tt_stream_t stream) {
  if (!sink) return fail_arg("tt_mfma_probe: null pointer");
  if (iters <= 0 || sink_floats < 512 * 256) return fail_arg("tt_mfma_probe: iters > 0, sink of at least 131072 floats");
  if (dtype == TT_BF16) mfma_probe_kernel<true><<<512, 256, 0, S(stream)>>>(sink, iters);
  else if (dtype == TT_F32) mfma_probe_kernel<false><<<512, 256, 0, S(stream)>>>(sink, iters);
  else return fail_arg("tt_mfma_probe: dtype TT_F32 or TT_BF16");
  return check_launch("mfma_probe_kernel");
}
