// What bounds the bf16 MIPS pass-1 kernel?  A stand-alone loop with the kernel's shape -- per 64-row tile and wave
// 2 sub-tiles x 8 k-groups x NQ = 4 query fragments of v_mfma_f32_32x32x16_bf16, then the group epilogue over the
// 4 x 16 scores of each sub-tile -- whose memory side is added piece by piece (template flags):
//   L  the A operand comes from LDS (one ds_read_b128 per k-group, issued one k-group ahead, explicit lgkmcnt waits)
//   B  one workgroup barrier per tile
//   D  the tile ring is filled by LDS-DMA from a global buffer (4 buffer_load ... lds per wave and tile, 3 tiles ahead)
//   G  (instead of L / B / D) no LDS at all: every wave loads its A operand straight from global memory into registers,
//      8 x global_load_dwordx4 per 32-row sub-tile, one sub-tile ahead (the four waves of a workgroup read the same rows)
// Epilogue variants: 0 = none, 1 = max only (v_max3, 0.5 / score), 2 = best + runner-up + best QUAD (1.75 / score,
// the product kernel), 3 = best + runner-up + best ROW (4 / score, the round-1/2 kernel).
// Random bf16 operands (the clock the chip sustains depends on the data: MI355X_MICROARCH.md, DVFS).
//   hipcc --offload-arch=gfx950 -O3 tools/mips_epilogue_probe.hip -o /tmp/mips_epilogue_probe && /tmp/mips_epilogue_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int NQ = 4, DPX = 8, STAGES = 4, TILE_BYTES = 64 * 256;

template <int V, bool L, bool B, bool D, bool G = false>
__global__ __launch_bounds__(256, 2) void spin(float* out, const uint4* __restrict__ qsrc, const char* __restrict__ corpus,
                                               int64_t corpus_tiles, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  u32x4 q[NQ][DPX];
  for (int n = 0; n < NQ; ++n)
    for (int g = 0; g < DPX; ++g) {
      const uint4 v = qsrc[((blockIdx.x * 4 + wave) * NQ + n) % 64 * 512 + g * 64 + lane];
      q[n][g] = u32x4{v.x, v.y, v.z, v.w};
    }
  // LDS ring: filled once from global memory (then re-filled by DMA if D)
  for (int i = threadIdx.x; i < STAGES * TILE_BYTES / 16; i += 256)
    reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(corpus)[(blockIdx.x * 977 + i) % (corpus_tiles * 1024)];
  __syncthreads();
  u32x4 areg[2];
  {
    const uint4 v0 = reinterpret_cast<const uint4*>(corpus)[lane], v1 = reinterpret_cast<const uint4*>(corpus)[64 + lane];
    areg[0] = u32x4{v0.x, v0.y, v0.z, v0.w}; areg[1] = u32x4{v1.x, v1.y, v1.z, v1.w};
  }
  float m1[NQ], m2[NQ];
  int arg[NQ];
  for (int n = 0; n < NQ; ++n) { m1[n] = -1e30f; m2[n] = -1e30f; arg[n] = 0; }
  int64_t tile = (int64_t)blockIdx.x * 131 % corpus_tiles;
  auto dma = [&](int stage) {
    if (!D) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(corpus + tile * TILE_BYTES), 0, TILE_BYTES, 0x00020000);
    const int w = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + stage * TILE_BYTES + (w * 4 + i) * 1024), 16,
                                               lane * 16, (w * 4 + i) * 1024, 0, 0);
    tile = tile + 1 < corpus_tiles ? tile + 1 : 0;
  };
  if (D) {
    for (int s = 0; s < STAGES - 1; ++s) dma(s);
    wait_vmcnt<(STAGES - 2) * 4>();
    __syncthreads();
  }
  int cur = 0;
  u32x4 ga[2][DPX];
  auto gload = [&](u32x4 (&dst)[DPX], int64_t tl, int jt) {
    const char* base = corpus + tl * TILE_BYTES + (jt * 32 + r) * 256 + h * 16;
#pragma unroll
    for (int g = 0; g < DPX; ++g) dst[g] = *reinterpret_cast<const u32x4*>(base + g * 32);
  };
  if (G) gload(ga[0], tile, 0);
  for (int i = 0; i < iters; ++i) {
    dma((cur + STAGES - 1) % STAGES);
    const char* ys = smem + cur * TILE_BYTES;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      f32x16 acc[NQ];
#pragma unroll
      for (int n = 0; n < NQ; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
      if (G) {
        if (jt == 0) {
          gload(ga[1], tile, 1);
        } else {
          tile = tile + 1 < corpus_tiles ? tile + 1 : 0;
          gload(ga[0], tile, 0);
        }
#pragma unroll
        for (int g = 0; g < DPX; ++g)
#pragma unroll
          for (int n = 0; n < NQ; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ga[jt][g]), __builtin_bit_cast(bf16x8, q[n][g]), acc[n], 0, 0, 0);
      } else if (L) {
        const int row = jt * 32 + r;
        const uint32_t lrow = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)ys + (uint32_t)row * 256;
        const int sw = row & 15;
        u32x4 yy[2];
        auto rd = [&](u32x4& dst, int g) {
          const uint32_t a = lrow + 16u * (uint32_t)((2 * g + h) ^ sw);
          asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(a) : "memory");
        };
        rd(yy[0], 0);
#pragma unroll
        for (int g = 0; g < DPX; ++g) {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yy[g & 1]) : : "memory");
          if (g + 1 < DPX) rd(yy[(g + 1) & 1], g + 1);
#pragma unroll
          for (int n = 0; n < NQ; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, yy[g & 1]), __builtin_bit_cast(bf16x8, q[n][g]), acc[n], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int g = 0; g < DPX; ++g) {
          asm volatile("" : "+v"(areg[g & 1]));  // the A operand "changes" every k-group
#pragma unroll
          for (int n = 0; n < NQ; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, areg[g & 1]), __builtin_bit_cast(bf16x8, q[n][g]), acc[n], 0, 0, 0);
        }
      }
      const int off0 = 16 * jt;
      if (V == 0) {
#pragma unroll
        for (int n = 0; n < NQ; ++n) m1[n] += acc[n][0];
      } else if (V == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int n = 0; n < NQ; ++n) m1[n] = max3f(m1[n], acc[n][2 * j], acc[n][2 * j + 1]);
      } else if (V == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < NQ; ++n) {
            const float x0 = acc[n][4 * j], x1 = acc[n][4 * j + 1], x2 = acc[n][4 * j + 2], x3 = acc[n][4 * j + 3];
            const float b1 = __builtin_amdgcn_fmed3f(m1[n], x0, x1), a1 = max3f(m1[n], x0, x1);
            const float b2 = __builtin_amdgcn_fmed3f(a1, x2, x3), a2 = max3f(a1, x2, x3);
            m2[n] = max3f(m2[n], b1, b2);
            const bool gt = a2 > m1[n];
            arg[n] = gt ? off0 / 4 + j : arg[n];
            m1[n] = a2;
          }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e)
#pragma unroll
          for (int n = 0; n < NQ; ++n) {
            const float x = acc[n][e];
            const bool gt = x > m1[n];
            m2[n] = __builtin_amdgcn_fmed3f(m1[n], m2[n], x);
            m1[n] = gt ? x : m1[n];
            arg[n] = gt ? off0 + e : arg[n];
          }
      }
    }
    if (D) wait_vmcnt<(STAGES - 2) * 4>();
    if (B) __syncthreads();
    cur = (cur + 1) % STAGES;
  }
  float s = 0.f;
  for (int n = 0; n < NQ; ++n) s += m1[n] + m2[n] + (float)arg[n];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float* g_out;
static uint4* g_q;
static char* g_corpus;
static int64_t g_tiles;

template <int V, bool L, bool B, bool D, bool G = false>
static void run(const char* what) {
  const int blocks = 256 * 2 * 4, iters = 600;
  auto k = spin<V, L, B, D, G>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, STAGES * TILE_BYTES);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<<<blocks, 256, STAGES * TILE_BYTES>>>(g_out, g_q, g_corpus, g_tiles, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<<<blocks, 256, STAGES * TILE_BYTES>>>(g_out, g_q, g_corpus, g_tiles, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 2 * 8 * 4 * (2.0 * 32 * 32 * 16);
  printf("%-26s LDS=%d barrier=%d DMA=%d global->VGPR=%d  %7.3f ms  %7.1f TFLOP/s\n", what, (int)L, (int)B, (int)D, (int)G, ms, flops / ms / 1e9);
}

template <int V>
static void run_all(const char* what) {
  run<V, false, false, false>(what);
  run<V, true, false, false>(what);
  run<V, true, true, false>(what);
  run<V, true, true, true>(what);
  run<V, false, false, false, true>(what);
}

int main() {
  g_tiles = 16384;  // 256 MiB of "corpus"
  const size_t cbytes = (size_t)g_tiles * TILE_BYTES, qbytes = 64 * 512 * 16;
  (void)hipMalloc(&g_out, 2048 * 256 * 4);
  (void)hipMalloc(&g_q, qbytes);
  (void)hipMalloc(&g_corpus, cbytes);
  // random bf16: random 16-bit patterns with a sane exponent
  uint16_t* hc = (uint16_t*)malloc(cbytes);
  srand(1);
  for (size_t i = 0; i < cbytes / 2; ++i) hc[i] = (uint16_t)(((rand() & 1) << 15) | ((120 + (rand() & 7)) << 7) | (rand() & 127));
  (void)hipMemcpy(g_corpus, hc, cbytes, hipMemcpyHostToDevice);
  (void)hipMemcpy(g_q, hc + 12345, qbytes, hipMemcpyHostToDevice);
  run_all<0>("no epilogue");
  run_all<1>("max only (0.5/score)");
  run_all<2>("top-2 + quad (1.75/score)");
  run_all<3>("top-2 + row (4/score)");
  return 0;
}
