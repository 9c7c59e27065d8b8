// What costs the second-product loop of the in-batch CE kernels its last 20 % of MFMA rate?
// The loop of ce_bwd_kept_kernel rebuilt from nothing, one ingredient at a time (2 workgroups per CU):
//   bit 0: B operands read from LDS (swizzled tile, one ds_read per MFMA, one step ahead)
//   bit 1: the gradient VALU block (16 x exp2 / compare / multiply) in front of each sub-tile
//   bit 2: per-row statistics read from LDS for that block
//   bit 3: workgroup barrier every two sub-tiles
//   bit 4: the 32 logits of the next tile loaded from global memory every tile (so the gradient
//          block is NOT loop-invariant, as it partly is without this bit)
//   bit 5: LDS-DMA of the next 32 KiB tile into a second named buffer, vmcnt(0) before the barrier
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_loop_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LD = 128, BJ = 64;

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float* out, const float* zin, int tiles) {
  __shared__ __attribute__((aligned(16))) float ys[BJ * LD + 2 * BJ];
  __shared__ __attribute__((aligned(16))) float other[(MODE & 32) ? BJ * LD : 4];
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < BJ * LD + 2 * BJ; i += 256) ys[i] = (float)((i * 37 + blockIdx.x) & 255) * (1.0f / 256.0f);
  __syncthreads();
  f32x16 dacc[4];
  for (int d = 0; d < 4; ++d)
    for (int e = 0; e < 16; ++e) dacc[d][e] = 0.f;
  int ybase[4];
  for (int q = 0; q < 4; ++q) ybase[q] = 4 * h * LD + 4 * ((((r >> 2) ^ (4 * h)) & 15) ^ q) + (r & 3);
  float zc[32];
  for (int i = 0; i < 32; ++i) zc[i] = zin[(threadIdx.x * 32 + i) & 4095];
  float gt[16];
  for (int e = 0; e < 16; ++e) gt[e] = zc[e];
  float zn[32];
  for (int i = 0; i < 32; ++i) zn[i] = zc[i];
  const int wave = threadIdx.x >> 6;
  for (int t = 0; t < tiles; ++t) {
    if (MODE & 16) {
#pragma unroll
      for (int i = 0; i < 32; ++i) zc[i] = zn[i];
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(zin), 0, 1 << 20, 0x00020000);
#pragma unroll
      for (int i = 0; i < 32; ++i)
        zn[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (threadIdx.x & 31) * 4, ((i * 97 + t * 13 + blockIdx.x) & 1023) * 128, 0));
    }
    if (MODE & 32) {
      const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(zin), 0, 1 << 20, 0x00020000);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (__attribute__((address_space(3))) void*)(other + (wave * 8 + i) * 256), 16,
                                                 lane * 16 + ((t * 8 + i) & 63) * 1024, 0, 0, 0);
    }
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      if (MODE & 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float lv[4] = {1.f, 2.f, 3.f, 4.f}, cv[4] = {0.5f, 0.25f, 0.125f, 1.f};
          if (MODE & 4) {
            const float4 l4 = *reinterpret_cast<const float4*>(ys + BJ * LD + jt * 32 + 4 * h + 8 * q);
            const float4 c4 = *reinterpret_cast<const float4*>(ys + BJ * LD + BJ + jt * 32 + 4 * h + 8 * q);
            lv[0] = l4.x; lv[1] = l4.y; lv[2] = l4.z; lv[3] = l4.w;
            cv[0] = c4.x; cv[1] = c4.y; cv[2] = c4.z; cv[3] = c4.w;
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int e = 4 * q + c;
            const float pr = __builtin_amdgcn_exp2f(zc[jt * 16 + e] - lv[c]);
            gt[e] = cv[c] * (pr - ((e + t == 1000000) ? 1.f : 0.f));
          }
        }
      }
      float yv[2][4];
      auto yread = [&](int e, float (&dst)[4]) {
        const int E = (e & 3) + 8 * (e >> 2), hi = (e >> 2) & 1;
#pragma unroll
        for (int d = 0; d < 4; ++d) dst[d] = (MODE & 1) ? ys[(jt * 32 + E) * LD + 32 * (d ^ hi) + ybase[e & 3]] : 1.0f + d;
      };
      yread(0, yv[0]);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (e + 1 < 16) yread(e + 1, yv[(e + 1) & 1]);
#pragma unroll
        for (int d = 0; d < 4; ++d) dacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(gt[e], yv[e & 1][d], dacc[d], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE & 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE & 8) __syncthreads();
  }
  float s = 0.f;
  for (int d = 0; d < 4; ++d)
    for (int e = 0; e < 16; ++e) s += dacc[d][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const float* zin, float* out) {
  const int blocks = 512, tiles = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<blocks, 256>>>(out, zin, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE><<<blocks, 256>>>(out, zin, tiles);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 4 * tiles * 128 * (2.0 * 32 * 32 * 2);
  printf("mode %2d (%s%s%s%s%s%s): %7.2f ms  %.1f TFLOP/s\n", MODE, (MODE & 1) ? "lds-B " : "", (MODE & 2) ? "valu " : "",
         (MODE & 4) ? "lds-stats " : "", (MODE & 8) ? "barrier " : "", (MODE & 16) ? "logit-loads " : "",
         (MODE & 32) ? "lds-dma" : "", ms, flop / ms / 1e9);
}

int main() {
  float *zin, *out;
  hipMalloc(&zin, 1 << 20);
  hipMalloc(&out, 512 * 256 * 4);
  hipMemset(zin, 0, 1 << 20);
  run<0>(zin, out); run<1>(zin, out); run<2>(zin, out); run<3>(zin, out); run<7>(zin, out); run<9>(zin, out); run<15>(zin, out);
  run<16 + 7>(zin, out); run<16 + 15>(zin, out); run<32 + 15>(zin, out); run<32 + 16 + 15>(zin, out);
  return 0;
}
