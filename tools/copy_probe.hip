// Which plain streaming copy reaches this box's HBM rate?  (The calibration kernel of bench.py's roofline.hbm_copy_GBps.)
//   hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o /tmp/copy_probe && /tmp/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float vf4 __attribute__((ext_vector_type(4)));

template <int ITERS, bool NT>
__global__ __launch_bounds__(256) void copy_static(const vf4* __restrict__ src, vf4* __restrict__ dst, long n4) {
  const long n_chunks = (n4 + 256 * ITERS - 1) / (256 * ITERS);
  for (long ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
    const long base = ch * (256 * ITERS) + threadIdx.x;
    vf4 v[ITERS];
#pragma unroll
    for (int k = 0; k < ITERS; ++k) {
      const long i = base + (long)k * 256;
      if (i < n4) v[k] = NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int k = 0; k < ITERS; ++k) {
      const long i = base + (long)k * 256;
      if (i < n4) { if (NT) __builtin_nontemporal_store(v[k], dst + i); else dst[i] = v[k]; }
    }
  }
}

// one-shot grid: one chunk per workgroup
template <int ITERS, bool NT>
__global__ __launch_bounds__(256) void copy_oneshot(const vf4* __restrict__ src, vf4* __restrict__ dst, long n4) {
  const long base = (long)blockIdx.x * (256 * ITERS) + threadIdx.x;
  vf4 v[ITERS];
#pragma unroll
  for (int k = 0; k < ITERS; ++k) {
    const long i = base + (long)k * 256;
    if (i < n4) v[k] = NT ? __builtin_nontemporal_load(src + i) : src[i];
  }
#pragma unroll
  for (int k = 0; k < ITERS; ++k) {
    const long i = base + (long)k * 256;
    if (i < n4) { if (NT) __builtin_nontemporal_store(v[k], dst + i); else dst[i] = v[k]; }
  }
}

template <typename F>
double time_ms(F launch, int reps = 7) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  std::vector<float> ms;
  launch();
  hipDeviceSynchronize();
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float t; hipEventElapsedTime(&t, a, b); ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}

int main() {
  const long bytes = 8448000000L;  // half the P sweep's footprint
  void *src, *dst;
  hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
  hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
  const long n4 = bytes / 16;
  auto report = [&](const char* name, int wgs, double ms) { printf("%-28s wgs %6d: %7.3f ms  %6.0f GB/s\n", name, wgs, ms, 2.0 * bytes / ms / 1e6); };
  for (int wgs : {512, 768, 1024, 1536, 2048, 4096}) {
    report("static ITERS=4 nt", wgs, time_ms([&] { copy_static<4, true><<<wgs, 256>>>((const vf4*)src, (vf4*)dst, n4); }));
    report("static ITERS=8 nt", wgs, time_ms([&] { copy_static<8, true><<<wgs, 256>>>((const vf4*)src, (vf4*)dst, n4); }));
    report("static ITERS=16 nt", wgs, time_ms([&] { copy_static<16, true><<<wgs, 256>>>((const vf4*)src, (vf4*)dst, n4); }));
    report("static ITERS=8 plain", wgs, time_ms([&] { copy_static<8, false><<<wgs, 256>>>((const vf4*)src, (vf4*)dst, n4); }));
  }
  report("oneshot ITERS=4 nt", 0, time_ms([&] { copy_oneshot<4, true><<<(unsigned)((n4 + 1023) / 1024), 256>>>((const vf4*)src, (vf4*)dst, n4); }));
  report("oneshot ITERS=8 nt", 0, time_ms([&] { copy_oneshot<8, true><<<(unsigned)((n4 + 2047) / 2048), 256>>>((const vf4*)src, (vf4*)dst, n4); }));
  report("oneshot ITERS=4 plain", 0, time_ms([&] { copy_oneshot<4, false><<<(unsigned)((n4 + 1023) / 1024), 256>>>((const vf4*)src, (vf4*)dst, n4); }));
  report("hipMemcpyDtoD", 0, time_ms([&] { hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0); }));
  // read-only and write-only rates
  return 0;
}
