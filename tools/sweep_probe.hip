// A/B of the dense-Adam sweep's memory access pattern on one MI355X (all variants in ONE process, interleaved
// rounds, same buffers): what does the streaming ceiling look like on THIS box, and how far is the product
// kernel (adam.hip: SoA p|m|v, persistent workgroups, dynamic 4 x 256-float4 chunks) from it?
//   hipcc --offload-arch=gfx950 -O3 tools/sweep_probe.hip -o /tmp/sweep_probe && /tmp/sweep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

struct C6 { float omb1, b2, eps, nss, bc2; };
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load(const float4* p) {
  const vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void nt_store(const float4& v, float4* p) {
  vf4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<vf4*>(p));
}
__device__ __forceinline__ void upd(float& p, float& m, float& v, const C6& c) {
  m = fmaf(c.omb1, -m, m);
  v = v * c.b2;
  const float denom = sqrtf(v) / c.bc2 + c.eps;
  p = fmaf(c.nss, m / denom, p);
}
__device__ __forceinline__ void upd4(float4& p, float4& m, float4& v, const C6& c) {
  upd(p.x, m.x, v.x, c); upd(p.y, m.y, v.y, c); upd(p.z, m.z, v.z, c); upd(p.w, m.w, v.w, c);
}

// ---- product form: SoA, persistent, dynamic chunks of ITERS x 256 float4; NT: 0 plain, 1 nt loads, 2 nt stores, 3 both
template <int ITERS, int NT>
__global__ __launch_bounds__(256) void soa_persist(float4* __restrict__ W, float4* __restrict__ M, float4* __restrict__ V,
                                                   int64_t n4, C6 c, unsigned* ctr) {
  const unsigned n_chunks = (unsigned)((n4 + 256 * ITERS - 1) / (256 * ITERS));
  __shared__ unsigned s_next[2];
  if (threadIdx.x == 0) s_next[0] = atomicAdd(&ctr[0], 1u);
  __syncthreads();
  unsigned ch = s_next[0];
  int par = 0;
  while (ch < n_chunks) {
    if (threadIdx.x == 0) s_next[par ^ 1] = atomicAdd(&ctr[0], 1u);
    const int64_t base = (int64_t)ch * (256 * ITERS) + threadIdx.x;
    float4 p[ITERS], m[ITERS], v[ITERS];
#pragma unroll
    for (int k = 0; k < ITERS; ++k) {
      const int64_t i = base + (int64_t)k * 256;
      if (i < n4) {
        if (NT & 1) { p[k] = nt_load(&W[i]); m[k] = nt_load(&M[i]); v[k] = nt_load(&V[i]); }
        else { p[k] = W[i]; m[k] = M[i]; v[k] = V[i]; }
      }
    }
#pragma unroll
    for (int k = 0; k < ITERS; ++k) {
      const int64_t i = base + (int64_t)k * 256;
      if (i < n4) {
        upd4(p[k], m[k], v[k], c);
        if (NT & 2) { nt_store(p[k], &W[i]); nt_store(m[k], &M[i]); nt_store(v[k], &V[i]); }
        else { W[i] = p[k]; M[i] = m[k]; V[i] = v[k]; }
      }
    }
    __syncthreads();
    par ^= 1;
    ch = s_next[par];
  }
  if (threadIdx.x == 0 && atomicAdd(&ctr[1], 1u) == gridDim.x - 1) { ctr[0] = 0; ctr[1] = 0; __threadfence(); }
}

// ---- interleaved rows: X = [rows][3][D4] float4 (p | m | v of a row contiguous): one read stream, one write stream
template <int ITERS>
__global__ __launch_bounds__(256) void aos_persist(float4* __restrict__ X, int64_t n4 /* float4 per plane */, int D4, C6 c,
                                                   unsigned* ctr) {
  // a chunk = ITERS x 256 float4 of EACH plane = rows [r0, r0 + ITERS*256/D4); thread handles float4 j of a plane
  const unsigned n_chunks = (unsigned)((n4 + 256 * ITERS - 1) / (256 * ITERS));
  __shared__ unsigned s_next[2];
  if (threadIdx.x == 0) s_next[0] = atomicAdd(&ctr[0], 1u);
  __syncthreads();
  unsigned ch = s_next[0];
  int par = 0;
  while (ch < n_chunks) {
    if (threadIdx.x == 0) s_next[par ^ 1] = atomicAdd(&ctr[0], 1u);
    const int64_t base = (int64_t)ch * (256 * ITERS) + threadIdx.x;
    float4 p[ITERS], m[ITERS], v[ITERS];
    int64_t at[ITERS];
#pragma unroll
    for (int k = 0; k < ITERS; ++k) {
      const int64_t i = base + (int64_t)k * 256;  // plane-linear float4 index
      const int64_t row = i / D4, col = i - row * D4;
      at[k] = row * 3 * D4 + col;
      if (i < n4) { p[k] = X[at[k]]; m[k] = X[at[k] + D4]; v[k] = X[at[k] + 2 * D4]; }
    }
#pragma unroll
    for (int k = 0; k < ITERS; ++k) {
      const int64_t i = base + (int64_t)k * 256;
      if (i < n4) { upd4(p[k], m[k], v[k], c); X[at[k]] = p[k]; X[at[k] + D4] = m[k]; X[at[k] + 2 * D4] = v[k]; }
    }
    __syncthreads();
    par ^= 1;
    ch = s_next[par];
  }
  if (threadIdx.x == 0 && atomicAdd(&ctr[1], 1u) == gridDim.x - 1) { ctr[0] = 0; ctr[1] = 0; __threadfence(); }
}

// ---- references: copy (1 read + 1 write stream), read-only, write-only; grid-stride, 8 float4 in flight
__global__ __launch_bounds__(256) void copy_k(const float4* __restrict__ a, float4* __restrict__ b, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 8;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 * 8 + threadIdx.x; i0 < n4; i0 += stride) {
    float4 t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) if (i0 + k * 256 < n4) t[k] = a[i0 + k * 256];
#pragma unroll
    for (int k = 0; k < 8; ++k) if (i0 + k * 256 < n4) b[i0 + k * 256] = t[k];
  }
}
__global__ __launch_bounds__(256) void read_k(const float4* __restrict__ a, float* out, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 8;
  float s = 0.f;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 * 8 + threadIdx.x; i0 < n4; i0 += stride) {
#pragma unroll
    for (int k = 0; k < 8; ++k) if (i0 + k * 256 < n4) { const float4 t = a[i0 + k * 256]; s += t.x + t.y + t.z + t.w; }
  }
  if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void write_k(float4* __restrict__ b, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 8;
  const float4 z = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int64_t i0 = (int64_t)blockIdx.x * 256 * 8 + threadIdx.x; i0 < n4; i0 += stride) {
#pragma unroll
    for (int k = 0; k < 8; ++k) if (i0 + k * 256 < n4) b[i0 + k * 256] = z;
  }
}

struct Timer {
  hipEvent_t a, b;
  Timer() { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
  template <typename F> float run(F f) {
    (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
  }
};

int main() {
  const int64_t rows = 11000000, D = 128, n = rows * D, n4 = n / 4;
  float *W, *M, *V, *X;
  unsigned* ctr;
  (void)hipMalloc(&W, n * 4); (void)hipMalloc(&M, n * 4); (void)hipMalloc(&V, n * 4); (void)hipMalloc(&X, 3 * n * 4);
  (void)hipMalloc(&ctr, 64); (void)hipMemset(ctr, 0, 64);
  (void)hipMemset(W, 0, n * 4); (void)hipMemset(M, 0, n * 4); (void)hipMemset(V, 0, n * 4); (void)hipMemset(X, 0, 3 * n * 4);
  const C6 c{0.1f, 0.999f, 1e-8f, -1e-3f, 0.5f};
  float4 *w4 = (float4*)W, *m4 = (float4*)M, *v4 = (float4*)V, *x4 = (float4*)X;
  Timer T;
  struct Var { const char* name; double bytes; std::vector<float> ms; };
  std::vector<Var> vars;
  auto add = [&](const char* name, double bytes) { vars.push_back(Var{name, bytes, {}}); return (int)vars.size() - 1; };
  const double B24 = 24.0 * n, B8 = 8.0 * n, B4 = 4.0 * n;
  const int v_p3 = add("soa persist 3/CU 4x256 (product)", B24), v_p2 = add("soa persist 2/CU 4x256", B24),
            v_p4 = add("soa persist 4/CU 4x256", B24), v_p6 = add("soa persist 6/CU 4x256", B24), v_p8 = add("soa persist 8/CU 4x256", B24),
            v_i8 = add("soa persist 3/CU 8x256", B24), v_i2 = add("soa persist 6/CU 2x256", B24),
            v_ntl = add("soa persist 3/CU nt loads", B24), v_nts = add("soa persist 3/CU nt stores", B24), v_ntb = add("soa persist 3/CU nt both", B24),
            v_n2 = add("soa persist 2/CU nt both", B24), v_n4 = add("soa persist 4/CU nt both", B24),
            v_n38 = add("soa persist 3/CU 8x256 nt both", B24), v_n28 = add("soa persist 2/CU 8x256 nt both", B24),
            v_n36 = add("soa persist 3/CU 6x256 nt both", B24), v_n1 = add("soa persist 1/CU 8x256 nt both", B24),
            v_a3 = add("aos persist 3/CU 4x256", B24), v_a6 = add("aos persist 6/CU 4x256", B24),
            v_cp = add("float4 copy (r+w)", B8), v_cp3 = add("float4 copy x3 arrays back-to-back", B24), v_rd = add("read only", B4), v_wr = add("write only", B4);
  for (int rnd = 0; rnd < 5; ++rnd) {
    vars[v_p3].ms.push_back(T.run([&] { soa_persist<4, 0><<<256 * 3, 256>>>(w4, m4, v4, n4, c, ctr); }));
    vars[v_p2].ms.push_back(T.run([&] { soa_persist<4, 0><<<256 * 2, 256>>>(w4, m4, v4, n4, c, ctr); }));
    vars[v_p4].ms.push_back(T.run([&] { soa_persist<4, 0><<<256 * 4, 256>>>(w4, m4, v4, n4, c, ctr); }));
    vars[v_p6].ms.push_back(T.run([&] { soa_persist<4, 0><<<256 * 6, 256>>>(w4, m4, v4, n4, c, ctr); }));
    vars[v_p8].ms.push_back(T.run([&] { soa_persist<4, 0><<<256 * 8, 256>>>(w4, m4, v4, n4, c, ctr); }));
    vars[v_i8].ms.push_back(T.run([&] { soa_persist<8, 0><<<256 * 3, 256>>>(w4, m4, v4, n4, c, ctr); }));
    vars[v_i2].ms.push_back(T.run([&] { soa_persist<2, 0><<<256 * 6, 256>>>(w4, m4, v4, n4, c, ctr); }));
    vars[v_ntl].ms.push_back(T.run([&] { soa_persist<4, 1><<<256 * 3, 256>>>(w4, m4, v4, n4, c, ctr); }));
    vars[v_nts].ms.push_back(T.run([&] { soa_persist<4, 2><<<256 * 3, 256>>>(w4, m4, v4, n4, c, ctr); }));
    vars[v_ntb].ms.push_back(T.run([&] { soa_persist<4, 3><<<256 * 3, 256>>>(w4, m4, v4, n4, c, ctr); }));
    vars[v_n2].ms.push_back(T.run([&] { soa_persist<4, 3><<<256 * 2, 256>>>(w4, m4, v4, n4, c, ctr); }));
    vars[v_n4].ms.push_back(T.run([&] { soa_persist<4, 3><<<256 * 4, 256>>>(w4, m4, v4, n4, c, ctr); }));
    vars[v_n38].ms.push_back(T.run([&] { soa_persist<8, 3><<<256 * 3, 256>>>(w4, m4, v4, n4, c, ctr); }));
    vars[v_n28].ms.push_back(T.run([&] { soa_persist<8, 3><<<256 * 2, 256>>>(w4, m4, v4, n4, c, ctr); }));
    vars[v_n36].ms.push_back(T.run([&] { soa_persist<6, 3><<<256 * 3, 256>>>(w4, m4, v4, n4, c, ctr); }));
    vars[v_n1].ms.push_back(T.run([&] { soa_persist<8, 3><<<256 * 1, 256>>>(w4, m4, v4, n4, c, ctr); }));
    vars[v_a3].ms.push_back(T.run([&] { aos_persist<4><<<256 * 3, 256>>>(x4, n4, (int)(D / 4), c, ctr); }));
    vars[v_a6].ms.push_back(T.run([&] { aos_persist<4><<<256 * 6, 256>>>(x4, n4, (int)(D / 4), c, ctr); }));
    vars[v_cp].ms.push_back(T.run([&] { copy_k<<<256 * 8, 256>>>(w4, m4, n4); }));
    vars[v_cp3].ms.push_back(T.run([&] { copy_k<<<256 * 8, 256>>>(w4, m4, n4); copy_k<<<256 * 8, 256>>>(m4, v4, n4); copy_k<<<256 * 8, 256>>>(v4, w4, n4); }));
    vars[v_rd].ms.push_back(T.run([&] { read_k<<<256 * 8, 256>>>(w4, (float*)ctr + 8, n4); }));
    vars[v_wr].ms.push_back(T.run([&] { write_k<<<256 * 8, 256>>>(v4, n4); }));
  }
  for (auto& v : vars) {
    std::sort(v.ms.begin() + 1, v.ms.end());  // first round = warm-up
    const float med = v.ms[1 + (v.ms.size() - 1) / 2], mn = v.ms[1];
    printf("%-40s median %7.3f ms  min %7.3f ms  %6.0f GB/s (of 8000: %.3f)\n", v.name, med, mn, v.bytes / med / 1e6, v.bytes / med / 1e6 / 8000.0);
  }
  return 0;
}
