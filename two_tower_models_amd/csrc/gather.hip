// K1: embedding-row gathers (HBM-bound).  One 512-byte row (dim=128 fp32) is read
// by 32 lanes as float4 -- two rows per wavefront instruction, fully coalesced
// 128-B lines; ids are bounds-checked on the device (zeros + flag, never a fault).
#include "common.hpp"

namespace tt {

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
  using type = float4;
  static __device__ __forceinline__ float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ float4 add(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
  static __device__ __forceinline__ float4 scale(float4 a, float s) {
    return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
  }
};
template <>
struct Vec<1> {
  using type = float;
  static __device__ __forceinline__ float zero() { return 0.f; }
  static __device__ __forceinline__ float add(float a, float b) { return a + b; }
  static __device__ __forceinline__ float scale(float a, float s) { return a * s; }
};

// LPR lanes cooperate on one row; 256/LPR rows per workgroup.
template <int VEC, int LPR>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table,
                                                          int64_t n_rows, int64_t dimv,
                                                          const int64_t* __restrict__ ids,
                                                          int64_t n_ids, float* __restrict__ out,
                                                          int64_t ld_outv, int32_t* oob_flag) {
  using V = typename Vec<VEC>::type;
  const int c = threadIdx.x % LPR;
  const int64_t i = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
  if (i >= n_ids) return;
  const int64_t id = ids[i];
  const bool ok = (id >= 0) && (id < n_rows);
  if (!ok && c == 0 && oob_flag) *oob_flag = 1;
  const V* src = reinterpret_cast<const V*>(table) + (ok ? id : 0) * dimv;
  V* dst = reinterpret_cast<V*>(out) + i * ld_outv;
  for (int64_t k = c; k < dimv; k += LPR) dst[k] = ok ? src[k] : Vec<VEC>::zero();
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

__global__ __launch_bounds__(256) void gather_rows_bf16_kernel(const uint16_t* __restrict__ table,
                                                               int64_t n_rows, int64_t dim,
                                                               const int64_t* __restrict__ ids,
                                                               int64_t n_ids, float* __restrict__ out,
                                                               int64_t ld_out, int32_t* oob_flag) {
  const int c = threadIdx.x & 31;
  const int64_t i = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (i >= n_ids) return;
  const int64_t id = ids[i];
  const bool ok = (id >= 0) && (id < n_rows);
  if (!ok && c == 0 && oob_flag) *oob_flag = 1;
  const uint16_t* src = table + (ok ? id : 0) * dim;
  for (int64_t k = c; k < dim; k += 32) out[i * ld_out + k] = ok ? bf16_to_f32(src[k]) : 0.f;
}

// round-to-nearest-even fp32 -> bf16 (NaN kept quiet)
__global__ void f32_to_bf16_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t u = __float_as_uint(in[i]);
    uint32_t r = ((u & 0x7fffffffu) > 0x7f800000u) ? ((u >> 16) | 0x40u)
                                                    : ((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    out[i] = (uint16_t)r;
  }
}

// History lookup fused with the positional add and the mean pool: every gathered
// row is read from HBM exactly once.  One workgroup per sample; G = 256/LPR row
// groups walk h = g, g+G, ...; group partial sums are combined in fixed order.
template <int VEC, int LPR>
__global__ __launch_bounds__(256) void hist_embed_pool_kernel(
    const float* __restrict__ table, int64_t n_rows, int64_t dimv, const int64_t* __restrict__ ids,
    int64_t H, const float* __restrict__ pe, float* __restrict__ x, float* __restrict__ pooled,
    int64_t ld_pooledv, int32_t* oob_flag) {
  using V = typename Vec<VEC>::type;
  constexpr int G = 256 / LPR;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  V* part = reinterpret_cast<V*>(smem_raw);  // [G][dimv]
  const int c = threadIdx.x % LPR, g = threadIdx.x / LPR;
  const int64_t b = blockIdx.x;
  const V* pev = reinterpret_cast<const V*>(pe);
  V* xb = reinterpret_cast<V*>(x) + b * H * dimv;
  // U history positions per round: first all ids, then all rows (unconditional loads from clamped
  // indices, masked afterwards) -- two memory round trips per round instead of two per position
  constexpr int U = 8;
  for (int64_t k = c; k < dimv; k += LPR) {
    V acc = Vec<VEC>::zero();
    for (int64_t h0 = g; h0 < H; h0 += (int64_t)G * U) {
      int64_t id[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t h = h0 + (int64_t)u * G;
        const int64_t hc = h < H ? h : H - 1;
        id[u] = ids ? ids[b * H + hc] : b * H + hc;
      }
      V v[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ok[u] = !ids || ((id[u] >= 0) && (id[u] < n_rows));
        v[u] = reinterpret_cast<const V*>(table)[(ok[u] ? id[u] : 0) * dimv + k];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t h = h0 + (int64_t)u * G;
        if (h < H) {
          if (!ok[u]) *oob_flag = 1;
          const V x_in = ok[u] ? v[u] : Vec<VEC>::zero();
          acc = Vec<VEC>::add(acc, x_in);
          xb[h * dimv + k] = pe ? Vec<VEC>::add(x_in, pev[h * dimv + k]) : x_in;
        }
      }
    }
    part[g * dimv + k] = acc;
  }
  __syncthreads();
  if (g == 0) {
    const float inv = 1.0f / (float)H;
    for (int64_t k = c; k < dimv; k += LPR) {
      V acc = part[k];
      for (int gg = 1; gg < G; ++gg) acc = Vec<VEC>::add(acc, part[gg * dimv + k]);
      reinterpret_cast<V*>(pooled)[b * ld_pooledv + k] = Vec<VEC>::scale(acc, inv);
    }
  }
}

// backward of the mean pool: dx[b,h,:] += d_pooled[b,:] / H
__global__ void hist_pool_bwd_kernel(float* __restrict__ dx, int64_t B, int64_t H, int64_t dim,
                                     const float* __restrict__ d_pooled, int64_t ld_pooled) {
  const int64_t total = B * H * dim;
  const float inv = 1.0f / (float)H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / (H * dim), d = i % dim;
    dx[i] += d_pooled[b * ld_pooled + d] * inv;
  }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace tt

using namespace tt;

extern "C" int tt_gather_rows(const float* table, int64_t n_rows, int64_t dim, const int64_t* ids,
                              int64_t n_ids, float* out, int64_t ld_out, int32_t* oob_flag,
                              tt_stream_t stream) {
  if (!table || !ids || !out) return fail_arg("tt_gather_rows: null pointer");
  if (n_rows <= 0 || dim <= 0 || n_ids < 0 || ld_out < dim) return fail_arg("tt_gather_rows: sizes");
  if (n_ids == 0) return 0;
  const bool vec = (dim % 4 == 0) && (ld_out % 4 == 0) && aligned16(table) && aligned16(out);
  if (vec) {
    const int64_t dv = dim / 4;
    if (dv <= 8)
      gather_rows_kernel<4, 8><<<ceil_div(n_ids, 32), 256, 0, S(stream)>>>(table, n_rows, dv, ids, n_ids, out, ld_out / 4, oob_flag);
    else if (dv <= 16)
      gather_rows_kernel<4, 16><<<ceil_div(n_ids, 16), 256, 0, S(stream)>>>(table, n_rows, dv, ids, n_ids, out, ld_out / 4, oob_flag);
    else
      gather_rows_kernel<4, 32><<<ceil_div(n_ids, 8), 256, 0, S(stream)>>>(table, n_rows, dv, ids, n_ids, out, ld_out / 4, oob_flag);
  } else {
    gather_rows_kernel<1, 64><<<ceil_div(n_ids, 4), 256, 0, S(stream)>>>(table, n_rows, dim, ids, n_ids, out, ld_out, oob_flag);
  }
  return check_launch("gather_rows_kernel");
}

extern "C" int tt_gather_rows_bf16(const uint16_t* table, int64_t n_rows, int64_t dim,
                                   const int64_t* ids, int64_t n_ids, float* out, int64_t ld_out,
                                   int32_t* oob_flag, tt_stream_t stream) {
  if (!table || !ids || !out) return fail_arg("tt_gather_rows_bf16: null pointer");
  if (n_rows <= 0 || dim <= 0 || n_ids < 0 || ld_out < dim) return fail_arg("tt_gather_rows_bf16: sizes");
  if (n_ids == 0) return 0;
  gather_rows_bf16_kernel<<<ceil_div(n_ids, 8), 256, 0, S(stream)>>>(table, n_rows, dim, ids, n_ids, out, ld_out, oob_flag);
  return check_launch("gather_rows_bf16_kernel");
}

extern "C" int tt_f32_to_bf16(const float* in, uint16_t* out, int64_t n, tt_stream_t stream) {
  if (!in || !out || n < 0) return fail_arg("tt_f32_to_bf16");
  if (n == 0) return 0;
  const int64_t blocks = ceil_div(n, 256) < 4096 ? ceil_div(n, 256) : 4096;
  f32_to_bf16_kernel<<<blocks, 256, 0, S(stream)>>>(in, out, n);
  return check_launch("f32_to_bf16_kernel");
}

extern "C" int tt_hist_embed_pool(const float* table, int64_t n_rows, int64_t dim, const int64_t* ids,
                                  int64_t B, int64_t H, const float* pe, float* x, float* pooled,
                                  int64_t ld_pooled, int32_t* oob_flag, tt_stream_t stream) {
  if (!table || !x || !pooled || !oob_flag) return fail_arg("tt_hist_embed_pool: null pointer");
  if (dim <= 0 || B < 0 || H <= 0 || ld_pooled < dim || (ids && n_rows <= 0))
    return fail_arg("tt_hist_embed_pool: sizes");
  if (B == 0) return 0;
  const bool vec = (dim % 4 == 0) && (ld_pooled % 4 == 0) && aligned16(table) && aligned16(x) &&
                   aligned16(pooled) && (!pe || aligned16(pe));
  if (vec) {
    const int64_t dv = dim / 4;
    const size_t lds = (size_t)(256 / 32) * dv * sizeof(float4);
    if (lds > 64 * 1024) { set_error("tt_hist_embed_pool: dim too large"); return TT_E_UNSUPPORTED; }
    hist_embed_pool_kernel<4, 32><<<B, 256, lds, S(stream)>>>(table, n_rows, dv, ids, H, pe, x, pooled, ld_pooled / 4, oob_flag);
  } else {
    const size_t lds = (size_t)(256 / 64) * dim * sizeof(float);
    if (lds > 64 * 1024) { set_error("tt_hist_embed_pool: dim too large"); return TT_E_UNSUPPORTED; }
    hist_embed_pool_kernel<1, 64><<<B, 256, lds, S(stream)>>>(table, n_rows, dim, ids, H, pe, x, pooled, ld_pooled, oob_flag);
  }
  return check_launch("hist_embed_pool_kernel");
}

namespace tt {
int gemm_ws16_pool_try(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc,
                       const float* pool, int64_t ld_pool, int64_t group, float scale, hipStream_t st);
}

extern "C" int tt_hist_dx_pool_bwd(const float* dqkv, const float* w_in, int64_t B, int64_t H, int64_t D, const float* d_pooled,
                                   int64_t ld_pooled, float* dx, tt_stream_t stream) {
  if (!dqkv || !w_in || !d_pooled || !dx) return fail_arg("tt_hist_dx_pool_bwd: null pointer");
  if (B <= 0 || H <= 0 || D <= 0 || ld_pooled < D) return fail_arg("tt_hist_dx_pool_bwd: sizes");
  const int rc = gemm_ws16_pool_try(B * H, D, 3 * D, dqkv, 3 * D, w_in, D, dx, D, d_pooled, ld_pooled, H, 1.0f / (float)H, S(stream));
  if (rc == -100) {
    set_error("tt_hist_dx_pool_bwd: takes D = 128, B * H >= 16384, 16-byte aligned operands");
    return TT_E_UNSUPPORTED;
  }
  return rc;
}

extern "C" int tt_hist_pool_bwd(float* dx, int64_t B, int64_t H, int64_t dim, const float* d_pooled,
                                int64_t ld_pooled, tt_stream_t stream) {
  if (!dx || !d_pooled) return fail_arg("tt_hist_pool_bwd: null pointer");
  if (B < 0 || H <= 0 || dim <= 0 || ld_pooled < dim) return fail_arg("tt_hist_pool_bwd: sizes");
  if (B == 0) return 0;
  const int64_t total = B * H * dim;
  const int64_t blocks = ceil_div(total, 256) < 4096 ? ceil_div(total, 256) : 4096;
  hist_pool_bwd_kernel<<<(unsigned)blocks, 256, 0, S(stream)>>>(dx, B, H, dim, d_pooled, ld_pooled);
  return check_launch("hist_pool_bwd_kernel");
}
