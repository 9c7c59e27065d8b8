// Exploratory: in-batch logits S = U I^T at fp32-grade accuracy on the 16-bit matrix pipes.
//
// Each fp32 operand is cut into TWO fp16 terms  x * s = h + l  (s = a power of two that brings the matrix' largest
// magnitude to (2^14, 2^15], the top of fp16's range; h = fp16(x s), l = fp16(x s - h): 11 + 11 significant bits for
// every element within 2^-17 of the largest, residual <= 2^-23 |x s|; smaller elements lose bits of l to fp16's
// denormal spacing, an ABSOLUTE error of 2^-39 of the largest), and the product runs as THREE
// v_mfma_f32_32x32x16_f16 into one fp32 accumulator (exact products, fp32 accumulate):
//     acc += ih uh + ih ul + il uh            S = acc / (s_u s_i)
// the dropped il ul term is <= 2^-24 |u||i| per element.  Three 8-pass MFMAs per 16 k instead of eight 16-pass fp32
// MFMAs: 96 vs 512 matrix-pipe cycles.  (The verdict proposed three bf16 terms and six products for 2^-22; fp16's 11-bit
// significand reaches the same with two terms and three products, at the price of a range argument -- the power-of-two
// scale -- that bf16 does not need.)
//
// What this program measures, on the sharded trainer's W = 8 shape (M = 8192 users, N = 65536 items, D = 128):
//   1. element-wise error of the split product against float64, next to a plain fp32 dot product and the bf16 x3 / x6 forms
//      (host emulation of the MFMA arithmetic: exact products, fp32 accumulation);
//   2. a device kernel for the row-wise log-sum-exp of S (the forward logits kernel without its dU half): users
//      stationary in registers as B fragments (64 per wave, both terms), item tiles by LDS-DMA into a swizzled
//      two-stage ring, online log-sum-exp per lane; time per launch and the error of lse against float64.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/f16x2_logits_probe.hip -o /tmp/f16x2 && /tmp/f16x2
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);       \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int D = 128;
constexpr int TILE = 64;                  // item rows per ring stage
constexpr int ROW_B = D * 2;              // bytes per fp16 row
constexpr int PART_B = TILE * ROW_B;      // one term of one stage: 16 KiB
constexpr int STAGE_B = 2 * PART_B;       // h | l
constexpr int USERS_PER_WG = 256;         // 4 waves x 64

__device__ __forceinline__ int brow(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// one stage = TILE rows of both terms; every wave instruction lands 1 KiB = 4 rows; the lane fetches the 16-B chunk
// that belongs at ITS linear LDS position after the XOR swizzle (chunk' = chunk ^ (row & 15))
__device__ __forceinline__ void stage_dma(const _Float16* __restrict__ Ih, const _Float16* __restrict__ Il, int64_t row0,
                                          int64_t n_rows, char* stage, int wave, int lane) {
  const int64_t left = n_rows - row0;
  const int rows_here = left < TILE ? (int)left : TILE;
#pragma unroll
  for (int part = 0; part < 2; ++part) {
    const _Float16* base = (part ? Il : Ih) + row0 * D;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base), 0, rows_here * ROW_B, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // 16 instructions per term and stage, 4 per wave
      const int rbase = (wave * 4 + i) * 4;
      const int row = rbase + (lane >> 4);
      const int c = (lane & 15) ^ (row & 15);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(stage + part * PART_B + rbase * ROW_B), 16,
                                               row * ROW_B + 16 * c, 0, 0, 0);
    }
  }
}

// one ring stage (TILE item rows) against the wave's 64 stationary users
__device__ __forceinline__ void tile_scores(const char* cur, const u32x4 (&uh)[2][8], const u32x4 (&ul)[2][8], float (&mx)[2],
                                            float (&sm)[2], int64_t tile_item0, int64_t n1, float out_scale_log2, int r, int h) {
#pragma unroll
  for (int jt = 0; jt < TILE / 32; ++jt) {
      f32x16 a1[2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) a1[s][e] = 0.f;
      const int row = jt * 32 + r;
      const char* rowp = cur + row * ROW_B;
      u32x4 ih[2], il[2];  // A fragments, read one k-step ahead of their MFMAs
      {
        const int off = (h ^ (row & 15)) * 16;
        ih[0] = *reinterpret_cast<const u32x4*>(rowp + off);
        il[0] = *reinterpret_cast<const u32x4*>(rowp + PART_B + off);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (t + 1 < 8) {
          const int off = ((2 * (t + 1) + h) ^ (row & 15)) * 16;
          ih[(t + 1) & 1] = *reinterpret_cast<const u32x4*>(rowp + off);
          il[(t + 1) & 1] = *reinterpret_cast<const u32x4*>(rowp + PART_B + off);
        }
        const f16x8 ah = __builtin_bit_cast(f16x8, ih[t & 1]), al = __builtin_bit_cast(f16x8, il[t & 1]);
#pragma unroll
        for (int s = 0; s < 2; ++s) a1[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, uh[s][t]), a1[s], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 2; ++s) a1[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, ul[s][t]), a1[s], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 2; ++s) a1[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, __builtin_bit_cast(f16x8, uh[s][t]), a1[s], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // keep the LDS reads one k-step ahead, not eight
      }
      // online log-sum-exp in the log2 domain; rows past n1 were zero-filled by the descriptor: mask them
      const int64_t item0 = tile_item0 + jt * 32;
      const bool partial = item0 + 32 > n1;  // wave-uniform
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[16];
        float tmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          v[e] = a1[s][e] * out_scale_log2;
          if (partial && item0 + brow(e, h) >= n1) v[e] = -INFINITY;
          tmax = fmaxf(tmax, v[e]);
        }
        const float nm = fmaxf(mx[s], tmax);
        float add = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) add += __builtin_amdgcn_exp2f(v[e] - nm);
        sm[s] = sm[s] * __builtin_amdgcn_exp2f(mx[s] - nm) + add;
        mx[s] = nm;
      }
    }
}

// lse[m] = log sum_n exp(S[m][n]) over the item range of this workgroup's split; partial (max, sum) per split
__global__ __launch_bounds__(256, 2) void lse_f16x2_kernel(const _Float16* __restrict__ Uh, const _Float16* __restrict__ Ul,
                                                           const _Float16* __restrict__ Ih, const _Float16* __restrict__ Il,
                                                           int64_t M, int64_t N, float out_scale_log2, float* __restrict__ pmax,
                                                           float* __restrict__ psum, int n_splits) {
  __shared__ __attribute__((aligned(1024))) char ring0[STAGE_B];
  __shared__ __attribute__((aligned(1024))) char ring1[STAGE_B];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int64_t user0 = (int64_t)blockIdx.x * USERS_PER_WG + wave * 64;
  const int split = blockIdx.y;
  const int64_t per = ((N + n_splits - 1) / n_splits + TILE - 1) / TILE * TILE;
  const int64_t n0 = split * per, n1 = n0 + per < N ? n0 + per : N;

  // stationary: B fragments of 2 x 32 users, both terms, 8 k-steps: lane (user r, half h) holds k = 16 t + 8 h .. + 7
  u32x4 uh[2][8], ul[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    int64_t u = user0 + 32 * s + r;
    if (u >= M) u = M - 1;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      uh[s][t] = *reinterpret_cast<const u32x4*>(Uh + u * D + 16 * t + 8 * h);
      ul[s][t] = *reinterpret_cast<const u32x4*>(Ul + u * D + 16 * t + 8 * h);
    }
  }
  float mx[2] = {-INFINITY, -INFINITY}, sm[2] = {0.f, 0.f};

  const int n_tiles = n1 > n0 ? (int)((n1 - n0 + TILE - 1) / TILE) : 0;
  if (n_tiles > 0) stage_dma(Ih, Il, n0, n1, ring0, wave, lane);
  for (int tile = 0; tile < n_tiles; tile += 2) {  // two NAMED stages, unrolled by two: distinct LDS objects carry alias scopes
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this tile has landed (the only vector-memory traffic in the loop)
    __builtin_amdgcn_s_barrier();        // ... for every wave, and everybody is done reading the other stage
    if (tile + 1 < n_tiles) stage_dma(Ih, Il, n0 + (int64_t)(tile + 1) * TILE, n1, ring1, wave, lane);
    tile_scores(ring0, uh, ul, mx, sm, n0 + (int64_t)tile * TILE, n1, out_scale_log2, r, h);
    if (tile + 1 >= n_tiles) break;
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __builtin_amdgcn_s_barrier();
    if (tile + 2 < n_tiles) stage_dma(Ih, Il, n0 + (int64_t)(tile + 2) * TILE, n1, ring0, wave, lane);
    tile_scores(ring1, uh, ul, mx, sm, n0 + (int64_t)(tile + 1) * TILE, n1, out_scale_log2, r, h);
  }
  // the two lane halves hold disjoint item subsets of the same user
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float om = __shfl_xor(mx[s], 32), os = __shfl_xor(sm[s], 32);
    const float nm = fmaxf(mx[s], om);
    const float tot = (nm == -INFINITY) ? 0.f : sm[s] * __builtin_amdgcn_exp2f(mx[s] - nm) + os * __builtin_amdgcn_exp2f(om - nm);
    const int64_t u = user0 + 32 * s + r;
    if (h == 0 && u < M) {
      pmax[(int64_t)split * M + u] = nm;
      psum[(int64_t)split * M + u] = tot;
    }
  }
}

__global__ void lse_merge_kernel(const float* __restrict__ pmax, const float* __restrict__ psum, int64_t M, int n_splits,
                                 float* __restrict__ lse) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= M) return;
  float m = -INFINITY;
  for (int s = 0; s < n_splits; ++s) m = fmaxf(m, pmax[(int64_t)s * M + u]);
  float t = 0.f;
  for (int s = 0; s < n_splits; ++s) t += psum[(int64_t)s * M + u] * exp2f(pmax[(int64_t)s * M + u] - m);
  lse[u] = (m + log2f(t)) * 0.6931471805599453f;
}

// ---------------------------------------------------------------- host side
static float scale_pow2(const std::vector<float>& x) {
  float mx = 0.f;
  for (float v : x) mx = std::max(mx, fabsf(v));
  int e;
  frexpf(mx, &e);  // mx = f * 2^e, f in [0.5, 1)
  return ldexpf(1.f, 15 - e);
}
static void split_f16(const std::vector<float>& x, float s, std::vector<_Float16>& hi, std::vector<_Float16>& lo) {
  hi.resize(x.size());
  lo.resize(x.size());
  for (size_t i = 0; i < x.size(); ++i) {
    const float xs = x[i] * s;
    const _Float16 hh = (_Float16)xs;
    hi[i] = hh;
    lo[i] = (_Float16)(xs - (float)hh);
  }
}
static float bf16_round(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  float y;
  memcpy(&y, &u, 4);
  return y;
}

int main() {
  const int64_t M = 8192, N = 65536;
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> U(M * D), I(N * D);
  for (auto& v : U) v = nd(rng) * 0.35f;  // logits ~ N(0, 0.35^4 * 128 = 1.9): spread of a few units, like a trained model
  for (auto& v : I) v = nd(rng) * 0.35f;
  const float su = scale_pow2(U), si = scale_pow2(I);
  std::vector<_Float16> Uh, Ul, Ih, Il;
  split_f16(U, su, Uh, Ul);
  split_f16(I, si, Ih, Il);
  printf("scales: s_u = %g, s_i = %g\n", su, si);

  // ---- 1. element-wise error of the product forms (host emulation: exact products, fp32 accumulate, k ascending)
  {
    const int pairs = 200000;
    double e32 = 0, e16 = 0, eb3 = 0, eb6 = 0, m32 = 0, m16 = 0, mb3 = 0, mb6 = 0, ref_abs = 0;
    std::uniform_int_distribution<int64_t> du(0, M - 1), di(0, N - 1);
    for (int p = 0; p < pairs; ++p) {
      const int64_t u = du(rng), i = di(rng);
      double ref = 0, norm = 0;
      float f32 = 0.f, a1 = 0.f, b3 = 0.f, b6 = 0.f;
      for (int k = 0; k < D; ++k) {
        const float x = U[u * D + k], y = I[i * D + k];
        ref += (double)x * y;
        norm += fabs((double)x * y);
        f32 = fmaf(x, y, f32);
        const float xh = (float)Uh[u * D + k], xl = (float)Ul[u * D + k], yh = (float)Ih[i * D + k], yl = (float)Il[i * D + k];
        a1 = fmaf(xh, yh, a1);
        a1 = fmaf(xh, yl, a1);
        a1 = fmaf(xl, yh, a1);
        const float x0 = bf16_round(x), x1 = bf16_round(x - x0), x2 = bf16_round(x - x0 - x1);
        const float y0 = bf16_round(y), y1 = bf16_round(y - y0), y2 = bf16_round(y - y0 - y1);
        b3 = fmaf(x0, y0, b3); b3 = fmaf(x0, y1, b3); b3 = fmaf(x1, y0, b3);
        b6 = fmaf(x0, y0, b6); b6 = fmaf(x0, y1, b6); b6 = fmaf(x1, y0, b6);
        b6 = fmaf(x0, y2, b6); b6 = fmaf(x2, y0, b6); b6 = fmaf(x1, y1, b6);
      }
      const float f16 = a1 / (su * si);
      const double d32 = fabs(f32 - ref) / norm, d16 = fabs(f16 - ref) / norm, d3 = fabs(b3 - ref) / norm, d6 = fabs(b6 - ref) / norm;
      e32 += d32 * d32; e16 += d16 * d16; eb3 += d3 * d3; eb6 += d6 * d6;
      m32 = std::max(m32, d32); m16 = std::max(m16, d16); mb3 = std::max(mb3, d3); mb6 = std::max(mb6, d6);
      ref_abs += fabs(ref);
    }
    printf("element-wise |S - S64| / sum_k |u_k i_k| over %d random (user, item) pairs, D = %d (mean |S| = %.3f):\n", pairs, D, ref_abs / pairs);
    printf("  fp32 fma chain     rms %.3e  max %.3e\n", sqrt(e32 / pairs), m32);
    printf("  fp16 x2, 3 products rms %.3e  max %.3e\n", sqrt(e16 / pairs), m16);
    printf("  bf16 x2, 3 products rms %.3e  max %.3e\n", sqrt(eb3 / pairs), mb3);
    printf("  bf16 x3, 6 products rms %.3e  max %.3e\n", sqrt(eb6 / pairs), mb6);
  }

  // ---- 2. device kernel
  _Float16 *dUh, *dUl, *dIh, *dIl;
  float *dpm, *dps, *dlse;
  const int n_splits = 16;
  CK(hipMalloc(&dUh, M * D * 2)); CK(hipMalloc(&dUl, M * D * 2)); CK(hipMalloc(&dIh, N * D * 2)); CK(hipMalloc(&dIl, N * D * 2));
  CK(hipMalloc(&dpm, n_splits * M * 4)); CK(hipMalloc(&dps, n_splits * M * 4)); CK(hipMalloc(&dlse, M * 4));
  CK(hipMemcpy(dUh, Uh.data(), M * D * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dUl, Ul.data(), M * D * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dIh, Ih.data(), N * D * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dIl, Il.data(), N * D * 2, hipMemcpyHostToDevice));
  const float out_scale_log2 = 1.4426950408889634f / (su * si);
  const dim3 grid((unsigned)((M + USERS_PER_WG - 1) / USERS_PER_WG), n_splits);
  auto launch = [&]() {
    hipLaunchKernelGGL(lse_f16x2_kernel, grid, dim3(256), 0, 0, dUh, dUl, dIh, dIl, M, N, out_scale_log2, dpm, dps, n_splits);
    hipLaunchKernelGGL(lse_merge_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, 0, dpm, dps, M, n_splits, dlse);
  };
  launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 20;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double flop = 2.0 * M * N * D;
  printf("lse_f16x2_kernel + merge: %.3f ms per launch at M = %lld, N = %lld, D = %d: %.0f TF/s fp32-equivalent "
         "(3 fp16 MFMA products: %.2f PF/s on the matrix pipe); %u workgroups\n",
         ms, (long long)M, (long long)N, D, flop / ms * 1e-9, 3 * flop / ms * 1e-12, grid.x * grid.y);
  std::vector<float> lse(M);
  CK(hipMemcpy(lse.data(), dlse, M * 4, hipMemcpyDeviceToHost));
  double worst = 0, rms = 0;
  const int check = 48;
  for (int q = 0; q < check; ++q) {
    const int64_t u = (int64_t)q * (M / check) + (q % 7);
    std::vector<double> s(N);
    double mxv = -1e300;
    for (int64_t i = 0; i < N; ++i) {
      double a = 0;
      for (int k = 0; k < D; ++k) a += (double)U[u * D + k] * I[i * D + k];
      s[i] = a;
      mxv = std::max(mxv, a);
    }
    double t = 0;
    for (int64_t i = 0; i < N; ++i) t += exp(s[i] - mxv);
    const double ref = mxv + log(t);
    worst = std::max(worst, fabs(lse[u] - ref));
    rms += (lse[u] - ref) * (lse[u] - ref);
    if (q < 3) printf("  user %lld: lse %.7f  float64 %.7f\n", (long long)u, lse[u], ref);
  }
  printf("lse vs float64 over %d users: max |err| %.3e, rms %.3e (fp32 resolution at lse ~ 12: 9.5e-7)\n", check, worst, sqrt(rms / check));
  return 0;
}
