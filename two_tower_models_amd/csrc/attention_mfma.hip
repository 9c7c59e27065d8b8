// K4 (fast path): the per-(sample, head) attention core on the matrix cores, for the BASELINE
// shape class H <= 64 (padded to 64), dh in {16, 32, 64}.  attention.hip keeps the generic
// VALU kernels for everything else.
//
// One WAVEFRONT owns one (sample, head); nothing is shared between waves, so the kernels
// contain no workgroup barrier.  The matrices that act as the MFMA A operand (and as the
// element-wise Y of the PY product) sit in the wave's own LDS slice as row-major [64][dh+4]
// images (rows >= H are zero); the matrix on the B side of an NT product is held in REGISTERS
// as fragments loaded straight from global memory.  Only TWO images are resident per wave
// (forward: K, V; backward: K, V in orientation 1, then Q*scale, dO in orientation 2 reuse the
// same slice), 18 KiB at dh = 32, so 8 waves fit on a CU (2 per SIMD) -- with all operands in
// LDS it was 4, every load and MFMA dependency exposed.  Both operand forms of the 32x32x2
// fp32 MFMA:
//   * "NT" product  D[x][y] = sum_d A[x][d] B[y][d]  (A-rows / B-rows fetched with
//     ds_read_b128, four consecutive k per read, the k->(step, lane-half) permutation of
//     gemm.hip) -- result layout: lane = y, registers = 16 values of x;
//   * "PY" product  D[x][d] += sum_y G[x][y] Y[y][d] where G already sits in an NT result
//     (lane = x, registers = y): the accumulator registers ARE the A operand, Y is read
//     element-wise -- the trick of inbatch_ce.hip.
// Forward  : St[j][i] = K.Qs  -> softmax over j is lane-local (one query per lane)
//            O[i][:]  = (P/l).V  (PY product)            lse[i] = max + log(sum)
// Backward : orientation 1 (lane = query i):  St, dPt = V.dO, delta_i = sum_j P dP,
//                dSt = P (dPt - delta),  dQ = scale * dS.K  (PY)
//            orientation 2 (lane = key j):    S, dP = dO.V (registers = i, stats per register row
//                from LDS),  dV = P^T.dO (PY),  dK = dS^T.Qs (PY)
// 128 MFMAs forward, 448 backward per (sample, head) at dh = 32.
#include <stdlib.h>

#include "common.hpp"

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HP = 64;  // padded history length (two 32-row tiles)

__device__ __forceinline__ int arow(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

template <int DH>
struct AttnLds {
  static constexpr int LD = DH + 4;
  static constexpr int MAT = HP * LD;  // floats per staged matrix
};

// stage X[0:H, 0:DH] (row stride `ld`) into the wave's LDS image, zero-padding rows >= H
template <int DH>
__device__ __forceinline__ void stage(float* dst, const float* __restrict__ src, int64_t ld, int H, float scale, int lane) {
  constexpr int C4 = DH / 4, LD = DH + 4;
#pragma unroll
  for (int it = 0; it < HP * C4 / 64; ++it) {
    const int f = it * 64 + lane;
    const int row = f / C4, c4 = f % C4;
    // unconditional load from a clamped row, masked afterwards: a guarded load makes hipcc wait
    // for each one before evaluating the next guard (one exposed round trip per load)
    const bool ok = row < H;
    const float* p = src + (int64_t)(ok ? row : H - 1) * ld + 4 * c4;
    const float x0 = p[0], x1 = p[1], x2 = p[2], x3 = p[3];
    const float4 v = make_float4(ok ? x0 * scale : 0.f, ok ? x1 * scale : 0.f, ok ? x2 * scale : 0.f, ok ? x3 * scale : 0.f);
    *reinterpret_cast<float4*>(dst + row * LD + 4 * c4) = v;
  }
}

// B-side fragments of an NT product, straight from global memory: lane (r, h) holds row
// `row` of the matrix, elements k = 8g + 4h + c (the k-permutation of gemm.hip).  Rows >= H are 0.
template <int DH>
__device__ __forceinline__ void load_bfrag(float4 (&bf)[DH / 8], const float* __restrict__ src, int64_t ld, int row,
                                           int H, float scale, int h) {
  const bool ok = row < H;
  const float* p = src + (int64_t)(ok ? row : H - 1) * ld + 4 * h;  // clamped row, masked below
#pragma unroll
  for (int g = 0; g < DH / 8; ++g) {
    const float x0 = p[8 * g], x1 = p[8 * g + 1], x2 = p[8 * g + 2], x3 = p[8 * g + 3];
    bf[g] = make_float4(ok ? x0 * scale : 0.f, ok ? x1 * scale : 0.f, ok ? x2 * scale : 0.f, ok ? x3 * scale : 0.f);
  }
}
// D[x][y] = sum_d A[a0 + x][d] * B[b0 + y][d] with the B rows in registers (load_bfrag of rows
// b0 + r); result: lane&31 = y, registers = x rows arow(e,h)
template <int DH>
__device__ __forceinline__ f32x16 nt_tile_rb(const float* A, int a0, const float4 (&bf)[DH / 8], int r, int h) {
  constexpr int LD = DH + 4;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const float* ap = A + (a0 + r) * LD + 4 * h;
#pragma unroll
  for (int g = 0; g < DH / 8; ++g) {
    const float4 a = *reinterpret_cast<const float4*>(ap + 8 * g);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bf[g].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bf[g].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bf[g].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bf[g].w, acc, 0, 0, 0);
  }
  return acc;
}

// out[dt][x][d] += sum over the 32 y of this tile: G[x][y0 + y] * Y[y0 + y][32*dt + d]
// G: an NT result (lane = x, registers = y).  out layout: lane&31 = d, registers = x rows.
template <int DH>
__device__ __forceinline__ void py_accum(f32x16 (&out)[(DH + 31) / 32], const f32x16& G, const float* Y, int y0,
                                         int r, int h) {
  constexpr int LD = DH + 4, TD = (DH + 31) / 32;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float* yrow = Y + (y0 + (e & 3) + 8 * (e >> 2) + 4 * h) * LD;
#pragma unroll
    for (int d = 0; d < TD; ++d) {
      const int col = 32 * d + r;
      const float yv = (DH % 32 == 0 || col < DH) ? yrow[col] : 0.f;
      out[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(G[e], yv, out[d], 0, 0, 0);
    }
  }
}

// store an output tile (lane = d, registers = rows x0 + arow) to dst[row*ld + d], rows < H only
template <int DH>
__device__ __forceinline__ void store_rows(float* __restrict__ dst, int64_t ld, const f32x16 (&t)[(DH + 31) / 32],
                                           int x0, int H, float scale, int r, int h) {
  constexpr int TD = (DH + 31) / 32;
#pragma unroll
  for (int d = 0; d < TD; ++d) {
    const int col = 32 * d + r;
    if (col >= DH) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = x0 + arow(e, h);
      if (row < H) dst[(int64_t)row * ld + col] = t[d][e] * scale;
    }
  }
}

// ------------------------------------------------------------------ forward
template <int DH, int WAVES>
__global__ __launch_bounds__(64 * WAVES, (WAVES == 4 ? 2 : 1)) void attn_fwd_mfma_kernel(const float* __restrict__ qkv, int64_t n_pairs,
                                                                   int H, int D, int heads,
                                                                   float* __restrict__ ctx, float* __restrict__ lse) {
  using L = AttnLds<DH>;
  constexpr int TD = (DH + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int64_t pair = (int64_t)blockIdx.x * WAVES + wave;
  if (pair >= n_pairs) return;
  float* Ks = reinterpret_cast<float*>(smem_raw) + wave * 2 * L::MAT;
  float* Vs = Ks + L::MAT;
  const int64_t b = pair / heads, hd = pair % heads;
  const float* base = qkv + b * H * 3 * (int64_t)D + hd * DH;
  const float scale = 1.0f / sqrtf((float)DH);
  // every global load of the pair goes out BEFORE the first wait: K and V rows for the LDS images and the Q fragments
  // of both query tiles.  (Staged one after the other -- K, V, then Q per tile -- the wave sat out four memory round
  // trips per pair with one partner wave to cover them: 136 us per layer against 55 us of MFMA time.)
  constexpr int C4 = DH / 4, NST = HP * C4 / 64;
  float4 kst[NST], vst[NST];
#pragma unroll
  for (int it = 0; it < NST; ++it) {
    const int f = it * 64 + lane, row = f / C4, c4 = f % C4;
    const float* pr = base + (int64_t)(row < H ? row : H - 1) * (3 * D) + 4 * c4;
    kst[it] = *reinterpret_cast<const float4*>(pr + D);
    vst[it] = *reinterpret_cast<const float4*>(pr + 2 * D);
  }
  float4 qf0[DH / 8], qf1[DH / 8];  // (two named arrays: a reference to a row of a 2-D array keeps the whole array in scratch memory)
  load_bfrag<DH>(qf0, base, 3 * D, r, H, scale, h);
  load_bfrag<DH>(qf1, base, 3 * D, 32 + r, H, scale, h);
#pragma unroll
  for (int it = 0; it < NST; ++it) {
    const int f = it * 64 + lane, row = f / C4, c4 = f % C4;
    const bool ok = row < H;
    // (by value, then masked: `ok ? kst[it] : z` selects between ADDRESSES of two structs and parks the arrays in scratch)
    float4 kv = kst[it], vv = vst[it];
    const float m = ok ? 1.f : 0.f;
    kv.x *= m; kv.y *= m; kv.z *= m; kv.w *= m;
    vv.x *= m; vv.y *= m; vv.z *= m; vv.w *= m;
    *reinterpret_cast<float4*>(Ks + row * L::LD + 4 * c4) = kv;
    *reinterpret_cast<float4*>(Vs + row * L::LD + 4 * c4) = vv;
  }
  __builtin_amdgcn_wave_barrier();

  float* out = ctx + b * H * (int64_t)D + hd * DH;
  auto query_tile = [&](int it, const float4 (&qf)[DH / 8]) {  // 32 queries at a time: lane r <-> query it*32 + r
    f32x16 st[2];
    float mx = -3.0e38f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      st[jt] = nt_tile_rb<DH>(Ks, jt * 32, qf, r, h);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const bool valid = jt * 32 + arow(e, h) < H;
        st[jt][e] = valid ? st[jt][e] : -3.0e38f;
        mx = fmaxf(mx, st[jt][e]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        st[jt][e] = __expf(st[jt][e] - mx);  // masked entries: exp(-huge) = 0
        l += st[jt][e];
      }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    f32x16 o[TD];
#pragma unroll
    for (int d = 0; d < TD; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) st[jt][e] *= inv;
      py_accum<DH>(o, st[jt], Vs, jt * 32, r, h);
    }
    store_rows<DH>(out, D, o, it * 32, H, 1.f, r, h);
    const int i = it * 32 + r;
    if (h == 0 && i < H) lse[(b * heads + hd) * H + i] = mx + __logf(l);
  };
  query_tile(0, qf0);
  query_tile(1, qf1);
}

// ------------------------------------------------------------------ backward
template <int DH, int WAVES>
__global__ __launch_bounds__(64 * WAVES, (WAVES == 4 ? 2 : 1)) void attn_bwd_mfma_kernel(const float* __restrict__ qkv,
                                                                   const float* __restrict__ lse,
                                                                   const float* __restrict__ d_ctx, int64_t n_pairs,
                                                                   int H, int D, int heads,
                                                                   float* __restrict__ d_qkv) {
  using L = AttnLds<DH>;
  constexpr int TD = (DH + 31) / 32;
  constexpr int PER_WAVE = 2 * L::MAT + 2 * HP;  // two images (K, V then Qs, dO) + lse[64] + delta[64]
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int64_t pair = (int64_t)blockIdx.x * WAVES + wave;
  if (pair >= n_pairs) return;
  float* M0 = reinterpret_cast<float*>(smem_raw) + wave * PER_WAVE;
  float* M1 = M0 + L::MAT;
  float* Ls = M1 + L::MAT;   // lse per query row (+huge for padding rows -> P = 0)
  float* Ds = Ls + HP;       // delta per query row
  const int64_t b = pair / heads, hd = pair % heads;
  const float* base = qkv + b * H * 3 * (int64_t)D + hd * DH;
  const float* gbase = d_ctx + b * H * (int64_t)D + hd * DH;
  const float scale = 1.0f / sqrtf((float)DH);
  float* Ks = M0;
  float* Vs = M1;
  stage<DH>(Ks, base + D, 3 * D, H, 1.f, lane);
  stage<DH>(Vs, base + 2 * D, 3 * D, H, 1.f, lane);
  Ls[lane] = (lane < H) ? lse[(b * heads + hd) * H + lane] : 3.0e38f;
  __builtin_amdgcn_wave_barrier();
  float* obase = d_qkv + b * H * 3 * (int64_t)D + hd * DH;

  // ---- orientation 1: lane = query i.  dQ and delta.
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = it * 32 + r;
    const float li = Ls[i];
    float4 qf[DH / 8], gf[DH / 8];
    load_bfrag<DH>(qf, base, 3 * D, i, H, scale, h);
    load_bfrag<DH>(gf, gbase, D, i, H, 1.f, h);
    f32x16 pt[2], dpt[2];
    float dl = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      pt[jt] = nt_tile_rb<DH>(Ks, jt * 32, qf, r, h);   // St[j][i]
      dpt[jt] = nt_tile_rb<DH>(Vs, jt * 32, gf, r, h);  // dPt[j][i] = V_j . dO_i
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const bool valid = jt * 32 + arow(e, h) < H;
        pt[jt][e] = valid ? __expf(pt[jt][e] - li) : 0.f;
        dl = fmaf(pt[jt][e], dpt[jt][e], dl);
      }
    }
    dl += __shfl_xor(dl, 32, 64);
    if (h == 0) Ds[i] = dl;
    f32x16 dq[TD];
#pragma unroll
    for (int d = 0; d < TD; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[d][e] = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) pt[jt][e] *= (dpt[jt][e] - dl);  // dSt[j][i]
      py_accum<DH>(dq, pt[jt], Ks, jt * 32, r, h);
    }
    store_rows<DH>(obase, 3 * D, dq, it * 32, H, scale, r, h);
  }
  __builtin_amdgcn_wave_barrier();

  // ---- orientation 2: lane = key j.  dV and dK (reductions over the queries i).  The slice now
  // holds Q*scale and dO (A operands and PY sources); K and V move to registers.
  float* Qs = M0;
  float* Gs = M1;
  stage<DH>(Qs, base, 3 * D, H, scale, lane);
  stage<DH>(Gs, gbase, D, H, 1.f, lane);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const bool jvalid = jt * 32 + r < H;
    float4 kf[DH / 8], vf[DH / 8];
    load_bfrag<DH>(kf, base + D, 3 * D, jt * 32 + r, H, 1.f, h);
    load_bfrag<DH>(vf, base + 2 * D, 3 * D, jt * 32 + r, H, 1.f, h);
    f32x16 dv[TD], dk[TD];
#pragma unroll
    for (int d = 0; d < TD; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) { dv[d][e] = 0.f; dk[d][e] = 0.f; }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      f32x16 p = nt_tile_rb<DH>(Qs, it * 32, kf, r, h);    // S[i][j]
      f32x16 dp = nt_tile_rb<DH>(Gs, it * 32, vf, r, h);   // dP[i][j] = dO_i . V_j
      const float* lrow = Ls + it * 32 + 4 * h;
      const float* drow = Ds + it * 32 + 4 * h;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 l4 = *reinterpret_cast<const float4*>(lrow + 8 * q);
        const float4 d4 = *reinterpret_cast<const float4*>(drow + 8 * q);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int e = 4 * q + c;
          const float pe = jvalid ? __expf(p[e] - lv[c]) : 0.f;  // padding query rows: lse = +huge -> 0
          p[e] = pe;
          dp[e] = pe * (dp[e] - dv4[c]);  // dS[i][j]
        }
      }
      py_accum<DH>(dv, p, Gs, it * 32, r, h);
      py_accum<DH>(dk, dp, Qs, it * 32, r, h);  // Qs carries the 1/sqrt(dh) factor
    }
    store_rows<DH>(obase + 2 * D, 3 * D, dv, jt * 32, H, 1.f, r, h);
    store_rows<DH>(obase + D, 3 * D, dk, jt * 32, H, 1.f, r, h);
  }
}

template <typename K>
static int lds_opt_in(K kernel, size_t lds, const char* name) {
  if (lds <= 64 * 1024) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e)); return (int)e; }
  return 0;
}

template <int DH>
static int launch_fwd(const float* qkv, int64_t n_pairs, int H, int D, int heads, float* ctx, float* lse, hipStream_t st) {
  constexpr int WAVES = 4;
  const size_t lds = (size_t)WAVES * 2 * AttnLds<DH>::MAT * sizeof(float);
  int rc = lds_opt_in(attn_fwd_mfma_kernel<DH, WAVES>, lds, "attn_fwd_mfma_kernel");
  if (rc) return rc;
  ProfScope prof("attn_fwd_mfma_kernel", st);
  attn_fwd_mfma_kernel<DH, WAVES><<<(unsigned)ceil_div(n_pairs, WAVES), 64 * WAVES, lds, st>>>(qkv, n_pairs, H, D, heads, ctx, lse);
  return check_launch("attn_fwd_mfma_kernel");
}
template <int DH>
static int launch_bwd(const float* qkv, const float* lse, const float* d_ctx, int64_t n_pairs, int H, int D, int heads,
                      float* d_qkv, hipStream_t st) {
  constexpr int WAVES = (DH <= 32) ? 4 : 2;
  const size_t lds = (size_t)WAVES * (2 * AttnLds<DH>::MAT + 2 * HP) * sizeof(float);
  int rc = lds_opt_in(attn_bwd_mfma_kernel<DH, WAVES>, lds, "attn_bwd_mfma_kernel");
  if (rc) return rc;
  ProfScope prof("attn_bwd_mfma_kernel", st);
  attn_bwd_mfma_kernel<DH, WAVES><<<(unsigned)ceil_div(n_pairs, WAVES), 64 * WAVES, lds, st>>>(qkv, lse, d_ctx, n_pairs, H, D, heads, d_qkv);
  return check_launch("attn_bwd_mfma_kernel");
}

bool attn_mfma_supported(const void* a, const void* b, int64_t H, int64_t D, int64_t dh) {
  const bool al = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
  return al && H <= HP && (dh == 16 || dh == 32 || dh == 64) && D % 4 == 0;
}

int attn_fwd_mfma(const float* qkv, int64_t B, int64_t H, int64_t D, int64_t heads, float* ctx, float* lse, hipStream_t st) {
  const int64_t dh = D / heads, n = B * heads;
  if (dh == 16) return launch_fwd<16>(qkv, n, (int)H, (int)D, (int)heads, ctx, lse, st);
  if (dh == 32) return launch_fwd<32>(qkv, n, (int)H, (int)D, (int)heads, ctx, lse, st);
  return launch_fwd<64>(qkv, n, (int)H, (int)D, (int)heads, ctx, lse, st);
}
int attn_bwd_mfma(const float* qkv, const float* lse, const float* d_ctx, int64_t B, int64_t H, int64_t D,
                  int64_t heads, float* d_qkv, hipStream_t st) {
  const int64_t dh = D / heads, n = B * heads;
  if (dh == 16) return launch_bwd<16>(qkv, lse, d_ctx, n, (int)H, (int)D, (int)heads, d_qkv, st);
  if (dh == 32) return launch_bwd<32>(qkv, lse, d_ctx, n, (int)H, (int)D, (int)heads, d_qkv, st);
  return launch_bwd<64>(qkv, lse, d_ctx, n, (int)H, (int)D, (int)heads, d_qkv, st);
}

}  // namespace tt
