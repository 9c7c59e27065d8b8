// K3 (tall-M fast path): "weights-stationary" fp32 MFMA GEMM for the Linear-layer products
// with a huge row count and a small weight matrix -- the encoder projections
// ([B*H, D] x [D, 3D], [B*H, D] x [D, D]) and their input gradients:
//     NT   C[m][n] = sum_k A[m][k] W[n][k]       (y  = x W^T + b)
//     NN   C[m][n] = sum_k A[m][k] W[k][n]       (dx = dy W)
// Each wave keeps its 32 output columns' weights in registers for the whole kernel and the
// activation rows stream through double-buffered LDS (LDS-DMA when K in {32,64,128,256} and
// rows are 16-B aligned), the structure of inbatch_ce.hip with the MFMA operand roles swapped:
// the weights are the A operand, so the 32x32 result tile has one output ROW per lane and, in
// that lane's 16 registers, four runs of four consecutive columns -> the tile is written with
// 16-B stores (4 per lane instead of 16 scalar ones; the scalar form was store-issue bound).  One workgroup covers 128 columns and
// walks its share of the 64-row tiles; A is read N/128 times (L2 / MALL), W once per workgroup.
// gemm.hip's tiled kernel remains the general path (TN products, small M, K > 256).
#include <stdlib.h>

#include "mfma_stream.hpp"

namespace tt {

struct WsArgs {
  const float* A;
  const float* W;
  float* C;
  const float* bias;
  const float* aux;
  int64_t M, N, K, lda, ldw, ldc, ldaux;
  int colblocks, rowgroups;  // grid = colblocks * rowgroups workgroups, 1-D
  int a_vec, w_vec, c_vec, epilogue, accumulate, epi_after_acc;
};

// stationary fragments from a [K][N] (row = reduction index) weight: xr[g][c] = W[8g+4h+c][n]
template <int DP8>
__device__ __forceinline__ void load_stationary_t(float (&xr)[DP8][4], const float* __restrict__ W, int64_t ld,
                                                  int64_t n, int64_t N, int64_t K, int h) {
#pragma unroll
  for (int g = 0; g < DP8; ++g)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int64_t k = 8 * g + 4 * h + c;
      xr[g][c] = (n < N && k < K) ? W[k * ld + n] : 0.f;
    }
}

// Tt[n][m] = sum_k W[n][k] * A[m][k]: lane&31 = streamed row m, registers = 16 of the wave's 32 columns
template <int DP8, bool GLDS>
__device__ __forceinline__ f32x16 out_tile(const float* Ys, const float (&xr)[DP8][4], int jt, int r, int h) {
  using TM = TileMap<DP8, GLDS>;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int row = jt * 32 + r;
  float4 y[2];
  y[0] = *reinterpret_cast<const float4*>(Ys + TM::chunk(row, h));
#pragma unroll
  for (int g = 0; g < DP8; ++g) {
    if (g + 1 < DP8) y[(g + 1) & 1] = *reinterpret_cast<const float4*>(Ys + TM::chunk(row, 2 * (g + 1) + h));
    const float4 v = y[g & 1];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[g][0], v.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[g][1], v.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[g][2], v.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[g][3], v.w, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  return acc;
}

// MODE is compile time on purpose: a run-time "maybe load the old value / the mask" in the store
// path makes hipcc place an unconditional s_waitcnt vmcnt(0) before every store, which also
// drains the LDS-DMA of the next tile.
//   WS_PLAIN   C = A.W + bias              16-B stores, no loads            (the forward projections)
//   WS_ACC     C += A.W                    16-B load + store                (second K pass)
//   WS_GENERIC run-time epilogue / accumulate / ragged or unaligned columns
constexpr int WS_PLAIN = 0, WS_ACC = 1, WS_GENERIC = 2;

template <int DP8, bool GLDS, bool W_TRANS, int MODE>
__global__ __launch_bounds__(256, ((GLDS && DP8 <= 16) ? 2 : 1)) void gemm_ws_kernel(const WsArgs p) {
  using TM = TileMap<DP8, GLDS>;
  constexpr int TILE_FLOATS = BJ * TM::LD;
  // two NAMED tile buffers, loop unrolled by two: LDS reads of one buffer then do not wait for the
  // LDS-DMA in flight into the other (hipcc cannot tell two halves of one dynamic block apart)
  __shared__ __attribute__((aligned(16))) float buf0[TILE_FLOATS];
  __shared__ __attribute__((aligned(16))) float buf1[TILE_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  // XCD-aware decomposition of the 1-D grid (workgroup b runs on XCD b % 8): the `colblocks`
  // workgroups that stream the SAME rows get consecutive slots on ONE XCD, so the activation
  // tile is fetched from HBM once and hit in that XCD's L2 by the others.
  const int L = blockIdx.x, nwg = p.colblocks * p.rowgroups;
  int cb, rg;
  if (nwg % 8 == 0 && p.rowgroups % 8 == 0) {
    const int xcd = L & 7, idx = L >> 3;
    rg = (idx / p.colblocks) * 8 + xcd;
    cb = idx % p.colblocks;
  } else {
    rg = L / p.colblocks;
    cb = L % p.colblocks;
  }
  const int64_t nw = (int64_t)cb * BI + wave * 32;  // first of this wave's 32 output columns
  const int64_t n = nw + r;                                 // the column whose weights this lane loads

  float xr[DP8][4];
  if constexpr (W_TRANS) load_stationary_t<DP8>(xr, p.W, p.ldw, n, p.N, p.K, h);
  else load_stationary<DP8>(xr, p.W, p.ldw, n, p.N, p.K, h, p.w_vec);
  // register e of a result tile is column nw + (e&3) + 8*(e>>2) + 4h
  float bias_e[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int64_t ne = nw + (e & 3) + 8 * (e >> 2) + 4 * h;
    bias_e[e] = (p.bias && ne < p.N) ? p.bias[ne] : 0.f;
  }
  const bool full_cols = nw + 32 <= p.N;

  // balanced partition of the 64-row tiles over the row groups (sizes differ by at most one)
  const int64_t ntiles_all = (p.M + BJ - 1) / BJ;
  const int64_t t0 = ntiles_all * rg / p.rowgroups;
  const int64_t t1 = ntiles_all * (rg + 1) / p.rowgroups;

  Stager<DP8, GLDS> stg;
  if (t0 < t1) {
    stg.issue(p.A, p.lda, t0 * BJ, p.M, p.K, p.a_vec, buf0, wave, lane);
    stg.land(buf0);
  }
  __syncthreads();
  // The tile's results are written by `emit`.  CDNA counts stores on vmcnt, the same counter the
  // LDS-DMA completion is waited on, so a store issued right before that wait would stall the
  // wave for a full HBM write latency: sub-tile 1's stores are DEFERRED into the next iteration
  // (after the next DMA issue), and sub-tile 0's are followed by sub-tile 1's 64 MFMAs.
  // always_inline: the run-time (WS_GENERIC) body is large enough that hipcc left it a FUNCTION with three call sites, and
  // everything it captures by reference (bias_e, the pending tile) then lived in scratch: 384 B per lane
  auto emit = [&](const f32x16& acc, int64_t m) __attribute__((always_inline)) {
    if constexpr (MODE != WS_GENERIC) {
      if (m < p.M && nw < p.N) {
        float* crow = p.C + ((int)m * (int)p.ldc + (int)nw + 4 * h);  // 32-bit offsets (host-checked)
        float4 o[4];
        if constexpr (MODE == WS_ACC) {
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = *reinterpret_cast<const float4*>(crow + 8 * q);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 v = make_float4(acc[4 * q] + bias_e[4 * q], acc[4 * q + 1] + bias_e[4 * q + 1],
                                 acc[4 * q + 2] + bias_e[4 * q + 2], acc[4 * q + 3] + bias_e[4 * q + 3]);
          if constexpr (MODE == WS_ACC) { v.x += o[q].x; v.y += o[q].y; v.z += o[q].z; v.w += o[q].w; }
          *reinterpret_cast<float4*>(crow + 8 * q) = v;
        }
      }
    } else {
      if (m < p.M) {
        // 32-bit element offsets (the host guarantees M*ldc < 2^31)
        const int coff = (int)m * (int)p.ldc + (int)nw + 4 * h;
        const int aoff = (int)m * (int)p.ldaux + (int)nw + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = acc[4 * q + c] + bias_e[4 * q + c];
          if (p.c_vec && full_cols) {
            float4 old = make_float4(0.f, 0.f, 0.f, 0.f), mk = make_float4(1.f, 1.f, 1.f, 1.f);
            float4* dst = reinterpret_cast<float4*>(p.C + coff + 8 * q);
            if (p.epilogue == TT_EPI_RELU_MASK) mk = *reinterpret_cast<const float4*>(p.aux + aoff + 8 * q);
            if (p.accumulate) old = *dst;
            const float mkv[4] = {mk.x, mk.y, mk.z, mk.w}, ov[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (p.epi_after_acc) v[c] += ov[c];
              if (p.epilogue == TT_EPI_RELU) v[c] = fmaxf(v[c], 0.f);
              else if (p.epilogue == TT_EPI_RELU_MASK) v[c] = (mkv[c] > 0.f) ? v[c] : 0.f;
              if (!p.epi_after_acc) v[c] += ov[c];
            }
            *dst = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (nw + 8 * q + 4 * h + c >= p.N) continue;
              float* dst = p.C + coff + 8 * q + c;
              const float old = p.accumulate ? *dst : 0.f;
              float x = v[c];
              if (p.epi_after_acc) x += old;
              if (p.epilogue == TT_EPI_RELU) x = fmaxf(x, 0.f);
              else if (p.epilogue == TT_EPI_RELU_MASK) x = (p.aux[aoff + 8 * q + c] > 0.f) ? x : 0.f;
              if (!p.epi_after_acc) x += old;
              *dst = x;
            }
          }
        }
      }
    }
  };
  f32x16 pend;
  int64_t pend_m = -1;
  auto step = [&](int64_t t, const float* ys, float* nxt) {
    // the deferred stores go out BEFORE the next DMA is queued: anything they wait on is older
    if (pend_m >= 0) emit(pend, pend_m);
    if (t + 1 < t1) stg.issue(p.A, p.lda, (t + 1) * BJ, p.M, p.K, p.a_vec, nxt, wave, lane);
    const f32x16 acc0 = out_tile<DP8, GLDS>(ys, xr, 0, r, h);
    emit(acc0, t * BJ + r);
    pend = out_tile<DP8, GLDS>(ys, xr, 1, r, h);
    pend_m = t * BJ + 32 + r;
    if (t + 1 < t1) stg.land(nxt);
    __syncthreads();
  };
  for (int64_t t = t0; t < t1; t += 2) {
    step(t, buf0, buf1);
    if (t + 1 < t1) step(t + 1, buf1, buf0);
  }
  if (pend_m >= 0) emit(pend, pend_m);
}

template <int DP8, bool GLDS, bool WT, int MODE>
static int launch_ws(const WsArgs& a, dim3 grid, hipStream_t st) {
  ProfScope prof("gemm_ws_kernel", st);
  gemm_ws_kernel<DP8, GLDS, WT, MODE><<<grid, 256, 0, st>>>(a);  // LDS is static: two named tile buffers
  return check_launch("gemm_ws_kernel");
}
template <bool GLDS, bool WT, int MODE>
static int dispatch_dp(int dp8, const WsArgs& a, dim3 grid, hipStream_t st) {
  switch (dp8) {
    case 4: return launch_ws<4, GLDS, WT, MODE>(a, grid, st);
    case 8: return launch_ws<8, GLDS, WT, MODE>(a, grid, st);
    case 16: return launch_ws<16, GLDS, WT, MODE>(a, grid, st);
    default: return launch_ws<32, GLDS, WT, MODE>(a, grid, st);
  }
}
template <bool GLDS, bool WT>
static int dispatch_mode(int mode, int dp8, const WsArgs& a, dim3 grid, hipStream_t st) {
  if (mode == WS_PLAIN) return dispatch_dp<GLDS, WT, WS_PLAIN>(dp8, a, grid, st);
  if (mode == WS_ACC) return dispatch_dp<GLDS, WT, WS_ACC>(dp8, a, grid, st);
  return dispatch_dp<GLDS, WT, WS_GENERIC>(dp8, a, grid, st);
}

static int ws_launch_one(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W,
                         int64_t ldw, float* C, int64_t ldc, const float* bias, int epilogue, const float* aux,
                         int64_t ldaux, int accumulate, int epi_after_acc, hipStream_t st) {
  const int dp8 = K <= 32 ? 4 : K <= 64 ? 8 : K <= 128 ? 16 : 32;
  WsArgs a{};
  a.A = A; a.W = W; a.C = C; a.bias = bias; a.aux = aux;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldaux = ldaux;
  a.epilogue = epilogue; a.accumulate = accumulate; a.epi_after_acc = epi_after_acc;
  a.a_vec = (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  a.w_vec = (ldw % 4 == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
  a.c_vec = (ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
            (!aux || ((ldaux % 4 == 0) && ((reinterpret_cast<uintptr_t>(aux) & 15) == 0)));
  const bool dma = a.a_vec && K == dp8 * 8 && lda <= (1 << 22);  // tile_dma: 32-bit byte offsets within a tile
  // one resident wave of workgroups: 256 CUs x (2 at <=128 reduction columns, else 1) slots
  const int64_t tiles = ceil_div(M, BJ), colblocks = ceil_div(N, BI);
  const int64_t slots = 256 * ((dma && dp8 <= 16) ? 2 : 1);
  int64_t rowgroups = (slots / colblocks) / 8 * 8;
  if (rowgroups < 8) rowgroups = 8;
  if (rowgroups > tiles) rowgroups = tiles;
  a.colblocks = (int)colblocks;
  a.rowgroups = (int)rowgroups;
  dim3 grid((unsigned)(colblocks * rowgroups));
  const bool wt = layout == TT_GEMM_NN;
  int mode = WS_GENERIC;
  if (a.c_vec && N % 32 == 0 && epilogue == TT_EPI_NONE) mode = accumulate ? WS_ACC : WS_PLAIN;
  if (dma) return wt ? dispatch_mode<true, true>(mode, dp8, a, grid, st) : dispatch_mode<true, false>(mode, dp8, a, grid, st);
  return wt ? dispatch_mode<false, true>(mode, dp8, a, grid, st) : dispatch_mode<false, false>(mode, dp8, a, grid, st);
}

// Called by tt_gemm_f32 (gemm.hip).  Returns -100 when the shape is not for this kernel.
// K in (256, 512] runs as two accumulating passes over the reduction range.
int gemm_ws_try(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W,
                int64_t ldw, float* C, int64_t ldc, const float* bias, int epilogue, const float* aux,
                int64_t ldaux, int accumulate, hipStream_t st) {
  if (layout == TT_GEMM_TN || K > 512 || K < 1 || M < 16384) return -100;
  if (M * ldc >= ((int64_t)1 << 31) || (aux && M * ldaux >= ((int64_t)1 << 31))) return -100;  // 32-bit offsets inside
  if (K <= 256)
    return ws_launch_one(layout, M, N, K, A, lda, W, ldw, C, ldc, bias, epilogue, aux, ldaux, accumulate, 0, st);
  if (accumulate && epilogue != TT_EPI_NONE) return -100;  // would need a third pass
  // two passes re-read A and C; when the generic kernel can take its all-interior vector-load
  // form (gemm.hip, FULL) it is faster for these shapes (K = 384: 245 vs 296 us)
  if (M % 128 == 0 && N % 128 == 0 && K % 32 == 0 && lda % 4 == 0 && ldw % 4 == 0 &&
      ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) & 15) == 0)
    return -100;
  const int64_t K1 = 256, K2 = K - 256;
  int rc = ws_launch_one(layout, M, N, K1, A, lda, W, ldw, C, ldc, bias, TT_EPI_NONE, nullptr, 0, accumulate, 0, st);
  if (rc) return rc;
  const float* W2 = (layout == TT_GEMM_NN) ? W + K1 * ldw : W + K1;
  return ws_launch_one(layout, M, N, K2, A + K1, lda, W2, ldw, C, ldc, nullptr, epilogue, aux, ldaux, 1, 1, st);
}

}  // namespace tt
