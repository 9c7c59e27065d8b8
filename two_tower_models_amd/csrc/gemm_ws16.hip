// The encoder's tall-M projections -- in-projection x W_in^T (N = 384 or, last layer, 256; K = 128), out-projection
// (N = K = 128), its gradient d_ctx = dY W_out and the input gradients dx = dQKV W_in (K = 384) / dKV W_in[D:]
// (K = 256): ref:src/user_history_encoder.py:103-108 and its autograd --
//     NT   C[m][n] = sum_k A[m][k] W[n][k] (+ bias[n])        NN   C[m][n] = sum_k A[m][k] W[k][n]
// with M = B*H = 204 800 rows and (N / 128) * (K / 128) <= 3: the weights (<= 48 K floats) fit the registers of ONE
// eight-wave workgroup, 96 per lane.
//
// gemm_ws.hip gives each wave 32 output columns and all of K in registers, which stops at K = 256 (128 registers) and
// leaves N = 128 with four waves and a two-tile LDS pipeline: dx (K = 384) fell to the generic tiled kernel (230 us,
// 89 TF/s) and the K = 128 products ran at half the matrix rate next to anything that loads the memory system
// (77 - 94 us).  Here a wave owns 16 N/128 columns in tiles of SIXTEEN (v_mfma_f32_16x16x4_f32: exact fp32, 32 cycles,
// the weights are the A operand, so a result tile has one row per lane and four consecutive columns in its registers:
// 16-byte stores), eight waves cover the N columns -- the activations are read ONCE, not N / 128 times --, and the rows stream through a three-stage LDS
// ring of 48 KB stages (32 / 48 / 96 rows at K = 384 / 256 / 128) filled by LDS-DMA with a source-side XOR swizzle of
// the 16-byte chunks (conflict-free ds_read_b128 with one row per lane).  One bare s_barrier per stage and counted
// vmcnt waits that leave the younger stage AND the result stores in flight (gemm_tn_stream.hip has the reasoning).
// One persistent workgroup per CU over a contiguous range of stages.
#include <stdlib.h>

#include "common.hpp"

namespace tt {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Ws16Args {
  const float* A;
  const float* W;
  const float* bias;
  float* C;
  int64_t M, lda, ldw, ldc;
  // POOL form only: C[m][:] += pool_scale * pool[m / group][:] (the mean pool's backward folded into the dx product)
  const float* pool;
  int64_t ld_pool;
  uint32_t group;
  float pool_scale;
};

template <int KB, int NB>  // K = 128 * KB, N = 128 * NB
struct Ws16Cfg {
  static constexpr int K = 128 * KB;
  // rows per stage: 48 KB of activations -- less where NB result tiles per row tile would not fit the registers
  static constexpr int ROWS = KB == 3 ? 32 : KB == 2 ? 48 : NB == 3 ? 48 : NB == 2 ? 64 : 96;
  static constexpr int TILES = ROWS / 16;
  static constexpr int CPR = K / 4;                 // 16-byte chunks per row
  static constexpr int PIECES = ROWS * K / 256;     // 1-KiB DMA pieces per stage
  static constexpr int PPW = PIECES / 8;
  static constexpr int STAGE = ROWS * K;            // floats
  static_assert(PIECES % 8 == 0, "pieces must divide over the eight waves");
};

template <int N>
__device__ __forceinline__ void ws16_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LDS-DMA of one stage: this wave's PPW pieces.  Piece q = LDS floats [256 q, 256 q + 256): 64 chunks = 64 / CPR rows
// (K = 384: two thirds of a row); the lane fetches the SOURCE chunk that belongs at its linear LDS position after the
// swizzle c' = c ^ (row & 15).  Rows past `left` lie outside the descriptor and land as zeros.  (A free function:
// with the buffer builtins inside a lambda of the kernel the host pass drops the kernel's launch stub.)
template <int KB, int NB>
__device__ __forceinline__ void ws16_issue(const Ws16Args& p, float* stage, int64_t t, int64_t t1, int wave, int lane) {
  using Cf = Ws16Cfg<KB, NB>;
  const int64_t row0 = t * Cf::ROWS;
  int64_t left = t < t1 ? p.M - row0 : 0;
  if (left > Cf::ROWS) left = Cf::ROWS;
  if (left < 0) left = 0;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(left ? p.A + row0 * p.lda : p.A), 0, (int)left * (int)p.lda * 4, 0x00020000);
#pragma unroll
  for (int i = 0; i < Cf::PPW; ++i) {
    const int q = wave * Cf::PPW + i;
    const int pos = q * 64 + lane;                   // linear chunk position in the stage
    const int row = pos / Cf::CPR, cp = pos % Cf::CPR;
    const int c = cp ^ (row & 15);                   // source chunk (the XOR stays inside a group of 16 chunks)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(stage + q * 256), 16,
                                             (row * (int)p.lda + 4 * c) * 4, 0, 0, 0);
  }
}

template <int KB, int NB, bool W_TRANS, bool POOL = false>  // K = 128 KB reduction columns, N = 128 NB output columns; KB * NB <= 3
__global__ __launch_bounds__(512, 2) void gemm_ws16_kernel(const Ws16Args p) {
  using Cf = Ws16Cfg<KB, NB>;
  constexpr int K = Cf::K, TILES = Cf::TILES, PPW = Cf::PPW;
  static_assert(KB * NB <= 3, "the stationary weights take 32 * KB * NB registers");
  // NAMED stage buffers and a ring loop unrolled by three: hipcc cannot tell the stage being read from the stage an
  // LDS-DMA is in flight to inside one array, and would wait vmcnt(0) before every LDS read
  __shared__ __attribute__((aligned(16))) float st0[Cf::STAGE];
  __shared__ __attribute__((aligned(16))) float st1[Cf::STAGE];
  __shared__ __attribute__((aligned(16))) float st2[Cf::STAGE];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tl = lane & 15, q4 = lane >> 4;  // MFMA lane roles: row-in-tile / k-quarter
  const int n0 = 16 * NB * wave;             // this wave's 16 NB output columns

  // stationary weights: wr[j][g][c] = W[n0 + 16 j + tl][16 g + 4 q4 + c]  (NT: a row of W; NN: a column)
  float wr[NB][K / 16][4];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int g = 0; g < K / 16; ++g) {
      const int n = n0 + 16 * j + tl;
      if constexpr (W_TRANS) {
#pragma unroll
        for (int c = 0; c < 4; ++c) wr[j][g][c] = p.W[(int64_t)(16 * g + 4 * q4 + c) * p.ldw + n];
      } else {
        const float4 v = *reinterpret_cast<const float4*>(p.W + (int64_t)n * p.ldw + 16 * g + 4 * q4);
        wr[j][g][0] = v.x; wr[j][g][1] = v.y; wr[j][g][2] = v.z; wr[j][g][3] = v.w;
      }
    }
  // a result tile: lane (tl, q4) holds row tl, columns n0 + 16 j + 4 q4 .. + 3
  f32x4 bias4[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4[j] = *reinterpret_cast<const f32x4*>(p.bias + n0 + 16 * j + 4 * q4);
  }

  const int64_t nst = (p.M + Cf::ROWS - 1) / Cf::ROWS;
  const int64_t t0 = nst * blockIdx.x / gridDim.x, t1 = nst * (blockIdx.x + 1) / gridDim.x;

  // POOL: the row-group term of a stage's rows is fetched BEFORE the next stage's DMA is issued (see WS16_STEP): the
  // compiler's wait for it then allows the DMA pieces issued after it to stay in flight
  f32x4 pool4[POOL ? TILES : 1][NB];
  auto fetch_pool = [&](int64_t t) {
    if constexpr (POOL) {
#pragma unroll
      for (int i = 0; i < TILES; ++i) {
        int64_t m = t * Cf::ROWS + tl + 16 * i;
        if (m >= p.M) m = p.M - 1;
        const uint32_t b = (uint32_t)m / p.group;  // (M < 2^32, host-checked)
#pragma unroll
        for (int j = 0; j < NB; ++j) pool4[i][j] = *reinterpret_cast<const f32x4*>(p.pool + (int64_t)b * p.ld_pool + n0 + 16 * j + 4 * q4);
      }
    }
  };

  auto compute = [&](const float* stg, int64_t t) {
    f32x4 acc[TILES][NB];
#pragma unroll
    for (int i = 0; i < TILES; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < K / 16; ++g) {
#pragma unroll
      for (int i = 0; i < TILES; ++i) {
        const int row = 16 * i + tl;
        const f32x4 x = *reinterpret_cast<const f32x4*>(stg + row * K + 4 * ((4 * g + q4) ^ (row & 15)));
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][g][0], x[0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][g][1], x[1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][g][2], x[2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][g][3], x[3], acc[i][j], 0, 0, 0);
        }
      }
    }
    const int64_t m0 = t * Cf::ROWS + tl;
#pragma unroll
    for (int i = 0; i < TILES; ++i) {
      const int64_t m = m0 + 16 * i;
      // (only the LAST stage of the matrix can be ragged and issue fewer than TILES * NB stores -- nothing is consumed
      // after it, so the store count the waits below assume holds wherever it matters)
      if (m < p.M) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          f32x4 v = acc[i][j] + bias4[j];
          if constexpr (POOL) v += pool4[i][j] * p.pool_scale;
          *reinterpret_cast<f32x4*>(p.C + m * p.ldc + n0 + 16 * j + 4 * q4) = v;
        }
      }
    }
  };
  // Stage t lives in buffer t % 3.  Before it is consumed: its pieces have landed (own wait + barrier) and every wave is
  // past stage t - 1, whose buffer takes stage t + 2.  In flight at that wait, youngest first: this wave's result
  // stores of stage t - 1 (at most TILES * NB), the DMA of stage t + 1 (PPW) -- vmcnt retires in order, so allowing that
  // many covers the DMA of stage t and never waits for a store.
#define WS16_STEP(CUR, NXT, T)                                                             \
  do {                                                                                     \
    if (first) ws16_wait_vmcnt<PPW>(); /* no result stores issued yet */                   \
    else ws16_wait_vmcnt<PPW + TILES * NB>();                                              \
    first = false;                                                                         \
    __builtin_amdgcn_s_barrier();                                                          \
    asm volatile("" ::: "memory");                                                         \
    fetch_pool(T);                                                                         \
    ws16_issue<KB, NB>(p, NXT, (T) + 2, t1, wave, lane);                                       \
    compute(CUR, (T));                                                                     \
  } while (0)
  if (t0 < t1) {
    bool first = true;
    ws16_issue<KB, NB>(p, st0, t0, t1, wave, lane);
    ws16_issue<KB, NB>(p, st1, t0 + 1, t1, wave, lane);
    for (int64_t t = t0; t < t1; t += 3) {
      WS16_STEP(st0, st2, t);
      if (t + 1 < t1) WS16_STEP(st1, st0, t + 1);
      if (t + 2 < t1) WS16_STEP(st2, st1, t + 2);
    }
    ws16_wait_vmcnt<0>();
  }
#undef WS16_STEP
}

// dx[m][:] = A[m][:] W + scale * pool[m / group][:] for the history encoder's first layer (K = 384, N = 128, NN): the mean
// pool's backward (ref:src/user_history_encoder.py:115-116, autograd of the mean over H) rides in the dx product's
// epilogue instead of a read-modify-write pass over dx.  -100 = shape not taken.
int gemm_ws16_pool_try(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc,
                       const float* pool, int64_t ld_pool, int64_t group, float scale, hipStream_t st) {
  if (M < 16384 || M >= ((int64_t)1 << 32) || N != 128 || K != 384 || group <= 0 || group >= ((int64_t)1 << 31)) return -100;
  const uintptr_t al = reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(C) |
                       reinterpret_cast<uintptr_t>(pool);
  if ((al & 15) || lda % 4 || ldw % 4 || ldc % 4 || ld_pool % 4 || lda > (1 << 20)) return -100;
  Ws16Args a{A, W, nullptr, C, M, lda, ldw, ldc, pool, ld_pool, (uint32_t)group, scale};
  const int64_t nst = ceil_div(a.M, Ws16Cfg<3, 1>::ROWS);
  const unsigned grid = (unsigned)(nst < 256 ? nst : 256);
  ProfScope prof("gemm_ws16_kernel", st);
  gemm_ws16_kernel<3, 1, true, true><<<grid, 512, 0, st>>>(a);
  return check_launch("gemm_ws16_kernel");
}

template <int KB, int NB, bool WT>
static int ws16_launch(const Ws16Args& a, hipStream_t st) {
  const int64_t nst = ceil_div(a.M, Ws16Cfg<KB, NB>::ROWS);
  const unsigned grid = (unsigned)(nst < 256 ? nst : 256);
  ProfScope prof("gemm_ws16_kernel", st);
  gemm_ws16_kernel<KB, NB, WT><<<grid, 512, 0, st>>>(a);
  return check_launch("gemm_ws16_kernel");
}

// Called by tt_gemm_f32 (gemm.hip) ahead of gemm_ws_try.  -100 = shape not for this kernel.
int gemm_ws16_try(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw,
                  float* C, int64_t ldc, const float* bias, int epilogue, int accumulate, hipStream_t st) {
  if (layout == TT_GEMM_TN || M < 16384 || N % 128 || K % 128 || N < 128 || K < 128 || (N / 128) * (K / 128) > 3) return -100;
  if (epilogue != TT_EPI_NONE || accumulate) return -100;
  const uintptr_t al = reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(C) |
                       reinterpret_cast<uintptr_t>(bias);
  if ((al & 15) || lda % 4 || ldw % 4 || ldc % 4 || lda > (1 << 20)) return -100;  // 32-bit byte offsets within a stage
  Ws16Args a{A, W, bias, C, M, lda, ldw, ldc, nullptr, 0, 1u, 0.f};
  const bool wt = layout == TT_GEMM_NN;
  const int kb = (int)(K / 128), nb = (int)(N / 128);
#define WS16_CASE(KBv, NBv) \
  if (kb == KBv && nb == NBv) return wt ? ws16_launch<KBv, NBv, true>(a, st) : ws16_launch<KBv, NBv, false>(a, st);
  WS16_CASE(1, 1) WS16_CASE(2, 1) WS16_CASE(3, 1) WS16_CASE(1, 2) WS16_CASE(1, 3)
#undef WS16_CASE
  return -100;
}

}  // namespace tt
