// K3: fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32 -- exact fp32,
// an fmaf chain per output element) for the tower MLPs / Linear layers, the
// attention projections and all of their backward passes.
//
// Tiling: workgroup = 4 wavefronts (2x2), tile BMxBNx32, each wave owns a
// (BM/2)x(BN/2) block of 32x32 MFMA tiles.  Operands are staged global ->
// registers -> LDS with one barrier per K-tile (double-buffered LDS): the loads
// of tile t+1 are in flight while tile t is on the MFMA pipe.
//
// LDS images (no transposes anywhere):
//   k-contiguous operand ([rows][K] in memory): [rows][32+4] floats.  A lane
//     fetches FOUR consecutive k with one ds_read_b128; the k -> (MFMA step,
//     lane half) assignment is permuted so that those four values feed four
//     consecutive MFMAs: within each group of 8 k, half h of the wave takes
//     k = 4h..4h+3.  Both operands use the same assignment, so the sum is just
//     evaluated in a fixed, different k order.  Row stride 36 floats = 9 16-B
//     slots (odd) makes the b128 reads bank-conflict free.
//   row-contiguous operand ([K][rows] in memory): [32][rows] floats, lanes read
//     consecutive rows with ds_read_b32 (conflict free by construction).
#include <stdlib.h>

#include "common.hpp"

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDK = BK + 4;  // row stride of a k-contiguous LDS image

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* aux;
  int64_t M, N, K, lda, ldb, ldc, ldaux;
  int64_t k_per_split;  // multiple of BK
  int epilogue, accumulate, a_vec, b_vec, splits;
  float* a_colsum;      // TN only, optional: [splits][M] partial column sums of A (sum over k)
};

// ---- global -> register staging -------------------------------------------
// k-contiguous operand: element (row, k) at X[row*ld + k]
// FULL: the tile lies inside the operand and rows are 16-B aligned -> plain vector loads.  The
// general form guards every element, and hipcc then waits for each load before the next guard
// (one fully exposed memory round trip per load), so interior tiles must not take it.
template <int ROWS, bool FULL>
__device__ __forceinline__ void load_kc(float4 (&st)[ROWS * 8 / 256], const float* __restrict__ X,
                                        int64_t ld, int64_t row0, int64_t nrows, int64_t k0,
                                        int64_t kend, bool vec) {
#pragma unroll
  for (int i = 0; i < ROWS * 8 / 256; ++i) {
    const int f = threadIdx.x + 256 * i;
    const int64_t row = row0 + f / 8, k = k0 + 4 * (f % 8);
    if constexpr (FULL) {
      const float* p = X + row * ld + k;
      st[i] = make_float4(p[0], p[1], p[2], p[3]);  // one global_load_dwordx4 (16-B aligned by contract)
    } else {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < nrows) {
        const float* p = X + row * ld + k;
        if (vec && k + 3 < kend) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (k + 0 < kend) v.x = p[0];
          if (k + 1 < kend) v.y = p[1];
          if (k + 2 < kend) v.z = p[2];
          if (k + 3 < kend) v.w = p[3];
        }
      }
      st[i] = v;
    }
  }
}
template <int ROWS>
__device__ __forceinline__ void store_kc(const float4 (&st)[ROWS * 8 / 256], float* Xs) {
#pragma unroll
  for (int i = 0; i < ROWS * 8 / 256; ++i) {
    const int f = threadIdx.x + 256 * i;
    *reinterpret_cast<float4*>(Xs + (f / 8) * LDK + 4 * (f % 8)) = st[i];
  }
}
// row-contiguous operand: element (row, k) at X[k*ld + row]
template <int ROWS, bool FULL>
__device__ __forceinline__ void load_rc(float4 (&st)[ROWS * 8 / 256], const float* __restrict__ X,
                                        int64_t ld, int64_t row0, int64_t nrows, int64_t k0,
                                        int64_t kend, bool vec) {
#pragma unroll
  for (int i = 0; i < ROWS * 8 / 256; ++i) {
    const int f = threadIdx.x + 256 * i;
    const int64_t k = k0 + f / (ROWS / 4), row = row0 + 4 * (f % (ROWS / 4));
    if constexpr (FULL) {
      const float* p = X + k * ld + row;
      st[i] = make_float4(p[0], p[1], p[2], p[3]);
    } else {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < kend) {
        const float* p = X + k * ld + row;
        if (vec && row + 3 < nrows) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (row + 0 < nrows) v.x = p[0];
          if (row + 1 < nrows) v.y = p[1];
          if (row + 2 < nrows) v.z = p[2];
          if (row + 3 < nrows) v.w = p[3];
        }
      }
      st[i] = v;
    }
  }
}
template <int ROWS>
__device__ __forceinline__ void store_rc(const float4 (&st)[ROWS * 8 / 256], float* Xs) {
#pragma unroll
  for (int i = 0; i < ROWS * 8 / 256; ++i) {
    const int f = threadIdx.x + 256 * i;
    *reinterpret_cast<float4*>(Xs + (f / (ROWS / 4)) * ROWS + 4 * (f % (ROWS / 4))) = st[i];
  }
}

template <int BM, int BN, bool A_KC, bool B_KC, bool FULL>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs g) {
  constexpr int TM = BM / 64, TN = BN / 64;  // 32x32 MFMA tiles per wave along m / n
  constexpr int A_FLOATS = A_KC ? BM * LDK : BK * BM;
  constexpr int B_FLOATS = B_KC ? BN * LDK : BK * BN;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;  // stage s: A at smem + s*STAGE, B right after it

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
  const int64_t kend = (kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 sa[BM * 8 / 256], sb[BN * 8 / 256];
#define TT_FETCH(k0)                                                                              \
  do {                                                                                            \
    if constexpr (A_KC) load_kc<BM, FULL>(sa, g.A, g.lda, m0, g.M, (k0), kend, g.a_vec);          \
    else load_rc<BM, FULL>(sa, g.A, g.lda, m0, g.M, (k0), kend, g.a_vec);                         \
    if constexpr (B_KC) load_kc<BN, FULL>(sb, g.B, g.ldb, n0, g.N, (k0), kend, g.b_vec);          \
    else load_rc<BN, FULL>(sb, g.B, g.ldb, n0, g.N, (k0), kend, g.b_vec);                         \
  } while (0)
#define TT_COMMIT(buf)                                                                            \
  do {                                                                                            \
    float* a_d = smem + (buf) * STAGE_FLOATS;                                                     \
    float* b_d = a_d + A_FLOATS;                                                                  \
    if constexpr (A_KC) store_kc<BM>(sa, a_d); else store_rc<BM>(sa, a_d);                        \
    if constexpr (B_KC) store_kc<BN>(sb, b_d); else store_rc<BN>(sb, b_d);                        \
  } while (0)

  // TN with a_colsum: the workgroups of the first column block also sum the A tile over k
  // (A = dY in a weight-gradient GEMM, so this is the bias gradient -- dY is read once)
  const bool do_colsum = !A_KC && g.a_colsum != nullptr && blockIdx.x == 0;
  float csum = 0.f;

  const int64_t ntiles = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
  if (ntiles > 0) {
    TT_FETCH(kbeg);
    TT_COMMIT(0);
  }
  __syncthreads();
  for (int64_t t = 0; t < ntiles; ++t) {
    const int cur = (int)(t & 1);
    if (t + 1 < ntiles) TT_FETCH(kbeg + (t + 1) * BK);
    const float* a_s = smem + cur * STAGE_FLOATS;
    const float* b_s = a_s + A_FLOATS;
    if constexpr (!A_KC) {
      if (do_colsum) {  // thread -> column t % BM, k rows t / BM, + 256/BM, ... (zero padded)
#pragma unroll
        for (int q = 0; q < BK * BM / 256; ++q) csum += a_s[(q * (256 / BM) + threadIdx.x / BM) * BM + threadIdx.x % BM];
      }
    }
#pragma unroll
    for (int grp = 0; grp < BK / 8; ++grp) {
      float av[TM][4], bv[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (A_KC) {
          const float4 v = *reinterpret_cast<const float4*>(a_s + (wm + 32 * i + r) * LDK + 8 * grp + 4 * h);
          av[i][0] = v.x; av[i][1] = v.y; av[i][2] = v.z; av[i][3] = v.w;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) av[i][c] = a_s[(8 * grp + 4 * h + c) * BM + wm + 32 * i + r];
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (B_KC) {
          const float4 v = *reinterpret_cast<const float4*>(b_s + (wn + 32 * j + r) * LDK + 8 * grp + 4 * h);
          bv[j][0] = v.x; bv[j][1] = v.y; bv[j][2] = v.z; bv[j][3] = v.w;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) bv[j][c] = b_s[(8 * grp + 4 * h + c) * BN + wn + 32 * j + r];
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c], bv[j][c], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < ntiles) TT_COMMIT(cur ^ 1);
    __syncthreads();
  }
#undef TT_FETCH
#undef TT_COMMIT

  if constexpr (!A_KC) {
    if (do_colsum) {  // combine the 256/BM k-groups in a fixed order (the LDS tiles are dead now)
      smem[threadIdx.x] = csum;
      __syncthreads();
      if (threadIdx.x < BM && m0 + threadIdx.x < g.M) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 256 / BM; ++q) v += smem[q * BM + threadIdx.x];
        g.a_colsum[(int64_t)blockIdx.z * g.M + m0 + threadIdx.x] = v;
      }
    }
  }

  // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const bool partial = g.splits > 1;
  float* out = partial ? g.C + (int64_t)blockIdx.z * g.M * g.N : g.C;
  const int64_t ldo = partial ? g.N : g.ldc;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int64_t col = n0 + wn + 32 * j + r;
      if (col >= g.N) continue;
      const float bcol = (!partial && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row >= g.M) continue;
        float v = acc[i][j][e];
        if (!partial) {
          v += bcol;
          if (g.epilogue == TT_EPI_RELU) v = fmaxf(v, 0.f);
          else if (g.epilogue == TT_EPI_RELU_MASK) v = (g.aux[row * g.ldaux + col] > 0.f) ? v : 0.f;
          if (g.accumulate) v += out[row * ldo + col];
        }
        out[row * ldo + col] = v;
      }
    }
}

// sum the split-K slabs in slab order (deterministic) and apply the epilogue.  Workgroups past `nb_reduce`
// (tt_gemm_tn_colsum_f32 only) turn the per-split column-sum partials into the column sums -- what was a
// launch of its own (colsum_stage2, same arithmetic in the same order: 13 launches per step at C3).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, GemmArgs g, int nb_reduce,
                                                            const float* __restrict__ cs_part, float* __restrict__ cs_out) {
  if ((int)blockIdx.x >= nb_reduce) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int64_t col = (int64_t)(blockIdx.x - nb_reduce) * 64 + c;
    float sum = 0.f;
    if (col < g.M)
      for (int64_t p = rg; p < g.splits; p += 4) sum += cs_part[p * g.M + col];
    red[rg][c] = sum;
    __syncthreads();
    if (rg == 0 && col < g.M) cs_out[col] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
    return;
  }
  const int64_t total = g.M * g.N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)nb_reduce * blockDim.x) {
    const int64_t row = i / g.N, col = i % g.N;
    float v = 0.f;
    for (int s = 0; s < g.splits; ++s) v += ws[(int64_t)s * total + i];
    if (g.bias) v += g.bias[col];
    if (g.epilogue == TT_EPI_RELU) v = fmaxf(v, 0.f);
    else if (g.epilogue == TT_EPI_RELU_MASK) v = (g.aux[row * g.ldaux + col] > 0.f) ? v : 0.f;
    if (g.accumulate) v += g.C[row * g.ldc + col];
    g.C[row * g.ldc + col] = v;
  }
}

// ---- column sums -----------------------------------------------------------
// stage 1: workgroup = 64 columns x 4 row groups over a 64-row chunk (each wave reads one
// 256-B row segment per instruction); the 4 row-group partials are combined in fixed order.
static inline int64_t colsum_rows_per_part(int64_t M) {  // 64 rows, more for tall inputs (<= 512 parts)
  int64_t r = 64;
  while (ceil_div(M, r) > 512) r *= 2;
  return r;
}
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ X, int64_t M, int64_t N,
                                                     int64_t ldx, float* __restrict__ part, int64_t CS_ROWS) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + c;
  const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS;
  const int64_t r1 = (r0 + CS_ROWS < M) ? r0 + CS_ROWS : M;
  float s = 0.f;
  if (col < N)
    for (int64_t rr = r0 + rg; rr < r1; rr += 4) s += X[rr * ldx + col];
  red[rg][c] = s;
  __syncthreads();
  if (rg == 0 && col < N) part[(int64_t)blockIdx.y * N + col] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
}
__global__ __launch_bounds__(256) void colsum_stage2(const float* __restrict__ part, int64_t nparts,
                                                     int64_t N, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + c;
  float s = 0.f;
  if (col < N)
    for (int64_t p = rg; p < nparts; p += 4) s += part[p * N + col];
  red[rg][c] = s;
  __syncthreads();
  if (rg == 0 && col < N) out[col] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
}

struct GemmPlan {
  int bm, bn, splits;
  int64_t k_per_split;
};

static GemmPlan plan_gemm(int64_t M, int64_t N, int64_t K) {
  GemmPlan p;
  const int64_t t128 = ceil_div(M, 128) * ceil_div(N, 128);
  // 128^2 tiles when there are enough of them, or when a huge reduction (split-K) supplies the
  // parallelism instead: the bigger tile halves the operand bytes moved per MFMA
  if (t128 >= 192 || (K >= 8192 && M >= 128 && N >= 128 && t128 >= 2)) { p.bm = 128; p.bn = 128; } else { p.bm = 64; p.bn = 64; }
  const int64_t tiles = ceil_div(M, p.bm) * ceil_div(N, p.bn);
  const int64_t ktiles = ceil_div(K, BK);
  int64_t splits = 1;
  if (tiles < 256 && ktiles >= 16) {
    // One workgroup per CU (256) measured as good as or better than two (512) for the split-K weight
    // gradients and halves the slabs the reduce kernel has to read; what must NOT happen is
    // tiles * splits landing just above a multiple of the resident slots (3 tiles x 171 splits = 513
    // cost 2x: one straggler workgroup ran a second round on an otherwise idle GPU).
    splits = 256 / tiles;
    if (splits < 1) splits = 1;
    if (splits > ktiles / 4) splits = ktiles / 4;  // at least 4 K-tiles per split
    if (splits > 256) splits = 256;
    if (splits < 1) splits = 1;
  }
  p.k_per_split = ceil_div(ktiles, splits) * BK;
  p.splits = (int)ceil_div(K, p.k_per_split);
  return p;
}

template <int BM, int BN, bool A_KC, bool B_KC, bool FULL>
static int launch_gemm_v(const GemmArgs& g, hipStream_t st) {
  constexpr int A_FLOATS = A_KC ? BM * LDK : BK * BM;
  constexpr int B_FLOATS = B_KC ? BN * LDK : BK * BN;
  const size_t lds = 2 * (A_FLOATS + B_FLOATS) * sizeof(float);
  dim3 grid((unsigned)ceil_div(g.N, BN), (unsigned)ceil_div(g.M, BM), (unsigned)g.splits);
  static bool lds_opt_in = false;  // >64 KiB of dynamic LDS needs an explicit opt-in, once per kernel
  if (!lds_opt_in && lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<BM, BN, A_KC, B_KC, FULL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("gemm_kernel: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    lds_opt_in = true;
  }
  gemm_kernel<BM, BN, A_KC, B_KC, FULL><<<grid, 256, lds, st>>>(g);
  return check_launch("gemm_kernel");
}
template <int BM, int BN, bool A_KC, bool B_KC>
static int launch_gemm(const GemmArgs& g, hipStream_t st) {
  // every tile interior, every row 16-B aligned, every split a whole number of K-tiles
  const bool full = g.M % BM == 0 && g.N % BN == 0 && g.K % BK == 0 && g.a_vec && g.b_vec;
  return full ? launch_gemm_v<BM, BN, A_KC, B_KC, true>(g, st) : launch_gemm_v<BM, BN, A_KC, B_KC, false>(g, st);
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// tall-M weights-stationary fast path (gemm_ws.hip); -100 = "not applicable"
int gemm_ws_try(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W,
                int64_t ldw, float* C, int64_t ldc, const float* bias, int epilogue, const float* aux,
                int64_t ldaux, int accumulate, hipStream_t st);
// tall-M products with exactly 128 output columns, K in {128, 256, 384} (gemm_ws16.hip); -100 = "not applicable"
int gemm_ws16_try(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw,
                  float* C, int64_t ldc, const float* bias, int epilogue, int accumulate, hipStream_t st);
// huge-K TN products with a register-resident result (gemm_tn_stream.hip); -100 = "not applicable"
int64_t gemm_tn_stream_ws_floats(int64_t M, int64_t N, int64_t K);
int gemm_tn_stream_try(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                       float* C, int64_t ldc, int accumulate, float* a_colsum, void* ws, int64_t ws_bytes, hipStream_t st);

}  // namespace tt

using namespace tt;

extern "C" int64_t tt_gemm_workspace_bytes(int layout, int64_t M, int64_t N, int64_t K) {
  (void)layout;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const GemmPlan p = plan_gemm(M, N, K);
  // split-K slabs + (tt_gemm_tn_colsum_f32) one row of column-sum partials per split
  const int64_t generic = p.splits > 1 ? round_up((int64_t)p.splits * M * (N + 1) * (int64_t)sizeof(float), 256) : 0;
  const int64_t stream = layout == TT_GEMM_TN ? round_up(gemm_tn_stream_ws_floats(M, N, K) * (int64_t)sizeof(float), 256) : 0;
  return generic > stream ? generic : stream;
}

static int gemm_impl(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                     const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                     int epilogue, const float* aux, int64_t ldaux, int accumulate, float* a_colsum, void* ws,
                     int64_t ws_bytes, tt_stream_t stream) {
  if (!A || !B || !C) return fail_arg("tt_gemm_f32: null pointer");
  if (M < 0 || N < 0 || K < 0 || ldc < N) return fail_arg("tt_gemm_f32: sizes");
  if (layout < TT_GEMM_NT || layout > TT_GEMM_TN) return fail_arg("tt_gemm_f32: layout");
  if (epilogue == TT_EPI_RELU_MASK && (!aux || ldaux < N)) return fail_arg("tt_gemm_f32: relu mask needs aux");
  if (M == 0 || N == 0) return 0;
  const bool a_kc = layout != TT_GEMM_TN, b_kc = layout == TT_GEMM_NT;
  if (lda < (a_kc ? K : M) || ldb < (b_kc ? K : N)) return fail_arg("tt_gemm_f32: leading dimension");

  if (layout == TT_GEMM_TN && !bias && epilogue == TT_EPI_NONE) {
    const int rc_st = gemm_tn_stream_try(M, N, K, A, lda, B, ldb, C, ldc, accumulate, a_colsum, ws, ws_bytes, S(stream));
    if (rc_st != -100) return rc_st;
  }
  if (!a_colsum) {
    const int rc_16 = gemm_ws16_try(layout, M, N, K, A, lda, B, ldb, C, ldc, bias, epilogue, accumulate, S(stream));
    if (rc_16 != -100) return rc_16;
    const int rc_ws = gemm_ws_try(layout, M, N, K, A, lda, B, ldb, C, ldc, bias, epilogue, aux, ldaux, accumulate, S(stream));
    if (rc_ws != -100) return rc_ws;
  }
  GemmArgs g;
  g.a_colsum = nullptr;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.aux = aux;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux;
  g.epilogue = epilogue; g.accumulate = accumulate;
  g.a_vec = (lda % 4 == 0) && al16(A);
  g.b_vec = (ldb % 4 == 0) && al16(B);
  const GemmPlan p = plan_gemm(M, N, K > 0 ? K : 1);
  g.splits = (K > 0) ? p.splits : 1;
  g.k_per_split = p.k_per_split;
  hipStream_t st = S(stream);

  GemmArgs gk = g;
  float* cs_part = nullptr;
  if (g.splits > 1) {
    const int64_t need = round_up((int64_t)g.splits * M * (N + 1) * (int64_t)sizeof(float), 256);
    if (!ws || ws_bytes < need) { set_error("tt_gemm_f32: workspace %lld < %lld", (long long)ws_bytes, (long long)need); return TT_E_WORKSPACE; }
    gk.C = reinterpret_cast<float*>(ws);
    cs_part = reinterpret_cast<float*>(ws) + (int64_t)g.splits * M * N;
  }
  if (a_colsum) gk.a_colsum = g.splits > 1 ? cs_part : a_colsum;
  int rc;
#define TT_DISPATCH(BMv, BNv)                                                        \
  (layout == TT_GEMM_NT   ? launch_gemm<BMv, BNv, true, true>(gk, st)                \
   : layout == TT_GEMM_NN ? launch_gemm<BMv, BNv, true, false>(gk, st)               \
                          : launch_gemm<BMv, BNv, false, false>(gk, st))
  rc = (p.bm == 128) ? TT_DISPATCH(128, 128) : TT_DISPATCH(64, 64);
#undef TT_DISPATCH
  if (rc) return rc;
  if (g.splits > 1) {
    const int64_t total = M * N;
    const int64_t blocks = ceil_div(total, 256) < 2048 ? ceil_div(total, 256) : 2048;
    // with a_colsum: ceil(M / 64) extra workgroups reduce the per-split column-sum partials, in split order
    const int64_t extra = a_colsum ? ceil_div(M, 64) : 0;
    splitk_reduce_kernel<<<(unsigned)(blocks + extra), 256, 0, st>>>(reinterpret_cast<const float*>(ws), g, (int)blocks, cs_part, a_colsum);
    if ((rc = check_launch("splitk_reduce_kernel"))) return rc;
  }
  return 0;
}

extern "C" int tt_gemm_f32(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                           const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                           int epilogue, const float* aux, int64_t ldaux, int accumulate, void* ws,
                           int64_t ws_bytes, tt_stream_t stream) {
  return gemm_impl(layout, M, N, K, A, lda, B, ldb, C, ldc, bias, epilogue, aux, ldaux, accumulate, nullptr, ws,
                   ws_bytes, stream);
}

extern "C" int tt_gemm_tn_colsum_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                                     int64_t ldb, float* C, int64_t ldc, int accumulate, float* a_colsum, void* ws,
                                     int64_t ws_bytes, tt_stream_t stream) {
  if (!a_colsum) return fail_arg("tt_gemm_tn_colsum_f32: null pointer");
  if (K <= 0) return fail_arg("tt_gemm_tn_colsum_f32: sizes");
  return gemm_impl(TT_GEMM_TN, M, N, K, A, lda, B, ldb, C, ldc, nullptr, TT_EPI_NONE, nullptr, 0, accumulate,
                   a_colsum, ws, ws_bytes, stream);
}

extern "C" int64_t tt_colsum_workspace_bytes(int64_t M, int64_t N) {
  if (M <= 0 || N <= 0) return 0;
  return round_up(ceil_div(M, colsum_rows_per_part(M)) * N * (int64_t)sizeof(float), 256);
}

extern "C" int tt_colsum_f32(const float* X, int64_t M, int64_t N, int64_t ldx, float* out, void* ws,
                             int64_t ws_bytes, tt_stream_t stream) {
  if (!X || !out) return fail_arg("tt_colsum_f32: null pointer");
  if (M <= 0 || N <= 0 || ldx < N) return fail_arg("tt_colsum_f32: sizes");
  if (!ws || ws_bytes < tt_colsum_workspace_bytes(M, N)) { set_error("tt_colsum_f32: workspace"); return TT_E_WORKSPACE; }
  const int64_t rpp = colsum_rows_per_part(M);
  const int64_t nparts = ceil_div(M, rpp);
  float* part = reinterpret_cast<float*>(ws);
  colsum_stage1<<<dim3((unsigned)ceil_div(N, 64), (unsigned)nparts), 256, 0, S(stream)>>>(X, M, N, ldx, part, rpp);
  int rc = check_launch("colsum_stage1");
  if (rc) return rc;
  colsum_stage2<<<(unsigned)ceil_div(N, 64), 256, 0, S(stream)>>>(part, nparts, N, out);
  return check_launch("colsum_stage2");
}
