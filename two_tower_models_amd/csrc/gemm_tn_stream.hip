// Weight-gradient products with a huge reduction length and a small result -- the encoder's
//     dW_in  [3D, D] = dQKV^T x      dW_out [D, D] = dY^T ctx      (+ the bias gradients = column sums)
// over K = B*H = 204 800 token rows (ref:src/user_history_encoder.py:104-108: the backward of
// nn.MultiheadAttention's two projections) -- as ONE pass in which every workgroup keeps the WHOLE result
// in registers:
//
//   C[m][n] = sum_k A[k][m] * B[k][n]        A = [K, M] (lda), B = [K, N = 128] (ldb),  M in {128, 256, 384}
//
// Both operands of a v_mfma_f32_32x32x2_f32 step are ROWS of A and B (two consecutive k: lane half h
// takes row k0 + h), so the row-major operands need no transpose: a wave owns 128 columns of A and 32 of
// B, reads its four consecutive A columns with ONE ds_read_b128 (they belong to four DIFFERENT result
// tiles -- tile i holds the columns = i mod 4 -- so no value ever moves between lanes; the reduce kernel
// undoes the permutation) and one B value per k-step, and issues four MFMAs.  M/128 x 4 waves per
// workgroup, one workgroup per CU (two at M = 128), each over a contiguous range of k; the rows stream
// through a 3-stage LDS ring filled by LDS-DMA (buffer_load_dwordx4 ... lds; rows past K are outside the
// descriptor and land as zeros), one barrier per stage of 16-32 rows.  Per-workgroup partial results go to
// the workspace and a second kernel adds them in workgroup order (deterministic).
//
// Why LDS and not "global -> registers -> MFMA" (the first version, tools/tns_probe.hip): a CU returns
// ~14 B/clk of global loads to registers whatever their width, i.e. 224 B per fp32 MFMA slot, and a wave
// with 4 tiles needs 320 B per MFMA (each A value is wanted by 4 waves, each B value by M/128): 95 TF/s
// on L1-resident data, 217 us for dW_in.  The ring brings every byte to the CU once (5 B/clk) and LDS
// reads at 0.5 per MFMA cost nothing (155 TF/s in the probe).
#include <stdlib.h>

#include "common.hpp"

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct TnsArgs {
  const float* A;
  const float* B;
  float* part;     // [nwg][4 CG waves][4][16][64]
  float* cs_part;  // [nwg][CG][4][32]
  int64_t K, lda, ldb;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CG, int S>
struct TnsCfg {
  static constexpr int M = 128 * CG, ROWS = 2 * S;            // rows (= 2 per k-step) of one ring stage
  static constexpr int A_FL = ROWS * M, B_FL = ROWS * 128;    // floats per stage
  static constexpr int A_P = A_FL / 256, B_P = B_FL / 256;    // 1-KiB DMA pieces (one wave instruction each)
  static constexpr int WAVES = 4 * CG, PPW = (A_P + B_P) / WAVES;
  static_assert((A_P + B_P) % WAVES == 0 && A_FL % 256 == 0 && B_FL % 256 == 0, "pieces must divide evenly");
};
constexpr int tns_s(int CG) { return CG == 3 ? 12 : CG == 2 ? 16 : 8; }   // k-steps per stage (48 / 48 / 16 KiB)
constexpr int tns_wps(int CG) { return CG == 2 ? 2 : 3; }                  // waves per SIMD (CG = 1: three workgroups per CU)

template <int N>
__device__ __forceinline__ void tns_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LDS-DMA of one ring stage: this wave's PPW pieces.  (A free function, not a lambda of the kernel: with the
// buffer builtins inside a lambda the HOST pass silently drops the kernel's launch stub.)
template <int PPW>
__device__ __forceinline__ void tns_issue(const TnsArgs& p, float* As, float* Bs, int64_t t, int64_t t1, int rows_per_stage,
                                          const bool* is_a, const int* voff, const int* loff) {
  const int64_t row0 = t * rows_per_stage;
  int64_t left = (t < t1 ? p.K - row0 : 0);  // rows of this stage inside the matrix (and inside this workgroup's range)
  if (left > rows_per_stage) left = rows_per_stage;
  if (left < 0) left = 0;
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const float* base = is_a[i] ? p.A + row0 * p.lda : p.B + row0 * p.ldb;
    const int bytes = (int)left * (int)(is_a[i] ? p.lda : p.ldb) * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(left ? base : p.A), 0, bytes, 0x00020000);
    float* dst = (is_a[i] ? As : Bs) + loff[i];
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff[i], 0, 0, 0);
  }
}

template <int CG, int S>
__global__ __launch_bounds__(256 * CG, tns_wps(CG)) void tn_stream_kernel(const TnsArgs p) {
  using Cf = TnsCfg<CG, S>;
  constexpr int M = Cf::M, PPW = Cf::PPW;
  // NAMED stage buffers (and a ring loop unrolled by four): with one LDS block hipcc cannot tell the stage
  // being read from the stage an LDS-DMA is in flight to and waits vmcnt(0) before every LDS read
  __shared__ __attribute__((aligned(16))) float As0[Cf::A_FL], As1[Cf::A_FL], As2[Cf::A_FL];
  __shared__ __attribute__((aligned(16))) float Bs0[Cf::B_FL], Bs1[Cf::B_FL], Bs2[Cf::B_FL];
  // wave index as a scalar: it selects the DMA descriptors, which must be wave-uniform (else: waterfall loops)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
  const int cg = wave >> 2, nt = wave & 3;

  // this workgroup's stages: a contiguous range of the ceil(K / ROWS) stages
  const int64_t nst = (p.K + Cf::ROWS - 1) / Cf::ROWS;
  const int64_t t0 = nst * blockIdx.x / gridDim.x, t1 = nst * (blockIdx.x + 1) / gridDim.x;

  // the wave's DMA pieces: piece q covers floats [256 q, 256 q + 256) of the stage's A image ([ROWS][M]) or,
  // from A_P on, of its B image ([ROWS][128]); the lane fetches 16 bytes of it
  int voff[PPW], loff[PPW];
  bool is_a[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int q = wave * PPW + i;
    is_a[i] = q < Cf::A_P;
    const int qq = is_a[i] ? q : q - Cf::A_P;
    const int f = qq * 256 + lane * 4;
    const int row = is_a[i] ? f / M : f / 128, col = is_a[i] ? f % M : f % 128;
    voff[i] = (row * (int)(is_a[i] ? p.lda : p.ldb) + col) * 4;
    loff[i] = qq * 256;
  }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  // The ring.  Stage t lives in buffer t % 3; operand reads run P k-steps ahead of their MFMAs, ACROSS stage
  // boundaries, so the barrier sits P k-steps before the end of a stage's MFMAs -- where the reads move on to
  // the next buffer -- and no wave ever faces an LDS round trip with an empty matrix pipe.  (First version:
  // barrier, then all of the stage's reads, then its MFMAs -- every wave of the CU waited for 180 KB of LDS
  // reads at once, matrix pipe 75 % busy.)  At that barrier: this wave's pieces of stage t + 1 have landed
  // (vmcnt; stage t + 2 may still be in flight), its reads of stage t have RETURNED (lgkmcnt(0): the youngest is
  // a k-step old), and after it every wave's have -- so stage t + 3 can be sent into stage t's buffer.  The
  // barrier is the bare instruction: __syncthreads() carries a release fence, which hipcc lowers to vmcnt(0).
  // (Macros, not lambdas: see tns_issue.)
  constexpr int P = 2;
  f32x4 ra[S + P];
  float rb[S + P];
  const int aoff = h * M + cg * 128 + 4 * r, boff = h * 128 + 32 * nt + r;
#define TNS_ISSUE(AN, BN, T) tns_issue<PPW>(p, AN, BN, (T), t1, Cf::ROWS, is_a, voff, loff)
#define TNS_READ(AB, BB, KS, SLOT)                                                        \
  do {                                                                                    \
    ra[SLOT] = *reinterpret_cast<const f32x4*>((AB) + aoff + 2 * (KS) * M);               \
    rb[SLOT] = (BB)[boff + 2 * (KS) * 128];                                               \
  } while (0)
#define TNS_STAGE(AC, BC, AN, BN, T)                                                      \
  do {                                                                                    \
    _Pragma("unroll") for (int s = 0; s < S; ++s) {                                       \
      if (s == S - P) {                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                \
        tns_wait_vmcnt<PPW>();                                                            \
        __builtin_amdgcn_s_barrier();                                                     \
        asm volatile("" ::: "memory");                                                    \
        TNS_ISSUE(AC, BC, (T) + 3);                                                       \
      }                                                                                   \
      if (s + P < S) TNS_READ(AC, BC, s + P, s + P);                                      \
      else TNS_READ(AN, BN, s + P - S, s + P);                                            \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                     \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s][i], rb[s], acc[i], 0, 0, 0);  \
        cs[i] += ra[s][i];                                                                \
      }                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                  \
    }                                                                                     \
    _Pragma("unroll") for (int j = 0; j < P; ++j) { ra[j] = ra[S + j]; rb[j] = rb[S + j]; } \
  } while (0)
  if (t0 < t1) {
    TNS_ISSUE(As0, Bs0, t0);
    TNS_ISSUE(As1, Bs1, t0 + 1);
    TNS_ISSUE(As2, Bs2, t0 + 2);
    tns_wait_vmcnt<2 * PPW>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < P; ++j) TNS_READ(As0, Bs0, j, j);
    for (int64_t t = t0; t < t1; t += 3) {
      TNS_STAGE(As0, Bs0, As1, Bs1, t);
      if (t + 1 < t1) TNS_STAGE(As1, Bs1, As2, Bs2, t + 1);
      if (t + 2 < t1) TNS_STAGE(As2, Bs2, As0, Bs0, t + 2);
    }
    tns_wait_vmcnt<0>();  // the (empty) look-ahead transfers
  }
#undef TNS_STAGE
#undef TNS_READ
#undef TNS_ISSUE

  float* out = p.part + ((int64_t)blockIdx.x * (4 * CG) + wave) * (4 * 16 * 64);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) out[(i * 16 + e) * 64 + lane] = acc[i][e];
  if (p.cs_part && nt == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = cs[i] + __shfl_xor(cs[i], 32, 64);
      if (h == 0) p.cs_part[(((int64_t)blockIdx.x * CG + cg) * 4 + i) * 32 + r] = v;
    }
  }
}

// C[m][n] (+)= sum over workgroups, in workgroup order; colsum[m] likewise.  A thread walks the partials
// in THEIR layout (consecutive threads = consecutive lanes of one accumulator register: coalesced) and
// maps its position back to (m, n).  Sixteen threads share an element (workgroups q, q + 16, ... each, combined
// in a fixed order through LDS) and every thread's loads are independent of one another: inside the train step
// this kernel runs next to the table sweep, which saturates HBM -- a chain of dependent adds, one memory
// latency each (first version: 4 threads per element), took 60-260 us there against 10 us on an idle chip.
template <int CG>
__global__ __launch_bounds__(1024) void tn_stream_reduce_kernel(const float* __restrict__ part, const float* __restrict__ cs_part,
                                                                int nwg, float* __restrict__ C, int64_t ldc, int accumulate,
                                                                float* __restrict__ colsum) {
  constexpr int PER_WG = 4 * CG * 4 * 16 * 64;
  __shared__ float sh[16][64];
  const int t = threadIdx.x & 63, q = threadIdx.x >> 6;
  if ((int)blockIdx.x < PER_WG / 64) {
    const int pos = blockIdx.x * 64 + t;  // [wave][i][e][lane]
    const float* src = part + pos;
    float s = 0.f;
    int w = q;
    for (; w + 112 < nwg; w += 128) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(w + 16 * u) * PER_WG];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; w < nwg; w += 16) s += src[(int64_t)w * PER_WG];
    sh[q][t] = s;
    __syncthreads();
    if (q == 0) {
      float tot = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) tot += sh[u][t];
      const int lane = pos & 63, e = (pos >> 6) & 15, i = (pos >> 10) & 3, wave = pos >> 12;
      const int cg = wave >> 2, nt = wave & 3, y = lane & 31, h = lane >> 5;
      const int x = (e & 3) + 8 * (e >> 2) + 4 * h;
      const int m = cg * 128 + 4 * x + i, n = 32 * nt + y;
      float* dst = C + (int64_t)m * ldc + n;
      *dst = accumulate ? *dst + tot : tot;
    }
  } else if (colsum) {
    // 128 * CG column sums: thread (m, slice) adds every 8th workgroup's partial, slices combined in order
    float* shc = &sh[0][0];
    const int m = threadIdx.x >> 1 < 128 * CG ? threadIdx.x >> 1 : 0, sl = threadIdx.x & 1;
    const int cg = m / 128, x = (m % 128) / 4, i = m % 4;
    const float* src = cs_part + (cg * 4 + i) * 32 + x;
    float s = 0.f;
    int w = sl;
    for (; w + 14 < nwg; w += 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(w + 2 * u) * (CG * 128)];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; w < nwg; w += 2) s += src[(int64_t)w * (CG * 128)];
    shc[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && (int)(threadIdx.x >> 1) < 128 * CG) colsum[m] = shc[threadIdx.x] + shc[threadIdx.x + 1];
  }
}

static int tns_nwg(int64_t M, int64_t K) {
  const int64_t chunks = ceil_div(K, 2 * tns_s((int)(M / 128)));
  const int64_t slots = M == 128 ? 768 : 256;  // resident workgroups: 3 x 4 waves per CU at M = 128, else one
  return (int)(chunks < slots ? chunks : slots);
}

// workspace floats for the streaming form (0 = shape not taken).  M = 128 / 256 work and are faster than the
// tiled kernel on an idle chip (72 vs 92 us, 125 vs 155 us at K = 204 800) but lose inside the train step, next
// to the table sweep: their partials (768 x 64 KB, 256 x 128 KB) cost as much to add up as the products
// themselves once HBM is shared.  Taken with TT_GEMM_TN_STREAM_ALL=1 (tests, A/B); M = 384 always.
int64_t gemm_tn_stream_ws_floats(int64_t M, int64_t N, int64_t K) {
  static const bool all = getenv("TT_GEMM_TN_STREAM_ALL") != nullptr;
  if (N != 128 || (M != 384 && !(all && (M == 128 || M == 256))) || K < 8192) return 0;
  const int64_t cg = M / 128;
  return (int64_t)tns_nwg(M, K) * (4 * cg * 4 * 16 * 64 + cg * 4 * 32);
}

template <int CG>
static int tns_launch(const TnsArgs& a, int nwg, float* C, int64_t ldc, int accumulate, float* colsum, hipStream_t st) {
  {
    ProfScope prof("tn_stream_kernel", st);
    tn_stream_kernel<CG, tns_s(CG)><<<nwg, 256 * CG, 0, st>>>(a);
  }
  int rc = check_launch("tn_stream_kernel");
  if (rc) return rc;
  const int blocks = 4 * CG * 4 * 16 + 1;
  tn_stream_reduce_kernel<CG><<<blocks, 1024, 0, st>>>(a.part, a.cs_part, nwg, C, ldc, accumulate, colsum);
  return check_launch("tn_stream_reduce_kernel");
}

// Called by tt_gemm_f32 / tt_gemm_tn_colsum_f32 (gemm.hip) for TN products.  -100 = shape not for this kernel.
int gemm_tn_stream_try(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                       float* C, int64_t ldc, int accumulate, float* a_colsum, void* ws, int64_t ws_bytes, hipStream_t st) {
  const int64_t need = gemm_tn_stream_ws_floats(M, N, K) * (int64_t)sizeof(float);
  if (need == 0) return -100;
  const uintptr_t al = reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B);
  if ((al & 15) || lda % 4 || ldb % 4 || lda < M || ldb < N) return -100;
  if (lda >= ((int64_t)1 << 22) || ldb >= ((int64_t)1 << 22)) return -100;  // 32-bit byte offsets within a stage
  if (!ws || ws_bytes < need) { set_error("tt_gemm_f32 (streaming TN): workspace %lld < %lld", (long long)ws_bytes, (long long)need); return TT_E_WORKSPACE; }
  const int nwg = tns_nwg(M, K);
  const int cg = (int)(M / 128);
  TnsArgs a{};
  a.A = A; a.B = B; a.K = K; a.lda = lda; a.ldb = ldb;
  a.part = reinterpret_cast<float*>(ws);
  a.cs_part = a.part + (int64_t)nwg * (4 * cg * 4 * 16 * 64);
  if (cg == 1) return tns_launch<1>(a, nwg, C, ldc, accumulate, a_colsum, st);
  if (cg == 2) return tns_launch<2>(a, nwg, C, ldc, accumulate, a_colsum, st);
  return tns_launch<3>(a, nwg, C, ldc, accumulate, a_colsum, st);
}

}  // namespace tt
