// K6: brute-force MIPS top-K (torch.topk(q @ corpus.T, K), ref:src/baseline_mips_module.py:57-61)
// whose [B, C] score matrix never reaches HBM, exact under the total order
// (score desc, index asc).
//
// The corpus is cut into GROUPS of 64 consecutive rows (128 through the LDS-DMA pass with more than 64 queries:
// MipsArgs::gshift).  Per batch of <= 1024 queries:
//   pass 1   dense score GEMM; each lane reduces the scores of one group to their max (the LDS-DMA form also to the
//            runner-up and the position of the best) and stores it:  gmax[group][query], gm2[group][query]
//            (4 B each; see gmax_ord for the two formats)                              (MFMA-bound)
//   select   the K best groups of every query under (max score desc, group asc): MSD radix
//            select of the K-th largest 64-bit group key (ord(max) << 32 | ~group) -> tau[q].
//            Groups are contiguous row ranges, so this is the order of their best items, and
//            every item of the true top-K lies in one of these exactly-K groups.
//   list     glist[q][0..K) = the groups with key >= tau[q].
//   pass 2   SPARSE: only the K selected groups of each query are scored again (64 K rows per
//            query instead of C), one wavefront per (query, group), with the same MFMA
//            instruction sequence as pass 1 (bit-identical scores; the query simply occupies
//            all 32 B-operand columns).  Items whose (ord(score), ~group) >= tau[q] become
//            candidates (64-bit keys ord(score) << 32 | ~row): at most group*K, typically ~1.01 K.  Groups whose
//            runner-up cannot reach tau[q] are not re-read whole: their best item is known (fp32: its row; bf16: its
//            4-row quad, of which the 4 rows are scored again).
//            The corpus rows are read straight from HBM (contiguous 16-32 KiB per group).
//            Ragged / unaligned D and corpora with fewer than K groups take the dense pass 2
//            (the pass-1 kernel with the candidate epilogue) instead.
//   sort     one wavefront per query: stable LSD radix sort of the candidates, emit top K.
//
// GEMM structure = the in-batch-softmax kernel's: each wave keeps 32 queries in registers
// as MFMA B-operand fragments, corpus rows stream through double-buffered LDS as the A
// operand, so the C/D layout gives ONE query per lane and 16 corpus rows per tile in that
// lane's registers.  Which corpus row feeds which A-operand row is free, and it is chosen so
// that over a 128-row chunk (4 tiles) lane-half h sees exactly the 64 CONSECUTIVE rows
// 64h .. 64h+63, in increasing order: the group max needs no cross-lane traffic.
//   fp32 : v_mfma_f32_32x32x2_f32   (fp32 products and accumulate)
//   bf16 : v_mfma_f32_32x32x16_bf16 (exact products, fp32 accumulate)
#include <type_traits>

#include "common.hpp"
#include "mfma_stream.hpp"

#ifndef TT_MIPS_EXP
#define TT_MIPS_EXP 0  // measurement variants of the bf16 pass 1 (tools/mips_variants.sh); 0 = the product kernel
#endif

namespace tt {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int QB_WG = 128;   // queries per workgroup (32 per wave)
constexpr int CT = 64;       // corpus rows per LDS tile
constexpr int CHUNK = 128;   // corpus rows per group-pair: lane-half h owns 64 of them
constexpr int GROUP = 64;

__device__ __forceinline__ uint32_t f2ord(float x) {
  const uint32_t u = __float_as_uint(x);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  const uint32_t u = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);
  return __uint_as_float(u);
}
__device__ __forceinline__ u64 make_key(float score, uint32_t idx) {
  return ((u64)f2ord(score) << 32) | (u64)(0xFFFFFFFFu - idx);
}
// order-preserving integer of a score, -0.0 canonicalised to +0.0 (they tie, like in torch)
__device__ __forceinline__ uint32_t score_ord(float s) { return f2ord(s + 0.0f); }
__device__ __forceinline__ u64 ord_key(uint32_t ord, uint32_t idx) {
  return ((u64)ord << 32) | (u64)(0xFFFFFFFFu - idx);
}
// LDS row L (0..63) of the t2-th (0/1) 64-row tile of a chunk -> row offset inside the chunk.
// MFMA tile T = 2*t2 + L/32, A-operand row a = L%32 lands in lane-half (a>>2)&1, element
// e = 4*(a>>3) + (a&3) of the accumulator; it is fed chunk row 64*half + 16*T + e.
// `hs` = rows between the two lane halves: 64 (a 128-row chunk per tile pair), or 128 when the bf16 LDS-DMA pass 1
// works on 256-row chunks (two tile pairs; lane-half h then sees rows 128h .. 128h+127: 128-row groups).
__device__ __forceinline__ int chunk_row(int t2, int L, int hs = 64) {
  const int a = L & 31, T = 2 * t2 + (L >> 5);
  return hs * ((a >> 2) & 1) + 16 * T + 4 * (a >> 3) + (a & 3);
}

// What the LDS-DMA pass 1 stores per (group, query) -- RAW, so that its tile loop spends no VALU instruction on it (every
// one is matrix-core time there); the readers, all memory-bound, decode:
//   gmax word: the bits of the best score (0xFFFFFFFF: empty group)
//   gm2 word : the bits of the runner-up with the low 7 bits replaced by the best item's position (its row in the
//              group, or its 4-row quad).  The runner-up is only ever used as an UPPER bound ("can a second item of
//              this group reach tau?"): ord(bits with the low 7 cleared) | 127 >= ord(runner-up) for either sign.
// The generic pass 1 and the D > 128 form store score_ord values and no runner-up (raw = 0).
__device__ __forceinline__ uint32_t gmax_ord(uint32_t v, int raw) {
  return raw ? (v == 0xFFFFFFFFu ? 0u : score_ord(__uint_as_float(v))) : v;
}
constexpr uint32_t POS_MASK = 127u;  // 7 bits: a row of a 128-row group at most
__device__ __forceinline__ uint32_t gm2_upper_ord(uint32_t v) { return score_ord(__uint_as_float(v & ~POS_MASK)) | POS_MASK; }

struct MipsArgs {
  const void* Q;       // [B, D] queries (fp32 or bf16)
  const void* Cm;      // [C, D] corpus
  int64_t B, C, D;
  int64_t q0, nq;      // this batch: queries q0 .. q0+nq
  int64_t chunks_per_split, n_chunks;
  uint32_t* gmax;      // [n_groups][nq] score_ord of the group max (0 = empty group)
  uint32_t* gm2;       // [n_groups][nq] runner-up + position of the best item (see gmax_ord), or NULL
  const u64* tau;      // [nq] K-th largest group key
  u64* cand;           // [nq][cap]
  int32_t* count;      // [nq]
  int64_t cap;
  const int32_t* glist;  // [nq][K] selected groups (sparse pass 2)
  const int32_t* lcount; // [nq]
  int64_t K;
  int64_t xblocks, splits;  // DMA pass 1: 1-D grid decomposition
  int vec_ok;
  int gshift;           // log2(rows per group): 6; 7 when the bf16 LDS-DMA pass 1 works on 256-row chunks (128 rows per lane half)
  int raw_scores;       // gmax / gm2 are in the raw format of the LDS-DMA pass 1 (see gmax_ord)
  int arg_quads;        // the position in a gm2 word is the 4-row QUAD (0..15) of the best item, not its row (bf16 DMA pass 1)
};

// ---------------------------------------------------------------- operand traits
// F32: DPX = D padded / 8 ; BF16: DPX = D padded / 16
template <int DT, int DPX>
struct Op;

template <int DPX>
struct Op<TT_F32, DPX> {
  static constexpr int DP = DPX * 8;
  static constexpr int LDB = (DP + 4) * 4;  // LDS row stride in bytes
  static constexpr int NLOAD = (CT * DP / 4 + 255) / 256;
  struct Frag { float v[DPX][4]; };
  static __device__ __forceinline__ void load_queries(Frag& f, const void* Q, int64_t row, int64_t nrows, int64_t D, int h, bool vec) {
    const float* X = reinterpret_cast<const float*>(Q);
#pragma unroll
    for (int g = 0; g < DPX; ++g) {
      const int64_t k = 8 * g + 4 * h;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < nrows) {
        const float* p = X + row * D + k;
        if (vec && k + 3 < D) v = *reinterpret_cast<const float4*>(p);
        else {
          if (k + 0 < D) v.x = p[0];
          if (k + 1 < D) v.y = p[1];
          if (k + 2 < D) v.z = p[2];
          if (k + 3 < D) v.w = p[3];
        }
      }
      f.v[g][0] = v.x; f.v[g][1] = v.y; f.v[g][2] = v.z; f.v[g][3] = v.w;
    }
  }
  static __device__ __forceinline__ void fetch(uint4 (&st)[NLOAD], const void* Cm, int64_t t, int64_t nrows, int64_t D, bool vec) {
    constexpr int C4 = DP / 4;
    const float* Y = reinterpret_cast<const float*>(Cm);
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int f = threadIdx.x + 256 * i;
      const int64_t row = (t >> 1) * CHUNK + chunk_row((int)(t & 1), (f / C4) & (CT - 1)), k = 4 * (f % C4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < CT * C4 && row < nrows) {
        const float* p = Y + row * D + k;
        if (vec && k + 3 < D) v = *reinterpret_cast<const float4*>(p);
        else {
          if (k + 0 < D) v.x = p[0];
          if (k + 1 < D) v.y = p[1];
          if (k + 2 < D) v.z = p[2];
          if (k + 3 < D) v.w = p[3];
        }
      }
      st[i] = make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w));
    }
  }
  static __device__ __forceinline__ void commit(const uint4 (&st)[NLOAD], char* Ys) {
    constexpr int C4 = DP / 4;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int f = threadIdx.x + 256 * i;
      if (f < CT * C4) *reinterpret_cast<uint4*>(Ys + (f / C4) * LDB + 16 * (f % C4)) = st[i];
    }
  }
  static constexpr int ESZ = 4;
  // yrow -> this lane's A-operand row (LDS or global), already advanced by 16*h bytes
  static __device__ __forceinline__ f32x16 tile_at(const char* yrow, const Frag& q) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int g = 0; g < DPX; ++g) {
      const float4 y = *reinterpret_cast<const float4*>(yrow + 32 * g);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.x, q.v[g][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.y, q.v[g][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.z, q.v[g][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.w, q.v[g][3], acc, 0, 0, 0);
    }
    return acc;
  }
  static __device__ __forceinline__ f32x16 tile(const char* Ys, const Frag& q, int jt, int r, int h) {
    return tile_at(Ys + (jt * 32 + r) * LDB + 16 * h, q);
  }
};

template <int DPX>
struct Op<TT_BF16, DPX> {
  static constexpr int DP = DPX * 16;
  static constexpr int LDB = DP * 2 + 16;
  static constexpr int NLOAD = (CT * DP / 8 + 255) / 256;
  struct Frag { uint4 v[DPX]; };
  static __device__ __forceinline__ uint4 load8(const uint16_t* X, int64_t row, int64_t k, int64_t D, bool vec) {
    const uint16_t* p = X + row * D + k;
    if (vec && k + 7 < D) return *reinterpret_cast<const uint4*>(p);
    uint16_t t[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) t[c] = (k + c < D) ? p[c] : (uint16_t)0;
    return make_uint4(t[0] | ((uint32_t)t[1] << 16), t[2] | ((uint32_t)t[3] << 16), t[4] | ((uint32_t)t[5] << 16), t[6] | ((uint32_t)t[7] << 16));
  }
  static __device__ __forceinline__ void load_queries(Frag& f, const void* Q, int64_t row, int64_t nrows, int64_t D, int h, bool vec) {
    const uint16_t* X = reinterpret_cast<const uint16_t*>(Q);
#pragma unroll
    for (int g = 0; g < DPX; ++g)
      f.v[g] = (row < nrows) ? load8(X, row, 16 * g + 8 * h, D, vec) : make_uint4(0, 0, 0, 0);
  }
  static __device__ __forceinline__ void fetch(uint4 (&st)[NLOAD], const void* Cm, int64_t t, int64_t nrows, int64_t D, bool vec) {
    constexpr int C8 = DP / 8;
    const uint16_t* Y = reinterpret_cast<const uint16_t*>(Cm);
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int f = threadIdx.x + 256 * i;
      const int64_t row = (t >> 1) * CHUNK + chunk_row((int)(t & 1), (f / C8) & (CT - 1)), k = 8 * (f % C8);
      st[i] = (f < CT * C8 && row < nrows) ? load8(Y, row, k, D, vec) : make_uint4(0, 0, 0, 0);
    }
  }
  static __device__ __forceinline__ void commit(const uint4 (&st)[NLOAD], char* Ys) {
    constexpr int C8 = DP / 8;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int f = threadIdx.x + 256 * i;
      if (f < CT * C8) *reinterpret_cast<uint4*>(Ys + (f / C8) * LDB + 16 * (f % C8)) = st[i];
    }
  }
  static constexpr int ESZ = 2;
  static __device__ __forceinline__ f32x16 tile_at(const char* yrow, const Frag& q) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int g = 0; g < DPX; ++g) {
      const uint4 y = *reinterpret_cast<const uint4*>(yrow + 32 * g);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, y), __builtin_bit_cast(bf16x8, q.v[g]), acc, 0, 0, 0);
    }
    return acc;
  }
  static __device__ __forceinline__ f32x16 tile(const char* Ys, const Frag& q, int jt, int r, int h) {
    return tile_at(Ys + (jt * 32 + r) * LDB + 16 * h, q);
  }
};

// EXPLORATORY (TT_F16X2): fp32-grade scores on the fp16 matrix pipe.  A row is the two-term split of an fp32 row after a
// power-of-two scale (csrc/ce_f16x2.hip explains the arithmetic): [D x fp16 h | D x fp16 l] -- 4 D bytes, the fp32 row's
// size, so staging, tile DMA and the LDS image are the fp32 path's, byte for byte.  A score is the sum over 16-wide
// k-steps of yh qh + yh ql + yl qh, three v_mfma_f32_32x32x16_f16 into one accumulator -- the SAME sequence in pass 1
// and in the sparse re-scoring pass, so their scores agree bit for bit, as the selection requires.  DPX = D / 8.
template <int DPX>
struct Op<TT_F16X2, DPX> : Op<TT_F32, DPX> {
  static constexpr int KS = DPX / 2;  // k-steps of 16
  static constexpr int DP = DPX * 8;
  struct Frag { uint4 h[KS], l[KS]; };
  static __device__ __forceinline__ void load_queries(Frag& f, const void* Q, int64_t row, int64_t nrows, int64_t D, int hh, bool) {
    const uint16_t* X = reinterpret_cast<const uint16_t*>(Q);
#pragma unroll
    for (int g = 0; g < KS; ++g) {
      f.h[g] = (row < nrows) ? *reinterpret_cast<const uint4*>(X + row * 2 * D + 16 * g + 8 * hh) : make_uint4(0, 0, 0, 0);
      f.l[g] = (row < nrows) ? *reinterpret_cast<const uint4*>(X + row * 2 * D + D + 16 * g + 8 * hh) : make_uint4(0, 0, 0, 0);
    }
  }
  // yrow -> this lane's A-operand row (LDS or global), already advanced by 16*h bytes
  static __device__ __forceinline__ f32x16 tile_at(const char* yrow, const Frag& q) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int g = 0; g < KS; ++g) {
      const uint4 yh = *reinterpret_cast<const uint4*>(yrow + 32 * g);
      const uint4 yl = *reinterpret_cast<const uint4*>(yrow + 2 * DP + 32 * g);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, yh), __builtin_bit_cast(f16x8, q.h[g]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, yh), __builtin_bit_cast(f16x8, q.l[g]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, yl), __builtin_bit_cast(f16x8, q.h[g]), acc, 0, 0, 0);
    }
    return acc;
  }
  static __device__ __forceinline__ f32x16 tile(const char* Ys, const Frag& q, int jt, int r, int h) {
    return tile_at(Ys + (jt * 32 + r) * Op<TT_F32, DPX>::LDB + 16 * h, q);
  }
};

// ---------------------------------------------------------------- the dense GEMM pass
// PASS 1: group maxima.  PASS 2 (fallback for ragged D / tiny corpora): candidates.
template <int DT, int DPX, int PASS>
__global__ __launch_bounds__(256) void mips_score_kernel(const MipsArgs p) {
  using O = Op<DT, DPX>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int TILE_BYTES = CT * O::LDB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int64_t ql = (int64_t)blockIdx.x * QB_WG + wave * 32 + r;  // query index inside the batch
  const bool q_ok = ql < p.nq;

  typename O::Frag qf;
  O::load_queries(qf, p.Q, p.q0 + ql, p.q0 + p.nq, p.D, h, p.vec_ok);

  const int64_t c0 = (int64_t)blockIdx.y * p.chunks_per_split;
  const int64_t c1 = (c0 + p.chunks_per_split < p.n_chunks) ? c0 + p.chunks_per_split : p.n_chunks;
  const int64_t t0 = c0 * (CHUNK / CT), t1 = c1 * (CHUNK / CT);

  u64 tau = 0;
  if (PASS == 2 && q_ok) tau = p.tau[ql];

  uint4 st[O::NLOAD];
  if (t0 < t1) {
    O::fetch(st, p.Cm, t0, p.C, p.D, p.vec_ok);
    O::commit(st, smem_raw);
  }
  __syncthreads();
  float best = 0.f;
  bool have = false;
  for (int64_t t = t0; t < t1; ++t) {
    const int cur = (int)((t - t0) & 1);
    if (t + 1 < t1) O::fetch(st, p.Cm, t + 1, p.C, p.D, p.vec_ok);
    const char* ys = smem_raw + cur * TILE_BYTES;
    const int64_t chunk = t >> 1;
    const bool full = (chunk + 1) * CHUNK <= p.C;  // no row of this chunk is past the corpus end
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      const f32x16 acc = O::tile(ys, qf, jt, r, h);
      // this lane's 16 scores are chunk rows 64h + 16T + e, T = 2*(t&1) + jt (see chunk_row)
      const int64_t b0 = chunk * CHUNK + 64 * h + 16 * (2 * (int)(t & 1) + jt);
      if (PASS == 1) {
        if (full) {
          float m = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
#pragma unroll
          for (int e = 4; e < 16; e += 4) m = fmaxf(m, fmaxf(fmaxf(acc[e], acc[e + 1]), fmaxf(acc[e + 2], acc[e + 3])));
          best = have ? fmaxf(best, m) : m;
          have = true;
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (b0 + e < p.C) { best = have ? fmaxf(best, acc[e]) : acc[e]; have = true; }
        }
      } else if (q_ok) {
        const uint32_t grp = (uint32_t)(2 * chunk + h);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t row = b0 + e;
          const uint32_t ord = score_ord(acc[e]);
          if (row < p.C && ord_key(ord, grp) >= tau) {
            const int pos = atomicAdd(&p.count[ql], 1);
            if (pos < p.cap) p.cand[ql * p.cap + pos] = ord_key(ord, (uint32_t)row);
          }
        }
      }
    }
    if (PASS == 1 && (t & 1)) {  // a 128-row chunk is complete: lane-half h holds group 2*chunk + h
      if (q_ok) p.gmax[(2 * chunk + h) * p.nq + ql] = have ? score_ord(best) : 0u;
      have = false;
    }
    if (t + 1 < t1) O::commit(st, smem_raw + (cur ^ 1) * TILE_BYTES);
    __syncthreads();
  }
}

// ---------------------------------------------------------------- pass 1, LDS-DMA form
// Same computation as mips_score_kernel<.., 1> for the common case (16-B aligned rows, D a
// multiple of 32 B/row-chunk sizes: D == DP, >= 8 chunks of 16 B per row):
//   * the corpus tile goes global -> LDS by DMA (no staging registers), unpadded rows with the
//     XOR chunk swizzle of mfma_stream.hpp; the chunk_row permutation is applied to the SOURCE
//     row of each DMA lane;
//   * 2 workgroups per CU, so one wave's max-epilogue overlaps another's MFMAs;
//   * NQ = 2 (bf16): every A fragment read from LDS feeds TWO MFMAs (64 queries per wave) -- at
//     one 1-KiB LDS read per 32-cycle bf16 MFMA the LDS port, not the matrix core, is the limit.
// The lane-dependent part of a tile DMA's addressing is the same for every tile: 2 VGPRs for the whole kernel.
// Everything else (which rows this wave instruction lands, the tile parity, the swizzle of its base row) is
// wave-uniform and goes into the scalar offset of the buffer load.  Keeping one precomputed 32-bit offset per DMA
// instruction and parity instead (8-16 VGPRs) pushed the NQ = 4 bf16 kernel past 256 registers, and a spill
// reload waits on vmcnt -- i.e. on the DMA of the tiles still in flight.
struct DmaLane {
  int vrow;  // chunk_row(0, lane / CPR) * row_bytes
  int vc;    // 16 * ((lane % CPR) ^ ((lane / CPR) & SW))
};
template <int DP8>
__device__ __forceinline__ DmaLane dma_lane(int lane, int64_t row_bytes, int hs) {
  using TM = TileMap<DP8, true>;
  const int lr = lane / TM::CPR;
  return DmaLane{chunk_row(0, lr, hs) * (int)row_bytes, 16 * ((lane % TM::CPR) ^ (lr & TM::SW))};
}

template <int DP8>
__device__ __forceinline__ void corpus_tile_dma(const char* __restrict__ Cm, int64_t row_bytes, int64_t t, int64_t C,
                                                float* Ys, int wave, const DmaLane& dl, int g7) {
  using TM = TileMap<DP8, true>;
  constexpr int RPI = 64 / TM::CPR;  // rows per wave instruction (1 KiB)
  constexpr int NI = CT / RPI / 4;   // instructions per wave
  // buffer loads (see tile_dma in mfma_stream.hpp): chunk base in a scalar descriptor; rows past the end of
  // the corpus are outside the descriptor and land as zeros (the epilogue masks them)
#if TT_MIPS_EXP & 16
  const int64_t chunk0 = ((t >> 1) & 63) * CHUNK, left = C - chunk0;  // measurement variant: the corpus "stream" stays in L2
#else
  // g7: 256-row chunks, this tile pair is its first / second 64 rows of each lane half
  const int64_t chunk0 = (t >> (1 + g7)) * (CHUNK << g7), left = C - chunk0;
#endif
  const int chunk_rows = CHUNK << g7, hs = 64 << g7, pair_rows = g7 ? 64 * (int)((t >> 1) & 1) : 0;
  const int rows_here = left < chunk_rows ? (int)left : chunk_rows;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Cm + chunk0 * row_bytes), 0,
                                                                      rows_here * (int)row_bytes, 0x00020000);
  const int w = __builtin_amdgcn_readfirstlane(wave);
  const int par = (int)(t & 1);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    // LDS rows rbase .. rbase + RPI - 1 (rbase a multiple of RPI, so its bits and the lane's row bits are disjoint:
    // chunk_row(par, rbase + lr) = chunk_row(par, rbase) + chunk_row(0, lr), and the swizzles XOR)
    const int rbase = (w * NI + i) * RPI;
    const int voff = dl.vrow + (dl.vc ^ (16 * (rbase & TM::SW)));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(Ys + rbase * TM::DP), 16, voff,
                                             (chunk_row(par, rbase, hs) + pair_rows) * (int)row_bytes, 0, 0);
  }
}

// max of three: the compiler fuses the two fmaxf into ONE v_max3_f32 (checked in the ISA: no canonicalising
// v_max_f32 x, x on the MFMA results).  Not inline assembly: the hazard recogniser does not see an asm statement read the
// MFMA results and leaves out the wait states between the last v_mfma and the first read of its accumulators
// (measured: stale scores).
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// STAGES: depth of the LDS tile ring.  A bf16 tile is 1024 MFMA cycles (0.4 us) of work per wave,
// less than the global-memory latency, so one tile of prefetch (STAGES = 2) leaves the matrix
// cores waiting for the DMA every iteration; fp32 tiles are 8x longer and 2 stages suffice.
// SHARE (query batches of <= 32, NQ = 1): all four waves hold the SAME 32 queries and split the
// corpus instead -- wave w scores sub-tile w of every 128-row chunk and the four partial
// (best, runner-up, row) triples of a group are merged through LDS at the end of the chunk.
// Without it three of the four waves multiply zero queries (B = 16 at C = 10 M: 3.0 -> 0.9 ms).
// SF = 2 (33..64 queries): two pairs of waves, each pair holds 32 queries and its two waves take
// sub-tile 0 / sub-tile 1 of every 64-row tile (alternating 16-row blocks of each group).
// TT_F16X2 scores of one 32-row sub-tile of the swizzled LDS tile (fp32 image: 32 DPX bytes per row, chunks 0 .. 2 KS - 1
// the h term, 2 KS .. 4 KS - 1 the l term; the swizzle flips the low four chunk bits only, so the halves stay apart):
// the MFMA sequence of Op<TT_F16X2>::tile_at, operand reads one k-step ahead
template <int DPX, int NQ>
__device__ __forceinline__ void score_tile_f16x2(const float* ys, const typename Op<TT_F16X2, DPX>::Frag (&q)[NQ], f32x16 (&acc)[NQ], int jt,
                                                 int r, int h) {
  using TM = TileMap<DPX, true>;
  constexpr int KS = DPX / 2;
#pragma unroll
  for (int n = 0; n < NQ; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
  const int row = jt * 32 + r;
  uint4 yh[2], yl[2];
  yh[0] = *reinterpret_cast<const uint4*>(ys + TM::chunk(row, h));
  yl[0] = *reinterpret_cast<const uint4*>(ys + TM::chunk(row, 2 * KS + h));
#pragma unroll
  for (int g = 0; g < KS; ++g) {
    if (g + 1 < KS) {
      yh[(g + 1) & 1] = *reinterpret_cast<const uint4*>(ys + TM::chunk(row, 2 * (g + 1) + h));
      yl[(g + 1) & 1] = *reinterpret_cast<const uint4*>(ys + TM::chunk(row, 2 * KS + 2 * (g + 1) + h));
    }
    // per query fragment the SAME three products in the same order as Op<TT_F16X2>::tile_at (bit-identical scores); the
    // fragments' chains interleave, and one pair of operand reads feeds all of them
#pragma unroll
    for (int n = 0; n < NQ; ++n)
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, yh[g & 1]), __builtin_bit_cast(f16x8, q[n].h[g]), acc[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NQ; ++n)
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, yh[g & 1]), __builtin_bit_cast(f16x8, q[n].l[g]), acc[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NQ; ++n)
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, yl[g & 1]), __builtin_bit_cast(f16x8, q[n].h[g]), acc[n], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int DT, int DPX, int NQ, int STAGES, int SF>
__global__ __launch_bounds__(256, 2) void mips_pass1_dma_kernel(const MipsArgs p) {
  constexpr bool SHARE = SF != 0;
#ifdef TT_MIPS_ROWARG  // measurement variant (tools/mips_variants.sh): the row-exact epilogue for bf16 as well
  constexpr bool QUADS = false;
#else
  constexpr bool QUADS = DT == TT_BF16;  // the best item of a group is tracked per 4-row quad (see the epilogue)
#endif
  static_assert(SF == 0 || SF == 2 || SF == 4, "shared-query form: waves per query block");
  using O = Op<DT, DPX>;
  using TM = TileMap<DPX, true>;  // row bytes = 32 * DPX for both dtypes
  static_assert(DT == TT_BF16 || NQ == 1 || (DT == TT_F16X2 && NQ == 2), "several query fragments only for the 16-bit forms");
  static_assert(!SHARE || NQ == 1, "shared queries: one fragment");
  constexpr int TILE_FLOATS = CT * TM::DP;
  // The ring's stages are separately NAMED LDS arrays and the tile loop is unrolled by STAGES: only
  // then can the compiler tell that an LDS read of one stage does not depend on the DMA in flight into
  // another (with one dynamic block it waits vmcnt(0) in front of the first LDS read after every DMA
  // issue, which made the ring depth irrelevant).
  // fp32 (2 stages): +14 % on the pass.  bf16 keeps the dynamic ring: its pass is bound by the
  // epilogue's VALU work, and four copies of the NQ = 4 body do not fit the register file.
  constexpr bool NAMED = (STAGES == 2);
  __shared__ __attribute__((aligned(16))) float ring0[NAMED ? TILE_FLOATS : 4];
  __shared__ __attribute__((aligned(16))) float ring1[NAMED ? TILE_FLOATS : 4];
  __shared__ float red[SHARE ? 4 * 64 * 3 : 4];  // SHARE: the four waves' partial triples of a chunk
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];  // the ring when it is not NAMED
  float* const smem = reinterpret_cast<float*>(smem_raw);
  auto stage = [&](int k) -> float* { return NAMED ? (k == 0 ? ring0 : ring1) : smem + k * TILE_FLOATS; };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  // XCD-aware decomposition of the 1-D grid (workgroup L runs on XCD L % 8, each XCD has its own
  // L2): the `xblocks` query blocks that stream the SAME corpus split get consecutive slots on
  // ONE XCD, so the split is fetched from HBM once and hit in that L2 by the others.
  const int64_t total = p.xblocks * p.splits, per_xcd = (total + 7) / 8;
  const int64_t w = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (w >= total) return;
  const int64_t bx = w % p.xblocks, by = w / p.xblocks;
  const int member = SF == 2 ? (wave & 1) : wave;          // position among the waves that share its queries
  const int64_t qbase = SF == 4 ? bx * 32 + r : SF == 2 ? bx * 64 + (wave >> 1) * 32 + r
                                                        : bx * (QB_WG * NQ) + wave * (32 * NQ) + r;

  typename O::Frag qf[NQ];
#pragma unroll
  for (int n = 0; n < NQ; ++n) O::load_queries(qf[n], p.Q, p.q0 + qbase + 32 * n, p.q0 + p.nq, p.D, h, true);

  const int64_t c0 = by * p.chunks_per_split;
  const int64_t c1 = (c0 + p.chunks_per_split < p.n_chunks) ? c0 + p.chunks_per_split : p.n_chunks;
  const int64_t t0 = c0 * (CHUNK / CT), t1 = c1 * (CHUNK / CT);
  const char* Cm = reinterpret_cast<const char*>(p.Cm);
  const int64_t row_bytes = p.D * O::ESZ;

  constexpr int NI = CT / (64 / TM::CPR) / 4;  // DMA instructions per wave per tile
  const int g7 = SHARE ? 0 : (p.gshift == 7);  // 256-row chunks: lane-half h holds rows 128h .. 128h+127 of it (128-row groups)
  const DmaLane dl = dma_lane<DPX>(lane, row_bytes, 64 << g7);
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (t0 + s < t1) corpus_tile_dma<DPX>(Cm, row_bytes, t0 + s, p.C, stage(s), wave, dl, g7);
  if (t0 + STAGES - 1 <= t1) wait_vmcnt<(STAGES - 2) * NI>();  // tile t0 has landed
  else wait_vmcnt<0>();
  __syncthreads();
  // per group (= this lane-half's 64 rows of the chunk): best score, second best, offset of the
  // first best.  med3(m1, m2, x) with m1 >= m2 is the new second best in one instruction.
  constexpr float NEG_INF = -__builtin_huge_valf();
  float m1[NQ], m2[NQ];
  int arg[NQ];
#pragma unroll
  for (int n = 0; n < NQ; ++n) { m1[n] = NEG_INF; m2[n] = NEG_INF; arg[n] = 0; }
  // one tile: `ys` is scored, `dst` (the stage of tile t-1, free since the barrier that ended the
  // previous step) receives tile t + STAGES - 1
  // PAR = t & 1 (t0 is even) as a compile-time constant: the group-relative row of every accumulator element is
  // then an inline constant of the v_cndmask that tracks the best row -- no v_mov per element.
  // a finished chunk: lane-half h holds group 2*chunk + h.  Scalar base (chunk) + one 32-bit lane offset: no 64-bit
  // per-lane address arithmetic in the tile loop (the NQ = 4 kernel has no register to spare for it)
  const uint32_t nq32 = (uint32_t)p.nq, lane_q = (uint32_t)qbase, lane_off = (uint32_t)h * nq32 + lane_q;
  constexpr int HALFPOS = QUADS ? 16 : 64;  // positions (quads / rows) per 64 rows
  auto store_chunk = [&](int64_t chunk) {
#if TT_MIPS_EXP & 32
    const int64_t base = 2 * (chunk & 7) * p.nq;  // measurement variant: the result stores stay in L2
#else
    const int64_t base = 2 * chunk * p.nq;
#endif
    uint32_t* const g1 = p.gmax + base;
    uint32_t* const g2 = p.gm2 + base;
    const bool second_empty = ((2 * chunk + 1) << p.gshift) >= p.C;  // only the corpus' last chunk (wave-uniform)
    // 256-row chunks: positions found in the first tile pair were shifted down by HALFPOS when it ended (see step)
    const int pos_base = g7 ? HALFPOS : 0;
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
#if TT_MIPS_EXP & 8
      if (m1[n] == 12345.678f) {  // measurement variant: (practically) no result stores
#else
      if (lane_q + 32u * n < nq32) {
#endif
        uint32_t best = __float_as_uint(m1[n]);
        if (second_empty && h) best = 0xFFFFFFFFu;
        __builtin_nontemporal_store(best, &g1[lane_off + 32u * n]);
        __builtin_nontemporal_store((__float_as_uint(m2[n]) & ~POS_MASK) | (uint32_t)(arg[n] + pos_base), &g2[lane_off + 32u * n]);
      }
      m1[n] = NEG_INF; m2[n] = NEG_INF; arg[n] = 0;
    }
  };
  auto step = [&](auto par_c, int64_t t, const float* ys, float* dst) {
    constexpr int PAR = decltype(par_c)::value;
    const bool more = t + STAGES - 1 < t1;
    const int64_t chunk = t >> (1 + g7);       // 128-row chunk, or 256-row chunk (g7: two tile pairs)
    const int pair = g7 ? (int)((t >> 1) & 1) : 0;  // which tile pair of a 256-row chunk
    // The previous chunk's results are stored HERE, in front of this step's tile DMA, not at the end of the step that
    // finished the chunk: the wait that ends a step counts outstanding vector-memory instructions of either kind, and
    // with the stores as the newest ones "all but 2 tiles' DMAs" also meant "wait for the tiles just requested and for
    // the stores' write acknowledgements" -- every second tile.
    if (!SHARE && PAR == 0 && pair == 0 && t > t0) store_chunk(chunk - 1);
    if (more) corpus_tile_dma<DPX>(Cm, row_bytes, t + STAGES - 1, p.C, dst, wave, dl, g7);
    const bool full = (chunk + 1) * (CHUNK << g7) <= p.C;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      if (SF == 4 && (2 * PAR + jt) != wave) continue;  // this sub-tile belongs to another wave
      if (SF == 2 && jt != member) continue;                     // the pair's other wave takes this sub-tile
      f32x16 acc[NQ];
      if constexpr (DT == TT_F32) {
        acc[0] = score_tile<DPX, true>(ys, qf[0].v, jt, r, h);
      } else if constexpr (DT == TT_F16X2) {
        score_tile_f16x2<DPX, NQ>(ys, qf, acc, jt, r, h);
      } else {
#pragma unroll
        for (int n = 0; n < NQ; ++n)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
        const int row = jt * 32 + r;
        if constexpr (NAMED) {
#pragma unroll
          for (int g = 0; g < DPX; ++g) {
            const uint4 y = *reinterpret_cast<const uint4*>(ys + TM::chunk(row, 2 * g + h));
#pragma unroll
            for (int n = 0; n < NQ; ++n)
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, y), __builtin_bit_cast(bf16x8, qf[n].v[g]), acc[n], 0, 0, 0);
          }
        } else {
          // Dynamic ring (its stages cannot be told apart by the compiler): the A-operand reads are issued
          // as ds_read_b128 from inline assembly, with our own lgkmcnt waits.  A compiler-visible LDS read
          // here gets `s_waitcnt vmcnt(0)` in front of it -- a wait for the tile DMA'd a moment ago, one
          // full memory latency per 0.4 us tile.  What must have landed (tile t) is guaranteed by the
          // wait_vmcnt + barrier that ended the previous step.  Reads run one k-group ahead of the MFMAs.
          const uint32_t lrow = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)ys + (uint32_t)row * (TM::LD * 4);
          const int sw = row & TM::SW;
          u32x4 yy[2];
          auto rd = [&](u32x4& dst, int g) {
            const uint32_t a = lrow + 16u * (uint32_t)((2 * g + h) ^ sw);
            asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(a) : "memory");
          };
#if TT_MIPS_EXP & 2
          // measurement variant: ONE LDS read per sub-tile instead of one per k-group
          rd(yy[0], 0);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yy[0]) : : "memory");
          yy[1] = yy[0];
#else
          rd(yy[0], 0);
#endif
#pragma unroll
          for (int g = 0; g < DPX; ++g) {
#if !(TT_MIPS_EXP & 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yy[g & 1]) : : "memory");
            if (g + 1 < DPX) rd(yy[(g + 1) & 1], g + 1);
#endif
#pragma unroll
            for (int n = 0; n < NQ; ++n)
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, yy[g & 1]), __builtin_bit_cast(bf16x8, qf[n].v[g]), acc[n], 0, 0, 0);
          }
        }
      }
      const int off0 = 16 * (2 * PAR + jt);  // group-relative row of element 0 (compile-time: jt is unrolled)
#if TT_MIPS_EXP & 1
      // measurement variant (tools/mips_variants.sh): no epilogue -- one add per accumulator quad keeps the MFMAs alive
#pragma unroll
      for (int n = 0; n < NQ; ++n) m1[n] += acc[n][0];
      continue;
#endif
      auto rows_inside = [&]() {  // elements of this lane's 16 rows inside the corpus (last chunk only)
        // the 64-bit part is wave-uniform (scalar registers); per lane only a 32-bit subtract and clamp
        const int64_t rows_left = p.C - chunk * (CHUNK << g7);
        const int left = (rows_left < (CHUNK << g7) ? (int)rows_left : (CHUNK << g7)) - (64 << g7) * h - 64 * pair - off0;
        return left < 0 ? 0 : left > 16 ? 16 : left;
      };
      if constexpr (QUADS) {
        // bf16: VALU and MFMA instructions do not co-issue on a gfx950 SIMD (tools/mfma_valu_coexec.hip), so every
        // epilogue instruction is MFMA time lost: a bf16 tile is 8 MFMAs = 256 cycles, and the row-exact epilogue
        // below (4 instructions per score, 64 per tile and query fragment) is another 256.  Here the best score and
        // the runner-up stay exact but the best item is tracked per QUAD of 4 consecutive rows (elements 4j .. 4j+3
        // are rows off0 + 4j .. + 3): 7 instructions per quad = 1.75 per score.  Pass 2 finds the row inside the quad
        // by scoring its 4 rows again (1 KiB per selected group instead of nothing -- but 16 KiB if it had to score
        // the whole group).  v_max3 / v_med3 take two new scores per instruction:
        //   b1 = med3(m1, x0, x1), a1 = max3(m1, x0, x1): runner-up and best of {m1, x0, x1};  b2, a2 likewise from a1;
        //   the new runner-up is max3(m2, b1, b2) (m2 <= m1 <= a1), the new best a2.
        auto quad = [&](int n, int j, float x0, float x1, float x2, float x3) {
          const float b1 = __builtin_amdgcn_fmed3f(m1[n], x0, x1), a1 = max3f(m1[n], x0, x1);
          const float b2 = __builtin_amdgcn_fmed3f(a1, x2, x3), a2 = max3f(a1, x2, x3);
          m2[n] = max3f(m2[n], b1, b2);
          const bool gt = a2 > m1[n];  // strict: equal scores keep the earlier quad
          arg[n] = gt ? off0 / 4 + j : arg[n];
          m1[n] = a2;
        };
        if (full) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NQ; ++n) quad(n, j, acc[n][4 * j], acc[n][4 * j + 1], acc[n][4 * j + 2], acc[n][4 * j + 3]);
        } else {  // last chunk: rows past the end of the corpus score -inf
          const int valid = rows_inside();
#pragma unroll
          for (int n = 0; n < NQ; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              quad(n, j, 4 * j < valid ? acc[n][4 * j] : NEG_INF, 4 * j + 1 < valid ? acc[n][4 * j + 1] : NEG_INF,
                   4 * j + 2 < valid ? acc[n][4 * j + 2] : NEG_INF, 4 * j + 3 < valid ? acc[n][4 * j + 3] : NEG_INF);
        }
      } else if (full) {
        // fp32 (the MFMA time of a tile is 8x the bf16 one, the epilogue is a few percent of it): the best ROW.
        // Per score: compare, med3 (new runner-up), two selects (best score, best row) = 4 VALU instructions.
        // fmaxf would add a canonicalising v_max per element, a runtime row offset a v_mov, and testing "row < C"
        // inside the loop a 64-bit compare + two selects: 10 instructions per score.  The last (partial) chunk takes
        // the masked copy of the loop through a wave-uniform branch.
        // element-major: the NQ chains are independent, so consecutive instructions never wait on each other's
        // compare result (chain-major order costs an s_nop per element for the VCC hazard)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
#pragma unroll
          for (int n = 0; n < NQ; ++n) {
            const float x = acc[n][e];
            const bool gt = x > m1[n];  // strict: equal scores keep the earlier (smaller) row
            m2[n] = __builtin_amdgcn_fmed3f(m1[n], m2[n], x);
            m1[n] = gt ? x : m1[n];
            arg[n] = gt ? off0 + e : arg[n];
          }
        }
      } else {
        const int valid = rows_inside();
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float x = e < valid ? acc[n][e] : NEG_INF;
            const bool gt = x > m1[n];
            m2[n] = __builtin_amdgcn_fmed3f(m1[n], m2[n], x);
            m1[n] = gt ? x : m1[n];
            arg[n] = gt ? off0 + e : arg[n];
          }
        }
      }
    }
    if (!SHARE && PAR && g7 && pair == 0) {
      // end of the first tile pair of a 256-row chunk: the positions found so far move to -HALFPOS .. -1, so that the
      // second pair can reuse the same compile-time positions 0 .. HALFPOS-1 (store_chunk adds HALFPOS back)
#pragma unroll
      for (int n = 0; n < NQ; ++n) arg[n] -= HALFPOS;
    }
    if (SHARE && PAR) {
      // merge the four waves' partial triples of this chunk (sub-tiles in ascending row order, so
      // "strictly greater" keeps the earlier row on ties, like the sequential scan)
      red[(wave * 64 + lane) * 3] = m1[0];
      red[(wave * 64 + lane) * 3 + 1] = m2[0];
      red[(wave * 64 + lane) * 3 + 2] = __int_as_float(arg[0]);
      __syncthreads();
      if (member == 0) {
        float M1 = NEG_INF, M2 = NEG_INF;
        int A = 0;
#pragma unroll
        for (int w2 = (SF == 2 ? wave : 0); w2 < (SF == 2 ? wave + 2 : 4); ++w2) {
          const float a1 = red[(w2 * 64 + lane) * 3], a2 = red[(w2 * 64 + lane) * 3 + 1];
          const int aa = __float_as_int(red[(w2 * 64 + lane) * 3 + 2]);
          // SF = 2: the two waves' rows interleave in 16-row blocks, so equal scores are decided by the row
          const bool gt = a1 > M1 || (SF == 2 && a1 == M1 && aa < A);
          M2 = fmaxf(fminf(M1, a1), fmaxf(M2, a2));  // second largest of {M1, M2, a1, a2}
          M1 = fmaxf(M1, a1);
          A = gt ? aa : A;
        }
        m1[0] = M1; m2[0] = M2; arg[0] = A;
      }
    }
    if (SHARE && PAR && member == 0) store_chunk(chunk);
    if (SHARE && PAR && member != 0) { m1[0] = NEG_INF; m2[0] = NEG_INF; arg[0] = 0; }
    // tile t+1 must have landed; tiles t+2 .. t+STAGES-1 may still be in flight
    // NOTE (round 3, ISA): __syncthreads() below is `s_waitcnt vmcnt(0); s_barrier` -- the release fence it carries
    // drains every DMA in flight, so the counted wait in front of it is moot and the 4-stage ring prefetches like a
    // 2-stage one.  Measured with the variants of tools/mips_variants.sh on one box, bf16 pass 1: product 2.43-2.44 ms,
    // counted wait + bare s_barrier (64) 2.38-2.42, one tile fewer in flight (192) 2.41-2.44, vmcnt(0) + bare barrier
    // (320) 2.39 -- all the same: with two workgroups per CU the other workgroup's MFMAs cover the drain, the kernel is
    // not waiting for its corpus stream at this point.  Left as it is.
#if TT_MIPS_EXP & 128
    if (more) wait_vmcnt<(STAGES >= 3 ? STAGES - 3 : 0) * NI>();
    else wait_vmcnt<0>();
#elif TT_MIPS_EXP & 256
    wait_vmcnt<0>();
#else
    if (more) wait_vmcnt<(STAGES - 2) * NI>();
    else wait_vmcnt<0>();
#endif
#if TT_MIPS_EXP & 64
    if constexpr (SHARE) {
      __syncthreads();
    } else {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
#elif !(TT_MIPS_EXP & 4)
    __syncthreads();
#endif
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  if constexpr (NAMED) {
    for (int64_t t = t0; t < t1; t += 2) {
      step(P0{}, t, ring0, ring1);
      if (t + 1 < t1) step(P1{}, t + 1, ring1, ring0);
    }
  } else {
    static_assert(NAMED || STAGES % 2 == 0, "the dynamic ring is walked two tiles (one chunk) at a time");
    int cur = 0;
    for (int64_t t = t0; t < t1; t += 2) {  // t1 - t0 is even: whole chunks
      step(P0{}, t, smem + cur * TILE_FLOATS, smem + ((cur + STAGES - 1) % STAGES) * TILE_FLOATS);
      cur = (cur + 1) % STAGES;
      step(P1{}, t + 1, smem + cur * TILE_FLOATS, smem + ((cur + STAGES - 1) % STAGES) * TILE_FLOATS);
      cur = (cur + 1) % STAGES;
    }
  }
  if (!SHARE && t1 > t0) store_chunk((t1 - 1) >> (1 + g7));  // the last chunk (t1 - t0 is a whole number of chunks)
}

// ---------------------------------------------------------------- sparse pass 2
// One wavefront per (query, selected group): the group's 64 rows come straight from global
// memory as the A operand (row a of the MFMA tile = row a of the 32-row half group), the query
// fills all 32 B-operand columns, so lanes 0 and 32 hold the 2 x 16 scores of column 0.  The
// per-element arithmetic is the MFMA's own and does not depend on the operand row / column
// an element sits in, so these scores are bit-identical to pass 1's.
template <int DT, int DPX>
__global__ __launch_bounds__(256) void mips_sparse_kernel(const MipsArgs p) {
  using O = Op<DT, DPX>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int64_t ql = blockIdx.x;
  typename O::Frag qf;
  O::load_queries(qf, p.Q, p.q0 + ql, p.q0 + p.nq, p.D, h, true);
  const u64 tau = p.tau[ql];
  int32_t n_sel = p.lcount[ql];
  if (n_sel > p.K) n_sel = (int32_t)p.K;
  const int32_t* gl = p.glist + ql * p.K;
  const char* base = reinterpret_cast<const char*>(p.Cm);
  const int64_t row_bytes = p.D * O::ESZ;
  const int nslots = gridDim.y * 4;
  const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  // 64 selected groups per wave step, one per lane: the single-candidate groups (the usual case
  // when pass 1 kept the runner-up scores) are emitted with one atomic per wave; the others are
  // re-scored one after the other by the whole wave
  for (int gbase = (blockIdx.y * 4 + wave) * 64; gbase < n_sel; gbase += nslots * 64) {
    const int gi = gbase + lane;
    const bool valid = gi < n_sel;
    const uint32_t my_grp = valid ? (uint32_t)gl[gi] : 0u;
    bool single = false;
    int64_t at = 0;
    uint32_t where = 0;  // position of the best item inside the group (row, or quad)
    if (p.gm2 && valid) {
      at = (int64_t)my_grp * p.nq + ql;
      const uint32_t w2 = p.gm2[at];
      where = w2 & POS_MASK;
      single = ord_key(gm2_upper_ord(w2), my_grp) < tau;  // the runner-up cannot qualify
    }
    const u64 smask = __ballot(single);
    if (smask) {
      int pos0 = 0;
      if (lane == 0) pos0 = atomicAdd(&p.count[ql], __popcll(smask));
      pos0 = __shfl(pos0, 0, 64);
      const int pos = pos0 + __popcll(smask & lt_mask);
      if (!p.arg_quads) {
        if (single && pos < p.cap)
          p.cand[ql * p.cap + pos] = ord_key(gmax_ord(p.gmax[at], 1), (my_grp << p.gshift) + where);
      } else {
        // pass 1 kept the QUAD of the best item: score the quad's 4 rows again, 8 quads per MFMA tile (A-operand row
        // a = 4 * (quad in tile) + row in quad).  Accumulator element e of lane-half hh is A row (e & 3) + 8 * (e >> 2)
        // + 4 * hh, so lane (rr, hh) with rr < 4 finds the four scores of tile quad 2 * rr + hh in acc[4 rr .. 4 rr + 3].
        const uint32_t my_row0 = single ? (my_grp << p.gshift) + 4u * where : 0u;  // rows < 2^32 (checked by the entry point)
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {
          if (!((smask >> (8 * j)) & 0xFFull)) continue;  // wave-uniform: none of these eight groups is single
          const uint32_t qrow0 = (uint32_t)__shfl((int)my_row0, 8 * j + (r >> 2), 64);
          int64_t arow = (int64_t)qrow0 + (r & 3);
          if (arow >= p.C) arow = p.C - 1;  // past the end (or not a single): any valid row, ignored below
          const f32x16 acc = O::tile_at(base + arow * row_bytes + 16 * h, qf);
          const int src = 8 * j + 2 * (r & 3) + h;  // the group lane this lane reports for (if r < 4)
          const uint32_t srow0 = (uint32_t)__shfl((int)my_row0, src, 64);
          const int spos = __shfl(pos, src, 64);
          const bool s_single = (smask >> src) & 1ull;
          float x0 = acc[0], x1 = acc[1], x2 = acc[2], x3 = acc[3];
#pragma unroll
          for (int jj = 1; jj < 4; ++jj)
            if (r == jj) { x0 = acc[4 * jj]; x1 = acc[4 * jj + 1]; x2 = acc[4 * jj + 2]; x3 = acc[4 * jj + 3]; }
          if (r < 4 && s_single && spos < p.cap) {
            // the best of the four, the first on ties: the group's best item (its score is gmax, bit for bit)
            uint32_t bo = score_ord(x0), be = 0;
            const uint32_t o1 = score_ord(x1), o2 = score_ord(x2), o3 = score_ord(x3);
            if ((int64_t)srow0 + 1 < p.C && o1 > bo) { bo = o1; be = 1; }
            if ((int64_t)srow0 + 2 < p.C && o2 > bo) { bo = o2; be = 2; }
            if ((int64_t)srow0 + 3 < p.C && o3 > bo) { bo = o3; be = 3; }
            p.cand[ql * p.cap + spos] = ord_key(bo, srow0 + be);
          }
        }
      }
    }
    u64 todo = __ballot(valid && !single);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const uint32_t grp = (uint32_t)__shfl((int)my_grp, src, 64);
      const int64_t row0 = (int64_t)grp << p.gshift;
      const int n_tiles = 1 << (p.gshift - 5);
      for (int jt = 0; jt < n_tiles; ++jt) {
        int64_t arow = row0 + 32 * jt + r;
        if (arow >= p.C) arow = p.C - 1;  // past the end: any valid row, its scores are ignored
        const f32x16 acc = O::tile_at(base + arow * row_bytes + 16 * h, qf);
        if (r == 0) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int64_t row = row0 + 32 * jt + (e & 3) + 8 * (e >> 2) + 4 * h;
            const uint32_t ord = score_ord(acc[e]);
            if (row < p.C && ord_key(ord, grp) >= tau) {
              const int pos = atomicAdd(&p.count[ql], 1);
              if (pos < p.cap) p.cand[ql * p.cap + pos] = ord_key(ord, (uint32_t)row);
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------- K-th largest group key
// MSD radix select, 8 bits per pass, one (histogram, pick) kernel pair per pass.
//   hist: workgroup = 32 queries x 8 group lanes over one slice of the groups (gridDim.y
//         slices); keys are read gmax[g][q] as 256-B coalesced rows; per-query 256-bin
//         histograms are built in LDS and merged into the global one with atomics.
//   pick: one thread per query walks its 256 bins from the top, fixes the next digit of the
//         threshold, and marks the query done as soon as the bin holds exactly the keys that
//         are still wanted (then the prefix itself is an exact threshold).
// `tau` doubles as the prefix being built.
constexpr int SEL_Q = 32;
__global__ __launch_bounds__(256) void mips_select_hist_kernel(const uint32_t* __restrict__ gmax, int raw, int64_t n_groups,
                                                               int64_t nq, int pass, const u64* __restrict__ tau,
                                                               const int32_t* __restrict__ done,
                                                               int32_t* __restrict__ ghist) {
  __shared__ int32_t hist[SEL_Q][257];
  __shared__ int32_t any_live;
  const int ql = threadIdx.x & 31, lane8 = threadIdx.x >> 5;
  const int64_t q = (int64_t)blockIdx.x * SEL_Q + ql;
  if (threadIdx.x == 0) any_live = 0;
  for (int i = threadIdx.x; i < SEL_Q * 257; i += 256) (&hist[0][0])[i] = 0;
  __syncthreads();
  const bool live = q < nq && !done[q];
  if (live) any_live = 1;
  __syncthreads();
  if (!any_live) return;
  const int shift = 56 - 8 * pass;
  const u64 prefix = live ? tau[q] : 0ull;
  const u64 himask = (pass == 0) ? 0ull : (~0ull << (shift + 8));
  const int64_t per = (n_groups + gridDim.y - 1) / gridDim.y;
  const int64_t g0 = (int64_t)blockIdx.y * per;
  const int64_t g1 = (g0 + per < n_groups) ? g0 + per : n_groups;
  if (live) {
    // eight loads in flight per thread (unconditional, from clamped rows): with one guarded load per
    // iteration the loop ran at one memory latency per element
    constexpr int U = 8;
    for (int64_t gb = g0 + lane8; gb < g1; gb += 8 * U) {
      uint32_t v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t g = gb + 8 * u;
        v[u] = gmax[(g < g1 ? g : g1 - 1) * nq + q];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t g = gb + 8 * u;
        const u64 key = ord_key(gmax_ord(v[u], raw), (uint32_t)g);
        if (g < g1 && (key & himask) == prefix) atomicAdd(&hist[ql][(int)((key >> shift) & 255)], 1);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SEL_Q * 256; i += 256) {
    const int qq = i >> 8, d = i & 255;
    const int32_t c = hist[qq][d];
    const int64_t gq = (int64_t)blockIdx.x * SEL_Q + qq;
    if (c && gq < nq) atomicAdd(&ghist[gq * 256 + d], c);
  }
}

// One WAVEFRONT per query: lane l holds bins 4l .. 4l+3, a suffix sum over the lanes gives "keys in higher bins" for
// every bin at once (a single thread walking the 256 bins with dependent loads took 17-26 us per pass, twice per call).
__global__ __launch_bounds__(256) void mips_select_pick_kernel(int32_t* __restrict__ ghist, int64_t nq, int pass,
                                                               u64* __restrict__ tau, int32_t* __restrict__ want,
                                                               int32_t* __restrict__ done) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq || done[q]) return;  // wave-uniform
  int4* h4 = reinterpret_cast<int4*>(ghist + q * 256);
  const int4 v = h4[lane];
  const int32_t s = (v.x + v.y) + (v.z + v.w);
  int32_t incl = s;  // keys in this lane's bins and in every higher lane's
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t t = __shfl_down(incl, o, 64);
    if (lane + o < 64) incl += t;
  }
  const int32_t w = want[q];
  // cum(d) = keys in bins >= d; the digit is the highest d with cum(d) >= w (0 if there is none)
  const int32_t c3 = incl - s + v.w, c2 = c3 + v.z, c1 = c2 + v.y, c0 = c1 + v.x;
  const int dl = c3 >= w ? 3 : c2 >= w ? 2 : c1 >= w ? 1 : c0 >= w ? 0 : -1;
  const u64 have = __ballot(dl >= 0);
  const int src = have ? 63 - __clzll((long long)have) : 0;
  const int d_src = __shfl(dl, src, 64);
  const int d = have ? 4 * src + d_src : 0;
  const int32_t cum_here = dl == 3 ? c3 : dl == 2 ? c2 : dl == 1 ? c1 : c0;
  const int32_t bin_here = dl == 3 ? v.w : dl == 2 ? v.z : dl == 1 ? v.y : v.x;
  // no bin reaches w: digit 0, everything above bin 0 is counted as "higher" (lane 0: c0 = all keys)
  int32_t cum_d = __shfl(have ? cum_here : c0, src, 64), in_bin = __shfl(have ? bin_here : v.x, src, 64);
  const int32_t acc = cum_d - in_bin;  // keys in bins above d
  h4[lane] = make_int4(0, 0, 0, 0);
  if (lane == 0) {
    tau[q] |= ((u64)d << (56 - 8 * pass));
    want[q] = w - acc;
    if (in_bin == w - acc || pass == 7) done[q] = 1;
  }
}

__global__ void mips_select_init_kernel(int32_t* __restrict__ ghist, int32_t* __restrict__ want,
                                        int32_t* __restrict__ done, int64_t nq, int32_t K) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq * 256) ghist[i] = 0;
  if (i < nq) { want[i] = K; done[i] = 0; }
}

// After the first two digit passes (full scans of gmax) tau's top 16 bits are fixed and only a
// per-cent or so of the groups still share them.  ONE more full scan splits every query's groups:
//   key prefix above tau's (or query already decided and key >= tau)  -> glist, selected for sure
//   same 16-bit prefix, still undecided                               -> surv[q][..] (the keys)
// and the remaining six digit passes run on the short survivor lists, one workgroup per query
// (mips_select_finish_kernel), which also appends the survivors that make it to glist.
constexpr int SPLIT_ITERS = 96;  // groups per thread and slice (3 x 32 match bits per list)

__global__ __launch_bounds__(256) void mips_select_split_kernel(const uint32_t* __restrict__ gmax, int raw, int64_t n_groups,
                                                                int64_t nq, const u64* __restrict__ tau,
                                                                const int32_t* __restrict__ done, int64_t K,
                                                                int32_t* __restrict__ glist, int32_t* __restrict__ lcount,
                                                                u64* __restrict__ surv, int32_t* __restrict__ scount) {
  // Same-address atomics serialise in L2 (~60 ns each): one atomicAdd per listed group put ~2.5 K of
  // them in a row on every query's counter, 0.15 ms whatever the batch size.  Each workgroup now
  // walks its slice ONCE, remembering its (rare: ~1.6 %) matches as bit masks in registers, reserves
  // one range per query and list, and then revisits only the groups whose bit is set.
  __shared__ int32_t cnt_sure[SEL_Q][8], cnt_surv[SEL_Q][8];
  const int ql = threadIdx.x & 31, lane8 = threadIdx.x >> 5;
  const int64_t q = (int64_t)blockIdx.x * SEL_Q + ql;
  const bool live = q < nq;
  const u64 t = live ? tau[q] : 0;
  const bool decided = live && done[q] != 0;
  const bool skip = !live || (decided && !glist);
  const int64_t per = (n_groups + gridDim.y - 1) / gridDim.y;  // host: per <= 8 * SPLIT_ITERS
  const int64_t g0 = (int64_t)blockIdx.y * per;
  const int64_t g1 = (g0 + per < n_groups) ? g0 + per : n_groups;
  uint32_t bits_sure[SPLIT_ITERS / 32], bits_surv[SPLIT_ITERS / 32];
  int32_t n_sure = 0, n_surv = 0;
#pragma unroll
  for (int w = 0; w < SPLIT_ITERS / 32; ++w) {
    bits_sure[w] = 0;
    bits_surv[w] = 0;
    if (skip || g0 + lane8 + 8 * (int64_t)(32 * w) >= g1) continue;
#pragma unroll
    for (int b8 = 0; b8 < 4; ++b8) {  // eight loads in flight, see mips_select_hist_kernel
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t g = g0 + lane8 + 8 * (int64_t)(32 * w + 8 * b8 + u);
        v[u] = gmax[(g < g1 ? g : g1 - 1) * nq + q];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t g = g0 + lane8 + 8 * (int64_t)(32 * w + 8 * b8 + u);
        const u64 key = ord_key(gmax_ord(v[u], raw), (uint32_t)g);
        const bool sure = g < g1 && (decided ? key >= t : (key >> 48) > (t >> 48));
        const bool sv = g < g1 && !sure && !decided && (key >> 48) == (t >> 48);
        bits_sure[w] |= (sure ? 1u : 0u) << (8 * b8 + u);
        bits_surv[w] |= (sv ? 1u : 0u) << (8 * b8 + u);
      }
    }
    n_sure += __popc(bits_sure[w]);
    n_surv += __popc(bits_surv[w]);
  }
  cnt_sure[ql][lane8] = n_sure;
  cnt_surv[ql][lane8] = n_surv;
  __syncthreads();
  if (lane8 == 0 && !skip) {  // exclusive prefix over the eight slots of this query; then add the reserved base
    int32_t tot_a = 0, tot_b = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int32_t a = cnt_sure[ql][k], b = cnt_surv[ql][k];
      cnt_sure[ql][k] = tot_a;
      cnt_surv[ql][k] = tot_b;
      tot_a += a;
      tot_b += b;
    }
    const int32_t base_a = (glist && tot_a > 0) ? atomicAdd(&lcount[q], tot_a) : 0;
    const int32_t base_b = tot_b > 0 ? atomicAdd(&scount[q], tot_b) : 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { cnt_sure[ql][k] += base_a; cnt_surv[ql][k] += base_b; }
  }
  __syncthreads();
  if (skip || (n_sure == 0 && n_surv == 0)) return;
  int32_t pa = cnt_sure[ql][lane8], pb = cnt_surv[ql][lane8];
#pragma unroll
  for (int w = 0; w < SPLIT_ITERS / 32; ++w) {
    uint32_t m = glist ? bits_sure[w] : 0u;
    while (m) {
      const int bit = __ffs(m) - 1;
      m &= m - 1;
      const int64_t g = g0 + lane8 + 8 * (int64_t)(32 * w + bit);
      if (pa < K) glist[q * K + pa] = (int32_t)g;
      ++pa;
    }
    m = bits_surv[w];
    while (m) {
      const int bit = __ffs(m) - 1;
      m &= m - 1;
      const int64_t g = g0 + lane8 + 8 * (int64_t)(32 * w + bit);
      surv[q * n_groups + pb++] = ord_key(gmax_ord(gmax[g * nq + q], raw), (uint32_t)g);
    }
  }
}

__global__ __launch_bounds__(256) void mips_select_finish_kernel(const u64* __restrict__ surv,
                                                                 const int32_t* __restrict__ scount, int64_t n_groups,
                                                                 u64* __restrict__ tau, const int32_t* __restrict__ want,
                                                                 const int32_t* __restrict__ done, int64_t K,
                                                                 int32_t* __restrict__ glist, int32_t* __restrict__ lcount) {
  __shared__ int32_t hist[256];
  __shared__ u64 s_tau;
  __shared__ int32_t s_want, s_done;
  const int64_t q = blockIdx.x;
  if (done[q]) return;  // decided after two digits: the split kernel already listed its groups
  const int32_t n = scount[q];
  const u64* keys = surv + q * n_groups;
  if (threadIdx.x == 0) { s_tau = tau[q]; s_want = want[q]; s_done = 0; }
  __syncthreads();
  for (int pass = 2; pass < 8; ++pass) {
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int shift = 56 - 8 * pass;
    const u64 himask = ~0ull << (shift + 8), prefix = s_tau;
    for (int32_t i = threadIdx.x; i < n; i += 256) {
      const u64 key = keys[i];
      if ((key & himask) == prefix) atomicAdd(&hist[(int)((key >> shift) & 255)], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // same walk as mips_select_pick_kernel
      const int32_t w = s_want;
      int32_t acc = 0;
      int d = 255;
      for (; d > 0; --d) {
        if (acc + hist[d] >= w) break;
        acc += hist[d];
      }
      s_tau = prefix | ((u64)d << shift);
      s_want = w - acc;
      if (hist[d] == w - acc || pass == 7) s_done = 1;
    }
    __syncthreads();
    if (s_done) break;
  }
  const u64 t = s_tau;
  if (threadIdx.x == 0) tau[q] = t;
  if (glist) {
    for (int32_t i = threadIdx.x; i < n; i += 256) {
      const u64 key = keys[i];
      if (key >= t) {
        const int pos = atomicAdd(&lcount[q], 1);
        if (pos < K) glist[q * K + pos] = (int32_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
      }
    }
  }
}

// ---------------------------------------------------------------- per-query candidate sort
// one wavefront per query: stable LSD radix sort (8-bit digits) on ~key (ascending ~key ==
// descending key), then emit top K.  Up to SORT_LDS candidates (the usual ~1.0 K at K = 1000) are
// sorted in LDS; longer lists ping-pong between the query's two global buffers.  Digits on which all
// keys agree (the top byte of the row index, usually the exponent byte of the score) are skipped:
// a stable pass over a constant digit is the identity.
constexpr int SORT_LDS = 2048;

template <typename Ptr>
__device__ __forceinline__ void sort_pass(Ptr a, Ptr b, int32_t n, int shift, int32_t* base, int lane, u64 lt_mask) {
  for (int d = lane; d < 256; d += 64) base[d] = 0;
  __syncthreads();
  for (int i = lane; i < n; i += 64) atomicAdd(&base[(int)((~a[i] >> shift) & 255)], 1);
  __syncthreads();
  {  // exclusive scan of the 256 counters, 4 per lane
    int32_t t[4], s = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) { t[c] = base[4 * lane + c]; s += t[c]; }
    int32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    int32_t run = inc - s;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) { base[4 * lane + c] = run; run += t[c]; }
  }
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    const bool active = i < n;
    const u64 key = active ? a[i] : 0ull;
    const int d = (int)((~key >> shift) & 255);
    u64 mask = __ballot(active);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const u64 bal = __ballot((d >> bit) & 1);
      mask &= ((d >> bit) & 1) ? bal : ~bal;
    }
    const int rank = __popcll(mask & lt_mask);
    int32_t pos = 0;
    if (active) pos = base[d] + rank;
    __builtin_amdgcn_wave_barrier();
    if (active && rank == 0) base[d] += __popcll(mask);
    __builtin_amdgcn_wave_barrier();
    if (active) b[pos] = key;
  }
  __syncthreads();
}

// The usual case (n <= 2048 candidates, ~1.01 K): a 256-thread bitonic sort in LDS.  Keys are distinct (the row index
// is part of the key), so an unstable network gives the same order as the stable radix sort; 66 compare-exchange
// stages of 4 pairs per thread at 2048 slots instead of ~6 radix passes by one wavefront (57 -> ~12 us per call:
// the sort is on the critical path of every call, and a third of the latency of a serving-size batch after the score
// pass).  Queries with more candidates than that (heavy ties) take mips_sort_emit_kernel.
__global__ __launch_bounds__(256) void mips_sort_emit_small_kernel(const u64* __restrict__ cand, const int32_t* __restrict__ count,
                                                                   int64_t cap, int64_t K, int64_t q0,
                                                                   int64_t* __restrict__ idx_out, float* __restrict__ score_out,
                                                                   int32_t* __restrict__ status) {
  __shared__ u64 keys[SORT_LDS];
  const int64_t ql = blockIdx.x;
  int32_t n = count[ql];
  if (n > SORT_LDS) return;  // the wave-per-query kernel sorts this query
  if (n > cap) { n = (int32_t)cap; if (threadIdx.x == 0) atomicOr(status, 1); }  // cannot happen (bound proven in the passes)
  if (n < K && threadIdx.x == 0) atomicOr(status, 2);
  int slots = 64;
  while (slots < n) slots <<= 1;  // power of two >= n (wave-uniform loop)
  const u64* ga = cand + ql * cap;
  for (int i = threadIdx.x; i < slots; i += 256) keys[i] = i < n ? ga[i] : 0ull;  // 0 sorts last (no real key is 0)
  __syncthreads();
  for (int k = 2; k <= slots; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (slots >> 1); t += 256) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
        const bool desc = (lo & k) == 0;  // descending overall: score desc, row asc (the row is stored inverted)
        const u64 a = keys[lo], b = keys[hi];
        if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  for (int64_t k = threadIdx.x; k < K; k += 256) {
    const u64 key = (k < n) ? keys[k] : 0ull;
    idx_out[(q0 + ql) * K + k] = (k < n) ? (int64_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull)) : (int64_t)-1;
    score_out[(q0 + ql) * K + k] = (k < n) ? ord2f((uint32_t)(key >> 32)) : 0.f;
  }
}

__global__ __launch_bounds__(64) void mips_sort_emit_kernel(u64* __restrict__ cand, u64* __restrict__ tmp,
                                                            const int32_t* __restrict__ count, int64_t cap, int64_t K,
                                                            int64_t q0, int64_t* __restrict__ idx_out,
                                                            float* __restrict__ score_out, int32_t* __restrict__ status,
                                                            int skip_small) {
  __shared__ int32_t base[256];
  __shared__ u64 lbuf[2][SORT_LDS];
  const int lane = threadIdx.x;
  const int64_t ql = blockIdx.x;
  int32_t n = count[ql];
  if (skip_small && n <= SORT_LDS) return;  // mips_sort_emit_small_kernel has done this query
  if (n > cap) { n = (int32_t)cap; if (lane == 0) atomicOr(status, 1); }  // cannot happen (bound proven above)
  if (n < K && lane == 0) atomicOr(status, 2);
  u64* ga = cand + ql * cap;
  u64* gb = tmp + ql * cap;
  const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const bool in_lds = n <= SORT_LDS;
  // which digits differ between any two keys
  u64 all_and = ~0ull, all_or = 0ull;
  for (int i = lane; i < n; i += 64) {
    const u64 k = ga[i];
    if (in_lds) lbuf[0][i] = k;
    all_and &= k;
    all_or |= k;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    all_and &= __shfl_xor(all_and, o, 64);
    all_or |= __shfl_xor(all_or, o, 64);
  }
  const u64 varies = all_and ^ all_or;
  __syncthreads();
  int cur = 0;  // buffer holding the current order (LDS: lbuf[cur]; global: cur == 0 -> ga)
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 8 * pass;
    if (((varies >> shift) & 255) == 0) continue;
    if (in_lds) sort_pass(&lbuf[cur][0], &lbuf[cur ^ 1][0], n, shift, base, lane, lt_mask);
    else sort_pass(cur ? gb : ga, cur ? ga : gb, n, shift, base, lane, lt_mask);
    cur ^= 1;
  }
  const u64* src = in_lds ? &lbuf[cur][0] : (cur ? gb : ga);
  for (int64_t k = lane; k < K; k += 64) {
    const u64 key = (k < n) ? src[k] : 0ull;
    idx_out[(q0 + ql) * K + k] = (k < n) ? (int64_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull)) : (int64_t)-1;
    score_out[(q0 + ql) * K + k] = (k < n) ? ord2f((uint32_t)(key >> 32)) : 0.f;
  }
}

// merge of per-shard results: keys from (score, global index) pairs, then the same per-query sort
__global__ void mips_merge_keys_kernel(const float* __restrict__ scores, const int64_t* __restrict__ idx,
                                       int64_t B, int64_t n_cand, u64* __restrict__ cand, int32_t* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * n_cand) {
    const int64_t id = idx[i];
    // a shard with fewer than K rows pads with index -1: lowest possible key, sorts last
    cand[i] = (id >= 0) ? make_key(scores[i], (uint32_t)id) : 0ull;
  }
  if (i < B) count[i] = (int32_t)n_cand;
}

__global__ void mips_zero_kernel(u64* tau, int32_t* count, int32_t* lcount, int32_t* scount, int64_t nq) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) { tau[i] = 0; count[i] = 0; lcount[i] = 0; scount[i] = 0; }
}

// ---------------------------------------------------------------- D > 128: generic form
// The register-stationary kernels above hold a query's D values in registers (D <= 128).  The reference
// accepts any width (torch.matmul, ref:src/baseline_mips_module.py:58), so wider corpora take the same
// pipeline with the two dense passes built from the library's fp32 GEMM instead: per corpus slab of
// WIDE_ROWS rows the [nq, rows] score block is materialised in the workspace (bf16 operands are widened to
// fp32 first: products of bf16 values are exact in fp32, accumulation is fp32 either way), then
//   pass 1: every (query, group of 64 rows) -> gmax          pass 2: scores >= tau -> candidates.
// Both passes run the SAME GEMM launches on the same slabs, so their scores are bit-identical and the
// select / sort stages in between are the ones above.  This is the slow path by construction.
constexpr int64_t WIDE_ROWS = 65536;

__global__ __launch_bounds__(256) void mips_widen_bf16_kernel(const uint16_t* __restrict__ in, float* __restrict__ out,
                                                              int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = __uint_as_float((uint32_t)in[i] << 16);
}

// S [nq][rows] (row stride lds) of corpus rows c0 .. c0+rows (c0 a multiple of 64).  PASS 1: gmax; PASS 2: candidates.
template <int PASS>
__global__ __launch_bounds__(256) void mips_wide_epilogue_kernel(const float* __restrict__ S, int64_t lds, int64_t nq,
                                                                 int64_t c0, int64_t rows, uint32_t* __restrict__ gmax,
                                                                 const u64* __restrict__ tau, u64* __restrict__ cand,
                                                                 int32_t* __restrict__ count, int64_t cap) {
  const int64_t ql = (int64_t)blockIdx.x * 256 + threadIdx.x;  // consecutive threads = consecutive queries (gmax is query-minor)
  const int64_t g = blockIdx.y;                                // group inside the slab
  if (ql >= nq) return;
  const int64_t r0 = g * GROUP, r1 = (r0 + GROUP < rows) ? r0 + GROUP : rows;
  const float* s = S + ql * lds;
  const uint32_t grp = (uint32_t)(c0 / GROUP + g);
  if (PASS == 1) {
    float best = s[r0];
    for (int64_t j = r0 + 1; j < r1; ++j) best = fmaxf(best, s[j]);
    gmax[(int64_t)grp * nq + ql] = score_ord(best);
  } else {
    const u64 t = tau[ql];
    for (int64_t j = r0; j < r1; ++j) {
      const uint32_t ord = score_ord(s[j]);
      if (ord_key(ord, grp) >= t) {
        const int pos = atomicAdd(&count[ql], 1);
        if (pos < cap) cand[ql * cap + pos] = ord_key(ord, (uint32_t)(c0 + j));
      }
    }
  }
}

static int64_t wide_extra_bytes(int64_t qb, int64_t C, int64_t D, int dtype) {
  const int64_t rows = C < WIDE_ROWS ? round_up(C, GROUP) : WIDE_ROWS;
  int64_t n = round_up(qb * rows * 4, 256) + round_up(tt_gemm_workspace_bytes(TT_GEMM_NT, qb, rows, D), 256);
  if (dtype == TT_BF16) n += round_up(qb * D * 4, 256) + round_up(rows * D * 4, 256);
  return n;
}

// one dense pass over the whole corpus for queries q0 .. q0+nq
static int mips_wide_pass(int pass, const MipsArgs& a, int dtype, void* extra, int64_t extra_bytes, int64_t qb, hipStream_t st) {
  const int64_t rows_max = a.C < WIDE_ROWS ? round_up(a.C, GROUP) : WIDE_ROWS;
  Carver cv(extra);
  float* S = cv.take<float>(qb * rows_max);
  const int64_t gemm_bytes = round_up(tt_gemm_workspace_bytes(TT_GEMM_NT, qb, rows_max, a.D), 256);
  char* gws = cv.take<char>(gemm_bytes);
  float *qf = nullptr, *cf = nullptr;
  if (dtype == TT_BF16) { qf = cv.take<float>(qb * a.D); cf = cv.take<float>(rows_max * a.D); }
  if (cv.off > extra_bytes) { set_error("tt_mips_topk (D > 128): workspace"); return TT_E_WORKSPACE; }
  tt_stream_t ts = reinterpret_cast<tt_stream_t>(st);
  const float* Q = reinterpret_cast<const float*>(a.Q) + a.q0 * a.D;
  int rc;
  if (dtype == TT_BF16) {
    mips_widen_bf16_kernel<<<(unsigned)(ceil_div(a.nq * a.D, 256) < 2048 ? ceil_div(a.nq * a.D, 256) : 2048), 256, 0, st>>>(
        reinterpret_cast<const uint16_t*>(a.Q) + a.q0 * a.D, qf, a.nq * a.D);
    if ((rc = check_launch("mips_widen_bf16_kernel"))) return rc;
    Q = qf;
  }
  if (pass == 1) {  // a trailing group with no rows is never visited below: it must read as "empty" (0)
    const int64_t n_groups = 2 * a.n_chunks, tail = n_groups >= 2 ? 2 : n_groups;
    hipError_t he = hipMemsetAsync(a.gmax + (n_groups - tail) * a.nq, 0, tail * a.nq * 4, st);
    if (he != hipSuccess) { set_error("tt_mips_topk: memset: %s", hipGetErrorString(he)); return (int)he; }
  }
  ProfScope prof("mips_score_kernel", st);
  for (int64_t c0 = 0; c0 < a.C; c0 += rows_max) {
    const int64_t rows = (a.C - c0 < rows_max) ? a.C - c0 : rows_max;
    const float* Cc = reinterpret_cast<const float*>(a.Cm) + c0 * a.D;
    if (dtype == TT_BF16) {
      mips_widen_bf16_kernel<<<2048, 256, 0, st>>>(reinterpret_cast<const uint16_t*>(a.Cm) + c0 * a.D, cf, rows * a.D);
      if ((rc = check_launch("mips_widen_bf16_kernel"))) return rc;
      Cc = cf;
    }
    if ((rc = tt_gemm_f32(TT_GEMM_NT, a.nq, rows, a.D, Q, a.D, Cc, a.D, S, rows_max, nullptr, TT_EPI_NONE, nullptr, 0, 0, gws,
                          gemm_bytes, ts)))
      return rc;
    dim3 grid((unsigned)ceil_div(a.nq, 256), (unsigned)ceil_div(rows, GROUP));
    if (pass == 1) mips_wide_epilogue_kernel<1><<<grid, 256, 0, st>>>(S, rows_max, a.nq, c0, rows, a.gmax, a.tau, a.cand, a.count, a.cap);
    else mips_wide_epilogue_kernel<2><<<grid, 256, 0, st>>>(S, rows_max, a.nq, c0, rows, a.gmax, a.tau, a.cand, a.count, a.cap);
    if ((rc = check_launch("mips_wide_epilogue_kernel"))) return rc;
  }
  return 0;
}

constexpr int64_t MIPS_QBATCH = 1024;

struct MipsPlan {
  int dpx;
  int64_t n_chunks, n_groups, cap, qb, splits, chunks_per_split;
};
static bool plan_mips(int64_t B, int64_t C, int64_t D, int64_t K, int dtype, MipsPlan& pl) {
  if (dtype == TT_F32) {
    if (D <= 32) pl.dpx = 4; else if (D <= 64) pl.dpx = 8; else if (D <= 128) pl.dpx = 16; else pl.dpx = 0;  // 0: generic form
  } else if (dtype == TT_BF16) {
    if (D <= 32) pl.dpx = 2; else if (D <= 64) pl.dpx = 4; else if (D <= 128) pl.dpx = 8; else pl.dpx = 0;
  } else if (dtype == TT_F16X2 && D == 128) {
    pl.dpx = 16;
  } else {
    return false;
  }
  pl.n_chunks = ceil_div(C, CHUNK);
  pl.n_groups = 2 * pl.n_chunks;
  const int64_t sel = K < pl.n_groups ? K : pl.n_groups;
  pl.cap = sel * GROUP;
  if (dtype == TT_BF16 || dtype == TT_F16X2 || (dtype == TT_F32 && D >= 32)) {  // the LDS-DMA pass may use 128-row groups: room for K whole groups of those
    const int64_t n7 = 2 * ceil_div(C, 2 * CHUNK), sel7 = K < n7 ? K : n7;
    if (sel7 * 2 * GROUP > pl.cap) pl.cap = sel7 * 2 * GROUP;
  }
  pl.qb = B < MIPS_QBATCH ? B : MIPS_QBATCH;
  const int64_t qblocks = ceil_div(pl.qb, QB_WG);
  int64_t splits = ceil_div(1024, qblocks);
  if (splits > pl.n_chunks) splits = pl.n_chunks;
  if (splits < 1) splits = 1;
  pl.chunks_per_split = ceil_div(pl.n_chunks, splits);
  pl.splits = ceil_div(pl.n_chunks, pl.chunks_per_split);
  return true;
}

template <int DT, int DPX, int PASS>
static int launch_score(const MipsArgs& a, dim3 grid, hipStream_t st) {
  const size_t lds = 2 * CT * Op<DT, DPX>::LDB;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mips_score_kernel<DT, DPX, PASS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("mips_score_kernel: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  ProfScope prof("mips_score_kernel", st);
  mips_score_kernel<DT, DPX, PASS><<<grid, 256, lds, st>>>(a);
  return check_launch("mips_score_kernel");
}
template <int PASS>
static int dispatch_score(int dtype, int dpx, const MipsArgs& a, dim3 grid, hipStream_t st) {
  if (dtype == TT_F32) {
    if (dpx == 4) return launch_score<TT_F32, 4, PASS>(a, grid, st);
    if (dpx == 8) return launch_score<TT_F32, 8, PASS>(a, grid, st);
    return launch_score<TT_F32, 16, PASS>(a, grid, st);
  }
  if (dtype == TT_F16X2) return launch_score<TT_F16X2, 16, PASS>(a, grid, st);
  if (dpx == 2) return launch_score<TT_BF16, 2, PASS>(a, grid, st);
  if (dpx == 4) return launch_score<TT_BF16, 4, PASS>(a, grid, st);
  return launch_score<TT_BF16, 8, PASS>(a, grid, st);
}

template <int DT, int DPX, int NQ, int SF = 0>
static int launch_pass1_dma(MipsArgs a, int64_t, hipStream_t st) {
  constexpr int SHARE = SF;
  // (a deeper ring for the shared-query form -- 4 / 8 stages, one workgroup per CU -- was measured: slower)
  constexpr int STAGES = (DT == TT_BF16) ? 4 : 2;
  const size_t lds = STAGES == 2 ? 0 : STAGES * (size_t)CT * 32 * DPX;  // two stages live in named static arrays
  // 2 workgroups per CU are resident; aim at ~4 rounds of them
  const int64_t xblocks = SF == 4 ? ceil_div(a.nq, 32) : SF == 2 ? ceil_div(a.nq, 64) : ceil_div(a.nq, QB_WG * NQ);
  int64_t splits = ceil_div(2048, xblocks);
  if (splits > a.n_chunks) splits = a.n_chunks;
  a.chunks_per_split = ceil_div(a.n_chunks, splits);
  if (a.gshift == 7) a.chunks_per_split += a.chunks_per_split & 1;  // whole 256-row chunks per split
  splits = ceil_div(a.n_chunks, a.chunks_per_split);
  a.xblocks = xblocks;
  a.splits = splits;
  const int64_t grid = 8 * ceil_div(xblocks * splits, 8);
  ProfScope prof("mips_score_kernel", st);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mips_pass1_dma_kernel<DT, DPX, NQ, STAGES, SHARE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("mips_pass1_dma_kernel: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  mips_pass1_dma_kernel<DT, DPX, NQ, STAGES, SHARE><<<(unsigned)grid, 256, lds, st>>>(a);
  return check_launch("mips_pass1_dma_kernel");
}
// -1: shape not covered by the DMA form
static int dispatch_pass1_dma(int dtype, int dpx, const MipsArgs& a, int64_t splits, hipStream_t st) {
  // small query batch: waves share queries and split the corpus instead (4 waves x 32, 2 pairs x 32)
  const int sf = a.nq <= 32 ? 4 : a.nq <= 64 ? 2 : 0;
  if (dtype == TT_F32) {
    if (dpx == 4) return sf == 4 ? launch_pass1_dma<TT_F32, 4, 1, 4>(a, splits, st) : sf == 2 ? launch_pass1_dma<TT_F32, 4, 1, 2>(a, splits, st) : launch_pass1_dma<TT_F32, 4, 1>(a, splits, st);
    if (dpx == 8) return sf == 4 ? launch_pass1_dma<TT_F32, 8, 1, 4>(a, splits, st) : sf == 2 ? launch_pass1_dma<TT_F32, 8, 1, 2>(a, splits, st) : launch_pass1_dma<TT_F32, 8, 1>(a, splits, st);
    return sf == 4 ? launch_pass1_dma<TT_F32, 16, 1, 4>(a, splits, st) : sf == 2 ? launch_pass1_dma<TT_F32, 16, 1, 2>(a, splits, st) : launch_pass1_dma<TT_F32, 16, 1>(a, splits, st);
  }
  if (dtype == TT_F16X2) {
    if (sf) return sf == 4 ? launch_pass1_dma<TT_F16X2, 16, 1, 4>(a, splits, st) : launch_pass1_dma<TT_F16X2, 16, 1, 2>(a, splits, st);
    // two query fragments per wave (64 queries) from 256 queries on: one pair of operand reads feeds six MFMAs
    const int nq2 = a.nq > 2 * QB_WG;
    return nq2 ? launch_pass1_dma<TT_F16X2, 16, 2>(a, splits, st) : launch_pass1_dma<TT_F16X2, 16, 1>(a, splits, st);
  }
  if (sf) {
    if (dpx == 4) return sf == 4 ? launch_pass1_dma<TT_BF16, 4, 1, 4>(a, splits, st) : launch_pass1_dma<TT_BF16, 4, 1, 2>(a, splits, st);
    if (dpx == 8) return sf == 4 ? launch_pass1_dma<TT_BF16, 8, 1, 4>(a, splits, st) : launch_pass1_dma<TT_BF16, 8, 1, 2>(a, splits, st);
    return -1;
  }
  const int nqf = a.nq > 2 * QB_WG ? 4 : a.nq > QB_WG ? 2 : 1;
  if (dpx == 4) {
    if (nqf == 4) return launch_pass1_dma<TT_BF16, 4, 4>(a, splits, st);
    return nqf == 2 ? launch_pass1_dma<TT_BF16, 4, 2>(a, splits, st) : launch_pass1_dma<TT_BF16, 4, 1>(a, splits, st);
  }
  if (dpx == 8) {
    if (nqf == 4) return launch_pass1_dma<TT_BF16, 8, 4>(a, splits, st);
    return nqf == 2 ? launch_pass1_dma<TT_BF16, 8, 2>(a, splits, st) : launch_pass1_dma<TT_BF16, 8, 1>(a, splits, st);
  }
  return -1;  // bf16 D = 32: 4 chunks per row, below the swizzle's width
}

template <int DT, int DPX>
static int launch_sparse(const MipsArgs& a, dim3 grid, hipStream_t st) {
  ProfScope prof("mips_sparse_kernel", st);
  mips_sparse_kernel<DT, DPX><<<grid, 256, 0, st>>>(a);
  return check_launch("mips_sparse_kernel");
}
static int dispatch_sparse(int dtype, int dpx, const MipsArgs& a, dim3 grid, hipStream_t st) {
  if (dtype == TT_F32) {
    if (dpx == 4) return launch_sparse<TT_F32, 4>(a, grid, st);
    if (dpx == 8) return launch_sparse<TT_F32, 8>(a, grid, st);
    return launch_sparse<TT_F32, 16>(a, grid, st);
  }
  if (dtype == TT_F16X2) return launch_sparse<TT_F16X2, 16>(a, grid, st);
  if (dpx == 2) return launch_sparse<TT_BF16, 2>(a, grid, st);
  if (dpx == 4) return launch_sparse<TT_BF16, 4>(a, grid, st);
  return launch_sparse<TT_BF16, 8>(a, grid, st);
}

}  // namespace tt

using namespace tt;

extern "C" int64_t tt_mips_workspace_bytes(int64_t B, int64_t C, int64_t D, int64_t K, int dtype) {
  MipsPlan pl;
  if (B <= 0 || C <= 0 || D <= 0 || K <= 0 || K > C || !plan_mips(B, C, D, K, dtype, pl)) return 256;
  return round_up(pl.n_groups * pl.qb * 4, 256)  // gmax
         + round_up(pl.n_groups * pl.qb * 4, 256)  // runner-up + position of the best per group
         + round_up(pl.qb * K * 4, 256)          // selected groups
         + round_up(pl.qb * 4, 256)              // their count
         + round_up(pl.n_groups * pl.qb * 8, 256) // select survivors (keys)
         + round_up(pl.qb * 4, 256)              // their count
         + round_up(pl.qb * 256 * 4, 256)        // select histograms
         + 2 * round_up(pl.qb * 4, 256)          // select want / done
         + round_up(pl.qb * 8, 256)              // tau
         + round_up(pl.qb * 4, 256)              // count
         + 2 * round_up(pl.qb * pl.cap * 8, 256) // candidates + sort ping-pong
         + 256                                   // status word
         + (pl.dpx == 0 ? wide_extra_bytes(pl.qb, C, D, dtype) : 0);  // D > 128: score slab, GEMM scratch, widened operands
}

extern "C" int tt_mips_topk(const void* query, const void* corpus, int dtype, int64_t B, int64_t C, int64_t D,
                            int64_t K, int64_t* idx_out, float* score_out, void* ws, int64_t ws_bytes,
                            tt_stream_t stream) {
  if (!query || !corpus || !idx_out || !score_out || !ws) return fail_arg("tt_mips_topk: null pointer");
  if (B <= 0 || C <= 0 || D <= 0 || K <= 0 || K > C || C >= ((int64_t)1 << 32)) return fail_arg("tt_mips_topk: sizes");
  MipsPlan pl;
  if (!plan_mips(B, C, D, K, dtype, pl)) { set_error("tt_mips_topk: unknown dtype %d", dtype); return TT_E_UNSUPPORTED; }
  if (ws_bytes < tt_mips_workspace_bytes(B, C, D, K, dtype)) { set_error("tt_mips_topk: workspace"); return TT_E_WORKSPACE; }
  hipStream_t st = S(stream);
  Carver cv(ws);
  uint32_t* gmax = cv.take<uint32_t>(pl.n_groups * pl.qb);
  uint32_t* gm2 = cv.take<uint32_t>(pl.n_groups * pl.qb);
  int32_t* glist = cv.take<int32_t>(pl.qb * K);
  int32_t* lcount = cv.take<int32_t>(pl.qb);
  u64* surv = cv.take<u64>(pl.n_groups * pl.qb);
  int32_t* scount = cv.take<int32_t>(pl.qb);
  int32_t* ghist = cv.take<int32_t>(pl.qb * 256);
  int32_t* want = cv.take<int32_t>(pl.qb);
  int32_t* done = cv.take<int32_t>(pl.qb);
  u64* tau = cv.take<u64>(pl.qb);
  int32_t* count = cv.take<int32_t>(pl.qb);
  u64* cand = cv.take<u64>(pl.qb * pl.cap);
  u64* tmp = cv.take<u64>(pl.qb * pl.cap);
  int32_t* status = cv.take<int32_t>(1);
  const bool wide = pl.dpx == 0;
  void* wide_ws = reinterpret_cast<char*>(ws) + cv.off;
  const int64_t wide_bytes = ws_bytes - cv.off;
  const int esz = dtype == TT_BF16 ? 2 : 4;  // (a TT_F16X2 row: two fp16 terms per element)
  const bool vec = ((reinterpret_cast<uintptr_t>(query) | reinterpret_cast<uintptr_t>(corpus)) & 15) == 0 &&
                   (D * esz) % 16 == 0;
  hipError_t he = hipMemsetAsync(status, 0, 4, st);
  if (he != hipSuccess) { set_error("tt_mips_topk: memset: %s", hipGetErrorString(he)); return (int)he; }
  int rc;
  for (int64_t q0 = 0; q0 < B; q0 += pl.qb) {
    const int64_t nq = (B - q0 < pl.qb) ? B - q0 : pl.qb;
    MipsArgs a{};
    a.Q = query; a.Cm = corpus; a.B = B; a.C = C; a.D = D; a.q0 = q0; a.nq = nq;
    a.chunks_per_split = pl.chunks_per_split; a.n_chunks = pl.n_chunks;
    a.gmax = gmax; a.tau = tau; a.cand = cand; a.count = count; a.cap = pl.cap; a.vec_ok = vec ? 1 : 0;
    a.glist = glist; a.lcount = lcount; a.K = K;
    a.gm2 = gm2;
    a.gshift = 6;
    dim3 grid((unsigned)ceil_div(nq, QB_WG), (unsigned)pl.splits);
    mips_zero_kernel<<<(unsigned)ceil_div(nq, 256), 256, 0, st>>>(tau, count, lcount, scount, nq);
    if ((rc = check_launch("mips_zero_kernel"))) return rc;
    // sparse pass 2 reads the corpus rows as MFMA fragments straight from global memory
    static const bool no_sparse = getenv("TT_MIPS_NO_SPARSE") != nullptr;
    const int dp = pl.dpx * (dtype == TT_BF16 ? 16 : 8);
    const bool sparse = pl.n_groups > K && vec && D == dp && !no_sparse && !wide;
    const bool no_dma = false;
    static const bool no_g128 = getenv("TT_MIPS_NO_G128") != nullptr;  // A/B: 64-row groups for bf16 too
    // the LDS-DMA pass 1 (not its shared-query forms) with the sparse pass 2: 256-row chunks, i.e. 128-row
    // groups -- half the result bytes of pass 1 and half the groups for the selection, which reads gmax three times
    const int64_t n_groups7 = 2 * ceil_div(C, 2 * CHUNK);
    const bool g128 = sparse && !no_dma && pl.dpx >= 4 && nq > 64 && n_groups7 > K && !no_g128;  // pl.dpx >= 4: every DMA form
    const int64_t n_groups = g128 ? n_groups7 : pl.n_groups;
    if (g128) {
      a.gshift = 7;
      a.n_chunks = n_groups7;  // in 128-row units, even: whole 256-row chunks
    }
    if (pl.n_groups > K) {  // otherwise tau = 0: every item is a candidate (cap == n_groups*64 >= C)
      if (wide) {  // D > 128: dense pass from the library GEMM, group maxima only
        a.gm2 = nullptr;
        rc = mips_wide_pass(1, a, dtype, wide_ws, wide_bytes, pl.qb, st);
      } else {
        rc = (vec && D == dp && !no_dma) ? dispatch_pass1_dma(dtype, pl.dpx, a, pl.splits, st) : -1;
#ifdef TT_MIPS_ROWARG
        a.arg_quads = 0;
#else
        a.arg_quads = dtype == TT_BF16;  // what the DMA pass left in the gm2 words
#endif
        a.raw_scores = 1;
        if (rc == -1) {  // generic pass 1 keeps the group maxima only, as score_ord values
          a.gm2 = nullptr;
          a.raw_scores = 0;
          rc = dispatch_score<1>(dtype, pl.dpx, a, grid, st);
        }
      }
      if (rc) return rc;
      static const bool no_top2 = getenv("TT_MIPS_NO_TOP2") != nullptr;  // A/B: re-score every selected group
      if (no_top2) a.gm2 = nullptr;
      mips_select_init_kernel<<<(unsigned)ceil_div(nq * 256, 256), 256, 0, st>>>(ghist, want, done, nq, (int32_t)K);
      if ((rc = check_launch("mips_select_init_kernel"))) return rc;
      const int64_t qblocks = ceil_div(nq, SEL_Q);
      int64_t slices = ceil_div(2048, qblocks);
      // every slice adds its bins to the same few global counters of a query (in the first pass nearly all keys share
      // one bin), and same-address atomics serialise: cap the slices of small batches
      const int64_t max_slices = 512;
      if (slices > max_slices) slices = max_slices;
      if (slices > ceil_div(n_groups, 64)) slices = ceil_div(n_groups, 64);
      if (slices < 1) slices = 1;
      for (int pass = 0; pass < 2; ++pass) {
        mips_select_hist_kernel<<<dim3((unsigned)qblocks, (unsigned)slices), 256, 0, st>>>(gmax, a.raw_scores, n_groups, nq, pass, tau, done, ghist);
        if ((rc = check_launch("mips_select_hist_kernel"))) return rc;
        mips_select_pick_kernel<<<(unsigned)ceil_div(nq, 4), 256, 0, st>>>(ghist, nq, pass, tau, want, done);
        if ((rc = check_launch("mips_select_pick_kernel"))) return rc;
      }
      int32_t* gl = sparse ? glist : nullptr;
      // fewer, longer slices than the histogram passes: one counter reservation per workgroup, query and list
      int64_t split_slices = slices < 256 ? slices : 256;
      if (split_slices * 8 * SPLIT_ITERS < n_groups) split_slices = ceil_div(n_groups, 8 * SPLIT_ITERS);  // match-bit capacity
      mips_select_split_kernel<<<dim3((unsigned)qblocks, (unsigned)split_slices), 256, 0, st>>>(gmax, a.raw_scores, n_groups, nq, tau, done, K, gl, lcount, surv, scount);
      if ((rc = check_launch("mips_select_split_kernel"))) return rc;
      mips_select_finish_kernel<<<(unsigned)nq, 256, 0, st>>>(surv, scount, n_groups, tau, want, done, K, gl, lcount);
      if ((rc = check_launch("mips_select_finish_kernel"))) return rc;
    }
    if (sparse) {
      int64_t gsplit = ceil_div(8192, nq);  // ~8 K workgroups in flight whatever the batch size
      const int64_t max_split = a.gm2 ? ceil_div(K, 4 * 64) : ceil_div(K, 4);  // a wave takes 64 groups per step
      if (gsplit > max_split) gsplit = max_split;
      if (gsplit < 1) gsplit = 1;
      if ((rc = dispatch_sparse(dtype, pl.dpx, a, dim3((unsigned)nq, (unsigned)gsplit), st))) return rc;
    } else if (wide) {
      if ((rc = mips_wide_pass(2, a, dtype, wide_ws, wide_bytes, pl.qb, st))) return rc;
    } else {
      if ((rc = dispatch_score<2>(dtype, pl.dpx, a, grid, st))) return rc;
    }
    mips_sort_emit_small_kernel<<<(unsigned)nq, 256, 0, st>>>(cand, count, pl.cap, K, q0, idx_out, score_out, status);
    if ((rc = check_launch("mips_sort_emit_small_kernel"))) return rc;
    mips_sort_emit_kernel<<<(unsigned)nq, 64, 0, st>>>(cand, tmp, count, pl.cap, K, q0, idx_out, score_out, status, 1);
    if ((rc = check_launch("mips_sort_emit_kernel"))) return rc;
  }
  return 0;
}

// ---------------------------------------------------------------- TT_F16X2 operands
// scale[0] = the power of two that brings the matrix' largest magnitude into [2^14, 2^15) (1 for an all-zero matrix)
__global__ __launch_bounds__(256) void mips_absmax_kernel(const float* __restrict__ X, int64_t n4, unsigned* __restrict__ out) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(X)[i];
    m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fmaxf(fabsf(v.z), fabsf(v.w)), m));
  }
  __shared__ float part[4];  // one atomic per workgroup, not per wavefront (same-address atomics serialise)
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));
}
__global__ void mips_scale_from_absmax_kernel(const unsigned* __restrict__ absmax, float* __restrict__ scale) {
  const float mx = __uint_as_float(*absmax);
  float s = 1.f;
  if (mx > 0.f && mx < 3.0e38f) {
    int e;
    frexpf(mx, &e);
    s = ldexpf(1.f, 15 - e);
  }
  *scale = s;
}
// one thread per 8 elements: [rows][D] fp32 -> [rows][D h | D l] fp16
__global__ __launch_bounds__(256) void mips_split_rows_kernel(const float* __restrict__ X, int64_t rows, int64_t D,
                                                              const float* __restrict__ scale, uint16_t* __restrict__ out) {
  const int64_t per_row = D / 8;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * per_row) return;
  const int64_t row = i / per_row, c = (i % per_row) * 8;
  const float s = *scale;
  const float4 a = *reinterpret_cast<const float4*>(X + row * D + c), b = *reinterpret_cast<const float4*>(X + row * D + c + 4);
  const float v[8] = {a.x * s, a.y * s, a.z * s, a.w * s, b.x * s, b.y * s, b.z * s, b.w * s};
  f16x8 hh, ll;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const _Float16 q = (_Float16)v[k];
    hh[k] = q;
    ll[k] = (_Float16)(v[k] - (float)q);
  }
  *reinterpret_cast<f16x8*>(out + row * 2 * D + c) = hh;
  *reinterpret_cast<f16x8*>(out + row * 2 * D + D + c) = ll;
}
__global__ void mips_unscale_kernel(float* __restrict__ scores, int64_t n, const float* __restrict__ sa, const float* __restrict__ sb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) scores[i] *= 1.f / (*sa * *sb);  // powers of two: exact
}

extern "C" int tt_mips_split_rows(const float* X, int64_t rows, int64_t D, uint16_t* out, float* scale, void* ws, int64_t ws_bytes,
                                  tt_stream_t stream) {
  if (!X || !out || !scale || !ws) return fail_arg("tt_mips_split_rows: null pointer");
  if (rows <= 0 || D <= 0 || D % 8 || ((uintptr_t)X | (uintptr_t)out) % 16) return fail_arg("tt_mips_split_rows: D % 8 == 0, 16-byte aligned rows");
  if (ws_bytes < 256) { set_error("tt_mips_split_rows: workspace (256 bytes)"); return TT_E_WORKSPACE; }
  hipStream_t st = S(stream);
  unsigned* am = reinterpret_cast<unsigned*>(ws);
  if (hipMemsetAsync(am, 0, 4, st) != hipSuccess) return check_launch("hipMemsetAsync");
  const int64_t n4 = rows * D / 4;
  const int blocks = (int)(n4 / 256 < 1024 ? (n4 + 255) / 256 : 1024);
  mips_absmax_kernel<<<blocks, 256, 0, st>>>(X, n4, am);
  if (int rc = check_launch("mips_absmax_kernel")) return rc;
  mips_scale_from_absmax_kernel<<<1, 1, 0, st>>>(am, scale);
  if (int rc = check_launch("mips_scale_from_absmax_kernel")) return rc;
  mips_split_rows_kernel<<<(unsigned)ceil_div(rows * (D / 8), 256), 256, 0, st>>>(X, rows, D, scale, out);
  return check_launch("mips_split_rows_kernel");
}

extern "C" int tt_mips_unscale(float* scores, int64_t n, const float* scale_a, const float* scale_b, tt_stream_t stream) {
  if (!scores || !scale_a || !scale_b) return fail_arg("tt_mips_unscale: null pointer");
  if (n <= 0) return 0;
  mips_unscale_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, S(stream)>>>(scores, n, scale_a, scale_b);
  return check_launch("mips_unscale_kernel");
}

extern "C" int64_t tt_mips_merge_workspace_bytes(int64_t B, int64_t n_cand) {
  if (B <= 0 || n_cand <= 0) return 256;
  return 2 * round_up(B * n_cand * 8, 256) + round_up(B * 4, 256) + 256;
}

extern "C" int tt_mips_merge(const float* scores, const int64_t* idx, int64_t B, int64_t n_cand, int64_t K,
                             int64_t* idx_out, float* score_out, void* ws, int64_t ws_bytes, tt_stream_t stream) {
  if (!scores || !idx || !idx_out || !score_out || !ws) return fail_arg("tt_mips_merge: null pointer");
  if (B <= 0 || n_cand <= 0 || K <= 0 || K > n_cand || n_cand >= ((int64_t)1 << 31)) return fail_arg("tt_mips_merge: sizes");
  if (ws_bytes < tt_mips_merge_workspace_bytes(B, n_cand)) { set_error("tt_mips_merge: workspace"); return TT_E_WORKSPACE; }
  hipStream_t st = S(stream);
  Carver cv(ws);
  u64* cand = cv.take<u64>(B * n_cand);
  u64* tmp = cv.take<u64>(B * n_cand);
  int32_t* count = cv.take<int32_t>(B);
  int32_t* status = cv.take<int32_t>(1);
  hipError_t he = hipMemsetAsync(status, 0, 4, st);
  if (he != hipSuccess) { set_error("tt_mips_merge: memset: %s", hipGetErrorString(he)); return (int)he; }
  mips_merge_keys_kernel<<<(unsigned)ceil_div(B * n_cand, 256), 256, 0, st>>>(scores, idx, B, n_cand, cand, count);
  int rc = check_launch("mips_merge_keys_kernel");
  if (rc) return rc;
  mips_sort_emit_kernel<<<(unsigned)B, 64, 0, st>>>(cand, tmp, count, n_cand, K, 0, idx_out, score_out, status, 0);
  return check_launch("mips_sort_emit_kernel");
}
