// Exploratory, opt-in (TT_CE_F16X2=1 in the sharded trainer; never the default): the wide-negatives in-batch CE pair
// (tt_inbatch_ce_fwd_du_keep / tt_inbatch_ce_bwd_kept, ref:src/two_tower_base_retrieval.py:287-312) on the 16-bit
// matrix pipe at fp32-grade accuracy.
//
// Every fp32 operand x of a product is cut into TWO fp16 terms, x * s = h + l (s = a power of two that brings the
// operand's largest magnitude to the top of fp16's range; h = fp16(x s), l = fp16(x s - h): 11 + 11 significant
// bits), and a product runs as THREE v_mfma_f32_32x32x16_f16 into one fp32 accumulator -- ah bh + ah bl + al bh; the
// dropped al bl is <= 2^-24 of |a||b|.  Measured element-wise error against float64: rms 4.9e-8 of sum |a_k b_k|
// (an fp32 fma chain: 2.8e-8; tools/f16x2_logits_probe.hip).  96 matrix-pipe cycles per 16 k instead of the 512 of
// eight v_mfma_f32_32x32x2_f32.
//
//   ce16_split_kernel      fp32 [R][128] -> four fp16 images: h, l row-major (operand of products that reduce over d)
//                         and th, tl = [R/16][2][128][8] blocks (operand of products that reduce over ROWS: the lane that
//                         owns column d reads its 8 rows of a 16-row k-step as one 16-B chunk, in the row order the
//                         MFMA result layout hands the other operand over in -- no transposing LDS read anywhere)
//   ce16_fwd_kernel       users stationary (B fragments of both terms in registers, 32 per wave), item tiles by LDS-DMA;
//                         per 32 x 32 tile: logits (24 MFMAs), online softmax per lane (= per user), the probabilities
//                         re-split in registers and E[u] += P I (24 MFMAs, the tile's th / tl image), log2-domain
//                         logits stored for the backward; per (split, user) partials
//   ce16_merge_kernel     splits -> lse (log2 domain), ce (diagonal logit as an fp32 dot product), unit user gradient E/sum - I_diag
//   ce16_bwd_items_kernel items stationary (accumulators only); per 32-user tile the kept logits come straight from
//                         HBM (lane = item, register = user: 128 contiguous bytes per half-wave and user row), the
//                         gradient tile G = coef (p - 1[diag]) is split in registers, dI += G^T U (24 MFMAs, U's th / tl)
//
// Shapes: D = 128, M a multiple of 256, N of 1024.
#include "common.hpp"

// measurement variants (tools/ce16_variants.sh; results are WRONG by design): 1 no logits stores, 2 no E product, 4 non-temporal
// logits stores [correct],
// 8 the tile wait leaves four vector-memory instructions (the logits stores) in flight, 32 backward without the logits loads
#ifndef TT_CE16_EXP
#define TT_CE16_EXP 0
#endif

namespace tt {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int C16_D = 128;
constexpr int C16_TILE = 32;                 // streamed rows per ring stage
constexpr int C16_NW = 8;                    // waves per workgroup: ONE workgroup per CU, its eight waves share one tile ring
                                             // (two 4-wave workgroups per CU fetched every tile twice: the L1 path -- tile
                                             // DMA + logits stores -- was as busy as the matrix pipe)
constexpr int C16_ROWS_WG = 32 * C16_NW;     // stationary rows per workgroup
constexpr int C16_ROW_B = C16_D * 2;         // bytes of one fp16 row
constexpr int C16_RM_B = C16_TILE * C16_ROW_B;   // one term, row-major image of a stage: 8 KiB
constexpr int C16_TR_B = C16_TILE * C16_ROW_B;   // one term, transposed image of a stage (2 blocks of 4 KiB)
constexpr float C16_PSCALE = 32768.f;        // probabilities / gradients are in [-1, 1]: 2^15 brings them to fp16's top
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ int brow(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

struct Images {
  _Float16 *h, *l, *th, *tl;
};

// the power of two that brings `absmax` into [2^14, 2^15)
__device__ __forceinline__ float scale_for(unsigned absmax_bits) {
  const float mx = __uint_as_float(absmax_bits);
  if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.f;
  int e;
  frexpf(mx, &e);  // mx = f 2^e, f in [0.5, 1)
  return ldexpf(1.f, 15 - e);
}
}  // namespace

// ---------------------------------------------------------------------------------------------------- split
// ONE atomic per workgroup (256 threads): with one per wavefront the 4096 same-address atomics of a 1024-block launch
// serialised in L2 and a 4 MB matrix took 50 us to scan (profiles/r04_timeline_emulated_W8_f16x2.txt, first version)
__device__ __forceinline__ void block_atomic_max(float m, unsigned* out) {
  __shared__ float part[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(out, __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));  // non-negative floats order like their bit patterns
}

__global__ __launch_bounds__(256) void ce16_absmax_kernel(const float* __restrict__ X, int64_t ld, int64_t rows, unsigned* __restrict__ out) {
  float m = 0.f;
  const int64_t n4 = rows * (C16_D / 4);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = *reinterpret_cast<const float4*>(X + (i >> 5) * ld + (i & 31) * 4);
    m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fmaxf(fabsf(v.z), fabsf(v.w)), m));
  }
  block_atomic_max(m, out);
}

__global__ __launch_bounds__(256) void ce16_absmax_vec_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
  block_atomic_max(m, out);
}

// one workgroup per 16-row block: row-major terms (thread = row, 8 columns) and the block's transposed image
// (thread = column d and lane half hh: slots q = 0..7 <-> rows (q & 3) + 8 (q >> 2) + 4 hh; [half][d][slot] so that the
// 16 lanes of a ds_read_b128 lane group -- all of one half -- cover 16 consecutive-modulo-16 chunks: no bank conflict;
// [d][half][slot] was a 2-way conflict on every read)
__global__ __launch_bounds__(256) void ce16_split_kernel(const float* __restrict__ X, int64_t ld, int64_t rows, const unsigned* __restrict__ absmax,
                                                      Images im) {
  __shared__ float xs[16][C16_D + 4];
  const float s = scale_for(*absmax);
  const int64_t blk = blockIdx.x;
  const int t = threadIdx.x;
  {
    const int r = t >> 4, c = (t & 15) * 8;
    const int64_t row = blk * 16 + r;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = 0.f;
    if (row < rows) {
      const float4 a = *reinterpret_cast<const float4*>(X + row * ld + c), b = *reinterpret_cast<const float4*>(X + row * ld + c + 4);
      v[0] = a.x * s; v[1] = a.y * s; v[2] = a.z * s; v[3] = a.w * s; v[4] = b.x * s; v[5] = b.y * s; v[6] = b.z * s; v[7] = b.w * s;
    }
    f16x8 hh, ll;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      xs[r][c + k] = v[k];
      const _Float16 q = (_Float16)v[k];
      hh[k] = q;
      ll[k] = (_Float16)(v[k] - (float)q);
    }
    *reinterpret_cast<f16x8*>(im.h + row * C16_D + c) = hh;
    *reinterpret_cast<f16x8*>(im.l + row * C16_D + c) = ll;
  }
  __syncthreads();
  {
    const int d = t >> 1, hh = t & 1;
    f16x8 a, b;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float v = xs[(q & 3) + 8 * (q >> 2) + 4 * hh][d];
      const _Float16 x = (_Float16)v;
      a[q] = x;
      b[q] = (_Float16)(v - (float)x);
    }
    const int64_t at = blk * (C16_D * 16) + (hh * C16_D + d) * 8;  // [block][lane half][d][8 slots]
    *reinterpret_cast<f16x8*>(im.th + at) = a;
    *reinterpret_cast<f16x8*>(im.tl + at) = b;
  }
}

namespace {
// ---------------------------------------------------------------------------------------------------- tile DMA
// row-major image of C16_TILE rows, one term: every wave instruction lands 1 KiB = 4 rows; XOR swizzle of the 16-B chunks
__device__ __forceinline__ void dma_rowmajor(const _Float16* __restrict__ base, char* dst, int wave, int lane) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base), 0, C16_RM_B, 0x00020000);
#pragma unroll
  for (int i = 0; i < 8 / C16_NW; ++i) {  // 8 instructions per term
    const int rbase = (wave * (8 / C16_NW) + i) * 4;
    const int row = rbase + (lane >> 4);
    const int c = (lane & 15) ^ (row & 15);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + rbase * C16_ROW_B), 16,
                                             row * C16_ROW_B + 16 * c, 0, 0, 0);
  }
}
// transposed image of C16_TILE rows (2 blocks), one term: already in fragment order, a linear copy
__device__ __forceinline__ void dma_linear(const _Float16* __restrict__ base, char* dst, int wave, int lane) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base), 0, C16_TR_B, 0x00020000);
#pragma unroll
  for (int i = 0; i < 8 / C16_NW; ++i) {
    const int off = (wave * (8 / C16_NW) + i) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + off), 16, off + lane * 16, 0, 0, 0);
  }
}

// v (already scaled into fp16's range) -> its two terms, 8 at a time
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
  f16x8 a, b;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const _Float16 x = (_Float16)v[k];
    a[k] = x;
    b[k] = (_Float16)(v[k] - (float)x);
  }
  hi = __builtin_bit_cast(u32x4, a);
  lo = __builtin_bit_cast(u32x4, b);
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)

// ---------------------------------------------------------------------------------------------------- forward
struct FwdArgs {
  Images u, it;                 // users (h, l used), items (all four)
  const unsigned* absmax;       // [0] users, [1] items
  int64_t M, N;
  float* logits;                // [M][N] log2-domain
  float *pmax, *psum, *pe;      // [splits][M], [splits][M], [splits][M][128]
  int n_splits;
};

struct FwdStage {  // LDS image of one item tile
  char rm_h[C16_RM_B], rm_l[C16_RM_B], tr_h[C16_TR_B], tr_l[C16_TR_B];
};

__device__ __forceinline__ void fwd_stage_dma(const FwdArgs& p, int64_t item0, FwdStage* st, int wave, int lane) {
  dma_rowmajor(p.it.h + item0 * C16_D, st->rm_h, wave, lane);
  dma_rowmajor(p.it.l + item0 * C16_D, st->rm_l, wave, lane);
  dma_linear(p.it.th + item0 * C16_D, st->tr_h, wave, lane);
  dma_linear(p.it.tl + item0 * C16_D, st->tr_l, wave, lane);
}

// one tile against the wave's 32 stationary users
__device__ __forceinline__ void fwd_tile(const FwdStage* st, const u32x4 (&uh)[8], const u32x4 (&ul)[8], f32x16 (&E)[4], float& mx, float& sm,
                                         float out_scale, float* __restrict__ logit_row, int64_t item0, int r, int h, char* xp) {
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const char* rowh = st->rm_h + r * C16_ROW_B;
  const char* rowl = st->rm_l + r * C16_ROW_B;
  u32x4 ih[2], il[2];
  {
    const int off = (h ^ (r & 15)) * 16;
    ih[0] = *reinterpret_cast<const u32x4*>(rowh + off);
    il[0] = *reinterpret_cast<const u32x4*>(rowl + off);
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t + 1 < 8) {
      const int off = ((2 * (t + 1) + h) ^ (r & 15)) * 16;
      ih[(t + 1) & 1] = *reinterpret_cast<const u32x4*>(rowh + off);
      il[(t + 1) & 1] = *reinterpret_cast<const u32x4*>(rowl + off);
    }
#if !(TT_CE16_EXP & 128)
    __builtin_amdgcn_sched_barrier(0);  // the reads of step t + 1 go out BEFORE step t's MFMAs (else hipcc reuses ih's registers
                                        // and issues the read one MFMA ahead of its use: an LDS latency per k-step)
#endif
    acc = MFMA16(ih[t & 1], uh[t], acc);
    acc = MFMA16(ih[t & 1], ul[t], acc);
    acc = MFMA16(il[t & 1], uh[t], acc);
    __builtin_amdgcn_sched_barrier(0);
  }
  // lane (user r, half h) holds the logits of items item0 + brow(e, h); both halves must use ONE reference maximum
  // (the E product below sums over the items of both)
  float v[16];
  float tmax = -INFINITY;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    v[e] = acc[e] * out_scale;
    tmax = fmaxf(tmax, v[e]);
  }
#if !(TT_CE16_EXP & 1)
  if (logit_row)  // (wave-uniform: null = the caller keeps no logits, tt_ce16_bwd_recompute forms them again)
#if TT_CE16_EXP & (4 | 16 | 256)
  {  // the wave's 32 x 32 tile is one contiguous 4 KiB block of the logits buffer, row-major inside (see tt_hotpath.h)
    float* tile = logit_row + (item0 >> 5) * 1024 + 8 * 0 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const f32x4 q = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
#if TT_CE16_EXP & 4
      __builtin_nontemporal_store(q, reinterpret_cast<f32x4*>(tile + 8 * g));
#else
      // (plain stores: non-temporal ones were measured 25 % slower here -- 1.30 vs 1.04 ms, WRITE_SIZE 3.1 GB for 2.15 GB of
      // logits: the 32-byte pieces of a line no longer merge in L2)
#if TT_CE16_EXP & 16  // (measurement: the same bytes as four fully coalesced 1-KiB stores, WRONG layout)
      *reinterpret_cast<f32x4*>(logit_row - r * 32 + (item0 >> 5) * 1024 + g * 256 + (h * 32 + r) * 4) = q;
#else
      *reinterpret_cast<f32x4*>(tile + 8 * g) = q;
#endif
#endif
    }
  }
#else
  {  // The wave's 32 x 32 tile is one contiguous 4 KiB block of the logits buffer, row-major inside (see tt_hotpath.h).  Straight
     // from the score registers (lane = user) a store instruction would write 32 bytes into each of 32 rows -- 8.5 % of the kernel
     // (variant 16) -- so the tile is turned through the wave's own 4-KiB LDS slice: written as [user][item] with the 16-byte
     // chunks of a row XOR-swizzled by the row (conflict-free both ways), read back 8 lanes per row, stored as four fully
     // coalesced 1-KiB pieces.  One wave, LDS in order: no barrier.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = h * 32 + r;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 q = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};  // items 8 g + 4 h + (0..3) = chunk 2 g + h of row r
      *reinterpret_cast<f32x4*>(xp + r * 128 + (((2 * g + h) ^ (r & 7)) * 16)) = q;
    }
    float* tile = logit_row - r * 32 + (item0 >> 5) * 1024 + lane * 4;
    const char* src = xp + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) * 16);
#pragma unroll
    for (int g = 0; g < 4; ++g)  // rows 8 g + (lane >> 3), chunk lane & 7
      *reinterpret_cast<f32x4*>(tile + g * 256) = *reinterpret_cast<const f32x4*>(src + g * 1024);
  }
#endif
#endif
  tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
  if (tmax > mx) {  // (lane-divergent, rare after the first tiles)
    const float f = __builtin_amdgcn_exp2f(mx - tmax);
    sm *= f;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) E[b][e] *= f;
    mx = tmax;
  }
  u32x4 ph[2], pl[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float pv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float pr = __builtin_amdgcn_exp2f(v[8 * s + q] - mx);
      sm += pr;
      pv[q] = pr * C16_PSCALE;
    }
    split8(pv, ph[s], pl[s]);
  }
  // E^T[d][user] += I^T[d][item] P[item][user]: A = the tile's transposed image, B = the probabilities just formed
#if TT_CE16_EXP & 2
  E[0][0] += __builtin_bit_cast(float, ph[0][0] ^ pl[1][3]);
  return;
#endif
#if TT_CE16_EXP & 128
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int at = s * 4096 + h * 2048 + (b * 32 + r) * 16;
      const u32x4 th = *reinterpret_cast<const u32x4*>(st->tr_h + at);
      const u32x4 tl = *reinterpret_cast<const u32x4*>(st->tr_l + at);
      E[b] = MFMA16(th, ph[s], E[b]);
      E[b] = MFMA16(th, pl[s], E[b]);
      E[b] = MFMA16(tl, ph[s], E[b]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#else
  // the eight (s, b) fragment pairs, each read one pair ahead of its three MFMAs
  u32x4 th[2], tl[2];
  {
    const int at = h * 2048 + r * 16;
    th[0] = *reinterpret_cast<const u32x4*>(st->tr_h + at);
    tl[0] = *reinterpret_cast<const u32x4*>(st->tr_l + at);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int s = i >> 2, b = i & 3;
    if (i + 1 < 8) {
      const int at = ((i + 1) >> 2) * 4096 + h * 2048 + (((i + 1) & 3) * 32 + r) * 16;
      th[(i + 1) & 1] = *reinterpret_cast<const u32x4*>(st->tr_h + at);
      tl[(i + 1) & 1] = *reinterpret_cast<const u32x4*>(st->tr_l + at);
    }
    __builtin_amdgcn_sched_barrier(0x6);  // (VALU / SALU may cross: the second half's exp2 + split still slide under these MFMAs)
    E[b] = MFMA16(th[i & 1], ph[s], E[b]);
    E[b] = MFMA16(th[i & 1], pl[s], E[b]);
    E[b] = MFMA16(tl[i & 1], ph[s], E[b]);
    __builtin_amdgcn_sched_barrier(0x6);
  }
#endif
}

}  // namespace

#if TT_CE16_EXP & 8
#define C16_FWD_WAIT 0x0f74
#else
#define C16_FWD_WAIT 0x0f70
#endif
__global__ __launch_bounds__(64 * C16_NW, 2) void ce16_fwd_kernel(const FwdArgs p) {
  __shared__ __attribute__((aligned(1024))) FwdStage ring0;
  __shared__ __attribute__((aligned(1024))) FwdStage ring1;
  __shared__ __attribute__((aligned(1024))) char xpose[C16_NW][4096];  // per wave: the logits tile on its way to memory (fwd_tile)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  char* xp = xpose[wave];
  const int64_t user = (int64_t)blockIdx.x * C16_ROWS_WG + wave * 32 + r;
  const int split = blockIdx.y;
  const int64_t per = p.N / p.n_splits;  // a multiple of C16_TILE (host)
  const int64_t n0 = split * per;
  const int n_tiles = (int)(per / C16_TILE);
  const float out_scale = LOG2E / (scale_for(p.absmax[0]) * scale_for(p.absmax[1]));

  u32x4 uh[8], ul[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    uh[t] = *reinterpret_cast<const u32x4*>(p.u.h + user * C16_D + 16 * t + 8 * h);
    ul[t] = *reinterpret_cast<const u32x4*>(p.u.l + user * C16_D + 16 * t + 8 * h);
  }
  f32x16 E[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) E[b][e] = 0.f;
  float mx = -INFINITY, sm = 0.f;
  // logits layout: [M / 32][N / 32] tiles of 32 users x 32 items, each 4 KiB contiguous and row-major inside -- the
  // forward writes whole tiles (row-major [M][N] meant 32-byte pieces in 32 rows 4 N bytes apart per store: 0.33 of the
  // kernel's 1.19 ms at N = 65536), the backward reads every tile as 16 coalesced 128-byte rows
  float* logit_row = p.logits ? p.logits + ((user >> 5) * (p.N >> 5)) * 1024 + (user & 31) * 32 : nullptr;

  fwd_stage_dma(p, n0, &ring0, wave, lane);
  for (int tile = 0; tile < n_tiles; tile += 2) {  // two NAMED stages, unrolled by two (distinct LDS objects carry alias scopes)
    __builtin_amdgcn_s_waitcnt(C16_FWD_WAIT);  // vmcnt(0): the tile has landed (and the logits stores of the previous one have left)
    __builtin_amdgcn_s_barrier();
    if (tile + 1 < n_tiles) fwd_stage_dma(p, n0 + (int64_t)(tile + 1) * C16_TILE, &ring1, wave, lane);
    fwd_tile(&ring0, uh, ul, E, mx, sm, out_scale, logit_row, n0 + (int64_t)tile * C16_TILE, r, h, xp);
    if (tile + 1 >= n_tiles) break;
    __builtin_amdgcn_s_waitcnt(C16_FWD_WAIT);
    __builtin_amdgcn_s_barrier();
    if (tile + 2 < n_tiles) fwd_stage_dma(p, n0 + (int64_t)(tile + 2) * C16_TILE, &ring0, wave, lane);
    fwd_tile(&ring1, uh, ul, E, mx, sm, out_scale, logit_row, n0 + (int64_t)(tile + 1) * C16_TILE, r, h, xp);
  }
  sm += __shfl_xor(sm, 32);
  const int64_t at = (int64_t)split * p.M + user;
  if (h == 0) {
    p.pmax[at] = mx;
    p.psum[at] = sm;
  }
  float* pe = p.pe + at * C16_D;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(pe + b * 32 + 8 * g + 4 * h) = make_float4(E[b][4 * g], E[b][4 * g + 1], E[b][4 * g + 2], E[b][4 * g + 3]);
}

// one wavefront per user: splits -> lse, ce, unit gradient
__global__ __launch_bounds__(256) void ce16_merge_kernel(const float* __restrict__ pmax, const float* __restrict__ psum, const float* __restrict__ pe,
                                                         int n_splits, int64_t M, const unsigned* __restrict__ absmax,
                                                         const float* __restrict__ U, int64_t ldu, const float* __restrict__ I, int64_t ldi,
                                                         int64_t diag_off, float* __restrict__ row_lse, float* __restrict__ row_ce,
                                                         float* __restrict__ du_unit, int64_t ld_du) {
  const int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= M) return;
  const int lane = threadIdx.x & 63;
  float m = -INFINITY;
  for (int s = 0; s < n_splits; ++s) m = fmaxf(m, pmax[(int64_t)s * M + u]);
  float tot = 0.f, e0 = 0.f, e1 = 0.f;
  for (int s = 0; s < n_splits; ++s) {
    const float f = exp2f(pmax[(int64_t)s * M + u] - m);
    tot += psum[(int64_t)s * M + u] * f;
    const float2 v = *reinterpret_cast<const float2*>(pe + ((int64_t)s * M + u) * C16_D + 2 * lane);
    e0 = fmaf(v.x, f, e0);
    e1 = fmaf(v.y, f, e1);
  }
  const float inv = 1.f / (tot * C16_PSCALE * scale_for(absmax[1]));
  const float2 iv = *reinterpret_cast<const float2*>(I + (u + diag_off) * ldi + 2 * lane);
  const float2 uv = *reinterpret_cast<const float2*>(U + u * ldu + 2 * lane);
  *reinterpret_cast<float2*>(du_unit + u * ld_du + 2 * lane) = make_float2(e0 * inv - iv.x, e1 * inv - iv.y);
  const float diag = wave_sum(fmaf(uv.x, iv.x, uv.y * iv.y));
  if (lane == 0) {
    // row_lse stays in the log2 domain, like the fp32 pair's: the backward subtracts it from log2-domain logits, and
    // a round trip through the natural logarithm costs an ulp of |lse| -- 1e-3 on saturated rows with logits of 1e4
    const float lse2 = m + log2f(tot);
    row_lse[u] = lse2;
    row_ce[u] = lse2 * LN2 - diag;
  }
}

// ---------------------------------------------------------------------------------------------------- backward, item side
namespace {
struct BwdArgs {
  Images u;                     // users: th, tl used (recomputing form: h, l too)
  Images it;                    // items: h, l (recomputing form only)
  const float* logits;          // [M][N] log2-domain (kept form only)
  const float *row_lse, *coef;  // [M]
  int64_t M, N, diag_off;
  float* dI;
  int64_t lddi;
  const unsigned* absmax;       // [0] users, [2] dL/dce
  float* part;                  // [n_splits][N][128] partial item gradients when the users are split (else unused)
  int n_splits;                 // blockIdx.y: user ranges, so that narrow item sets (N < 65536) still fill the chip
};
struct BwdStage {
  char tr_h[C16_TR_B], tr_l[C16_TR_B];
};
// buffer loads: tile base in a scalar resource descriptor, row offset in a scalar register, ONE 32-bit per-lane offset
// (global loads would carry a 64-bit VGPR address each: 32 registers and 32 VALU adds per tile).  The row stride is made
// opaque so the 16 row offsets are a scalar multiply per tile, not 16 hoisted scalar registers (csrc/inbatch_ce.hip).
__device__ __forceinline__ void bwd_fetch(const BwdArgs& p, int64_t user0, int64_t item_blk, int lane_off, int wave, int lane, float (&s)[16], float& stat) {
  // buffer loads: tile base in a scalar resource descriptor, row offset an immediate, ONE 32-bit per-lane offset
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.logits + ((user0 >> 5) * (p.N >> 5) + item_blk) * 1024), 0, 4096, 0x00020000);
#if TT_CE16_EXP & 32
#pragma unroll
  for (int e = 0; e < 16; ++e) s[e] = (float)(lane_off + e) * 1e-9f;
#else
#pragma unroll
  for (int e = 0; e < 16; ++e)
    s[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane_off, ((e & 3) + 8 * (e >> 2)) * 128, 0));
#endif
  // the tile's 32 (lse, coef) pairs: one value per lane of wave 0, handed to everybody through LDS (every lane needs 16 of
  // each; fetched per lane they would be twice the logits' load traffic and 64 registers of double buffer)
  if (wave == 0) stat = (lane < 32 ? p.row_lse : p.coef)[user0 + (lane & 31)];
}
__device__ __forceinline__ void bwd_tile(const BwdStage* st, const float* stat, const float (&s)[16], f32x16 (&acc)[4],
                                         int64_t user0, int64_t diag_user, float gscale, int r, int h) {
  float lse[16], cf[16];
#pragma unroll
  for (int g = 0; g < 4; ++g) {  // users 8 g + 4 h + (0..3) = brow(4 g + k, h)
    const float4 a = *reinterpret_cast<const float4*>(stat + 8 * g + 4 * h);
    const float4 b = *reinterpret_cast<const float4*>(stat + 32 + 8 * g + 4 * h);
    lse[4 * g] = a.x; lse[4 * g + 1] = a.y; lse[4 * g + 2] = a.z; lse[4 * g + 3] = a.w;
    cf[4 * g] = b.x; cf[4 * g + 1] = b.y; cf[4 * g + 2] = b.z; cf[4 * g + 3] = b.w;
  }
  u32x4 gh[2], gl[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float gv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = 8 * k + q;
      float pr = __builtin_amdgcn_exp2f(s[e] - lse[e]);
      if (user0 + brow(e, h) == diag_user) pr -= 1.f;
      gv[q] = pr * (cf[e] * gscale);  // |p - 1[diag]| <= 1 and max |coef| gscale < 2^15: inside fp16's range whatever coef holds
    }
    split8(gv, gh[k], gl[k]);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int at = k * 4096 + h * 2048 + (b * 32 + r) * 16;
      const u32x4 th = *reinterpret_cast<const u32x4*>(st->tr_h + at);
      const u32x4 tl = *reinterpret_cast<const u32x4*>(st->tr_l + at);
      acc[b] = MFMA16(th, gh[k], acc[b]);
      acc[b] = MFMA16(th, gl[k], acc[b]);
      acc[b] = MFMA16(tl, gh[k], acc[b]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- the same tile WITHOUT kept logits: the 32 x 32 logits are formed again on the fp16 pipe -- 24 matrix instructions, the
// user tile's row-major terms (a second pair of LDS images per stage) against the wave's items held as B fragments -- in the
// register layout the kept form loads them in (lane = item, register e = user brow(e, h)).  At W = 8 the step with the split
// pair moved half of its HBM bytes as kept logits: 2.1 GB less written by the forward, 2.1 GB less read here.
__device__ __forceinline__ void bwd_tile_rc(const FwdStage* st, const float* stat, const unsigned (&koff)[8], unsigned tbase, const u32x4 (&ih)[8],
                                            const char* ilp, f32x16 (&acc)[4], float out_scale, int64_t user0, int64_t diag_user,
                                            float gscale, int h) {
  f32x16 sacc;
#pragma unroll
  for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const u32x4 ah = *reinterpret_cast<const u32x4*>(st->rm_h + koff[t]);
    const u32x4 al = *reinterpret_cast<const u32x4*>(st->rm_l + koff[t]);
    const u32x4 ilt = *reinterpret_cast<const u32x4*>(ilp + t * 1024);  // (the items' low terms live in the wave's LDS slice: 32 registers)
    // rows = the tile's users (LDS), columns = this wave's items.  The three products in the FORWARD's order (item high x user
    // high, item high x user low, item low x user high): the tile must come out bit for bit as the forward saw it -- the row's
    // lse was formed from those values, and at logits of 1e4 a last-bit difference is 1e-3 in the exponent
    sacc = MFMA16(ah, ih[t], sacc);
    sacc = MFMA16(al, ih[t], sacc);
    sacc = MFMA16(ah, ilt, sacc);
  }
  u32x4 gh[2], gl[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float lse[8], cf[8];
#pragma unroll
    for (int g = 0; g < 2; ++g) {  // users 8 (2 k + g) + 4 h + (0..3) = brow(8 k + 4 g + (0..3), h)
      const float4 a = *reinterpret_cast<const float4*>(stat + 8 * (2 * k + g) + 4 * h);
      const float4 b = *reinterpret_cast<const float4*>(stat + 32 + 8 * (2 * k + g) + 4 * h);
      lse[4 * g] = a.x; lse[4 * g + 1] = a.y; lse[4 * g + 2] = a.z; lse[4 * g + 3] = a.w;
      cf[4 * g] = b.x; cf[4 * g + 1] = b.y; cf[4 * g + 2] = b.z; cf[4 * g + 3] = b.w;
    }
    float gv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = 8 * k + q;
      // the product ROUNDED, then the difference -- not one fma: the row's lse was formed from the rounded logits, and at
      // logits of 1e4 the unrounded product is up to 1e-3 away from what the forward saw (tools/fuzz_ce16.py, scale 30)
      float lg = sacc[e] * out_scale;
      asm volatile("" : "+v"(lg));  // (keeps hipcc from contracting product and difference into one v_fma: __fmul_rn does not)
      float pr = __builtin_amdgcn_exp2f(lg - lse[q]);
      if (user0 + brow(e, h) == diag_user) pr -= 1.f;
      gv[q] = pr * (cf[q] * gscale);
    }
    split8(gv, gh[k], gl[k]);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const u32x4 th = *reinterpret_cast<const u32x4*>(st->tr_h + tbase + (k * 4096 + b * 512));
      const u32x4 tl = *reinterpret_cast<const u32x4*>(st->tr_l + tbase + (k * 4096 + b * 512));
      acc[b] = MFMA16(th, gh[k], acc[b]);
      acc[b] = MFMA16(th, gl[k], acc[b]);
      acc[b] = MFMA16(tl, gh[k], acc[b]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
__device__ __forceinline__ void bwd_stage_dma_rc(const BwdArgs& p, int64_t user0, FwdStage* st, int wave, int lane) {
  dma_rowmajor(p.u.h + user0 * C16_D, st->rm_h, wave, lane);
  dma_rowmajor(p.u.l + user0 * C16_D, st->rm_l, wave, lane);
  dma_linear(p.u.th + user0 * C16_D, st->tr_h, wave, lane);
  dma_linear(p.u.tl + user0 * C16_D, st->tr_l, wave, lane);
}
}  // namespace

// SPLIT: the users are cut into p.n_splits ranges (blockIdx.y), partial results; otherwise one workgroup streams all of them
template <bool SPLIT>
__global__ __launch_bounds__(64 * C16_NW, 2) void ce16_bwd_items_kernel(const BwdArgs p) {
  __shared__ __attribute__((aligned(1024))) BwdStage ring0;
  __shared__ __attribute__((aligned(1024))) BwdStage ring1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int64_t item = (int64_t)blockIdx.x * C16_ROWS_WG + wave * 32 + r;
  const int64_t diag_user = item - p.diag_off;  // the user whose positive this item is (outside [0, M): none)
  const int tiles_all = (int)(p.M / C16_TILE), per = SPLIT ? (tiles_all + p.n_splits - 1) / p.n_splits : tiles_all;
  const int tile0 = SPLIT ? blockIdx.y * per : 0;
  const int n_tiles = SPLIT ? (tile0 + per < tiles_all ? tile0 + per : tiles_all) - tile0 : tiles_all;  // (SPLIT: may be <= 0)
  const float gscale = scale_for(p.absmax[2]);  // from max |coef|
  const int lane_off = (4 * h * 32 + r) * 4;  // bytes from the tile's first logit to this lane's column, rows 4 h ..
  const int64_t item_blk = (int64_t)blockIdx.x * C16_NW + __builtin_amdgcn_readfirstlane(wave);  // (wave-uniform: scalar descriptor)
  f32x16 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
  __shared__ __attribute__((aligned(16))) float stat0[64];
  __shared__ __attribute__((aligned(16))) float stat1[64];
  float s0[16], s1[16];
  float sv0 = 0.f, sv1 = 0.f;
  const int64_t ubase = (int64_t)tile0 * C16_TILE;  // this split's first user
  if (n_tiles > 0) {
    dma_linear(p.u.th + ubase * C16_D, ring0.tr_h, wave, lane);
    dma_linear(p.u.tl + ubase * C16_D, ring0.tr_l, wave, lane);
    bwd_fetch(p, ubase, item_blk, lane_off, wave, lane, s0, sv0);
  }
  for (int tile = 0; tile < n_tiles; tile += 2) {
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if (wave == 0) stat0[lane] = sv0;
    __syncthreads();
    if (tile + 1 < n_tiles) {
      const int64_t u1 = ubase + (int64_t)(tile + 1) * C16_TILE;
      dma_linear(p.u.th + u1 * C16_D, ring1.tr_h, wave, lane);
      dma_linear(p.u.tl + u1 * C16_D, ring1.tr_l, wave, lane);
      bwd_fetch(p, u1, item_blk, lane_off, wave, lane, s1, sv1);
    }
    bwd_tile(&ring0, stat0, s0, acc, ubase + (int64_t)tile * C16_TILE, diag_user, gscale, r, h);
    if (tile + 1 >= n_tiles) break;
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if (wave == 0) stat1[lane] = sv1;
    __syncthreads();
    if (tile + 2 < n_tiles) {
      const int64_t u2 = ubase + (int64_t)(tile + 2) * C16_TILE;
      dma_linear(p.u.th + u2 * C16_D, ring0.tr_h, wave, lane);
      dma_linear(p.u.tl + u2 * C16_D, ring0.tr_l, wave, lane);
      bwd_fetch(p, u2, item_blk, lane_off, wave, lane, s0, sv0);
    }
    bwd_tile(&ring1, stat1, s1, acc, ubase + (int64_t)(tile + 1) * C16_TILE, diag_user, gscale, r, h);
  }
  const float inv = 1.f / (gscale * scale_for(p.absmax[0]));
  float* out = SPLIT ? p.part + ((int64_t)blockIdx.y * p.N + item) * C16_D : p.dI + item * p.lddi;
  const int64_t ldo = SPLIT ? C16_D : p.lddi;
  (void)ldo;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(out + b * 32 + 8 * g + 4 * h) =
          make_float4(acc[b][4 * g] * inv, acc[b][4 * g + 1] * inv, acc[b][4 * g + 2] * inv, acc[b][4 * g + 3] * inv);
}

template <bool SPLIT>
__global__ __launch_bounds__(64 * C16_NW, 2) void ce16_bwd_items_rc_kernel(const BwdArgs p) {
  __shared__ __attribute__((aligned(1024))) FwdStage ring0;
  __shared__ __attribute__((aligned(1024))) FwdStage ring1;
  __shared__ __attribute__((aligned(16))) float stat0[64];
  __shared__ __attribute__((aligned(16))) float stat1[64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int64_t item = (int64_t)blockIdx.x * C16_ROWS_WG + wave * 32 + r;
  const int64_t diag_user = item - p.diag_off;
  const int tiles_all = (int)(p.M / C16_TILE), per = SPLIT ? (tiles_all + p.n_splits - 1) / p.n_splits : tiles_all;
  const int tile0 = SPLIT ? blockIdx.y * per : 0;
  const int n_tiles = SPLIT ? (tile0 + per < tiles_all ? tile0 + per : tiles_all) - tile0 : tiles_all;
  const float gscale = scale_for(p.absmax[2]);
  const float out_scale = LOG2E / (scale_for(p.absmax[0]) * scale_for(p.absmax[1]));
  unsigned koff[8];  // the lane's LDS offsets inside a stage, computed once (else hipcc keeps a set per stage)
#pragma unroll
  for (int t = 0; t < 8; ++t) koff[t] = (unsigned)(r * C16_ROW_B + ((2 * t + h) ^ (r & 15)) * 16);
  const unsigned tbase = (unsigned)(h * 2048 + r * 16);
  // this lane's item as B fragments (8 consecutive d per k-group): the high terms in registers, the low terms in the wave's own
  // 8-KiB LDS slice (acc + both terms = 192 registers left no room for the tile's work: spills inside the loop)
  __shared__ __attribute__((aligned(1024))) char ilfrag[C16_NW][8192];
  char* ilp = ilfrag[wave] + lane * 16;
  u32x4 ih[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    ih[t] = *reinterpret_cast<const u32x4*>(p.it.h + item * C16_D + 16 * t + 8 * h);
    *reinterpret_cast<u32x4*>(ilp + t * 1024) = *reinterpret_cast<const u32x4*>(p.it.l + item * C16_D + 16 * t + 8 * h);
  }
  f32x16 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
  float sv0 = 0.f, sv1 = 0.f;
  const int64_t ubase = (int64_t)tile0 * C16_TILE;
  if (n_tiles > 0) {
    bwd_stage_dma_rc(p, ubase, &ring0, wave, lane);
    if (wave == 0) sv0 = (lane < 32 ? p.row_lse : p.coef)[ubase + (lane & 31)];
  }
  for (int tile = 0; tile < n_tiles; tile += 2) {
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if (wave == 0) stat0[lane] = sv0;
    __syncthreads();
    if (tile + 1 < n_tiles) {
      const int64_t u1 = ubase + (int64_t)(tile + 1) * C16_TILE;
      bwd_stage_dma_rc(p, u1, &ring1, wave, lane);
      if (wave == 0) sv1 = (lane < 32 ? p.row_lse : p.coef)[u1 + (lane & 31)];
    }
    bwd_tile_rc(&ring0, stat0, koff, tbase, ih, ilp, acc, out_scale, ubase + (int64_t)tile * C16_TILE, diag_user, gscale, h);
    if (tile + 1 >= n_tiles) break;
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if (wave == 0) stat1[lane] = sv1;
    __syncthreads();
    if (tile + 2 < n_tiles) {
      const int64_t u2 = ubase + (int64_t)(tile + 2) * C16_TILE;
      bwd_stage_dma_rc(p, u2, &ring0, wave, lane);
      if (wave == 0) sv0 = (lane < 32 ? p.row_lse : p.coef)[u2 + (lane & 31)];
    }
    bwd_tile_rc(&ring1, stat1, koff, tbase, ih, ilp, acc, out_scale, ubase + (int64_t)(tile + 1) * C16_TILE, diag_user, gscale, h);
  }
  const float inv = 1.f / (gscale * scale_for(p.absmax[0]));
  float* out = SPLIT ? p.part + ((int64_t)blockIdx.y * p.N + item) * C16_D : p.dI + item * p.lddi;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(out + b * 32 + 8 * g + 4 * h) =
          make_float4(acc[b][4 * g] * inv, acc[b][4 * g + 1] * inv, acc[b][4 * g + 2] * inv, acc[b][4 * g + 3] * inv);
}


// dI[i][:] = sum over user splits of part[s][i][:], in split order (deterministic)
__global__ __launch_bounds__(256) void ce16_bwd_reduce_kernel(const float* __restrict__ part, int n_splits, int64_t N, float* __restrict__ dI,
                                                              int64_t lddi) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4
  if (i >= N * (C16_D / 4)) return;
  const int64_t row = i >> 5, c = (i & 31) * 4;
  float4 a = *reinterpret_cast<const float4*>(part + row * C16_D + c);
  for (int s = 1; s < n_splits; ++s) {
    const float4 b = *reinterpret_cast<const float4*>(part + ((int64_t)s * N + row) * C16_D + c);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  *reinterpret_cast<float4*>(dI + row * lddi + c) = a;
}

// ---------------------------------------------------------------------------------------------------- host
namespace {
// user splits of the backward: one workgroup per 256 items is enough from N = 65536 on; narrower item sets split the users
int pick_bwd_splits(int64_t M, int64_t N) {
  int want = 1;
  while (want < 16 && (N / C16_ROWS_WG) * want < 256 && (M / C16_TILE) / (2 * want) >= 8) want *= 2;
  return want;
}

// item splits of the forward: enough workgroups for the 256 CUs, N / splits a multiple of the tile
int pick_splits(int64_t M, int64_t N) {
  int want = 8;
  while (want < 32 && (M / C16_ROWS_WG) * want < 256) want *= 2;
  while (want > 1 && N % ((int64_t)want * 128)) want /= 2;
  return want;
}

struct Ws {
  unsigned* absmax;
  Images u, it;
  float *pmax, *psum, *pe, *part;
};
int64_t carve(void* base, int64_t M, int64_t N, Ws* w) {
  Carver c(base);
  const int splits = pick_splits(M, N);
  unsigned* am = c.take<unsigned>(64);
  Images u, it;
  u.h = c.take<_Float16>(M * C16_D); u.l = c.take<_Float16>(M * C16_D); u.th = c.take<_Float16>(M * C16_D); u.tl = c.take<_Float16>(M * C16_D);
  it.h = c.take<_Float16>(N * C16_D); it.l = c.take<_Float16>(N * C16_D); it.th = c.take<_Float16>(N * C16_D); it.tl = c.take<_Float16>(N * C16_D);
  float* pmax = c.take<float>((int64_t)splits * M);
  float* psum = c.take<float>((int64_t)splits * M);
  float* pe = c.take<float>((int64_t)splits * M * C16_D);
  const int bs = pick_bwd_splits(M, N);
  float* part = c.take<float>(bs > 1 ? (int64_t)bs * N * C16_D : 0);
  if (w) { w->absmax = am; w->u = u; w->it = it; w->pmax = pmax; w->psum = psum; w->pe = pe; w->part = part; }
  return c.off;
}
int split_matrix(const float* X, int64_t ld, int64_t rows, unsigned* absmax, const Images& im, hipStream_t st) {
  const int64_t n4 = rows * (C16_D / 4);
  int blocks = (int)(n4 / 256 < 256 ? (n4 + 255) / 256 : 256);
  hipLaunchKernelGGL(ce16_absmax_kernel, dim3(blocks), dim3(256), 0, st, X, ld, rows, absmax);
  if (int rc = check_launch("ce16_absmax_kernel")) return rc;
  hipLaunchKernelGGL(ce16_split_kernel, dim3((unsigned)(rows / 16)), dim3(256), 0, st, X, ld, rows, absmax, im);
  return check_launch("ce16_split_kernel");
}
}  // namespace
}  // namespace tt

using namespace tt;

extern "C" int tt_ce16_supported(int64_t M, int64_t N, int64_t D) {
  return (D == C16_D && M > 0 && N > 0 && M % C16_ROWS_WG == 0 && N % 1024 == 0) ? 1 : 0;
}

extern "C" int64_t tt_ce16_workspace_bytes(int64_t M, int64_t N, int64_t D) {
  if (!tt_ce16_supported(M, N, D)) return 0;
  return carve(nullptr, M, N, nullptr);
}

extern "C" int tt_ce16_fwd_du_keep(const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M, int64_t N, int64_t D,
                                   int64_t diag_offset, float* row_lse, float* row_ce, float* du_unit, int64_t ld_du, float* logits,
                                   int64_t logits_bytes, void* ws, int64_t ws_bytes, tt_stream_t stream) {
  if (!U || !I || !row_lse || !row_ce || !du_unit || !ws) return fail_arg("tt_ce16_fwd_du_keep: null pointer");
  if (!tt_ce16_supported(M, N, D)) {
    set_error("tt_ce16_fwd_du_keep: needs D = 128, M %% 256 == 0, N %% 1024 == 0");
    return TT_E_UNSUPPORTED;
  }
  if (ldu < D || ldi < D || ld_du < D || (ldu | ldi | ld_du) % 4 || ((uintptr_t)U | (uintptr_t)I | (uintptr_t)du_unit) % 16)
    return fail_arg("tt_ce16_fwd_du_keep: rows must be 16-byte aligned");
  if (diag_offset < 0 || diag_offset + M > N) return fail_arg("tt_ce16_fwd_du_keep: diagonal outside the item range");
  if (logits && logits_bytes < M * N * (int64_t)sizeof(float)) return fail_arg("tt_ce16_fwd_du_keep: logits buffer");
  if (ws_bytes < tt_ce16_workspace_bytes(M, N, D)) {
    set_error("tt_ce16_fwd_du_keep: workspace");
    return TT_E_WORKSPACE;
  }
  hipStream_t st = S(stream);
  Ws w;
  carve(ws, M, N, &w);
  if (hipMemsetAsync(w.absmax, 0, 256, st) != hipSuccess) return check_launch("hipMemsetAsync");
  if (int rc = split_matrix(U, ldu, M, w.absmax, w.u, st)) return rc;
  if (int rc = split_matrix(I, ldi, N, w.absmax + 1, w.it, st)) return rc;
  FwdArgs a;
  a.u = w.u; a.it = w.it; a.absmax = w.absmax; a.M = M; a.N = N; a.logits = logits;
  const int splits = pick_splits(M, N);
  a.pmax = w.pmax; a.psum = w.psum; a.pe = w.pe; a.n_splits = splits;
  {
    ProfScope prof("ce_fwd_kernel", st);
    hipLaunchKernelGGL(ce16_fwd_kernel, dim3((unsigned)(M / C16_ROWS_WG), splits), dim3(64 * C16_NW), 0, st, a);
  }
  if (int rc = check_launch("ce16_fwd_kernel")) return rc;
  hipLaunchKernelGGL(ce16_merge_kernel, dim3((unsigned)ceil_div(M, 4)), dim3(256), 0, st, w.pmax, w.psum, w.pe, splits, M, w.absmax, U, ldu, I,
                     ldi, diag_offset, row_lse, row_ce, du_unit, ld_du);
  return check_launch("ce16_merge_kernel");
}

extern "C" int tt_ce16_bwd_kept(const float* U, int64_t ldu, int64_t M, int64_t N, int64_t D, int64_t diag_offset, const float* row_lse,
                                const float* coef, const float* logits, int64_t logits_bytes, float* dI, int64_t lddi, void* ws,
                                int64_t ws_bytes, tt_stream_t stream) {
  if (!U || !row_lse || !coef || !logits || !dI || !ws) return fail_arg("tt_ce16_bwd_kept: null pointer");
  if (!tt_ce16_supported(M, N, D)) {
    set_error("tt_ce16_bwd_kept: needs D = 128, M %% 256 == 0, N %% 1024 == 0");
    return TT_E_UNSUPPORTED;
  }
  if (ldu < D || lddi < D || (ldu | lddi) % 4 || ((uintptr_t)U | (uintptr_t)dI) % 16) return fail_arg("tt_ce16_bwd_kept: rows must be 16-byte aligned");
  if (logits_bytes < M * N * (int64_t)sizeof(float)) return fail_arg("tt_ce16_bwd_kept: logits buffer");
  if (ws_bytes < tt_ce16_workspace_bytes(M, N, D)) {
    set_error("tt_ce16_bwd_kept: workspace");
    return TT_E_WORKSPACE;
  }
  hipStream_t st = S(stream);
  Ws w;
  carve(ws, M, N, &w);
  // the user images are formed again (two small launches) instead of being trusted to have survived in a shared workspace
  if (hipMemsetAsync(w.absmax, 0, 256, st) != hipSuccess) return check_launch("hipMemsetAsync");
  if (int rc = split_matrix(U, ldu, M, w.absmax, w.u, st)) return rc;
  hipLaunchKernelGGL(ce16_absmax_vec_kernel, dim3((unsigned)(M / 256 < 64 ? ceil_div(M, 256) : 64)), dim3(256), 0, st, coef, M, w.absmax + 2);
  if (int rc = check_launch("ce16_absmax_vec_kernel")) return rc;
  BwdArgs a;
  a.u = w.u; a.logits = logits; a.row_lse = row_lse; a.coef = coef; a.M = M; a.N = N; a.diag_off = diag_offset; a.dI = dI; a.lddi = lddi;
  a.absmax = w.absmax;
  a.n_splits = pick_bwd_splits(M, N);
  a.part = w.part;
  {
    ProfScope prof("ce_bwd_kernel", st);
    if (a.n_splits > 1) hipLaunchKernelGGL(ce16_bwd_items_kernel<true>, dim3((unsigned)(N / C16_ROWS_WG), (unsigned)a.n_splits), dim3(64 * C16_NW), 0, st, a);
    else hipLaunchKernelGGL(ce16_bwd_items_kernel<false>, dim3((unsigned)(N / C16_ROWS_WG)), dim3(64 * C16_NW), 0, st, a);
  }
  if (int rc = check_launch("ce16_bwd_items_kernel")) return rc;
  if (a.n_splits > 1) {
    hipLaunchKernelGGL(ce16_bwd_reduce_kernel, dim3((unsigned)ceil_div(N * (C16_D / 4), 256)), dim3(256), 0, st, w.part, a.n_splits, N, dI, lddi);
    return check_launch("ce16_bwd_reduce_kernel");
  }
  return 0;
}

extern "C" int tt_ce16_bwd_recompute(const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M, int64_t N, int64_t D,
                                     int64_t diag_offset, const float* row_lse, const float* coef, float* dI, int64_t lddi, void* ws,
                                     int64_t ws_bytes, int reuse_images, tt_stream_t stream) {
  if (!U || !I || !row_lse || !coef || !dI || !ws) return fail_arg("tt_ce16_bwd_recompute: null pointer");
  if (!tt_ce16_supported(M, N, D)) {
    set_error("tt_ce16_bwd_recompute: needs D = 128, M %% 256 == 0, N %% 1024 == 0");
    return TT_E_UNSUPPORTED;
  }
  if (ldu < D || ldi < D || lddi < D || (ldu | ldi | lddi) % 4 || ((uintptr_t)U | (uintptr_t)I | (uintptr_t)dI) % 16)
    return fail_arg("tt_ce16_bwd_recompute: rows must be 16-byte aligned");
  if (ws_bytes < tt_ce16_workspace_bytes(M, N, D)) {
    set_error("tt_ce16_bwd_recompute: workspace");
    return TT_E_WORKSPACE;
  }
  hipStream_t st = S(stream);
  Ws w;
  carve(ws, M, N, &w);
  if (reuse_images) {  // the workspace still holds the forward's images and scales of THESE operands (the caller's promise)
    if (hipMemsetAsync(w.absmax + 2, 0, 4, st) != hipSuccess) return check_launch("hipMemsetAsync");
  } else {
    if (hipMemsetAsync(w.absmax, 0, 256, st) != hipSuccess) return check_launch("hipMemsetAsync");
    if (int rc = split_matrix(U, ldu, M, w.absmax, w.u, st)) return rc;
    if (int rc = split_matrix(I, ldi, N, w.absmax + 1, w.it, st)) return rc;
  }
  hipLaunchKernelGGL(ce16_absmax_vec_kernel, dim3((unsigned)(M / 256 < 64 ? ceil_div(M, 256) : 64)), dim3(256), 0, st, coef, M, w.absmax + 2);
  if (int rc = check_launch("ce16_absmax_vec_kernel")) return rc;
  BwdArgs a;
  a.u = w.u; a.it = w.it; a.logits = nullptr; a.row_lse = row_lse; a.coef = coef; a.M = M; a.N = N; a.diag_off = diag_offset; a.dI = dI; a.lddi = lddi;
  a.absmax = w.absmax;
  a.n_splits = pick_bwd_splits(M, N);
  a.part = w.part;
  {
    ProfScope prof("ce_bwd_kernel", st);
    if (a.n_splits > 1) hipLaunchKernelGGL(ce16_bwd_items_rc_kernel<true>, dim3((unsigned)(N / C16_ROWS_WG), (unsigned)a.n_splits), dim3(64 * C16_NW), 0, st, a);
    else hipLaunchKernelGGL(ce16_bwd_items_rc_kernel<false>, dim3((unsigned)(N / C16_ROWS_WG)), dim3(64 * C16_NW), 0, st, a);
  }
  if (int rc = check_launch("ce16_bwd_items_rc_kernel")) return rc;
  if (a.n_splits > 1) {
    hipLaunchKernelGGL(ce16_bwd_reduce_kernel, dim3((unsigned)ceil_div(N * (C16_D / 4), 256)), dim3(256), 0, st, w.part, a.n_splits, N, dI, lddi);
    return check_launch("ce16_bwd_reduce_kernel");
  }
  return 0;
}
