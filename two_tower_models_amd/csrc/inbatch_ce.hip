// K5: in-batch softmax cross entropy over S = U I^T without ever writing S.
//
// Decomposition (forward and both backward products share it):
//   workgroup = 4 wavefronts; each wave owns 32 "stationary" rows a (users for the
//   forward / dU, items for dI) and keeps them in REGISTERS as MFMA B-operand
//   fragments for the whole kernel.  The "streamed" rows b arrive in 64-row tiles
//   through double-buffered LDS (global -> registers -> LDS, one barrier per tile,
//   next tile's loads in flight under the current tile's MFMAs).
//   Per 32x32 sub-tile the wave computes the TRANSPOSED score tile
//       St[b][a] = sum_k Y[b][k] * X[a][k]        (v_mfma_f32_32x32x2_f32)
//   whose C/D register layout puts ONE stationary row a = lane&31 in each lane
//   (the 16 registers hold 16 streamed rows b).  Consequences:
//     * forward: the online-softmax state (running max, running sum) is two
//       registers per lane; no cross-lane traffic until the single merge at the end.
//     * backward: the gradient tile Gt[b][a] sits in exactly the A-operand layout of
//       the second product dX[a][:] += sum_b G[a][b] * Y[b][:], whose reduction
//       index b is permuted the same way on both operands -- the accumulator
//       registers are fed to the MFMA directly, no LDS round trip, no shuffles.
//   The streamed range is split over gridDim.y workgroups; partial softmax states
//   / partial dX slabs are merged by a second tiny kernel in fixed order
//   (deterministic, atomic-free).
//   All exponentials are base-2 on pre-scaled logits (v_exp_f32).
//
// Staging of the streamed tile, two forms (template GLDS):
//   GLDS = true  (D in {32,64,128}, 16-B aligned rows): LDS-DMA (global_load_lds_dwordx4), no
//     staging registers and no ds_write pass -> the D=128 kernels fit 2 waves per SIMD, so one
//     wave's softmax epilogue hides under the other's MFMAs.  The DMA image is lane-linear
//     (1 KiB per wave instruction), so the tile is UNPADDED [64][D] and bank conflicts are
//     removed by an XOR swizzle of the 16-B chunk index, applied to the per-lane SOURCE
//     address and again on every read:  chunk' = chunk ^ (row & min(D/4,16)-1).
//   GLDS = false: global -> registers -> LDS with zero padding (any D <= 128, any alignment),
//     row stride D+4 floats.
#include <stdlib.h>

#include "common.hpp"
#include "mfma_stream.hpp"

namespace tt {

struct CeArgs {
  const float* X;  // stationary [RX, D]
  const float* Y;  // streamed   [RY, D]
  int64_t ldx, ldy, RX, RY, D;
  int64_t diag_offset;      // positive of user i is item i + diag_offset
  int64_t tiles_per_split;  // streamed 64-row tiles per gridDim.y slice
  int x_vec, y_vec;
  // forward outputs (per split): running max (log2 domain), running sum, diagonal logit
  float* part_m;
  float* part_s;
  float* diag;
  // backward inputs / outputs
  const float* lse;   // row LSE in the log2 domain (as written by the forward), indexed by USER row
  const float* coef;  // dLoss/d row_ce, indexed by USER row
  float* out;         // [splits][RX][D] slabs (or final [RX, ldo] when splits == 1)
  int64_t ldo;
  int splits;
  // kept logits (log2 domain, masked): Z[user][item], row stride ldk, both extents padded to 128
  float* keep;
  int64_t ldk;
  unsigned* counter;  // fused loss epilogue: arrival counter of the finish kernel's blocks (zeroed by the product kernel)
};

// ------------------------------------------------------------------ forward
// Streams tile t+1 while tile t is on the MFMA pipe.  GLDS: DMA issued at the top of the
// iteration, drained (vmcnt(0)) right before the barrier that ends it.
template <int DP8, bool GLDS>
__global__ __launch_bounds__(256, ((GLDS || DP8 < 16) ? 2 : 1)) void ce_fwd_kernel(const CeArgs p) {
  using TM = TileMap<DP8, GLDS>;
  constexpr int TILE_FLOATS = BJ * TM::LD;
  // two NAMED tile buffers, loop unrolled by two (see ce_bwd_kept_kernel)
  __shared__ __attribute__((aligned(16))) float buf0[TILE_FLOATS];
  __shared__ __attribute__((aligned(16))) float buf1[TILE_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int64_t a = (int64_t)blockIdx.x * BI + wave * 32 + r;  // this lane's user row

  float xr[DP8][4];
  load_stationary<DP8>(xr, p.X, p.ldx, a, p.RX, p.D, h, p.x_vec);

  const int64_t ntiles_all = (p.RY + BJ - 1) / BJ;
  const int64_t t0 = (int64_t)blockIdx.y * p.tiles_per_split;
  const int64_t t1 = (t0 + p.tiles_per_split < ntiles_all) ? t0 + p.tiles_per_split : ntiles_all;

  float m = NEG_BIG, s = 0.f, dg = 0.f;
  bool has_dg = false;
  const int64_t want = a + p.diag_offset;

  Stager<DP8, GLDS> stg;
  if (t0 < t1) {
    stg.issue(p.Y, p.ldy, t0 * BJ, p.RY, p.D, p.y_vec, buf0, wave, lane);
    stg.land(buf0);
  }
  __syncthreads();
  auto step = [&](int64_t t, const float* ys, float* nxt) {
    if (t + 1 < t1) stg.issue(p.Y, p.ldy, (t + 1) * BJ, p.RY, p.D, p.y_vec, nxt, wave, lane);
    // tile-relative 32-bit indices with the lane term 4h folded in: row (li + 4h) of this tile
    // is the diagonal iff li == want4, and is a real item iff li < lim4
    const int64_t wrel = want - t * BJ, lrel = p.RY - t * BJ;
    const int want4 = (wrel >= 0 && wrel < BJ) ? (int)wrel - 4 * h : -1000;
    const int lim4 = (lrel < BJ ? (int)lrel : BJ) - 4 * h;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      const f32x16 acc = score_tile<DP8, GLDS>(ys, xr, jt, r, h);
      float v2[16];
      float tmax = NEG_BIG;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int li = jt * 32 + (e & 3) + 8 * (e >> 2);  // tile-local row, minus the 4h lane term
        const float s2 = acc[e] * LOG2E;  // logits are handled in the log2 domain throughout
        if (li == want4) { dg = s2; has_dg = true; }
        v2[e] = (li < lim4) ? s2 : NEG_BIG;
        tmax = fmaxf(tmax, v2[e]);
      }
      const float mn = fmaxf(m, tmax);
      float add = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) add += fast_exp2(v2[e] - mn);  // invalid -> exp2(-huge) = 0
      s = s * fast_exp2(m - mn) + add;
      m = mn;
    }
    if (t + 1 < t1) stg.land(nxt);
    __syncthreads();
  };
  for (int64_t t = t0; t < t1; t += 2) {
    step(t, buf0, buf1);
    if (t + 1 < t1) step(t + 1, buf1, buf0);
  }
  // merge the two lane halves (same row a, disjoint b subsets)
  const float mo = __shfl_xor(m, 32, 64), so = __shfl_xor(s, 32, 64), dgo = __shfl_xor(dg, 32, 64);
  const bool has_o = __shfl_xor((int)has_dg, 32, 64) != 0;
  const float M = fmaxf(m, mo);
  const float Ssum = s * fast_exp2(m - M) + so * fast_exp2(mo - M);
  if (h == 0 && a < p.RX) {
    p.part_m[(int64_t)blockIdx.y * p.RX + a] = M;
    p.part_s[(int64_t)blockIdx.y * p.RX + a] = Ssum;
    if (has_dg || has_o) p.diag[a] = has_dg ? dg : dgo;
  }
}

__global__ void ce_fwd_finish_kernel(const float* __restrict__ part_m, const float* __restrict__ part_s,
                                     const float* __restrict__ diag, int64_t M, int splits,
                                     float* __restrict__ row_lse, float* __restrict__ row_ce) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  float mx = NEG_BIG;
  for (int z = 0; z < splits; ++z) mx = fmaxf(mx, part_m[(int64_t)z * M + i]);
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += part_s[(int64_t)z * M + i] * exp2f(part_m[(int64_t)z * M + i] - mx);
  // row_lse stays in the log2 domain (it is only ever consumed by the backward kernels, which work
  // there too): no ln2 / log2e round trip, and a 1-column row gives exactly p = 1, ce = 0
  const float lse2 = mx + log2f(s);
  row_lse[i] = lse2;
  row_ce[i] = (lse2 - diag[i]) * LN2;
}

// ------------------------------------------------------------------ backward
// STREAM_STATS = false: stationary = users (stats per lane), streamed = items   -> dU
// STREAM_STATS = true : stationary = items, streamed = users (stats per b)      -> dI
template <int DP8, bool STREAM_STATS, bool GLDS>
__global__ __launch_bounds__(256, ((GLDS || DP8 < 16) ? 2 : 1)) void ce_bwd_kernel(const CeArgs p) {
  using TM = TileMap<DP8, GLDS>;
  constexpr int TD = (DP8 + 3) / 4;  // 32-column tiles of the output
  constexpr int TILE_FLOATS = BJ * TM::LD + (STREAM_STATS ? 2 * BJ : 0);
  // two NAMED tile buffers (see ce_bwd_kept_kernel): LDS reads of one do not wait for the DMA into the other
  __shared__ __attribute__((aligned(16))) float buf0[TILE_FLOATS];
  __shared__ __attribute__((aligned(16))) float buf1[TILE_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int64_t a = (int64_t)blockIdx.x * BI + wave * 32 + r;

  float xr[DP8][4];
  load_stationary<DP8>(xr, p.X, p.ldx, a, p.RX, p.D, h, p.x_vec);

  float lse2_a = 0.f, coef_a = 0.f;
  if (!STREAM_STATS && a < p.RX) { lse2_a = p.lse[a]; coef_a = p.coef[a]; }

  const int64_t ntiles_all = (p.RY + BJ - 1) / BJ;
  const int64_t t0 = (int64_t)blockIdx.y * p.tiles_per_split;
  const int64_t t1 = (t0 + p.tiles_per_split < ntiles_all) ? t0 + p.tiles_per_split : ntiles_all;

  f32x16 dacc[TD];
#pragma unroll
  for (int d = 0; d < TD; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) dacc[d][e] = 0.f;

  // per-lane float offsets of the second product's B operand (see the loop below)
  int ybase[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    ybase[q] = GLDS ? 4 * h * TM::LD + 4 * ((((r >> 2) ^ (4 * h)) & TM::SW) ^ q) + (r & 3) : 4 * h * TM::LD + r;

  Stager<DP8, GLDS> stg;
  float st_lse = 0.f, st_coef = 0.f;  // threads 0..63 stage the streamed rows' stats
  auto issue = [&](int64_t t, float* dst) {
    stg.issue(p.Y, p.ldy, t * BJ, p.RY, p.D, p.y_vec, dst, wave, lane);
    if (STREAM_STATS && threadIdx.x < BJ) {
      const int64_t b = t * BJ + threadIdx.x;
      st_lse = (b < p.RY) ? p.lse[b] : 3.0e38f;
      st_coef = (b < p.RY) ? p.coef[b] : 0.f;
    }
  };
  auto land = [&](float* dst) {
    if (STREAM_STATS && threadIdx.x < BJ) {
      dst[BJ * TM::LD + threadIdx.x] = st_lse;
      dst[BJ * TM::LD + BJ + threadIdx.x] = st_coef;
    }
    stg.land(dst);
  };

  if (t0 < t1) { issue(t0, buf0); land(buf0); }
  __syncthreads();
  auto step = [&](int64_t t, const float* ys, float* nxt) {
    if (t + 1 < t1) issue(t + 1, nxt);
    // tile-relative 32-bit indices, lane term 4h folded in (see ce_fwd_kernel).  The streamed
    // row that pairs with stationary row a is  a + diag_offset (dU) / a - diag_offset (dI).
    const int64_t wrel = (STREAM_STATS ? a - p.diag_offset : a + p.diag_offset) - t * BJ, lrel = p.RY - t * BJ;
    const int want4 = (wrel >= 0 && wrel < BJ) ? (int)wrel - 4 * h : -1000;
    const int lim4 = (lrel < BJ ? (int)lrel : BJ) - 4 * h;
    // VALU instructions are paid in MFMA issue time (they do not co-execute): the diagonal test is
    // kept out of the tiles that do not hold this wave's diagonal (scalar ballot -> uniform branch)
    const bool no_diag = STREAM_STATS && __builtin_amdgcn_ballot_w64(want4 != -1000) == 0ull;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      const f32x16 acc = score_tile<DP8, GLDS>(ys, xr, jt, r, h);
      float gt[16];
      if constexpr (!STREAM_STATS) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int li = jt * 32 + (e & 3) + 8 * (e >> 2);
          const float pr = fast_exp2(mul_rounded(acc[e], LOG2E) - lse2_a);  // same rounded s2 as the forward
          const float gval = coef_a * (pr - ((li == want4) ? 1.f : 0.f));
          gt[e] = (li < lim4) ? gval : 0.f;
        }
      } else if (no_diag) {  // no diagonal in this tile for any lane of the wave: 4 instead of 7 VALU ops per element
        const float* sl = ys + BJ * TM::LD + jt * 32 + 4 * h;
        const float* sc = sl + BJ;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 l4 = *reinterpret_cast<const float4*>(sl + 8 * q);
          const float4 c4 = *reinterpret_cast<const float4*>(sc + 8 * q);
          gt[4 * q + 0] = c4.x * fast_exp2(mul_rounded(acc[4 * q + 0], LOG2E) - l4.x);
          gt[4 * q + 1] = c4.y * fast_exp2(mul_rounded(acc[4 * q + 1], LOG2E) - l4.y);
          gt[4 * q + 2] = c4.z * fast_exp2(mul_rounded(acc[4 * q + 2], LOG2E) - l4.z);
          gt[4 * q + 3] = c4.w * fast_exp2(mul_rounded(acc[4 * q + 3], LOG2E) - l4.w);
        }
      } else {
        const float* sl = ys + BJ * TM::LD + jt * 32 + 4 * h;
        const float* sc = sl + BJ;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 l4 = *reinterpret_cast<const float4*>(sl + 8 * q);
          const float4 c4 = *reinterpret_cast<const float4*>(sc + 8 * q);
          const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int e = 4 * q + c;
            const int li = jt * 32 + (e & 3) + 8 * (e >> 2);
            const float pr = fast_exp2(mul_rounded(acc[e], LOG2E) - lv[c]);
            gt[e] = cv[c] * (pr - ((li == want4) ? 1.f : 0.f));  // coef 0 beyond RY
          }
        }
      }
      // dX[a][d] += sum_b G[a][b] * Y[b][d]; reduction index b = brow(e, h) on both operands.
      // B-operand element Y[b][32d + r].  In the swizzled image its chunk is
      //   (8d + r/4) ^ (b & SW) = 8*(d ^ hi(e)) | ((r/4 ^ 4h) ^ (e & 3)),
      // i.e. one of FOUR per-lane base offsets (ybase[e&3]) plus a compile-time immediate.
      // Reads run one step ahead of the MFMAs that consume them (register double buffer);
      // the scheduling barriers keep the compiler from hoisting all 64 reads at once.
      float yv[2][TD];
      auto yread = [&](int e, float (&dst)[TD]) {
        const int E = (e & 3) + 8 * (e >> 2);
        const int hi = (GLDS && TM::SW >= 8) ? ((e >> 2) & 1) : 0;
#pragma unroll
        for (int d = 0; d < TD; ++d) dst[d] = ys[(jt * 32 + E) * TM::LD + 32 * (d ^ hi) + ybase[e & 3]];
      };
      yread(0, yv[0]);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (e + 1 < 16) yread(e + 1, yv[(e + 1) & 1]);
#pragma unroll
        for (int d = 0; d < TD; ++d)
          dacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(gt[e], yv[e & 1][d], dacc[d], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (t + 1 < t1) land(nxt);
    __syncthreads();
  };
  for (int64_t t = t0; t < t1; t += 2) {
    step(t, buf0, buf1);
    if (t + 1 < t1) step(t + 1, buf1, buf0);
  }

  float* out = p.out + (p.splits > 1 ? (int64_t)blockIdx.y * p.RX * p.D : 0);
  const int64_t ldo = p.splits > 1 ? p.D : p.ldo;
  const int64_t abase = (int64_t)blockIdx.x * BI + wave * 32;
#pragma unroll
  for (int d = 0; d < TD; ++d) {
    const int64_t col = 32 * d + r;
    if (col >= p.D) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t row = abase + brow(e, h);
      if (row < p.RX) out[row * ldo + col] = dacc[d][e];
    }
  }
}

// ------------------------------------------------------------------ item-side backward from kept logits
// dI[a][:] = sum_b G[b][a] * U[b][:] with G rebuilt from the logits the forward kept (KEEP above)
// instead of from a second U I^T product: half the MFMA work of ce_bwd_kernel<.., true, ..>, paid for
// with one read of Z (M x N x 4 B) that streams underneath.  Lane = item a, register e = user
// brow(e, h) of the sub-tile, exactly the layout the recomputed score tile would have; for a fixed
// user the 32 lanes of a half-wave read 128 consecutive bytes of its row.  The 32 values of the NEXT
// tile are requested before this tile's MFMAs and land with the next tile's LDS-DMA.
template <int DP8>
__global__ __launch_bounds__(256, 2) void ce_bwd_kept_kernel(const CeArgs p) {
  using TM = TileMap<DP8, true>;
  constexpr int TD = (DP8 + 3) / 4;
  constexpr int TILE_FLOATS = BJ * TM::LD + 2 * BJ;
  // TWO named LDS arrays (not one dynamic block): the compiler tags accesses to distinct LDS variables
  // with alias scopes, and only then does it let a ds_read of one buffer proceed while the LDS-DMA
  // into the OTHER is in flight -- with a single block every LDS read after a DMA issue waits vmcnt(0)
  __shared__ __attribute__((aligned(16))) float buf0[TILE_FLOATS];
  __shared__ __attribute__((aligned(16))) float buf1[TILE_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int64_t a = (int64_t)blockIdx.x * BI + wave * 32 + r;  // item; columns of Z are padded to 128

  const int64_t ntiles_all = (p.RY + BJ - 1) / BJ;
  const int64_t t0 = (int64_t)blockIdx.y * p.tiles_per_split;
  const int64_t t1 = (t0 + p.tiles_per_split < ntiles_all) ? t0 + p.tiles_per_split : ntiles_all;

  f32x16 dacc[TD];
#pragma unroll
  for (int d = 0; d < TD; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) dacc[d][e] = 0.f;
  int ybase[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) ybase[q] = 4 * h * TM::LD + 4 * ((((r >> 2) ^ (4 * h)) & TM::SW) ^ q) + (r & 3);

  // Z[user][item]: uniform row base (scalar registers) + one 32-bit per-lane offset
  const float* const zblock = p.keep + (int64_t)blockIdx.x * BI;
  const int zlane = wave * 32 + r + 4 * h * (int)p.ldk;
  float zn[32];
  auto zload = [&](int64_t t) {
    // the row stride is made opaque here so that the 32 row offsets are recomputed with a scalar
    // multiply per tile: hoisted out of the loop they occupy 64 scalar registers, which spill into
    // VGPR lanes and come back through v_readlane in front of every load
    int ldk = (int)p.ldk;
    asm volatile("" : "+s"(ldk));
    // buffer loads: tile base in a scalar resource descriptor, row offset in a scalar register, ONE
    // 32-bit per-lane offset -- global_load would carry a 64-bit VGPR address per load (two VALU
    // adds each, 64 per tile)
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(zblock + t * BJ * (int64_t)ldk), 0, -1, 0x00020000);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int row = (i >> 4) * 32 + (i & 3) + 8 * ((i & 15) >> 2);
      zn[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, zlane * 4, row * ldk * 4, 0));
    }
  };
  float st_lse = 0.f, st_coef = 0.f;  // threads 0..63 stage the streamed rows' stats (raw: masked when landed)
  bool st_ok = false;
  auto issue = [&](int64_t t, float* dst) {
    tile_dma<DP8>(p.Y, p.ldy, t * BJ, p.RY, dst, wave, lane);
    if (threadIdx.x < BJ) {
      const int64_t b = t * BJ + threadIdx.x;
      const int64_t bc = b < p.RY ? b : p.RY - 1;
      st_lse = p.lse[bc];
      st_coef = p.coef[bc];
      st_ok = b < p.RY;
    }
  };
  auto land = [&](float* dst) {
    if (threadIdx.x < BJ) {
      dst[BJ * TM::LD + threadIdx.x] = st_ok ? st_lse : 3.0e38f;
      dst[BJ * TM::LD + BJ + threadIdx.x] = st_ok ? st_coef : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // one tile: `ys` is read, `nxt` receives tile t + 1
  auto step = [&](int64_t t, const float* ys, float* nxt) {
    float zc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) zc[i] = zn[i];
    if (t + 1 < t1) { issue(t + 1, nxt); zload(t + 1); }
    const int64_t wrel = a - p.diag_offset - t * BJ;
    const int want4 = (wrel >= 0 && wrel < BJ) ? (int)wrel - 4 * h : -1000;
    // MFMA and VALU instructions do not co-execute on a SIMD (SQ_VALU_MFMA_COEXEC_CYCLES = 0): every
    // VALU instruction here is paid in MFMA issue time.  Only the one or two tiles that hold this
    // wave's diagonal need the per-element test (wave-uniform branch).
    const bool no_diag = __builtin_amdgcn_ballot_w64(want4 != -1000) == 0ull;  // scalar: a uniform branch
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      float gt[16];
      const float* sl = ys + BJ * TM::LD + jt * 32 + 4 * h;
      const float* sc = sl + BJ;
      if (no_diag) {  // three VALU instructions per element instead of six
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 l4 = *reinterpret_cast<const float4*>(sl + 8 * q);
          const float4 c4 = *reinterpret_cast<const float4*>(sc + 8 * q);
          gt[4 * q + 0] = c4.x * fast_exp2(zc[jt * 16 + 4 * q + 0] - l4.x);
          gt[4 * q + 1] = c4.y * fast_exp2(zc[jt * 16 + 4 * q + 1] - l4.y);
          gt[4 * q + 2] = c4.z * fast_exp2(zc[jt * 16 + 4 * q + 2] - l4.z);
          gt[4 * q + 3] = c4.w * fast_exp2(zc[jt * 16 + 4 * q + 3] - l4.w);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 l4 = *reinterpret_cast<const float4*>(sl + 8 * q);
          const float4 c4 = *reinterpret_cast<const float4*>(sc + 8 * q);
          const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int e = 4 * q + c;
            const int li = jt * 32 + (e & 3) + 8 * (e >> 2);
            const float pr = fast_exp2(zc[jt * 16 + e] - lv[c]);
            gt[e] = cv[c] * (pr - ((li == want4) ? 1.f : 0.f));  // coef 0 beyond RY
          }
        }
      }
      float yv[2][TD];
      auto yread = [&](int e, float (&dst)[TD]) {
        const int E = (e & 3) + 8 * (e >> 2);
        const int hi = (TM::SW >= 8) ? ((e >> 2) & 1) : 0;
#pragma unroll
        for (int d = 0; d < TD; ++d) dst[d] = ys[(jt * 32 + E) * TM::LD + 32 * (d ^ hi) + ybase[e & 3]];
      };
      yread(0, yv[0]);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (e + 1 < 16) yread(e + 1, yv[(e + 1) & 1]);
#pragma unroll
        for (int d = 0; d < TD; ++d)
          dacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(gt[e], yv[e & 1][d], dacc[d], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (t + 1 < t1) land(nxt);
    __syncthreads();
  };

  if (t0 < t1) { zload(t0); issue(t0, buf0); land(buf0); }
  __syncthreads();
  for (int64_t t = t0; t < t1; t += 2) {
    step(t, buf0, buf1);
    if (t + 1 < t1) step(t + 1, buf1, buf0);
  }

  float* out = p.out + (p.splits > 1 ? (int64_t)blockIdx.y * p.RX * p.D : 0);
  const int64_t ldo = p.splits > 1 ? p.D : p.ldo;
  const int64_t abase = (int64_t)blockIdx.x * BI + wave * 32;
#pragma unroll
  for (int d = 0; d < TD; ++d) {
    const int64_t col = 32 * d + r;
    if (col >= p.D) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t row = abase + brow(e, h);
      if (row < p.RX) out[row * ldo + col] = dacc[d][e];
    }
  }
}

// ------------------------------------------------------------------ forward fused with dU
// Forward that also returns E[a][:] = sum_b softmax(S)[a][b] * Y[b][:], the expected item embedding
// of every user -- from which dU[a] = dLoss/dce[a] * (E[a] - Y[a + diag_offset]) is an elementwise
// step, so the backward needs no dU kernel and its recomputation of the logits (5 -> 4 logit-sized
// products per training step; at 8 GPUs with global negatives the logits ARE the step).
// It is the dU kernel above with the online-softmax state of the forward: probabilities are formed
// against the RUNNING maximum, and whenever that moves the accumulated rows are rescaled.  The
// accumulator tile has one output column per lane and 16 USER rows in the registers, so the 32
// per-user factors of a wave travel through 128 B of LDS (written by the lane that owns the user,
// read as four float4 by everybody).  Both lane halves of a user share one running maximum (one
// cross-half exchange per sub-tile), because their probabilities are summed by the same MFMA.
// KEEP: the masked log2-domain logits are also written to p.keep (16 B per lane and 4-row group), so
// the item-side backward (ce_bwd_kept_kernel) reads them back instead of recomputing the product.
template <int DP8, bool GLDS, bool KEEP = false>
__global__ __launch_bounds__(256, ((GLDS || DP8 < 16) ? 2 : 1)) void ce_fwd_du_kernel(const CeArgs p) {
  using TM = TileMap<DP8, GLDS>;
  constexpr int TD = (DP8 + 3) / 4;
  constexpr int TILE_FLOATS = BJ * TM::LD;
  // two NAMED tile buffers (see ce_bwd_kept_kernel): LDS reads of one no longer wait for the DMA into the other
  __shared__ __attribute__((aligned(16))) float buf0[TILE_FLOATS];
  __shared__ __attribute__((aligned(16))) float buf1[TILE_FLOATS];
  __shared__ __attribute__((aligned(16))) float fbuf_all[4 * 32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int64_t a = (int64_t)blockIdx.x * BI + wave * 32 + r;
  float* const fbuf = fbuf_all + wave * 32;  // this wave's 32 rescale factors

  float xr[DP8][4];
  load_stationary<DP8>(xr, p.X, p.ldx, a, p.RX, p.D, h, p.x_vec);
  if (p.counter && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *p.counter = 0u;  // see ce_fwd_du_finish_loss_kernel

  const int64_t ntiles_all = (p.RY + BJ - 1) / BJ;
  const int64_t t0 = (int64_t)blockIdx.y * p.tiles_per_split;
  const int64_t t1 = (t0 + p.tiles_per_split < ntiles_all) ? t0 + p.tiles_per_split : ntiles_all;

  f32x16 dacc[TD];
#pragma unroll
  for (int d = 0; d < TD; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) dacc[d][e] = 0.f;
  // second product's B operand offsets (see ce_bwd_kernel): base + 4 * (swz ^ q), q = e & 3
  const int ybase0 = GLDS ? 4 * h * TM::LD + (r & 3) : 4 * h * TM::LD + r;
  const int yswz = GLDS ? (((r >> 2) ^ (4 * h)) & TM::SW) : 0;

  float m = NEG_BIG, s = 0.f;
  const int64_t want = (a < p.RX) ? a + p.diag_offset : -1;  // rows past the end own no diagonal
  const int zlane = KEEP ? ((wave * 32 + r) * (int)p.ldk + 4 * h) * 4 : 0;  // byte offset of this lane's logits row

  Stager<DP8, GLDS> stg;
  if (t0 < t1) {
    stg.issue(p.Y, p.ldy, t0 * BJ, p.RY, p.D, p.y_vec, buf0, wave, lane);
    stg.land(buf0);
  }
  __syncthreads();
  auto step = [&](int64_t t, const float* ys, float* nxt) {
    if (t + 1 < t1) stg.issue(p.Y, p.ldy, (t + 1) * BJ, p.RY, p.D, p.y_vec, nxt, wave, lane);
    const int64_t wrel = want - t * BJ, lrel = p.RY - t * BJ;
    const int want4 = (wrel >= 0 && wrel < BJ) ? (int)wrel - 4 * h : -1000;
    const int lim4 = (lrel < BJ ? (int)lrel : BJ) - 4 * h;
    // MFMA and VALU instructions do not co-execute on a SIMD (SQ_VALU_MFMA_COEXEC_CYCLES = 0 for every
    // kernel of this file): each VALU instruction costs its 4+ cycles of MFMA issue.  The diagonal
    // test and the end-of-range mask are therefore taken out of the per-element code of the tiles
    // that need neither (wave-uniform branch): all but the last tile, and all but the one or two
    // tiles that hold this wave's diagonal.
    // (not in the KEEP form: there the second code path costs more in spilled registers than it saves)
    const bool plain = !KEEP && lrel >= BJ && __builtin_amdgcn_ballot_w64(want4 != -1000) == 0ull;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      // v2 lives in the score tile's own registers from here on (masked logits, then probabilities):
      // the kernel sits at the 256-VGPR limit of 2 waves per SIMD, and a spill reload waits on
      // vmcnt -- the counter the next tile's LDS-DMA completes on
      f32x16 v2 = score_tile<DP8, GLDS>(ys, xr, jt, r, h);
      float tmax = NEG_BIG;
      if (plain) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          v2[e] = v2[e] * LOG2E;
          tmax = fmaxf(tmax, v2[e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int li = jt * 32 + (e & 3) + 8 * (e >> 2);
          const float s2 = v2[e] * LOG2E;
          if (li == want4) p.diag[a] = s2;  // executes for exactly one (lane, e) per user row
          v2[e] = (li < lim4) ? s2 : NEG_BIG;
          tmax = fmaxf(tmax, v2[e]);
        }
      }
      if constexpr (KEEP) {  // rows past RX / columns past RY of the padded buffer get 0-logit / NEG_BIG filler
        // buffer stores: tile base in a scalar descriptor, one loop-invariant 32-bit lane offset
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t zs = __builtin_amdgcn_make_buffer_rsrc(
            p.keep + (int64_t)blockIdx.x * BI * p.ldk + t * BJ, 0, -1, 0x00020000);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 o = {v2[4 * g], v2[4 * g + 1], v2[4 * g + 2], v2[4 * g + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o), zs, zlane,
                                                 (jt * 32 + 8 * g) * 4, 0);
        }
      }
      float mn = fmaxf(m, tmax);
      mn = fmaxf(mn, __shfl_xor(mn, 32, 64));  // one reference per USER: both lane halves feed the same MFMA
      const float f = fast_exp2(m - mn);       // m is identical in both halves by construction
      float add = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        v2[e] = fast_exp2(v2[e] - mn);  // in place: v2 now holds the probabilities (rows past the end: 0)
        add += v2[e];
      }
      s = s * f + add;
      m = mn;
      // rescale the accumulated rows: register e of dacc is user row brow(e, h) of this wave.
      // A running maximum moves O(log n) times per row, so after the first tiles most sub-tiles
      // find every factor equal to 1 and skip the exchange (wave-uniform branch).
      if (__any(f != 1.0f)) {
        if (h == 0) fbuf[r] = f;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 f4 = *reinterpret_cast<const float4*>(fbuf + 8 * q + 4 * h);
#pragma unroll
          for (int d = 0; d < TD; ++d) {
            dacc[d][4 * q] *= f4.x; dacc[d][4 * q + 1] *= f4.y; dacc[d][4 * q + 2] *= f4.z; dacc[d][4 * q + 3] *= f4.w;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      // O[a][d] += sum_b p[a][b] * Y[b][d]  (the dU kernel's second product, same operand addressing)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int E = (e & 3) + 8 * (e >> 2);
        const int hi = (GLDS && TM::SW >= 8) ? ((e >> 2) & 1) : 0;
        float yv[TD];
#pragma unroll
        for (int d = 0; d < TD; ++d) yv[d] = ys[(jt * 32 + E) * TM::LD + 32 * (d ^ hi) + ybase0 + (GLDS ? 4 * (yswz ^ (e & 3)) : 0)];
#pragma unroll
        for (int d = 0; d < TD; ++d)
          dacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2[e], yv[d], dacc[d], 0, 0, 0);
      }
    }
    if (t + 1 < t1) stg.land(nxt);
    __syncthreads();
  };
  for (int64_t t = t0; t < t1; t += 2) {
    step(t, buf0, buf1);
    if (t + 1 < t1) step(t + 1, buf1, buf0);
  }
  // the two lane halves hold the same m and disjoint parts of the sum
  const float so = __shfl_xor(s, 32, 64);
  if (h == 0 && a < p.RX) {
    p.part_m[(int64_t)blockIdx.y * p.RX + a] = m;
    p.part_s[(int64_t)blockIdx.y * p.RX + a] = s + so;
  }
  float* out = p.out + (int64_t)blockIdx.y * p.RX * p.D;  // unnormalised, relative to 2^m of this split
  const int64_t abase = (int64_t)blockIdx.x * BI + wave * 32;
#pragma unroll
  for (int d = 0; d < TD; ++d) {
    const int64_t col = 32 * d + r;
    if (col >= p.D) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t row = abase + brow(e, h);
      if (row < p.RX) out[row * p.D + col] = dacc[d][e];
    }
  }
}

// merge the splits: row statistics as ce_fwd_finish_kernel, E = sum_z O_z 2^(m_z - M) / S, and
// du_unit[a] = E[a] - Y[a + diag_offset]  (dU[a] = dLoss/dce[a] * du_unit[a])
template <int ROWS_PER_BLOCK, bool AGENT_CE = false>
__device__ __forceinline__ void ce_fwd_du_finish_row(const float* __restrict__ part_m, const float* __restrict__ part_s,
                                                     const float* __restrict__ diag, const float* __restrict__ slabs, int splits,
                                                     int64_t M, int64_t D, const float* __restrict__ Y, int64_t ldy,
                                                     int64_t diag_offset, float* __restrict__ row_lse, float* __restrict__ row_ce,
                                                     float* __restrict__ du_unit, int64_t ld_du) {
  // one wavefront per user row; lane z holds split z's statistics (splits <= 64), so the maximum, the
  // normaliser and the per-split factors are wave reductions / shuffles instead of per-lane loops,
  // and the slab reads of a column are issued together (they were one dependent round trip each)
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  // the per-split factors and weighted sums go through LDS: the column loop below is divergent for
  // D < 64, where a shuffle could read from an inactive lane
  __shared__ float s_f[ROWS_PER_BLOCK][64], s_w[ROWS_PER_BLOCK][64];
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  const bool has = lane < splits;
  const float pm = has ? part_m[(int64_t)lane * M + row] : NEG_BIG;
  const float ps = has ? part_s[(int64_t)lane * M + row] : 0.f;
  const float mx = wave_max(pm);
  const int wv = threadIdx.x >> 6;
  const float fz = has ? exp2f(pm - mx) : 0.f;  // this split's rescale factor
  s_f[wv][lane] = fz;
  s_w[wv][lane] = ps * fz;
  __builtin_amdgcn_wave_barrier();
  // S in split order: the same sequence of additions as a serial loop over z
  float S = 0.f;
  for (int z = 0; z < splits; ++z) S += s_w[wv][z];
  const float inv = 1.0f / S;
  for (int64_t col = lane; col < D; col += 64) {
    float e = 0.f;
    int z = 0;
    for (; z + 8 <= splits; z += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = slabs[((int64_t)(z + u) * M + row) * D + col];
#pragma unroll
      for (int u = 0; u < 8; ++u) e += v[u] * s_f[wv][z + u];
    }
    for (; z < splits; ++z) e += slabs[((int64_t)z * M + row) * D + col] * s_f[wv][z];
    du_unit[row * ld_du + col] = e * inv - Y[(row + diag_offset) * ldy + col];
  }
  if (lane == 0) {
    const float lse2 = mx + log2f(S);
    row_lse[row] = lse2;
    const float cev = (lse2 - diag[row]) * LN2;
    // AGENT_CE: written through to the device-wide coherence point (another XCD's workgroup reads it in this launch)
    if constexpr (AGENT_CE) __hip_atomic_store(row_ce + row, cev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else row_ce[row] = cev;
  }
}

__global__ __launch_bounds__(256) void ce_fwd_du_finish_kernel(const float* __restrict__ part_m,
                                                               const float* __restrict__ part_s,
                                                               const float* __restrict__ diag,
                                                               const float* __restrict__ slabs, int splits, int64_t M,
                                                               int64_t D, const float* __restrict__ Y, int64_t ldy,
                                                               int64_t diag_offset, float* __restrict__ row_lse,
                                                               float* __restrict__ row_ce, float* __restrict__ du_unit,
                                                               int64_t ld_du) {
  ce_fwd_du_finish_row<4>(part_m, part_s, diag, slabs, splits, M, D, Y, ldy, diag_offset, row_lse, row_ce, du_unit, ld_du);
}

__global__ void slab_reduce_kernel(const float* __restrict__ slabs, int splits, int64_t rows, int64_t D,
                                   float* __restrict__ out, int64_t ldo) {
  const int64_t total = rows * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += slabs[(int64_t)z * total + i];
    out[(i / D) * ldo + (i % D)] = v;
  }
}

// the same sums, four columns per lane (16-byte loads and stores; D % 4 == 0, 16-byte aligned rows): the scalar form moved
// 100 MB in 66 us at the 8-GPU shape (2 slabs of [65 536, 128]) -- 1.5 TB/s, a fifth of what the copy reaches
__global__ __launch_bounds__(256) void slab_reduce4_kernel(const float4* __restrict__ slabs, int splits, int64_t rows, int64_t D4,
                                                           float4* __restrict__ out, int64_t ldo4) {
  const int64_t total = rows * D4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = slabs[i];
    for (int z = 1; z < splits; ++z) {
      const float4 w = slabs[(int64_t)z * total + i];
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    out[(i / D4) * ldo4 + (i % D4)] = v;
  }
}
static int launch_slab_reduce(const float* slabs, int splits, int64_t rows, int64_t D, float* out, int64_t ldo, hipStream_t st) {
  if (D % 4 == 0 && ldo % 4 == 0 && ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    const int64_t blocks = ceil_div(rows * (D / 4), 256);
    slab_reduce4_kernel<<<(unsigned)(blocks < 8192 ? blocks : 8192), 256, 0, st>>>(reinterpret_cast<const float4*>(slabs), splits, rows, D / 4,
                                                                                 reinterpret_cast<float4*>(out), ldo / 4);
  } else {
    slab_reduce_kernel<<<(unsigned)(ceil_div(rows * D, 256) < 2048 ? ceil_div(rows * D, 256) : 2048), 256, 0, st>>>(slabs, splits, rows, D, out, ldo);
  }
  return check_launch("slab_reduce_kernel");
}

// out[i, :] = x[i, :] * coef[i]  (dU = dL/dce (.) du_unit, the chain-rule step behind tt_inbatch_ce_fwd_du)
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ coef,
                                                         int64_t rows, int64_t D, float* __restrict__ out, int64_t ldo) {
  const int64_t total = rows * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D, c = i - r * D;
    out[r * ldo + c] = x[r * ldx + c] * coef[r];
  }
}

// ref:src/two_tower_base_retrieval.py:322,334-343 for [B,T] labels, one workgroup of 1024 threads.
template <bool AGENT_CE = false>
__device__ __forceinline__ void weighted_mean_loss_block(const float* __restrict__ labels, int64_t B, int64_t T,
                                                         const float* __restrict__ uvw, float* row_ce,
                                                         float* __restrict__ w_out, float* __restrict__ coef_out,
                                                         float* __restrict__ loss_out) {
  __shared__ float red[16];
  __shared__ float bcast;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mx = NEG_BIG;
  for (int64_t i = threadIdx.x; i < B; i += blockDim.x) {
    float nuv = 1.f;
    if (labels) {
      nuv = 0.f;
      for (int64_t t = 0; t < T; ++t) nuv += labels[i * T + t] * uvw[t];
      nuv = fmaxf(nuv, 0.000001f);
    }
    w_out[i] = nuv;
    mx = fmaxf(mx, nuv);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = red[0];
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k) v = fmaxf(v, red[k]);
    bcast = v;
  }
  __syncthreads();
  const float wmax = bcast;
  float acc = 0.f;
  const float invB = 1.0f / (float)B;
  for (int64_t i = threadIdx.x; i < B; i += blockDim.x) {
    const float w = w_out[i] / wmax;
    w_out[i] = w;
    coef_out[i] = w * invB;
    const float cev = AGENT_CE ? __hip_atomic_load(row_ce + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : row_ce[i];
    acc += cev * w;
  }
  acc = wave_sum(acc);
  __syncthreads();
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) v += red[k];
    *loss_out = v * invB;
  }
}

// The same head split around the two scalar exchanges of the row-sharded step (every rank holds B of the W*B rows):
//   value_weights_kernel        nuv[i] = clamp(sum_t labels[i,t] uvw[t], 1e-6), *max_out = max_i nuv   -> all-reduce MAX
//   weighted_loss_global_kernel w = nuv / *gmax, coef = w / denom, *loss_out = sum_i ce[i] w[i] / denom -> all-reduce SUM
// (denom = W*B).  One 1024-thread workgroup each; they replace nine B-sized torch launches per step.
__global__ __launch_bounds__(1024) void value_weights_kernel(const float* __restrict__ labels, int64_t B, int64_t T,
                                                             const float* __restrict__ uvw, float* __restrict__ nuv_out,
                                                             float* __restrict__ max_out) {
  __shared__ float red[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mx = NEG_BIG;
  for (int64_t i = threadIdx.x; i < B; i += blockDim.x) {
    float nuv = 0.f;
    for (int64_t t = 0; t < T; ++t) nuv += labels[i * T + t] * uvw[t];
    nuv = fmaxf(nuv, 0.000001f);
    nuv_out[i] = nuv;
    mx = fmaxf(mx, nuv);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = red[0];
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k) v = fmaxf(v, red[k]);
    *max_out = v;
  }
}
__global__ __launch_bounds__(1024) void weighted_loss_global_kernel(const float* __restrict__ nuv, const float* __restrict__ gmax,
                                                                    const float* __restrict__ row_ce, int64_t B, float denom,
                                                                    float* __restrict__ coef_out, float* __restrict__ loss_out) {
  __shared__ float red[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float wmax = *gmax;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < B; i += blockDim.x) {
    const float w = nuv[i] / wmax;
    coef_out[i] = w / denom;
    acc += row_ce[i] * w;
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) v += red[k];
    *loss_out = v / denom;
  }
}

__global__ __launch_bounds__(1024) void weighted_mean_loss_kernel(const float* __restrict__ labels, int64_t B,
                                                                  int64_t T, const float* __restrict__ uvw,
                                                                  const float* __restrict__ row_ce,
                                                                  float* __restrict__ w_out,
                                                                  float* __restrict__ coef_out,
                                                                  float* __restrict__ loss_out) {
  weighted_mean_loss_block(labels, B, T, uvw, const_cast<float*>(row_ce), w_out, coef_out, loss_out);
}

// ce_fwd_du_finish_kernel + weighted_mean_loss_kernel in one launch (16 rows per 1024-thread block): the block that
// ARRIVES LAST (device-scope counter, zeroed by the product kernel that precedes this one in the stream) runs the loss
// head over all M rows -- the same code on the same 1024-thread shape as the stand-alone kernel, reading complete
// arrays in a fixed order, so the loss is bit-identical to the two-launch form whichever block happens to be last.
// Inside the train step every launch of this latency-bound chain (finish -> weights -> loss) is stretched 5-7x by
// the table sweep's HBM traffic; one launch instead of two is 40 us of the 1.2 ms C2 step.
// NO __threadfence(): a device-scope release fence is an L2 write-back on this chip (buffer_wbl2), and 4096 waves each
// flushing an L2 that the Adam sweep keeps full of dirty lines made the step 0.2 ms LONGER.  Only row_ce crosses
// workgroups: it is stored and loaded as a device-scope relaxed atomic (write-through / cache-bypassing accesses), the
// workgroup barrier waits for the stores' acknowledgement, and the counter's atomic publishes them.
__global__ __launch_bounds__(1024) void ce_fwd_du_finish_loss_kernel(const float* __restrict__ part_m, const float* __restrict__ part_s,
                                                                     const float* __restrict__ diag, const float* __restrict__ slabs,
                                                                     int splits, int64_t M, int64_t D, const float* __restrict__ Y,
                                                                     int64_t ldy, int64_t diag_offset, float* __restrict__ row_lse,
                                                                     float* row_ce, float* __restrict__ du_unit, int64_t ld_du,
                                                                     const float* __restrict__ labels, int64_t T,
                                                                     const float* __restrict__ uvw, float* __restrict__ w_out,
                                                                     float* __restrict__ coef_out, float* __restrict__ loss_out,
                                                                     unsigned* counter) {
  ce_fwd_du_finish_row<16, true>(part_m, part_s, diag, slabs, splits, M, D, Y, ldy, diag_offset, row_lse, row_ce, du_unit, ld_du);
  __shared__ int is_last;
  // The hand-off is MI355X_MICROARCH.md's "handoff-flag, drained sc1" form: write-through (sc1) payload stores ->
  // s_waitcnt vmcnt(0) in the storing wave -> sc1 flag; reader: sc1 loads.  The wait is written out as inline assembly so it
  // does not depend on what hipcc chooses to emit for __syncthreads() (today: vmcnt(0) + s_barrier; the LLVM memory model
  // would allow a workgroup-scope barrier without it -- ADVICE r3); inline asm is invisible to the wait-count pass, so it
  // cannot be optimised away.  tt_inbatch_ce_fwd_du + tt_weighted_loss is the two-launch form, which has no cross-workgroup hand-off.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // every wave's stores (row_ce among them) have been acknowledged
  if (threadIdx.x == 0)
    is_last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!is_last) return;
  weighted_mean_loss_block<true>(labels, M, T, uvw, row_ce, w_out, coef_out, loss_out);
}

// dU[i, :] = du_unit[i, :] * coef[i] * g and coef_g[i] = coef[i] * g, g = the upstream gradient of the scalar loss (device
// scalar): the `coef * g` elementwise launch and tt_scale_rows of the two-op form in one
__global__ __launch_bounds__(256) void scale_rows_g_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ coef,
                                                           const float* __restrict__ g, int64_t rows, int64_t D,
                                                           float* __restrict__ out, int64_t ldo, float* __restrict__ coef_g) {
  const float gv = *g;
  const int64_t total = rows * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D, c = i - r * D;
    const float cg = coef[r] * gv;
    out[r * ldo + c] = x[r * ldx + c] * cg;
    if (c == 0) coef_g[r] = cg;
  }
}

// generic form for D > 128 (ce_wide.hip): logits materialised per row chunk, library GEMM
int64_t ce_wide_workspace_bytes(int64_t M, int64_t N, int64_t D);
int ce_wide_run(int mode, const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M, int64_t N, int64_t D,
                int64_t diag_offset, float* row_lse, float* row_ce, const float* coef, float* dU, int64_t lddu, float* dI,
                int64_t lddi, void* ws, int64_t ws_bytes, hipStream_t st);

struct CePlan {
  int dp8, splits;
  int64_t tiles_per_split;
};
static bool plan_ce(int64_t RX, int64_t RY, int64_t D, CePlan& pl) {
  if (D <= 32) pl.dp8 = 4; else if (D <= 64) pl.dp8 = 8; else if (D <= 128) pl.dp8 = 16; else return false;
  const int64_t rowblocks = ceil_div(RX, BI), tiles = ceil_div(RY, BJ);
  // 512 workgroups = one resident round (2 per CU).  Long streams are cut finer, up to 1024 units of
  // >= 16 tiles: when the Adam sweep is resident only ONE of these 256-VGPR workgroups fits per CU, and
  // with a single static round the late starters set the kernel time (8192 x 65536 under the sweep:
  // 3.5 ms with 512 units, 2.3 ms with 1024; alone 2.18 ms either way).
  int64_t splits = ceil_div(512, rowblocks);
  while (rowblocks * splits < 1024 && tiles >= 32 * splits) splits *= 2;
  if (splits > ceil_div(tiles, 4)) splits = ceil_div(tiles, 4);
  if (splits > 64) splits = 64;
  if (splits < 1) splits = 1;
  pl.tiles_per_split = ceil_div(tiles, splits);
  pl.splits = (int)ceil_div(tiles, pl.tiles_per_split);
  return true;
}
static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int DP8, bool GLDS>
static int launch_fwd(const CeArgs& a, dim3 grid, hipStream_t st) {
  ProfScope prof("ce_fwd_kernel", st);
  ce_fwd_kernel<DP8, GLDS><<<grid, 256, 0, st>>>(a);  // LDS is static: two named tile buffers
  return check_launch("ce_fwd_kernel");
}
template <int DP8, bool SS, bool GLDS>
static int launch_bwd(const CeArgs& a, dim3 grid, hipStream_t st) {
  ProfScope prof("ce_bwd_kernel", st);
  ce_bwd_kernel<DP8, SS, GLDS><<<grid, 256, 0, st>>>(a);  // LDS is static: two named tile buffers
  return check_launch("ce_bwd_kernel");
}
// LDS-DMA staging needs an unpadded, fully valid row: D == padded D, 16-B aligned rows
static bool can_dma(const float* Y, int64_t ld, int64_t D, int dp8) {
  static const bool off = getenv("TT_CE_NO_DMA") != nullptr;
  return !off && D == dp8 * 8 && (ld % 4 == 0) && ld <= (1 << 22) && al16(Y);  // tile_dma: 32-bit byte offsets within a tile
}
static int dispatch_fwd(int dp8, bool dma, const CeArgs& a, dim3 grid, hipStream_t st) {
  if (dma) return dp8 == 4 ? launch_fwd<4, true>(a, grid, st) : dp8 == 8 ? launch_fwd<8, true>(a, grid, st) : launch_fwd<16, true>(a, grid, st);
  return dp8 == 4 ? launch_fwd<4, false>(a, grid, st) : dp8 == 8 ? launch_fwd<8, false>(a, grid, st) : launch_fwd<16, false>(a, grid, st);
}
template <int DP8, bool GLDS, bool KEEP = false>
static int launch_fwd_du(const CeArgs& a, dim3 grid, hipStream_t st) {
  ProfScope prof("ce_fwd_kernel", st);
  ce_fwd_du_kernel<DP8, GLDS, KEEP><<<grid, 256, 0, st>>>(a);  // LDS is static: two named tile buffers
  return check_launch("ce_fwd_du_kernel");
}
template <int DP8>
static int launch_bwd_kept(const CeArgs& a, dim3 grid, hipStream_t st) {
  ProfScope prof("ce_bwd_kernel", st);
  ce_bwd_kept_kernel<DP8><<<grid, 256, 0, st>>>(a);  // LDS is static: two named tile buffers
  return check_launch("ce_bwd_kept_kernel");
}
static int dispatch_fwd_du_keep(int dp8, const CeArgs& a, dim3 grid, hipStream_t st) {  // LDS-DMA form only
  return dp8 == 4 ? launch_fwd_du<4, true, true>(a, grid, st) : dp8 == 8 ? launch_fwd_du<8, true, true>(a, grid, st) : launch_fwd_du<16, true, true>(a, grid, st);
}
static int dispatch_fwd_du(int dp8, bool dma, const CeArgs& a, dim3 grid, hipStream_t st) {
  if (dma) return dp8 == 4 ? launch_fwd_du<4, true>(a, grid, st) : dp8 == 8 ? launch_fwd_du<8, true>(a, grid, st) : launch_fwd_du<16, true>(a, grid, st);
  return dp8 == 4 ? launch_fwd_du<4, false>(a, grid, st) : dp8 == 8 ? launch_fwd_du<8, false>(a, grid, st) : launch_fwd_du<16, false>(a, grid, st);
}
template <bool SS>
static int dispatch_bwd(int dp8, bool dma, const CeArgs& a, dim3 grid, hipStream_t st) {
  if (dma) return dp8 == 4 ? launch_bwd<4, SS, true>(a, grid, st) : dp8 == 8 ? launch_bwd<8, SS, true>(a, grid, st) : launch_bwd<16, SS, true>(a, grid, st);
  return dp8 == 4 ? launch_bwd<4, SS, false>(a, grid, st) : dp8 == 8 ? launch_bwd<8, SS, false>(a, grid, st) : launch_bwd<16, SS, false>(a, grid, st);
}

}  // namespace tt

using namespace tt;

extern "C" int64_t tt_inbatch_ce_workspace_bytes(int64_t M, int64_t N, int64_t D) {
  if (M <= 0 || N <= 0 || D <= 0) return 0;
  if (D > 128) return ce_wide_workspace_bytes(M, N, D);
  CePlan pu, pi;
  if (!plan_ce(M, N, D, pu) || !plan_ce(N, M, D, pi)) return 0;
  const int64_t fwd = round_up((2 * (int64_t)pu.splits * M + M) * 4, 256);
  const int64_t du = pu.splits > 1 ? round_up((int64_t)pu.splits * M * D * 4, 256) : 0;
  const int64_t di = pi.splits > 1 ? round_up((int64_t)pi.splits * N * D * 4, 256) : 0;
  const int64_t bwd = du + di;
  // tt_inbatch_ce_fwd_du keeps the statistics and ALWAYS writes its per-split slabs
  const int64_t fwd_du = fwd + round_up((int64_t)pu.splits * M * D * 4, 256);
  const int64_t most = fwd > bwd ? fwd : bwd;
  return (most > fwd_du ? most : fwd_du) + 256;  // + the fused loss epilogue's arrival counter (last 256 B)
}

extern "C" int tt_inbatch_ce_fwd(const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M,
                                 int64_t N, int64_t D, int64_t diag_offset, float* row_lse, float* row_ce,
                                 void* ws, int64_t ws_bytes, tt_stream_t stream) {
  if (!U || !I || !row_lse || !row_ce || !ws) return fail_arg("tt_inbatch_ce_fwd: null pointer");
  if (M <= 0 || N <= 0 || D <= 0 || ldu < D || ldi < D) return fail_arg("tt_inbatch_ce_fwd: sizes");
  if (diag_offset < 0 || diag_offset + M > N) return fail_arg("tt_inbatch_ce_fwd: diagonal outside the item block");
  if (D > 128)
    return ce_wide_run(0, U, ldu, I, ldi, M, N, D, diag_offset, row_lse, row_ce, nullptr, nullptr, 0, nullptr, 0, ws, ws_bytes, S(stream));
  CePlan pl;
  if (!plan_ce(M, N, D, pl)) { set_error("tt_inbatch_ce: D=%lld > 128 not implemented", (long long)D); return TT_E_UNSUPPORTED; }
  if (ws_bytes < tt_inbatch_ce_workspace_bytes(M, N, D)) { set_error("tt_inbatch_ce_fwd: workspace"); return TT_E_WORKSPACE; }
  float* w = reinterpret_cast<float*>(ws);
  CeArgs a{};
  a.X = U; a.Y = I; a.ldx = ldu; a.ldy = ldi; a.RX = M; a.RY = N; a.D = D;
  a.diag_offset = diag_offset; a.tiles_per_split = pl.tiles_per_split; a.splits = pl.splits;
  a.x_vec = (ldu % 4 == 0) && al16(U); a.y_vec = (ldi % 4 == 0) && al16(I);
  a.part_m = w; a.part_s = w + (int64_t)pl.splits * M; a.diag = w + 2 * (int64_t)pl.splits * M;
  dim3 grid((unsigned)ceil_div(M, BI), (unsigned)pl.splits);
  hipStream_t st = S(stream);
  int rc = dispatch_fwd(pl.dp8, can_dma(I, ldi, D, pl.dp8), a, grid, st);
  if (rc) return rc;
  ce_fwd_finish_kernel<<<(unsigned)ceil_div(M, 256), 256, 0, st>>>(a.part_m, a.part_s, a.diag, M, pl.splits, row_lse, row_ce);
  return check_launch("ce_fwd_finish_kernel");
}

// the kept-logits kernels address a workgroup's 128 rows of the buffer with 32-bit byte offsets
constexpr int KEPT_MAX_COLS = 4 * 1024 * 1024 - 128;

extern "C" int64_t tt_inbatch_ce_logits_bytes(int64_t M, int64_t N) {
  if (M <= 0 || N <= 0) return 0;
  return round_up(M, 128) * round_up(N, 128) * 4;
}

struct CeLossTail {  // the weighted-mean loss head fused behind the forward (tt_inbatch_ce_fwd_du_loss)
  const float* labels; int64_t T; const float* uvw;
  float *w_out, *coef_out, *loss_out;
};
static int fwd_du_impl(const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M, int64_t N,
                       int64_t D, int64_t diag_offset, float* row_lse, float* row_ce, float* du_unit,
                       int64_t ld_du, float* logits, int64_t logits_bytes, void* ws, int64_t ws_bytes,
                       tt_stream_t stream, const CeLossTail* tail = nullptr);

extern "C" int tt_inbatch_ce_fwd_du(const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M, int64_t N,
                                    int64_t D, int64_t diag_offset, float* row_lse, float* row_ce, float* du_unit,
                                    int64_t ld_du, void* ws, int64_t ws_bytes, tt_stream_t stream) {
  return fwd_du_impl(U, ldu, I, ldi, M, N, D, diag_offset, row_lse, row_ce, du_unit, ld_du, nullptr, 0, ws, ws_bytes, stream);
}

extern "C" int tt_inbatch_ce_fwd_du_keep(const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M, int64_t N,
                                         int64_t D, int64_t diag_offset, float* row_lse, float* row_ce, float* du_unit,
                                         int64_t ld_du, float* logits, int64_t logits_bytes, void* ws, int64_t ws_bytes,
                                         tt_stream_t stream) {
  if (!logits) return fail_arg("tt_inbatch_ce_fwd_du_keep: null pointer");
  return fwd_du_impl(U, ldu, I, ldi, M, N, D, diag_offset, row_lse, row_ce, du_unit, ld_du, logits, logits_bytes, ws, ws_bytes, stream);
}

static int fwd_du_impl(const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M, int64_t N,
                       int64_t D, int64_t diag_offset, float* row_lse, float* row_ce, float* du_unit,
                       int64_t ld_du, float* logits, int64_t logits_bytes, void* ws, int64_t ws_bytes,
                       tt_stream_t stream, const CeLossTail* tail) {
  if (!U || !I || !row_lse || !row_ce || !du_unit || !ws) return fail_arg("tt_inbatch_ce_fwd_du: null pointer");
  if (M <= 0 || N <= 0 || D <= 0 || ldu < D || ldi < D || ld_du < D) return fail_arg("tt_inbatch_ce_fwd_du: sizes");
  if (diag_offset < 0 || diag_offset + M > N) return fail_arg("tt_inbatch_ce_fwd_du: diagonal outside the item block");
  if (D > 128) {
    if (logits) { set_error("tt_inbatch_ce_fwd_du_keep: needs D in {32, 64, 128} (use tt_inbatch_ce_fwd_du)"); return TT_E_UNSUPPORTED; }
    int rc = ce_wide_run(1, U, ldu, I, ldi, M, N, D, diag_offset, row_lse, row_ce, nullptr, du_unit, ld_du, nullptr, 0, ws, ws_bytes,
                         S(stream));
    if (rc || !tail) return rc;
    weighted_mean_loss_kernel<<<1, 1024, 0, S(stream)>>>(tail->labels, M, tail->T, tail->uvw, row_ce, tail->w_out, tail->coef_out,
                                                         tail->loss_out);
    return check_launch("weighted_mean_loss_kernel");
  }
  CePlan pl;
  if (!plan_ce(M, N, D, pl)) { set_error("tt_inbatch_ce: D=%lld > 128 not implemented", (long long)D); return TT_E_UNSUPPORTED; }
  if (ws_bytes < tt_inbatch_ce_workspace_bytes(M, N, D)) { set_error("tt_inbatch_ce_fwd_du: workspace"); return TT_E_WORKSPACE; }
  float* w = reinterpret_cast<float*>(ws);
  CeArgs a{};
  a.X = U; a.Y = I; a.ldx = ldu; a.ldy = ldi; a.RX = M; a.RY = N; a.D = D;
  a.diag_offset = diag_offset; a.tiles_per_split = pl.tiles_per_split; a.splits = pl.splits;
  a.x_vec = (ldu % 4 == 0) && al16(U); a.y_vec = (ldi % 4 == 0) && al16(I);
  a.part_m = w; a.part_s = w + (int64_t)pl.splits * M; a.diag = w + 2 * (int64_t)pl.splits * M;
  a.out = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + round_up((2 * (int64_t)pl.splits * M + M) * 4, 256));
  if (tail) a.counter = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + tt_inbatch_ce_workspace_bytes(M, N, D) - 256);
  dim3 grid((unsigned)ceil_div(M, BI), (unsigned)pl.splits);
  hipStream_t st = S(stream);
  int rc;
  if (logits) {
    if (!can_dma(I, ldi, D, pl.dp8) || !al16(logits)) {
      set_error("tt_inbatch_ce_fwd_du_keep: needs D in {32, 64, 128} with 16-B aligned rows (use tt_inbatch_ce_fwd_du)");
      return TT_E_UNSUPPORTED;
    }
    if (round_up(N, 128) > KEPT_MAX_COLS) { set_error("tt_inbatch_ce_fwd_du_keep: N > %d (32-bit offsets inside)", KEPT_MAX_COLS); return TT_E_UNSUPPORTED; }
    if (logits_bytes < tt_inbatch_ce_logits_bytes(M, N)) { set_error("tt_inbatch_ce_fwd_du_keep: logits buffer"); return TT_E_WORKSPACE; }
    a.keep = logits; a.ldk = round_up(N, 128);
    rc = dispatch_fwd_du_keep(pl.dp8, a, grid, st);
  } else {
    rc = dispatch_fwd_du(pl.dp8, can_dma(I, ldi, D, pl.dp8), a, grid, st);
  }
  if (rc) return rc;
  if (tail) {
    ce_fwd_du_finish_loss_kernel<<<(unsigned)ceil_div(M, 16), 1024, 0, st>>>(a.part_m, a.part_s, a.diag, a.out, pl.splits, M, D, I, ldi,
                                                                              diag_offset, row_lse, row_ce, du_unit, ld_du, tail->labels,
                                                                              tail->T, tail->uvw, tail->w_out, tail->coef_out,
                                                                              tail->loss_out, a.counter);
    return check_launch("ce_fwd_du_finish_loss_kernel");
  }
  ce_fwd_du_finish_kernel<<<(unsigned)ceil_div(M, 4), 256, 0, st>>>(a.part_m, a.part_s, a.diag, a.out, pl.splits, M, D, I,
                                                                     ldi, diag_offset, row_lse, row_ce, du_unit, ld_du);
  return check_launch("ce_fwd_du_finish_kernel");
}

extern "C" int tt_inbatch_ce_fwd_du_loss(const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M, int64_t N,
                                         int64_t D, int64_t diag_offset, const float* labels, int64_t T, const float* uvw,
                                         float* row_lse, float* row_ce, float* du_unit, int64_t ld_du, float* w_out,
                                         float* coef_out, float* loss_out, void* ws, int64_t ws_bytes, tt_stream_t stream) {
  if (!uvw || !w_out || !coef_out || !loss_out) return fail_arg("tt_inbatch_ce_fwd_du_loss: null pointer");
  if (labels && T <= 0) return fail_arg("tt_inbatch_ce_fwd_du_loss: sizes");
  const CeLossTail tail{labels, T, uvw, w_out, coef_out, loss_out};
  return fwd_du_impl(U, ldu, I, ldi, M, N, D, diag_offset, row_lse, row_ce, du_unit, ld_du, nullptr, 0, ws, ws_bytes, stream, &tail);
}

extern "C" int tt_scale_rows_g(const float* x, int64_t ldx, const float* coef, const float* g, int64_t rows, int64_t D,
                               float* out, int64_t ldo, float* coef_g, tt_stream_t stream) {
  if (!x || !coef || !g || !out || !coef_g) return fail_arg("tt_scale_rows_g: null pointer");
  if (rows <= 0 || D <= 0 || ldx < D || ldo < D) return fail_arg("tt_scale_rows_g: sizes");
  const int64_t total = rows * D;
  const unsigned blocks = (unsigned)(ceil_div(total, 256) < 4096 ? ceil_div(total, 256) : 4096);
  scale_rows_g_kernel<<<blocks, 256, 0, S(stream)>>>(x, ldx, coef, g, rows, D, out, ldo, coef_g);
  return check_launch("scale_rows_g_kernel");
}

extern "C" int tt_inbatch_ce_bwd(const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M,
                                 int64_t N, int64_t D, int64_t diag_offset, const float* row_lse,
                                 const float* coef, float* dU, int64_t lddu, float* dI, int64_t lddi,
                                 void* ws, int64_t ws_bytes, tt_stream_t stream) {
  if (!U || !I || !row_lse || !coef || !dI || !ws) return fail_arg("tt_inbatch_ce_bwd: null pointer");
  if (M <= 0 || N <= 0 || D <= 0 || ldu < D || ldi < D || (dU && lddu < D) || lddi < D) return fail_arg("tt_inbatch_ce_bwd: sizes");
  if (D > 128)
    return ce_wide_run(2, U, ldu, I, ldi, M, N, D, diag_offset, const_cast<float*>(row_lse), nullptr, coef, dU, lddu, dI, lddi, ws,
                       ws_bytes, S(stream));
  CePlan pu, pi;
  if (!plan_ce(M, N, D, pu) || !plan_ce(N, M, D, pi)) { set_error("tt_inbatch_ce: D=%lld > 128 not implemented", (long long)D); return TT_E_UNSUPPORTED; }
  if (ws_bytes < tt_inbatch_ce_workspace_bytes(M, N, D)) { set_error("tt_inbatch_ce_bwd: workspace"); return TT_E_WORKSPACE; }
  hipStream_t st = S(stream);
  char* wsb = reinterpret_cast<char*>(ws);
  float* slab_u = reinterpret_cast<float*>(wsb);
  float* slab_i = reinterpret_cast<float*>(wsb + (pu.splits > 1 ? round_up((int64_t)pu.splits * M * D * 4, 256) : 0));
  int rc;
  if (dU) {  // dU: stationary users, streamed items (skipped when the forward already produced it)
    CeArgs a{};
    a.X = U; a.Y = I; a.ldx = ldu; a.ldy = ldi; a.RX = M; a.RY = N; a.D = D; a.diag_offset = diag_offset;
    a.tiles_per_split = pu.tiles_per_split; a.splits = pu.splits;
    a.x_vec = (ldu % 4 == 0) && al16(U); a.y_vec = (ldi % 4 == 0) && al16(I);
    a.lse = row_lse; a.coef = coef; a.out = pu.splits > 1 ? slab_u : dU; a.ldo = lddu;
    dim3 grid((unsigned)ceil_div(M, BI), (unsigned)pu.splits);
    rc = dispatch_bwd<false>(pu.dp8, can_dma(I, ldi, D, pu.dp8), a, grid, st);
    if (rc) return rc;
    if (pu.splits > 1) {
      if ((rc = launch_slab_reduce(slab_u, pu.splits, M, D, dU, lddu, st))) return rc;
    }
  }
  {  // dI: stationary items, streamed users
    CeArgs a{};
    a.X = I; a.Y = U; a.ldx = ldi; a.ldy = ldu; a.RX = N; a.RY = M; a.D = D; a.diag_offset = diag_offset;
    a.tiles_per_split = pi.tiles_per_split; a.splits = pi.splits;
    a.x_vec = (ldi % 4 == 0) && al16(I); a.y_vec = (ldu % 4 == 0) && al16(U);
    a.lse = row_lse; a.coef = coef; a.out = pi.splits > 1 ? slab_i : dI; a.ldo = lddi;
    dim3 grid((unsigned)ceil_div(N, BI), (unsigned)pi.splits);
    rc = dispatch_bwd<true>(pi.dp8, can_dma(U, ldu, D, pi.dp8), a, grid, st);
    if (rc) return rc;
    if (pi.splits > 1) {
      if ((rc = launch_slab_reduce(slab_i, pi.splits, N, D, dI, lddi, st))) return rc;
    }
  }
  return 0;
}

extern "C" int tt_inbatch_ce_bwd_kept(const float* U, int64_t ldu, int64_t M, int64_t N, int64_t D, int64_t diag_offset,
                                      const float* row_lse, const float* coef, const float* logits,
                                      int64_t logits_bytes, float* dI, int64_t lddi, void* ws, int64_t ws_bytes,
                                      tt_stream_t stream) {
  if (!U || !row_lse || !coef || !logits || !dI || !ws) return fail_arg("tt_inbatch_ce_bwd_kept: null pointer");
  if (M <= 0 || N <= 0 || D <= 0 || ldu < D || lddi < D) return fail_arg("tt_inbatch_ce_bwd_kept: sizes");
  CePlan pu, pi;
  if (!plan_ce(M, N, D, pu) || !plan_ce(N, M, D, pi)) { set_error("tt_inbatch_ce: D=%lld > 128 not implemented", (long long)D); return TT_E_UNSUPPORTED; }
  if (!can_dma(U, ldu, D, pi.dp8) || !al16(logits)) {
    set_error("tt_inbatch_ce_bwd_kept: needs D in {32, 64, 128} with 16-B aligned rows");
    return TT_E_UNSUPPORTED;
  }
  if (round_up(N, 128) > KEPT_MAX_COLS) { set_error("tt_inbatch_ce_bwd_kept: N > %d (32-bit offsets inside)", KEPT_MAX_COLS); return TT_E_UNSUPPORTED; }
  if (logits_bytes < tt_inbatch_ce_logits_bytes(M, N)) { set_error("tt_inbatch_ce_bwd_kept: logits buffer"); return TT_E_WORKSPACE; }
  if (ws_bytes < tt_inbatch_ce_workspace_bytes(M, N, D)) { set_error("tt_inbatch_ce_bwd_kept: workspace"); return TT_E_WORKSPACE; }
  hipStream_t st = S(stream);
  float* slab_i = reinterpret_cast<float*>(ws);
  CeArgs a{};
  a.Y = U; a.ldy = ldu; a.RX = N; a.RY = M; a.D = D; a.diag_offset = diag_offset;
  a.tiles_per_split = pi.tiles_per_split; a.splits = pi.splits;
  a.lse = row_lse; a.coef = coef; a.out = pi.splits > 1 ? slab_i : dI; a.ldo = lddi;
  a.keep = const_cast<float*>(logits); a.ldk = round_up(N, 128);
  dim3 grid((unsigned)ceil_div(N, BI), (unsigned)pi.splits);
  int rc = pi.dp8 == 4 ? launch_bwd_kept<4>(a, grid, st) : pi.dp8 == 8 ? launch_bwd_kept<8>(a, grid, st) : launch_bwd_kept<16>(a, grid, st);
  if (rc) return rc;
  if (pi.splits > 1) {
    if ((rc = launch_slab_reduce(slab_i, pi.splits, N, D, dI, lddi, st))) return rc;
  }
  return 0;
}

extern "C" int tt_scale_rows(const float* x, int64_t ldx, const float* coef, int64_t rows, int64_t D, float* out,
                             int64_t ldo, tt_stream_t stream) {
  if (!x || !coef || !out) return fail_arg("tt_scale_rows: null pointer");
  if (rows <= 0 || D <= 0 || ldx < D || ldo < D) return fail_arg("tt_scale_rows: sizes");
  const int64_t blocks = ceil_div(rows * D, 256) < 2048 ? ceil_div(rows * D, 256) : 2048;
  scale_rows_kernel<<<(unsigned)blocks, 256, 0, S(stream)>>>(x, ldx, coef, rows, D, out, ldo);
  return check_launch("scale_rows_kernel");
}

extern "C" int tt_value_weights(const float* labels, int64_t B, int64_t T, const float* uvw, float* nuv_out, float* max_out,
                                tt_stream_t stream) {
  if (!labels || !uvw || !nuv_out || !max_out) return fail_arg("tt_value_weights: null pointer");
  if (B <= 0 || T <= 0) return fail_arg("tt_value_weights: sizes");
  value_weights_kernel<<<1, 1024, 0, S(stream)>>>(labels, B, T, uvw, nuv_out, max_out);
  return check_launch("value_weights_kernel");
}

extern "C" int tt_weighted_loss_global(const float* nuv, const float* gmax, const float* row_ce, int64_t B, float denom,
                                       float* coef_out, float* loss_out, tt_stream_t stream) {
  if (!nuv || !gmax || !row_ce || !coef_out || !loss_out) return fail_arg("tt_weighted_loss_global: null pointer");
  if (B <= 0 || !(denom > 0.f)) return fail_arg("tt_weighted_loss_global: sizes");
  weighted_loss_global_kernel<<<1, 1024, 0, S(stream)>>>(nuv, gmax, row_ce, B, denom, coef_out, loss_out);
  return check_launch("weighted_loss_global_kernel");
}

extern "C" int tt_weighted_mean_loss(const float* labels, int64_t B, int64_t T, const float* uvw,
                                     const float* row_ce, float* w_out, float* coef_out, float* loss_out,
                                     tt_stream_t stream) {
  if (!uvw || !row_ce || !w_out || !coef_out || !loss_out) return fail_arg("tt_weighted_mean_loss: null pointer");
  if (B <= 0 || T <= 0) return fail_arg("tt_weighted_mean_loss: sizes");
  weighted_mean_loss_kernel<<<1, 1024, 0, S(stream)>>>(labels, B, T, uvw, row_ce, w_out, coef_out, loss_out);
  return check_launch("weighted_mean_loss_kernel");
}
