// Shared device code of the "stationary operand in registers, streamed operand through
// double-buffered LDS" fp32-MFMA kernels (inbatch_ce.hip, gemm_ws.hip): tile addressing,
// LDS-DMA / register staging, and the 32x32 transposed product tile.
#pragma once
#include "common.hpp"

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BI = 128;  // stationary rows per workgroup (32 per wave)
constexpr int BJ = 64;   // streamed rows per LDS tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float NEG_BIG = -3.0e38f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// a * b rounded to fp32 on its own: never contracted into a following add (hipcc defaults to
// -ffp-contract=fast, and __fmul_rn is a plain multiply there)
__device__ __forceinline__ float mul_rounded(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}

// this wave's 32 stationary rows -> B-operand fragments: xr[g][c] = X[a][8g + 4h + c]
template <int DP8>
__device__ __forceinline__ void load_stationary(float (&xr)[DP8][4], const float* __restrict__ X,
                                                int64_t ld, int64_t row, int64_t nrows, int64_t D,
                                                int h, bool vec) {
  if (vec && D == 8 * DP8) {
    // full-width aligned rows: unconditional loads from a clamped row, masked afterwards (guarded
    // loads make hipcc wait for each load before the next guard: DP8 exposed round trips per wave)
    const bool ok = row < nrows;
    const float* p = X + (ok ? row : nrows - 1) * ld + 4 * h;
#pragma unroll
    for (int g = 0; g < DP8; ++g) {
      const float x0 = p[8 * g], x1 = p[8 * g + 1], x2 = p[8 * g + 2], x3 = p[8 * g + 3];
      xr[g][0] = ok ? x0 : 0.f; xr[g][1] = ok ? x1 : 0.f; xr[g][2] = ok ? x2 : 0.f; xr[g][3] = ok ? x3 : 0.f;
    }
    return;
  }
#pragma unroll
  for (int g = 0; g < DP8; ++g) {
    const int64_t k = 8 * g + 4 * h;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows) {
      const float* p = X + row * ld + k;
      if (vec && k + 3 < D) {
        v = *reinterpret_cast<const float4*>(p);
      } else {
        if (k + 0 < D) v.x = p[0];
        if (k + 1 < D) v.y = p[1];
        if (k + 2 < D) v.z = p[2];
        if (k + 3 < D) v.w = p[3];
      }
    }
    xr[g][0] = v.x; xr[g][1] = v.y; xr[g][2] = v.z; xr[g][3] = v.w;
  }
}

// streamed tile [BJ][DP] : DP8/2 float4 per thread
template <int DP8>
__device__ __forceinline__ void tile_fetch(float4 (&st)[(DP8 + 1) / 2], const float* __restrict__ Y,
                                           int64_t ld, int64_t row0, int64_t nrows, int64_t D, bool vec) {
  constexpr int C4 = DP8 * 2;  // float4 per row
#pragma unroll
  for (int i = 0; i < (DP8 + 1) / 2; ++i) {
    const int f = threadIdx.x + 256 * i;
    const int64_t row = row0 + f / C4, k = 4 * (f % C4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < BJ * C4 && row < nrows) {
      const float* p = Y + row * ld + k;
      if (vec && k + 3 < D) {
        v = *reinterpret_cast<const float4*>(p);
      } else {
        if (k + 0 < D) v.x = p[0];
        if (k + 1 < D) v.y = p[1];
        if (k + 2 < D) v.z = p[2];
        if (k + 3 < D) v.w = p[3];
      }
    }
    st[i] = v;
  }
}
template <int DP8>
__device__ __forceinline__ void tile_commit(const float4 (&st)[(DP8 + 1) / 2], float* Ys) {
  constexpr int C4 = DP8 * 2, LD = DP8 * 8 + 4;
#pragma unroll
  for (int i = 0; i < (DP8 + 1) / 2; ++i) {
    const int f = threadIdx.x + 256 * i;
    if (f < BJ * C4) *reinterpret_cast<float4*>(Ys + (f / C4) * LD + 4 * (f % C4)) = st[i];
  }
}

// ---- LDS image of one streamed tile: addressing for both staging forms
template <int DP8, bool GLDS>
struct TileMap {
  static constexpr int DP = DP8 * 8;
  static constexpr int LD = GLDS ? DP : DP + 4;          // row stride in floats
  static constexpr int CPR = DP / 4;                     // 16-B chunks per row
  static constexpr int SW = (CPR < 16 ? CPR : 16) - 1;   // swizzle mask
  // float offset of 16-B chunk `c` of row `row`
  static __device__ __forceinline__ int chunk(int row, int c) {
    return GLDS ? row * LD + 4 * (c ^ (row & SW)) : row * LD + 4 * c;
  }
  // float offset of element (row, col)
  static __device__ __forceinline__ int elem(int row, int col) {
    return GLDS ? row * LD + 4 * ((col >> 2) ^ (row & SW)) + (col & 3) : row * LD + col;
  }
};

// LDS-DMA of one 64-row tile: each wave instruction lands 1 KiB (= 64/CPR rows); the lane
// fetches the chunk that belongs at ITS linear LDS position after swizzling.
// Issued as buffer loads (buffer_load_dwordx4 ... lds): the tile base sits in a scalar resource
// descriptor and the lane contributes ONE 32-bit offset, so there is no 64-bit per-lane address
// arithmetic and no clamping (rows past the end are outside the descriptor's range and land as
// zeros; every consumer masks them anyway).  In the kernels at the register limit the 64-bit
// addresses of the global_load form were spilled, and a reload waits on vmcnt -- the DMA just issued.
template <int DP8>
__device__ __forceinline__ void tile_dma(const float* __restrict__ Y, int64_t ld, int64_t row0, int64_t nrows,
                                         float* Ys, int wave, int lane) {
  using TM = TileMap<DP8, true>;
  constexpr int RPI = 64 / TM::CPR;        // rows per wave instruction
  constexpr int NI = BJ / RPI / 4;         // instructions per wave
  const int64_t left = nrows - row0;
  const int rows_here = left < BJ ? (int)left : BJ;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Y + row0 * ld), 0, rows_here * (int)ld * 4, 0x00020000);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int rbase = (wave * NI + i) * RPI;
    const int row = rbase + lane / TM::CPR;
    const int c = (lane % TM::CPR) ^ (row & TM::SW);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(Ys + rbase * TM::DP), 16,
                                             (row * (int)ld + 4 * c) * 4, 0, 0, 0);
  }
}

// St[b][a] for one 32-row sub-tile `jt` of the LDS tile
template <int DP8, bool GLDS>
__device__ __forceinline__ f32x16 score_tile(const float* Ys, const float (&xr)[DP8][4], int jt, int r, int h) {
  using TM = TileMap<DP8, GLDS>;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int row = jt * 32 + r;
  // A-operand reads run one k-group ahead of the MFMAs (register double buffer); the
  // scheduling barrier stops the compiler from hoisting all DP8 reads (64 VGPRs at D=128)
  float4 y[2];
  y[0] = *reinterpret_cast<const float4*>(Ys + TM::chunk(row, h));
#pragma unroll
  for (int g = 0; g < DP8; ++g) {
    if (g + 1 < DP8) y[(g + 1) & 1] = *reinterpret_cast<const float4*>(Ys + TM::chunk(row, 2 * (g + 1) + h));
    const float4 v = y[g & 1];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, xr[g][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, xr[g][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, xr[g][2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, xr[g][3], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  return acc;
}

__device__ __forceinline__ int brow(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

template <int DP8, bool GLDS>
struct Stager {
  float4 st[GLDS ? 1 : (DP8 + 1) / 2];
  __device__ __forceinline__ void issue(const float* Y, int64_t ld, int64_t row0, int64_t nrows, int64_t D, bool vec,
                                        float* dst, int wave, int lane) {
    if constexpr (GLDS) tile_dma<DP8>(Y, ld, row0, nrows, dst, wave, lane);
    else tile_fetch<DP8>(st, Y, ld, row0, nrows, D, vec);
  }
  __device__ __forceinline__ void land(float* dst) {
    if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else tile_commit<DP8>(st, dst);
  }
};

}  // namespace tt
