// K3: one tower of TwoTowerBaseRetrieval as ONE kernel per direction.
//   forward   y = Linear(2D -> D)( [ table[id] | Linear(256 -> D)(ReLU(Linear(F -> 256)(features))) ] )
//             ref:src/two_tower_base_retrieval.py:129-162 (user), :193-219 (item): nn.Embedding lookup,
//             nn.Sequential feature MLP, torch.cat, tower nn.Linear -- the cat never exists, nothing but
//             the ids / features is read from HBM and only y (+ what the backward needs) is written
//   backward  d_tin = dy W3;  d_emb = d_tin[:, :D] (the embedding-row gradients);  d_f = d_tin[:, D:];
//             dh = (d_f W2) (.) [h > 0]          (autograd of the same lines, data side)
// The weight gradients (dW3 = dy^T tin, dW2 = d_f^T h, dW1 = dh^T x and the three bias sums) are reductions over
// the batch and stay with tt_gemm_tn_colsum_f32.
//
// One workgroup = 64 batch rows, 4 waves.  The row block's activations live in LDS ([64][K + 4] images, the layout
// of mfma_stream.hpp's register-staged tiles), each wave owns 32 output columns and holds their weights in
// registers as MFMA A-operand fragments (v_mfma_f32_32x32x2_f32: exact fp32), so a result tile has one batch row
// per lane: the same structure as gemm_ws.hip with the "stream" being a single resident tile.  Layer 1 has K = F
// (8 at the BASELINE shapes): VALU, one hidden unit per thread.  Against the unfused path (gather + 3 GEMM launches
// forward, 3 NN GEMMs backward, each a separate pass over [B, 256]-sized intermediates): 1 launch per direction.
// Shapes: hidden = 256 (fixed upstream), D in {32, 64, 128}, F <= 64, 16-B aligned rows; anything else returns
// TT_E_UNSUPPORTED and the caller takes the GEMM path.
//
// XTRA variants (tt_tower_*_x): the tower input has a third, dense block -- [ id | MLP | extra[B, 2D] ], tower
// Linear(4D -> D) -- which is TwoTowerWithUserHistoryEncoder's user tower (extra = the encoder's [recent | mean]
// summary; ref:src/two_tower_with_user_history_encoder.py:81-83,85-122).  The extra block takes the place of the hidden
// activations in LDS once they are consumed (same [64][2D + 4] image as the tower input), the last product runs as
// two K chunks over the same accumulators, the backward's d_tin has 2D more columns (-> d_extra) and the weight
// gradient one more operand pass (dW3[:, 2D:] = dy^T extra).
#include "mfma_stream.hpp"

namespace tt {

constexpr int TW_ROWS = 64;
constexpr int TW_HID = 256;
constexpr int TW_FMAX = 64;

struct TowerFwdArgs {
  const float* table; int64_t n_rows; const int64_t* ids;
  const float* feats; int64_t ldf;
  int64_t B, F;
  const float *W1, *b1, *W2, *b2, *W3, *b3;
  float* y; int64_t ldy;
  float* h_out;    // [B][256]
  float* tin_out;  // [B][2D]
  int32_t* oob;
  const float* extra; int64_t ldx;  // XTRA: [B][2D] third block of the tower input
};

// weights as the MFMA A operand (rows = this wave's 32 output columns), activations as B from the LDS image:
// acc[e] = out[row = jt*32 + (lane & 31)][col = nw + (e&3) + 8*(e>>2) + 4*(lane>>5)]
template <int DP8>
__device__ __forceinline__ f32x16 tw_tile_acc(f32x16 acc, const float* Ys, const float (&xr)[DP8][4], int jt, int r, int h) {
  using TM = TileMap<DP8, false>;
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
template <int DP8>
__device__ __forceinline__ f32x16 tw_tile(const float* Ys, const float (&xr)[DP8][4], int jt, int r, int h) {
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  return tw_tile_acc<DP8>(acc, Ys, xr, jt, r, h);
}

// stationary fragments from a [K][N] (row = reduction index) weight: xr[g][c] = W[8g+4h+c][n]
template <int DP8>
__device__ __forceinline__ void tw_load_t(float (&xr)[DP8][4], const float* __restrict__ W, int64_t ld, int64_t n, int64_t N,
                                          int h) {
#pragma unroll
  for (int g = 0; g < DP8; ++g)
#pragma unroll
    for (int c = 0; c < 4; ++c) xr[g][c] = (n < N) ? W[(int64_t)(8 * g + 4 * h + c) * ld + n] : 0.f;
}

// RT = 32-row tiles per workgroup: 2 (64 batch rows) when that still gives every CU a workgroup, else 1 -- at B = 8192 the
// 64-row form is 128 workgroups on a 256-CU chip, and the tower is latency-bound (weights from L2, three dependent stages).
template <int DE8, bool XTRA, int RT>  // D = 8 * DE8
__device__ __forceinline__ void tower_fwd_body(const TowerFwdArgs& p) {
  constexpr int ROWS = 32 * RT;
  constexpr int D = 8 * DE8, K3 = 2 * D, LDH = TW_HID + 4, LDT = K3 + 4, LDW3 = XTRA ? 2 * K3 : K3;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* hT = smem;                    // [64][260]  hidden activations
  float* tT = hT + ROWS * LDH;      // [64][2D+4] tower input: id row | feature-MLP output
  float* fT = tT + ROWS * LDT;      // [64][F]    dense features
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int F = (int)p.F;

  // ---- stage 0: features and id rows of the 64 batch rows into LDS
  for (int i = threadIdx.x; i < ROWS * F; i += 256) {
    const int m = i / F, f = i - m * F;
    fT[i] = (row0 + m < p.B) ? p.feats[(row0 + m) * p.ldf + f] : 0.f;
  }
  {
    constexpr int C4 = D / 4;  // float4 per id row
    for (int i = threadIdx.x; i < ROWS * C4; i += 256) {
      const int m = i / C4, c = i - m * C4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + m < p.B) {
        const int64_t id = p.ids[row0 + m];
        if (id >= 0 && id < p.n_rows) v = *reinterpret_cast<const float4*>(p.table + id * D + 4 * c);
        else if (c == 0 && p.oob) *p.oob = 1;
      }
      *reinterpret_cast<float4*>(tT + m * LDT + 4 * c) = v;
    }
  }
  __syncthreads();
  // ---- stage 1: h = ReLU(x W1^T + b1), one hidden unit per thread (K = F is tiny: VALU)
  {
    const int j = threadIdx.x;
    float w1r[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) w1r[f] = f < F ? p.W1[(int64_t)j * F + f] : 0.f;
    const float bj = p.b1[j];
    for (int m = 0; m < ROWS; ++m) {
      float acc = bj;
#pragma unroll
      for (int f = 0; f < 16; ++f)
        if (f < F) acc = fmaf(fT[m * F + f], w1r[f], acc);
      for (int f = 16; f < F; ++f) acc = fmaf(fT[m * F + f], p.W1[(int64_t)j * F + f], acc);
      hT[m * LDH + j] = fmaxf(acc, 0.f);
    }
  }
  __syncthreads();
  const int nw = wave * 32;  // this wave's output columns (of f, then of y)
  // ---- stage 2: f = h W2^T + b2 -> columns D .. 2D-1 of the tower input (LDS)
  if (nw < D) {
    float xr[TW_HID / 8][4];
    load_stationary<TW_HID / 8>(xr, p.W2, TW_HID, nw + r, D, TW_HID, h, true);
    float be[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) be[e] = p.b2[nw + (e & 3) + 8 * (e >> 2) + 4 * h];
#pragma unroll
    for (int jt = 0; jt < RT; ++jt) {
      const f32x16 acc = tw_tile<TW_HID / 8>(hT, xr, jt, r, h);
      float* dst = tT + (jt * 32 + r) * LDT + D + nw + 4 * h;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(dst + 8 * q) = make_float4(acc[4 * q] + be[4 * q], acc[4 * q + 1] + be[4 * q + 1],
                                                             acc[4 * q + 2] + be[4 * q + 2], acc[4 * q + 3] + be[4 * q + 3]);
    }
  }
  // what the backward needs: h (ReLU mask, dW2) -- coalesced copy of the LDS image
  for (int i = threadIdx.x; i < ROWS * (TW_HID / 4); i += 256) {
    const int m = i / (TW_HID / 4), c = i - m * (TW_HID / 4);
    if (row0 + m < p.B)
      *reinterpret_cast<float4*>(p.h_out + (row0 + m) * TW_HID + 4 * c) = *reinterpret_cast<const float4*>(hT + m * LDH + 4 * c);
  }
  __syncthreads();
  // ---- stage 3: y = tin W3^T + b3; tin goes out for dW3
  for (int i = threadIdx.x; i < ROWS * (K3 / 4); i += 256) {
    const int m = i / (K3 / 4), c = i - m * (K3 / 4);
    if (row0 + m < p.B)
      *reinterpret_cast<float4*>(p.tin_out + (row0 + m) * K3 + 4 * c) = *reinterpret_cast<const float4*>(tT + m * LDT + 4 * c);
  }
  float* xT = hT;  // XTRA: the third block takes the hidden activations' place ([64][2D + 4], 2D <= 256)
  if constexpr (XTRA) {
    for (int i = threadIdx.x; i < ROWS * (K3 / 4); i += 256) {
      const int m = i / (K3 / 4), c = i - m * (K3 / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + m < p.B) v = *reinterpret_cast<const float4*>(p.extra + (row0 + m) * p.ldx + 4 * c);
      *reinterpret_cast<float4*>(xT + m * LDT + 4 * c) = v;
    }
    __syncthreads();
  }
  if (nw < D) {
    float xr[K3 / 8][4];
    load_stationary<K3 / 8>(xr, p.W3, LDW3, nw + r, D, K3, h, true);
    float be[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) be[e] = p.b3[nw + (e & 3) + 8 * (e >> 2) + 4 * h];
    f32x16 accs[RT];
#pragma unroll
    for (int jt = 0; jt < RT; ++jt) accs[jt] = tw_tile<K3 / 8>(tT, xr, jt, r, h);
    if constexpr (XTRA) {  // second K chunk: W3[:, 2D:] against the extra block, same accumulators
      load_stationary<K3 / 8>(xr, p.W3 + K3, LDW3, nw + r, D, K3, h, true);
#pragma unroll
      for (int jt = 0; jt < RT; ++jt) accs[jt] = tw_tile_acc<K3 / 8>(accs[jt], xT, xr, jt, r, h);
    }
#pragma unroll
    for (int jt = 0; jt < RT; ++jt) {
      const f32x16 acc = accs[jt];
      const int64_t m = row0 + jt * 32 + r;
      if (m < p.B) {
        float* dst = p.y + m * p.ldy + nw + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(dst + 8 * q) = make_float4(acc[4 * q] + be[4 * q], acc[4 * q + 1] + be[4 * q + 1],
                                                               acc[4 * q + 2] + be[4 * q + 2], acc[4 * q + 3] + be[4 * q + 3]);
      }
    }
  }
}

template <int DE8, bool XTRA, int RT>
__global__ __launch_bounds__(256) void tower_fwd_kernel(const TowerFwdArgs p) { tower_fwd_body<DE8, XTRA, RT>(p); }
// BOTH towers of the base model in one launch (blockIdx.y = tower; tt_tower_fwd_pair): the same body, so the same bits.  One
// tower is at most 128 workgroups -- half the chip -- and a step that runs on ONE stream (a hipGraph capture, batches too
// small for the two-stream fork) paid for the two launches back to back.
struct TowerFwdArgs2 { TowerFwdArgs t[2]; };
template <int DE8, int RT>
__global__ __launch_bounds__(256) void tower_fwd_pair_kernel(const TowerFwdArgs2 q) { tower_fwd_body<DE8, false, RT>(q.t[blockIdx.y]); }

struct TowerBwdArgs {
  const float* dy; int64_t ldy;
  int64_t B;
  const float *W2, *W3;
  const float* h;        // [B][256] saved by the forward
  float* d_emb; int64_t ld_demb;  // [B][D]
  float* d_f;            // [B][D]
  float* dh;             // [B][256]
  float* d_x; int64_t ld_dx;  // XTRA: [B][2D] gradient of the third block
};

template <int DE8, bool XTRA, int RT>
__device__ __forceinline__ void tower_bwd_body(const TowerBwdArgs& p) {
  constexpr int ROWS = 32 * RT;
  constexpr int D = 8 * DE8, K3 = 2 * D, LDY = D + 4, LDW3 = XTRA ? 2 * K3 : K3;
  __shared__ __attribute__((aligned(16))) float dyT[ROWS * LDY];
  __shared__ __attribute__((aligned(16))) float dfT[ROWS * LDY];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  for (int i = threadIdx.x; i < ROWS * (D / 4); i += 256) {
    const int m = i / (D / 4), c = i - m * (D / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + m < p.B) v = *reinterpret_cast<const float4*>(p.dy + (row0 + m) * p.ldy + 4 * c);
    *reinterpret_cast<float4*>(dyT + m * LDY + 4 * c) = v;
  }
  __syncthreads();
  // ---- d_tin = dy W3 (W3 [D][2D (+ 2D)]: reduction index = row): columns < D -> d_emb, D .. 2D-1 -> d_f, >= 2D -> d_x
  for (int cb = 0; cb * 128 < LDW3; ++cb) {
    const int nw = cb * 128 + wave * 32;
    if (nw >= LDW3) continue;
    float xr[DE8][4];
    tw_load_t<DE8>(xr, p.W3, LDW3, nw + r, LDW3, h);
#pragma unroll
    for (int jt = 0; jt < RT; ++jt) {
      const f32x16 acc = tw_tile<DE8>(dyT, xr, jt, r, h);
      const int ml = jt * 32 + r;
      const int64_t m = row0 + ml;
      if (nw < D) {  // the whole 32-column block lies in the id half (D is a multiple of 32)
        if (m < p.B) {
          float* dst = p.d_emb + m * p.ld_demb + nw + 4 * h;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(dst + 8 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
      } else if (XTRA && nw >= K3) {
        if (m < p.B) {
          float* dst = p.d_x + m * p.ld_dx + (nw - K3) + 4 * h;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(dst + 8 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
      } else {
        float* l = dfT + ml * LDY + (nw - D) + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
          *reinterpret_cast<float4*>(l + 8 * q) = v;
          if (m < p.B) *reinterpret_cast<float4*>(p.d_f + m * D + (nw - D) + 4 * h + 8 * q) = v;
        }
      }
    }
  }
  __syncthreads();
  // ---- dh = (d_f W2) (.) [h > 0]   (W2 [D][256]: reduction index = row)
#pragma unroll 1
  for (int cb = 0; cb < 2; ++cb) {
    const int nw = cb * 128 + wave * 32;
    float xr[DE8][4];
    tw_load_t<DE8>(xr, p.W2, TW_HID, nw + r, TW_HID, h);
#pragma unroll
    for (int jt = 0; jt < RT; ++jt) {
      const f32x16 acc = tw_tile<DE8>(dfT, xr, jt, r, h);
      const int64_t m = row0 + jt * 32 + r;
      if (m < p.B) {
        const float* hm = p.h + m * TW_HID + nw + 4 * h;
        float* dst = p.dh + m * TW_HID + nw + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 hv = *reinterpret_cast<const float4*>(hm + 8 * q);
          *reinterpret_cast<float4*>(dst + 8 * q) =
              make_float4(hv.x > 0.f ? acc[4 * q] : 0.f, hv.y > 0.f ? acc[4 * q + 1] : 0.f, hv.z > 0.f ? acc[4 * q + 2] : 0.f,
                          hv.w > 0.f ? acc[4 * q + 3] : 0.f);
        }
      }
    }
  }
}

template <int DE8, bool XTRA, int RT>
__global__ __launch_bounds__(256) void tower_bwd_kernel(const TowerBwdArgs p) { tower_bwd_body<DE8, XTRA, RT>(p); }
struct TowerBwdArgs2 { TowerBwdArgs t[2]; };
template <int DE8, int RT>
__global__ __launch_bounds__(256) void tower_bwd_pair_kernel(const TowerBwdArgs2 q) { tower_bwd_body<DE8, false, RT>(q.t[blockIdx.y]); }

// ---------------------------------------------------------------- weight gradients of one tower
// dW3 = dy^T tin, dW2 = d_f^T h, dW1 = dh^T x and the three bias sums: reductions over the batch.  As three
// tt_gemm_tn_colsum_f32 calls they were six launches per tower (split-K product + slab reduce each).  Here ONE launch
// writes every 64-row block's partial of all six tensors (TN products on the matrix cores: the two operands of a
// v_mfma_f32_32x32x2_f32 step are one LDS row pair of the dy / d_f tile and of the tin / h tile), and one reduce launch
// adds the partials in block order (deterministic).  Layout of a partial (floats):
//   dW3 [D][2D] | dW2 [D][256] | dW1 [256][F] | db3 [D] | db2 [D] | db1 [256]
struct TowerWgradArgs {
  const float* dy; int64_t ldy;
  const float *tin, *d_f, *h, *dh, *feats;
  int64_t ldf, F, B;
  float* part;
  const float* extra; int64_t ldx;  // XTRA
};

template <int D, bool XTRA>
__host__ __device__ constexpr int64_t tw_part_floats(int64_t F) {
  return (int64_t)D * (XTRA ? 4 : 2) * D + (int64_t)D * TW_HID + TW_HID * F + 2 * D + TW_HID;
}

// out[n0 + i][k0 + j] = sum_m A[m][n0 + i] * Bm[m][k0 + j] over the 64 rows of the tiles; one 32 x 32 tile per call
template <int LDA, int LDB>
__device__ __forceinline__ void tw_tn_tile(const float* At, const float* Bt, int n0, int k0, float* out, int ldo, int r, int h) {
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const float* a = At + h * LDA + n0 + r;
  const float* b = Bt + h * LDB + k0 + r;
#pragma unroll 8
  for (int s = 0; s < TW_ROWS / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * s * LDA], b[2 * s * LDB], acc, 0, 0, 0);
#pragma unroll
  for (int e = 0; e < 16; ++e) out[(n0 + (e & 3) + 8 * (e >> 2) + 4 * h) * ldo + k0 + r] = acc[e];
}

template <int DE8, bool XTRA>
__device__ __forceinline__ void tower_wgrad_body(const TowerWgradArgs& p) {
  constexpr int D = 8 * DE8, K3 = 2 * D, LDW3 = XTRA ? 2 * K3 : K3;
  extern __shared__ __attribute__((aligned(16))) float tw_smem[];
  float* At = tw_smem;                   // [64][D]     dy, then d_f
  float* Bt = At + TW_ROWS * D;          // [64][256]   tin (2D <= 256 columns used), then h, then dh
  float* Xt = Bt + TW_ROWS * TW_HID;     // [64][F]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * TW_ROWS;
  const int F = (int)p.F;
  float* part = p.part + (int64_t)blockIdx.x * tw_part_floats<D, XTRA>(F);
  float* pW3 = part;
  float* pW2 = pW3 + D * LDW3;
  float* pW1 = pW2 + D * TW_HID;
  float* pb3 = pW1 + TW_HID * F;
  float* pb2 = pb3 + D;
  float* pb1 = pb2 + D;
  auto load = [&](float* dst, int ld_dst, const float* src, int64_t ld_src, int cols) {  // rows past B: zeros
    const int c4 = cols / 4;
    for (int i = threadIdx.x; i < TW_ROWS * c4; i += 256) {
      const int m = i / c4, c = i - m * c4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + m < p.B) v = *reinterpret_cast<const float4*>(src + (row0 + m) * ld_src + 4 * c);
      *reinterpret_cast<float4*>(dst + m * ld_dst + 4 * c) = v;
    }
  };
  auto colsum = [&](const float* T, int ld, int cols, float* out) {
    for (int c = threadIdx.x; c < cols; c += 256) {
      float s = 0.f;
#pragma unroll 8
      for (int m = 0; m < TW_ROWS; ++m) s += T[m * ld + c];
      out[c] = s;
    }
  };
  // ---- dW3 = dy^T tin, db3
  load(At, D, p.dy, p.ldy, D);
  load(Bt, TW_HID, p.tin, K3, K3);
  __syncthreads();
  for (int t = wave; t < (D / 32) * (K3 / 32); t += 4) tw_tn_tile<D, TW_HID>(At, Bt, 32 * (t / (K3 / 32)), 32 * (t % (K3 / 32)), pW3, LDW3, r, h);
  colsum(At, D, D, pb3);
  __syncthreads();
  if constexpr (XTRA) {  // dW3[:, 2D:] = dy^T extra
    load(Bt, TW_HID, p.extra, p.ldx, K3);
    __syncthreads();
    for (int t = wave; t < (D / 32) * (K3 / 32); t += 4)
      tw_tn_tile<D, TW_HID>(At, Bt, 32 * (t / (K3 / 32)), 32 * (t % (K3 / 32)), pW3 + K3, LDW3, r, h);
    __syncthreads();
  }
  // ---- dW2 = d_f^T h, db2
  load(At, D, p.d_f, D, D);
  load(Bt, TW_HID, p.h, TW_HID, TW_HID);
  __syncthreads();
  for (int t = wave; t < (D / 32) * (TW_HID / 32); t += 4) tw_tn_tile<D, TW_HID>(At, Bt, 32 * (t / (TW_HID / 32)), 32 * (t % (TW_HID / 32)), pW2, TW_HID, r, h);
  colsum(At, D, D, pb2);
  __syncthreads();
  // ---- dW1 = dh^T x, db1: K = F is small (8 at the BASELINE shapes): one hidden unit per thread
  load(Bt, TW_HID, p.dh, TW_HID, TW_HID);
  for (int i = threadIdx.x; i < TW_ROWS * F; i += 256) {
    const int m = i / F, f = i - m * F;
    Xt[i] = (row0 + m < p.B) ? p.feats[(row0 + m) * p.ldf + f] : 0.f;
  }
  __syncthreads();
  {
    const int u = threadIdx.x;  // 256 threads = 256 hidden units
    float sb = 0.f;
#pragma unroll 8
    for (int m = 0; m < TW_ROWS; ++m) sb += Bt[m * TW_HID + u];
    pb1[u] = sb;
    for (int f = 0; f < F; ++f) {
      float s = 0.f;
#pragma unroll 8
      for (int m = 0; m < TW_ROWS; ++m) s = fmaf(Bt[m * TW_HID + u], Xt[m * F + f], s);
      pW1[u * F + f] = s;
    }
  }
}

template <int DE8, bool XTRA>
__global__ __launch_bounds__(256) void tower_wgrad_kernel(const TowerWgradArgs p) { tower_wgrad_body<DE8, XTRA>(p); }
struct TowerWgradArgs2 { TowerWgradArgs t[2]; };
template <int DE8>
__global__ __launch_bounds__(256) void tower_wgrad_pair_kernel(const TowerWgradArgs2 q) { tower_wgrad_body<DE8, false>(q.t[blockIdx.y]); }

// out = sum over the blocks' partials, in a FIXED order (deterministic): 64 elements of the six tensors per 1024-thread
// workgroup, sixteen threads per element -- thread (q, t) adds blocks q, q + 16, ... with eight independent loads in
// flight, the sixteen slices are combined in order through LDS.  (One thread per element walking all the blocks was a
// chain of ~128 loads issued a few at a time: 32 us for 35 MB, 270-310 us next to the table sweep.)
__device__ __forceinline__ void tower_wgrad_reduce_body(const float* __restrict__ part, int n_blocks, int64_t part_floats,
                                                        int64_t n3, int64_t n2, int64_t n1, int64_t D, float* dW3,
                                                        float* dW2, float* dW1, float* db3, float* db2, float* db1) {
  __shared__ float sh[16][64];
  const int t = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + t;
  float s = 0.f;
  if (i < part_floats) {
    const float* src = part + i;
    int w = q;
    for (; w + 112 < n_blocks; w += 128) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(w + 16 * u) * part_floats];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; w < n_blocks; w += 16) s += src[(int64_t)w * part_floats];
  }
  sh[q][t] = s;
  __syncthreads();
  if (q != 0 || i >= part_floats) return;
  float tot = 0.f;
#pragma unroll
  for (int u = 0; u < 16; ++u) tot += sh[u][t];
  int64_t j = i;
  if (j < n3) { dW3[j] = tot; return; }
  j -= n3;
  if (j < n2) { dW2[j] = tot; return; }
  j -= n2;
  if (j < n1) { dW1[j] = tot; return; }
  j -= n1;
  if (j < D) { db3[j] = tot; return; }
  j -= D;
  if (j < D) { db2[j] = tot; return; }
  j -= D;
  db1[j] = tot;
}

__global__ __launch_bounds__(1024) void tower_wgrad_reduce_kernel(const float* __restrict__ part, int n_blocks, int64_t part_floats,
                                                                  int64_t n3, int64_t n2, int64_t n1, int64_t D, float* dW3,
                                                                  float* dW2, float* dW1, float* db3, float* db2, float* db1) {
  tower_wgrad_reduce_body(part, n_blocks, part_floats, n3, n2, n1, D, dW3, dW2, dW1, db3, db2, db1);
}
struct TowerReduceArgs2 {
  const float* part[2];
  int64_t part_floats[2], n3[2], n2[2], n1[2];
  float *dW3[2], *dW2[2], *dW1[2], *db3[2], *db2[2], *db1[2];
};
__global__ __launch_bounds__(1024) void tower_wgrad_reduce_pair_kernel(const TowerReduceArgs2 q, int n_blocks, int64_t D) {
  const int k = blockIdx.y;
  if ((int64_t)blockIdx.x * 64 >= q.part_floats[k]) return;  // (the grid is sized for the larger of the two partials)
  tower_wgrad_reduce_body(q.part[k], n_blocks, q.part_floats[k], q.n3[k], q.n2[k], q.n1[k], D, q.dW3[k], q.dW2[k], q.dW1[k], q.db3[k],
                          q.db2[k], q.db1[k]);
}

// 32-row tiles per workgroup of the forward / backward-data kernels: one (32 batch rows) up to B = 4096, else two.  Either
// way a tower of the BASELINE shapes is at most 128 workgroups -- HALF the chip, on purpose: in the overlapped train step
// the row plan's single 118 KB-LDS workgroup (plan_small_kernel, third stream) has to find a CU with that much LDS free
// while the towers run, and with 256 tower workgroups of 68 KB it could not start before the logits kernels (132 KB per
// CU) had come and gone -- 1.5 ms late, the step's tail 0.1 ms longer at the headline shape.  Below B = 4096 the 32-row
// form halves the towers' latency (B = 4096: 64 -> 128 workgroups; C2 step 1.216 -> 1.202 ms).
static int tower_row_tiles(int64_t B) {
  return B <= 4096 ? 1 : 2;
}
static bool tower_shape_ok(int64_t D, int64_t F, int64_t hidden, int64_t d_out) {
  return hidden == TW_HID && d_out == D && (D == 32 || D == 64 || D == 128) && F >= 1 && F <= TW_FMAX;
}
static inline bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace tt

using namespace tt;

extern "C" int tt_tower_supported(int64_t D, int64_t F, int64_t hidden, int64_t d_out) {
  return tower_shape_ok(D, F, hidden, d_out) ? 1 : 0;
}
extern "C" int tt_tower_x_supported(int64_t D, int64_t F, int64_t hidden, int64_t d_out, int64_t E) {
  return (tower_shape_ok(D, F, hidden, d_out) && (E == 0 || E == 2 * D)) ? 1 : 0;
}

extern "C" int tt_tower_fwd_x(const float* table, int64_t n_rows, const int64_t* ids, const float* feats, int64_t ldf,
                              int64_t B, int64_t D, int64_t F, int64_t hidden, const float* W1, const float* b1,
                              const float* W2, const float* b2, const float* W3, const float* b3, int64_t d_out,
                              const float* extra, int64_t ldx, int64_t E, float* y, int64_t ldy, float* h_out,
                              float* tin_out, int32_t* oob_flag, tt_stream_t stream) {
  if (!table || !ids || !feats || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !y || !h_out || !tin_out || (E > 0 && !extra))
    return fail_arg("tt_tower_fwd: null pointer");
  if (B <= 0 || n_rows <= 0 || ldf < F || ldy < d_out || E < 0 || (E > 0 && ldx < E)) return fail_arg("tt_tower_fwd: sizes");
  if (!tt_tower_x_supported(D, F, hidden, d_out, E) || ldy % 4 || !al16p(table) || !al16p(W2) || !al16p(W3) || !al16p(y) ||
      !al16p(h_out) || !al16p(tin_out) || (E > 0 && (ldx % 4 || !al16p(extra)))) {
    set_error("tt_tower_fwd: needs hidden = 256, D = d_out in {32, 64, 128}, F <= 64, extra width 0 or 2D, 16-B aligned operands");
    return TT_E_UNSUPPORTED;
  }
  TowerFwdArgs a{table, n_rows, ids, feats, ldf, B, F, W1, b1, W2, b2, W3, b3, y, ldy, h_out, tin_out, oob_flag, extra, ldx};
  hipStream_t st = S(stream);
  const int rt = tower_row_tiles(B);
  const int rows = 32 * rt;
  const unsigned grid = (unsigned)ceil_div(B, rows);
  const size_t lds = (size_t)(rows * (TW_HID + 4) + rows * (2 * D + 4) + rows * F) * sizeof(float);
#define TT_TWF1(E8, X, R)                                                                                                \
  {                                                                                                                      \
    if (lds > 64 * 1024) {                                                                                               \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tower_fwd_kernel<E8, X, R>),                     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
      if (e != hipSuccess) { set_error("tower_fwd_kernel: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; } \
    }                                                                                                                    \
    tower_fwd_kernel<E8, X, R><<<grid, 256, lds, st>>>(a);                                                               \
  }
#define TT_TWF(E8, X) { if (rt == 1) TT_TWF1(E8, X, 1) else TT_TWF1(E8, X, 2) }
  if (E == 0) {
    if (D == 32) TT_TWF(4, false) else if (D == 64) TT_TWF(8, false) else TT_TWF(16, false)
  } else {
    if (D == 32) TT_TWF(4, true) else if (D == 64) TT_TWF(8, true) else TT_TWF(16, true)
  }
#undef TT_TWF
#undef TT_TWF1
  return check_launch("tower_fwd_kernel");
}

extern "C" int tt_tower_fwd(const float* table, int64_t n_rows, const int64_t* ids, const float* feats, int64_t ldf,
                            int64_t B, int64_t D, int64_t F, int64_t hidden, const float* W1, const float* b1,
                            const float* W2, const float* b2, const float* W3, const float* b3, int64_t d_out, float* y,
                            int64_t ldy, float* h_out, float* tin_out, int32_t* oob_flag, tt_stream_t stream) {
  return tt_tower_fwd_x(table, n_rows, ids, feats, ldf, B, D, F, hidden, W1, b1, W2, b2, W3, b3, d_out, nullptr, 0, 0, y, ldy,
                        h_out, tin_out, oob_flag, stream);
}

extern "C" int tt_tower_bwd_data_x(const float* dy, int64_t ldy, int64_t B, int64_t D, int64_t hidden, const float* W2,
                                   const float* W3, const float* h, float* d_emb, int64_t ld_demb, float* d_f, float* dh,
                                   float* d_extra, int64_t ld_dx, int64_t E, tt_stream_t stream) {
  if (!dy || !W2 || !W3 || !h || !d_emb || !d_f || !dh || (E > 0 && !d_extra)) return fail_arg("tt_tower_bwd_data: null pointer");
  if (B <= 0 || ldy < D || ld_demb < D || E < 0 || (E > 0 && ld_dx < E)) return fail_arg("tt_tower_bwd_data: sizes");
  if (!tt_tower_x_supported(D, 1, hidden, D, E) || ldy % 4 || ld_demb % 4 || !al16p(dy) || !al16p(h) || !al16p(d_emb) ||
      !al16p(d_f) || !al16p(dh) || !al16p(W3) || (E > 0 && (ld_dx % 4 || !al16p(d_extra)))) {
    set_error("tt_tower_bwd_data: needs hidden = 256, D in {32, 64, 128}, extra width 0 or 2D, 16-B aligned operands");
    return TT_E_UNSUPPORTED;
  }
  TowerBwdArgs a{dy, ldy, B, W2, W3, h, d_emb, ld_demb, d_f, dh, d_extra, ld_dx};
  hipStream_t st = S(stream);
  const int rt = tower_row_tiles(B);
  const unsigned grid = (unsigned)ceil_div(B, 32 * rt);
#define TT_TWB(E8, X) { if (rt == 1) tower_bwd_kernel<E8, X, 1><<<grid, 256, 0, st>>>(a); else tower_bwd_kernel<E8, X, 2><<<grid, 256, 0, st>>>(a); }
  if (E == 0) {
    if (D == 32) TT_TWB(4, false) else if (D == 64) TT_TWB(8, false) else TT_TWB(16, false)
  } else {
    if (D == 32) TT_TWB(4, true) else if (D == 64) TT_TWB(8, true) else TT_TWB(16, true)
  }
#undef TT_TWB
  return check_launch("tower_bwd_kernel");
}

extern "C" int tt_tower_bwd_data(const float* dy, int64_t ldy, int64_t B, int64_t D, int64_t hidden, const float* W2,
                                 const float* W3, const float* h, float* d_emb, int64_t ld_demb, float* d_f, float* dh,
                                 tt_stream_t stream) {
  return tt_tower_bwd_data_x(dy, ldy, B, D, hidden, W2, W3, h, d_emb, ld_demb, d_f, dh, nullptr, 0, 0, stream);
}

static int64_t tower_part_floats(int64_t D, int64_t F, int64_t E) { return D * (2 * D + E) + D * TW_HID + TW_HID * F + 2 * D + TW_HID; }

extern "C" int64_t tt_tower_bwd_weights_x_workspace_bytes(int64_t B, int64_t D, int64_t F, int64_t hidden, int64_t E) {
  if (B <= 0 || !tt_tower_x_supported(D, F, hidden, D, E)) return 256;
  return round_up(ceil_div(B, TW_ROWS) * tower_part_floats(D, F, E) * (int64_t)sizeof(float), 256);
}
extern "C" int64_t tt_tower_bwd_weights_workspace_bytes(int64_t B, int64_t D, int64_t F, int64_t hidden) {
  return tt_tower_bwd_weights_x_workspace_bytes(B, D, F, hidden, 0);
}

extern "C" int tt_tower_bwd_weights_x(const float* dy, int64_t ldy, const float* tin, const float* d_f, const float* h,
                                      const float* dh, const float* feats, int64_t ldf, const float* extra, int64_t ldx,
                                      int64_t E, int64_t B, int64_t D, int64_t F, int64_t hidden, float* dW1, float* db1,
                                      float* dW2, float* db2, float* dW3, float* db3, void* ws, int64_t ws_bytes,
                                      tt_stream_t stream) {
  if (!dy || !tin || !d_f || !h || !dh || !feats || !dW1 || !db1 || !dW2 || !db2 || !dW3 || !db3 || !ws || (E > 0 && !extra))
    return fail_arg("tt_tower_bwd_weights: null pointer");
  if (B <= 0 || ldy < D || ldf < F || E < 0 || (E > 0 && ldx < E)) return fail_arg("tt_tower_bwd_weights: sizes");
  if (!tt_tower_x_supported(D, F, hidden, D, E) || ldy % 4 || !al16p(dy) || !al16p(tin) || !al16p(d_f) || !al16p(h) ||
      !al16p(dh) || (E > 0 && (ldx % 4 || !al16p(extra)))) {
    set_error("tt_tower_bwd_weights: needs hidden = 256, D in {32, 64, 128}, F <= 64, extra width 0 or 2D, 16-B aligned operands");
    return TT_E_UNSUPPORTED;
  }
  if (ws_bytes < tt_tower_bwd_weights_x_workspace_bytes(B, D, F, hidden, E)) { set_error("tt_tower_bwd_weights: workspace"); return TT_E_WORKSPACE; }
  TowerWgradArgs a{dy, ldy, tin, d_f, h, dh, feats, ldf, F, B, reinterpret_cast<float*>(ws), extra, ldx};
  hipStream_t st = S(stream);
  const unsigned grid = (unsigned)ceil_div(B, TW_ROWS);
  const size_t lds = (size_t)(TW_ROWS * D + TW_ROWS * TW_HID + TW_ROWS * F) * sizeof(float);
#define TT_TWG(E8, X)                                                                                                    \
  {                                                                                                                      \
    if (lds > 64 * 1024) {                                                                                               \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tower_wgrad_kernel<E8, X>),                      \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
      if (e != hipSuccess) { set_error("tower_wgrad_kernel: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; } \
    }                                                                                                                    \
    tower_wgrad_kernel<E8, X><<<grid, 256, lds, st>>>(a);                                                                \
  }
  if (E == 0) {
    if (D == 32) TT_TWG(4, false) else if (D == 64) TT_TWG(8, false) else TT_TWG(16, false)
  } else {
    if (D == 32) TT_TWG(4, true) else if (D == 64) TT_TWG(8, true) else TT_TWG(16, true)
  }
#undef TT_TWG
  int rc = check_launch("tower_wgrad_kernel");
  if (rc) return rc;
  const int64_t n3 = D * (2 * D + E), n2 = D * TW_HID, n1 = TW_HID * F, part = tower_part_floats(D, F, E);
  tower_wgrad_reduce_kernel<<<(unsigned)ceil_div(part, 64), 1024, 0, st>>>(reinterpret_cast<const float*>(ws), (int)grid, part, n3, n2,
                                                                           n1, D, dW3, dW2, dW1, db3, db2, db1);
  return check_launch("tower_wgrad_reduce_kernel");
}

extern "C" int tt_tower_bwd_weights(const float* dy, int64_t ldy, const float* tin, const float* d_f, const float* h,
                                    const float* dh, const float* feats, int64_t ldf, int64_t B, int64_t D, int64_t F,
                                    int64_t hidden, float* dW1, float* db1, float* dW2, float* db2, float* dW3, float* db3,
                                    void* ws, int64_t ws_bytes, tt_stream_t stream) {
  return tt_tower_bwd_weights_x(dy, ldy, tin, d_f, h, dh, feats, ldf, nullptr, 0, 0, B, D, F, hidden, dW1, db1, dW2, db2, dW3, db3,
                                ws, ws_bytes, stream);
}

// ---------------------------------------------------------------- both towers of the base model per launch
static int pair_shape_check(const char* who, int64_t B, int64_t D, int64_t hidden, int64_t F0, int64_t F1) {
  if (B <= 0) return fail_arg(who);
  if (!tower_shape_ok(D, F0, hidden, D) || !tower_shape_ok(D, F1, hidden, D)) {
    set_error("%s: needs hidden = 256, D in {32, 64, 128} (the same for both towers), F <= 64", who);
    return TT_E_UNSUPPORTED;
  }
  return 0;
}

extern "C" int tt_tower_fwd_pair(const tt_tower_fwd_side* sides, int64_t B, int64_t D, int64_t hidden, int32_t* oob_flag,
                                 tt_stream_t stream) {
  if (!sides) return fail_arg("tt_tower_fwd_pair: null pointer");
  int rc = pair_shape_check("tt_tower_fwd_pair", B, D, hidden, sides[0].F, sides[1].F);
  if (rc) return rc;
  TowerFwdArgs2 q{};
  int64_t Fmax = 0;
  for (int k = 0; k < 2; ++k) {
    const tt_tower_fwd_side& s = sides[k];
    if (!s.table || !s.ids || !s.feats || !s.W1 || !s.b1 || !s.W2 || !s.b2 || !s.W3 || !s.b3 || !s.y || !s.h_out || !s.tin_out)
      return fail_arg("tt_tower_fwd_pair: null pointer");
    if (s.n_rows <= 0 || s.ldf < s.F || s.ldy < D) return fail_arg("tt_tower_fwd_pair: sizes");
    if (s.ldy % 4 || !al16p(s.table) || !al16p(s.W2) || !al16p(s.W3) || !al16p(s.y) || !al16p(s.h_out) || !al16p(s.tin_out)) {
      set_error("tt_tower_fwd_pair: 16-B aligned operands");
      return TT_E_UNSUPPORTED;
    }
    q.t[k] = TowerFwdArgs{s.table, s.n_rows, s.ids, s.feats, s.ldf, B, s.F, s.W1, s.b1, s.W2, s.b2, s.W3, s.b3, s.y, s.ldy,
                          s.h_out, s.tin_out, oob_flag, nullptr, 0};
    Fmax = s.F > Fmax ? s.F : Fmax;
  }
  hipStream_t st = S(stream);
  const int rt = tower_row_tiles(B);
  const int rows = 32 * rt;
  const dim3 grid((unsigned)ceil_div(B, rows), 2);
  const size_t lds = (size_t)(rows * (TW_HID + 4) + rows * (2 * D + 4) + rows * Fmax) * sizeof(float);
#define TT_TWP1(E8, R)                                                                                                   \
  {                                                                                                                      \
    if (lds > 64 * 1024) {                                                                                               \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tower_fwd_pair_kernel<E8, R>),                   \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
      if (e != hipSuccess) { set_error("tower_fwd_pair_kernel: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; } \
    }                                                                                                                    \
    tower_fwd_pair_kernel<E8, R><<<grid, 256, lds, st>>>(q);                                                             \
  }
#define TT_TWP(E8) { if (rt == 1) TT_TWP1(E8, 1) else TT_TWP1(E8, 2) }
  if (D == 32) TT_TWP(4) else if (D == 64) TT_TWP(8) else TT_TWP(16)
#undef TT_TWP
#undef TT_TWP1
  return check_launch("tower_fwd_pair_kernel");
}

extern "C" int tt_tower_bwd_data_pair(const tt_tower_bwd_side* sides, int64_t B, int64_t D, int64_t hidden, tt_stream_t stream) {
  if (!sides) return fail_arg("tt_tower_bwd_data_pair: null pointer");
  int rc = pair_shape_check("tt_tower_bwd_data_pair", B, D, hidden, 1, 1);
  if (rc) return rc;
  TowerBwdArgs2 q{};
  for (int k = 0; k < 2; ++k) {
    const tt_tower_bwd_side& s = sides[k];
    if (!s.dy || !s.W2 || !s.W3 || !s.h || !s.d_emb || !s.d_f || !s.dh) return fail_arg("tt_tower_bwd_data_pair: null pointer");
    if (s.ldy < D || s.ld_demb < D) return fail_arg("tt_tower_bwd_data_pair: sizes");
    if (s.ldy % 4 || s.ld_demb % 4 || !al16p(s.dy) || !al16p(s.h) || !al16p(s.d_emb) || !al16p(s.d_f) || !al16p(s.dh) || !al16p(s.W3)) {
      set_error("tt_tower_bwd_data_pair: 16-B aligned operands");
      return TT_E_UNSUPPORTED;
    }
    q.t[k] = TowerBwdArgs{s.dy, s.ldy, B, s.W2, s.W3, s.h, s.d_emb, s.ld_demb, s.d_f, s.dh, nullptr, 0};
  }
  hipStream_t st = S(stream);
  const int rt = tower_row_tiles(B);
  const dim3 grid((unsigned)ceil_div(B, 32 * rt), 2);
#define TT_TBP(E8) { if (rt == 1) tower_bwd_pair_kernel<E8, 1><<<grid, 256, 0, st>>>(q); else tower_bwd_pair_kernel<E8, 2><<<grid, 256, 0, st>>>(q); }
  if (D == 32) TT_TBP(4) else if (D == 64) TT_TBP(8) else TT_TBP(16)
#undef TT_TBP
  return check_launch("tower_bwd_pair_kernel");
}

extern "C" int tt_tower_bwd_weights_pair(const tt_tower_wgrad_side* sides, int64_t B, int64_t D, int64_t hidden, tt_stream_t stream) {
  if (!sides) return fail_arg("tt_tower_bwd_weights_pair: null pointer");
  int rc = pair_shape_check("tt_tower_bwd_weights_pair", B, D, hidden, sides[0].F, sides[1].F);
  if (rc) return rc;
  TowerWgradArgs2 q{};
  TowerReduceArgs2 r{};
  int64_t Fmax = 0, part_max = 0;
  for (int k = 0; k < 2; ++k) {
    const tt_tower_wgrad_side& s = sides[k];
    if (!s.dy || !s.tin || !s.d_f || !s.h || !s.dh || !s.feats || !s.dW1 || !s.db1 || !s.dW2 || !s.db2 || !s.dW3 || !s.db3 || !s.ws)
      return fail_arg("tt_tower_bwd_weights_pair: null pointer");
    if (s.ldy < D || s.ldf < s.F) return fail_arg("tt_tower_bwd_weights_pair: sizes");
    if (s.ldy % 4 || !al16p(s.dy) || !al16p(s.tin) || !al16p(s.d_f) || !al16p(s.h) || !al16p(s.dh)) {
      set_error("tt_tower_bwd_weights_pair: 16-B aligned operands");
      return TT_E_UNSUPPORTED;
    }
    if (s.ws_bytes < tt_tower_bwd_weights_x_workspace_bytes(B, D, s.F, hidden, 0)) { set_error("tt_tower_bwd_weights_pair: workspace"); return TT_E_WORKSPACE; }
    q.t[k] = TowerWgradArgs{s.dy, s.ldy, s.tin, s.d_f, s.h, s.dh, s.feats, s.ldf, s.F, B, reinterpret_cast<float*>(s.ws), nullptr, 0};
    r.part[k] = reinterpret_cast<const float*>(s.ws);
    r.part_floats[k] = tower_part_floats(D, s.F, 0);
    r.n3[k] = D * 2 * D; r.n2[k] = D * TW_HID; r.n1[k] = TW_HID * s.F;
    r.dW3[k] = s.dW3; r.dW2[k] = s.dW2; r.dW1[k] = s.dW1; r.db3[k] = s.db3; r.db2[k] = s.db2; r.db1[k] = s.db1;
    Fmax = s.F > Fmax ? s.F : Fmax;
    part_max = r.part_floats[k] > part_max ? r.part_floats[k] : part_max;
  }
  hipStream_t st = S(stream);
  const unsigned blocks = (unsigned)ceil_div(B, TW_ROWS);
  const size_t lds = (size_t)(TW_ROWS * D + TW_ROWS * TW_HID + TW_ROWS * Fmax) * sizeof(float);
#define TT_TGP(E8)                                                                                                       \
  {                                                                                                                      \
    if (lds > 64 * 1024) {                                                                                               \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tower_wgrad_pair_kernel<E8>),                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
      if (e != hipSuccess) { set_error("tower_wgrad_pair_kernel: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; } \
    }                                                                                                                    \
    tower_wgrad_pair_kernel<E8><<<dim3(blocks, 2), 256, lds, st>>>(q);                                                   \
  }
  if (D == 32) TT_TGP(4) else if (D == 64) TT_TGP(8) else TT_TGP(16)
#undef TT_TGP
  rc = check_launch("tower_wgrad_pair_kernel");
  if (rc) return rc;
  tower_wgrad_reduce_pair_kernel<<<dim3((unsigned)ceil_div(part_max, 64), 2), 1024, 0, st>>>(r, (int)blocks, D);
  return check_launch("tower_wgrad_reduce_pair_kernel");
}
