// K4, last layer: the encoder consumes ONLY ROW 0 of its last attention layer
// (ref:src/user_history_encoder.py:113-116), so that layer never needs K and V of the
// other positions as matrices.  With X = the layer's input [H, D] of one sample,
// q0 = W_q X[0] + b_q and per head h (rows h*dh .. of W_k / W_v):
//
//   score_h[j] = scale * q0_h . (W_k,h X[j] + b_k,h) = scale * (W_k,h^T q0_h) . X[j] + const
//                                                     = scale * t_h . X[j]      (softmax drops the constant)
//   ctx0_h     = sum_j p_h[j] (W_v,h X[j] + b_v,h)   = W_v,h (sum_j p_h[j] X[j]) + b_v,h = W_v,h xbar_h + b_v,h
//
// i.e. the K / V projections of H rows (2*H*D*2D flop per sample) become two D-wide vectors
// per head (t_h, xbar_h) and the sample's X is read ONCE: 0.4 GFLOP + 105 MB instead of
// 13.4 GFLOP + 525 MB at B = 4096, H = 50, D = 128.  The backward has the same shape
// (d_xbar_h = W_v,h^T d_ctx0_h; dX[j] = sum_h p_h[j] d_xbar_h + ds_h[j] t_h; dt_h = sum_j ds_h[j] X[j]);
// the weight gradients are [D, B] x [B, D] products over the batch, one partial per 32 samples,
// added up in a fixed order.  The bias of K receives exactly zero (it only shifts every
// score of a head by the same amount; autograd produces rounding noise there).
//
// Launches: forward  enc_last_pre (q0, t) -> enc_last_main_fwd (probs, xbar) -> enc_last_post (ctx0, recent)
//           backward, data path     enc_last_bwd_a (d_ctx0, d_xbar) -> enc_last_main_bwd (dt, dX)
//                                   -> enc_last_bwd_b (dq0; dX[0] += W_q^T dq0)
//           backward, weights       enc_last_wgrad (dW_out, dW_v | dW_q, dW_k partials) -> enc_last_reduce
//           (tt_enc_last_bwd_data / _weights: the weight half feeds nothing but the optimiser, so the caller may
//           run it on another stream -- it used to be the tail of kernels a and b, on the step's critical path)
// pre / post / a / b take 32 samples per workgroup and run their [32, D] x [D, D] products (and the
// per-head [32, dh] x [dh, D] ones) on the matrix cores, operands straight from global memory /
// L2 (the weights are 64 KB each) and a [32, D] LDS image for the intermediate that feeds the
// next product; the main kernels are one workgroup per sample, HBM-bound on X / dX.
#include "common.hpp"

namespace tt {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int EL_MAXD = 128;   // columns lane and lane + 64 of the main kernels
constexpr int EL_MAXH = 64;    // one history position per lane
constexpr int EL_MAXHEADS = 16;
constexpr int EL_ROWS = 32;    // samples per workgroup of the pre / post / a / b kernels (half of a 32-row MFMA tile: these
                               // kernels are latency-bound chains of small products, 2x the workgroups beats full tiles)
constexpr int EL_IMG = 32;     // rows of an LDS image (one MFMA tile)

__device__ __forceinline__ float dot4(const float4 a, const float4 b, float acc) {
  acc = fmaf(a.x, b.x, acc);
  acc = fmaf(a.y, b.y, acc);
  acc = fmaf(a.z, b.z, acc);
  return fmaf(a.w, b.w, acc);
}

// y = acc + sum_c v[c] * W[i][c]   (W image with row stride D+4, v in LDS)
__device__ __forceinline__ float row_dot(const float* W, const float* v, int i, int D, float acc) {
  const float* w = W + i * (D + 4);
  for (int c = 0; c < D; c += 4)
    acc = dot4(*reinterpret_cast<const float4*>(w + c), *reinterpret_cast<const float4*>(v + c), acc);
  return acc;
}

__device__ __forceinline__ void fma4(float4& acc, float s, const float4 v) {
  acc.x = fmaf(s, v.x, acc.x);
  acc.y = fmaf(s, v.y, acc.y);
  acc.z = fmaf(s, v.z, acc.z);
  acc.w = fmaf(s, v.w, acc.w);
}

// ------------------------------------------------------------------ one 32x32 tile on the matrix cores
// acc[m][n] += sum_{k < K} (am A(m,k)) (bm B(k,n)) with v_mfma_f32_32x32x2_f32 (exact fp32).  `pa` points at the
// lane's row m of A (lane & 31), `pb` at the lane's column n of B:
//   KC  operand (k contiguous):  element k at p[k]        -- K % 4 == 0, p 16-byte aligned; one 16-B load feeds
//                                                             four MFMA steps
//   !KC operand (k strided):     element k at p[k * ld]   -- any K
// k -> (group g, lane half h, step c) = 8g + 4h + c on BOTH operands: a fixed summation order.
// am / bm are the lane's operand masks (1 or 0): a 32-wide tile may span several heads.
template <bool KC>
__device__ __forceinline__ float4 el_fetch(const float* p, int64_t ld, int k, int K) {
  // UNCONDITIONAL loads from clamped positions, masked afterwards: a guarded load sits in its own basic block and hipcc
  // waits for it before evaluating the next guard -- one exposed memory round trip per load (gemm.hip has the same note)
  float4 v;
  if constexpr (KC) {
    const int kc = k < K ? k : K - 4;
    v = *reinterpret_cast<const float4*>(p + kc);
    const float m = k < K ? 1.f : 0.f;
    v.x *= m; v.y *= m; v.z *= m; v.w *= m;
  } else {
    const int k0 = k + 0 < K ? k + 0 : K - 1, k1 = k + 1 < K ? k + 1 : K - 1, k2 = k + 2 < K ? k + 2 : K - 1, k3 = k + 3 < K ? k + 3 : K - 1;
    const float x0 = p[(int64_t)k0 * ld], x1 = p[(int64_t)k1 * ld], x2 = p[(int64_t)k2 * ld], x3 = p[(int64_t)k3 * ld];
    v.x = k + 0 < K ? x0 : 0.f;
    v.y = k + 1 < K ? x1 : 0.f;
    v.z = k + 2 < K ? x2 : 0.f;
    v.w = k + 3 < K ? x3 : 0.f;
  }
  return v;
}
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void mma_tile(f32x16& acc, const float* pa, int64_t lda, float am, const float* pb,
                                         int64_t ldb, float bm, int K) {
  // operands come straight from global memory / L2 (or LDS): FOUR k-groups (32 k) are requested ahead of the MFMAs that
  // consume them -- with one group ahead every group paid most of an L2 round trip (4 MFMAs = 256 cycles of cover)
  const int h4 = ((threadIdx.x & 63) >> 5) * 4;
  float4 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = el_fetch<A_KC>(pa, lda, 8 * i + h4, K);
    b[i] = el_fetch<B_KC>(pb, ldb, 8 * i + h4, K);
  }
  for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (k0 + 8 * i < K) {  // (wave-uniform)
        const float4 ac = a[i], bc = b[i];
        a[i] = el_fetch<A_KC>(pa, lda, k0 + 32 + 8 * i + h4, K);
        b[i] = el_fetch<B_KC>(pb, ldb, k0 + 32 + 8 * i + h4, K);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac.x * am, bc.x * bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac.y * am, bc.y * bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac.z * am, bc.z * bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac.w * am, bc.w * bm, acc, 0, 0, 0);
      }
    }
  }
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
// row of accumulator register r within the tile (C/D layout of the 32x32 MFMAs); column = lane & 31
__device__ __forceinline__ int acc_row(int r) { return (r & 3) + 8 * (r >> 2) + 4 * ((threadIdx.x & 63) >> 5); }

struct ElLane {
  int w, li;  // wave, lane & 31
  __device__ ElLane() : w(threadIdx.x >> 6), li(threadIdx.x & 31) {}
};

}  // namespace

// Arguments shared by the four small kernels.  When the layer in front of this one hands over its attention CONTEXT
// instead of its output (x = c W_o^T + b_o never formed), the CALLER composes the weights -- W_in W_o, W_in b_o + b_in:
// q, k, v are linear in x -- and these kernels run on c unchanged (ops.HistoryEncoder; round 4 carried W_o through the
// kernels instead: 3.5x the matrix work here for the same result).
struct ElArgs {
  const float* x;      // [B*H, D]: this layer's input (or the previous layer's context, with composed w_in / b_in)
  const float* w_in;   // [3D, D]
  const float* b_in;   // [3D]
  const float* w_out;  // [D, D]
  const float* b_out;  // [D]
  int64_t B;
  int H, D, heads;
  // saved by the forward, read by the backward
  float* q0;     // [B, D]
  float* t;      // [B, heads, D]   W_k,h^T q0_h
  float* xbar;   // [B, heads, D]
  float* ctx0;   // [B, D]
  float* recent; int64_t ld_recent;            // forward output [B, D]
  const float* d_recent; int64_t ld_dr;        // backward input
  float* d_ctx0;  // backward scratch [B, D]: kernel a's first product, read again by the weight kernel
  float* dq0;     // backward scratch [B, D]: kernel b's first product, likewise
  float* d_xbar;  // backward scratch [B, heads, D]: what the main backward consumes
  float* dt;      // backward scratch [B, heads, D]: what the main backward produces
  float* dx;      // backward output [B*H, D]
  float* part_a;  // per-workgroup partial sums, backward a
  float* part_b;  // per-workgroup partial sums, backward b
};

// floats per workgroup in the partial buffers:  a = [dW_out | dW_v | db_out | db_v]      b = [dW_q | dW_k | db_q]
__host__ __device__ inline int64_t el_part_a(int64_t D) { return 2 * D * D + 2 * D; }
__host__ __device__ inline int64_t el_part_b(int64_t D) { return 2 * D * D + D; }

// the [32, D] result tile `acc` of column tile ct -> LDS image (row stride ldq) and / or global rows (row stride ldg)
#define EL_TILE_COLS(ct) const int col = 32 * (ct) + L.li, nc = col < D ? col : D - 1

// ------------------------------------------------------------------ forward: q0, t
__global__ __launch_bounds__(256, 1) void enc_last_pre_kernel(const ElArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int D = p.D, H = p.H, heads = p.heads;
  const int ldq = D + 4, dh = D / heads, NT = (D + 31) / 32;
  float* qs = reinterpret_cast<float*>(smem_raw);  // [32][ldq] q0
  const ElLane L;
  const int64_t b0 = (int64_t)blockIdx.x * EL_ROWS;
  const int nb = (int)((p.B - b0) < EL_ROWS ? (p.B - b0) : EL_ROWS);
  const int rowA = L.li < nb ? L.li : nb - 1;
  for (int ct = L.w; ct < NT; ct += 4) {  // q0 = X0 W_q^T + b_q
    EL_TILE_COLS(ct);
    f32x16 acc = zero16();
    mma_tile<true, true>(acc, p.x + (b0 + rowA) * H * D, 0, 1.f, p.w_in + (int64_t)nc * D, 0, 1.f, D);
    const float bias = p.b_in[nc];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r);
      const float v = acc[r] + bias;
      if (col < D) {
        qs[row * ldq + col] = v;
        if (row < nb) p.q0[(b0 + row) * D + col] = v;
      }
    }
  }
  __syncthreads();
  for (int u = L.w; u < heads * NT; u += 4) {  // t_h = q0_h W_k,h
    const int hh = u / NT, ct = u % NT;
    EL_TILE_COLS(ct);
    f32x16 acc = zero16();
    mma_tile<true, false>(acc, qs + L.li * ldq + hh * dh, 0, 1.f, p.w_in + (int64_t)(D + hh * dh) * D + nc, D, 1.f, dh);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r);
      if (col < D && row < nb) p.t[((b0 + row) * heads + hh) * D + col] = acc[r];
    }
  }
}

// ------------------------------------------------------------------ forward: scores, softmax, xbar
// one workgroup per sample, wave w takes heads w, w+4, ...; lane = history position for the
// scores, lane = column for the weighted row sum
__global__ __launch_bounds__(256) void enc_last_main_fwd_kernel(const float* __restrict__ x, int H, int D, int heads,
                                                                const float* __restrict__ t, float* __restrict__ probs,
                                                                float* __restrict__ xbar) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int ld = D + 4, q = D / 4;
  float* Xs = reinterpret_cast<float*>(smem_raw);  // [H][ld]
  float* ts = Xs + H * ld;                         // [heads][D]
  const int64_t b = blockIdx.x;
  const float* xb = x + b * H * D;
  for (int f = threadIdx.x; f < H * q; f += 256) {
    const int r = f / q, c4 = f % q;
    *reinterpret_cast<float4*>(Xs + r * ld + 4 * c4) = *reinterpret_cast<const float4*>(xb + (int64_t)r * D + 4 * c4);
  }
  for (int f = threadIdx.x; f < heads * q; f += 256)
    *reinterpret_cast<float4*>(ts + 4 * f) = *reinterpret_cast<const float4*>(t + b * heads * D + 4 * f);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float scale = 1.0f / sqrtf((float)(D / heads));
  for (int h = wave; h < heads; h += 4) {
    float sc = -3.0e38f;
    if (lane < H) sc = scale * row_dot(Xs, ts + h * D, lane, D, 0.f);
    const float mx = wave_max(sc);
    const float e = (lane < H) ? __expf(sc - mx) : 0.f;
    const float l = wave_sum(e);
    const float pj = e / l;
    if (lane < H) probs[(b * heads + h) * H + lane] = pj;
    {  // xbar_h[c] = sum_j p_j X[j][c], columns lane and lane + 64 (D <= 128)
      const int c0 = lane < D ? lane : 0, c1 = lane + 64 < D ? lane + 64 : 0;
      float a0 = 0.f, a1 = 0.f;
      for (int j = 0; j < H; ++j) {
        const float pb = __shfl(pj, j, 64);
        a0 = fmaf(pb, Xs[j * ld + c0], a0);
        a1 = fmaf(pb, Xs[j * ld + c1], a1);
      }
      if (lane < D) xbar[(b * heads + h) * D + lane] = a0;
      if (lane + 64 < D) xbar[(b * heads + h) * D + lane + 64] = a1;
    }
  }
}

// ------------------------------------------------------------------ forward: ctx0, recent
// ctx0[b, i] = W_v[i] . xbar[b, head(i)] + b_v[i] ;  recent[b] = W_out ctx0[b] + b_out
__global__ __launch_bounds__(256, 1) void enc_last_post_kernel(const ElArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int D = p.D, heads = p.heads;
  const int ldq = D + 4, dh = D / heads, NT = (D + 31) / 32;
  float* cs = reinterpret_cast<float*>(smem_raw);  // [32][ldq] ctx0
  const ElLane L;
  const int64_t b0 = (int64_t)blockIdx.x * EL_ROWS;
  const int nb = (int)((p.B - b0) < EL_ROWS ? (p.B - b0) : EL_ROWS);
  const int rowA = L.li < nb ? L.li : nb - 1;
  for (int ct = L.w; ct < NT; ct += 4) {
    EL_TILE_COLS(ct);
    const int h_lo = (32 * ct) / dh, h_hi = ((32 * ct + 31 < D ? 32 * ct + 31 : D - 1)) / dh;
    f32x16 acc = zero16();
    for (int hh = h_lo; hh <= h_hi; ++hh) {
      const float bm = nc / dh == hh ? 1.f : 0.f;
      mma_tile<true, true>(acc, p.xbar + ((b0 + rowA) * heads + hh) * D, 0, 1.f, p.w_in + (int64_t)(2 * D + nc) * D, 0, bm, D);
    }
    const float bias = p.b_in[2 * D + nc];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r);
      const float v = acc[r] + bias;
      if (col < D) {
        cs[row * ldq + col] = v;
        if (row < nb) p.ctx0[(b0 + row) * D + col] = v;
      }
    }
  }
  __syncthreads();
  for (int ct = L.w; ct < NT; ct += 4) {
    EL_TILE_COLS(ct);
    f32x16 acc = zero16();
    mma_tile<true, true>(acc, cs + L.li * ldq, 0, 1.f, p.w_out + (int64_t)nc * D, 0, 1.f, D);
    const float bias = p.b_out[nc];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r);
      if (col < D && row < nb) p.recent[(b0 + row) * p.ld_recent + col] = acc[r] + bias;
    }
  }
}

// ------------------------------------------------------------------ backward a: d_ctx0, d_xbar
__global__ __launch_bounds__(256, 1) void enc_last_bwd_a_kernel(const ElArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int D = p.D, heads = p.heads;
  const int ldq = D + 4, dh = D / heads, NT = (D + 31) / 32;
  float* dc = reinterpret_cast<float*>(smem_raw);  // [32][ldq] d_ctx0, rows past the batch = 0
  const ElLane L;
  const int64_t b0 = (int64_t)blockIdx.x * EL_ROWS;
  const int nb = (int)((p.B - b0) < EL_ROWS ? (p.B - b0) : EL_ROWS);
  const int rowA = L.li < nb ? L.li : nb - 1;
  // d_ctx0[b][n] = sum_k d_recent[b][k] W_out[k][n]
  for (int ct = L.w; ct < NT; ct += 4) {
    EL_TILE_COLS(ct);
    f32x16 acc = zero16();
    mma_tile<true, false>(acc, p.d_recent + (b0 + rowA) * p.ld_dr, 0, 1.f, p.w_out + nc, D, 1.f, D);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r);
      if (col < D) dc[row * ldq + col] = row < nb ? acc[r] : 0.f;
      if (col < D && row < nb) p.d_ctx0[(b0 + row) * D + col] = acc[r];
    }
  }
  __syncthreads();
  // d_xbar[b][h][n] = sum_{k in head h} d_ctx0[b][k] W_v[k][n]
  for (int u = L.w; u < heads * NT; u += 4) {
    const int hh = u / NT, ct = u % NT;
    EL_TILE_COLS(ct);
    f32x16 acc = zero16();
    mma_tile<true, false>(acc, dc + L.li * ldq + hh * dh, 0, 1.f, p.w_in + (int64_t)(2 * D + hh * dh) * D + nc, D, 1.f, dh);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r);
      if (col < D && row < nb) p.d_xbar[((b0 + row) * heads + hh) * D + col] = acc[r];
    }
  }
}

// ------------------------------------------------------------------ backward main: dt, dX
__global__ __launch_bounds__(256) void enc_last_main_bwd_kernel(const float* __restrict__ x, int H, int D, int heads,
                                                                const float* __restrict__ t, const float* __restrict__ probs,
                                                                const float* __restrict__ d_xbar, float* __restrict__ dt,
                                                                float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int ld = D + 4, q = D / 4;
  float* Xs = reinterpret_cast<float*>(smem_raw);  // [H][ld]
  float* ts = Xs + H * ld;                         // [heads][D]
  float* gs = ts + heads * D;                      // [heads][D] d_xbar
  float* pc = gs + heads * D;                      // [heads][64] p
  float* sc = pc + heads * 64;                     // [heads][64] scale * ds
  const int64_t b = blockIdx.x;
  const float* xb = x + b * H * D;
  for (int f = threadIdx.x; f < H * q; f += 256) {
    const int r = f / q, c4 = f % q;
    *reinterpret_cast<float4*>(Xs + r * ld + 4 * c4) = *reinterpret_cast<const float4*>(xb + (int64_t)r * D + 4 * c4);
  }
  for (int f = threadIdx.x; f < heads * q; f += 256) {
    *reinterpret_cast<float4*>(ts + 4 * f) = *reinterpret_cast<const float4*>(t + b * heads * D + 4 * f);
    *reinterpret_cast<float4*>(gs + 4 * f) = *reinterpret_cast<const float4*>(d_xbar + b * heads * D + 4 * f);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float scale = 1.0f / sqrtf((float)(D / heads));
  for (int h = wave; h < heads; h += 4) {
    float pj = 0.f, dp = 0.f;
    if (lane < H) {
      pj = probs[(b * heads + h) * H + lane];
      dp = row_dot(Xs, gs + h * D, lane, D, 0.f);
    }
    const float delta = wave_sum(pj * dp);
    const float ds = scale * pj * (dp - delta);
    pc[h * 64 + lane] = pj;
    sc[h * 64 + lane] = ds;
    {  // dt_h[c] = sum_j (scale ds_j) X[j][c]
      const int c0 = lane < D ? lane : 0, c1 = lane + 64 < D ? lane + 64 : 0;
      float a0 = 0.f, a1 = 0.f;
      for (int j = 0; j < H; ++j) {
        const float db = __shfl(ds, j, 64);
        a0 = fmaf(db, Xs[j * ld + c0], a0);
        a1 = fmaf(db, Xs[j * ld + c1], a1);
      }
      if (lane < D) dt[(b * heads + h) * D + lane] = a0;
      if (lane + 64 < D) dt[(b * heads + h) * D + lane + 64] = a1;
    }
  }
  __syncthreads();
  float* dxb = dx + b * H * D;
  for (int f = threadIdx.x; f < H * q; f += 256) {
    const int j = f / q, c4 = f % q;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int h = 0; h < heads; ++h) {
      fma4(acc, pc[h * 64 + j], *reinterpret_cast<const float4*>(gs + h * D + 4 * c4));
      fma4(acc, sc[h * 64 + j], *reinterpret_cast<const float4*>(ts + h * D + 4 * c4));
    }
    *reinterpret_cast<float4*>(dxb + (int64_t)j * D + 4 * c4) = acc;
  }
}

// ------------------------------------------------------------------ backward b: dq0 -> dX[0]
__global__ __launch_bounds__(256, 1) void enc_last_bwd_b_kernel(const ElArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int D = p.D, H = p.H, heads = p.heads;
  const int ldq = D + 4, dh = D / heads, NT = (D + 31) / 32;
  float* dq = reinterpret_cast<float*>(smem_raw);  // [32][ldq] dq0, rows past the batch = 0
  const ElLane L;
  const int64_t b0 = (int64_t)blockIdx.x * EL_ROWS;
  const int nb = (int)((p.B - b0) < EL_ROWS ? (p.B - b0) : EL_ROWS);
  const int rowA = L.li < nb ? L.li : nb - 1;
  // dq0[b][n] = W_k[n] . dt[b][head(n)]
  for (int ct = L.w; ct < NT; ct += 4) {
    EL_TILE_COLS(ct);
    const int h_lo = (32 * ct) / dh, h_hi = ((32 * ct + 31 < D ? 32 * ct + 31 : D - 1)) / dh;
    f32x16 acc = zero16();
    for (int hh = h_lo; hh <= h_hi; ++hh) {
      const float bm = nc / dh == hh ? 1.f : 0.f;
      mma_tile<true, true>(acc, p.dt + ((b0 + rowA) * heads + hh) * D, 0, 1.f, p.w_in + (int64_t)(D + nc) * D, 0, bm, D);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r);
      if (col < D) dq[row * ldq + col] = row < nb ? acc[r] : 0.f;
      if (col < D && row < nb) p.dq0[(b0 + row) * D + col] = acc[r];
    }
  }
  __syncthreads();
  // dX0[b][n] = sum_k dq0[b][k] W_q[k][n]:  added to row 0 of the sample's dx
  for (int ct = L.w; ct < NT; ct += 4) {
    EL_TILE_COLS(ct);
    f32x16 acc = zero16();
    mma_tile<true, false>(acc, dq + L.li * ldq, 0, 1.f, p.w_in + nc, D, 1.f, D);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r);
      if (col < D && row < nb) p.dx[(b0 + row) * H * D + col] += acc[r];
    }
  }
}

// ------------------------------------------------------------------ backward, weights: the [D, 32] x [32, D] products
// One workgroup per 32 samples and HALF: blockIdx.y = 0 writes the partials [dW_out | dW_v | db_out | db_v], 1 writes
// [dW_q | dW_k | db_q].  The same products in the same order as when they were the tails of kernels a and b (the
// operands d_ctx0 / dq0 now come from the [B, D] scratch those kernels leave behind instead of their LDS images):
// bit-identical gradients, off the data path.
__global__ __launch_bounds__(256, 1) void enc_last_wgrad_kernel(const ElArgs p) {
  const int D = p.D, H = p.H, heads = p.heads;
  const int dh = D / heads, NT = (D + 31) / 32;
  const ElLane L;
  const int64_t b0 = (int64_t)blockIdx.x * EL_ROWS;
  const int nb = (int)((p.B - b0) < EL_ROWS ? (p.B - b0) : EL_ROWS);
  const int64_t DD = (int64_t)D * D;
  if (blockIdx.y == 0) {
    // dW_out[m][n] = sum_b d_recent[b][m] ctx0[b][n] ;  dW_v[m][n] = sum_b d_ctx0[b][m] xbar[b][head(m)][n]
    float* mine = p.part_a + (int64_t)blockIdx.x * el_part_a(D);
    for (int u = L.w; u < NT * NT; u += 4) {
      const int rt = u / NT, ct = u % NT;
      const int mrow = 32 * rt + L.li, mc = mrow < D ? mrow : D - 1;
      EL_TILE_COLS(ct);
      f32x16 acc = zero16();
      mma_tile<false, false>(acc, p.d_recent + b0 * p.ld_dr + mc, p.ld_dr, 1.f, p.ctx0 + b0 * D + nc, D, 1.f, nb);
      f32x16 acv = zero16();
      const int h_lo = (32 * rt) / dh, h_hi = ((32 * rt + 31 < D ? 32 * rt + 31 : D - 1)) / dh;
      for (int hh = h_lo; hh <= h_hi; ++hh)
        mma_tile<false, false>(acv, p.d_ctx0 + b0 * D + mc, D, mc / dh == hh ? 1.f : 0.f, p.xbar + (b0 * heads + hh) * D + nc,
                               (int64_t)heads * D, 1.f, nb);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * rt + acc_row(r);
        if (row < D && col < D) {
          mine[row * D + col] = acc[r];
          mine[DD + row * D + col] = acv[r];
        }
      }
    }
    if ((int)threadIdx.x < D) {
      float so = 0.f, sv = 0.f;
      for (int b = 0; b < nb; ++b) {
        so += p.d_recent[(b0 + b) * p.ld_dr + threadIdx.x];
        sv += p.d_ctx0[(b0 + b) * D + threadIdx.x];
      }
      mine[2 * DD + threadIdx.x] = so;
      mine[2 * DD + D + threadIdx.x] = sv;
    }
  } else {
    // dW_q[m][n] = sum_b dq0[b][m] X0[b][n] ;  dW_k[m][n] = sum_b q0[b][m] dt[b][head(m)][n]
    float* mine = p.part_b + (int64_t)blockIdx.x * el_part_b(D);
    for (int u = L.w; u < NT * NT; u += 4) {
      const int rt = u / NT, ct = u % NT;
      const int mrow = 32 * rt + L.li, mc = mrow < D ? mrow : D - 1;
      EL_TILE_COLS(ct);
      f32x16 acc = zero16();
      mma_tile<false, false>(acc, p.dq0 + b0 * D + mc, D, 1.f, p.x + b0 * H * D + nc, (int64_t)H * D, 1.f, nb);
      f32x16 ack = zero16();
      const int h_lo = (32 * rt) / dh, h_hi = ((32 * rt + 31 < D ? 32 * rt + 31 : D - 1)) / dh;
      for (int hh = h_lo; hh <= h_hi; ++hh) {
        const float am = mc / dh == hh ? 1.f : 0.f;
        mma_tile<false, false>(ack, p.q0 + b0 * D + mc, D, am, p.dt + (b0 * heads + hh) * D + nc, (int64_t)heads * D, 1.f, nb);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * rt + acc_row(r);
        if (row < D && col < D) {
          mine[row * D + col] = acc[r];
          mine[DD + row * D + col] = ack[r];
        }
      }
    }
    if ((int)threadIdx.x < D) {
      float sq = 0.f;
      for (int b = 0; b < nb; ++b) sq += p.dq0[(b0 + b) * D + threadIdx.x];
      mine[2 * DD + threadIdx.x] = sq;
    }
  }
}

// ------------------------------------------------------------------ reduce the partials (fixed order)
// dW_in [3D, D] = [dW_q | dW_k | dW_v], db_in [3D] = [db_q | 0 | db_v], dW_out [D, D], db_out [D]
__device__ __forceinline__ float el_sum_parts(const float* src, int64_t stride, int n) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four interleaved chains: independent loads in flight
  int w = 0;
  for (; w + 3 < n; w += 4) {
    s0 += src[(int64_t)w * stride];
    s1 += src[(int64_t)(w + 1) * stride];
    s2 += src[(int64_t)(w + 2) * stride];
    s3 += src[(int64_t)(w + 3) * stride];
  }
  for (; w < n; ++w) s0 += src[(int64_t)w * stride];
  return (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void enc_last_reduce_kernel(const float* __restrict__ part_a, const float* __restrict__ part_b,
                                                              int n, int D, float* __restrict__ dW_in, float* __restrict__ db_in,
                                                              float* __restrict__ dW_out, float* __restrict__ db_out) {
  const int DD = D * D;
  const int64_t sa = el_part_a(D), sb = el_part_b(D);
  const int total = 4 * DD + 3 * D;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    if (e < 2 * DD) {  // dW_q | dW_k
      dW_in[e] = el_sum_parts(part_b + e, sb, n);
    } else if (e < 3 * DD) {  // dW_v
      dW_in[e] = el_sum_parts(part_a + DD + (e - 2 * DD), sa, n);
    } else if (e < 4 * DD) {  // dW_out
      dW_out[e - 3 * DD] = el_sum_parts(part_a + (e - 3 * DD), sa, n);
    } else {
      const int k = e - 4 * DD, which = k / D, i = k % D;
      if (which == 0) {  // db_q, and the K third: exactly zero
        db_in[i] = el_sum_parts(part_b + 2 * DD + i, sb, n);
        db_in[D + i] = 0.f;
      } else if (which == 1) {
        db_in[2 * D + i] = el_sum_parts(part_a + 2 * DD + D + i, sa, n);
      } else {
        db_out[i] = el_sum_parts(part_a + 2 * DD + i, sa, n);
      }
    }
  }
}

namespace {

template <typename K>
int el_opt_in(K kernel, size_t lds, const char* name) {
  if (lds <= 64 * 1024) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e)); return (int)e; }
  return 0;
}

int64_t el_groups(int64_t B) { return ceil_div(B, EL_ROWS); }

bool el_aligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// LDS of the small kernels: one [32][D+4] image
size_t el_small_lds(int64_t D) { return (size_t)EL_IMG * (D + 4) * sizeof(float); }

}  // namespace
}  // namespace tt

using namespace tt;

extern "C" int tt_enc_last_supported(int64_t H, int64_t D, int64_t heads) {
  return (H >= 1 && H <= EL_MAXH && D >= 4 && D <= EL_MAXD && D % 4 == 0 && heads >= 1 && heads <= EL_MAXHEADS &&
          D % heads == 0 && (D / heads) % 4 == 0) ? 1 : 0;
}

extern "C" int tt_enc_last_fwd(const float* x, int64_t B, int64_t H, int64_t D, int64_t heads, const float* w_in,
                               const float* b_in, const float* w_out, const float* b_out, float* recent, int64_t ld_recent,
                               float* q0, float* t, float* probs, float* xbar, float* ctx0, tt_stream_t stream) {
  if (!x || !w_in || !b_in || !w_out || !b_out || !recent || !q0 || !t || !probs || !xbar || !ctx0)
    return fail_arg("tt_enc_last_fwd: null pointer");
  if (B < 0 || !tt_enc_last_supported(H, D, heads) || ld_recent < D) {
    set_error("tt_enc_last_fwd: shape outside H <= 64, D <= 128, D %% 4 == 0, head width %% 4 == 0 (H = %lld, D = %lld, heads = %lld)",
              (long long)H, (long long)D, (long long)heads);
    return TT_E_UNSUPPORTED;
  }
  if (!el_aligned(x) || !el_aligned(w_in) || !el_aligned(w_out) || !el_aligned(t) || !el_aligned(xbar))
    return fail_arg("tt_enc_last_fwd: x, w_in, w_out, t, xbar must be 16-byte aligned");
  if (B == 0) return 0;
  hipStream_t st = S(stream);
  const unsigned G = (unsigned)el_groups(B);
  ElArgs a{};
  a.x = x; a.w_in = w_in; a.b_in = b_in; a.w_out = w_out; a.b_out = b_out;
  a.B = B; a.H = (int)H; a.D = (int)D; a.heads = (int)heads;
  a.q0 = q0; a.t = t; a.xbar = xbar; a.ctx0 = ctx0;
  a.recent = recent; a.ld_recent = ld_recent;
  int rc;
  enc_last_pre_kernel<<<G, 256, el_small_lds(D), st>>>(a);
  if ((rc = check_launch("enc_last_pre_kernel"))) return rc;
  {
    const size_t lds = ((size_t)H * (D + 4) + heads * D) * sizeof(float);
    if ((rc = el_opt_in(enc_last_main_fwd_kernel, lds, "enc_last_main_fwd_kernel"))) return rc;
    ProfScope prof("enc_last_main_fwd_kernel", st);
    enc_last_main_fwd_kernel<<<(unsigned)B, 256, lds, st>>>(x, (int)H, (int)D, (int)heads, t, probs, xbar);
    if ((rc = check_launch("enc_last_main_fwd_kernel"))) return rc;
  }
  enc_last_post_kernel<<<G, 256, el_small_lds(D), st>>>(a);
  return check_launch("enc_last_post_kernel");
}

extern "C" int64_t tt_enc_last_bwd_workspace_bytes(int64_t B, int64_t H, int64_t D, int64_t heads) {
  if (B <= 0 || !tt_enc_last_supported(H, D, heads)) return 0;
  const int64_t G = el_groups(B);
  // d_xbar [B, heads, D] | dt [B, heads, D] | d_ctx0 [B, D] | dq0 [B, D] | partials a | partials b
  return round_up(B * heads * D * 4, 256) * 2 + round_up(B * D * 4, 256) * 2 + round_up(G * el_part_a(D) * 4, 256) +
         round_up(G * el_part_b(D) * 4, 256);
}

namespace {
// the workspace, carved the same way by the data half and the weight half
void el_carve(ElArgs& a, void* ws, int64_t B, int64_t D, int64_t heads) {
  const int64_t G = el_groups(B);
  Carver cv(ws);
  a.d_xbar = cv.take<float>(B * heads * D);
  a.dt = cv.take<float>(B * heads * D);
  a.d_ctx0 = cv.take<float>(B * D);
  a.dq0 = cv.take<float>(B * D);
  a.part_a = cv.take<float>(G * el_part_a(D));
  a.part_b = cv.take<float>(G * el_part_b(D));
}
}  // namespace

extern "C" int tt_enc_last_bwd_data(const float* x, int64_t B, int64_t H, int64_t D, int64_t heads, const float* w_in,
                                    const float* w_out, const float* d_recent, int64_t ld_dr, const float* t,
                                    const float* probs, float* dx, void* ws, int64_t ws_bytes, tt_stream_t stream) {
  if (!x || !w_in || !w_out || !d_recent || !t || !probs || !dx) return fail_arg("tt_enc_last_bwd_data: null pointer");
  if (B < 0 || !tt_enc_last_supported(H, D, heads) || ld_dr < D) {
    set_error("tt_enc_last_bwd: shape outside H <= 64, D <= 128, D %% 4 == 0, head width %% 4 == 0");
    return TT_E_UNSUPPORTED;
  }
  if (!el_aligned(x) || !el_aligned(w_in) || !el_aligned(w_out) || !el_aligned(t) || !el_aligned(dx) || !el_aligned(ws) ||
      !el_aligned(d_recent) || (ld_dr % 4) != 0)
    return fail_arg("tt_enc_last_bwd: x, dx, w_in, w_out, t, xbar, d_recent, ws must be 16-byte aligned, ld_dr % 4 == 0");
  if (B == 0) return 0;
  if (!ws || ws_bytes < tt_enc_last_bwd_workspace_bytes(B, H, D, heads)) return fail_arg("tt_enc_last_bwd: workspace too small");
  hipStream_t st = S(stream);
  const int64_t G = el_groups(B);
  ElArgs a{};
  a.x = x; a.w_in = w_in; a.w_out = w_out;
  a.B = B; a.H = (int)H; a.D = (int)D; a.heads = (int)heads;
  a.t = const_cast<float*>(t);
  a.d_recent = d_recent; a.ld_dr = ld_dr;
  el_carve(a, ws, B, D, heads);
  a.dx = dx;
  int rc;
  enc_last_bwd_a_kernel<<<(unsigned)G, 256, el_small_lds(D), st>>>(a);
  if ((rc = check_launch("enc_last_bwd_a_kernel"))) return rc;
  {
    const size_t lds = ((size_t)H * (D + 4) + 2 * heads * D + 2 * heads * 64) * sizeof(float);
    if ((rc = el_opt_in(enc_last_main_bwd_kernel, lds, "enc_last_main_bwd_kernel"))) return rc;
    ProfScope prof("enc_last_main_bwd_kernel", st);
    enc_last_main_bwd_kernel<<<(unsigned)B, 256, lds, st>>>(x, (int)H, (int)D, (int)heads, t, probs, a.d_xbar, a.dt, dx);
    if ((rc = check_launch("enc_last_main_bwd_kernel"))) return rc;
  }
  enc_last_bwd_b_kernel<<<(unsigned)G, 256, el_small_lds(D), st>>>(a);
  return check_launch("enc_last_bwd_b_kernel");
}

extern "C" int tt_enc_last_bwd_weights(const float* x, int64_t B, int64_t H, int64_t D, int64_t heads, const float* d_recent,
                                       int64_t ld_dr, const float* q0, const float* xbar, const float* ctx0, float* dW_in,
                                       float* db_in, float* dW_out, float* db_out, void* ws, int64_t ws_bytes,
                                       tt_stream_t stream) {
  if (!x || !d_recent || !q0 || !xbar || !ctx0 || !dW_in || !db_in || !dW_out || !db_out)
    return fail_arg("tt_enc_last_bwd_weights: null pointer");
  if (B < 0 || !tt_enc_last_supported(H, D, heads) || ld_dr < D) {
    set_error("tt_enc_last_bwd: shape outside H <= 64, D <= 128, D %% 4 == 0, head width %% 4 == 0");
    return TT_E_UNSUPPORTED;
  }
  if (!el_aligned(x) || !el_aligned(xbar) || !el_aligned(ws) || !el_aligned(d_recent) || (ld_dr % 4) != 0)
    return fail_arg("tt_enc_last_bwd: x, xbar, d_recent, ws must be 16-byte aligned, ld_dr % 4 == 0");
  hipStream_t st = S(stream);
  if (B == 0) {
    (void)hipMemsetAsync(dW_in, 0, sizeof(float) * 3 * D * D, st);
    (void)hipMemsetAsync(db_in, 0, sizeof(float) * 3 * D, st);
    (void)hipMemsetAsync(dW_out, 0, sizeof(float) * D * D, st);
    (void)hipMemsetAsync(db_out, 0, sizeof(float) * D, st);
    return 0;
  }
  if (!ws || ws_bytes < tt_enc_last_bwd_workspace_bytes(B, H, D, heads)) return fail_arg("tt_enc_last_bwd: workspace too small");
  const int64_t G = el_groups(B);
  ElArgs a{};
  a.x = x;
  a.B = B; a.H = (int)H; a.D = (int)D; a.heads = (int)heads;
  a.q0 = const_cast<float*>(q0); a.xbar = const_cast<float*>(xbar); a.ctx0 = const_cast<float*>(ctx0);
  a.d_recent = d_recent; a.ld_dr = ld_dr;
  el_carve(a, ws, B, D, heads);
  int rc;
  enc_last_wgrad_kernel<<<dim3((unsigned)G, 2), 256, 0, st>>>(a);
  if ((rc = check_launch("enc_last_wgrad_kernel"))) return rc;
  const int total = (int)(4 * D * D + 3 * D);
  enc_last_reduce_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, st>>>(a.part_a, a.part_b, (int)G, (int)D, dW_in, db_in, dW_out, db_out);
  return check_launch("enc_last_reduce_kernel");
}

extern "C" int tt_enc_last_bwd(const float* x, int64_t B, int64_t H, int64_t D, int64_t heads, const float* w_in,
                               const float* w_out, const float* d_recent, int64_t ld_dr, const float* q0, const float* t,
                               const float* probs, const float* xbar, const float* ctx0, float* dx, float* dW_in, float* db_in,
                               float* dW_out, float* db_out, void* ws, int64_t ws_bytes, tt_stream_t stream) {
  if (!q0 || !xbar || !ctx0 || !dW_in || !db_in || !dW_out || !db_out) return fail_arg("tt_enc_last_bwd: null pointer");
  const int rc = tt_enc_last_bwd_data(x, B, H, D, heads, w_in, w_out, d_recent, ld_dr, t, probs, dx, ws, ws_bytes, stream);
  if (rc) return rc;
  return tt_enc_last_bwd_weights(x, B, H, D, heads, d_recent, ld_dr, q0, xbar, ctx0, dW_in, db_in, dW_out, db_out, ws, ws_bytes,
                                 stream);
}
