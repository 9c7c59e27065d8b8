// In-batch softmax CE for embedding widths the stationary-operand kernels of inbatch_ce.hip do not
// hold in registers (D > 128).  The reference accepts any width (torch.matmul + F.cross_entropy,
// ref:src/two_tower_base_retrieval.py:287,310-312); this is the generic -- slower -- form of the same
// arithmetic, built from the library's own fp32-MFMA GEMM:
//
//   per chunk of Mc user rows (Mc x N logits stay within ~512 MB of workspace):
//     S = U_chunk . I^T                       tt_gemm_f32 (NT)
//     forward      lse_i, ce_i = lse_i - S[i, i + off]      one workgroup per row
//     forward+dU   S <- softmax rows;  du_unit = S . I  (NN)  - I[i + off]
//     backward     S <- (softmax - onehot) * coef_i;  dU = S . I (NN);  dI += S^T . U_chunk (TN)
// The logits ARE written here (that is what makes it the slow path); nothing else differs.
#include "common.hpp"

namespace tt {

constexpr float WIDE_NEG = -3.0e38f;

// MODE 0: statistics only.  MODE 1: statistics, then S <- exp(S - lse).
template <int MODE>
__global__ __launch_bounds__(256) void wide_rows_fwd_kernel(float* __restrict__ S, int64_t N, int64_t lds, int64_t row0,
                                                            int64_t diag_offset, float* __restrict__ row_lse,
                                                            float* __restrict__ row_ce) {
  __shared__ float red[4];
  const int64_t r = blockIdx.x;  // row inside the chunk
  float* s = S + r * lds;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mx = WIDE_NEG;
  for (int64_t j = threadIdx.x; j < N; j += 256) mx = fmaxf(mx, s[j]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int64_t j = threadIdx.x; j < N; j += 256) sum += expf(s[j] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = (red[0] + red[1]) + (red[2] + red[3]);
  const float lse = mx + logf(sum);
  const int64_t g = row0 + r;
  if (threadIdx.x == 0) {
    row_lse[g] = lse;
    row_ce[g] = lse - s[g + diag_offset];
  }
  if constexpr (MODE == 1) {
    __syncthreads();  // the diagonal logit has been read
    for (int64_t j = threadIdx.x; j < N; j += 256) s[j] = expf(s[j] - lse);
  }
}

__global__ __launch_bounds__(256) void wide_rows_bwd_kernel(float* __restrict__ S, int64_t N, int64_t lds, int64_t row0,
                                                            int64_t diag_offset, const float* __restrict__ row_lse,
                                                            const float* __restrict__ coef) {
  const int64_t r = blockIdx.x, g = row0 + r;
  float* s = S + r * lds;
  const float lse = row_lse[g], c = coef[g];
  const int64_t d = g + diag_offset;
  for (int64_t j = threadIdx.x; j < N; j += 256) s[j] = (expf(s[j] - lse) - (j == d ? 1.f : 0.f)) * c;
}

__global__ __launch_bounds__(256) void wide_sub_diag_kernel(float* __restrict__ du, int64_t ld_du, const float* __restrict__ I,
                                                            int64_t ldi, int64_t rows, int64_t D, int64_t row0,
                                                            int64_t diag_offset) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * D) return;
  const int64_t r = i / D, d = i - r * D;
  du[(row0 + r) * ld_du + d] -= I[(row0 + r + diag_offset) * ldi + d];
}

static int64_t wide_chunk_rows(int64_t M, int64_t N) {
  int64_t mc = (512ll << 20) / (N * 4);
  mc = mc / 128 * 128;
  if (mc < 128) mc = 128;
  return mc < M ? mc : M;
}

int64_t ce_wide_workspace_bytes(int64_t M, int64_t N, int64_t D) {
  const int64_t mc = wide_chunk_rows(M, N);
  // the GEMM scratch for BOTH chunk sizes that occur: a short ragged last chunk (M % mc rows) has fewer tiles,
  // so the split-K planner may give it MORE splits -- and more scratch -- than the full chunk
  int64_t g = 0;
  const int64_t sizes[2] = {mc, M % mc};
  for (int64_t rows : sizes) {
    if (rows <= 0) continue;
    const int64_t cand[3] = {tt_gemm_workspace_bytes(TT_GEMM_NT, rows, N, D), tt_gemm_workspace_bytes(TT_GEMM_NN, rows, D, N),
                             tt_gemm_workspace_bytes(TT_GEMM_TN, N, D, rows)};
    for (int64_t c : cand) if (c > g) g = c;
  }
  return round_up(mc * N * 4, 256) + round_up(g, 256);
}

// mode 0: forward (lse, ce); 1: forward + du_unit; 2: backward (dU optional, dI)
int ce_wide_run(int mode, const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M, int64_t N, int64_t D,
                int64_t diag_offset, float* row_lse, float* row_ce, const float* coef, float* dU, int64_t lddu, float* dI,
                int64_t lddi, void* ws, int64_t ws_bytes, hipStream_t st) {
  if (ws_bytes < ce_wide_workspace_bytes(M, N, D)) { set_error("tt_inbatch_ce (D > 128): workspace"); return TT_E_WORKSPACE; }
  const int64_t mc = wide_chunk_rows(M, N);
  float* S = reinterpret_cast<float*>(ws);
  char* gws = reinterpret_cast<char*>(ws) + round_up(mc * N * 4, 256);
  const int64_t gws_bytes = ws_bytes - round_up(mc * N * 4, 256);
  tt_stream_t ts = reinterpret_cast<tt_stream_t>(st);
  int rc;
  for (int64_t r0 = 0; r0 < M; r0 += mc) {
    const int64_t rows = (M - r0 < mc) ? M - r0 : mc;
    const float* Uc = U + r0 * ldu;
    if ((rc = tt_gemm_f32(TT_GEMM_NT, rows, N, D, Uc, ldu, I, ldi, S, N, nullptr, TT_EPI_NONE, nullptr, 0, 0, gws, gws_bytes, ts)))
      return rc;
    if (mode == 0) {
      wide_rows_fwd_kernel<0><<<(unsigned)rows, 256, 0, st>>>(S, N, N, r0, diag_offset, row_lse, row_ce);
      if ((rc = check_launch("wide_rows_fwd_kernel"))) return rc;
    } else if (mode == 1) {
      wide_rows_fwd_kernel<1><<<(unsigned)rows, 256, 0, st>>>(S, N, N, r0, diag_offset, row_lse, row_ce);
      if ((rc = check_launch("wide_rows_fwd_kernel"))) return rc;
      if ((rc = tt_gemm_f32(TT_GEMM_NN, rows, D, N, S, N, I, ldi, dU + r0 * lddu, lddu, nullptr, TT_EPI_NONE, nullptr, 0, 0, gws,
                            gws_bytes, ts)))
        return rc;
      wide_sub_diag_kernel<<<(unsigned)ceil_div(rows * D, 256), 256, 0, st>>>(dU, lddu, I, ldi, rows, D, r0, diag_offset);
      if ((rc = check_launch("wide_sub_diag_kernel"))) return rc;
    } else {
      wide_rows_bwd_kernel<<<(unsigned)rows, 256, 0, st>>>(S, N, N, r0, diag_offset, row_lse, coef);
      if ((rc = check_launch("wide_rows_bwd_kernel"))) return rc;
      if (dU && (rc = tt_gemm_f32(TT_GEMM_NN, rows, D, N, S, N, I, ldi, dU + r0 * lddu, lddu, nullptr, TT_EPI_NONE, nullptr, 0, 0,
                                  gws, gws_bytes, ts)))
        return rc;
      if ((rc = tt_gemm_f32(TT_GEMM_TN, N, D, rows, S, N, Uc, ldu, dI, lddi, nullptr, TT_EPI_NONE, nullptr, 0, r0 > 0 ? 1 : 0, gws,
                            gws_bytes, ts)))
        return rc;
    }
  }
  return 0;
}

}  // namespace tt
