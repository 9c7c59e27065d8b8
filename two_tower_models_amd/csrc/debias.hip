// Combined debias loss head, fused:
//   ref:src/two_tower_with_debiasing.py:77-129 (position prior + user prior, two sum-MSE terms, the
//   division by the clamped user prior) on top of ref:src/two_tower_base_retrieval.py:322-345
//   (clamp, division by the batch maximum, weighted mean of the per-row cross entropy).
//
//   n_i = sum_t labels[i,t] * uvw[t]                     net user value
//   p_i = pos_table[position_i]                          position prior              (Embedding(100, 1))
//   e_i = <user_emb[i,:], W[:DI]> + p_i * W[DI] + b      user prior                  (Linear(DI + 1, 1))
//   aux = sum_i (e_i - n_i)^2 + sum_i sum_j (p_i - n_j)^2        upstream compares [B,1] with [B]: a
//         [B,B] broadcast.  Here in closed form: B*sum p^2 - 2*sum p*sum n + B*sum n^2 -- O(B), not O(B^2)
//   r_i = max(n_i / max(e_i, 1e-3), 1e-6);  M = max_i r_i;  w_i = r_i / M
//   loss = (1/B) sum_i row_ce_i * w_i + aux
// Gradients follow torch's: clamp(min) passes the gradient where input >= min, max() splits it evenly
// over ties.  Every reduction runs in a fixed order (double accumulators) => bit-reproducible.
//
// `mode` selects the head (round 4: the two single-term siblings take the same kernels):
//   TT_DEBIAS_COMBINED  (0)  the above
//   TT_DEBIAS_POSITION  (1)  ref:src/two_tower_with_position_debiased_weights.py:76-113:  prior = p_i,
//                            aux = sum_i (p_i - n_i)^2  (the RAW prior),  r_i = max(n_i / max(p_i, 1e-3), 1e-6)
//   TT_DEBIAS_USER      (2)  ref:src/two_tower_with_user_debiased_weights.py:100-135:  e_i = <user_emb[i,:], W[:DI]> + b,
//                            c_i = max(e_i, 1e-1) FIRST,  aux = sum_i (c_i - n_i)^2  (the CLAMPED prior: a clamped row
//                            sends no gradient to the head),  r_i = max(n_i / c_i, 1e-6)
#include "common.hpp"

namespace tt {

constexpr float USER_PRIOR_MIN = 1.0e-3f;  // ref:src/two_tower_with_debiasing.py:119-121; position-only: ref:...position_debiased_weights.py:103
constexpr float USER_ONLY_PRIOR_MIN = 1.0e-1f;  // ref:src/two_tower_with_user_debiased_weights.py:127
__device__ __forceinline__ float prior_min(int mode) { return mode == TT_DEBIAS_USER ? USER_ONLY_PRIOR_MIN : USER_PRIOR_MIN; }
constexpr float WEIGHT_MIN = 1.0e-6f;      // ref:src/two_tower_base_retrieval.py:335-337

struct DebiasScalars {  // lives in the workspace between forward and backward
  double max_r, ties, sum_ce_r, sum_n, sum_p;
};

// one wavefront per batch row: n_i, p_i, e_i
__global__ __launch_bounds__(256) void debias_rows_fwd_kernel(const float* __restrict__ labels, int64_t B, int64_t T,
                                                              const float* __restrict__ uvw,
                                                              const int64_t* __restrict__ position, int64_t n_pos,
                                                              const float* __restrict__ pos_table,
                                                              const float* __restrict__ ue, int64_t ld_ue, int64_t DI,
                                                              const float* __restrict__ lin_w,
                                                              const float* __restrict__ lin_b, float* __restrict__ n_out,
                                                              float* __restrict__ p_out, float* __restrict__ e_out,
                                                              int32_t* __restrict__ oob_flag, int mode) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B) return;
  const int lane = threadIdx.x & 63;
  float nv = 0.f;
  for (int64_t t = lane; t < T; t += 64) nv += labels[i * T + t] * uvw[t];
  nv = wave_sum(nv);
  float pv = 0.f;
  if (mode != TT_DEBIAS_USER) {
    int64_t pos = position[i];
    if (pos < 0 || pos >= n_pos) { if (lane == 0) *oob_flag = 1; pos = 0; }
    pv = pos_table[pos];
  }
  float dot = 0.f;
  if (mode != TT_DEBIAS_POSITION) {
    for (int64_t k = lane; k < DI; k += 64) dot += ue[i * ld_ue + k] * lin_w[k];
    dot = wave_sum(dot);
  }
  if (lane == 0) {
    n_out[i] = nv;
    p_out[i] = pv;
    // the value the weight is divided by (before its clamp): the user prior, or (position-only) the position prior
    e_out[i] = mode == TT_DEBIAS_POSITION ? pv : mode == TT_DEBIAS_USER ? dot + lin_b[0] : dot + pv * lin_w[DI] + lin_b[0];
  }
}

__device__ __forceinline__ double block_sum(double v, double* red /*[16]*/) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}

// single workgroup: r_i, the five sums, the maximum with its tie count, the loss
__global__ __launch_bounds__(1024) void debias_reduce_fwd_kernel(const float* __restrict__ row_ce, int64_t B,
                                                                 const float* __restrict__ nv,
                                                                 const float* __restrict__ pv,
                                                                 const float* __restrict__ ev, float* __restrict__ rv,
                                                                 DebiasScalars* __restrict__ sc,
                                                                 float* __restrict__ loss_out, int mode) {
  __shared__ double red[16];
  __shared__ float redm[16];
  double s_en2 = 0.0, s_p2 = 0.0, s_p = 0.0, s_n = 0.0, s_n2 = 0.0, s_cr = 0.0;
  float mx = -3.0e38f;
  for (int64_t i = threadIdx.x; i < B; i += blockDim.x) {
    const float n = nv[i], p = pv[i], e = ev[i];
    const float c = fmaxf(e, prior_min(mode));
    const float r = fmaxf(n / c, WEIGHT_MIN);
    rv[i] = r;
    const double d = (double)(mode == TT_DEBIAS_USER ? c : e) - (double)n;  // user-only: the CLAMPED prior enters the MSE
    s_en2 += d * d;
    s_p2 += (double)p * p;
    s_p += p;
    s_n += n;
    s_n2 += (double)n * n;
    s_cr += (double)row_ce[i] * r;
    mx = fmaxf(mx, r);
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = mx;
  __syncthreads();
  float M = redm[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) M = fmaxf(M, redm[w]);
  double ties = 0.0;
  for (int64_t i = threadIdx.x; i < B; i += blockDim.x) ties += (rv[i] == M) ? 1.0 : 0.0;
  s_en2 = block_sum(s_en2, red);
  s_p2 = block_sum(s_p2, red);
  s_p = block_sum(s_p, red);
  s_n = block_sum(s_n, red);
  s_n2 = block_sum(s_n2, red);
  s_cr = block_sum(s_cr, red);
  ties = block_sum(ties, red);
  if (threadIdx.x == 0) {
    const double b = (double)B;
    const double aux = s_en2 + (mode == TT_DEBIAS_COMBINED ? (b * s_p2 - 2.0 * s_p * s_n + b * s_n2) : 0.0);
    sc->max_r = M;
    sc->ties = ties;
    sc->sum_ce_r = s_cr;
    sc->sum_n = s_n;
    sc->sum_p = s_p;
    *loss_out = (float)(s_cr / (double)M / b + aux);
  }
}

// one wavefront per row: d row_ce, d user_emb, and the per-row gradients of e and p
__global__ __launch_bounds__(256) void debias_rows_bwd_kernel(const float* __restrict__ grad_loss,
                                                              const float* __restrict__ row_ce, int64_t B,
                                                              const float* __restrict__ nv,
                                                              const float* __restrict__ pv,
                                                              const float* __restrict__ ev,
                                                              const float* __restrict__ rv,
                                                              const DebiasScalars* __restrict__ sc, int64_t DI,
                                                              const float* __restrict__ lin_w,
                                                              float* __restrict__ d_row_ce, float* __restrict__ d_ue,
                                                              int64_t ld_due, float* __restrict__ ge_out,
                                                              float* __restrict__ gp_out, int mode) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B) return;
  const int lane = threadIdx.x & 63;
  const float g = grad_loss[0];
  const float M = (float)sc->max_r;
  const float b = (float)B;
  const float n = nv[i], p = pv[i], e = ev[i], r = rv[i], ce = row_ce[i];
  const float cmin = prior_min(mode);
  const float c = fmaxf(e, cmin);
  const float q = n / c;
  float gr = g * (ce / (b * M));
  if (r == M) gr -= g * (float)(sc->sum_ce_r / ((double)b * (double)M * (double)M * sc->ties));
  const float gq = (q >= WEIGHT_MIN) ? gr : 0.f;
  const float gc = -gq * ((n / c) / c);
  float ge, gp;
  if (mode == TT_DEBIAS_COMBINED) {
    ge = ((e >= cmin) ? gc : 0.f) + g * 2.f * (e - n);
    gp = ge * lin_w[DI] + g * 2.f * (float)((double)b * (double)p - sc->sum_n);
  } else if (mode == TT_DEBIAS_POSITION) {  // e IS the position prior: the MSE sees it raw, the division clamped
    ge = 0.f;
    gp = ((e >= cmin) ? gc : 0.f) + g * 2.f * (e - n);
  } else {  // user-only: clamp first, so both terms pass through it
    ge = (e >= cmin) ? gc + g * 2.f * (c - n) : 0.f;
    gp = 0.f;
  }
  for (int64_t k = lane; k < DI; k += 64) d_ue[i * ld_due + k] = mode == TT_DEBIAS_POSITION ? 0.f : ge * lin_w[k];
  if (lane == 0) {
    d_row_ce[i] = g * (r / M) / b;
    ge_out[i] = ge;
    gp_out[i] = gp;
  }
}

// d lin_w, d lin_b, d pos_table in two steps, fixed summation order throughout: every workgroup
// reduces ROWS_PER_WG consecutive rows into one partial row [DI + 2 + n_pos]; a single workgroup then
// adds the partial rows in order.  (One workgroup walking all B rows took 0.55 ms at B = 4096.)
constexpr int ROWS_PER_WG = 64;

__global__ __launch_bounds__(256) void debias_partial_bwd_kernel(int64_t B, const float* __restrict__ ge,
                                                                 const float* __restrict__ gp,
                                                                 const float* __restrict__ pv,
                                                                 const int64_t* __restrict__ position, int64_t n_pos,
                                                                 const float* __restrict__ ue, int64_t ld_ue, int64_t DI,
                                                                 float* __restrict__ partial) {
  __shared__ float part[2][128];
  __shared__ float s_ge[ROWS_PER_WG], s_gp[ROWS_PER_WG], s_p[ROWS_PER_WG];
  __shared__ int s_pos[ROWS_PER_WG];
  const int64_t r0 = (int64_t)blockIdx.x * ROWS_PER_WG;
  const int rows = (int)((B - r0 < ROWS_PER_WG) ? B - r0 : ROWS_PER_WG);
  float* out = partial + (int64_t)blockIdx.x * (DI + 2 + n_pos);
  if (threadIdx.x < ROWS_PER_WG) {
    const bool ok = (int)threadIdx.x < rows;
    const int64_t i = r0 + (ok ? threadIdx.x : 0);
    int64_t pos = position ? position[i] : 0;
    if (pos < 0 || pos >= n_pos) pos = 0;  // as in the forward (which raised the flag)
    s_ge[threadIdx.x] = ok ? ge[i] : 0.f;
    s_gp[threadIdx.x] = ok ? gp[i] : 0.f;
    s_p[threadIdx.x] = ok ? pv[i] : 0.f;
    s_pos[threadIdx.x] = ok ? (int)pos : -1;
  }
  __syncthreads();
  const int col = threadIdx.x & 127, grp = threadIdx.x >> 7;
  for (int64_t k0 = 0; k0 < DI; k0 += 128) {
    const int64_t k = k0 + col;
    float acc = 0.f;
    if (k < DI)
      for (int i = grp; i < rows; i += 2) acc += s_ge[i] * ue[(r0 + i) * ld_ue + k];
    part[grp][col] = acc;
    __syncthreads();
    if (grp == 0 && k < DI) out[k] = part[0][col] + part[1][col];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float a = 0.f, c = 0.f;
    for (int i = 0; i < rows; ++i) { a += s_ge[i] * s_p[i]; c += s_ge[i]; }
    out[DI] = a;
    out[DI + 1] = c;
  }
  for (int64_t j = threadIdx.x; j < n_pos; j += blockDim.x) {
    float a = 0.f;
    for (int i = 0; i < rows; ++i) a += (s_pos[i] == (int)j) ? s_gp[i] : 0.f;
    out[DI + 2 + j] = a;
  }
}

__global__ __launch_bounds__(256) void debias_final_bwd_kernel(const float* __restrict__ partial, int64_t n_wg, int64_t DI,
                                                               int64_t n_pos, float* __restrict__ d_pos_table,
                                                               float* __restrict__ d_lin_w, float* __restrict__ d_lin_b,
                                                               int mode) {
  const int64_t width = DI + 2 + n_pos;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < width; c += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int64_t g = 0; g < n_wg; ++g) s += partial[g * width + c];
    if (c < DI) { if (mode != TT_DEBIAS_POSITION) d_lin_w[c] = (float)s; }
    else if (c == DI) { if (mode == TT_DEBIAS_COMBINED) d_lin_w[c] = (float)s; }  // the weight of the position-prior input
    else if (c == DI + 1) { if (mode != TT_DEBIAS_POSITION) d_lin_b[0] = (float)s; }
    else if (mode != TT_DEBIAS_USER) d_pos_table[c - DI - 2] = (float)s;
  }
}

struct DebiasWs {
  float *n, *p, *e, *r, *ge, *gp, *partial;
  DebiasScalars* sc;
};
static DebiasWs carve_debias(void* ws, int64_t B, int64_t DI, int64_t n_pos) {
  Carver cv(ws);
  DebiasWs w;
  w.n = cv.take<float>(B); w.p = cv.take<float>(B); w.e = cv.take<float>(B); w.r = cv.take<float>(B);
  w.ge = cv.take<float>(B); w.gp = cv.take<float>(B);
  w.sc = cv.take<DebiasScalars>(1);
  w.partial = cv.take<float>(ceil_div(B, ROWS_PER_WG) * (DI + 2 + n_pos));
  return w;
}

}  // namespace tt

using namespace tt;

extern "C" int64_t tt_debias_loss_workspace_bytes(int64_t B, int64_t DI, int64_t n_pos) {
  if (B <= 0 || DI <= 0 || n_pos <= 0) return 0;
  return 6 * round_up(B * 4, 256) + round_up((int64_t)sizeof(DebiasScalars), 256) +
         round_up(ceil_div(B, ROWS_PER_WG) * (DI + 2 + n_pos) * 4, 256);
}

static bool debias_ptrs_ok(int mode, const void* position, const void* pos_table, const void* lin_w, const void* lin_b) {
  if (mode != TT_DEBIAS_USER && (!position || !pos_table)) return false;
  if (mode != TT_DEBIAS_POSITION && (!lin_w || !lin_b)) return false;
  return mode == TT_DEBIAS_COMBINED || mode == TT_DEBIAS_POSITION || mode == TT_DEBIAS_USER;
}

extern "C" int tt_debias_loss_fwd(int mode, const float* row_ce, const float* labels, int64_t B, int64_t T, const float* uvw,
                                  const int64_t* position, int64_t n_pos, const float* pos_table, const float* user_emb,
                                  int64_t ld_ue, int64_t DI, const float* lin_w, const float* lin_b, float* loss_out,
                                  void* ws, int64_t ws_bytes, int32_t* oob_flag, tt_stream_t stream) {
  if (!row_ce || !labels || !uvw || !user_emb || !loss_out || !ws || !oob_flag || !debias_ptrs_ok(mode, position, pos_table, lin_w, lin_b))
    return fail_arg("tt_debias_loss_fwd: null pointer (or unknown mode)");
  if (B <= 0 || T <= 0 || n_pos <= 0 || DI <= 0 || ld_ue < DI) return fail_arg("tt_debias_loss_fwd: sizes");
  if (ws_bytes < tt_debias_loss_workspace_bytes(B, DI, n_pos)) { set_error("tt_debias_loss_fwd: workspace"); return TT_E_WORKSPACE; }
  const DebiasWs w = carve_debias(ws, B, DI, n_pos);
  hipStream_t st = S(stream);
  debias_rows_fwd_kernel<<<(unsigned)ceil_div(B, 4), 256, 0, st>>>(labels, B, T, uvw, position, n_pos, pos_table, user_emb,
                                                                    ld_ue, DI, lin_w, lin_b, w.n, w.p, w.e, oob_flag, mode);
  int rc = check_launch("debias_rows_fwd_kernel");
  if (rc) return rc;
  debias_reduce_fwd_kernel<<<1, 1024, 0, st>>>(row_ce, B, w.n, w.p, w.e, w.r, w.sc, loss_out, mode);
  return check_launch("debias_reduce_fwd_kernel");
}

extern "C" int tt_debias_loss_bwd(int mode, const float* grad_loss, const float* row_ce, int64_t B, const int64_t* position,
                                  int64_t n_pos, const float* user_emb, int64_t ld_ue, int64_t DI, const float* lin_w,
                                  const void* ws, int64_t ws_bytes, float* d_row_ce, float* d_user_emb, int64_t ld_due,
                                  float* d_pos_table, float* d_lin_w, float* d_lin_b, tt_stream_t stream) {
  if (!grad_loss || !row_ce || !user_emb || !ws || !d_row_ce || !d_user_emb ||
      !debias_ptrs_ok(mode, position, d_pos_table, lin_w, d_lin_b) || (mode != TT_DEBIAS_POSITION && !d_lin_w))
    return fail_arg("tt_debias_loss_bwd: null pointer (or unknown mode)");
  if (B <= 0 || n_pos <= 0 || DI <= 0 || ld_ue < DI || ld_due < DI) return fail_arg("tt_debias_loss_bwd: sizes");
  if (ws_bytes < tt_debias_loss_workspace_bytes(B, DI, n_pos)) { set_error("tt_debias_loss_bwd: workspace"); return TT_E_WORKSPACE; }
  const DebiasWs w = carve_debias(const_cast<void*>(ws), B, DI, n_pos);
  hipStream_t st = S(stream);
  debias_rows_bwd_kernel<<<(unsigned)ceil_div(B, 4), 256, 0, st>>>(grad_loss, row_ce, B, w.n, w.p, w.e, w.r, w.sc, DI, lin_w,
                                                                    d_row_ce, d_user_emb, ld_due, w.ge, w.gp, mode);
  int rc = check_launch("debias_rows_bwd_kernel");
  if (rc) return rc;
  const int64_t n_wg = ceil_div(B, ROWS_PER_WG);
  debias_partial_bwd_kernel<<<(unsigned)n_wg, 256, 0, st>>>(B, w.ge, w.gp, w.p, position, n_pos, user_emb, ld_ue, DI, w.partial);
  if ((rc = check_launch("debias_partial_bwd_kernel"))) return rc;
  debias_final_bwd_kernel<<<(unsigned)ceil_div(DI + 2 + n_pos, 256), 256, 0, st>>>(w.partial, n_wg, DI, n_pos, d_pos_table, d_lin_w,
                                                                                    d_lin_b, mode);
  return check_launch("debias_final_bwd_kernel");
}
