/*
 * tt_hotpath.h -- C ABI of libtt_hotpath.so: the MI355X (gfx950) hot path of a
 * two-tower retrieval trainer, as hand-written HIP kernels.
 *
 * The reference (gauravchak/two_tower_models) has no FFI of its own: its
 * boundary is the Python nn.Module surface.  Every entry point
 * below replaces one torch call site of that surface; the site is cited as
 * ref:<file>:<line> (paths relative to the reference root).  The Python
 * modules in two_tower_models_amd/ bind these symbols with ctypes and keep the
 * reference's class / method / keyword names (INTEGRATION.md shows the stub a
 * maintainer of the reference would add).
 *
 * Conventions
 *   - plain C, no torch / C++ types.  Pointers are DEVICE pointers unless a
 *     parameter is documented "host".
 *   - every function is asynchronous on `stream` (a hipStream_t passed as
 *     void*), allocates nothing, frees nothing and keeps no pointer after it
 *     returns.  Scratch comes in through (`ws`, `ws_bytes`); the matching
 *     tt_*_workspace_bytes() says how much is needed.
 *   - return value: 0 = ok; >0 = hipError_t of a failed launch; <0 = TT_E_*.
 *     tt_last_error_string() describes the last failure on the calling thread.
 *   - all tensors are row-major fp32 unless stated; `ld*` are row strides in
 *     ELEMENTS; ids / indices are int64 (torch.long) at the boundary.
 *   - out-of-range ids never fault: the kernel writes zeros for that row and
 *     sets *oob_flag (device int32) to 1; the Python layer raises IndexError
 *     (what torch's nn.Embedding raises) when it next reads the flag.
 */
#ifndef TT_HOTPATH_H
#define TT_HOTPATH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TT_ABI_VERSION 4

#define TT_E_BADARG (-1)      /* null pointer / negative size / unsupported shape */
#define TT_E_WORKSPACE (-2)   /* ws_bytes smaller than tt_*_workspace_bytes()     */
#define TT_E_UNSUPPORTED (-3) /* shape outside what the kernels implement         */

typedef void* tt_stream_t; /* hipStream_t */

int tt_abi_version(void);
const char* tt_last_error_string(void);

/* Per-kernel timing for the benchmark: when enabled, the launches of the named hot
 * kernels ("adam_sweep_kernel", "ce_fwd_kernel", "ce_bwd_kernel", "mips_score_kernel")
 * are bracketed by HIP events on their own stream.  tt_profile_read synchronises those
 * events and returns the summed duration and the launch count. */
int tt_profile_enable(int on);
int tt_profile_read(const char* kernel, double* total_ms, int64_t* launches);
/* Bracket only the named kernels ("a,b,c"; NULL or "" = all of them again).  An event pair ends the overlap between the
 * kernel and its neighbours on the stream and delays what follows it by a few microseconds -- enough to change which of
 * two streams' kernels reaches the CUs first -- so a measurement brackets the kernel it reports and nothing else. */
int tt_profile_filter(const char* kernels);
/* Suspend / resume the bracketing WITHOUT clearing what was recorded (the optimiser's one-off timing of candidate table
 * arenas at its first step is not part of the caller's measurement). */
int tt_profile_pause(int paused);

/* ---------------------------------------------------------------- K1 gather
 * out[i, 0:dim] = table[ids[i], 0:dim]          (out row stride ld_out >= dim)
 * replaces nn.Embedding.__call__ at ref:src/two_tower_base_retrieval.py:126,209
 * and the history lookup ref:src/two_tower_with_user_history_encoder.py:105.
 * ld_out lets the caller write straight into a column slice of the tower
 * input, which is how torch.cat (ref:...base_retrieval.py:159,214) disappears.
 * oob_flag may be NULL: out-of-range ids then zero-fill silently (the sharded trainer uses
 * this to gather "my rows, zeros for everyone else's" ahead of a reduce-scatter). */
int tt_gather_rows(const float* table, int64_t n_rows, int64_t dim, const int64_t* ids,
                   int64_t n_ids, float* out, int64_t ld_out, int32_t* oob_flag,
                   tt_stream_t stream);

/* ---------------------------------------------------------------- K3 GEMM
 * C[M,N] (+)= A(M,K) * B(K,N) (+ bias[N]) with an optional epilogue, fp32 MFMA
 * (v_mfma_f32_32x32x2_f32).  `layout`:
 *   TT_GEMM_NT  A is [M,K] row-major, B is [N,K] row-major   y = x W^T
 *               (nn.Linear forward, ref:...base_retrieval.py:76-80,90-93,101-110)
 *   TT_GEMM_NN  A is [M,K], B is [K,N]                        dx = dy W
 *   TT_GEMM_TN  A is [K,M], B is [K,N]                        dW = dy^T x
 * `epilogue`: 0 none; TT_EPI_RELU max(.,0); TT_EPI_RELU_MASK multiply by
 * (aux[m,n] > 0) -- the ReLU backward mask (aux = saved forward activation).
 * `accumulate` != 0 adds into C instead of overwriting it. */
#define TT_GEMM_NT 0
#define TT_GEMM_NN 1
#define TT_GEMM_TN 2
#define TT_EPI_NONE 0
#define TT_EPI_RELU 1
#define TT_EPI_RELU_MASK 2
int64_t tt_gemm_workspace_bytes(int layout, int64_t M, int64_t N, int64_t K);
int tt_gemm_f32(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                int epilogue, const float* aux, int64_t ldaux, int accumulate, void* ws,
                int64_t ws_bytes, tt_stream_t stream);
/* The weight-gradient GEMM with the bias gradient for free: C = A^T B (TT_GEMM_TN, A = dy [K,M],
 * B = x [K,N]) and a_colsum[m] = sum_k A[k,m] (= db) from the same pass over A.  Same workspace
 * as tt_gemm_f32 (tt_gemm_workspace_bytes(TT_GEMM_TN, M, N, K)); deterministic. */
int tt_gemm_tn_colsum_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                          int64_t ldb, float* C, int64_t ldc, int accumulate, float* a_colsum, void* ws,
                          int64_t ws_bytes, tt_stream_t stream);

/* out[n] = sum_m X[m,n]  (bias gradients), deterministic two-stage reduction */
int64_t tt_colsum_workspace_bytes(int64_t M, int64_t N);
int tt_colsum_f32(const float* X, int64_t M, int64_t N, int64_t ldx, float* out, void* ws,
                  int64_t ws_bytes, tt_stream_t stream);

/* ---------------------------------------------------------------- K3 fused tower
 * One tower of TwoTowerBaseRetrieval in ONE launch per direction:
 *   y = Linear(2D -> D)([ table[id] | Linear(hidden -> D)(ReLU(Linear(F -> hidden)(features))) ])
 * replaces nn.Embedding + nn.Sequential + torch.cat + nn.Linear at ref:src/two_tower_base_retrieval.py:129-162,164-191
 * (user tower) and :193-219 (item tower).  The forward also writes what the backward needs: h_out [B, hidden] (the ReLU
 * output) and tin_out [B, 2D] (the tower input).  tt_tower_bwd_data is the data side of their autograd:
 *   d_tin = dy W3;  d_emb = d_tin[:, :D] (embedding-row gradients);  d_f = d_tin[:, D:];  dh = (d_f W2) (.) [h > 0]
 * and tt_tower_bwd_weights the parameter side -- dW3 = dy^T tin, dW2 = d_f^T h, dW1 = dh^T features and the three
 * bias sums (autograd of the nn.Linear layers of the same lines) -- in one product launch over 64-row blocks plus one
 * deterministic reduce over the blocks' partials (`ws`: tt_tower_bwd_weights_workspace_bytes) instead of three
 * tt_gemm_tn_colsum_f32 calls.
 * Shapes: hidden = 256, D = d_out in {32, 64, 128}, F <= 64, 16-B aligned rows (tt_tower_supported says);
 * anything else returns TT_E_UNSUPPORTED -- use tt_gather_rows + tt_gemm_f32. */
int tt_tower_supported(int64_t D, int64_t F, int64_t hidden, int64_t d_out);
int tt_tower_fwd(const float* table, int64_t n_rows, const int64_t* ids, const float* feats, int64_t ldf, int64_t B,
                 int64_t D, int64_t F, int64_t hidden, const float* W1, const float* b1, const float* W2, const float* b2,
                 const float* W3, const float* b3, int64_t d_out, float* y, int64_t ldy, float* h_out, float* tin_out,
                 int32_t* oob_flag, tt_stream_t stream);
int tt_tower_bwd_data(const float* dy, int64_t ldy, int64_t B, int64_t D, int64_t hidden, const float* W2, const float* W3,
                      const float* h, float* d_emb, int64_t ld_demb, float* d_f, float* dh, tt_stream_t stream);
int64_t tt_tower_bwd_weights_workspace_bytes(int64_t B, int64_t D, int64_t F, int64_t hidden);
int tt_tower_bwd_weights(const float* dy, int64_t ldy, const float* tin, const float* d_f, const float* h, const float* dh,
                         const float* feats, int64_t ldf, int64_t B, int64_t D, int64_t F, int64_t hidden, float* dW1,
                         float* db1, float* dW2, float* db2, float* dW3, float* db3, void* ws, int64_t ws_bytes,
                         tt_stream_t stream);
/* The same three entry points for a tower whose input has a THIRD, dense block:
 *   y = Linear(2D + E -> D)([ table[id] | feature MLP | extra[B, E] ]),  E = 2D  (E = 0: the functions above)
 * = TwoTowerWithUserHistoryEncoder's user tower, whose input gains the history encoder's [recent | mean] summary
 * (ref:src/two_tower_with_user_history_encoder.py:81-83 the Linear(2*DU + 2*DI -> DI), :85-122 the cat).  W3 / dW3 are
 * [D, 2D + E] row-major, tin_out stays [B, 2D] (the extra block is the caller's own tensor), d_extra [B, E] is the
 * gradient that flows back into the encoder. */
/* BOTH towers of TwoTowerBaseRetrieval per launch (user tower = sides[0], item tower = sides[1]; blockIdx.y picks the tower,
 * the arithmetic is tt_tower_fwd / _bwd_data / _bwd_weights' own, so the results are the same bits): for steps that run on
 * ONE stream -- a whole-step hipGraph, batches too small for the two-stream fork -- where two launches of a 128-workgroup
 * kernel ran back to back on a 256-CU chip.  Same shape limits, D the same for both towers, no third input block.
 * ref:src/two_tower_base_retrieval.py:129-219 (both towers' forward), their autograd. */
typedef struct {
  const float* table; int64_t n_rows; const int64_t* ids; const float* feats; int64_t ldf; int64_t F;
  const float *W1, *b1, *W2, *b2, *W3, *b3;
  float* y; int64_t ldy; float* h_out; float* tin_out;
} tt_tower_fwd_side;
typedef struct {
  const float* dy; int64_t ldy; const float *W2, *W3, *h;
  float* d_emb; int64_t ld_demb; float* d_f; float* dh;
} tt_tower_bwd_side;
typedef struct {
  const float* dy; int64_t ldy; const float *tin, *d_f, *h, *dh, *feats; int64_t ldf; int64_t F;
  float *dW1, *db1, *dW2, *db2, *dW3, *db3;
  void* ws; int64_t ws_bytes; /* tt_tower_bwd_weights_workspace_bytes(B, D, F, hidden), one per side */
} tt_tower_wgrad_side;
int tt_tower_fwd_pair(const tt_tower_fwd_side* sides /*host, 2*/, int64_t B, int64_t D, int64_t hidden, int32_t* oob_flag,
                      tt_stream_t stream);
int tt_tower_bwd_data_pair(const tt_tower_bwd_side* sides /*host, 2*/, int64_t B, int64_t D, int64_t hidden, tt_stream_t stream);
int tt_tower_bwd_weights_pair(const tt_tower_wgrad_side* sides /*host, 2*/, int64_t B, int64_t D, int64_t hidden,
                              tt_stream_t stream);
int tt_tower_x_supported(int64_t D, int64_t F, int64_t hidden, int64_t d_out, int64_t E);
int tt_tower_fwd_x(const float* table, int64_t n_rows, const int64_t* ids, const float* feats, int64_t ldf, int64_t B,
                   int64_t D, int64_t F, int64_t hidden, const float* W1, const float* b1, const float* W2, const float* b2,
                   const float* W3, const float* b3, int64_t d_out, const float* extra, int64_t ldx, int64_t E, float* y,
                   int64_t ldy, float* h_out, float* tin_out, int32_t* oob_flag, tt_stream_t stream);
int tt_tower_bwd_data_x(const float* dy, int64_t ldy, int64_t B, int64_t D, int64_t hidden, const float* W2, const float* W3,
                        const float* h, float* d_emb, int64_t ld_demb, float* d_f, float* dh, float* d_extra, int64_t ld_dx,
                        int64_t E, tt_stream_t stream);
int64_t tt_tower_bwd_weights_x_workspace_bytes(int64_t B, int64_t D, int64_t F, int64_t hidden, int64_t E);
int tt_tower_bwd_weights_x(const float* dy, int64_t ldy, const float* tin, const float* d_f, const float* h, const float* dh,
                           const float* feats, int64_t ldf, const float* extra, int64_t ldx, int64_t E, int64_t B, int64_t D,
                           int64_t F, int64_t hidden, float* dW1, float* db1, float* dW2, float* db2, float* dW3, float* db3,
                           void* ws, int64_t ws_bytes, tt_stream_t stream);

/* ---------------------------------------------------------------- K5 in-batch softmax CE
 * Forward: S = U I^T is never written to memory.
 *   row_lse[i] = log2 sum_j 2^(S[i,j] log2 e)   (= logsumexp_j S[i,j] / ln 2; opaque to the caller,
 *                                                 saved for tt_inbatch_ce_bwd which works in base 2)
 *   row_ce[i]  = logsumexp_j S[i,j] - S[i, i+diag_offset]
 * replaces torch.matmul + F.cross_entropy(reduction="none") at
 * ref:src/two_tower_base_retrieval.py:287,301-312.  U is [M,D], I is [N,D];
 * M == N and diag_offset == 0 for the reference; the sharded trainer passes
 * the all-gathered item block (N = world*M, diag_offset = rank*M).
 * Backward (recomputes S tile by tile): with G[i,j] = coef[i]*(softmax_j(S[i,:]) - [j == i+diag_offset]),
 *   dU = G I   [M,D],   dI = G^T U   [N,D]
 * where coef[i] = dLoss/d row_ce[i] (the normalised net_user_value weight / B,
 * ref:...base_retrieval.py:334-343).  Requires D <= 128 (TT_E_UNSUPPORTED otherwise). */
int64_t tt_inbatch_ce_workspace_bytes(int64_t M, int64_t N, int64_t D);
int tt_inbatch_ce_fwd(const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M,
                      int64_t N, int64_t D, int64_t diag_offset, float* row_lse, float* row_ce,
                      void* ws, int64_t ws_bytes, tt_stream_t stream);
int tt_inbatch_ce_bwd(const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M,
                      int64_t N, int64_t D, int64_t diag_offset, const float* row_lse,
                      const float* coef, float* dU /* may be NULL */, int64_t lddu, float* dI, int64_t lddi,
                      void* ws, int64_t ws_bytes, tt_stream_t stream);

/* Forward fused with the user-side gradient: additionally returns
 *   du_unit[i,:] = sum_j softmax(S)[i,j] I[j,:] - I[i + diag_offset,:]
 * so that dU[i,:] = dLoss/drow_ce[i] * du_unit[i,:] is an elementwise step and tt_inbatch_ce_bwd can
 * be called with dU == NULL (item side only): 4 instead of 5 logit-sized products per training
 * step.  Same workspace query as the other two. */
int tt_inbatch_ce_fwd_du(const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M, int64_t N,
                         int64_t D, int64_t diag_offset, float* row_lse, float* row_ce, float* du_unit,
                         int64_t ld_du, void* ws, int64_t ws_bytes, tt_stream_t stream);

/* The same pair with the logits KEPT between forward and backward (wide negative sets: several
 * ranks' items per user).  The forward also writes the masked log2-domain logits to `logits`
 * (tt_inbatch_ce_logits_bytes(M, N) bytes: [round_up(M,128)][round_up(N,128)] fp32); the item-side
 * backward rebuilds the gradient tile from them instead of from a second U I^T product -- 3 instead
 * of 4 logit-sized products per training step, for one write and one read of the buffer.  Needs
 * D in {32, 64, 128}, 16-B aligned rows and N < 4 Mi (TT_E_UNSUPPORTED otherwise: use the pair above).
 * Same workspace query as the others; dU comes from du_unit as above. */
int64_t tt_inbatch_ce_logits_bytes(int64_t M, int64_t N);
int tt_inbatch_ce_fwd_du_keep(const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M, int64_t N,
                              int64_t D, int64_t diag_offset, float* row_lse, float* row_ce, float* du_unit,
                              int64_t ld_du, float* logits, int64_t logits_bytes, void* ws, int64_t ws_bytes,
                              tt_stream_t stream);
int tt_inbatch_ce_bwd_kept(const float* U, int64_t ldu, int64_t M, int64_t N, int64_t D, int64_t diag_offset,
                           const float* row_lse, const float* coef, const float* logits, int64_t logits_bytes,
                           float* dI, int64_t lddi, void* ws, int64_t ws_bytes, tt_stream_t stream);

/* EXPLORATORY, opt-in (the sharded trainer takes it with TT_CE_F16X2=1; nothing takes it by default): the kept-logits pair
 * above on the 16-bit matrix pipe at fp32-grade accuracy (csrc/ce_f16x2.hip).  Every fp32 operand is cut into two fp16
 * terms after a power-of-two scale (11 + 11 significant bits) and every product runs as three v_mfma_f32_32x32x16_f16
 * into one fp32 accumulator: element-wise error at the level of an fp32 fma chain (tools/f16x2_logits_probe.hip), a
 * fifth of the matrix-pipe cycles.  Same arguments, meaning and outputs as tt_inbatch_ce_fwd_du_keep /
 * tt_inbatch_ce_bwd_kept -- same reference lines: ref:src/two_tower_base_retrieval.py:287-312 -- except: `logits`
 * is M * N * 4 bytes holding [M / 32][N / 32] tiles of 32 users x 32 items, each tile 4 KiB contiguous and row-major
 * inside (log2-domain logits), row_lse is in the log2 domain as well (only the backward consumes it), the workspace is
 * tt_ce16_workspace_bytes, and the shapes are D = 128,
 * M % 256 == 0, N % 1024 == 0 (tt_ce16_supported; TT_E_UNSUPPORTED otherwise). */
int tt_ce16_supported(int64_t M, int64_t N, int64_t D);
int64_t tt_ce16_workspace_bytes(int64_t M, int64_t N, int64_t D);
int tt_ce16_fwd_du_keep(const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M, int64_t N, int64_t D,
                        int64_t diag_offset, float* row_lse, float* row_ce, float* du_unit, int64_t ld_du, float* logits,
                        int64_t logits_bytes, void* ws, int64_t ws_bytes, tt_stream_t stream);
int tt_ce16_bwd_kept(const float* U, int64_t ldu, int64_t M, int64_t N, int64_t D, int64_t diag_offset,
                     const float* row_lse, const float* coef, const float* logits, int64_t logits_bytes, float* dI,
                     int64_t lddi, void* ws, int64_t ws_bytes, tt_stream_t stream);
/* The same backward WITHOUT kept logits: tt_ce16_fwd_du_keep accepts logits = NULL (nothing is written), and this call
 * forms every 32 x 32 logits tile again on the fp16 pipe (24 more matrix instructions per tile) instead of reading it
 * back -- 2 * 4 * M * N bytes of HBM traffic less per step, half of what the W = 8 step with this pair moved
 * (DESIGN section 5).  Needs the item rows as well.  reuse_images != 0: `ws` still holds the images and scales the
 * forward call formed from THESE U and I (nothing else touched it) -- they are not formed again. */
int tt_ce16_bwd_recompute(const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M, int64_t N, int64_t D,
                          int64_t diag_offset, const float* row_lse, const float* coef, float* dI, int64_t lddi, void* ws,
                          int64_t ws_bytes, int reuse_images, tt_stream_t stream);

/* dU[i, :] = du_unit[i, :] * coef[i]: the chain-rule step that turns tt_inbatch_ce_fwd_du's unit gradient into the
 * user-side gradient once dL/dce is known (autograd of ref:...base_retrieval.py:287-312,342). */
int tt_scale_rows(const float* x, int64_t ldx, const float* coef, int64_t rows, int64_t D, float* out, int64_t ldo,
                  tt_stream_t stream);

/* net_user_value weights, ref:...base_retrieval.py:322,334-339 for 2-D labels:
 *   nuv[i] = sum_t labels[i,t]*uvw[t];  w = clamp(nuv,1e-6);  w /= max_i w
 * then loss = mean_i(row_ce[i]*w[i]) and coef[i] = w[i]/B (gradient seed 1).
 * labels == NULL: every weight is 1 -- what the reference computes for train.py's 1-D [B] labels
 * (ref:train/train.py:53-55,78: labels*uvw sums to ONE scalar, which clamp and /max turn into 1.0). */
int tt_weighted_mean_loss(const float* labels, int64_t B, int64_t T, const float* uvw,
                          const float* row_ce, float* w_out, float* coef_out, float* loss_out,
                          tt_stream_t stream);

/* The same head cut at its two batch-wide reductions, for a batch that is split over ranks (row-sharded step: every rank
 * holds B of the global rows; ref:...base_retrieval.py:322,334-343 on the concatenated batch):
 *   tt_value_weights         nuv[i] = clamp(sum_t labels[i,t]*uvw[t], 1e-6);  *max_out = max_i nuv      (-> all-reduce MAX)
 *   tt_weighted_loss_global  w = nuv / *gmax;  coef[i] = w[i] / denom;  *loss_out = sum_i row_ce[i]*w[i] / denom
 *                            (denom = global batch size; -> all-reduce SUM of the scalar). */
int tt_value_weights(const float* labels, int64_t B, int64_t T, const float* uvw, float* nuv_out, float* max_out,
                     tt_stream_t stream);
int tt_weighted_loss_global(const float* nuv, const float* gmax, const float* row_ce, int64_t B, float denom,
                            float* coef_out, float* loss_out, tt_stream_t stream);

/* tt_inbatch_ce_fwd_du followed by tt_weighted_mean_loss, the loss head running in the forward's finishing launch
 * (ref:src/two_tower_base_retrieval.py:287-312 logits + cross entropy, :322,334-343 value weights + mean): same
 * outputs as the two calls, the loss bit-identical to theirs; needs M user rows = B label rows.  `ws`:
 * tt_inbatch_ce_workspace_bytes(M, N, D).  tt_scale_rows_g is the matching first step of the backward:
 *   coef_g[i] = coef[i] * g;  out[i, :] = x[i, :] * coef_g[i]      (g: device scalar dL/dloss)
 * i.e. dU from du_unit and the row factors the item-side backward (tt_inbatch_ce_bwd) takes as `coef`. */
int tt_inbatch_ce_fwd_du_loss(const float* U, int64_t ldu, const float* I, int64_t ldi, int64_t M, int64_t N, int64_t D,
                              int64_t diag_offset, const float* labels, int64_t T, const float* uvw, float* row_lse,
                              float* row_ce, float* du_unit, int64_t ld_du, float* w_out, float* coef_out, float* loss_out,
                              void* ws, int64_t ws_bytes, tt_stream_t stream);
int tt_scale_rows_g(const float* x, int64_t ldx, const float* coef, const float* g, int64_t rows, int64_t D, float* out,
                    int64_t ldo, float* coef_g, tt_stream_t stream);

/* Combined debias loss head (ref:src/two_tower_with_debiasing.py:77-129 on top of
 * ref:src/two_tower_base_retrieval.py:322-345), fused -- SURVEY 8f item 2:
 *   n = labels.uvw;  p = pos_table[position];  e = <user_emb, lin_w[:DI]> + p*lin_w[DI] + lin_b
 *   aux = sum_i (e_i-n_i)^2 + sum_i sum_j (p_i-n_j)^2   (upstream's [B,1]-vs-[B] broadcast, in closed form)
 *   r = max(n / max(e, 1e-3), 1e-6);  loss = mean_i(row_ce_i * r_i / max_j r_j) + aux
 * The forward leaves n, p, e, r and the reduction scalars in `ws` (tt_debias_loss_workspace_bytes(B, DI, n_pos)),
 * which the backward reads; `grad_loss` is the device scalar dL/dloss.  Gradients as torch defines
 * them: clamp(min) passes where input >= min, max() splits evenly over ties.  Positions outside
 * [0, n_pos) raise *oob_flag (torch: IndexError) and read row 0.  d_user_emb is written, not added.
 * `mode` (round 4): TT_DEBIAS_COMBINED the above; TT_DEBIAS_POSITION the position-only sibling
 * (ref:src/two_tower_with_position_debiased_weights.py:76-113: r = max(n / max(p, 1e-3), 1e-6), aux = sum (p - n)^2, lin_w /
 * lin_b unused, d_user_emb written as zeros); TT_DEBIAS_USER the user-only sibling
 * (ref:src/two_tower_with_user_debiased_weights.py:100-135: c = max(<user_emb, lin_w[:DI]> + lin_b, 1e-1) first, aux =
 * sum (c - n)^2, r = max(n / c, 1e-6); lin_w has DI entries, position / pos_table unused and may be NULL). */
#define TT_DEBIAS_COMBINED 0
#define TT_DEBIAS_POSITION 1
#define TT_DEBIAS_USER 2
int64_t tt_debias_loss_workspace_bytes(int64_t B, int64_t DI, int64_t n_pos);
int tt_debias_loss_fwd(int mode, const float* row_ce, const float* labels, int64_t B, int64_t T, const float* uvw,
                       const int64_t* position, int64_t n_pos, const float* pos_table, const float* user_emb,
                       int64_t ld_ue, int64_t DI, const float* lin_w /*[DI+1]*/, const float* lin_b /*[1]*/,
                       float* loss_out, void* ws, int64_t ws_bytes, int32_t* oob_flag, tt_stream_t stream);
int tt_debias_loss_bwd(int mode, const float* grad_loss, const float* row_ce, int64_t B, const int64_t* position,
                       int64_t n_pos, const float* user_emb, int64_t ld_ue, int64_t DI, const float* lin_w,
                       const void* ws, int64_t ws_bytes, float* d_row_ce, float* d_user_emb, int64_t ld_due,
                       float* d_pos_table /*[n_pos]*/, float* d_lin_w /*[DI+1]*/, float* d_lin_b /*[1]*/,
                       tt_stream_t stream);

/* ---------------------------------------------------------------- K2 row-gradient plan + dense-exact Adam
 * The embedding backward of the reference is a dense [n_rows,dim] gradient
 * (autograd embedding_dense_backward) consumed by optim.Adam over EVERY row
 * (ref:train/train.py:123-125,179).  Here the gradient stays in row form:
 *   tt_rowgrad_plan   stable-sorts the looked-up ids, finds the unique rows and
 *                     their occurrence lists  (deterministic, atomic-free)
 *   tt_rowgrad_dense  materialises the dense gradient (for torch.optim users)
 *   tt_adam_table     one dense-exact Adam step on the whole table: every row
 *                     gets the zero-gradient update, rows that were looked up
 *                     get the update with their summed gradient.
 * Plan outputs (device): sorted_ids int32[n], perm int32[n] (original position
 * of each sorted id), seg_begin int32[n+1] (start of each unique row's run in
 * the sorted order; seg_begin[n_unique] == n), n_unique int32[1].
 * Sentinel rows: tt_adam_table / _stash / _finish ignore runs whose row id is >= the
 * n_rows THEY are given, so a caller may plan with n_rows+1 and map "not my row" ids to
 * n_rows (the sharded trainer does, after all-gathering every rank's ids). */
int64_t tt_rowgrad_workspace_bytes(int64_t n_ids);
int tt_rowgrad_plan(const int64_t* ids, int64_t n_ids, int64_t n_rows, int32_t* sorted_ids,
                    int32_t* perm, int32_t* seg_begin, int32_t* n_unique, int32_t* oob_flag,
                    void* ws, int64_t ws_bytes, tt_stream_t stream);
/* tt_rowgrad_plan for SEVERAL tables' id lists in one launch (one workgroup per list): lists of at most 10 240 ids
 * (tt_rowgrad_plan_jobs_supported), which is every lookup of the base model up to B = 10 240 -- the two sorts of a step
 * (user ids, item ids) otherwise run back to back.  Same outputs as tt_rowgrad_plan, no workspace. */
#define TT_PLAN_MAX_JOBS 4
typedef struct {
  const int64_t* ids; int64_t n_ids; int64_t n_rows;
  int32_t *sorted_ids, *perm, *seg_begin, *n_unique;
} tt_plan_job;
int tt_rowgrad_plan_jobs_supported(int64_t n_ids);
int tt_rowgrad_plan_jobs(const tt_plan_job* jobs /*host*/, int32_t n_jobs, int32_t* oob_flag, tt_stream_t stream);

/* up to TT_MAX_GRAD_SOURCES row-gradient blocks; occurrence p in [0,n_ids) lives in
 * the block whose [first, first+rows) range contains p (e.g. item_id rows then
 * history rows for the item table). */
#define TT_MAX_GRAD_SOURCES 4
typedef struct {
  const float* rows[TT_MAX_GRAD_SOURCES];
  int64_t ld[TT_MAX_GRAD_SOURCES];
  int64_t first[TT_MAX_GRAD_SOURCES + 1]; /* first[k+1]-first[k] = rows in block k */
  int32_t n_sources;
} tt_grad_sources;

int tt_rowgrad_dense(const tt_grad_sources* src /*host*/, int64_t n_ids, int64_t dim,
                     const int32_t* sorted_ids, const int32_t* perm, const int32_t* seg_begin,
                     const int32_t* n_unique, float* dense_grad /*[n_rows,dim], pre-zeroed*/,
                     tt_stream_t stream);

/* Adam hyper-parameters + step live in DEVICE memory (8 doubles) so that a
 * captured hipGraph replays with a moving step count:
 *   [0] lr [1] beta1 [2] beta2 [3] eps [4] step [5] lr/(1-beta1^step)
 *   [6] sqrt(1-beta2^step) [7] library scratch (the sweep's chunk counters; initialise to 0).
 * tt_adam_advance bumps [4] and recomputes [5],[6] in double, as torch.optim.Adam does on the host. */
int tt_adam_advance(double* hyper, tt_stream_t stream);

int64_t tt_adam_table_workspace_bytes(int64_t n_ids, int64_t dim);
int tt_adam_table(float* W, float* M, float* V, int64_t n_rows, int64_t dim, const double* hyper,
                  const tt_grad_sources* src /*host*/, int64_t n_ids, const int32_t* sorted_ids,
                  const int32_t* perm, const int32_t* seg_begin, const int32_t* n_unique,
                  void* ws, int64_t ws_bytes, tt_stream_t stream);

/* The same table step in three phases, for the overlapped schedule: the zero-gradient sweep
 * does not depend on this step's gradients, only on the lookups having finished.
 *   tt_adam_table_stash   park the OLD p,m,v of the looked-up rows in `side` (needs the plan; one
 *                         copy per unique row).  `side` = three planes [n_ids][dim] (p | m | v);
 *                         a row's slot is the position of its FIRST occurrence in the id list
 *   tt_adam_table_stash_ids  the same from the raw id list, no plan needed: occurrence i parks
 *                         row ids[i] in slot i (ids outside [0, n_rows) park zeros).  The sweep
 *                         can then start before the ids are sorted, and the p plane IS the
 *                         lookup result of the step's forward (rows in id-list order), which
 *                         must read it instead of the table the sweep is rewriting
 *   tt_adam_table_sweep   every row, gradient = 0 -- run it on a SECOND stream, concurrently
 *                         with the backward pass (HBM-bound vs MFMA/latency-bound)
 *   tt_adam_table_finish  Adam on the looked-up rows from `side` + their summed gradients,
 *                         then write them over the swept rows (after the sweep completed)
 * stash -> sweep -> finish leaves bit-identical results to tt_adam_table.  `side` needs
 * tt_adam_table_workspace_bytes(n_ids, dim) and must survive from stash to finish. */
int tt_adam_table_stash(const float* W, const float* M, const float* V, int64_t n_rows, int64_t dim,
                        int64_t n_ids, const int32_t* sorted_ids, const int32_t* perm,
                        const int32_t* seg_begin, const int32_t* n_unique, void* side, int64_t side_bytes,
                        tt_stream_t stream);
int tt_adam_table_stash_ids(const float* W, const float* M, const float* V, int64_t n_rows, int64_t dim,
                            const int64_t* ids, int64_t n_ids, void* side, int64_t side_bytes,
                            tt_stream_t stream);
int tt_adam_table_sweep(float* W, float* M, float* V, int64_t n_rows, int64_t dim, const double* hyper,
                        tt_stream_t stream);
/* tt_adam_table_sweep for up to 4 tables in ONE launch (descriptor .p/.m/.v/.n = weights, moments, element count;
 * .g unused): the chunk list spans the tables, so a step has one sweep launch and one tail.  Bit-identical to the
 * per-table calls.  n_wgs > 0 caps the number of persistent workgroups (a THROTTLED sweep: when the step is much longer
 * than the sweep, a thin sweep that lasts most of the step takes less HBM bandwidth from the forward / backward kernels
 * at any moment than a saturating one at its start); 0 = the default (3 per CU). */
struct tt_adam_tensor_s;
int tt_adam_tables_sweep(const struct tt_adam_tensor_s* tables /*host*/, int32_t n_tables, const double* hyper,
                         int32_t n_wgs, tt_stream_t stream);
/* Steps that look up MANY rows (the history model: 213 K of 1 M item rows) MARK them instead of parking them: one bit per
 * row in `marks` (tt_adam_marks_words(n_rows) 32-bit words, cleared and set by tt_adam_mark_rows from the step's id list; ids
 * outside [0, n_rows) are ignored), tt_adam_tables_sweep_marked steps over marked rows (marks[t] == NULL: no row of table t
 * is marked), and tt_adam_table_finish / tt_adam_tables_finish with side == NULL read the rows' old p, m, v from the table
 * itself.  Same arithmetic per row as the parked schedule: the same bits.  dims[t] = row width, a power of two in
 * [32, 4096] (tt_adam_marked_supported), 16-byte aligned tables.  (ref:train/train.py:123-125: the dense Adam step.) */
int64_t tt_adam_marks_words(int64_t n_rows);
int tt_adam_marked_supported(int64_t dim);
int tt_adam_mark_rows(const int64_t* ids, int64_t n_ids, int64_t n_rows, uint32_t* marks, int64_t marks_words,
                      tt_stream_t stream);
int tt_adam_tables_sweep_marked(const struct tt_adam_tensor_s* tables /*host*/, const int64_t* dims /*host*/,
                                const uint32_t* const* marks /*host array of device pointers*/, int32_t n_tables,
                                const double* hyper, int32_t n_wgs, tt_stream_t stream);
int tt_adam_table_finish(float* W, float* M, float* V, int64_t n_rows, int64_t dim, const double* hyper,
                         const tt_grad_sources* src /*host*/, int64_t n_ids, const int32_t* sorted_ids,
                         const int32_t* perm, const int32_t* seg_begin, const int32_t* n_unique,
                         void* side, int64_t side_bytes, tt_stream_t stream);

/* The two ends of an overlapped step, each as ONE launch over all tables (the step has three tiny dependent launches in
 * front of the sweep and three behind it otherwise; at 1 M-row tables the sweep's start is the critical path):
 *   tt_adam_begin_ids      = tt_adam_advance (tab == NULL) or tt_adam_advance_tab, then tt_adam_table_stash_ids per job
 *   tt_adam_tables_finish  = tt_adam_table_finish per job
 * Same results as the separate calls, which they fall back to for more than 4 tables / mixed row widths / unaligned rows.
 * Replaces the same reference lines (ref:train/train.py:123-125, optimizer.step()). */
typedef struct {
  const float *W, *M, *V;
  int64_t n_rows, dim;
  const int64_t* ids;
  int64_t n_ids;
  void* side;
  int64_t side_bytes;
} tt_adam_stash_job;
typedef struct {
  float *W, *M, *V;
  int64_t n_rows, dim;
  const tt_grad_sources* src;
  int64_t n_ids;
  const int32_t *sorted_ids, *perm, *seg_begin, *n_unique;
  void* side;
  int64_t side_bytes;
} tt_adam_finish_job;
int tt_adam_begin_ids(double* hyper, float* tab, int64_t tab_steps, const tt_adam_stash_job* jobs, int32_t n_jobs,
                      tt_stream_t stream);
int tt_adam_tables_finish(const tt_adam_finish_job* jobs, int32_t n_jobs, const double* hyper, tt_stream_t stream);
/* tt_adam_begin_ids in two halves, for steps whose lookups are large (the history model parks 217 K rows: 0.67 GB): `planes`
 * is a mask of 1 = p (+ the step-count advance), 2 = m, 4 = v.  The forward's lookups read the parked p plane only, so a
 * caller runs planes = 1 on its own stream and planes = 6 on the sweep's stream in front of the sweep -- the moments leave
 * the step's critical path.  planes = 7 is tt_adam_begin_ids.  At most 4 tables. */
int tt_adam_begin_ids_planes(double* hyper, float* tab, int64_t tab_steps, const tt_adam_stash_job* jobs, int32_t n_jobs,
                             int32_t planes, tt_stream_t stream);

/* A HIP stream of the device's least priority (hipStreamCreateWithPriority) for
 * tt_adam_table_sweep, so the backward pass on the caller's stream is dispatched first. */
int tt_stream_create_low_priority(void** out_stream);
int tt_stream_destroy(void* stream);

/* Deferred ("lazy") schedule of the SAME dense Adam -- value-exact, reported separately from the
 * dense-sweep figure.  A row that receives no gradient evolves by a recurrence that
 * needs only the row and the per-step constants, so its zero-gradient steps are replayed in
 * registers (same fp32 operations, same order) when the row is next needed instead of being
 * applied by a sweep every step; results are bit-identical to tt_adam_table.
 *   last_step[n_rows] (int32, zero-initialised): the step each row is current for.
 *   tab[2*tab_steps] (float): per-step constants, maintained by tt_adam_advance_tab (use it
 *     instead of tt_adam_advance); steps beyond the table are recomputed in double in-kernel.
 *   tt_adam_rows_catchup  rows ids[0..n) (duplicates allowed) -> current step.  Call before any
 *                         lookup reads them.
 *   tt_adam_table_lazy    this step's gradient update on the planned rows only (after
 *                         tt_adam_advance_tab), rows first brought to the previous step.
 *   tt_adam_table_flush   every row -> current step: before anything reads the table as a whole
 *                         (checkpoint, export).  */
int tt_adam_advance_tab(double* hyper, float* tab, int64_t tab_steps, tt_stream_t stream);
int tt_adam_rows_catchup(float* W, float* M, float* V, int64_t n_rows, int64_t dim, const int64_t* ids,
                         int64_t n_ids, int32_t* last_step, const double* hyper, const float* tab,
                         int64_t tab_steps, tt_stream_t stream);
int tt_adam_table_lazy(float* W, float* M, float* V, int64_t n_rows, int64_t dim, const double* hyper,
                       const tt_grad_sources* src /*host*/, int64_t n_ids, const int32_t* sorted_ids,
                       const int32_t* perm, const int32_t* seg_begin, const int32_t* n_unique, void* ws,
                       int64_t ws_bytes, int32_t* last_step, const float* tab, int64_t tab_steps,
                       tt_stream_t stream);
int tt_adam_table_flush(float* W, float* M, float* V, int64_t n_rows, int64_t dim, int32_t* last_step,
                        const double* hyper, const float* tab, int64_t tab_steps, tt_stream_t stream);

/* dense parameters: `tensors` is a HOST array of n_tensors {p,g,m,v,n} descriptors
 * (device pointers inside); they are passed to the kernel by value, 64 per launch. */
typedef struct tt_adam_tensor_s {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
} tt_adam_tensor;
int tt_adam_dense(const tt_adam_tensor* tensors /*host*/, int32_t n_tensors, const double* hyper,
                  tt_stream_t stream);
/* Row-sharded training: copy each replicated parameter's gradient (`g`, n floats) into its slice
 * (`p`) of ONE flat buffer -- the operand of the single dense-gradient all-reduce; m, v are ignored.  One launch per 64
 * tensors.  The reference has no counterpart (single process, ref:train/train.py:123-125). */
int tt_pack_grads(const tt_adam_tensor* tensors /*host*/, int32_t n_tensors, tt_stream_t stream);
/* Whole-step hipGraph (graphs.GraphedTrainStep): copy a new batch's input tensors into the captured static buffers in ONE
 * launch -- `p` = destination, `g` = source, `n` = BYTES (any dtype), m / v ignored.  Replaces the seven `.to(device)` /
 * copy_ calls of ref:train/train.py:91-99 per step. */
int tt_copy_buffers(const tt_adam_tensor* buffers /*host*/, int32_t n_buffers, tt_stream_t stream);
/* Measurement aid (bench.py `roofline.hbm_copy_GBps`; no counterpart in the reference): the plain HBM streaming copy --
 * 16-byte non-temporal loads and stores, one 16 KB chunk per workgroup -- that calibrates what THIS box's HBM streams next
 * to the sweep's own figure.  `bytes` are read and `bytes` written. */
int tt_stream_copy(const void* src, void* dst, int64_t bytes, tt_stream_t stream);
/* Measurement aid (bench.py `roofline.sustained_peak`): a register-only MFMA loop on random operands over the whole chip
 * (512 workgroups x 4 waves x `iters` x 4 independent 32x32 MFMAs; dtype 0 = fp32 32x32x2, 1 = bf16 32x32x16 -- TT_F32 /
 * TT_BF16 below) -- what the matrix pipe of THIS box sustains at its power budget, which is what a kernel priced against
 * the spec peak can reach at most.  tt_mfma_probe_flops = the flops one launch executes; time it with events. */
int64_t tt_mfma_probe_flops(int dtype, int32_t iters);
int tt_mfma_probe(int dtype, int32_t iters, float* sink /* >= 131072 floats */, int64_t sink_floats, tt_stream_t stream);

/* ---------------------------------------------------------------- K4 history encoder pieces
 * tt_hist_embed_pool: x[b,h,:] = table[ids[b,h],:] (+ pe[h,:]);  pooled[b,:] = mean_h table[ids[b,h],:]
 *   (mean of the RAW rows, ref:src/user_history_encoder.py:89; PE add :91-95).
 *   If `ids` is NULL, `table` is read as an already-embedded [B,H,dim] tensor
 *   (the UserHistoryEncoder.forward([B,H,DI]) entry, ref:...encoder.py:80).
 * tt_attn_fwd / tt_attn_bwd: unmasked multi-head softmax attention over one
 *   sample's H rows, heads split along columns of the packed projection
 *   qkv[B*H, 3D] = [Q | K | V]  (nn.MultiheadAttention internals,
 *   ref:...encoder.py:103-108; algebra in SURVEY.md 3.3).  ctx is [B*H, D]
 *   (heads concatenated), lse is [B, heads, H]. */
int tt_hist_embed_pool(const float* table, int64_t n_rows, int64_t dim, const int64_t* ids,
                       int64_t B, int64_t H, const float* pe, float* x, float* pooled,
                       int64_t ld_pooled, int32_t* oob_flag, tt_stream_t stream);
/* The first encoder layer's input gradient and the mean pool's backward in ONE pass:
 * dx[b, h, :] = dqkv[b, h, :] . w_in + d_pooled[b, :] / H   (dqkv [B*H, 3D], w_in [3D, D]; autograd of
 * ref:src/user_history_encoder.py:103-116: the in-projection's input gradient plus the mean over H) -- the row-group term
 * rides in the product's epilogue, so dx is written once and never read back (tt_gemm_f32 + tt_hist_pool_bwd otherwise).
 * D = 128, B*H >= 16384, 16-byte aligned operands; TT_E_UNSUPPORTED otherwise. */
int tt_hist_dx_pool_bwd(const float* dqkv, const float* w_in, int64_t B, int64_t H, int64_t D, const float* d_pooled,
                        int64_t ld_pooled, float* dx, tt_stream_t stream);

/* backward of the mean pool (ref:...encoder.py:89): dx[b,h,:] += d_pooled[b,:] / H */
int tt_hist_pool_bwd(float* dx, int64_t B, int64_t H, int64_t dim, const float* d_pooled,
                     int64_t ld_pooled, tt_stream_t stream);
int tt_attn_fwd(const float* qkv, int64_t B, int64_t H, int64_t D, int64_t heads, float* ctx,
                float* lse, tt_stream_t stream);
int tt_attn_bwd(const float* qkv, const float* ctx, const float* lse, const float* d_ctx,
                int64_t B, int64_t H, int64_t D, int64_t heads, float* d_qkv, tt_stream_t stream);
/* The encoder's LAST layer is consumed at row 0 only (ref:src/user_history_encoder.py:103-116: only row 0 of the last
 * nn.MultiheadAttention's output is used): it runs WITHOUT projecting K and V.  x [B*H, D] is the layer's input, w_in
 * [3D, D] / b_in [3D] / w_out [D, D] / b_out [D] its packed parameters.  With q0 = W_q x[b,0] + b_q:
 * score_h[j] = scale (W_k,h^T q0_h) . x[b,j]  (the K bias shifts every score of a head alike and drops out of the softmax)
 * and ctx0_h = W_v,h (sum_j p_h[j] x[b,j]) + b_v,h, so each sample's rows are read once and the [B*H, 2D] projection
 * never exists.  Forward writes recent [B, D] (row stride ld_recent) = W_out ctx0 + b_out and, for the backward, q0
 * [B, D], t [B, heads, D] (= W_k,h^T q0_h), probs [B, heads, H], xbar [B, heads, D], ctx0 [B, D].  Backward writes dx
 * [B*H, D] (every row), dW_in [3D, D], db_in [3D] (the K third exactly zero), dW_out [D, D], db_out [D]; weight gradients
 * are per-workgroup partial sums added in a fixed order (deterministic).
 * A caller whose PREVIOUS layer hands over its attention context c instead of its output c W_o^T + b_o passes x = c with
 * the composed parameters w_in = W_in W_o, b_in = W_in b_o + b_in (q, k, v are linear in the never-formed output) and maps
 * dW_in / db_in back itself (ops.HistoryEncoder).
 * Shapes: H <= 64, D <= 128, D % 4 == 0, D % heads == 0, (D / heads) % 4 == 0, heads <= 16 (tt_enc_last_supported);
 * x, dx, d_recent, the weights, t, xbar and ws 16-byte aligned. */
int tt_enc_last_supported(int64_t H, int64_t D, int64_t heads);
int tt_enc_last_fwd(const float* x, int64_t B, int64_t H, int64_t D, int64_t heads, const float* w_in,
                    const float* b_in, const float* w_out, const float* b_out, float* recent, int64_t ld_recent,
                    float* q0, float* t, float* probs, float* xbar, float* ctx0, tt_stream_t stream);
int64_t tt_enc_last_bwd_workspace_bytes(int64_t B, int64_t H, int64_t D, int64_t heads);
int tt_enc_last_bwd(const float* x, int64_t B, int64_t H, int64_t D, int64_t heads, const float* w_in,
                    const float* w_out, const float* d_recent, int64_t ld_dr, const float* q0, const float* t,
                    const float* probs, const float* xbar, const float* ctx0, float* dx, float* dW_in, float* db_in,
                    float* dW_out, float* db_out, void* ws, int64_t ws_bytes, tt_stream_t stream);
/* tt_enc_last_bwd in two halves sharing ONE workspace (same size; it must stay untouched between the two calls):
 *   _data     dx (everything the rest of the backward pass waits for)
 *   _weights  dW_in, db_in, dW_out, db_out from what _data left in the workspace -- feeds nothing but the optimiser, so a
 *             caller may enqueue it on another stream, behind _data
 * tt_enc_last_bwd = the two in a row on one stream; bit-identical results either way. */
int tt_enc_last_bwd_data(const float* x, int64_t B, int64_t H, int64_t D, int64_t heads, const float* w_in,
                         const float* w_out, const float* d_recent, int64_t ld_dr, const float* t, const float* probs,
                         float* dx, void* ws, int64_t ws_bytes, tt_stream_t stream);
int tt_enc_last_bwd_weights(const float* x, int64_t B, int64_t H, int64_t D, int64_t heads, const float* d_recent,
                            int64_t ld_dr, const float* q0, const float* xbar, const float* ctx0, float* dW_in, float* db_in,
                            float* dW_out, float* db_out, void* ws, int64_t ws_bytes, tt_stream_t stream);

/* ---------------------------------------------------------------- R1 owner routing (row-sharded tables)
 * New design -- the reference has no parallelism.  Tables are split into `world`
 * contiguous blocks of rows_per_rank rows; a rank asks each owner only for the ids that owner holds,
 * through a padded fixed-capacity all-to-all (cap slots per peer).  Semantics to keep: the lookups of
 * ref:src/two_tower_base_retrieval.py:126,209 and ref:src/two_tower_with_user_history_encoder.py:105.
 * Bucketing a rank's ids by owner (= id / rows_per_rank) is one stable counting pass; slots are assigned in
 * list order, so an owner sees the ids of one requester in request order.
 *   tt_route_count    counts[o] (int32 [world]) = ids owned by rank o; *max_count (device int32, atomicMax'ed:
 *                     zero it first) = the largest bucket -- the caller all-reduces it (MAX) into `cap`.  Leaves
 *                     per-tile offsets in `ws` (tt_route_workspace_bytes) for tt_route_build.  Ids outside
 *                     [0, n_rows) set *oob_flag (may be NULL) and are routed as row 0.
 *   tt_route_build    (same ids, same ws) send_ids[o*cap + r] = r-th id owned by o (-1 = padding); slot_of[i] =
 *                     slot in which the row of the caller's i-th id comes back; src_of[slot] = i (-1 = padding):
 *                     the backward sends gradient row src_of[slot] in that slot.  A bucket larger than cap sets
 *                     *overflow_flag and slot_of = -1 for the ids that did not fit (cannot happen when cap
 *                     comes from the all-reduced count).  world <= 1024.
 *   tt_route_localize owner side: local[i] = ids[i] - lo for lo <= ids[i] < lo + n_local, else the
 *                     sentinel n_local (padding, foreign ids): tt_gather_rows then yields a zero row and
 *                     the Adam kernels skip the run. */
int64_t tt_route_workspace_bytes(int64_t n_ids, int32_t world);
int tt_route_count(const int64_t* ids, int64_t n_ids, int64_t n_rows, int64_t rows_per_rank, int32_t world,
                   int32_t* counts, int32_t* max_count, int32_t* oob_flag, void* ws, int64_t ws_bytes,
                   tt_stream_t stream);
int tt_route_build(const int64_t* ids, int64_t n_ids, int64_t n_rows, int64_t rows_per_rank, int32_t world,
                   int64_t cap, const void* ws, int64_t ws_bytes, int64_t* send_ids, int64_t* slot_of,
                   int64_t* src_of, int32_t* overflow_flag, tt_stream_t stream);
int tt_route_localize(const int64_t* ids, int64_t n_ids, int64_t lo, int64_t n_local, int64_t* local,
                      tt_stream_t stream);
/* The same stages for ALL lookups of a step per launch (a step routes 2-3 id lists of a few thousand ids: 3-7 us kernels
 * that stood 6-15 us apart).  Up to TT_ROUTE_MAX_JOBS jobs per call; host arrays of descriptors, device pointers inside.
 *   tt_route_count_jobs  tt_route_count per job in two launches for all jobs; *max_count is WRITTEN (not atomicMax'ed:
 *                        nothing to zero first).
 *   tt_route_build_jobs  tt_route_build per job in ONE launch, the -1 fill of send_ids / src_of included (a slot
 *                        o*cap + r is padding iff r >= counts[o], which the count stage left in `counts`).
 *   tt_route_serve_jobs  owner side, ONE launch: tt_route_localize + the row gather of every received id list --
 *                        local[i] as above, rows[i] = table[local[i]] widened to fp32 (dtype TT_F32 / TT_BF16 = the
 *                        table's storage) or a zero row for the sentinel. */
#define TT_ROUTE_MAX_JOBS 8
typedef struct {
  const int64_t* ids;    /* this rank's ids [n_ids] */
  int64_t n_ids, n_rows, rows_per_rank;
  int32_t* counts;       /* [world]: out of _count_jobs, in of _build_jobs */
  int32_t* max_count;    /* [1]: out of _count_jobs */
  void* ws;              /* tt_route_workspace_bytes(n_ids, world): written by _count_jobs, read by _build_jobs */
  int64_t ws_bytes;
  int64_t cap;           /* _build_jobs only (ignored by _count_jobs), with the three outputs below */
  int64_t* send_ids;     /* [world * cap] */
  int64_t* slot_of;      /* [n_ids] */
  int64_t* src_of;       /* [world * cap] */
} tt_route_job;
typedef struct {
  const int64_t* ids;    /* received global ids [n_ids] (-1 = padding) */
  int64_t n_ids, lo, n_local;
  int64_t* local;        /* out [n_ids] */
  const void* table;     /* this rank's row block [n_local, dim], fp32 or bf16 */
  int dtype;             /* TT_F32 / TT_BF16 */
  int64_t dim;
  float* rows;           /* out [n_ids, dim] */
} tt_route_serve_job;
int tt_route_count_jobs(const tt_route_job* jobs /*host*/, int32_t n_jobs, int32_t world, int32_t* oob_flag, tt_stream_t stream);
int tt_route_build_jobs(const tt_route_job* jobs /*host*/, int32_t n_jobs, int32_t world, int32_t* overflow_flag,
                        tt_stream_t stream);
int tt_route_serve_jobs(const tt_route_serve_job* jobs /*host*/, int32_t n_jobs, tt_stream_t stream);

/* ---------------------------------------------------------------- R collectives (RCCL over xGMI)
 * New design (SURVEY.md 2b R1-R4, 8b "tt_comm_*"): the reference has no communication.  One communicator per
 * process (= per GPU: the current HIP device at tt_comm_init), created from a 128-byte id that rank 0 obtains
 * with tt_comm_unique_id and hands to the other ranks by any host-side channel.  The handle is the ONLY
 * long-lived native state of this library.  RCCL is bound at run time (librccl.so.1; TT_RCCL_PATH overrides):
 * without it these return TT_E_UNSUPPORTED and everything else keeps working.  RCCL failures return
 * -(100 + ncclResult_t).  All calls are asynchronous on `stream`; counts are in ELEMENTS of `dtype`; a count
 * of 0 is a no-op that succeeds (pointers may then be null), as an empty tensor is for torch.distributed.
 *   tt_comm_alltoall       chunk r of `send` (count_per_peer elements) goes to rank r; chunk r of `recv` came
 *                          from rank r -- routed lookups: ids, rows, row gradients (R1)
 *   tt_comm_allgather      item embeddings for the global in-batch negatives (R2)
 *   tt_comm_reduce_scatter partial dI over the gathered items -> this rank's block (R2)
 *   tt_comm_allreduce      dense gradients (SUM), value-weight / bucket maxima (MAX), loss (SUM) (R3); in place
 *                          when send == recv
 *   tt_comm_broadcast      replicated parameters at start-up */
#define TT_COMM_ID_BYTES 128
#define TT_COMM_F32 0
#define TT_COMM_I32 1
#define TT_COMM_I64 2
#define TT_COMM_U8 3
#define TT_COMM_SUM 0
#define TT_COMM_MAX 1
typedef struct tt_comm_s* tt_comm_t;
int tt_comm_unique_id(void* id_out /* host, TT_COMM_ID_BYTES */);
int tt_comm_init(const void* id /* host, TT_COMM_ID_BYTES */, int32_t rank, int32_t world, tt_comm_t* out);
int tt_comm_destroy(tt_comm_t comm);
int tt_comm_size(tt_comm_t comm, int32_t* rank_out, int32_t* world_out /* as RCCL reports it */);
int tt_comm_alltoall(tt_comm_t comm, const void* send, void* recv, int64_t count_per_peer, int dtype,
                     tt_stream_t stream);
int tt_comm_allgather(tt_comm_t comm, const void* send, void* recv, int64_t count_per_rank, int dtype,
                      tt_stream_t stream);
int tt_comm_reduce_scatter(tt_comm_t comm, const void* send, void* recv, int64_t count_per_rank, int dtype, int op,
                           tt_stream_t stream);
int tt_comm_allreduce(tt_comm_t comm, const void* send, void* recv, int64_t count, int dtype, int op,
                      tt_stream_t stream);
int tt_comm_broadcast(tt_comm_t comm, void* buf, int64_t count, int dtype, int32_t root, tt_stream_t stream);

/* ---------------------------------------------------------------- K6 MIPS top-K
 * idx[b, 0:K], score[b, 0:K] = the K largest inner products q[b,:].corpus[c,:]
 * sorted by (score desc, index asc); replaces torch.topk(torch.matmul(q,
 * corpus.T), k) at ref:src/baseline_mips_module.py:57-61.  The [B,C] score
 * matrix is never written.  dtype: TT_F32 (corpus/query fp32) or TT_BF16
 * (both stored as bf16, fp32 accumulate -- BASELINE config 5). */
#define TT_F32 0
#define TT_BF16 1
#define TT_F16X2 2 /* EXPLORATORY: both operands as two-term fp16 splits of fp32 rows, see tt_mips_split_rows */
int64_t tt_mips_workspace_bytes(int64_t B, int64_t C, int64_t D, int64_t K, int dtype);
int tt_mips_topk(const void* query, const void* corpus, int dtype, int64_t B, int64_t C,
                 int64_t D, int64_t K, int64_t* idx_out, float* score_out, void* ws,
                 int64_t ws_bytes, tt_stream_t stream);
/* Exact merge of per-shard top-K lists (row-sharded corpus): per query, the K best of n_cand
 * (score, GLOBAL index) candidates under the same (score desc, index asc) order.  idx < 0
 * marks padding. */
int64_t tt_mips_merge_workspace_bytes(int64_t B, int64_t n_cand);
int tt_mips_merge(const float* scores, const int64_t* idx, int64_t B, int64_t n_cand, int64_t K,
                  int64_t* idx_out, float* score_out, void* ws, int64_t ws_bytes, tt_stream_t stream);
int tt_f32_to_bf16(const float* in, uint16_t* out, int64_t n, tt_stream_t stream);

/* EXPLORATORY: fp32-grade MIPS scores on the fp16 matrix pipe (dtype TT_F16X2; D = 128 only).  tt_mips_split_rows turns
 * fp32 rows into [rows][D x fp16 h | D x fp16 l] (x * scale = h + l, scale[0] = the power of two that brings the
 * matrix' largest magnitude to the top of fp16's range; 4 D bytes per row, the fp32 row's size; ws: 256 bytes) -- once
 * for a corpus, per call for the queries.  tt_mips_topk(query_split, corpus_split, TT_F16X2, ...) then scores every pair
 * as three fp16 MFMA products per 16-wide k-step (element-wise error at an fp32 fma chain's level, csrc/ce_f16x2.hip)
 * and returns scores times scale_q * scale_c; tt_mips_unscale divides that out (powers of two: exact, order untouched).
 * Same contract as the fp32 path otherwise -- (score desc, index asc), bit-exact indices and scores on exact-arithmetic
 * corpora; ref:src/baseline_mips_module.py:57-61. */
int tt_mips_split_rows(const float* X, int64_t rows, int64_t D, uint16_t* out, float* scale, void* ws, int64_t ws_bytes,
                       tt_stream_t stream);
int tt_mips_unscale(float* scores, int64_t n, const float* scale_a, const float* scale_b, tt_stream_t stream);
/* corpus[idx] for a bf16 corpus (ref:src/baseline_mips_module.py:63-69), rows widened exactly to fp32.  Ids outside
 * [0, n_rows) produce zero rows and set *oob_flag; oob_flag may be NULL (the owner side of a row-sharded corpus: the
 * padding slots of the fixed-size exchange carry the sentinel n_rows on purpose, like tt_gather_rows). */
int tt_gather_rows_bf16(const uint16_t* table, int64_t n_rows, int64_t dim, const int64_t* ids,
                        int64_t n_ids, float* out, int64_t ld_out, int32_t* oob_flag,
                        tt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TT_HOTPATH_H */
