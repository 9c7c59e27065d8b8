// K4: unmasked multi-head softmax self-attention over one sample's H history rows
// (the core of nn.MultiheadAttention as used at ref:src/user_history_encoder.py:103-108).
// The projections around it (packed QKV in-projection, out-projection) are the MFMA
// GEMMs of gemm.hip; what remains per (sample, head) is tiny -- H x H x dh with H = 50,
// dh = 32 at the BASELINE shapes -- so one workgroup takes one (sample, head): K and V
// (and Q, dO in the backward) sit in LDS, every lane owns one query (or key) row in
// registers and walks the other side with wave-uniform LDS reads (hardware broadcast,
// conflict free).  Softmax uses the exact row max, like the reference.
//
// qkv is the packed projection [B*H, 3D] = [Q | K | V]; head h owns columns
// h*dh .. (h+1)*dh of each third.  The 1/sqrt(dh) scale is applied to q.
#include "common.hpp"

namespace tt {

template <int DHP>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, int H, int D, int heads, int dh,
                                                       float* __restrict__ ctx, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Ks = reinterpret_cast<float*>(smem_raw);  // [H][DHP]
  float* Vs = Ks + (size_t)H * DHP;                // [H][DHP]
  const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
  const float* base = qkv + (size_t)b * H * 3 * D + hd * dh;
  for (int idx = threadIdx.x; idx < H * DHP; idx += blockDim.x) {
    const int j = idx / DHP, d = idx % DHP;
    const bool in = d < dh;
    Ks[idx] = in ? base[(size_t)j * 3 * D + D + d] : 0.f;
    Vs[idx] = in ? base[(size_t)j * 3 * D + 2 * D + d] : 0.f;
  }
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)dh);
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    float q[DHP], o[DHP];
#pragma unroll
    for (int d = 0; d < DHP; ++d) {
      q[d] = (d < dh) ? base[(size_t)i * 3 * D + d] * scale : 0.f;
      o[d] = 0.f;
    }
    float mx = -3.0e38f;
    for (int j = 0; j < H; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DHP; ++d) s = fmaf(q[d], Ks[j * DHP + d], s);
      mx = fmaxf(mx, s);
    }
    float l = 0.f;
    for (int j = 0; j < H; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DHP; ++d) s = fmaf(q[d], Ks[j * DHP + d], s);
      const float p = __expf(s - mx);
      l += p;
#pragma unroll
      for (int d = 0; d < DHP; ++d) o[d] = fmaf(p, Vs[j * DHP + d], o[d]);
    }
    const float inv = 1.0f / l;
    float* out = ctx + ((size_t)b * H + i) * D + hd * dh;
#pragma unroll
    for (int d = 0; d < DHP; ++d)
      if (d < dh) out[d] = o[d] * inv;
    lse[((size_t)b * heads + hd) * H + i] = mx + __logf(l);
  }
}

template <int DHP>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                       const float* __restrict__ lse, const float* __restrict__ d_ctx,
                                                       int H, int D, int heads, int dh, float* __restrict__ d_qkv) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Qs = reinterpret_cast<float*>(smem_raw);  // [H][DHP]  q * scale
  float* Ks = Qs + (size_t)H * DHP;
  float* Vs = Ks + (size_t)H * DHP;
  float* Gs = Vs + (size_t)H * DHP;  // dO
  float* Ls = Gs + (size_t)H * DHP;  // [H] row lse
  float* Ds = Ls + H;                // [H] delta_i = dO_i . O_i
  const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
  const float* base = qkv + (size_t)b * H * 3 * D + hd * dh;
  const float* cbase = ctx + (size_t)b * H * D + hd * dh;
  const float* gbase = d_ctx + (size_t)b * H * D + hd * dh;
  const float scale = 1.0f / sqrtf((float)dh);
  for (int idx = threadIdx.x; idx < H * DHP; idx += blockDim.x) {
    const int j = idx / DHP, d = idx % DHP;
    const bool in = d < dh;
    Qs[idx] = in ? base[(size_t)j * 3 * D + d] * scale : 0.f;
    Ks[idx] = in ? base[(size_t)j * 3 * D + D + d] : 0.f;
    Vs[idx] = in ? base[(size_t)j * 3 * D + 2 * D + d] : 0.f;
    Gs[idx] = in ? gbase[(size_t)j * D + d] : 0.f;
  }
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    float dl = 0.f;
    for (int d = 0; d < dh; ++d) dl = fmaf(gbase[(size_t)i * D + d], cbase[(size_t)i * D + d], dl);
    Ds[i] = dl;
    Ls[i] = lse[((size_t)b * heads + hd) * H + i];
  }
  __syncthreads();
  float* obase = d_qkv + (size_t)b * H * 3 * D + hd * dh;
  // phase 1: lane = query row i  -> dq_i
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    float q[DHP], g[DHP], dq[DHP];
#pragma unroll
    for (int d = 0; d < DHP; ++d) { q[d] = Qs[i * DHP + d]; g[d] = Gs[i * DHP + d]; dq[d] = 0.f; }
    const float li = Ls[i], di = Ds[i];
    for (int j = 0; j < H; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < DHP; ++d) { s = fmaf(q[d], Ks[j * DHP + d], s); dp = fmaf(g[d], Vs[j * DHP + d], dp); }
      const float ds = __expf(s - li) * (dp - di);
#pragma unroll
      for (int d = 0; d < DHP; ++d) dq[d] = fmaf(ds, Ks[j * DHP + d], dq[d]);
    }
#pragma unroll
    for (int d = 0; d < DHP; ++d)
      if (d < dh) obase[(size_t)i * 3 * D + d] = dq[d] * scale;
  }
  // phase 2: lane = key row j  -> dk_j, dv_j
  for (int j = threadIdx.x; j < H; j += blockDim.x) {
    float k[DHP], v[DHP], dk[DHP], dv[DHP];
#pragma unroll
    for (int d = 0; d < DHP; ++d) { k[d] = Ks[j * DHP + d]; v[d] = Vs[j * DHP + d]; dk[d] = 0.f; dv[d] = 0.f; }
    for (int i = 0; i < H; ++i) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < DHP; ++d) { s = fmaf(Qs[i * DHP + d], k[d], s); dp = fmaf(Gs[i * DHP + d], v[d], dp); }
      const float p = __expf(s - Ls[i]);
      const float ds = p * (dp - Ds[i]);
#pragma unroll
      for (int d = 0; d < DHP; ++d) { dk[d] = fmaf(ds, Qs[i * DHP + d], dk[d]); dv[d] = fmaf(p, Gs[i * DHP + d], dv[d]); }
    }
#pragma unroll
    for (int d = 0; d < DHP; ++d)
      if (d < dh) {
        obase[(size_t)j * 3 * D + D + d] = dk[d];
        obase[(size_t)j * 3 * D + 2 * D + d] = dv[d];
      }
  }
}

// fast path, attention_mfma.hip
bool attn_mfma_supported(const void* a, const void* b, int64_t H, int64_t D, int64_t dh);
// one sample per 8-wave workgroup (attention_wg.hip): 4 heads x 32, H <= 56
bool attn_bwd_wg_supported(const void* qkv, const void* ctx, const void* d_ctx, const void* d_qkv, int64_t H, int64_t D,
                           int64_t heads);
int attn_bwd_wg(const float* qkv, const float* ctx, const float* lse, const float* d_ctx, int64_t B, int64_t H, float* d_qkv,
                hipStream_t st);
int attn_fwd_mfma(const float* qkv, int64_t B, int64_t H, int64_t D, int64_t heads, float* ctx, float* lse, hipStream_t st);
int attn_bwd_mfma(const float* qkv, const float* lse, const float* d_ctx, int64_t B, int64_t H, int64_t D,
                  int64_t heads, float* d_qkv, hipStream_t st);

// head widths above 64 (embedding_dim 512 with the history model's 4 heads) take the same kernels with 128 values per
// lane: they spill past the VGPR file, which makes them slow, not wrong.  Wider heads are refused.
static int pick_dhp(int64_t dh) { return dh <= 4 ? 4 : dh <= 16 ? 16 : dh <= 32 ? 32 : dh <= 64 ? 64 : dh <= 128 ? 128 : 0; }

template <typename K>
static int opt_in(K kernel, size_t lds, const char* name) {
  if (lds <= 64 * 1024) return 0;
  if (lds > 160 * 1024) { set_error("%s: H*dh too large for LDS (%zu bytes)", name, lds); return TT_E_UNSUPPORTED; }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e)); return (int)e; }
  return 0;
}


// ------------------------------------------------------------------ single-query attention
// The encoder consumes only ROW 0 of its last layer (ref:src/user_history_encoder.py:113), so
// that layer needs one query per sample: 1/H of the score / context work, no Q projection for the
// other rows, and a backward that touches K and V once.  One wavefront per (sample, head), H <= 64:
// lane j owns key j for the scores, lane d owns column d for the context.
//   fwd: s_j = scale * q . K_j ; p = softmax(s) ; ctx = sum_j p_j V_j        probs saved [B, heads, H]
//   bwd: dP_j = d_ctx . V_j ; delta = sum_j p_j dP_j ; dS_j = p_j (dP_j - delta)
//        dq = scale * sum_j dS_j K_j ; dK_j = scale * dS_j q ; dV_j = p_j d_ctx
// one head slice [H][dh] (K or V) of a (sample, head) -> the wave's LDS image [64][dh+1], coalesced;
// rows >= H are never read.  Only the matrix that is consumed "one row per lane" is staged; the
// one consumed "one column per lane" is read from global memory directly (already coalesced).
__device__ __forceinline__ void stage_slice(float* Xs, const float* __restrict__ Xb, int64_t ldkv, int H, int dh, int lane) {
  const int ld = dh + 1;
  for (int i = lane; i < H * dh; i += 64) Xs[(i / dh) * ld + (i % dh)] = Xb[(int64_t)(i / dh) * ldkv + (i % dh)];
}

__global__ __launch_bounds__(256) void attn_row0_fwd_kernel(const float* __restrict__ q0, int64_t ldq,
                                                            const float* __restrict__ kv, int64_t ldkv, int64_t n_pairs,
                                                            int H, int D, int heads, float* __restrict__ ctx0,
                                                            float* __restrict__ probs) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t pair = (int64_t)blockIdx.x * 4 + wave;
  if (pair >= n_pairs) return;
  const int dh = D / heads, ld = dh + 1;
  float* Ks = reinterpret_cast<float*>(smem_raw) + wave * (64 * ld + 128);
  float* qs = Ks + 64 * ld;  // [dh] scaled query
  float* ps = qs + 64;       // [64] probabilities
  const int64_t b = pair / heads, hd = pair % heads;
  const float scale = 1.0f / sqrtf((float)dh);
  const float* Kb = kv + b * H * ldkv + hd * dh;
  const float* Vb = Kb + D;
  stage_slice(Ks, Kb, ldkv, H, dh, lane);
  if (lane < dh) qs[lane] = q0[b * ldq + hd * dh + lane] * scale;
  __builtin_amdgcn_wave_barrier();
  float sc = -3.0e38f;
  if (lane < H) {
    float acc = 0.f;
    for (int d = 0; d < dh; ++d) acc = fmaf(qs[d], Ks[lane * ld + d], acc);
    sc = acc;
  }
  const float mx = wave_max(sc);
  const float e = (lane < H) ? __expf(sc - mx) : 0.f;
  const float l = wave_sum(e);
  const float pj = e / l;
  if (lane < H) probs[pair * H + lane] = pj;
  ps[lane] = pj;
  __builtin_amdgcn_wave_barrier();
  if (lane < dh) {  // column `lane` of V, rows in sequence: 128-B coalesced per row, loads independent
    float acc = 0.f;
#pragma unroll 8
    for (int j = 0; j < H; ++j) acc = fmaf(ps[j], Vb[(int64_t)j * ldkv + lane], acc);
    ctx0[b * D + hd * dh + lane] = acc;
  }
}

__global__ __launch_bounds__(256) void attn_row0_bwd_kernel(const float* __restrict__ q0, int64_t ldq,
                                                            const float* __restrict__ kv, int64_t ldkv,
                                                            const float* __restrict__ probs,
                                                            const float* __restrict__ d_ctx0, int64_t n_pairs, int H,
                                                            int D, int heads, float* __restrict__ d_q0,
                                                            float* __restrict__ d_kv, int64_t ld_dkv) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t pair = (int64_t)blockIdx.x * 4 + wave;
  if (pair >= n_pairs) return;
  const int dh = D / heads, ld = dh + 1;
  float* Vs = reinterpret_cast<float*>(smem_raw) + wave * (64 * ld + 256);
  float* qs = Vs + 64 * ld;  // [dh] query (unscaled)
  float* dcs = qs + 64;      // [dh] d_ctx0
  float* gs = dcs + 64;      // [64] scale * dS_j
  float* pss = gs + 64;      // [64] p_j
  const int64_t b = pair / heads, hd = pair % heads;
  const float scale = 1.0f / sqrtf((float)dh);
  const float* Kb = kv + b * H * ldkv + hd * dh;
  const float* Vb = Kb + D;
  stage_slice(Vs, Vb, ldkv, H, dh, lane);
  if (lane < dh) {
    qs[lane] = q0[b * ldq + hd * dh + lane];
    dcs[lane] = d_ctx0[b * D + hd * dh + lane];
  }
  __builtin_amdgcn_wave_barrier();
  float pj = 0.f, dp = 0.f;
  if (lane < H) {
    pj = probs[pair * H + lane];
    for (int d = 0; d < dh; ++d) dp = fmaf(dcs[d], Vs[lane * ld + d], dp);
  }
  const float delta = wave_sum(pj * dp);
  const float dS = pj * (dp - delta);
  gs[lane] = scale * dS;
  pss[lane] = pj;
  __builtin_amdgcn_wave_barrier();
  // dK_j = (scale dS_j) q, dV_j = p_j d_ctx0: consecutive lanes on consecutive columns
  float* dKb = d_kv + b * H * ld_dkv + hd * dh;
  for (int i = lane; i < H * dh; i += 64) {
    const int row = i / dh, d = i % dh;
    float* dst = dKb + (int64_t)row * ld_dkv + d;
    dst[0] = gs[row] * qs[d];
    dst[D] = pss[row] * dcs[d];
  }
  if (lane < dh) {  // dq = scale * sum_j dS_j K_j: column `lane` of K, rows in sequence
    float acc = 0.f;
#pragma unroll 8
    for (int j = 0; j < H; ++j) acc = fmaf(gs[j], Kb[(int64_t)j * ldkv + lane], acc);
    d_q0[b * D + hd * dh + lane] = acc;
  }
}

}  // namespace tt

using namespace tt;

extern "C" int tt_attn_fwd(const float* qkv, int64_t B, int64_t H, int64_t D, int64_t heads, float* ctx,
                           float* lse, tt_stream_t stream) {
  if (!qkv || !ctx || !lse) return fail_arg("tt_attn_fwd: null pointer");
  if (B < 0 || H <= 0 || D <= 0 || heads <= 0 || D % heads != 0) return fail_arg("tt_attn_fwd: sizes");
  if (B == 0) return 0;
  const int64_t dh = D / heads;
  if (attn_mfma_supported(qkv, ctx, H, D, dh)) return attn_fwd_mfma(qkv, B, H, D, heads, ctx, lse, S(stream));
  const int dhp = pick_dhp(dh);
  if (!dhp) { set_error("tt_attn_fwd: head dim %lld > 128 not implemented", (long long)dh); return TT_E_UNSUPPORTED; }
  const unsigned threads = (unsigned)(H >= 256 ? 256 : round_up(H, 64));
  const size_t lds = (size_t)2 * H * dhp * sizeof(float);
  const unsigned grid = (unsigned)(B * heads);
  int rc;
#define TT_FWD(P)                                                                     \
  if ((rc = opt_in(attn_fwd_kernel<P>, lds, "attn_fwd_kernel"))) return rc;           \
  attn_fwd_kernel<P><<<grid, threads, lds, S(stream)>>>(qkv, (int)H, (int)D, (int)heads, (int)dh, ctx, lse);
  if (dhp == 4) { TT_FWD(4) } else if (dhp == 16) { TT_FWD(16) } else if (dhp == 32) { TT_FWD(32) } else if (dhp == 64) { TT_FWD(64) } else { TT_FWD(128) }
#undef TT_FWD
  return check_launch("attn_fwd_kernel");
}

extern "C" int tt_attn_bwd(const float* qkv, const float* ctx, const float* lse, const float* d_ctx,
                           int64_t B, int64_t H, int64_t D, int64_t heads, float* d_qkv, tt_stream_t stream) {
  if (!qkv || !ctx || !lse || !d_ctx || !d_qkv) return fail_arg("tt_attn_bwd: null pointer");
  if (B < 0 || H <= 0 || D <= 0 || heads <= 0 || D % heads != 0) return fail_arg("tt_attn_bwd: sizes");
  if (B == 0) return 0;
  const int64_t dh = D / heads;
  if (attn_bwd_wg_supported(qkv, ctx, d_ctx, d_qkv, H, D, heads)) return attn_bwd_wg(qkv, ctx, lse, d_ctx, B, H, d_qkv, S(stream));
  if (attn_mfma_supported(qkv, d_ctx, H, D, dh) && (reinterpret_cast<uintptr_t>(d_qkv) & 15) == 0)
    return attn_bwd_mfma(qkv, lse, d_ctx, B, H, D, heads, d_qkv, S(stream));
  const int dhp = pick_dhp(dh);
  if (!dhp) { set_error("tt_attn_bwd: head dim %lld > 128 not implemented", (long long)dh); return TT_E_UNSUPPORTED; }
  const unsigned threads = (unsigned)(H >= 256 ? 256 : round_up(H, 64));
  const size_t lds = ((size_t)4 * H * dhp + 2 * H) * sizeof(float);
  const unsigned grid = (unsigned)(B * heads);
  int rc;
#define TT_BWD(P)                                                                     \
  if ((rc = opt_in(attn_bwd_kernel<P>, lds, "attn_bwd_kernel"))) return rc;           \
  attn_bwd_kernel<P><<<grid, threads, lds, S(stream)>>>(qkv, ctx, lse, d_ctx, (int)H, (int)D, (int)heads, (int)dh, d_qkv);
  if (dhp == 4) { TT_BWD(4) } else if (dhp == 16) { TT_BWD(16) } else if (dhp == 32) { TT_BWD(32) } else if (dhp == 64) { TT_BWD(64) } else { TT_BWD(128) }
#undef TT_BWD
  return check_launch("attn_bwd_kernel");
}

extern "C" int tt_attn_row0_fwd(const float* q0, int64_t ldq, const float* kv, int64_t ldkv, int64_t B, int64_t H,
                                int64_t D, int64_t heads, float* ctx0, float* probs, tt_stream_t stream) {
  if (!q0 || !kv || !ctx0 || !probs) return fail_arg("tt_attn_row0_fwd: null pointer");
  if (B < 0 || H <= 0 || H > 64 || D <= 0 || heads <= 0 || D % heads != 0 || ldq < D || ldkv < 2 * D)
    return fail_arg("tt_attn_row0_fwd: sizes (H <= 64)");
  if (B == 0) return 0;
  const int64_t n = B * heads, dh = D / heads;
  if (dh > 64) { set_error("tt_attn_row0_fwd: head dim %lld > 64 not implemented", (long long)dh); return TT_E_UNSUPPORTED; }
  const size_t lds = 4 * (64 * (size_t)(dh + 1) + 128) * sizeof(float);
  int rc = opt_in(attn_row0_fwd_kernel, lds, "attn_row0_fwd_kernel");
  if (rc) return rc;
  attn_row0_fwd_kernel<<<(unsigned)ceil_div(n, 4), 256, lds, S(stream)>>>(q0, ldq, kv, ldkv, n, (int)H, (int)D, (int)heads, ctx0, probs);
  return check_launch("attn_row0_fwd_kernel");
}

extern "C" int tt_attn_row0_bwd(const float* q0, int64_t ldq, const float* kv, int64_t ldkv, const float* probs,
                                const float* d_ctx0, int64_t B, int64_t H, int64_t D, int64_t heads, float* d_q0,
                                float* d_kv, int64_t ld_dkv, tt_stream_t stream) {
  if (!q0 || !kv || !probs || !d_ctx0 || !d_q0 || !d_kv) return fail_arg("tt_attn_row0_bwd: null pointer");
  if (B < 0 || H <= 0 || H > 64 || D <= 0 || heads <= 0 || D % heads != 0 || ldq < D || ldkv < 2 * D || ld_dkv < 2 * D)
    return fail_arg("tt_attn_row0_bwd: sizes (H <= 64)");
  if (B == 0) return 0;
  const int64_t n = B * heads, dh = D / heads;
  if (dh > 64) { set_error("tt_attn_row0_bwd: head dim %lld > 64 not implemented", (long long)dh); return TT_E_UNSUPPORTED; }
  const size_t lds = 4 * (64 * (size_t)(dh + 1) + 256) * sizeof(float);
  int rc = opt_in(attn_row0_bwd_kernel, lds, "attn_row0_bwd_kernel");
  if (rc) return rc;
  attn_row0_bwd_kernel<<<(unsigned)ceil_div(n, 4), 256, lds, S(stream)>>>(q0, ldq, kv, ldkv, probs, d_ctx0, n, (int)H, (int)D, (int)heads, d_q0, d_kv, ld_dkv);
  return check_launch("attn_row0_bwd_kernel");
}
