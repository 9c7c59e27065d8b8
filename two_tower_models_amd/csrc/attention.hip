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
