// K4 (BASELINE shape): the attention backward of one SAMPLE per workgroup, for 4 heads x 32 and H <= 50
// (ref:src/user_history_encoder.py:103-108: the autograd of nn.MultiheadAttention's softmax(Q K^T / sqrt(dh)) V).
//
// attention_mfma.hip gives every (sample, head) its own wavefront in a barrier-free kernel, which (a) makes each wave
// fetch its operands from global memory five times per pair with nothing to hide the round trips behind -- 331 us per
// layer on an idle chip, 431 us next to the table sweep (B = 4096, H = 50) -- and (b) forces the products that reduce
// over the QUERIES (dK, dV) and the one that reduces over the KEYS (dQ) to each recompute S and dP in their own
// orientation: 448 MFMAs per pair.  Here a persistent workgroup -- four COMPUTE waves (wave = head) and four LOADER
// waves -- walks its samples:
//   * Q * scale, dO and K of all four heads sit in LDS ([56][36] images, rows H..55 zero, reads of rows >= 56 clamped
//     onto a zero row); V is needed as an MFMA B operand only and stays in registers.  The NEXT sample's rows are
//     requested right after the current one's have been written to LDS and ride in the LOADER waves' registers (28
//     float4 per thread) through the whole computation; the loaders also scale Q and form delta, VALU work that runs
//     beside the compute waves' MFMAs.  (With the staging in the compute waves' own registers the kernel is over its
//     register budget: 100 staging + 64 accumulator + operand registers per thread.)  Global latency is off the critical
//     path whatever the memory system is busy with.
//   * S and dP are computed once, in the "lane = key" orientation: P and dS are then A operands of dV = P^T dO and
//     dK = dS^T Q as they stand.  dQ = dS K reduces over the keys: dS goes through the wave's own [56][68] LDS scratch
//     (written as four 16-byte runs per lane, read back one element per lane) instead of being recomputed:
//     312 MFMAs per pair.
//   * delta_i = sum_j P_ij dP_ij equals dO_i . O_i (O = the forward's context rows), a row-wise dot product formed
//     while the rows are staged: no cross-lane reduction in the tile code.
// Three workgroup barriers per sample.  159 744 B of LDS, one workgroup per CU.
#include <stdlib.h>

#include "common.hpp"

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace awg {
constexpr int ROWS = 56, LD = 36, IMG = ROWS * LD;  // one head's [56][32 + 4] image
constexpr int SLD = 68, SCR = ROWS * SLD;           // one head's dS^T scratch: [key][query]
constexpr int HEADS = 4, DH = 32, D = 128;
__device__ __forceinline__ int arow(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }
}  // namespace awg

struct AttnWgArgs {
  const float* qkv;
  const float* ctx;
  const float* lse;
  const float* d_ctx;
  float* d_qkv;
  int64_t B;
  int H;
};

__global__ __launch_bounds__(512, 2) void attn_bwd_wg_kernel(const AttnWgArgs p) {
  using namespace awg;
  __shared__ __attribute__((aligned(16))) float Qs[HEADS * IMG];
  __shared__ __attribute__((aligned(16))) float Gs[HEADS * IMG];
  __shared__ __attribute__((aligned(16))) float Ks[HEADS * IMG];
  __shared__ __attribute__((aligned(16))) float Sc[HEADS * SCR];
  __shared__ __attribute__((aligned(16))) float Ls[HEADS * 64];
  __shared__ __attribute__((aligned(16))) float Ds[HEADS * 64];
  // waves 0..3 compute (wave = head, one per SIMD); waves 4..7 are LOADERS: they own the staging registers, fetch the
  // next sample's rows, and at the top of a sample write them into the LDS images and form delta.  Wave-uniform roles;
  // every wave meets the same two barriers per sample.  (Eight compute waves -- two per SIMD, head x key tile -- plus
  // four loaders measured 10 % slower: 168 registers per wave are not enough for the tile code.)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool loader = wave >= 4;
  const int tid = threadIdx.x & 255, lane = tid & 63, head = tid >> 6, r = lane & 31, h2 = lane >> 5;
  const int H = p.H, n4 = H * 32;
  const float scale = 0.17677669529663687f;  // 1 / sqrt(32)

  // images: rows H..55 are zero for the life of the kernel (staging writes rows < H only); statistics of the padding
  // query rows: lse = +huge -> P = 0
  for (int i = threadIdx.x; i < HEADS * IMG; i += 512) {
    const int row = (i % IMG) / LD;
    if (row >= H) { Qs[i] = 0.f; Gs[i] = 0.f; Ks[i] = 0.f; }
  }
  if (loader && lane >= H) { Ls[tid] = 3.0e38f; Ds[tid] = 0.f; }

  float* const Qh = Qs + head * IMG;
  float* const Gh = Gs + head * IMG;
  float* const Kh = Ks + head * IMG;
  float* const Sh = Sc + head * SCR;
  const float* const Lh = Ls + head * 64;
  const float* const Dh = Ds + head * 64;

  // Two loops, one per role, that meet the same barriers in the same order (prologue, then A and C per sample).  Not
  // one loop with role branches inside: the staging arrays are loop-carried, so in a shared loop they would be live
  // in the compute path as well -- and spilled there.
  if (loader) {
    // ---- staging registers.  A sample's [H][128] rows are 32 H <= 1600 float4 per matrix: six per thread (float4
    // #f = tid + 256 k, rows 0..47) plus a remainder of <= 64 (rows 48, 49).  The remainders of Q, K, dO and O share ONE
    // extra quad -- threads 0..63 carry Q's, 64..127 K's, 128..191 dO's, 192..255 O's -- instead of a seventh,
    // three-quarters empty quad per matrix: 100 staging registers instead of 112.
    constexpr int NS = 6;
    float4 sq[NS], sk[NS], sg[NS], so[NS], sx;
    float sl = 0.f;
    const int fx = NS * 256 + (tid & 63), mx = tid >> 6;  // the remainder element / whose it is
    // NOTE on the staging arrays: what looked like hipcc "spilling one of them" through several versions of this kernel
    // was an alloca it could not split into registers (ScratchSize > 0 with vgpr_spill_count = 0): a whole-struct copy
    // `*(float4*)dst = sk[k]` out of the array is a memcpy from the alloca (see the store of K below), as is a select
    // between two float4 lvalues.  A staging register in scratch waits for the very load it is meant to hide.  The
    // macro (instead of a lambda capturing the arrays by reference) and the unconditional fetch below date from that
    // hunt and are kept: both forms are known to stay in registers.
#define AWG_FETCH_ROWS(BB)                                                                          \
    do {                                                                                            \
      const char* __restrict__ qb = reinterpret_cast<const char*>(p.qkv + (BB) * H * (3 * D));      \
      const char* __restrict__ gb = reinterpret_cast<const char*>(p.d_ctx + (BB) * H * D);          \
      const char* __restrict__ ob = reinterpret_cast<const char*>(p.ctx + (BB) * H * D);            \
      /* 32-bit BYTE offsets: each load takes a scalar base + one offset register (not a 64-bit pair) */ \
      _Pragma("unroll") for (int k = 0; k < NS; ++k) {                                              \
        const int fi = tid + 256 * k;                                                               \
        const unsigned f = (unsigned)(fi < n4 ? fi : n4 - 1), row = f >> 5, c4 = f & 31;            \
        const unsigned o3 = (row * (3 * D) + 4 * c4) * 4u, o1 = (row * D + 4 * c4) * 4u;            \
        sq[k] = *reinterpret_cast<const float4*>(qb + o3);                                          \
        sk[k] = *reinterpret_cast<const float4*>(qb + o3 + D * 4u);                                 \
        sg[k] = *reinterpret_cast<const float4*>(gb + o1);                                          \
        so[k] = *reinterpret_cast<const float4*>(ob + o1);                                          \
      }                                                                                             \
      {                                                                                             \
        const unsigned f = (unsigned)(fx < n4 ? fx : n4 - 1), row = f >> 5, c4 = f & 31;            \
        const unsigned o3 = (row * (3 * D) + 4 * c4) * 4u, o1 = (row * D + 4 * c4) * 4u;            \
        const char* src = mx == 0 ? qb + o3 : mx == 1 ? qb + o3 + D * 4u : mx == 2 ? gb + o1 : ob + o1; \
        sx = *reinterpret_cast<const float4*>(src);                                                 \
      }                                                                                             \
      sl = p.lse[((BB) * HEADS + head) * H + (lane < H ? lane : 0)];                                \
    } while (0)
    auto dot8 = [](const float4 x, const float4 y, bool ok) {  // sum over the 8 lanes that hold one head's 32 columns of a row
      float part = ok ? x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w : 0.f;
      part += __shfl_xor(part, 1, 64);
      part += __shfl_xor(part, 2, 64);
      part += __shfl_xor(part, 4, 64);
      return part;
    };
    int64_t b = blockIdx.x;
    if (b < p.B) AWG_FETCH_ROWS(b);
    __syncthreads();  // the zero rows above
    for (; b < p.B; b += gridDim.x) {
      // the staged rows into the LDS images; delta_i = dO_i . O_i per head
      if (mx == 3) *reinterpret_cast<float4*>(Sc + 4 * (tid & 63)) = sx;  // O's remainder -> exchange space (the scratch is idle here)
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const int f = tid + 256 * k;
        const bool ok = f < n4;
        const int row = f >> 5, c4 = f & 31, hh = c4 >> 3, cc = c4 & 7;
        const float part = dot8(sg[k], so[k], ok);
        if (ok) {
          const int off = hh * IMG + row * LD + 4 * cc;
          *reinterpret_cast<float4*>(Qs + off) = make_float4(sq[k].x * scale, sq[k].y * scale, sq[k].z * scale, sq[k].w * scale);
          // (member by member: a whole-struct copy out of the array is a memcpy from an alloca, and the array then stays in scratch)
          *reinterpret_cast<float4*>(Ks + off) = make_float4(sk[k].x, sk[k].y, sk[k].z, sk[k].w);
          *reinterpret_cast<float4*>(Gs + off) = sg[k];
          if (cc == 0) Ds[hh * 64 + row] = part;
        }
      }
      if (lane < H) Ls[tid] = sl;
      {  // rows 48, 49
        const bool ok = fx < n4;
        const int row = fx >> 5, c4 = fx & 31, hh = c4 >> 3, cc = c4 & 7;
        const int off = hh * IMG + row * LD + 4 * cc;
        if (ok && mx == 0) *reinterpret_cast<float4*>(Qs + off) = make_float4(sx.x * scale, sx.y * scale, sx.z * scale, sx.w * scale);
        if (ok && mx == 1) *reinterpret_cast<float4*>(Ks + off) = sx;
        if (mx == 2) {  // wave 6 reads what wave 7 wrote above: same-workgroup LDS, ordered by the loaders' own barrier below
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
      __syncthreads();  // A0: the images (but for dO's remainder rows) and O's remainder are in LDS
      if (mx == 2) {
        const bool ok = fx < n4;
        const int row = fx >> 5, c4 = fx & 31, hh = c4 >> 3, cc = c4 & 7;
        const float4 o = *reinterpret_cast<const float4*>(Sc + 4 * (tid & 63));
        const float part = dot8(sx, o, ok);
        if (ok) {
          *reinterpret_cast<float4*>(Gs + hh * IMG + row * LD + 4 * cc) = sx;
          if (cc == 0) Ds[hh * 64 + row] = part;
        }
      }
      __syncthreads();  // A: images and statistics of sample b are in place
      // unconditional (past the end: the current sample again, unused): a conditional fetch makes every staging
      // register a merge of "old" and "new" across the loop's back edge (see the NOTE above)
      const int64_t bn = b + gridDim.x < p.B ? b + gridDim.x : b;
      AWG_FETCH_ROWS(bn);  // in flight while the compute waves work; consumed at the next "top"
      __syncthreads();  // C: every read of the images is done
    }
#undef AWG_FETCH_ROWS
    return;
  }

  // ---- compute waves.  V rows as B-operand fragments, both key tiles: lane (r, h2) holds V[32 t + r][8 g + 4 h2 + c]
  float4 vf[2][4];
  auto fetch_v = [&](int64_t bb) {
    const char* __restrict__ vb = reinterpret_cast<const char*>(p.qkv + bb * H * (3 * D));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = 32 * t + r;
      const bool ok = j < H;
      const unsigned vo = (unsigned)((ok ? j : 0) * (3 * D) + 2 * D + DH * head + 4 * h2) * 4u;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 x = *reinterpret_cast<const float4*>(vb + vo + 32u * g);
        vf[t][g] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  int64_t b = blockIdx.x;
  if (b < p.B) fetch_v(b);
  __syncthreads();  // the zero rows above
  for (; b < p.B; b += gridDim.x) {
    __syncthreads();  // A0 (the loaders' remainder exchange)
    __syncthreads();  // A: images and statistics of sample b are in place
    const int64_t bn = b + gridDim.x;
    float* __restrict__ ob = p.d_qkv + (b * H) * (3 * D);  // wave-uniform base of this sample's gradient rows
    // ---- phase X: lane = key j = 32 t + r
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int jrow = 32 * t + r;
      const bool jvalid = jrow < H;
      const int jr = jrow < ROWS ? jrow : ROWS - 1;  // rows >= 56: the zero row
      const float* const ka = Kh + jr * LD + 4 * h2;
      f32x16 dv, dk;
#pragma unroll
      for (int e = 0; e < 16; ++e) { dv[e] = 0.f; dk[e] = 0.f; }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int ir = (32 * it + r) < ROWS ? 32 * it + r : ROWS - 1;
        f32x16 s, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
        const float* qa = Qh + ir * LD + 4 * h2;
        const float* ga = Gh + ir * LD + 4 * h2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 a = *reinterpret_cast<const float4*>(qa + 8 * g);
          const float4 kq = *reinterpret_cast<const float4*>(ka + 8 * g);
          const float4 c = *reinterpret_cast<const float4*>(ga + 8 * g);
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, kq.x, s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, kq.y, s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, kq.z, s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, kq.w, s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.x, vf[t][g].x, dp, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.y, vf[t][g].y, dp, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.z, vf[t][g].z, dp, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.w, vf[t][g].w, dp, 0, 0, 0);
        }
        // registers e <-> query i = 32 it + arow(e, h2): P = exp(S - lse_i), dS = P (dP - delta_i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 l4 = *reinterpret_cast<const float4*>(Lh + 32 * it + 8 * q + 4 * h2);
          const float4 d4 = *reinterpret_cast<const float4*>(Dh + 32 * it + 8 * q + 4 * h2);
          const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int e = 4 * q + c;
            const float pe = jvalid ? __expf(s[e] - lv[c]) : 0.f;
            s[e] = pe;
            dp[e] = pe * (dp[e] - dl[c]);
          }
        }
        // dV[j][d] += sum_i P[i][j] dO[i][d];  dK[j][d] += sum_i dS[i][j] (Q scale)[i][d]
        const float* gy = Gh + (32 * it + 4 * h2) * LD + r;  // row 32 it + arow(e, h2) = 32 it + 4 h2 + (e & 3) + 8 (e >> 2)
        const float* qy = Qh + (32 * it + 4 * h2) * LD + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ro = (e & 3) + 8 * (e >> 2);          // compile-time row offset
          if (it == 1 && ro + 4 >= ROWS - 32) {           // rows 56..63 (either lane half): the zero row
            const int i = 32 + ro + 4 * h2;
            const int ic = i < ROWS ? i : ROWS - 1;
            dv = __builtin_amdgcn_mfma_f32_32x32x2f32(s[e], Gh[ic * LD + r], dv, 0, 0, 0);
            dk = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[e], Qh[ic * LD + r], dk, 0, 0, 0);
          } else {
            dv = __builtin_amdgcn_mfma_f32_32x32x2f32(s[e], gy[ro * LD], dv, 0, 0, 0);
            dk = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[e], qy[ro * LD], dk, 0, 0, 0);
          }
        }
        // dS^T: row = key, four runs of four consecutive queries (this wave's own scratch: no workgroup barrier)
        if (jrow < ROWS) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(Sh + jrow * SLD + 32 * it + 8 * q + 4 * h2) =
                make_float4(dp[4 * q], dp[4 * q + 1], dp[4 * q + 2], dp[4 * q + 3]);
        }
      }
      // lane = d, registers = keys 32 t + arow(e, h2)
      const unsigned oc = (unsigned)((32 * t + 4 * h2) * (3 * D) + DH * head + r);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ro = (e & 3) + 8 * (e >> 2);
        if (32 * t + 4 * h2 + ro < H) {
          ob[oc + (unsigned)(ro * 3 * D + 2 * D)] = dv[e];
          ob[oc + (unsigned)(ro * 3 * D + D)] = dk[e];
        }
      }
    }
    if (bn < p.B) fetch_v(bn);  // the V fragments are dead from here on
    __builtin_amdgcn_wave_barrier();  // this wave's scratch rows are written (a wave's LDS operations complete in order)

    // ---- phase Y: dQ[i][d] = scale * sum_j dS[i][j] K[j][d], both query tiles
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      f32x16 dq;
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[e] = 0.f;
      const float* sy = Sh + (4 * h2) * SLD + 32 * it + r;
      const float* ky = Kh + (4 * h2) * LD + r;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if (jt == 1 && (e >> 2) == 3) continue;  // keys 56..63: beyond every supported H
          const int ro = 32 * jt + (e & 3) + 8 * (e >> 2);
          dq = __builtin_amdgcn_mfma_f32_32x32x2f32(sy[ro * SLD], ky[ro * LD], dq, 0, 0, 0);
        }
      }
      const unsigned oc = (unsigned)((32 * it + 4 * h2) * (3 * D) + DH * head + r);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ro = (e & 3) + 8 * (e >> 2);
        if (32 * it + 4 * h2 + ro < H) ob[oc + (unsigned)(ro * 3 * D)] = dq[e] * scale;
      }
    }
    __syncthreads();  // C: every read of the images is done; the loaders may overwrite them
  }
}

bool attn_bwd_wg_supported(const void* qkv, const void* ctx, const void* d_ctx, const void* d_qkv, int64_t H, int64_t D,
                           int64_t heads) {
  const uintptr_t al = reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(ctx) | reinterpret_cast<uintptr_t>(d_ctx) |
                       reinterpret_cast<uintptr_t>(d_qkv);
  return (al & 15) == 0 && heads == awg::HEADS && D == awg::D && H >= 1 && H <= 50;  // 50: the loaders' staging split
}

int attn_bwd_wg(const float* qkv, const float* ctx, const float* lse, const float* d_ctx, int64_t B, int64_t H, float* d_qkv,
                hipStream_t st) {
  AttnWgArgs a{qkv, ctx, lse, d_ctx, d_qkv, B, (int)H};
  const unsigned grid = (unsigned)(B < 256 ? B : 256);
  ProfScope prof("attn_bwd_wg_kernel", st);
  attn_bwd_wg_kernel<<<grid, 512, 0, st>>>(a);
  return check_launch("attn_bwd_wg_kernel");
}

}  // namespace tt
