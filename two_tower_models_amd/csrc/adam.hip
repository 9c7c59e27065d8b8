// K2b: dense-exact Adam.  torch.optim.Adam on an nn.Embedding(sparse=False)
// updates EVERY row every step (ref:train/train.py:123-125,179): rows that were not
// looked up still decay their moments and move by their momentum.  The table step is
// therefore an HBM-bound streaming sweep -- 24 B per element (read+write p, m, v) --
// and that sweep is the roofline of the whole train step.
//
//   adam_touched_kernel   rows that WERE looked up: sum their gradient rows (run order
//                         from the plan => deterministic), full Adam update computed
//                         from the OLD p,m,v, result parked in a side buffer
//   adam_sweep_kernel     every row, gradient = 0, in place, pure float4 streaming
//   adam_writeback_kernel side buffer -> the looked-up rows (overwrites the sweep's
//                         zero-gradient result for them)
//
// Update rule, in torch's single-tensor operation order:
//   m += (g - m)*(1-b1);  v = v*b2 + (1-b2)*g*g;
//   p += -(lr/(1-b1^t)) * ( m / (sqrt(v)/sqrt(1-b2^t) + eps) )
#include <stdlib.h>

#include "common.hpp"

namespace tt {

constexpr int SWEEP_MAX_TABLES_DECL = 4;  // = SWEEP_MAX_TABLES (tables per multi-table launch)

constexpr int SWEEP_DEFAULT_PERSIST = 3;

struct AdamConst {
  float one_minus_b1, b2, one_minus_b2, eps, neg_step_size, inv_bc2_sqrt;
};

__device__ __forceinline__ AdamConst load_hyper(const double* __restrict__ h) {
  AdamConst c;
  c.one_minus_b1 = (float)(1.0 - h[1]);
  c.b2 = (float)h[2];
  c.one_minus_b2 = (float)(1.0 - h[2]);
  c.eps = (float)h[3];
  c.neg_step_size = (float)(-h[5]);
  c.inv_bc2_sqrt = (float)(1.0 / h[6]);
  return c;
}

// Every fused multiply-add is spelled out: left to -ffp-contract the compiler picks a different
// pairing in different kernels (fma(t, g, v*b2) in one, fma(v, b2, t*g) in another), and the
// schedules (serial / overlapped / deferred) must agree to the bit.
//
// m / (sqrt(v) / c + eps) is ONE function shared by every kernel and schedule (so they keep agreeing to the bit) and it is
// built from the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32 and the reciprocal of c prepared per step: 5 VALU issues, two
// of them quarter-rate, instead of an IEEE square root and two IEEE divisions (~30).  The dense sweep is HBM-bound and
// does not notice; the deferred schedule's replay (pure VALU work) is what gets cheaper.  Error: a few ulp of an update
// that is itself <= lr in size, i.e. orders of magnitude below the rounding of p += ... -- the trajectory tests against
// the reference fixture hold at their old tolerances.  v_sqrt_f32 takes a denormal v as 0: sqrt(v) < 1.1e-19 there,
// invisible next to any eps in use (torch's default is 1e-8); the floor keeps eps = 0 with v = 0 from producing
// 0 * inf (torch itself returns NaN there: 0 / 0).
__device__ __forceinline__ float adam_ratio(float m, float v, const AdamConst& c) {
  const float denom = fmaxf(fmaf(__builtin_amdgcn_sqrtf(v), c.inv_bc2_sqrt, c.eps), 1e-30f);
  return m * __builtin_amdgcn_rcpf(denom);
}
__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, float g, const AdamConst& c) {
  m = fmaf(c.one_minus_b1, g - m, m);
  v = fmaf(c.one_minus_b2 * g, g, v * c.b2);
  p = fmaf(c.neg_step_size, adam_ratio(m, v, c), p);
}
__device__ __forceinline__ void adam_elem_zero_grad(float& p, float& m, float& v, const AdamConst& c) {
  m = fmaf(c.one_minus_b1, -m, m);
  v = v * c.b2;
  p = fmaf(c.neg_step_size, adam_ratio(m, v, c), p);
}

// step-dependent constants exactly as load_hyper() hands them to the kernels
__device__ __forceinline__ void step_consts_from_doubles(double lr, double b1, double b2, double step,
                                                         double& h5, double& h6) {
  h5 = lr / (1.0 - pow(b1, step));
  h6 = sqrt(1.0 - pow(b2, step));
}
// tab (optional): per-step constants for the deferred schedule, tab[2j] = (float)(-h5_j),
// tab[2j+1] = (float)(1 / h6_j) -- the values every dense kernel of step j saw
__global__ void adam_advance_kernel(double* h, float* tab, int64_t tab_cap) {
  const double step = h[4] + 1.0;
  h[4] = step;
  step_consts_from_doubles(h[0], h[1], h[2], step, h[5], h[6]);
  const int64_t j = (int64_t)step;
  if (tab && j < tab_cap) {
    tab[2 * j] = (float)(-h[5]);
    tab[2 * j + 1] = (float)(1.0 / h[6]);
  }
}

__device__ __forceinline__ const float* source_row(const tt_grad_sources& s, int64_t pos) {
  int k = 0;
#pragma unroll
  for (int q = 1; q < TT_MAX_GRAD_SOURCES; ++q)
    if (q < s.n_sources && pos >= s.first[q]) k = q;
  return s.rows[k] + (pos - s.first[k]) * s.ld[k];
}

// Side buffer = three planes [cap][dim]: p | m | v, cap = n_ids.  A looked-up row lives in the slot
// numbered by its FIRST occurrence among the step's ids (perm[seg_begin[u]]: the sort is stable),
// so the buffer can be filled either from the plan (adam_stash_kernel: unique rows only) or
// straight from the id list before any plan exists (adam_stash_ids_kernel: every occurrence;
// the p plane is then also the forward's lookup result).
//
// one wavefront per unique row; 4 rows per workgroup.  FROM_SIDE: the row's OLD p,m,v were
// parked in the side buffer (overlapped schedule: by now the sweep may already have
// overwritten them in the table).
template <bool FROM_SIDE>
__global__ __launch_bounds__(256) void adam_touched_kernel(const float* __restrict__ W, const float* __restrict__ M,
                                                           const float* __restrict__ V, int64_t n_rows, int64_t dim,
                                                           const double* __restrict__ hyper, const tt_grad_sources src,
                                                           const int32_t* __restrict__ sorted_ids,
                                                           const int32_t* __restrict__ perm,
                                                           const int32_t* __restrict__ seg_begin,
                                                           const int32_t* __restrict__ n_unique,
                                                           float* __restrict__ side, int64_t cap) {
  const int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= *n_unique) return;
  const int lane = threadIdx.x & 63;
  const AdamConst c = load_hyper(hyper);
  const int32_t t0 = seg_begin[u], t1 = seg_begin[u + 1];
  const int64_t row = sorted_ids[t0];
  if (row >= n_rows) return;  // sentinel run: ids that belong to another rank's block
  float* sp = side + (int64_t)perm[t0] * dim;
  float* sm = sp + cap * dim;
  float* sv = sm + cap * dim;
  for (int64_t d = lane; d < dim; d += 64) {
    float g = 0.f;
    for (int32_t t = t0; t < t1; ++t) g += source_row(src, perm[t])[d];
    float p, m, v;
    if constexpr (FROM_SIDE) { p = sp[d]; m = sm[d]; v = sv[d]; }
    else { p = W[row * dim + d]; m = M[row * dim + d]; v = V[row * dim + d]; }
    adam_elem(p, m, v, g, c);
    sp[d] = p;
    sm[d] = m;
    sv[d] = v;
  }
}

__global__ __launch_bounds__(256) void adam_stash_kernel(const float* __restrict__ W, const float* __restrict__ M,
                                                         const float* __restrict__ V, int64_t n_rows, int64_t dim,
                                                         const int32_t* __restrict__ sorted_ids,
                                                         const int32_t* __restrict__ seg_begin,
                                                         const int32_t* __restrict__ perm,
                                                         const int32_t* __restrict__ n_unique,
                                                         float* __restrict__ side, int64_t cap) {
  const int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= *n_unique) return;
  const int lane = threadIdx.x & 63;
  const int32_t t0 = seg_begin[u];
  const int64_t row = sorted_ids[t0];
  if (row >= n_rows) return;
  float* sp = side + (int64_t)perm[t0] * dim;
  float* sm = sp + cap * dim;
  float* sv = sm + cap * dim;
  for (int64_t d = lane; d < dim; d += 64) {
    sp[d] = W[row * dim + d];
    sm[d] = M[row * dim + d];
    sv[d] = V[row * dim + d];
  }
}

// Plan-free stash: occurrence i of the step's id list parks row ids[i] in slot i (duplicates park
// the same values several times; the finish uses the first).  Needs nothing but the ids, so the
// sweep can start a few microseconds into the step while the sort runs underneath it.  Ids
// outside [0, n_rows) (another rank's rows, or invalid input that the plan will flag) park zeros.
// planes: bit 0 = p, bit 1 = m, bit 2 = v (tt_adam_begin_ids_planes: the p plane is all the forward's lookups need, the
// moments can be parked on the sweep's own stream in front of it)
__device__ __forceinline__ void adam_stash_ids_body(const float* __restrict__ W, const float* __restrict__ M,
                                                    const float* __restrict__ V, int64_t n_rows, int64_t dim,
                                                    const int64_t* __restrict__ ids, int64_t n_ids, float* __restrict__ side,
                                                    int64_t block, int planes = 7) {
  const int64_t i = block * 4 + (threadIdx.x >> 6);
  if (i >= n_ids) return;
  const int lane = threadIdx.x & 63;
  const int64_t row = ids[i];
  const bool ok = row >= 0 && row < n_rows;
  float* sp = side + i * dim;
  float* sm = sp + n_ids * dim;
  float* sv = sm + n_ids * dim;
  if (((dim & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(M) |
                            reinterpret_cast<uintptr_t>(V) | reinterpret_cast<uintptr_t>(side)) & 15) == 0) {
    // (the load is unconditional, from row 0 for an id that is not ours, and masked afterwards: `ok ? *ptr : zero` on a
    // float4 is a select between two ADDRESSES -- the zero lands in scratch memory, 32 bytes per lane of it)
    const int64_t src = (ok ? row : 0) * dim;
    for (int64_t d = 4 * lane; d < dim; d += 256) {
      if (planes & 1) {
        float4 v = *reinterpret_cast<const float4*>(W + src + d);
        if (!ok) v.x = v.y = v.z = v.w = 0.f;
        *reinterpret_cast<float4*>(sp + d) = v;
      }
      if (planes & 2) {
        float4 v = *reinterpret_cast<const float4*>(M + src + d);
        if (!ok) v.x = v.y = v.z = v.w = 0.f;
        *reinterpret_cast<float4*>(sm + d) = v;
      }
      if (planes & 4) {
        float4 v = *reinterpret_cast<const float4*>(V + src + d);
        if (!ok) v.x = v.y = v.z = v.w = 0.f;
        *reinterpret_cast<float4*>(sv + d) = v;
      }
    }

  } else {
    for (int64_t d = lane; d < dim; d += 64) {
      if (planes & 1) sp[d] = ok ? W[row * dim + d] : 0.f;
      if (planes & 2) sm[d] = ok ? M[row * dim + d] : 0.f;
      if (planes & 4) sv[d] = ok ? V[row * dim + d] : 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void adam_stash_ids_kernel(const float* __restrict__ W, const float* __restrict__ M,
                                                             const float* __restrict__ V, int64_t n_rows, int64_t dim,
                                                             const int64_t* __restrict__ ids, int64_t n_ids,
                                                             float* __restrict__ side) {
  adam_stash_ids_body(W, M, V, n_rows, dim, ids, n_ids, side, blockIdx.x);
}

// The top of an overlapped step in ONE launch (tt_adam_begin_ids): adam_advance_kernel's step-constant update (last
// workgroup; nothing below reads the constants) and the plan-free stash of every table (the workgroups before it, table
// after table).  Three tiny dependent launches cost three trips through the queue's inter-kernel dependency latency
// in front of the sweep -- and at the 1 M-row shapes the sweep's start is the step's critical path.
struct StashJobs {
  const float* W[SWEEP_MAX_TABLES_DECL];
  const float* M[SWEEP_MAX_TABLES_DECL];
  const float* V[SWEEP_MAX_TABLES_DECL];
  const int64_t* ids[SWEEP_MAX_TABLES_DECL];
  float* side[SWEEP_MAX_TABLES_DECL];
  int64_t n_rows[SWEEP_MAX_TABLES_DECL], dim[SWEEP_MAX_TABLES_DECL], n_ids[SWEEP_MAX_TABLES_DECL];
  unsigned first_block[SWEEP_MAX_TABLES_DECL + 1];
  int n;
};
__global__ __launch_bounds__(256) void adam_begin_ids_kernel(const StashJobs jobs, double* h, float* tab, int64_t tab_cap, int planes,
                                                             int advance) {
  const unsigned n_stash = jobs.first_block[jobs.n];
  if (blockIdx.x == n_stash) {
    if (threadIdx.x == 0 && advance) {
      const double step = h[4] + 1.0;
      h[4] = step;
      step_consts_from_doubles(h[0], h[1], h[2], step, h[5], h[6]);
      const int64_t j = (int64_t)step;
      if (tab && j < tab_cap) {
        tab[2 * j] = (float)(-h[5]);
        tab[2 * j + 1] = (float)(1.0 / h[6]);
      }
    }
    return;
  }
  int t = 0;
#pragma unroll
  for (int q = 1; q < SWEEP_MAX_TABLES_DECL; ++q)
    if (q < jobs.n && blockIdx.x >= jobs.first_block[q]) t = q;
  adam_stash_ids_body(jobs.W[t], jobs.M[t], jobs.V[t], jobs.n_rows[t], jobs.dim[t], jobs.ids[t], jobs.n_ids[t], jobs.side[t],
                      blockIdx.x - jobs.first_block[t], planes);
}

__global__ __launch_bounds__(256) void adam_writeback_kernel(float* __restrict__ W, float* __restrict__ M,
                                                             float* __restrict__ V, int64_t n_rows, int64_t dim,
                                                             const int32_t* __restrict__ sorted_ids,
                                                             const int32_t* __restrict__ seg_begin,
                                                             const int32_t* __restrict__ perm,
                                                             const int32_t* __restrict__ n_unique,
                                                             const float* __restrict__ side, int64_t cap) {
  const int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= *n_unique) return;
  const int lane = threadIdx.x & 63;
  const int32_t t0 = seg_begin[u];
  const int64_t row = sorted_ids[t0];
  if (row >= n_rows) return;
  const float* sp = side + (int64_t)perm[t0] * dim;
  const float* sm = sp + cap * dim;
  const float* sv = sm + cap * dim;
  for (int64_t d = lane; d < dim; d += 64) {
    W[row * dim + d] = sp[d];
    M[row * dim + d] = sm[d];
    V[row * dim + d] = sv[d];
  }
}

// Last phase of the overlapped schedule in ONE pass (dim % 4 == 0, 16-B aligned operands): the
// sweep has finished, so the looked-up rows go from the stash (old p,m,v) + their summed gradient
// straight into the table -- no second trip through the side buffer.  LPR lanes per unique row,
// one float4 per lane and plane; gradient runs are summed in plan order, as adam_touched_kernel does.
template <int LPR>
__device__ __forceinline__ void adam_finish_body(float* __restrict__ W, float* __restrict__ M, float* __restrict__ V,
                                                 int64_t n_rows, int64_t dim, const double* __restrict__ hyper,
                                                 const tt_grad_sources& src, const int32_t* __restrict__ sorted_ids,
                                                 const int32_t* __restrict__ perm, const int32_t* __restrict__ seg_begin,
                                                 const int32_t* __restrict__ n_unique, const float* __restrict__ side,
                                                 int64_t cap, int64_t block) {
  constexpr int ROWS = 256 / LPR;
  const int64_t u = block * ROWS + threadIdx.x / LPR;
  if (u >= *n_unique) return;
  const int sub = threadIdx.x % LPR;
  const AdamConst c = load_hyper(hyper);
  const int32_t t0 = seg_begin[u], t1 = seg_begin[u + 1];
  const int64_t row = sorted_ids[t0];
  if (row >= n_rows) return;  // sentinel run
  const int64_t n4 = dim / 4;
  float4* wp = reinterpret_cast<float4*>(W + row * dim);
  float4* wm = reinterpret_cast<float4*>(M + row * dim);
  float4* wv = reinterpret_cast<float4*>(V + row * dim);
  // side == nullptr: the rows were MARKED, not parked (tt_adam_mark_rows + tt_adam_tables_sweep_marked): the sweep left
  // them alone and their old p, m, v are where they always were
  const float4* sp = side ? reinterpret_cast<const float4*>(side + (int64_t)perm[t0] * dim) : wp;
  const float4* sm = side ? sp + cap * n4 : wm;
  const float4* sv = side ? sm + cap * n4 : wv;
  for (int64_t d = sub; d < n4; d += LPR) {
    float4 p = sp[d], m = sm[d], v = sv[d];
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int32_t t = t0; t < t1; ++t) {
      const float4 x = reinterpret_cast<const float4*>(source_row(src, perm[t]))[d];
      g.x += x.x; g.y += x.y; g.z += x.z; g.w += x.w;
    }
    adam_elem(p.x, m.x, v.x, g.x, c);
    adam_elem(p.y, m.y, v.y, g.y, c);
    adam_elem(p.z, m.z, v.z, g.z, c);
    adam_elem(p.w, m.w, v.w, g.w, c);
    wp[d] = p;
    wm[d] = m;
    wv[d] = v;
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void adam_finish_kernel(float* __restrict__ W, float* __restrict__ M, float* __restrict__ V,
                                                          int64_t n_rows, int64_t dim, const double* __restrict__ hyper,
                                                          const tt_grad_sources src, const int32_t* __restrict__ sorted_ids,
                                                          const int32_t* __restrict__ perm,
                                                          const int32_t* __restrict__ seg_begin,
                                                          const int32_t* __restrict__ n_unique,
                                                          const float* __restrict__ side, int64_t cap) {
  adam_finish_body<LPR>(W, M, V, n_rows, dim, hyper, src, sorted_ids, perm, seg_begin, n_unique, side, cap, blockIdx.x);
}

// The same for SEVERAL tables in one launch (tt_adam_tables_finish): after the sweep the step's tail is finish, finish,
// dense Adam -- each a few microseconds of work behind a dependent launch.
struct FinishJobs {
  float* W[SWEEP_MAX_TABLES_DECL];
  float* M[SWEEP_MAX_TABLES_DECL];
  float* V[SWEEP_MAX_TABLES_DECL];
  int64_t n_rows[SWEEP_MAX_TABLES_DECL], dim[SWEEP_MAX_TABLES_DECL], cap[SWEEP_MAX_TABLES_DECL];
  tt_grad_sources src[SWEEP_MAX_TABLES_DECL];
  const int32_t* sorted_ids[SWEEP_MAX_TABLES_DECL];
  const int32_t* perm[SWEEP_MAX_TABLES_DECL];
  const int32_t* seg_begin[SWEEP_MAX_TABLES_DECL];
  const int32_t* n_unique[SWEEP_MAX_TABLES_DECL];
  const float* side[SWEEP_MAX_TABLES_DECL];
  unsigned first_block[SWEEP_MAX_TABLES_DECL + 1];
  int n;
};
template <int LPR>
__global__ __launch_bounds__(256) void adam_finish_tables_kernel(const FinishJobs jobs, const double* __restrict__ hyper) {
  int t = 0;
#pragma unroll
  for (int q = 1; q < SWEEP_MAX_TABLES_DECL; ++q)
    if (q < jobs.n && blockIdx.x >= jobs.first_block[q]) t = q;
  adam_finish_body<LPR>(jobs.W[t], jobs.M[t], jobs.V[t], jobs.n_rows[t], jobs.dim[t], hyper, jobs.src[t], jobs.sorted_ids[t],
                        jobs.perm[t], jobs.seg_begin[t], jobs.n_unique[t], jobs.side[t], jobs.cap[t],
                        blockIdx.x - jobs.first_block[t]);
}

// The roofline kernel: 3 x 16-B loads + 3 x 16-B stores per lane per float4, nothing else.  Persistent: gridDim.x =
// (#CUs x workgroups-per-CU) workgroups take 256 x ITERS-float4 chunks in order, so neighbouring workgroups stream
// neighbouring chunks while the sweep's footprint on every CU stays FIXED: a few light waves per SIMD, no LDS.  That
// leaves registers, LDS and wave slots for the forward / backward kernels' big workgroups for the whole 5 ms the sweep
// lasts (a one-shot grid lets the dispatcher refill every freed slot with another small sweep workgroup, and a
// 256-VGPR / 64-KiB workgroup never finds a whole CU's worth of room: sweep + compute in series).
typedef float vf4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 sweep_load(const float4* p) {
  if constexpr (NT) {
    const vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
  } else {
    return *p;
  }
}
template <bool NT>
__device__ __forceinline__ void sweep_store(const float4& v, float4* p) {
  if constexpr (NT) {
    vf4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<vf4*>(p));
  } else {
    *p = v;
  }
}

// NT: non-temporal loads AND stores (every byte is touched exactly once per step, nothing is worth keeping in
// L2 / MALL).  Measured with tools/sweep_probe.hip, all variants in one process on the same buffers: nt on both
// sides +4 % at 3 workgroups per CU, +9.5 % at 2 per CU (6.16 vs 5.62 TB/s on a slower box, 6.33 vs 6.10 on a
// faster one); nt on the loads alone +1.5 %, on the stores alone -3 %.  (Round 1 tried nt on the one-shot grid
// form only, where it lost.)  FEWER resident sweep waves stream better, not worse: 1-2 workgroups per CU beat 3,
// 4, 6, 8 -- fewer DRAM pages open at once -- as long as each lane keeps 12 x 16 B in flight.
template <int ITERS, int UNR, bool NT>
__global__ __launch_bounds__(256) void adam_sweep_persistent_kernel(float4* __restrict__ W, float4* __restrict__ M,
                                                                    float4* __restrict__ V, int64_t n4,
                                                                    const double* __restrict__ hyper,
                                                                    unsigned* __restrict__ ctr, int prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);  // A/B (TT_SWEEP_PRIO): the sweep's waves win instruction issue on their SIMD
  const AdamConst c = load_hyper(hyper);
  const unsigned n_chunks = (unsigned)((n4 + 256 * ITERS - 1) / (256 * ITERS));
  __shared__ unsigned s_next[2];
  // Chunks are handed out through a device counter (ctr[0]), one fetch ahead of the chunk being
  // streamed, so workgroups that share a CU with a heavy kernel simply take fewer chunks (with a
  // static round-robin assignment the step time was bimodal).  The last workgroup out (ctr[1])
  // re-arms both counters for the next launch; launches sharing `ctr` must be stream-ordered,
  // which they are (one optimiser = one hyper buffer = one sweep stream).
  if (threadIdx.x == 0) s_next[0] = atomicAdd(&ctr[0], 1u);
  __syncthreads();
  unsigned ch = s_next[0];
  int par = 0;
  while (ch < n_chunks) {
    if (threadIdx.x == 0) s_next[par ^ 1] = atomicAdd(&ctr[0], 1u);
    const int64_t base = (int64_t)ch * (256 * ITERS) + threadIdx.x;
#pragma unroll UNR
    for (int k = 0; k < ITERS; ++k) {
      const int64_t i = base + (int64_t)k * 256;
      if (i >= n4) break;
      float4 p = sweep_load<NT>(W + i), m = sweep_load<NT>(M + i), v = sweep_load<NT>(V + i);
      adam_elem_zero_grad(p.x, m.x, v.x, c);
      adam_elem_zero_grad(p.y, m.y, v.y, c);
      adam_elem_zero_grad(p.z, m.z, v.z, c);
      adam_elem_zero_grad(p.w, m.w, v.w, c);
      sweep_store<NT>(p, W + i); sweep_store<NT>(m, M + i); sweep_store<NT>(v, V + i);
    }
    __syncthreads();
    par ^= 1;
    ch = s_next[par];
  }
  if (threadIdx.x == 0) {
    if (atomicAdd(&ctr[1], 1u) == gridDim.x - 1) {
      ctr[0] = 0;
      ctr[1] = 0;
      __threadfence();
    }
  }
}

// The same sweep over SEVERAL tables in one launch (tt_adam_tables_sweep): the chunk list simply spans them,
// so a step has one sweep launch, one tail and no launch gap between the user and the item table.
constexpr int SWEEP_MAX_TABLES = SWEEP_MAX_TABLES_DECL;
struct SweepTables {
  float4* W[SWEEP_MAX_TABLES];
  float4* M[SWEEP_MAX_TABLES];
  float4* V[SWEEP_MAX_TABLES];
  int64_t n4[SWEEP_MAX_TABLES];
  unsigned first_chunk[SWEEP_MAX_TABLES + 1];  // table t owns chunks [first_chunk[t], first_chunk[t + 1])
  int n;
};
template <int ITERS, int UNR, bool NT>
__global__ __launch_bounds__(256) void adam_sweep_tables_kernel(const SweepTables tabs, const double* __restrict__ hyper,
                                                                unsigned* __restrict__ ctr, int prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);
  const AdamConst c = load_hyper(hyper);
  const unsigned n_chunks = tabs.first_chunk[tabs.n];
  __shared__ unsigned s_next[2];
  if (threadIdx.x == 0) s_next[0] = atomicAdd(&ctr[0], 1u);
  __syncthreads();
  unsigned ch = s_next[0];
  int par = 0;
  while (ch < n_chunks) {
    if (threadIdx.x == 0) s_next[par ^ 1] = atomicAdd(&ctr[0], 1u);
    int t = 0;
#pragma unroll
    for (int q = 1; q < SWEEP_MAX_TABLES; ++q)
      if (q < tabs.n && ch >= tabs.first_chunk[q]) t = q;
    float4* __restrict__ W = tabs.W[t];
    float4* __restrict__ M = tabs.M[t];
    float4* __restrict__ V = tabs.V[t];
    const int64_t n4 = tabs.n4[t];
    const int64_t base = (int64_t)(ch - tabs.first_chunk[t]) * (256 * ITERS) + threadIdx.x;
#pragma unroll UNR
    for (int k = 0; k < ITERS; ++k) {
      const int64_t i = base + (int64_t)k * 256;
      if (i >= n4) break;
      float4 p = sweep_load<NT>(W + i), m = sweep_load<NT>(M + i), v = sweep_load<NT>(V + i);
      adam_elem_zero_grad(p.x, m.x, v.x, c);
      adam_elem_zero_grad(p.y, m.y, v.y, c);
      adam_elem_zero_grad(p.z, m.z, v.z, c);
      adam_elem_zero_grad(p.w, m.w, v.w, c);
      sweep_store<NT>(p, W + i); sweep_store<NT>(m, M + i); sweep_store<NT>(v, V + i);
    }
    __syncthreads();
    par ^= 1;
    ch = s_next[par];
  }
  if (threadIdx.x == 0) {
    if (atomicAdd(&ctr[1], 1u) == gridDim.x - 1) {
      ctr[0] = 0;
      ctr[1] = 0;
      __threadfence();
    }
  }
}

// The sweep for steps that look up MANY rows (history model: 213 K of 1 M item rows per step).  Parking such a step's rows
// (p, m, v out to a side buffer before the sweep, back in afterwards) moved 0.67 GB in front of the sweep and the sweep
// then spent another 0.58 GB on rows whose result the finish overwrites.  Here the looked-up rows are MARKED in a bitmap
// (one bit per row, tt_adam_mark_rows) and the sweep steps over them: they keep their old p, m, v until the finish
// (side == nullptr above) gives them their real update.  Same arithmetic for every row, so the same bits as the parked
// schedule.  A chunk (256 x ITERS float4) covers (1024 >> sh) whole rows, sh = log2(dim / 4) in [3, 10]: at most 128 rows
// = 4 words of the bitmap, fetched once per chunk (the chunk index is made wave-uniform first).  The unmarked form above stays
// what it is: the headline's sweep touches 16 K of 11 M rows and parks them (this kernel with no row marked in its place:
// P 5.436 -> 5.429 ms, C2 1.086 -> 1.082, i.e. nothing).
struct SweepTablesMarked {
  SweepTables t;
  const unsigned* marks[SWEEP_MAX_TABLES];
  int sh[SWEEP_MAX_TABLES];
};
template <int ITERS, int PAIR, bool NT>
__global__ __launch_bounds__(256) void adam_sweep_tables_marked_kernel(const SweepTablesMarked tm, const double* __restrict__ hyper,
                                                                       unsigned* __restrict__ ctr) {
  const AdamConst c = load_hyper(hyper);
  const unsigned n_chunks = tm.t.first_chunk[tm.t.n];
  __shared__ unsigned s_next[2];
  if (threadIdx.x == 0) s_next[0] = atomicAdd(&ctr[0], 1u);
  __syncthreads();
  unsigned ch = s_next[0];
  int par = 0;
  while (ch < n_chunks) {
    if (threadIdx.x == 0) s_next[par ^ 1] = atomicAdd(&ctr[0], 1u);
    const unsigned chu = __builtin_amdgcn_readfirstlane(ch);
    int t = 0;
#pragma unroll
    for (int q = 1; q < SWEEP_MAX_TABLES; ++q)
      if (q < tm.t.n && chu >= tm.t.first_chunk[q]) t = q;
    float4* __restrict__ W = tm.t.W[t];
    float4* __restrict__ M = tm.t.M[t];
    float4* __restrict__ V = tm.t.V[t];
    const int64_t n4 = tm.t.n4[t];
    const unsigned lc = chu - tm.t.first_chunk[t];
    const int sh = tm.sh[t];
    const int64_t r0 = (int64_t)lc * ((256 * ITERS) >> sh);  // first row of the chunk
    const unsigned* __restrict__ mk = tm.marks[t];
    unsigned w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    if (mk) {  // (the bitmap is padded by four words)
      const unsigned* mw = mk + (r0 >> 5);
      w0 = mw[0]; w1 = mw[1]; w2 = mw[2]; w3 = mw[3];
    }
    const int64_t base = (int64_t)lc * (256 * ITERS) + threadIdx.x;
    // Register budget: the kernels that run NEXT TO this sweep fill the register file -- the encoder's in-projection holds
    // 2 x 218 (allocated as 2 x 224) of a SIMD's 512 registers, so a CU takes it together with sweep waves of <= 64 registers per SIMD in total, and
    // the attention backward (2 x 250) with none.  All ITERS loads up front (92 registers) shut the in-projection out of every
    // CU with a sweep workgroup (651 us instead of 211); one row triple at a time (46) lets ONE sweep wave per SIMD in, so that
    // a sweep wide enough to finish in time (1.5 workgroups per CU) halved the in-projection's CUs.  PAIR triples in flight per
    // wave: half as many waves stream as much.
#pragma unroll
    for (int k = 0; k < ITERS; k += PAIR) {
      float4 p[PAIR], m[PAIR], v[PAIR];
      bool live[PAIR];
#pragma unroll
      for (int q = 0; q < PAIR; ++q) {
        const int64_t i = base + (int64_t)(k + q) * 256;
        const int64_t row = r0 + (((k + q) * 256 + (int)threadIdx.x) >> sh);
        const int wj = (int)((row >> 5) - (r0 >> 5));
        const unsigned word = wj == 0 ? w0 : wj == 1 ? w1 : wj == 2 ? w2 : w3;
        live[q] = i < n4 && !((word >> (row & 31)) & 1u);
      }
#pragma unroll
      for (int q = 0; q < PAIR; ++q) {
        const int64_t i = base + (int64_t)(k + q) * 256;
        if (live[q]) { p[q] = sweep_load<NT>(W + i); m[q] = sweep_load<NT>(M + i); v[q] = sweep_load<NT>(V + i); }
      }
#pragma unroll
      for (int q = 0; q < PAIR; ++q) {
        const int64_t i = base + (int64_t)(k + q) * 256;
        if (live[q]) {
          adam_elem_zero_grad(p[q].x, m[q].x, v[q].x, c);
          adam_elem_zero_grad(p[q].y, m[q].y, v[q].y, c);
          adam_elem_zero_grad(p[q].z, m[q].z, v[q].z, c);
          adam_elem_zero_grad(p[q].w, m[q].w, v[q].w, c);
          sweep_store<NT>(p[q], W + i); sweep_store<NT>(m[q], M + i); sweep_store<NT>(v[q], V + i);
        }
      }
    }
    __syncthreads();
    par ^= 1;
    ch = s_next[par];
  }
  if (threadIdx.x == 0) {
    if (atomicAdd(&ctr[1], 1u) == gridDim.x - 1) {
      ctr[0] = 0;
      ctr[1] = 0;
      __threadfence();
    }
  }
}

__global__ __launch_bounds__(256) void adam_mark_rows_kernel(const int64_t* __restrict__ ids, int64_t n_ids, int64_t n_rows,
                                                             unsigned* __restrict__ marks) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_ids) return;
  const int64_t row = ids[i];
  if (row >= 0 && row < n_rows) atomicOr(&marks[row >> 5], 1u << (row & 31));  // ids of another rank's rows / invalid ids: nothing
}

__global__ __launch_bounds__(256) void adam_sweep_scalar_kernel(float* __restrict__ W, float* __restrict__ M,
                                                                float* __restrict__ V, int64_t i0, int64_t n,
                                                                const double* __restrict__ hyper) {
  const AdamConst c = load_hyper(hyper);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = i0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float p = W[i], m = M[i], v = V[i];
    adam_elem_zero_grad(p, m, v, c);
    W[i] = p; M[i] = m; V[i] = v;
  }
}

// ------------------------------------------------------------------ deferred ("lazy") schedule
// Dense Adam moves an untouched row by a recurrence that needs nothing but the row itself and
// the step-dependent constants, so the zero-gradient steps of a row can be REPLAYED, in
// registers, the next time the row is needed (looked up, updated, or flushed), with the very
// same fp32 operations in the same order: bit-identical to sweeping the table every step, but
// the table is only touched where it is used.  last_step[row] = the step the row is current for.
__device__ __forceinline__ void replay_consts(AdamConst& c, const double* __restrict__ h, const float* __restrict__ tab,
                                              int64_t cap, int64_t j) {
  if (j < cap) {
    c.neg_step_size = tab[2 * j];
    c.inv_bc2_sqrt = tab[2 * j + 1];
  } else {  // beyond the table: the same double arithmetic adam_advance_kernel performs
    double h5, h6;
    step_consts_from_doubles(h[0], h[1], h[2], (double)j, h5, h6);
    c.neg_step_size = (float)(-h5);
    c.inv_bc2_sqrt = (float)(1.0 / h6);
  }
}

// one wavefront replays steps from+1 .. to of one row (zero gradient), Q elements per lane and
// chunk of 64*Q columns.  The replay is pure VALU work (adam_ratio: 7 VALU issues per element
// and step), so Q is matched to the row width -- no lane computes padding.
template <int Q>
__device__ __forceinline__ void replay_row_q(float* __restrict__ W, float* __restrict__ M, float* __restrict__ V,
                                             int64_t row, int64_t dim, int32_t from, int32_t to,
                                             const double* __restrict__ hyper, const float* __restrict__ tab,
                                             int64_t cap, int lane) {
  AdamConst c = load_hyper(hyper);
  for (int64_t d0 = 0; d0 < dim; d0 += 64 * Q) {
    float p[Q], m[Q], v[Q];
    bool live = false;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int64_t d = d0 + lane + 64 * q;
      const int64_t at = row * dim + (d < dim ? d : dim - 1);
      p[q] = W[at]; m[q] = M[at]; v[q] = V[at];
      live = live || (d < dim && (m[q] != 0.f || v[q] != 0.f));
    }
    // m = v = 0 (a row no gradient ever reached): the update is exactly p += -s * (0 / eps) = p
    if (!__any(live)) continue;
    for (int32_t j = from + 1; j <= to; ++j) {  // from / to are wave-uniform: scalar loop, scalar loads
      replay_consts(c, hyper, tab, cap, j);
#pragma unroll
      for (int q = 0; q < Q; ++q) adam_elem_zero_grad(p[q], m[q], v[q], c);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int64_t d = d0 + lane + 64 * q;
      if (d < dim) { W[row * dim + d] = p[q]; M[row * dim + d] = m[q]; V[row * dim + d] = v[q]; }
    }
  }
}
__device__ __forceinline__ void replay_row(float* __restrict__ W, float* __restrict__ M, float* __restrict__ V,
                                           int64_t row, int64_t dim, int32_t from, int32_t to,
                                           const double* __restrict__ hyper, const float* __restrict__ tab,
                                           int64_t cap, int lane) {
  from = __builtin_amdgcn_readfirstlane(from);
  to = __builtin_amdgcn_readfirstlane(to);
  if (dim <= 64) replay_row_q<1>(W, M, V, row, dim, from, to, hyper, tab, cap, lane);
  else if (dim <= 128) replay_row_q<2>(W, M, V, row, dim, from, to, hyper, tab, cap, lane);
  else replay_row_q<4>(W, M, V, row, dim, from, to, hyper, tab, cap, lane);
}

// rows about to be read or updated: bring ids[i] up to step (current + offset).  Duplicate ids are
// resolved by an atomic claim on last_step (the first wave replays, the others find it current).
__global__ __launch_bounds__(256) void adam_catchup_ids_kernel(float* __restrict__ W, float* __restrict__ M,
                                                               float* __restrict__ V, int64_t n_rows, int64_t dim,
                                                               const int64_t* __restrict__ ids, int64_t n_ids,
                                                               int32_t* __restrict__ last_step,
                                                               const double* __restrict__ hyper, int offset,
                                                               const float* __restrict__ tab, int64_t cap) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_ids) return;
  const int lane = threadIdx.x & 63;
  const int64_t row = ids[i];
  if (row < 0 || row >= n_rows) return;  // reported by the lookup / plan that follows
  const int32_t target = (int32_t)hyper[4] + offset;
  int32_t prev = 0;
  if (lane == 0) prev = atomicMax(&last_step[row], target);
  prev = __shfl(prev, 0, 64);
  if (prev >= target) return;
  replay_row(W, M, V, row, dim, prev, target, hyper, tab, cap, lane);
}

// the same for the unique rows of a plan (no duplicates: no claim needed)
__global__ __launch_bounds__(256) void adam_catchup_plan_kernel(float* __restrict__ W, float* __restrict__ M,
                                                                float* __restrict__ V, int64_t n_rows, int64_t dim,
                                                                const int32_t* __restrict__ sorted_ids,
                                                                const int32_t* __restrict__ seg_begin,
                                                                const int32_t* __restrict__ n_unique,
                                                                int32_t* __restrict__ last_step,
                                                                const double* __restrict__ hyper, int offset, int mark,
                                                                const float* __restrict__ tab, int64_t cap) {
  const int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= *n_unique) return;
  const int lane = threadIdx.x & 63;
  const int64_t row = sorted_ids[seg_begin[u]];
  if (row >= n_rows) return;
  const int32_t target = (int32_t)hyper[4] + offset;
  const int32_t prev = last_step[row];
  if (prev < target) replay_row(W, M, V, row, dim, prev, target, hyper, tab, cap, lane);
  if (lane == 0) last_step[row] = target + mark;  // mark = 1: the gradient step that follows
}

// every row up to the current step (before anything reads the table as a whole)
__global__ __launch_bounds__(256) void adam_flush_kernel(float* __restrict__ W, float* __restrict__ M,
                                                         float* __restrict__ V, int64_t n_rows, int64_t dim,
                                                         int32_t* __restrict__ last_step,
                                                         const double* __restrict__ hyper,
                                                         const float* __restrict__ tab, int64_t cap) {
  const int lane = threadIdx.x & 63;
  const int32_t target = (int32_t)hyper[4];
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += stride) {
    const int32_t prev = last_step[row];
    if (prev >= target) continue;
    replay_row(W, M, V, row, dim, prev, target, hyper, tab, cap, lane);
    if (lane == 0) last_step[row] = target;
  }
}

// dense parameters: blockIdx.y = tensor, blockIdx.x = 1024-element chunk.  The descriptors
// travel in the kernel arguments (by value), so there is no host->device copy to race with
// and a captured graph keeps its own copy.
constexpr int ADAM_BATCH = 64;
struct AdamBatch { tt_adam_tensor t[ADAM_BATCH]; };
__global__ __launch_bounds__(256) void adam_dense_kernel(const AdamBatch ts, const double* __restrict__ hyper) {
  const tt_adam_tensor t = ts.t[blockIdx.y];
  const AdamConst c = load_hyper(hyper);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < t.n; i += stride) {
    float p = t.p[i], m = t.m[i], v = t.v[i];
    adam_elem(p, m, v, t.g[i], c);
    t.p[i] = p; t.m[i] = m; t.v[i] = v;
  }
}

// row-sharded training: every replicated parameter's gradient into its slice of ONE flat buffer (the operand of the single
// dense-gradient all-reduce): t.p = destination slice, t.g = the gradient autograd produced, t.n elements
__global__ __launch_bounds__(256) void pack_grads_kernel(const AdamBatch ts) {
  const tt_adam_tensor t = ts.t[blockIdx.y];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < t.n; i += stride) t.p[i] = t.g[i];
}

// whole-step hipGraph: a new batch's seven input tensors into the captured static buffers in ONE launch
// (t.p = destination, t.g = source, t.n = BYTES; 16-byte words where both ends are aligned, bytes otherwise)
__global__ __launch_bounds__(256) void copy_buffers_kernel(const AdamBatch ts) {
  const tt_adam_tensor t = ts.t[blockIdx.y];
  char* dst = reinterpret_cast<char*>(t.p);
  const char* src = reinterpret_cast<const char*>(t.g);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool vec = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
  const int64_t n16 = vec ? t.n / 16 : 0;
  for (int64_t i = i0; i < n16; i += stride) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
  for (int64_t i = n16 * 16 + i0; i < t.n; i += stride) dst[i] = src[i];
}

// dense gradient for torch.optim users: dense[row,:] = sum of that row's gradient rows
__global__ __launch_bounds__(256) void rowgrad_dense_kernel(const tt_grad_sources src, int64_t n_rows, int64_t dim,
                                                            const int32_t* __restrict__ sorted_ids,
                                                            const int32_t* __restrict__ perm,
                                                            const int32_t* __restrict__ seg_begin,
                                                            const int32_t* __restrict__ n_unique,
                                                            float* __restrict__ dense) {
  const int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= *n_unique) return;
  const int lane = threadIdx.x & 63;
  const int32_t t0 = seg_begin[u], t1 = seg_begin[u + 1];
  const int64_t row = sorted_ids[t0];
  if (row >= n_rows) return;
  for (int64_t d = lane; d < dim; d += 64) {
    float g = 0.f;
    for (int32_t t = t0; t < t1; ++t) g += source_row(src, perm[t])[d];
    dense[row * dim + d] = g;
  }
}

static bool check_sources(const tt_grad_sources* s, int64_t n_ids, int64_t dim) {
  if (!s || s->n_sources < 1 || s->n_sources > TT_MAX_GRAD_SOURCES || s->first[0] != 0) return false;
  for (int k = 0; k < s->n_sources; ++k)
    if (!s->rows[k] || s->ld[k] < dim || s->first[k + 1] < s->first[k]) return false;
  return s->first[s->n_sources] == n_ids;
}

// CUs of the current device (256 on MI355X), queried once per device
static int device_cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cached[dev];
}

// n_wgs > 0: upper limit of persistent workgroups (the caller's sweep throttle; 0 = TT_SWEEP_PERSIST per CU)
static int launch_sweep(float* W, float* M, float* V, int64_t n_rows, int64_t dim, const double* hyper,
                        hipStream_t st, int n_wgs = 0) {
  int rc;
  const int64_t total = n_rows * dim;
  const bool vec = ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(M) | reinterpret_cast<uintptr_t>(V)) & 15) == 0;
  const int64_t n4 = vec ? total / 4 : 0;
  if (n4 > 0) {
    float4 *w4 = reinterpret_cast<float4*>(W), *m4 = reinterpret_cast<float4*>(M), *v4 = reinterpret_cast<float4*>(V);
    ProfScope prof("adam_sweep_kernel", st);
    // Persistent, dynamically chunked sweep with SWEEP_DEFAULT_PERSIST = 3 workgroups per CU (3 waves per SIMD, ~12
    // 16-byte loads in flight per lane: HBM-saturating on its own and small enough that a 256-VGPR / 64-KiB forward /
    // backward workgroup still fits next to it; from 5 per CU upwards it no longer does and the overlapped step degrades
    // to sweep + compute in series), non-temporal loads / stores.  The one-shot bounded grid, the temporal form and the
    // raised-priority form lost their A/Bs (profiles/HISTORY.md) and are gone.
    // hyper[7] is the library's scratch slot: two 32-bit chunk counters, zero between launches
    unsigned* ctr = reinterpret_cast<unsigned*>(const_cast<double*>(hyper) + 7);
    unsigned grid = (unsigned)(device_cu_count() * SWEEP_DEFAULT_PERSIST);
    if (n_wgs > 0 && (unsigned)n_wgs < grid) grid = (unsigned)n_wgs;
    adam_sweep_persistent_kernel<4, 4, true><<<grid, 256, 0, st>>>(w4, m4, v4, n4, hyper, ctr, 0);
    if ((rc = check_launch("adam_sweep_kernel"))) return rc;
  }
  if (n4 * 4 < total) {
    const int64_t rem = total - n4 * 4;
    const int64_t blocks = ceil_div(rem, 256) < 2048 ? ceil_div(rem, 256) : 2048;
    adam_sweep_scalar_kernel<<<(unsigned)blocks, 256, 0, st>>>(W, M, V, n4 * 4, total, hyper);
    if ((rc = check_launch("adam_sweep_scalar_kernel"))) return rc;
  }
  return 0;
}

}  // namespace tt

using namespace tt;

extern "C" int tt_adam_advance(double* hyper, tt_stream_t stream) {
  if (!hyper) return fail_arg("tt_adam_advance: null pointer");
  adam_advance_kernel<<<1, 1, 0, S(stream)>>>(hyper, nullptr, 0);
  return check_launch("adam_advance_kernel");
}

extern "C" int64_t tt_adam_table_workspace_bytes(int64_t n_ids, int64_t dim) {
  if (n_ids <= 0 || dim <= 0) return 256;
  return round_up(n_ids * 3 * dim * 4, 256);
}

extern "C" int tt_adam_table(float* W, float* M, float* V, int64_t n_rows, int64_t dim, const double* hyper,
                             const tt_grad_sources* src, int64_t n_ids, const int32_t* sorted_ids,
                             const int32_t* perm, const int32_t* seg_begin, const int32_t* n_unique,
                             void* ws, int64_t ws_bytes, tt_stream_t stream) {
  if (!W || !M || !V || !hyper) return fail_arg("tt_adam_table: null pointer");
  if (n_rows <= 0 || dim <= 0 || n_ids < 0) return fail_arg("tt_adam_table: sizes");
  hipStream_t st = S(stream);
  int rc;
  if (n_ids > 0) {
    if (!sorted_ids || !perm || !seg_begin || !n_unique || !ws) return fail_arg("tt_adam_table: null plan");
    if (!check_sources(src, n_ids, dim)) return fail_arg("tt_adam_table: gradient sources");
    if (ws_bytes < tt_adam_table_workspace_bytes(n_ids, dim)) { set_error("tt_adam_table: workspace"); return TT_E_WORKSPACE; }
    adam_touched_kernel<false><<<(unsigned)ceil_div(n_ids, 4), 256, 0, st>>>(W, M, V, n_rows, dim, hyper, *src, sorted_ids, perm, seg_begin, n_unique, reinterpret_cast<float*>(ws), n_ids);
    if ((rc = check_launch("adam_touched_kernel"))) return rc;
  }
  if ((rc = launch_sweep(W, M, V, n_rows, dim, hyper, st))) return rc;
  if (n_ids > 0) {
    adam_writeback_kernel<<<(unsigned)ceil_div(n_ids, 4), 256, 0, st>>>(W, M, V, n_rows, dim, sorted_ids, seg_begin, perm, n_unique, reinterpret_cast<const float*>(ws), n_ids);
    if ((rc = check_launch("adam_writeback_kernel"))) return rc;
  }
  return 0;
}

// ---- the same table step in three phases, so the sweep can run on its own stream while
// the backward pass is still producing the row gradients (DESIGN.md "overlap")
extern "C" int tt_adam_table_stash(const float* W, const float* M, const float* V, int64_t n_rows, int64_t dim,
                                   int64_t n_ids, const int32_t* sorted_ids, const int32_t* perm,
                                   const int32_t* seg_begin, const int32_t* n_unique, void* side,
                                   int64_t side_bytes, tt_stream_t stream) {
  if (!W || !M || !V || !sorted_ids || !perm || !seg_begin || !n_unique || !side) return fail_arg("tt_adam_table_stash: null pointer");
  if (n_rows <= 0 || dim <= 0 || n_ids <= 0) return fail_arg("tt_adam_table_stash: sizes");
  if (side_bytes < tt_adam_table_workspace_bytes(n_ids, dim)) { set_error("tt_adam_table_stash: side buffer"); return TT_E_WORKSPACE; }
  adam_stash_kernel<<<(unsigned)ceil_div(n_ids, 4), 256, 0, S(stream)>>>(W, M, V, n_rows, dim, sorted_ids, seg_begin, perm, n_unique, reinterpret_cast<float*>(side), n_ids);
  return check_launch("adam_stash_kernel");
}

extern "C" int tt_adam_table_stash_ids(const float* W, const float* M, const float* V, int64_t n_rows, int64_t dim,
                                       const int64_t* ids, int64_t n_ids, void* side, int64_t side_bytes,
                                       tt_stream_t stream) {
  if (!W || !M || !V || !ids || !side) return fail_arg("tt_adam_table_stash_ids: null pointer");
  if (n_rows <= 0 || dim <= 0 || n_ids <= 0) return fail_arg("tt_adam_table_stash_ids: sizes");
  if (side_bytes < tt_adam_table_workspace_bytes(n_ids, dim)) { set_error("tt_adam_table_stash_ids: side buffer"); return TT_E_WORKSPACE; }
  adam_stash_ids_kernel<<<(unsigned)ceil_div(n_ids, 4), 256, 0, S(stream)>>>(W, M, V, n_rows, dim, ids, n_ids, reinterpret_cast<float*>(side));
  return check_launch("adam_stash_ids_kernel");
}

static int begin_ids_impl(double* hyper, float* tab, int64_t tab_steps, const tt_adam_stash_job* jobs, int32_t n_jobs, int planes,
                          tt_stream_t stream);

extern "C" int tt_adam_begin_ids(double* hyper, float* tab, int64_t tab_steps, const tt_adam_stash_job* jobs, int32_t n_jobs,
                                 tt_stream_t stream) {
  return begin_ids_impl(hyper, tab, tab_steps, jobs, n_jobs, 7, stream);
}

extern "C" int tt_adam_begin_ids_planes(double* hyper, float* tab, int64_t tab_steps, const tt_adam_stash_job* jobs, int32_t n_jobs,
                                        int32_t planes, tt_stream_t stream) {
  if (planes <= 0 || planes > 7) return fail_arg("tt_adam_begin_ids_planes: planes is a mask of 1 (p), 2 (m), 4 (v)");
  if (n_jobs > SWEEP_MAX_TABLES) return fail_arg("tt_adam_begin_ids_planes: at most 4 tables");
  return begin_ids_impl(hyper, tab, tab_steps, jobs, n_jobs, planes, stream);
}

// planes & 1: the step-count advance rides along (it belongs to the launch that parks the p plane)
static int begin_ids_impl(double* hyper, float* tab, int64_t tab_steps, const tt_adam_stash_job* jobs, int32_t n_jobs, int planes,
                          tt_stream_t stream) {
  if (!hyper || (n_jobs > 0 && !jobs) || n_jobs < 0) return fail_arg("tt_adam_begin_ids: null pointer");
  if (tab && tab_steps <= 0) return fail_arg("tt_adam_begin_ids: sizes");
  hipStream_t st = S(stream);
  if (n_jobs > SWEEP_MAX_TABLES) {  // more tables than one launch takes: the separate entry points, same result
    int rc = tab ? tt_adam_advance_tab(hyper, tab, tab_steps, stream) : tt_adam_advance(hyper, stream);
    for (int i = 0; i < n_jobs && !rc; ++i)
      rc = tt_adam_table_stash_ids(jobs[i].W, jobs[i].M, jobs[i].V, jobs[i].n_rows, jobs[i].dim, jobs[i].ids, jobs[i].n_ids,
                                   jobs[i].side, jobs[i].side_bytes, stream);
    return rc;
  }
  StashJobs a{};
  a.n = n_jobs;
  unsigned blocks = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const tt_adam_stash_job& j = jobs[i];
    if (!j.W || !j.M || !j.V || !j.ids || !j.side) return fail_arg("tt_adam_begin_ids: null pointer");
    if (j.n_rows <= 0 || j.dim <= 0 || j.n_ids <= 0) return fail_arg("tt_adam_begin_ids: sizes");
    if (j.side_bytes < tt_adam_table_workspace_bytes(j.n_ids, j.dim)) { set_error("tt_adam_begin_ids: side buffer"); return TT_E_WORKSPACE; }
    a.W[i] = j.W; a.M[i] = j.M; a.V[i] = j.V; a.ids[i] = j.ids; a.side[i] = reinterpret_cast<float*>(j.side);
    a.n_rows[i] = j.n_rows; a.dim[i] = j.dim; a.n_ids[i] = j.n_ids;
    a.first_block[i] = blocks;
    blocks += (unsigned)ceil_div(j.n_ids, 4);
  }
  for (int i = n_jobs; i <= SWEEP_MAX_TABLES; ++i) a.first_block[i] = blocks;
  adam_begin_ids_kernel<<<blocks + 1, 256, 0, st>>>(a, hyper, tab, tab ? tab_steps : 0, planes, planes & 1);
  return check_launch("adam_begin_ids_kernel");
}

extern "C" int tt_adam_table_sweep(float* W, float* M, float* V, int64_t n_rows, int64_t dim, const double* hyper,
                                   tt_stream_t stream) {
  if (!W || !M || !V || !hyper) return fail_arg("tt_adam_table_sweep: null pointer");
  if (n_rows <= 0 || dim <= 0) return fail_arg("tt_adam_table_sweep: sizes");
  return launch_sweep(W, M, V, n_rows, dim, hyper, S(stream));
}

extern "C" int tt_adam_tables_sweep(const tt_adam_tensor* tables, int32_t n_tables, const double* hyper, int32_t n_wgs,
                                    tt_stream_t stream) {
  if (!tables || !hyper) return fail_arg("tt_adam_tables_sweep: null pointer");
  if (n_tables <= 0 || n_tables > SWEEP_MAX_TABLES) return fail_arg("tt_adam_tables_sweep: 1..4 tables");
  hipStream_t st = S(stream);
  bool fused = n_tables > 1;
  SweepTables tabs{};
  unsigned chunks = 0;
  for (int t = 0; t < n_tables; ++t) {
    const tt_adam_tensor& d = tables[t];
    if (!d.p || !d.m || !d.v || d.n <= 0) return fail_arg("tt_adam_tables_sweep: descriptor");
    // the fused launch streams float4 only: every table 16-B aligned with n % 4 == 0 (dim % 4 == 0), else launch per table
    if (((reinterpret_cast<uintptr_t>(d.p) | reinterpret_cast<uintptr_t>(d.m) | reinterpret_cast<uintptr_t>(d.v)) & 15) || d.n % 4) fused = false;
    tabs.W[t] = reinterpret_cast<float4*>(d.p); tabs.M[t] = reinterpret_cast<float4*>(d.m); tabs.V[t] = reinterpret_cast<float4*>(d.v);
    tabs.n4[t] = d.n / 4;
    tabs.first_chunk[t] = chunks;
    const int64_t c = ceil_div(d.n / 4, 256 * 4);
    if (c + chunks >= (1ll << 32)) fused = false;
    chunks += (unsigned)c;
  }
  tabs.first_chunk[n_tables] = chunks;
  tabs.n = n_tables;
  static const int wgs_env = getenv("TT_SWEEP_WGS") ? atoi(getenv("TT_SWEEP_WGS")) : 0;  // A/B: absolute workgroup count
  const int want = getenv("TT_SWEEP_WGS") ? wgs_env : n_wgs;  // the A/B switch wins over the caller's choice
  if (!fused) {  // one table (a rank that owns rows of one table only) or unaligned tables: per-table launches, same throttle
    for (int t = 0; t < n_tables; ++t) {
      const int rc = launch_sweep(tables[t].p, tables[t].m, tables[t].v, tables[t].n, 1, hyper, st, want);
      if (rc) return rc;
    }
    return 0;
  }
  unsigned* ctr = reinterpret_cast<unsigned*>(const_cast<double*>(hyper) + 7);
  unsigned grid = (unsigned)(device_cu_count() * SWEEP_DEFAULT_PERSIST);
  if (want > 0 && (unsigned)want < grid) grid = (unsigned)want;
  ProfScope prof("adam_sweep_kernel", st);
  adam_sweep_tables_kernel<4, 4, true><<<grid, 256, 0, st>>>(tabs, hyper, ctr, 0);
  return check_launch("adam_sweep_tables_kernel");
}

extern "C" int64_t tt_adam_marks_words(int64_t n_rows) { return n_rows > 0 ? (n_rows + 31) / 32 + 4 : 4; }

extern "C" int tt_adam_mark_rows(const int64_t* ids, int64_t n_ids, int64_t n_rows, uint32_t* marks, int64_t marks_words,
                                 tt_stream_t stream) {
  if (!ids || !marks) return fail_arg("tt_adam_mark_rows: null pointer");
  if (n_ids <= 0 || n_rows <= 0) return fail_arg("tt_adam_mark_rows: sizes");
  if (marks_words < tt_adam_marks_words(n_rows)) { set_error("tt_adam_mark_rows: bitmap of %lld words < %lld", (long long)marks_words, (long long)tt_adam_marks_words(n_rows)); return TT_E_WORKSPACE; }
  hipStream_t st = S(stream);
  // cleared HERE, every step, not by the finish: a step that is abandoned between its begin and its finish leaves nothing behind
  hipError_t e = hipMemsetAsync(marks, 0, (size_t)marks_words * 4, st);
  if (e != hipSuccess) { set_error("tt_adam_mark_rows: hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
  adam_mark_rows_kernel<<<(unsigned)ceil_div(n_ids, 256), 256, 0, st>>>(ids, n_ids, n_rows, marks);
  return check_launch("adam_mark_rows_kernel");
}

extern "C" int tt_adam_marked_supported(int64_t dim) {
  if (dim < 32 || dim > 4096 || (dim & (dim - 1))) return 0;  // whole rows per 16 KB chunk, at most 128 of them
  return 1;
}

extern "C" int tt_adam_tables_sweep_marked(const tt_adam_tensor* tables, const int64_t* dims, const uint32_t* const* marks,
                                           int32_t n_tables, const double* hyper, int32_t n_wgs, tt_stream_t stream) {
  if (!tables || !dims || !marks || !hyper) return fail_arg("tt_adam_tables_sweep_marked: null pointer");
  if (n_tables <= 0 || n_tables > SWEEP_MAX_TABLES) return fail_arg("tt_adam_tables_sweep_marked: 1..4 tables");
  SweepTablesMarked tm{};
  unsigned chunks = 0;
  for (int t = 0; t < n_tables; ++t) {
    const tt_adam_tensor& d = tables[t];
    if (!d.p || !d.m || !d.v || d.n <= 0) return fail_arg("tt_adam_tables_sweep_marked: descriptor");
    if (!tt_adam_marked_supported(dims[t]) || d.n % dims[t]) return fail_arg("tt_adam_tables_sweep_marked: dim must be a power of two in [32, 4096]");
    if ((reinterpret_cast<uintptr_t>(d.p) | reinterpret_cast<uintptr_t>(d.m) | reinterpret_cast<uintptr_t>(d.v)) & 15)
      return fail_arg("tt_adam_tables_sweep_marked: 16-byte aligned tables");
    tm.t.W[t] = reinterpret_cast<float4*>(d.p); tm.t.M[t] = reinterpret_cast<float4*>(d.m); tm.t.V[t] = reinterpret_cast<float4*>(d.v);
    tm.t.n4[t] = d.n / 4;
    tm.t.first_chunk[t] = chunks;
    const int64_t c = ceil_div(d.n / 4, 256 * 4);
    if (c + chunks >= (1ll << 32)) return fail_arg("tt_adam_tables_sweep_marked: too many chunks");
    chunks += (unsigned)c;
    tm.marks[t] = marks[t];  // NULL: no row of this table is marked
    int sh = 0;
    while ((4ll << sh) < dims[t]) ++sh;
    tm.sh[t] = sh;
  }
  tm.t.first_chunk[n_tables] = chunks;
  tm.t.n = n_tables;
  static const int wgs_env = getenv("TT_SWEEP_WGS") ? atoi(getenv("TT_SWEEP_WGS")) : 0;
  const int want = getenv("TT_SWEEP_WGS") ? wgs_env : n_wgs;
  unsigned* ctr = reinterpret_cast<unsigned*>(const_cast<double*>(hyper) + 7);
  unsigned grid = (unsigned)(device_cu_count() * SWEEP_DEFAULT_PERSIST);
  if (want > 0 && (unsigned)want < grid) grid = (unsigned)want;
  hipStream_t st = S(stream);
  ProfScope prof("adam_sweep_kernel", st);
  // two row triples in flight per wave: 62 registers (see the kernel); one at a time (46) 3.27 ms per C3 step at its best
  // width (384 workgroups), two 3.09 (256), four (92 registers) 3.34 (128) -- one process each, profiles/r06_marked_sweep_AB.txt
  adam_sweep_tables_marked_kernel<4, 2, true><<<grid, 256, 0, st>>>(tm, hyper, ctr);
  return check_launch("adam_sweep_tables_marked_kernel");
}

// A HIP stream of the device's LEAST priority for the sweep: the backward kernels on the
// caller's (normal-priority) stream win the dispatcher whenever both have work.
extern "C" int tt_stream_create_low_priority(void** out) {
  if (!out) return fail_arg("tt_stream_create_low_priority: null pointer");
  int least = 0, greatest = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (e != hipSuccess) { set_error("hipDeviceGetStreamPriorityRange: %s", hipGetErrorString(e)); return (int)e; }
  hipStream_t s;
  e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least);
  if (e != hipSuccess) { set_error("hipStreamCreateWithPriority: %s", hipGetErrorString(e)); return (int)e; }
  *out = reinterpret_cast<void*>(s);
  return 0;
}
extern "C" int tt_stream_destroy(void* stream) {
  if (!stream) return 0;
  hipError_t e = hipStreamDestroy(reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) { set_error("hipStreamDestroy: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

extern "C" int tt_adam_table_finish(float* W, float* M, float* V, int64_t n_rows, int64_t dim, const double* hyper,
                                    const tt_grad_sources* src, int64_t n_ids, const int32_t* sorted_ids,
                                    const int32_t* perm, const int32_t* seg_begin, const int32_t* n_unique,
                                    void* side, int64_t side_bytes, tt_stream_t stream) {
  if (!W || !M || !V || !hyper || !sorted_ids || !perm || !seg_begin || !n_unique)
    return fail_arg("tt_adam_table_finish: null pointer");
  if (n_rows <= 0 || dim <= 0 || n_ids <= 0) return fail_arg("tt_adam_table_finish: sizes");
  if (!check_sources(src, n_ids, dim)) return fail_arg("tt_adam_table_finish: gradient sources");
  // side == NULL: the rows were marked, not parked (tt_adam_tables_sweep_marked); vector form only
  if (side && side_bytes < tt_adam_table_workspace_bytes(n_ids, dim)) { set_error("tt_adam_table_finish: side buffer"); return TT_E_WORKSPACE; }
  hipStream_t st = S(stream);
  float* sd = reinterpret_cast<float*>(side);
  bool vec = dim % 4 == 0 && ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(M) | reinterpret_cast<uintptr_t>(V) |
                               reinterpret_cast<uintptr_t>(side)) & 15) == 0;
  for (int k = 0; vec && k < src->n_sources; ++k)
    vec = (reinterpret_cast<uintptr_t>(src->rows[k]) & 15) == 0 && src->ld[k] % 4 == 0;
  if (vec) {
    if (dim <= 64) adam_finish_kernel<16><<<(unsigned)ceil_div(n_ids, 16), 256, 0, st>>>(W, M, V, n_rows, dim, hyper, *src, sorted_ids, perm, seg_begin, n_unique, sd, n_ids);
    else if (dim <= 128) adam_finish_kernel<32><<<(unsigned)ceil_div(n_ids, 8), 256, 0, st>>>(W, M, V, n_rows, dim, hyper, *src, sorted_ids, perm, seg_begin, n_unique, sd, n_ids);
    else adam_finish_kernel<64><<<(unsigned)ceil_div(n_ids, 4), 256, 0, st>>>(W, M, V, n_rows, dim, hyper, *src, sorted_ids, perm, seg_begin, n_unique, sd, n_ids);
    return check_launch("adam_finish_kernel");
  }
  if (!side) return fail_arg("tt_adam_table_finish: marked rows (side == NULL) need dim % 4 == 0 and 16-byte aligned operands");
  adam_touched_kernel<true><<<(unsigned)ceil_div(n_ids, 4), 256, 0, st>>>(W, M, V, n_rows, dim, hyper, *src, sorted_ids, perm, seg_begin, n_unique, sd, n_ids);
  int rc = check_launch("adam_touched_kernel");
  if (rc) return rc;
  adam_writeback_kernel<<<(unsigned)ceil_div(n_ids, 4), 256, 0, st>>>(W, M, V, n_rows, dim, sorted_ids, seg_begin, perm, n_unique, sd, n_ids);
  return check_launch("adam_writeback_kernel");
}

extern "C" int tt_adam_tables_finish(const tt_adam_finish_job* jobs, int32_t n_jobs, const double* hyper, tt_stream_t stream) {
  if (!jobs || !hyper || n_jobs <= 0) return fail_arg("tt_adam_tables_finish: null pointer");
  // one launch when every table takes the vector form with the same lanes-per-row class; else table by table
  bool one = n_jobs <= SWEEP_MAX_TABLES;
  int lpr = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const tt_adam_finish_job& j = jobs[i];
    if (!j.W || !j.M || !j.V || !j.src || !j.sorted_ids || !j.perm || !j.seg_begin || !j.n_unique)
      return fail_arg("tt_adam_tables_finish: null pointer");
    if (j.n_rows <= 0 || j.dim <= 0 || j.n_ids <= 0) return fail_arg("tt_adam_tables_finish: sizes");
    if (!check_sources(j.src, j.n_ids, j.dim)) return fail_arg("tt_adam_tables_finish: gradient sources");
    if (j.side && j.side_bytes < tt_adam_table_workspace_bytes(j.n_ids, j.dim)) { set_error("tt_adam_tables_finish: side buffer"); return TT_E_WORKSPACE; }
    bool vec = j.dim % 4 == 0 && ((reinterpret_cast<uintptr_t>(j.W) | reinterpret_cast<uintptr_t>(j.M) | reinterpret_cast<uintptr_t>(j.V) |
                                   reinterpret_cast<uintptr_t>(j.side)) & 15) == 0;
    for (int k = 0; vec && k < j.src->n_sources; ++k)
      vec = (reinterpret_cast<uintptr_t>(j.src->rows[k]) & 15) == 0 && j.src->ld[k] % 4 == 0;
    const int cls = j.dim <= 64 ? 16 : j.dim <= 128 ? 32 : 64;
    if (!vec || (lpr && cls != lpr)) one = false;
    lpr = lpr ? lpr : cls;
  }
  if (!one) {
    for (int i = 0; i < n_jobs; ++i) {
      const tt_adam_finish_job& j = jobs[i];
      int rc = tt_adam_table_finish(j.W, j.M, j.V, j.n_rows, j.dim, hyper, j.src, j.n_ids, j.sorted_ids, j.perm, j.seg_begin,
                                    j.n_unique, j.side, j.side_bytes, stream);
      if (rc) return rc;
    }
    return 0;
  }
  FinishJobs a{};
  a.n = n_jobs;
  unsigned blocks = 0;
  const int rows_per_block = 256 / lpr;
  for (int i = 0; i < n_jobs; ++i) {
    const tt_adam_finish_job& j = jobs[i];
    a.W[i] = j.W; a.M[i] = j.M; a.V[i] = j.V; a.n_rows[i] = j.n_rows; a.dim[i] = j.dim; a.cap[i] = j.n_ids;
    a.src[i] = *j.src; a.sorted_ids[i] = j.sorted_ids; a.perm[i] = j.perm; a.seg_begin[i] = j.seg_begin; a.n_unique[i] = j.n_unique;
    a.side[i] = reinterpret_cast<const float*>(j.side);
    a.first_block[i] = blocks;
    blocks += (unsigned)ceil_div(j.n_ids, rows_per_block);
  }
  for (int i = n_jobs; i <= SWEEP_MAX_TABLES; ++i) a.first_block[i] = blocks;
  hipStream_t st = S(stream);
  if (lpr == 16) adam_finish_tables_kernel<16><<<blocks, 256, 0, st>>>(a, hyper);
  else if (lpr == 32) adam_finish_tables_kernel<32><<<blocks, 256, 0, st>>>(a, hyper);
  else adam_finish_tables_kernel<64><<<blocks, 256, 0, st>>>(a, hyper);
  return check_launch("adam_finish_tables_kernel");
}

// ---- deferred ("lazy") schedule: see the kernels above
extern "C" int tt_adam_advance_tab(double* hyper, float* tab, int64_t tab_steps, tt_stream_t stream) {
  if (!hyper || !tab) return fail_arg("tt_adam_advance_tab: null pointer");
  if (tab_steps <= 0) return fail_arg("tt_adam_advance_tab: sizes");
  adam_advance_kernel<<<1, 1, 0, S(stream)>>>(hyper, tab, tab_steps);
  return check_launch("adam_advance_kernel");
}

extern "C" int tt_adam_rows_catchup(float* W, float* M, float* V, int64_t n_rows, int64_t dim, const int64_t* ids,
                                    int64_t n_ids, int32_t* last_step, const double* hyper, const float* tab,
                                    int64_t tab_steps, tt_stream_t stream) {
  if (!W || !M || !V || !ids || !last_step || !hyper || !tab) return fail_arg("tt_adam_rows_catchup: null pointer");
  if (n_rows <= 0 || dim <= 0 || n_ids <= 0) return fail_arg("tt_adam_rows_catchup: sizes");
  adam_catchup_ids_kernel<<<(unsigned)ceil_div(n_ids, 4), 256, 0, S(stream)>>>(W, M, V, n_rows, dim, ids, n_ids, last_step, hyper, 0, tab, tab_steps);
  return check_launch("adam_catchup_ids_kernel");
}

extern "C" int tt_adam_table_lazy(float* W, float* M, float* V, int64_t n_rows, int64_t dim, const double* hyper,
                                  const tt_grad_sources* src, int64_t n_ids, const int32_t* sorted_ids,
                                  const int32_t* perm, const int32_t* seg_begin, const int32_t* n_unique,
                                  void* ws, int64_t ws_bytes, int32_t* last_step, const float* tab,
                                  int64_t tab_steps, tt_stream_t stream) {
  if (!W || !M || !V || !hyper || !sorted_ids || !perm || !seg_begin || !n_unique || !ws || !last_step || !tab)
    return fail_arg("tt_adam_table_lazy: null pointer");
  if (n_rows <= 0 || dim <= 0 || n_ids <= 0) return fail_arg("tt_adam_table_lazy: sizes");
  if (!check_sources(src, n_ids, dim)) return fail_arg("tt_adam_table_lazy: gradient sources");
  if (ws_bytes < tt_adam_table_workspace_bytes(n_ids, dim)) { set_error("tt_adam_table_lazy: workspace"); return TT_E_WORKSPACE; }
  hipStream_t st = S(stream);
  const unsigned blocks = (unsigned)ceil_div(n_ids, 4);
  // rows to the step before this one (a no-op for rows the forward's lookups already brought
  // up to date), stamped with this step; then this step's gradient update on them alone
  adam_catchup_plan_kernel<<<blocks, 256, 0, st>>>(W, M, V, n_rows, dim, sorted_ids, seg_begin, n_unique, last_step, hyper, -1, 1, tab, tab_steps);
  int rc = check_launch("adam_catchup_plan_kernel");
  if (rc) return rc;
  adam_touched_kernel<false><<<blocks, 256, 0, st>>>(W, M, V, n_rows, dim, hyper, *src, sorted_ids, perm, seg_begin, n_unique, reinterpret_cast<float*>(ws), n_ids);
  if ((rc = check_launch("adam_touched_kernel"))) return rc;
  adam_writeback_kernel<<<blocks, 256, 0, st>>>(W, M, V, n_rows, dim, sorted_ids, seg_begin, perm, n_unique, reinterpret_cast<const float*>(ws), n_ids);
  return check_launch("adam_writeback_kernel");
}

extern "C" int tt_adam_table_flush(float* W, float* M, float* V, int64_t n_rows, int64_t dim, int32_t* last_step,
                                   const double* hyper, const float* tab, int64_t tab_steps, tt_stream_t stream) {
  if (!W || !M || !V || !last_step || !hyper || !tab) return fail_arg("tt_adam_table_flush: null pointer");
  if (n_rows <= 0 || dim <= 0) return fail_arg("tt_adam_table_flush: sizes");
  const int64_t want = ceil_div(n_rows, 4);
  const int64_t cap = (int64_t)device_cu_count() * 32;
  ProfScope prof("adam_flush_kernel", S(stream));
  adam_flush_kernel<<<(unsigned)(want < cap ? want : cap), 256, 0, S(stream)>>>(W, M, V, n_rows, dim, last_step, hyper, tab, tab_steps);
  return check_launch("adam_flush_kernel");
}

extern "C" int tt_adam_dense(const tt_adam_tensor* tensors, int32_t n_tensors, const double* hyper,
                             tt_stream_t stream) {
  if (!tensors || !hyper) return fail_arg("tt_adam_dense: null pointer");
  if (n_tensors <= 0) return fail_arg("tt_adam_dense: sizes");
  for (int32_t base = 0; base < n_tensors; base += ADAM_BATCH) {
    const int32_t cnt = (n_tensors - base < ADAM_BATCH) ? n_tensors - base : ADAM_BATCH;
    AdamBatch b;
    int64_t max_n = 1;
    for (int32_t i = 0; i < cnt; ++i) {
      b.t[i] = tensors[base + i];
      if (!b.t[i].p || !b.t[i].g || !b.t[i].m || !b.t[i].v || b.t[i].n < 0) return fail_arg("tt_adam_dense: descriptor");
      if (b.t[i].n > max_n) max_n = b.t[i].n;
    }
    const int64_t bx = ceil_div(max_n, 1024) < 1024 ? ceil_div(max_n, 1024) : 1024;
    adam_dense_kernel<<<dim3((unsigned)bx, (unsigned)cnt), 256, 0, S(stream)>>>(b, hyper);
    const int rc = check_launch("adam_dense_kernel");
    if (rc) return rc;
  }
  return 0;
}

extern "C" int tt_pack_grads(const tt_adam_tensor* tensors, int32_t n_tensors, tt_stream_t stream) {
  if (!tensors) return fail_arg("tt_pack_grads: null pointer");
  if (n_tensors <= 0) return fail_arg("tt_pack_grads: sizes");
  for (int32_t base = 0; base < n_tensors; base += ADAM_BATCH) {
    const int32_t cnt = (n_tensors - base < ADAM_BATCH) ? n_tensors - base : ADAM_BATCH;
    AdamBatch b;
    int64_t max_n = 1;
    for (int32_t i = 0; i < cnt; ++i) {
      b.t[i] = tensors[base + i];
      if (!b.t[i].p || !b.t[i].g || b.t[i].n < 0) return fail_arg("tt_pack_grads: descriptor");
      if (b.t[i].n > max_n) max_n = b.t[i].n;
    }
    const int64_t bx = ceil_div(max_n, 1024) < 1024 ? ceil_div(max_n, 1024) : 1024;
    pack_grads_kernel<<<dim3((unsigned)bx, (unsigned)cnt), 256, 0, S(stream)>>>(b);
    const int rc = check_launch("pack_grads_kernel");
    if (rc) return rc;
  }
  return 0;
}

extern "C" int tt_copy_buffers(const tt_adam_tensor* buffers, int32_t n_buffers, tt_stream_t stream) {
  if (!buffers) return fail_arg("tt_copy_buffers: null pointer");
  if (n_buffers <= 0) return fail_arg("tt_copy_buffers: sizes");
  for (int32_t base = 0; base < n_buffers; base += ADAM_BATCH) {
    const int32_t cnt = (n_buffers - base < ADAM_BATCH) ? n_buffers - base : ADAM_BATCH;
    AdamBatch b;
    int64_t max_n = 16;
    for (int32_t i = 0; i < cnt; ++i) {
      b.t[i] = buffers[base + i];
      if (!b.t[i].p || !b.t[i].g || b.t[i].n < 0) return fail_arg("tt_copy_buffers: descriptor");
      if (b.t[i].n > max_n) max_n = b.t[i].n;
    }
    const int64_t bx = ceil_div(max_n / 16, 256) < 256 ? ceil_div(max_n / 16, 256) : 256;
    copy_buffers_kernel<<<dim3((unsigned)(bx < 1 ? 1 : bx), (unsigned)cnt), 256, 0, S(stream)>>>(b);
    const int rc = check_launch("copy_buffers_kernel");
    if (rc) return rc;
  }
  return 0;
}

// HBM calibration beside the sweep (bench.py `roofline.hbm_copy_GBps`): the plain streaming copy the guide's "measured
// copy rate" refers to -- per lane four 16-B non-temporal loads, then four 16-B non-temporal stores; a ONE-SHOT grid, one
// 16 KB chunk per workgroup.  Picked by measurement (tools/copy_probe.hip, profiles/r06_copy_probe.txt): 6.10 TB/s
// against 5.0-5.6 for persistent grids of 512-4096 workgroups, 4.4 for 8 float4 per lane and 4.45 for hipMemcpyDtoD.
// It moves 2 x bytes through HBM.
namespace tt {
__global__ __launch_bounds__(256) void stream_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
  constexpr int ITERS = 4;
  const int64_t base = (int64_t)blockIdx.x * (256 * ITERS) + threadIdx.x;
  float4 v[ITERS];
#pragma unroll
  for (int k = 0; k < ITERS; ++k) {
    const int64_t i = base + (int64_t)k * 256;
    if (i < n4) v[k] = sweep_load<true>(src + i);
  }
#pragma unroll
  for (int k = 0; k < ITERS; ++k) {
    const int64_t i = base + (int64_t)k * 256;
    if (i < n4) sweep_store<true>(v[k], dst + i);
  }
}
}  // namespace tt

extern "C" int tt_stream_copy(const void* src, void* dst, int64_t bytes, tt_stream_t stream) {
  if (!src || !dst) return fail_arg("tt_stream_copy: null pointer");
  if (bytes <= 0 || bytes % 16 || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15))
    return fail_arg("tt_stream_copy: 16-byte aligned buffers, a multiple of 16 bytes");
  const int64_t n4 = bytes / 16;
  const int64_t blocks = ceil_div(n4, 256 * 4);
  if (blocks >= (1ll << 31)) return fail_arg("tt_stream_copy: at most 32 TiB per call");
  stream_copy_kernel<<<(unsigned)blocks, 256, 0, S(stream)>>>(reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n4);
  return check_launch("stream_copy_kernel");
}

extern "C" int tt_rowgrad_dense(const tt_grad_sources* src, int64_t n_ids, int64_t dim, const int32_t* sorted_ids,
                                const int32_t* perm, const int32_t* seg_begin, const int32_t* n_unique,
                                float* dense_grad, tt_stream_t stream) {
  if (!sorted_ids || !perm || !seg_begin || !n_unique || !dense_grad) return fail_arg("tt_rowgrad_dense: null pointer");
  if (n_ids <= 0 || dim <= 0) return fail_arg("tt_rowgrad_dense: sizes");
  if (!check_sources(src, n_ids, dim)) return fail_arg("tt_rowgrad_dense: gradient sources");
  rowgrad_dense_kernel<<<(unsigned)ceil_div(n_ids, 4), 256, 0, S(stream)>>>(*src, INT64_MAX, dim, sorted_ids, perm, seg_begin, n_unique, dense_grad);
  return check_launch("rowgrad_dense_kernel");
}
