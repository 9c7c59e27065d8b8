// K2a: row-gradient plan.  The ids looked up in one step (8 K .. 420 K int64) are
// stable-radix-sorted by row id (8-bit digits, only as many passes as the table
// height needs), then cut into runs of equal id.  Everything downstream (summing
// the gradient rows that hit the same table row, dense-exact Adam on those rows,
// the dense gradient for torch.optim users) walks these runs in a fixed order, so
// the embedding backward is atomic-free and bit-reproducible.
//
// All sizes are device-independent of the DATA: kernels are launched for the worst
// case (n_ids runs) and read the actual run count from device memory, which keeps
// the whole step capturable in a hipGraph.
#include <stdlib.h>

#include "common.hpp"

namespace tt {

constexpr int TILE = 256;  // keys per workgroup per pass (one wavefront, 4 rounds): 209 K ids -> 816 waves

__global__ void plan_init_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t n_rows,
                                 int32_t* __restrict__ keys, int32_t* __restrict__ vals, int32_t* oob_flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t id = ids[i];
  if (id < 0 || id >= n_rows) { *oob_flag = 1; id = 0; }
  keys[i] = (int32_t)id;
  vals[i] = (int32_t)i;
}

__global__ __launch_bounds__(64) void radix_hist_kernel(const int32_t* __restrict__ keys, int64_t n, int shift,
                                                        int32_t* __restrict__ hist, int nblk) {
  __shared__ int32_t h[256];
  const int lane = threadIdx.x;
  for (int d = lane; d < 256; d += 64) h[d] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * TILE;
  // all 16 keys of this lane first (unconditional loads from clamped indices: one memory round
  // trip for the tile instead of one per round), then the LDS histogram
  int32_t k[TILE / 64];
#pragma unroll
  for (int rnd = 0; rnd < TILE / 64; ++rnd) {
    const int64_t i = base + rnd * 64 + lane;
    k[rnd] = keys[i < n ? i : n - 1];
  }
#pragma unroll
  for (int rnd = 0; rnd < TILE / 64; ++rnd) {
    const int64_t i = base + rnd * 64 + lane;
    if (i < n) atomicAdd(&h[(k[rnd] >> shift) & 255], 1);
  }
  __syncthreads();
  for (int d = lane; d < 256; d += 64) hist[(int64_t)d * nblk + blockIdx.x] = h[d];
}

// one wavefront per digit: exclusive scan of that digit's per-block counts (in place) + digit total
__global__ __launch_bounds__(64) void radix_scan_kernel(int32_t* __restrict__ hist, int nblk, int32_t* __restrict__ totals) {
  const int lane = threadIdx.x;
  int32_t* row = hist + (int64_t)blockIdx.x * nblk;
  int32_t carry = 0;
  for (int b0 = 0; b0 < nblk; b0 += 64) {
    const int b = b0 + lane;
    const int32_t v = (b < nblk) ? row[b] : 0;
    int32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (b < nblk) row[b] = carry + inc - v;
    carry += __shfl(inc, 63, 64);
  }
  if (lane == 0) totals[blockIdx.x] = carry;
}

__global__ __launch_bounds__(64) void radix_scatter_kernel(const int32_t* __restrict__ keys_in,
                                                           const int32_t* __restrict__ vals_in, int64_t n, int shift,
                                                           const int32_t* __restrict__ hist, int nblk,
                                                           const int32_t* __restrict__ totals,
                                                           int32_t* __restrict__ keys_out, int32_t* __restrict__ vals_out) {
  __shared__ int32_t base[256];
  const int lane = threadIdx.x;
  {  // base[d] = (sum of totals of smaller digits) + (this digit's count in earlier blocks)
    int32_t t[4], s = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) { t[c] = totals[4 * lane + c]; s += t[c]; }
    int32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    int32_t run = inc - s;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      base[4 * lane + c] = run + hist[(int64_t)(4 * lane + c) * nblk + blockIdx.x];
      run += t[c];
    }
  }
  __syncthreads();
  const int64_t tile0 = (int64_t)blockIdx.x * TILE;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int32_t kreg[TILE / 64], vreg[TILE / 64];  // the whole tile up front: see radix_hist_kernel
#pragma unroll
  for (int rnd = 0; rnd < TILE / 64; ++rnd) {
    const int64_t i = tile0 + rnd * 64 + lane;
    const int64_t ic = i < n ? i : n - 1;
    kreg[rnd] = keys_in[ic];
    vreg[rnd] = vals_in[ic];
  }
#pragma unroll
  for (int rnd = 0; rnd < TILE / 64; ++rnd) {
    const int64_t i = tile0 + rnd * 64 + lane;
    const bool active = i < n;
    const int32_t key = kreg[rnd];
    const int32_t val = vreg[rnd];
    const int d = (key >> shift) & 255;
    unsigned long long mask = __ballot(active);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const unsigned long long bal = __ballot((d >> bit) & 1);
      mask &= ((d >> bit) & 1) ? bal : ~bal;
    }
    // `mask` = active lanes holding the same digit; stable rank = those in lower lanes
    const int rank = __popcll(mask & lt_mask);
    int32_t pos = 0;
    if (active) pos = base[d] + rank;
    __builtin_amdgcn_wave_barrier();
    if (active && rank == 0) base[d] += __popcll(mask);
    __builtin_amdgcn_wave_barrier();
    if (active) { keys_out[pos] = key; vals_out[pos] = val; }
  }
}

// ---- runs of equal key -----------------------------------------------------
constexpr int SEG_TILE = 2048;  // keys per workgroup (256 threads x 8)
__global__ __launch_bounds__(256) void seg_count_kernel(const int32_t* __restrict__ keys, int64_t n,
                                                        int32_t* __restrict__ blk_heads) {
  __shared__ int32_t red[4];
  const int64_t base = (int64_t)blockIdx.x * SEG_TILE;
  int32_t c = 0;
  for (int k = 0; k < 8; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    if (i < n) c += (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
  }
  int32_t w = c;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) blk_heads[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// single workgroup: exclusive scan of the per-block head counts; writes n_unique and the end sentinel
__global__ __launch_bounds__(256) void seg_scan_kernel(int32_t* __restrict__ blk_heads, int nblk, int64_t n,
                                                       int32_t* __restrict__ n_unique, int32_t* __restrict__ seg_begin) {
  __shared__ int32_t wsum[4];
  __shared__ int32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b0 = 0; b0 < nblk; b0 += 256) {
    const int b = b0 + threadIdx.x;
    const int32_t v = (b < nblk) ? blk_heads[b] : 0;
    int32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int32_t carry = carry_s;
    if (b < nblk) blk_heads[b] = carry + woff + inc - v;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *n_unique = carry_s;
    seg_begin[carry_s] = (int32_t)n;
  }
}

__global__ __launch_bounds__(256) void seg_write_kernel(const int32_t* __restrict__ keys, int64_t n,
                                                        const int32_t* __restrict__ blk_off,
                                                        int32_t* __restrict__ seg_begin) {
  __shared__ int32_t wsum[4];
  __shared__ int32_t carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = blk_off[blockIdx.x];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SEG_TILE;
  for (int k = 0; k < 8; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    const int32_t head = (i < n && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
    int32_t inc = head;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int32_t carry = carry_s;
    if (head) seg_begin[carry + woff + inc - 1] = (int32_t)i;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + woff + inc;
    __syncthreads();
  }
}

// ---- small plans: the whole thing in ONE workgroup -------------------------
// n <= PS_MAX ids (a base-model lookup: B = 8192; the sharded trainer's W * cap ~ 9 K per table): LSD radix sort with
// 4-bit digits entirely in LDS + the run structure, one launch instead of ~15 (two tables per step: 30 of the ~70
// kernel launches the HOST enqueues per step).
// 512 threads; thread t owns positions [t*kpt, (t+1)*kpt) of the current order (blocked: position order = (thread,
// slot) order, which keeps every pass stable).  Keys (u32) and original positions (u16) ping-pong between two LDS
// images; per pass each thread counts its digits into its own column of cnt[digit][thread] (u16), the 16 x 512 counts
// are scanned in digit-major order, and each thread scatters its slots in order, bumping its own (now exclusive)
// counters.  Nothing but loop counters lives in registers: the sort runs NEXT TO the persistent sweep (3 waves per SIMD
// on every CU), so it must be a light workgroup -- the first version (1024 threads, the tile in registers, 127 VGPRs)
// could not be placed on any CU until the sweep was over and sat in the queue for 5.7 ms.  Images are padded by one word
// per 32 (a thread's slots are kpt words apart: unpadded that is a 16-way bank conflict).  Positions >= n carry the key
// 0xFFFFFFFF: digit 15 in every pass, they stay behind.
constexpr int PS_THREADS = 512;
constexpr int PS_KPT_MAX = 20;
constexpr int PS_MAX = PS_THREADS * PS_KPT_MAX;  // 10 240 ids
__device__ __forceinline__ int ps_pad(int i) { return i + (i >> 5); }

__device__ __forceinline__ uint32_t ps_block_excl_scan(uint32_t v, uint32_t* wsum, uint32_t& total) {
  // exclusive scan of one value per thread over the workgroup (PS_THREADS / 64 waves)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < PS_THREADS / 64; ++w) {
    const uint32_t x = wsum[w];
    if (w < wave) woff += x;
    tot += x;
  }
  total = tot;
  __syncthreads();
  return woff + inc - v;
}

__device__ __forceinline__ void plan_small_body(const int64_t* __restrict__ ids, int n, int64_t n_rows, int kpt,
                                                int passes, int32_t* __restrict__ sorted_ids,
                                                int32_t* __restrict__ perm, int32_t* __restrict__ seg_begin,
                                                int32_t* __restrict__ n_unique, int32_t* __restrict__ oob_flag) {
  extern __shared__ __attribute__((aligned(16))) uint32_t ps_smem[];
  const int cap = ps_pad(PS_THREADS * kpt) + 1;
  uint32_t* kbuf[2] = {ps_smem, ps_smem + cap};
  uint16_t* vbuf[2] = {reinterpret_cast<uint16_t*>(ps_smem + 2 * cap), reinterpret_cast<uint16_t*>(ps_smem + 2 * cap) + cap};
  uint16_t* cnt = vbuf[1] + cap + (cap & 1);  // [16][512]
  __shared__ uint32_t wsum[PS_THREADS / 64];
  const int t = threadIdx.x;
  for (int i = t; i < PS_THREADS * kpt; i += PS_THREADS) {  // coalesced load of the id list
    uint32_t key = 0xFFFFFFFFu;
    if (i < n) {
      int64_t id = ids[i];
      if (id < 0 || id >= n_rows) { *oob_flag = 1; id = 0; }
      key = (uint32_t)id;
    }
    kbuf[0][ps_pad(i)] = key;
    vbuf[0][ps_pad(i)] = (uint16_t)i;
  }
  __syncthreads();
  int cur = 0;
  for (int p = 0; p < passes; ++p) {
    const int shift = 4 * p;
    const uint32_t* kin = kbuf[cur];
    const uint16_t* vin = vbuf[cur];
#pragma unroll
    for (int d = 0; d < 16; ++d) cnt[d * PS_THREADS + t] = 0;
    for (int s = 0; s < kpt; ++s) cnt[((kin[ps_pad(t * kpt + s)] >> shift) & 15) * PS_THREADS + t] += 1;
    __syncthreads();
    {  // exclusive scan of cnt in flat (digit-major) order: thread t takes flat entries [16t, 16t + 16)
      uint16_t* mine = cnt + 16 * t;
      uint32_t loc[16], sum = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) { loc[j] = sum; sum += mine[j]; }
      uint32_t total;
      const uint32_t base = ps_block_excl_scan(sum, wsum, total);
#pragma unroll
      for (int j = 0; j < 16; ++j) mine[j] = (uint16_t)(base + loc[j]);
    }
    __syncthreads();
    uint32_t* kout = kbuf[cur ^ 1];
    uint16_t* vout = vbuf[cur ^ 1];
    for (int s = 0; s < kpt; ++s) {  // in slot order: equal digits keep their order
      const uint32_t key = kin[ps_pad(t * kpt + s)];
      uint16_t* c = cnt + ((key >> shift) & 15) * PS_THREADS + t;
      const int pos = ps_pad((int)*c);
      *c += 1;
      kout[pos] = key;
      vout[pos] = vin[ps_pad(t * kpt + s)];
    }
    __syncthreads();
    cur ^= 1;
  }
  const uint32_t* ks = kbuf[cur];
  const uint16_t* vs = vbuf[cur];
  // outputs (coalesced) and run heads (blocked, so that the scan below yields run numbers in position order)
  for (int i = t; i < n; i += PS_THREADS) {
    sorted_ids[i] = (int32_t)ks[ps_pad(i)];
    perm[i] = (int32_t)vs[ps_pad(i)];
  }
  uint32_t heads = 0;
  for (int s = 0; s < kpt; ++s) {
    const int i = t * kpt + s;
    if (i < n && (i == 0 || ks[ps_pad(i - 1)] != ks[ps_pad(i)])) heads += 1;
  }
  uint32_t total;
  uint32_t seg = ps_block_excl_scan(heads, wsum, total);
  for (int s = 0; s < kpt; ++s) {
    const int i = t * kpt + s;
    if (i < n && (i == 0 || ks[ps_pad(i - 1)] != ks[ps_pad(i)])) seg_begin[seg++] = i;
  }
  if (t == 0) {
    *n_unique = (int32_t)total;
    seg_begin[total] = n;
  }
}

__global__ __launch_bounds__(PS_THREADS) void plan_small_kernel(const int64_t* __restrict__ ids, int n, int64_t n_rows, int kpt,
                                                                int passes, int32_t* __restrict__ sorted_ids,
                                                                int32_t* __restrict__ perm, int32_t* __restrict__ seg_begin,
                                                                int32_t* __restrict__ n_unique, int32_t* __restrict__ oob_flag) {
  plan_small_body(ids, n, n_rows, kpt, passes, sorted_ids, perm, seg_begin, n_unique, oob_flag);
}

// the same one-workgroup sort for SEVERAL tables' id lists in one launch (tt_rowgrad_plan_jobs): one workgroup per list --
// the two 40-us sorts of a base-model step (user ids, item ids) ran back to back on one stream
struct PlanSmallJobs {
  const int64_t* ids[TT_PLAN_MAX_JOBS];
  int n[TT_PLAN_MAX_JOBS], kpt[TT_PLAN_MAX_JOBS], passes[TT_PLAN_MAX_JOBS];
  int64_t n_rows[TT_PLAN_MAX_JOBS];
  int32_t *sorted_ids[TT_PLAN_MAX_JOBS], *perm[TT_PLAN_MAX_JOBS], *seg_begin[TT_PLAN_MAX_JOBS], *n_unique[TT_PLAN_MAX_JOBS];
};
__global__ __launch_bounds__(PS_THREADS) void plan_small_jobs_kernel(const PlanSmallJobs q, int32_t* __restrict__ oob_flag) {
  const int j = blockIdx.x;
  plan_small_body(q.ids[j], q.n[j], q.n_rows[j], q.kpt[j], q.passes[j], q.sorted_ids[j], q.perm[j], q.seg_begin[j], q.n_unique[j],
                  oob_flag);
}

static int radix_passes(int64_t n_rows) {
  int bits = 1;
  while (bits < 31 && ((int64_t)1 << bits) < n_rows) ++bits;
  return (bits + 7) / 8;
}

}  // namespace tt

using namespace tt;

extern "C" int64_t tt_rowgrad_workspace_bytes(int64_t n_ids) {
  if (n_ids <= 0) return 256;
  const int64_t nblk = ceil_div(n_ids, TILE);
  const int64_t nseg = ceil_div(n_ids, SEG_TILE);
  return round_up(n_ids * 4, 256) * 2      // ping-pong keys / vals
         + round_up(256 * nblk * 4, 256)   // per-block digit histograms
         + round_up(256 * 4, 256)          // digit totals
         + round_up(nseg * 4, 256);        // per-block run-head counts
}

extern "C" int tt_rowgrad_plan(const int64_t* ids, int64_t n_ids, int64_t n_rows, int32_t* sorted_ids,
                               int32_t* perm, int32_t* seg_begin, int32_t* n_unique, int32_t* oob_flag,
                               void* ws, int64_t ws_bytes, tt_stream_t stream) {
  if (!ids || !sorted_ids || !perm || !seg_begin || !n_unique || !oob_flag || !ws)
    return fail_arg("tt_rowgrad_plan: null pointer");
  if (n_ids <= 0 || n_ids >= ((int64_t)1 << 31) || n_rows <= 0 || n_rows > ((int64_t)1 << 31))
    return fail_arg("tt_rowgrad_plan: sizes");
  if (ws_bytes < tt_rowgrad_workspace_bytes(n_ids)) { set_error("tt_rowgrad_plan: workspace"); return TT_E_WORKSPACE; }
  hipStream_t st = S(stream);
  if (n_ids <= PS_MAX) {
    int bits = 1;
    while (bits < 31 && ((int64_t)1 << bits) < n_rows) ++bits;
    const int kpt = (int)ceil_div(n_ids, PS_THREADS);
    const size_t cap = (size_t)(PS_THREADS * kpt + (PS_THREADS * kpt >> 5)) + 1;
    const size_t lds = 2 * cap * 4 + 2 * cap * 2 + 4 + 16 * PS_THREADS * 2;
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(plan_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("plan_small_kernel: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    }
    plan_small_kernel<<<1, PS_THREADS, lds, st>>>(ids, (int)n_ids, n_rows, kpt, n_rows > 1 ? (bits + 3) / 4 : 0, sorted_ids, perm,
                                                  seg_begin, n_unique, oob_flag);
    return check_launch("plan_small_kernel");
  }
  const int nblk = (int)ceil_div(n_ids, TILE);
  const int nseg = (int)ceil_div(n_ids, SEG_TILE);
  Carver cv(ws);
  int32_t* tkeys = cv.take<int32_t>(n_ids);
  int32_t* tvals = cv.take<int32_t>(n_ids);
  int32_t* hist = cv.take<int32_t>((int64_t)256 * nblk);
  int32_t* totals = cv.take<int32_t>(256);
  int32_t* blk_heads = cv.take<int32_t>(nseg);

  const int passes = radix_passes(n_rows);
  // ping-pong so that the LAST pass lands in (sorted_ids, perm)
  int32_t* kbuf[2] = {sorted_ids, tkeys};
  int32_t* vbuf[2] = {perm, tvals};
  int cur = (passes % 2 == 0) ? 0 : 1;
  plan_init_kernel<<<(unsigned)ceil_div(n_ids, 256), 256, 0, st>>>(ids, n_ids, n_rows, kbuf[cur], vbuf[cur], oob_flag);
  int rc = check_launch("plan_init_kernel");
  if (rc) return rc;
  for (int p = 0; p < passes; ++p) {
    const int shift = 8 * p;
    radix_hist_kernel<<<nblk, 64, 0, st>>>(kbuf[cur], n_ids, shift, hist, nblk);
    if ((rc = check_launch("radix_hist_kernel"))) return rc;
    radix_scan_kernel<<<256, 64, 0, st>>>(hist, nblk, totals);
    if ((rc = check_launch("radix_scan_kernel"))) return rc;
    radix_scatter_kernel<<<nblk, 64, 0, st>>>(kbuf[cur], vbuf[cur], n_ids, shift, hist, nblk, totals, kbuf[cur ^ 1], vbuf[cur ^ 1]);
    if ((rc = check_launch("radix_scatter_kernel"))) return rc;
    cur ^= 1;
  }
  seg_count_kernel<<<nseg, 256, 0, st>>>(sorted_ids, n_ids, blk_heads);
  if ((rc = check_launch("seg_count_kernel"))) return rc;
  seg_scan_kernel<<<1, 256, 0, st>>>(blk_heads, nseg, n_ids, n_unique, seg_begin);
  if ((rc = check_launch("seg_scan_kernel"))) return rc;
  seg_write_kernel<<<nseg, 256, 0, st>>>(sorted_ids, n_ids, blk_heads, seg_begin);
  return check_launch("seg_write_kernel");
}

extern "C" int tt_rowgrad_plan_jobs_supported(int64_t n_ids) { return (n_ids > 0 && n_ids <= PS_MAX) ? 1 : 0; }

extern "C" int tt_rowgrad_plan_jobs(const tt_plan_job* jobs, int32_t n_jobs, int32_t* oob_flag, tt_stream_t stream) {
  if (!jobs || !oob_flag) return fail_arg("tt_rowgrad_plan_jobs: null pointer");
  if (n_jobs <= 0 || n_jobs > TT_PLAN_MAX_JOBS) return fail_arg("tt_rowgrad_plan_jobs: 1..4 jobs");
  PlanSmallJobs q{};
  size_t lds = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const tt_plan_job& b = jobs[j];
    if (!b.ids || !b.sorted_ids || !b.perm || !b.seg_begin || !b.n_unique) return fail_arg("tt_rowgrad_plan_jobs: null pointer");
    if (b.n_ids <= 0 || b.n_ids > PS_MAX || b.n_rows <= 0 || b.n_rows > ((int64_t)1 << 31)) {
      set_error("tt_rowgrad_plan_jobs: lists of 1..%d ids (longer ones: tt_rowgrad_plan)", PS_MAX);
      return TT_E_UNSUPPORTED;
    }
    int bits = 1;
    while (bits < 31 && ((int64_t)1 << bits) < b.n_rows) ++bits;
    const int kpt = (int)ceil_div(b.n_ids, PS_THREADS);
    const size_t cap = (size_t)(PS_THREADS * kpt + (PS_THREADS * kpt >> 5)) + 1;
    const size_t need = 2 * cap * 4 + 2 * cap * 2 + 4 + 16 * PS_THREADS * 2;
    lds = need > lds ? need : lds;
    q.ids[j] = b.ids; q.n[j] = (int)b.n_ids; q.n_rows[j] = b.n_rows; q.kpt[j] = kpt; q.passes[j] = b.n_rows > 1 ? (bits + 3) / 4 : 0;
    q.sorted_ids[j] = b.sorted_ids; q.perm[j] = b.perm; q.seg_begin[j] = b.seg_begin; q.n_unique[j] = b.n_unique;
  }
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(plan_small_jobs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("plan_small_jobs_kernel: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  plan_small_jobs_kernel<<<(unsigned)n_jobs, PS_THREADS, lds, S(stream)>>>(q, oob_flag);
  return check_launch("plan_small_jobs_kernel");
}
