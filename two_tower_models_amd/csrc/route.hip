// R1: owner routing for row-sharded embedding tables.  The reference has no
// parallelism; this is the device side of the padded all-to-all that replaces "every owner gathers
// rows for every rank's ids": a rank sends each owner only the ids that owner holds.
//
// Tables are split into `world` contiguous blocks of rows_per_rank rows, so owner(id) = id / rows_per_rank.
// Bucketing a rank's n ids (8 K .. 420 K) by owner is ONE stable counting pass -- no sort:
//
//   tt_route_count   per 1024-id tile an owner histogram, then one small scan: tile_off[tile][o] = ids of
//                    owner o in earlier tiles, and the largest bucket -> *max_count.  The caller all-reduces
//                    that (MAX) into the per-peer capacity `cap` of the fixed-size exchange (one step ahead,
//                    so the host never waits for it).
//   tt_route_build   slot of id i = owner * cap + (number of earlier ids with the same owner): each tile
//                    ranks its ids with wave ballots + a 16-wave prefix in LDS on top of tile_off.
//                    send_ids[slot] = id (-1 padding), slot_of[i] = slot (its row comes back there),
//                    src_of[slot] = i or -1 (the backward sends gradient row src_of[slot] in that slot).
//   tt_route_localize (owner side) received global ids -> row offsets inside the block, with the
//                    sentinel n_local for padding: what tt_gather_rows (zero rows) and the Adam plan
//                    (skipped run) expect.
// Placement is by position in the id list, hence deterministic: the owner sees the ids of one requester in
// request order, and sums the gradients of equal ids in that order.
#include "common.hpp"

namespace tt {

constexpr int RT = 1024;  // ids per tile = threads per workgroup

__device__ __forceinline__ int32_t owner_of(int64_t id, int64_t n_rows, int64_t rows_per_rank, int32_t world, bool& bad) {
  bad = id < 0 || id >= n_rows;
  if (bad) return 0;
  const int64_t o = id / rows_per_rank;
  return (int32_t)(o < world ? o : world - 1);
}

__global__ __launch_bounds__(RT) void route_hist_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t n_rows,
                                                        int64_t rows_per_rank, int32_t world,
                                                        int32_t* __restrict__ tile_hist, int32_t* __restrict__ oob_flag) {
  extern __shared__ int32_t cnt[];
  for (int o = threadIdx.x; o < world; o += RT) cnt[o] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * RT + threadIdx.x;
  if (i < n) {
    bool bad;
    const int32_t o = owner_of(ids[i], n_rows, rows_per_rank, world, bad);
    if (bad && oob_flag) *oob_flag = 1;
    atomicAdd(&cnt[o], 1);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < world; o += RT) tile_hist[(int64_t)blockIdx.x * world + o] = cnt[o];
}

// one thread per owner: exclusive prefix of its counts over the tiles (in place) and the bucket total
__global__ __launch_bounds__(256) void route_scan_kernel(int32_t* __restrict__ tile_hist, int64_t n_tiles, int32_t world,
                                                         int32_t* __restrict__ counts, int32_t* __restrict__ max_count) {
  int32_t mx = 0;
  for (int32_t o = threadIdx.x; o < world; o += blockDim.x) {
    int32_t run = 0;
    for (int64_t t = 0; t < n_tiles; ++t) {
      const int32_t c = tile_hist[t * world + o];
      tile_hist[t * world + o] = run;
      run += c;
    }
    counts[o] = run;
    mx = max(mx, run);
  }
  if (mx > 0) atomicMax(max_count, mx);
}

__global__ __launch_bounds__(256) void route_fill_kernel(int64_t* __restrict__ send_ids, int64_t* __restrict__ src_of,
                                                         int64_t n_slots) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  send_ids[s] = -1;
  src_of[s] = -1;
}

__global__ __launch_bounds__(RT) void route_build_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t n_rows,
                                                         int64_t rows_per_rank, int32_t world, int64_t cap,
                                                         const int32_t* __restrict__ tile_off,
                                                         int64_t* __restrict__ send_ids, int64_t* __restrict__ slot_of,
                                                         int64_t* __restrict__ src_of, int32_t* __restrict__ overflow) {
  extern __shared__ int32_t wcnt[];  // [16 waves][world]: ids of owner o held by wave w of this tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < 16 * world; k += RT) wcnt[k] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * RT + threadIdx.x;
  const bool live = i < n;
  int64_t id = 0;
  int32_t o = -1;
  if (live) {
    bool bad;
    id = ids[i];
    o = owner_of(id, n_rows, rows_per_rank, world, bad);
    if (bad) id = 0;  // reported by tt_route_count's flag; row 0 keeps every later index in range
  }
  // rank among the wave's ids with the same owner (lane order = list order)
  int32_t rank = 0;
  uint64_t todo = __ballot(live);
  while (todo) {
    const int lead = __ffsll((unsigned long long)todo) - 1;
    const int32_t ol = __shfl(o, lead, 64);
    const uint64_t same = __ballot(live && o == ol);
    if (live && o == ol) {
      rank = __popcll(same & ((1ull << lane) - 1ull));
      if (lane == lead) wcnt[wave * world + ol] = __popcll(same);
    }
    todo &= ~same;
  }
  __syncthreads();
  if (!live) return;
  for (int w = 0; w < wave; ++w) rank += wcnt[w * world + o];
  const int64_t r = (int64_t)tile_off[(int64_t)blockIdx.x * world + o] + rank;
  if (r >= cap) {  // cap came from an older count: nothing is sent for this id, and it is reported
    *overflow = 1;
    slot_of[i] = -1;
    return;
  }
  const int64_t slot = (int64_t)o * cap + r;
  send_ids[slot] = id;
  src_of[slot] = i;
  slot_of[i] = slot;
}

__global__ __launch_bounds__(256) void route_localize_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t lo,
                                                             int64_t n_local, int64_t* __restrict__ local) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t r = ids[i] - lo;
  local[i] = (ids[i] >= 0 && r >= 0 && r < n_local) ? r : n_local;
}

constexpr int32_t ROUTE_MAX_WORLD = 1024;  // 16 x world x 4 B of LDS in route_build_kernel (64 KiB)

// ---------------------------------------------------------------------------------------------------------------------
// The same three stages for ALL lookups of a step at once (tt_route_*_jobs): a step of the base model routes two id lists
// (user ids, item ids), the history model three, each a few thousand ids -- 3-7 us kernels that were 6-15 us apart.  One
// launch per stage covers every lookup (blockIdx -> job through first_block[], like the Adam stash / finish jobs), the
// -1 fill of the send lists is folded into the build launch (a slot o*cap + r is padding iff r >= counts[o]: no second
// writer), the scan WRITES the bucket maximum (nothing to zero first), and the owner side localises the received ids and
// gathers their rows in the same launch.
struct RouteJobs {
  const int64_t* ids[TT_ROUTE_MAX_JOBS];
  int64_t n_ids[TT_ROUTE_MAX_JOBS], n_rows[TT_ROUTE_MAX_JOBS], rows_per_rank[TT_ROUTE_MAX_JOBS], cap[TT_ROUTE_MAX_JOBS];
  int32_t* counts[TT_ROUTE_MAX_JOBS];
  int32_t* max_count[TT_ROUTE_MAX_JOBS];
  int32_t* tile[TT_ROUTE_MAX_JOBS];
  int64_t* send_ids[TT_ROUTE_MAX_JOBS];
  int64_t* slot_of[TT_ROUTE_MAX_JOBS];
  int64_t* src_of[TT_ROUTE_MAX_JOBS];
  unsigned first_block[TT_ROUTE_MAX_JOBS + 1];
  int n;
};

__device__ __forceinline__ int job_of(const unsigned* first_block, int n, unsigned b) {
  int j = 0;
#pragma unroll
  for (int q = 1; q < TT_ROUTE_MAX_JOBS; ++q)
    if (q < n && b >= first_block[q]) j = q;
  return j;
}

__global__ __launch_bounds__(RT) void route_hist_jobs_kernel(const RouteJobs jobs, int32_t world, int32_t* __restrict__ oob_flag) {
  extern __shared__ int32_t cnt[];
  const int j = job_of(jobs.first_block, jobs.n, blockIdx.x);
  const unsigned tile = blockIdx.x - jobs.first_block[j];
  for (int o = threadIdx.x; o < world; o += RT) cnt[o] = 0;
  __syncthreads();
  const int64_t i = (int64_t)tile * RT + threadIdx.x;
  if (i < jobs.n_ids[j]) {
    bool bad;
    const int32_t o = owner_of(jobs.ids[j][i], jobs.n_rows[j], jobs.rows_per_rank[j], world, bad);
    if (bad && oob_flag) *oob_flag = 1;
    atomicAdd(&cnt[o], 1);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < world; o += RT) jobs.tile[j][(int64_t)tile * world + o] = cnt[o];
}

// one workgroup per job: exclusive prefix over the tiles per owner (in place), bucket totals, the largest bucket
__global__ __launch_bounds__(256) void route_scan_jobs_kernel(const RouteJobs jobs, int32_t world) {
  __shared__ int32_t s_max[256];
  const int j = blockIdx.x;
  int32_t* tile_hist = jobs.tile[j];
  const int64_t n_tiles = (jobs.n_ids[j] + RT - 1) / RT;
  int32_t mx = 0;
  for (int32_t o = threadIdx.x; o < world; o += 256) {
    int32_t run = 0;
    for (int64_t t = 0; t < n_tiles; ++t) {
      const int32_t c = tile_hist[t * world + o];
      tile_hist[t * world + o] = run;
      run += c;
    }
    jobs.counts[j][o] = run;
    mx = max(mx, run);
  }
  s_max[threadIdx.x] = mx;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) s_max[threadIdx.x] = max(s_max[threadIdx.x], s_max[threadIdx.x + k]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *jobs.max_count[j] = s_max[0];
}

// blocks [first_block[j], first_block[j] + tiles_j): placement of job j's ids; the following pad blocks of the job write -1
// into the slots no id will claim
__global__ __launch_bounds__(RT) void route_build_jobs_kernel(const RouteJobs jobs, int32_t world, int32_t* __restrict__ overflow) {
  extern __shared__ int32_t wcnt[];  // [16 waves][world]
  const int j = job_of(jobs.first_block, jobs.n, blockIdx.x);
  const unsigned blk = blockIdx.x - jobs.first_block[j];
  const int64_t n = jobs.n_ids[j], cap = jobs.cap[j];
  const unsigned tiles = (unsigned)((n + RT - 1) / RT);
  if (blk >= tiles) {  // padding slots: r >= counts[o]
    const int64_t s = (int64_t)(blk - tiles) * RT + threadIdx.x;
    if (s < (int64_t)world * cap) {
      const int64_t o = s / cap, r = s - o * cap;
      if (r >= jobs.counts[j][o]) {
        jobs.send_ids[j][s] = -1;
        jobs.src_of[j][s] = -1;
      }
    }
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < 16 * world; k += RT) wcnt[k] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blk * RT + threadIdx.x;
  const bool live = i < n;
  int64_t id = 0;
  int32_t o = -1;
  if (live) {
    bool bad;
    id = jobs.ids[j][i];
    o = owner_of(id, jobs.n_rows[j], jobs.rows_per_rank[j], world, bad);
    if (bad) id = 0;
  }
  int32_t rank = 0;
  uint64_t todo = __ballot(live);
  while (todo) {
    const int lead = __ffsll((unsigned long long)todo) - 1;
    const int32_t ol = __shfl(o, lead, 64);
    const uint64_t same = __ballot(live && o == ol);
    if (live && o == ol) {
      rank = __popcll(same & ((1ull << lane) - 1ull));
      if (lane == lead) wcnt[wave * world + ol] = __popcll(same);
    }
    todo &= ~same;
  }
  __syncthreads();
  if (!live) return;
  for (int w = 0; w < wave; ++w) rank += wcnt[w * world + o];
  const int64_t r = (int64_t)jobs.tile[j][(int64_t)blk * world + o] + rank;
  if (r >= cap) {
    *overflow = 1;
    jobs.slot_of[j][i] = -1;
    return;
  }
  const int64_t slot = (int64_t)o * cap + r;
  jobs.send_ids[j][slot] = id;
  jobs.src_of[j][slot] = i;
  jobs.slot_of[j][i] = slot;
}

// owner side: received global ids -> local row numbers (sentinel n_local) AND the rows themselves, 32 lanes per row
struct ServeJobs {
  const int64_t* ids[TT_ROUTE_MAX_JOBS];
  int64_t n_ids[TT_ROUTE_MAX_JOBS], lo[TT_ROUTE_MAX_JOBS], n_local[TT_ROUTE_MAX_JOBS], dim[TT_ROUTE_MAX_JOBS];
  int64_t* local[TT_ROUTE_MAX_JOBS];
  const void* table[TT_ROUTE_MAX_JOBS];
  int dtype[TT_ROUTE_MAX_JOBS];
  float* rows[TT_ROUTE_MAX_JOBS];
  unsigned first_block[TT_ROUTE_MAX_JOBS + 1];
  int n;
};

__global__ __launch_bounds__(256) void route_serve_jobs_kernel(const ServeJobs jobs) {
  const int j = job_of(jobs.first_block, jobs.n, blockIdx.x);
  const int c = threadIdx.x & 31;
  const int64_t i = (int64_t)(blockIdx.x - jobs.first_block[j]) * 8 + (threadIdx.x >> 5);
  if (i >= jobs.n_ids[j]) return;
  const int64_t id = jobs.ids[j][i], r = id - jobs.lo[j], n_local = jobs.n_local[j], dim = jobs.dim[j];
  const bool mine = id >= 0 && r >= 0 && r < n_local;
  if (c == 0) jobs.local[j][i] = mine ? r : n_local;
  float* __restrict__ out = jobs.rows[j] + i * dim;
  if (jobs.dtype[j] == TT_BF16) {
    const uint16_t* src = reinterpret_cast<const uint16_t*>(jobs.table[j]) + (mine ? r : 0) * dim;
    for (int64_t k = c; k < dim; k += 32) out[k] = mine ? __uint_as_float((uint32_t)src[k] << 16) : 0.f;
  } else if ((dim & 3) == 0) {
    const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(jobs.table[j]) + (mine ? r : 0) * dim);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (int64_t k = c; k < dim / 4; k += 32) o4[k] = mine ? src[k] : make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    const float* src = reinterpret_cast<const float*>(jobs.table[j]) + (mine ? r : 0) * dim;
    for (int64_t k = c; k < dim; k += 32) out[k] = mine ? src[k] : 0.f;
  }
}

}  // namespace tt

using namespace tt;

extern "C" int64_t tt_route_workspace_bytes(int64_t n_ids, int32_t world) {
  if (n_ids <= 0 || world <= 0) return 256;
  return round_up(ceil_div(n_ids, RT) * (int64_t)world * 4, 256);
}

extern "C" int tt_route_count(const int64_t* ids, int64_t n_ids, int64_t n_rows, int64_t rows_per_rank, int32_t world,
                              int32_t* counts, int32_t* max_count, int32_t* oob_flag, void* ws, int64_t ws_bytes,
                              tt_stream_t stream) {
  if (!ids || !counts || !max_count || !ws) return fail_arg("tt_route_count: null pointer");
  if (n_ids <= 0 || n_rows <= 0 || rows_per_rank <= 0 || world <= 0 || world > ROUTE_MAX_WORLD)
    return fail_arg("tt_route_count: sizes");
  if (ws_bytes < tt_route_workspace_bytes(n_ids, world)) { set_error("tt_route_count: workspace"); return TT_E_WORKSPACE; }
  hipStream_t st = S(stream);
  const int64_t tiles = ceil_div(n_ids, RT);
  int32_t* hist = reinterpret_cast<int32_t*>(ws);
  route_hist_kernel<<<(unsigned)tiles, RT, world * sizeof(int32_t), st>>>(ids, n_ids, n_rows, rows_per_rank, world, hist, oob_flag);
  int rc = check_launch("route_hist_kernel");
  if (rc) return rc;
  route_scan_kernel<<<1, 256, 0, st>>>(hist, tiles, world, counts, max_count);
  return check_launch("route_scan_kernel");
}

extern "C" int tt_route_build(const int64_t* ids, int64_t n_ids, int64_t n_rows, int64_t rows_per_rank, int32_t world,
                              int64_t cap, const void* ws, int64_t ws_bytes, int64_t* send_ids, int64_t* slot_of,
                              int64_t* src_of, int32_t* overflow_flag, tt_stream_t stream) {
  if (!ids || !ws || !send_ids || !slot_of || !src_of || !overflow_flag) return fail_arg("tt_route_build: null pointer");
  if (n_ids <= 0 || n_rows <= 0 || rows_per_rank <= 0 || world <= 0 || world > ROUTE_MAX_WORLD || cap <= 0)
    return fail_arg("tt_route_build: sizes");
  if (ws_bytes < tt_route_workspace_bytes(n_ids, world)) { set_error("tt_route_build: workspace"); return TT_E_WORKSPACE; }
  hipStream_t st = S(stream);
  const int64_t n_slots = (int64_t)world * cap;
  route_fill_kernel<<<(unsigned)ceil_div(n_slots, 256), 256, 0, st>>>(send_ids, src_of, n_slots);
  int rc = check_launch("route_fill_kernel");
  if (rc) return rc;
  route_build_kernel<<<(unsigned)ceil_div(n_ids, RT), RT, 16 * world * sizeof(int32_t), st>>>(
      ids, n_ids, n_rows, rows_per_rank, world, cap, reinterpret_cast<const int32_t*>(ws), send_ids, slot_of, src_of,
      overflow_flag);
  return check_launch("route_build_kernel");
}

extern "C" int tt_route_localize(const int64_t* ids, int64_t n_ids, int64_t lo, int64_t n_local, int64_t* local,
                                 tt_stream_t stream) {
  if (!ids || !local) return fail_arg("tt_route_localize: null pointer");
  if (n_ids <= 0 || lo < 0 || n_local < 0) return fail_arg("tt_route_localize: sizes");
  route_localize_kernel<<<(unsigned)ceil_div(n_ids, 256), 256, 0, S(stream)>>>(ids, n_ids, lo, n_local, local);
  return check_launch("route_localize_kernel");
}

// ---- all lookups of a step per launch (see RouteJobs above)
static int fill_jobs(const tt_route_job* jobs, int32_t n_jobs, int32_t world, bool build, RouteJobs& a, const char* who) {
  if (!jobs) return fail_arg(who);
  if (n_jobs <= 0 || n_jobs > TT_ROUTE_MAX_JOBS || world <= 0 || world > ROUTE_MAX_WORLD) return fail_arg(who);
  a.n = n_jobs;
  unsigned blocks = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const tt_route_job& q = jobs[j];
    if (!q.ids || !q.counts || !q.max_count || !q.ws) return fail_arg(who);
    if (q.n_ids <= 0 || q.n_rows <= 0 || q.rows_per_rank <= 0) return fail_arg(who);
    if (q.ws_bytes < tt_route_workspace_bytes(q.n_ids, world)) { set_error("%s: workspace", who); return TT_E_WORKSPACE; }
    if (build && (!q.send_ids || !q.slot_of || !q.src_of || q.cap <= 0)) return fail_arg(who);
    a.ids[j] = q.ids; a.n_ids[j] = q.n_ids; a.n_rows[j] = q.n_rows; a.rows_per_rank[j] = q.rows_per_rank; a.cap[j] = q.cap;
    a.counts[j] = q.counts; a.max_count[j] = q.max_count; a.tile[j] = reinterpret_cast<int32_t*>(q.ws);
    a.send_ids[j] = q.send_ids; a.slot_of[j] = q.slot_of; a.src_of[j] = q.src_of;
    a.first_block[j] = blocks;
    blocks += (unsigned)ceil_div(q.n_ids, RT);
    if (build) blocks += (unsigned)ceil_div((int64_t)world * q.cap, RT);
  }
  for (int j = n_jobs; j <= TT_ROUTE_MAX_JOBS; ++j) a.first_block[j] = blocks;
  return 0;
}

extern "C" int tt_route_count_jobs(const tt_route_job* jobs, int32_t n_jobs, int32_t world, int32_t* oob_flag,
                                   tt_stream_t stream) {
  RouteJobs a{};
  int rc = fill_jobs(jobs, n_jobs, world, false, a, "tt_route_count_jobs");
  if (rc) return rc;
  hipStream_t st = S(stream);
  route_hist_jobs_kernel<<<a.first_block[n_jobs], RT, world * sizeof(int32_t), st>>>(a, world, oob_flag);
  rc = check_launch("route_hist_jobs_kernel");
  if (rc) return rc;
  route_scan_jobs_kernel<<<(unsigned)n_jobs, 256, 0, st>>>(a, world);
  return check_launch("route_scan_jobs_kernel");
}

extern "C" int tt_route_build_jobs(const tt_route_job* jobs, int32_t n_jobs, int32_t world, int32_t* overflow_flag,
                                   tt_stream_t stream) {
  if (!overflow_flag) return fail_arg("tt_route_build_jobs: null pointer");
  RouteJobs a{};
  int rc = fill_jobs(jobs, n_jobs, world, true, a, "tt_route_build_jobs");
  if (rc) return rc;
  route_build_jobs_kernel<<<a.first_block[n_jobs], RT, 16 * world * sizeof(int32_t), S(stream)>>>(a, world, overflow_flag);
  return check_launch("route_build_jobs_kernel");
}

extern "C" int tt_route_serve_jobs(const tt_route_serve_job* jobs, int32_t n_jobs, tt_stream_t stream) {
  if (!jobs) return fail_arg("tt_route_serve_jobs: null pointer");
  if (n_jobs <= 0 || n_jobs > TT_ROUTE_MAX_JOBS) return fail_arg("tt_route_serve_jobs: 1..8 jobs");
  ServeJobs a{};
  a.n = n_jobs;
  unsigned blocks = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const tt_route_serve_job& q = jobs[j];
    if (!q.ids || !q.local || !q.rows || (!q.table && q.n_local > 0)) return fail_arg("tt_route_serve_jobs: null pointer");
    if (q.n_ids <= 0 || q.lo < 0 || q.n_local < 0 || q.dim <= 0 || (q.dtype != TT_F32 && q.dtype != TT_BF16))
      return fail_arg("tt_route_serve_jobs: sizes");
    if (q.dtype == TT_F32 && (q.dim & 3) == 0 && ((reinterpret_cast<uintptr_t>(q.table) | reinterpret_cast<uintptr_t>(q.rows)) & 15))
      return fail_arg("tt_route_serve_jobs: 16-byte aligned table and rows");
    a.ids[j] = q.ids; a.n_ids[j] = q.n_ids; a.lo[j] = q.lo; a.n_local[j] = q.n_local; a.dim[j] = q.dim;
    a.local[j] = q.local; a.table[j] = q.table; a.dtype[j] = q.dtype; a.rows[j] = q.rows;
    a.first_block[j] = blocks;
    blocks += (unsigned)ceil_div(q.n_ids, 8);
  }
  for (int j = n_jobs; j <= TT_ROUTE_MAX_JOBS; ++j) a.first_block[j] = blocks;
  route_serve_jobs_kernel<<<blocks, 256, 0, S(stream)>>>(a);
  return check_launch("route_serve_jobs_kernel");
}
