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
