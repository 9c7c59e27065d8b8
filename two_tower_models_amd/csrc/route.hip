// R1: owner routing for row-sharded embedding tables.  The reference has no
// parallelism; this is the device side of the padded all-to-all that replaces "every owner gathers
// rows for every rank's ids": a rank sends each owner only the ids that owner holds.
//
// Tables are split into `world` contiguous blocks of rows_per_rank rows, so the stable sort the row
// plan already performs (tt_rowgrad_plan: sorted_ids, perm) groups a rank's ids BY OWNER.  From it:
//
//   tt_route_count   starts[o] = first sorted position owned by rank o (binary search), counts, and
//                    the largest bucket -- all-reduced (MAX) by the caller into the per-peer capacity
//                    `cap` of the fixed-size exchange (planned one step ahead, so the host never waits)
//   tt_route_build   send_ids[o*cap + r] = r-th id owned by o (-1 padding), slot_of[i] = slot of the
//                    caller's i-th id (its row comes back in that slot), src_of[slot] = i or -1
//                    (the backward sends gradient row src_of[slot] in that slot)
//   tt_route_localize (owner side) received global ids -> row offsets inside the block, with the
//                    sentinel n_local for padding: what tt_gather_rows (zero rows) and the Adam plan
//                    (skipped run) expect.
// Placement is by sorted position, hence deterministic: the owner sees equal ids of one requester in
// request order, and sums their gradients in that order.
#include "common.hpp"

namespace tt {

__device__ __forceinline__ int64_t lower_bound_i32(const int32_t* __restrict__ a, int64_t n, int64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void route_count_kernel(const int32_t* __restrict__ sorted_ids, int64_t n,
                                                          int64_t rows_per_rank, int32_t world,
                                                          int32_t* __restrict__ starts, int32_t* __restrict__ max_count) {
  int32_t mx = 0;
  for (int32_t o = threadIdx.x; o <= world; o += blockDim.x) {
    const int64_t a = o == world ? n : lower_bound_i32(sorted_ids, n, (int64_t)o * rows_per_rank);
    starts[o] = (int32_t)a;
    if (o < world) {
      const int64_t b = o + 1 == world ? n : lower_bound_i32(sorted_ids, n, (int64_t)(o + 1) * rows_per_rank);
      mx = max(mx, (int32_t)(b - a));
    }
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

__global__ __launch_bounds__(256) void route_build_kernel(const int32_t* __restrict__ sorted_ids,
                                                          const int32_t* __restrict__ perm, int64_t n,
                                                          int64_t rows_per_rank, int32_t world, int64_t cap,
                                                          const int32_t* __restrict__ starts,
                                                          int64_t* __restrict__ send_ids, int64_t* __restrict__ slot_of,
                                                          int64_t* __restrict__ src_of, int32_t* __restrict__ overflow) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int64_t id = sorted_ids[t];
  int64_t o = id / rows_per_rank;
  if (o >= world) o = world - 1;  // cannot happen for ids < n_rows; keeps a flagged (out-of-range) id in range
  const int64_t r = t - starts[o];
  const int64_t i = perm[t];
  if (r >= cap) {  // the caller sized cap from an older count: nothing is sent for this id, and it is reported
    *overflow = 1;
    slot_of[i] = -1;
    return;
  }
  const int64_t slot = o * cap + r;
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

}  // namespace tt

using namespace tt;

extern "C" int tt_route_count(const int32_t* sorted_ids, int64_t n_ids, int64_t rows_per_rank, int32_t world,
                              int32_t* starts, int32_t* max_count, tt_stream_t stream) {
  if (!sorted_ids || !starts || !max_count) return fail_arg("tt_route_count: null pointer");
  if (n_ids <= 0 || rows_per_rank <= 0 || world <= 0) return fail_arg("tt_route_count: sizes");
  route_count_kernel<<<1, 256, 0, S(stream)>>>(sorted_ids, n_ids, rows_per_rank, world, starts, max_count);
  return check_launch("route_count_kernel");
}

extern "C" int tt_route_build(const int32_t* sorted_ids, const int32_t* perm, int64_t n_ids, int64_t rows_per_rank,
                              int32_t world, int64_t cap, const int32_t* starts, int64_t* send_ids, int64_t* slot_of,
                              int64_t* src_of, int32_t* overflow_flag, tt_stream_t stream) {
  if (!sorted_ids || !perm || !starts || !send_ids || !slot_of || !src_of || !overflow_flag)
    return fail_arg("tt_route_build: null pointer");
  if (n_ids <= 0 || rows_per_rank <= 0 || world <= 0 || cap <= 0) return fail_arg("tt_route_build: sizes");
  hipStream_t st = S(stream);
  const int64_t n_slots = (int64_t)world * cap;
  route_fill_kernel<<<(unsigned)ceil_div(n_slots, 256), 256, 0, st>>>(send_ids, src_of, n_slots);
  int rc = check_launch("route_fill_kernel");
  if (rc) return rc;
  route_build_kernel<<<(unsigned)ceil_div(n_ids, 256), 256, 0, st>>>(sorted_ids, perm, n_ids, rows_per_rank, world, cap,
                                                                     starts, send_ids, slot_of, src_of, overflow_flag);
  return check_launch("route_build_kernel");
}

extern "C" int tt_route_localize(const int64_t* ids, int64_t n_ids, int64_t lo, int64_t n_local, int64_t* local,
                                 tt_stream_t stream) {
  if (!ids || !local) return fail_arg("tt_route_localize: null pointer");
  if (n_ids <= 0 || lo < 0 || n_local < 0) return fail_arg("tt_route_localize: sizes");
  route_localize_kernel<<<(unsigned)ceil_div(n_ids, 256), 256, 0, S(stream)>>>(ids, n_ids, lo, n_local, local);
  return check_launch("route_localize_kernel");
}
