// Is v_mfma_f32_32x32x16_f16 symmetric in its operand roles bit for bit?  D1 = A x B and D2 = B^T x A^T (the same scalar products
// and the same k order per output element) on random fp16 data, chained over 8 k-steps like the split-fp16 logits tile.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_role_probe.hip -o /tmp/mfma_role_probe && /tmp/mfma_role_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int brow(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }
// X, Y: [32 rows][128 k] fp16 row-major.  out1[i][j] = sum_k X[i][k] Y[j][k] with X as A; out2 the same with Y as A (stored transposed back)
__global__ void probe(const _Float16* X, const _Float16* Y, float* out1, float* out2, int chain) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 d1, d2;
  for (int e = 0; e < 16; ++e) d1[e] = d2[e] = 0.f;
  for (int c = 0; c < chain; ++c)
    for (int t = 0; t < 8; ++t) {
      f16x8 x = *reinterpret_cast<const f16x8*>(X + r * 128 + 16 * t + 8 * h);
      f16x8 y = *reinterpret_cast<const f16x8*>(Y + r * 128 + 16 * t + 8 * h);
      d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, d1, 0, 0, 0);  // rows = X's rows (registers), columns = Y's rows (lanes)
      d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, d2, 0, 0, 0);  // rows = Y's rows, columns = X's rows
    }
  for (int e = 0; e < 16; ++e) {
    out1[brow(e, h) * 32 + r] = d1[e];  // [i = X row][j = Y row]
    out2[r * 32 + brow(e, h)] = d2[e];  // [i = X row (lane)][j = Y row (register)]
  }
}
int main() {
  static _Float16 hx[32 * 128], hy[32 * 128];
  srand(1);
  for (int i = 0; i < 32 * 128; ++i) { hx[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 200.f); hy[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 200.f); }
  _Float16 *dx, *dy; float *o1, *o2;
  hipMalloc(&dx, sizeof(hx)); hipMalloc(&dy, sizeof(hy)); hipMalloc(&o1, 4096); hipMalloc(&o2, 4096);
  hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice); hipMemcpy(dy, hy, sizeof(hy), hipMemcpyHostToDevice);
  for (int chain = 1; chain <= 3; chain += 2) {
    probe<<<1, 64>>>(dx, dy, o1, o2, chain);
    static float a[1024], b[1024];
    hipMemcpy(a, o1, 4096, hipMemcpyDeviceToHost); hipMemcpy(b, o2, 4096, hipMemcpyDeviceToHost);
    int diff = 0; float worst = 0.f;
    for (int i = 0; i < 1024; ++i) if (memcmp(&a[i], &b[i], 4)) { ++diff; float d = fabsf(a[i] - b[i]) / fabsf(a[i]); if (d > worst) worst = d; }
    printf("chain %d x 8 k-steps: %d of 1024 elements differ between A x B and (B x A)^T, worst relative %.2e (values ~%.3e)\n", chain, diff, worst, a[5]);
  }
  return 0;
}
