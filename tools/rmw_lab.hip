// Lab (round 4): what one CU can do on the panel epilogue's memory pattern — read 16 B per lane of two bf16 planes (rows of 768 B per
// 384-column half), write both back — as a function of rows in flight per wave and of how many CUs run it at once.  No MFMA, almost no VALU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rmw_lab.hip -o tools/bin/rmw_lab && tools/bin/rmw_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <int DEPTH>
__global__ __launch_bounds__(512) void rmw_kernel(unsigned short* hi, unsigned short* lo, int rows_per_tile, int ld, int tile_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * tile_stride;
  const int panel = tile >> 1, nh = tile & 1;
  const size_t base = (size_t)panel * rows_per_tile * ld + nh * 384 + lane * 8;
  if (lane >= 48) return;
  const int nrows = (rows_per_tile - wave + 7) / 8;      // rows wave, wave + 8, ...
  for (int r0 = 0; r0 < nrows; r0 += DEPTH) {
    u32x4_t h[DEPTH], l[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const int r = r0 + j;
      if (r < nrows) {
        const size_t o = base + (size_t)(wave + 8 * r) * ld;
        h[j] = *reinterpret_cast<const u32x4_t*>(hi + o);
        l[j] = *reinterpret_cast<const u32x4_t*>(lo + o);
      }
    }
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const int r = r0 + j;
      if (r < nrows) {
        const size_t o = base + (size_t)(wave + 8 * r) * ld;
        *reinterpret_cast<u32x4_t*>(hi + o) = h[j] + l[j];
        *reinterpret_cast<u32x4_t*>(lo + o) = h[j] ^ l[j];
      }
    }
  }
}

template <int DEPTH>
static void run(unsigned short* hi, unsigned short* lo, int nwg, int stride) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 8; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rmw_kernel<DEPTH>, dim3(nwg), dim3(512), 0, 0, hi, lo, 196, 768, stride);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double bytes = (double)nwg * 196 * 768 * 2 * 2;
  printf("  %3d workgroups, %2d rows in flight per wave: %6.1f us  (%.2f TB/s read + write, %.1f B/clk per CU at 2.4 GHz)\n", nwg, DEPTH, best * 1e3,
         bytes / best / 1e9, bytes / nwg / (best * 1e-3 * 2.4e9));
}

int main() {
  const size_t n = (size_t)25088 * 768;
  unsigned short *hi, *lo;
  hipMalloc(&hi, n * 2); hipMalloc(&lo, n * 2);
  hipMemset(hi, 1, n * 2); hipMemset(lo, 2, n * 2);
  for (int nwg : {256, 128, 64, 32}) {
    const int stride = 256 / nwg;
    run<1>(hi, lo, nwg, stride);
    run<2>(hi, lo, nwg, stride);
    run<4>(hi, lo, nwg, stride);
    run<8>(hi, lo, nwg, stride);
    run<13>(hi, lo, nwg, stride);
    run<25>(hi, lo, nwg, stride);
  }
  return 0;
}
