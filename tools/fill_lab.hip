// Fill-rate lab: how fast can CUs pull L2-resident data (the re-read operand tiles of the small-M GEMMs)?
//   mode 0: buffer/global_load ... lds (LDS-DMA), 16 B per lane, ring of 64 KB
//   mode 1: global_load_dwordx4 into VGPRs (consumed by an xor chain)
// footprint F bytes (all CUs sweep the SAME region: it lives in every XCD's L2 after the first touch), R sweeps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__global__ __launch_bounds__(256) void k_dma(const char* src, size_t bytes_per_wg, int wg_stride_bytes, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 64 KB ring
  const int tid = threadIdx.x, wave = tid >> 6;
  const char* base = src + (size_t)blockIdx.x * wg_stride_bytes;
  const int chunks = (int)(bytes_per_wg / 4096);       // 4 KB per WG-instruction (256 lanes x 16 B)
  for (int c = 0; c < chunks; ++c) {
    const char* g = base + (size_t)c * 4096 + tid * 16;
    char* d = smem + (c & 15) * 4096 + wave * 1024;
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)d, 16, 0, 0);
    if ((c & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned*>(smem + 64);
}

__global__ __launch_bounds__(256) void k_reg(const char* src, size_t bytes_per_wg, int wg_stride_bytes, unsigned* sink) {
  const int tid = threadIdx.x;
  const char* base = src + (size_t)blockIdx.x * wg_stride_bytes;
  const int chunks = (int)(bytes_per_wg / 4096);
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll 8
  for (int c = 0; c < chunks; ++c) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(base + (size_t)c * 4096 + tid * 16);
    acc ^= v;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = 1;
}

int main(int argc, char** argv) {
  const size_t foot = 8u << 20;
  char* src; unsigned* sink;
  CK(hipMalloc(&src, foot + (1 << 20))); CK(hipMemset(src, 1, foot + (1 << 20))); CK(hipMalloc(&sink, 1 << 16));
  CK(hipFuncSetAttribute((const void*)k_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-6s %6s %10s %10s %8s %10s %12s\n", "mode", "WGs", "KB/WG", "stride", "us", "TB/s", "GB/s/WG");
  for (int mode = 0; mode < 2; ++mode)
    for (int wgs : {64, 128, 256, 512, 1024})
      for (size_t per : {(size_t)98304, (size_t)196608, (size_t)393216})
        for (int stride : {0, 4096}) {        // 0: every WG reads the same bytes; 4096: overlapping windows (A-tile like)
          auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(k_dma, dim3(wgs), dim3(256), 65536, 0, src, per, stride, sink);
            else hipLaunchKernelGGL(k_reg, dim3(wgs), dim3(256), 0, 0, src, per, stride, sink);
          };
          for (int i = 0; i < 3; ++i) launch();
          CK(hipEventRecord(e0, 0));
          const int it = 50;
          for (int i = 0; i < it; ++i) launch();
          CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          const double us = ms * 1e3 / it;
          printf("%-6s %6d %10zu %10d %8.2f %10.2f %12.1f\n", mode ? "reg" : "dma", wgs, per / 1024, stride, us, wgs * (double)per / us / 1e6, per / us / 1e3);
        }
  return 0;
}
