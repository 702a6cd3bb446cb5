// Lab (VERDICT r2 #4a): the barrier interval of the persistent 256-column GEMM (sf_gemm256.hip) with its 16 x
// v_mfma_f32_16x16x32_bf16 per phase against the same work as 8 x v_mfma_f32_32x32x16_bf16 — same fragment reads
// (12 ds_read_b128 per phase from the product's XOR-swizzled piece images), same accumulator count (32 VGPRs per
// quadrant), same two-barrier phase with the wave rows staggered by one barrier, s_setprio around the MFMA segment.
// Operands are LDS-resident (no DMA, no HBM): this isolates instruction issue + LDS fragment reads, which is what the
// instruction shape can change.  Prints cycles per phase (s_memtime), wall time and the effective shader clock.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_shape_lab.hip -o tools/bin/mfma_shape_lab && tools/bin/mfma_shape_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
#define PIECE 16384

__device__ __forceinline__ bf16x8_t rd16(const char* piece, int row, int kc) {      // [128 rows][64 k], 16-byte slot XOR (row >> 1) & 7
  return *reinterpret_cast<const bf16x8_t*>(piece + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
}

// SHAPE 0: 16x16x32 (product), 1: 32x32x16.  Per phase and wave: one 64 x 32 quadrant over K = 64.
template <int SHAPE>
__global__ __launch_bounds__(512) void lab_kernel(const unsigned short* src, float* out, unsigned long long* cyc, int phases) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  for (int i = tid; i < 8 * PIECE / 2; i += 512) reinterpret_cast<unsigned short*>(smem)[i] = src[(blockIdx.x * 977 + i) & 0xfffff];
  __syncthreads();
  f32x4_t a16[8];
  f32x16_t a32[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) a16[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) a32[i][j] = 0.f;
  if (wm == 1) __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int ph = 0; ph < phases; ++ph) {
    const char* pa = smem + ((ph & 3) * 2 + 1) * PIECE;
    const char* pb = smem + ((ph & 3) * 2) * PIECE;
    bf16x8_t af[8], bf[4];
    if (SHAPE == 0) {
      const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bf[nt * 2 + ks] = rd16(pb, wn * 32 + nt * 16 + l15, ks * 4 + g);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[mt * 2 + ks] = rd16(pa, wm * 64 + mt * 16 + l15, ks * 4 + g);
    } else {
      const int l31 = lane & 31, h = lane >> 5;          // 32x32x16: lane = (row l31, k half h): 8 k values of a 16-k step
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bf[ks] = rd16(pb, wn * 32 + l31, ks * 2 + h);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) af[mt * 4 + ks] = rd16(pa, wm * 64 + mt * 32 + l31, ks * 2 + h);
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
    if (SHAPE == 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            a16[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, bf[nt * 2 + ks]), __builtin_bit_cast(v8bf, af[mt * 2 + ks]), a16[mt * 2 + nt], 0, 0, 0);
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          a32[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, bf[ks]), __builtin_bit_cast(v8bf, af[mt * 4 + ks]), a32[mt], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (wm == 0) __builtin_amdgcn_s_barrier();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a16[i][0] + a16[i][1] + a16[i][2] + a16[i][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += a32[i][j];
  out[blockIdx.x * 512 + tid] = s;
  if (lane == 0 && wn == 0 && blockIdx.x == 17) cyc[wm] = t1 - t0;
}

template <int SHAPE>
static void run(const char* name, const unsigned short* src, int phases, int zeros) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_kernel<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PIECE);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(lab_kernel<SHAPE>, dim3(256), dim3(512), 8 * PIECE, 0, src, out, cyc, phases);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long c[2];
  hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
  const double flop = 256.0 * 8 * phases * 2.0 * 64 * 32 * 64;
  printf("%-34s %s: %7.1f us  %6.0f TFLOP/s  cycles/phase %6.1f (row 0) %6.1f (row 1)  effective clock %.2f GHz\n", name, zeros ? "zeros" : "noise",
         ms * 1e3, flop / ms / 1e9, (double)c[0] / phases, (double)c[1] / phases, (double)c[0] / (ms * 1e6));
  hipFree(out); hipFree(cyc);
}

int main(int argc, char** argv) {
  const int phases = argc > 1 ? atoi(argv[1]) : 20000;
  for (int zeros = 0; zeros < 2; ++zeros) {
    std::vector<unsigned short> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = zeros ? 0 : (unsigned short)(0x3c00 + (unsigned short)((i * 2654435761u) >> 23));
    unsigned short* src; hipMalloc(&src, h.size() * 2);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<0>("16 x mfma_f32_16x16x32_bf16 / phase", src, phases, zeros);
    run<1>(" 8 x mfma_f32_32x32x16_bf16 / phase", src, phases, zeros);
    hipFree(src);
  }
  return 0;
}
