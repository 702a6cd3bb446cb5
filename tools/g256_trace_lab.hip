// Lab: where does a barrier interval of the persistent 256-column GEMM go?  Builds the product kernel with
// SF_G256_TRACE (s_memtime stamps around the read segment, both barriers and the MFMA segment, summed per wave row
// of one workgroup) and prints cycles per interval.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSF_G256_TRACE=17
//   -Istreamformer_amd/csrc tools/g256_trace_lab.hip -o /tmp/g256_trace && /tmp/g256_trace [N] [K]
#include "../streamformer_amd/csrc/sf_gemm256.hip"
#include <cstdio>
#include <cstring>
#include <vector>
int sf_wall_clock_ticks(int ns) { return ns / 10; }
template <bool SPLIT, bool WIDE>
static void run(int M, int N, int K, const char* name, int zeros) {
  bf16_t *a, *al, *w, *wl, *o;
  hipMalloc(&a, (size_t)M * K * 2); hipMalloc(&al, (size_t)M * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&wl, (size_t)N * K * 2);
  hipMalloc(&o, (size_t)M * N * 4 * 2);
  std::vector<unsigned short> h((size_t)M * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = zeros ? 0 : (unsigned short)(0x3c00 + (unsigned short)((i * 2654435761u) >> 23));      // finite bf16 noise, or all zero (no toggling: the DVFS check)
  hipMemcpy(a, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(al, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice); hipMemcpy(wl, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  SfGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.a_hi = a; g.a_lo = SPLIT ? al : nullptr; g.w_hi = w; g.w_lo = SPLIT ? wl : nullptr; g.M = M; g.N = N; g.K = K; g.ldc = N;
  g.epi = SPLIT ? SF_EPI_F32 : SF_EPI_BF16; g.out_hi = o; g.out_f32 = (float*)o;
  constexpr int BM = 224;
  const int tiles = ((M + BM - 1) / BM) * (N / 256);
  auto kern = sf_gemm256_kernel<SPLIT ? SF_EPI_F32 : SF_EPI_BF16, false, BM, SPLIT, WIDE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PIECE_BYTES);
  unsigned long long z[16] = {0};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipMemcpyToSymbol(HIP_SYMBOL(g256_trace), z, sizeof(z));
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(G_THREADS), 8 * PIECE_BYTES, 0, g, tiles, 0, 3);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long t[16];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(g256_trace), sizeof(t));
  printf("%s: %.1f us per launch (instrumented)\n", name, ms * 1e3);
  for (int r = 0; r < 2; ++r) {
    const double n = (double)t[r * 8 + 4];
    printf("  wave row %d: %g intervals-pairs; per phase: read+issue+wait %.0f  barrier-1 wait %.0f  mfma issue %.0f  barrier-2 wait %.0f  = %.0f cycles\n", r, n,
           t[r * 8 + 0] / n, t[r * 8 + 1] / n, t[r * 8 + 2] / n, t[r * 8 + 3] / n, (t[r * 8 + 0] + t[r * 8 + 1] + t[r * 8 + 2] + t[r * 8 + 3]) / n);
  }
  hipFree(a); hipFree(al); hipFree(w); hipFree(wl); hipFree(o);
}
int main(int argc, char** argv) {
  const int M = 25088, N = argc > 1 ? atoi(argv[1]) : 2304, K = argc > 2 ? atoi(argv[2]) : 768, zeros = argc > 3 ? atoi(argv[3]) : 0;
  printf("N = %d, K = %d, %s operands\n", N, K, zeros ? "all-zero" : "noise");
  run<false, false>(M, N, K, "bf16 narrow (16 MFMAs per interval)", zeros);
  run<false, true>(M, N, K, "bf16 wide (32 MFMAs per interval)", zeros);
  run<true, false>(M, N, K, "bf16x3 narrow (24 MFMAs per interval)", zeros);
  run<true, true>(M, N, K, "bf16x3 wide (48 MFMAs per interval)", zeros);
  return 0;
}
