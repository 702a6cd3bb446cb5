// Lab (round 4): where the waves of the role-split panel kernel (sf_gemm_pipe.hip) spend their cycles.  Builds the product kernel with
// SF_PIPE_TRACE (s_memtime stamps per role, summed over the workgroups) on synthetic operands:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSF_PIPE_TRACE -Istreamformer_amd/csrc tools/pipe_trace_lab.hip -o tools/bin/pipe_trace
//   tools/bin/pipe_trace [K]
#include "../streamformer_amd/csrc/sf_gemm_pipe.hip"
#include <cstdio>
#include <cstring>
#include <vector>
int main(int argc, char** argv) {
  const int M = 25088, N = 768, K = argc > 1 ? atoi(argv[1]) : 768;
  bf16_t *a, *w, *rh, *rl, *oh, *ol; float *st, *bias;
  hipMalloc(&a, (size_t)M * K * 2); hipMalloc(&w, (size_t)N * K * 2);
  hipMalloc(&rh, (size_t)M * N * 2); hipMalloc(&rl, (size_t)M * N * 2); hipMalloc(&oh, (size_t)M * N * 2); hipMalloc(&ol, (size_t)M * N * 2);
  hipMalloc(&st, (size_t)M * 8 * 4); hipMalloc(&bias, N * 4);
  std::vector<unsigned short> h((size_t)M * (K > N ? K : N));
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (unsigned short)((i * 2654435761u) >> 23));
  hipMemcpy(a, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice); hipMemcpy(w, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(rh, h.data(), (size_t)M * N * 2, hipMemcpyHostToDevice); hipMemcpy(rl, h.data(), (size_t)M * N * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, N * 4);
  SfGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.a_hi = a; g.w_hi = w; g.bias = bias; g.M = M; g.N = N; g.K = K; g.ldc = N; g.epi = SF_EPI_RESID_F32; g.alpha = 1.f;
  g.resid_hi = rh; g.resid_lo = rl; g.out_hi = oh; g.out_lo = ol; g.ln_stats_out = st; g.ln_stats_wide = 1;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PI_LDS_BYTES);
  int* fail; hipMalloc(&fail, 4); hipMemset(fail, 0, 4);
  unsigned long long z[32] = {0};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int it = 0; it < 4; ++it) {
    hipMemcpyToSymbol(HIP_SYMBOL(pipe_trace), z, sizeof(z));
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(sf_gemm_pipe_kernel, dim3(256), dim3(PI_THREADS), PI_LDS_BYTES, 0, g, 98, 256, fail);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long t[32];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(pipe_trace), sizeof(t));
  int hf = 0; hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
  const int steps = 4 * (K / 32);
  printf("K = %d: %.1f us per launch (instrumented), %d steps per workgroup, spin failures %d\n", K, ms * 1e3, steps, hf);
  auto row = [&](const char* name, int b, const char* n0, const char* n1, const char* n2, const char* n3, double per) {
    const double n = (double)t[b + 5];
    if (n == 0) return;
    printf("  %-9s total %8.0f cycles | %s %6.0f  %s %6.0f  %s %6.0f  %s %6.0f   (per %s)\n", name, t[b + 4] / n, n0, t[b] / n / per, n1, t[b + 1] / n / per, n2,
           t[b + 2] / n / per, n3, t[b + 3] / n / per, per == 1 ? "launch" : "step / tile");
  };
  row("mfma", 0, "poll+issue reads", "lgkm wait+publish", "mfma issue", "dump (per tile /steps)", steps);
  row("A loader", 8, "wait landed", "poll consumed", "issue", "-", steps);
  row("W loader", 16, "wait landed", "poll consumed", "issue", "-", steps);
  row("storer", 24, "wait staged", "epilogue", "request", "-", 4);
  return 0;
}
