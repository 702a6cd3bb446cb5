// Lab (round 4): where the waves of the PRODUCT panel kernel (sf_gemm_panel.hip) spend the epilogue of a K = 768 residual projection.
// Builds the kernel with SF_PANEL_TRACE = 1 (stamps only), 3 (no residual loads), 5 (no plane stores), 7 (neither):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSF_PANEL_TRACE=1 -Istreamformer_amd/csrc tools/panel_trace_lab.hip -o tools/bin/panel_trace_1
//   tools/bin/panel_trace_1 [K]
#include "../streamformer_amd/csrc/sf_gemm_panel.hip"
#include <cstdio>
#include <cstring>
#include <vector>
int sf_wall_clock_ticks(int ns) { return (int)((long long)ns / 10); }      // 100 MHz wall clock
int main(int argc, char** argv) {
  const int M = 25088, N = 768, K = argc > 1 ? atoi(argv[1]) : 768;
  bf16_t *a, *w, *rh, *rl; float *st, *bias;
  hipMalloc(&a, (size_t)M * K * 2); hipMalloc(&w, (size_t)N * K * 2);
  hipMalloc(&rh, (size_t)M * N * 2); hipMalloc(&rl, (size_t)M * N * 2);
  hipMalloc(&st, (size_t)M * 8 * 4); hipMalloc(&bias, N * 4);
  std::vector<unsigned short> h((size_t)M * (K > N ? K : N));
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (unsigned short)((i * 2654435761u) >> 23));
  hipMemcpy(a, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice); hipMemcpy(w, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(rh, h.data(), (size_t)M * N * 2, hipMemcpyHostToDevice); hipMemcpy(rl, h.data(), (size_t)M * N * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, N * 4); hipMemset(st, 0, (size_t)M * 8 * 4);
  SfGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.a_hi = a; g.w_hi = w; g.bias = bias; g.M = M; g.N = N; g.K = K; g.ldc = N; g.epi = SF_EPI_RESID_F32; g.alpha = 1.f;
  g.resid_hi = rh; g.resid_lo = rl; g.out_hi = rh; g.out_lo = rl; g.ln_stats_out = st; g.ln_stats_wide = 1;      // in place, as in the forward
  unsigned long long z[16] = {0};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0, best = 1e9f;
  for (int it = 0; it < 6; ++it) {
    hipMemcpyToSymbol(HIP_SYMBOL(panel_trace), z, sizeof(z));
    hipEventRecord(e0, 0);
    if (sf_launch_gemm_panel(g, 0) != hipSuccess) { printf("launch failed\n"); return 1; }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  unsigned long long t[16];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(panel_trace), sizeof(t));
  const double n = (double)t[7];
  printf("SF_PANEL_TRACE=%d K=%d: %.1f us per launch (instrumented, best of 6; last %.1f), %g waves\n", (int)SF_PANEL_TRACE, K, best * 1e3, ms * 1e3, n);
  printf("  cycles per wave (100 MHz s_memtime ticks x 24 at 2.4 GHz are NOT used: s_memtime counts shader clocks): main loop %.0f | per tile, 4 row groups: "
         "loads issued + staged %.0f, barrier %.0f, row loop %.0f, closing barrier %.0f | whole tile %.0f\n",
         t[0] / n, t[1] / n, t[2] / n, t[3] / n, t[4] / n, t[6] / n);
  return 0;
}
