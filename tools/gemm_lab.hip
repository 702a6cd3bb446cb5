// GEMM lab: times the 128^2 (v1) and 256^2 8-phase kernels on the encoder's shapes with random bf16
// data and checks the 256^2 result against v1.   Build: see tools/run_gemm_lab.sh
#include "../streamformer_amd/csrc/sf_common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

static uint16_t f2bf_h(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
  struct Shape { int M, N, K, epi; const char* name; };
  std::vector<Shape> shapes = {
    {25088, 3072, 768, SF_EPI_ACT_BF16, "mlp_up"}, {25088, 768, 3072, SF_EPI_RESID_F32, "mlp_down"},
    {25088, 2304, 768, SF_EPI_BF16, "qkv"}, {25088, 768, 768, SF_EPI_RESID_F32, "out_proj"},
    {25088, 1536, 768, SF_EPI_BF16, "head_kv"}, {4000, 768, 256, SF_EPI_F32, "ragged"},
    {8192, 8192, 8192, SF_EPI_BF16, "sq8k"},
    {12544, 768, 768, SF_EPI_RESID_F32, "out_B4"}, {23000, 768, 3072, SF_EPI_RESID_F32, "down_ragged"},
  };
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  for (auto& sh : shapes) {
    const size_t na = (size_t)sh.M * sh.K, nw = (size_t)sh.N * sh.K, nc = (size_t)sh.M * sh.N;
    std::vector<uint16_t> ha(na), hw(nw);
    std::vector<float> hb(sh.N), hr(nc);
    srand(1234);
    for (auto& v : ha) v = f2bf_h((float)rand() / RAND_MAX * 2.f - 1.f);
    for (auto& v : hw) v = f2bf_h(((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f);
    for (auto& v : hb) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : hr) v = (float)rand() / RAND_MAX - 0.5f;
    bf16_t *da, *dw, *oh1, *oh2; float *db, *dr, *of1, *of2;
    CK(hipMalloc(&da, na * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&db, sh.N * 4)); CK(hipMalloc(&dr, nc * 4));
    CK(hipMalloc(&oh1, nc * 2)); CK(hipMalloc(&oh2, nc * 2)); CK(hipMalloc(&of1, nc * 4)); CK(hipMalloc(&of2, nc * 4));
    CK(hipMemcpy(da, ha.data(), na * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), sh.N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr.data(), nc * 4, hipMemcpyHostToDevice));
    CK(hipMemset(oh1, 0, nc * 2)); CK(hipMemset(oh2, 0xff, nc * 2)); CK(hipMemset(of1, 0, nc * 4)); CK(hipMemset(of2, 0xff, nc * 4));
    SfGemmArgs g; memset(&g, 0, sizeof(g));
    g.a_hi = da; g.w_hi = dw; g.bias = db; g.M = sh.M; g.N = sh.N; g.K = sh.K; g.epi = sh.epi < 0 ? -sh.epi : sh.epi; g.act = sh.epi < 0 ? 99 : 0; g.alpha = 0.7f; g.resid = dr; g.ldc = sh.N;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms1 = 0, ms2 = 0;
    g.out_f32 = of1; g.out_hi = oh1;
    CK(sf_launch_gemm128(g, false, 0));
    CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) CK(sf_launch_gemm128(g, false, 0)); CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms1, e0, e1));
    double maxd = -1;
    const bool use_panel = sf_gemm_panel_supported(g, false);
    if (use_panel || sf_gemm256_supported(g, false)) {
      g.out_f32 = of2; g.out_hi = oh2;
      CK(use_panel ? sf_launch_gemm_panel(g, 0) : sf_launch_gemm256(g, 0));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) CK(use_panel ? sf_launch_gemm_panel(g, 0) : sf_launch_gemm256(g, 0)); CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms2, e0, e1));
      const bool f32out = sh.epi == SF_EPI_F32 || sh.epi == SF_EPI_RESID_F32;
      maxd = 0;
      size_t bad = 0;
      if (f32out) {
        std::vector<float> r1(nc), r2(nc);
        CK(hipMemcpy(r1.data(), of1, nc * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), of2, nc * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < nc; ++i) { double d = fabs((double)r1[i] - r2[i]); if (!(d <= 1e-3)) { if (bad < 5) printf("   mismatch at row %zu col %zu: %g vs %g\n", i / sh.N, i % sh.N, r1[i], r2[i]); ++bad; } if (d > maxd) maxd = d; }
      } else {
        std::vector<uint16_t> r1(nc), r2(nc);
        CK(hipMemcpy(r1.data(), oh1, nc * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), oh2, nc * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < nc; ++i) { uint32_t a = (uint32_t)r1[i] << 16, b = (uint32_t)r2[i] << 16; float fa, fb; memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
          double d = fabs((double)fa - fb); if (!(d <= 2e-2)) { if (bad < 5) printf("   mismatch at row %zu col %zu: %g vs %g\n", i / sh.N, i % sh.N, fa, fb); ++bad; } if (d > maxd) maxd = d; }
      }
      if (bad) printf("   %zu mismatching elements\n", bad);
    }
    const double fl = 2.0 * sh.M * sh.N * sh.K;
    printf("%-9s M=%d N=%d K=%d  v1 %.1f us %.0f TF | g256 %.1f us %.0f TF  maxdiff %.3g\n", sh.name, sh.M, sh.N, sh.K,
           ms1 / iters * 1e3, fl / (ms1 / iters) / 1e9, ms2 / iters * 1e3, ms2 > 0 ? fl / (ms2 / iters) / 1e9 : 0.0, maxd);
    hipFree(da); hipFree(dw); hipFree(db); hipFree(dr); hipFree(oh1); hipFree(oh2); hipFree(of1); hipFree(of2);
  }
  return 0;
}
