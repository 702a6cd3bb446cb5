// LNF lab: the small-M LayerNorm-folded skinny GEMM against a host double computation of LayerNorm(x) W^T + b.
#include "../streamformer_amd/csrc/sf_common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
static uint16_t f2bf_h(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f_h(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  for (auto sh : std::vector<std::array<int, 3>>{{9, 384, 128}, {196, 2304, 768}, {40, 256, 768}}) {
    const int M = sh[0], N = sh[1], K = sh[2];
    std::vector<float> x((size_t)M * K), w((size_t)N * K), gam(K), bet(K), b(N);
    srand(7);
    auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& v : x) v = rnd() * 2.f + 0.7f;
    for (auto& v : w) v = rnd() * 0.05f;
    for (auto& v : gam) v = 1.f + 0.1f * rnd();
    for (auto& v : bet) v = 0.05f * rnd();
    for (auto& v : b) v = 0.1f * rnd();
    std::vector<uint16_t> xb(x.size()), wfb(w.size());
    std::vector<float> bf(N), sn(N);
    for (size_t i = 0; i < x.size(); ++i) xb[i] = f2bf_h(x[i]);
    for (int n = 0; n < N; ++n) {
      double bb = b[n], ss = 0;
      for (int k = 0; k < K; ++k) {
        const float wg = w[(size_t)n * K + k] * gam[k];
        wfb[(size_t)n * K + k] = f2bf_h(wg);
        ss += bf2f_h(wfb[(size_t)n * K + k]);
        bb += (double)w[(size_t)n * K + k] * bet[k];
      }
      bf[n] = (float)bb; sn[n] = (float)ss;
    }
    bf16_t *dx, *dw, *dout; float *db, *ds;
    CK(hipMalloc(&dx, xb.size() * 2)); CK(hipMalloc(&dw, wfb.size() * 2)); CK(hipMalloc(&dout, (size_t)M * N * 2)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&ds, N * 4));
    CK(hipMemcpy(dx, xb.data(), xb.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, wfb.data(), wfb.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, bf.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, sn.data(), N * 4, hipMemcpyHostToDevice));
    SfGemmArgs g; memset(&g, 0, sizeof(g));
    g.a_hi = dx; g.w_hi = dw; g.bias = db; g.M = M; g.N = N; g.K = K; g.epi = SF_EPI_BF16; g.out_hi = dout; g.ldc = N;
    g.ln_inkernel = 1; g.ln_s = ds; g.ln_eps = 1e-6f;
    hipError_t e = sf_launch_gemm_skinny(g, false, 0);
    if (e != hipSuccess) { printf("launch: %s\n", hipGetErrorString(e)); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> out((size_t)M * N);
    CK(hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0, worst_ref = 0;
    for (int m = 0; m < M; ++m) {
      double mu = 0, var = 0;
      for (int k = 0; k < K; ++k) mu += x[(size_t)m * K + k];
      mu /= K;
      for (int k = 0; k < K; ++k) var += (x[(size_t)m * K + k] - mu) * (x[(size_t)m * K + k] - mu);
      var /= K;
      const double rs = 1.0 / sqrt(var + 1e-6);
      for (int n = 0; n < N; n += 7) {
        double y = b[n];
        for (int k = 0; k < K; ++k) y += ((x[(size_t)m * K + k] - mu) * rs * gam[k] + bet[k]) * w[(size_t)n * K + k];
        worst = fmax(worst, fabs(y - bf2f_h(out[(size_t)m * N + n])));
        worst_ref = fmax(worst_ref, fabs(y));
      }
    }
    printf("M=%d N=%d K=%d: max-abs error %.4g (max |y| %.3g)\n", M, N, K, worst, worst_ref);
  }
  return 0;
}
