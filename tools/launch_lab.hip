// Launch lab: marginal cost of one dependent kernel inside a hipGraph replay, by kernel "weight".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
struct Big { int pad[256]; };
__global__ void k_empty(float* o) { if (threadIdx.x == 9999) o[0] = 1; }
__global__ void k_bigarg(float* o, Big b) { if (threadIdx.x == 9999) o[0] = b.pad[5]; }
__global__ void k_rw(const float* __restrict__ in, float* __restrict__ o, int n) {   // n floats: read + write, one float4 per thread
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 4 < n) reinterpret_cast<float4*>(o)[i] = reinterpret_cast<const float4*>(in)[i];
}
int main() {
  float *a, *b; CK(hipMalloc(&a, 64 << 20)); CK(hipMalloc(&b, 64 << 20)); CK(hipMemset(a, 0, 64 << 20));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  Big big = {};
  auto time_graph = [&](const char* name, auto launch) {
    const int NK = 100;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NK; ++i) launch(i);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %6.2f us per kernel (graph of %d)\n", name, ms * 1e3 / 20 / NK, NK);
    // eager back-to-back
    for (int i = 0; i < NK; ++i) launch(i);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 5; ++r) for (int i = 0; i < NK; ++i) launch(i);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %6.2f us per kernel (eager)\n", "", ms * 1e3 / 5 / NK);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  };
  time_graph("empty 1 WG x 64", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, b); });
  time_graph("empty 256 WG x 256", [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, b); });
  time_graph("empty 1024 WG x 256", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, b); });
  time_graph("empty 256 WG x 256, 1 KB kernarg", [&](int) { hipLaunchKernelGGL(k_bigarg, dim3(256), dim3(256), 0, s, b, big); });
  time_graph("empty 168 WG x 1024 (128 KB LDS)", [&](int) { hipLaunchKernelGGL(k_empty, dim3(168), dim3(1024), 0, s, b); });
  for (int kb : {64, 600, 2400, 9600}) {
    char nm[64]; snprintf(nm, 64, "copy %d KB (read+write), ping-pong", kb);
    const int n = kb * 256;
    time_graph(nm, [&](int i) { hipLaunchKernelGGL(k_rw, dim3((n / 4 + 255) / 256), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n); });
  }
  return 0;
}
