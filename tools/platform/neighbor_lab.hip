// Platform lab for DESIGN.md 4 "Device sharing" (no library code): a synthetic VICTIM and synthetic NEIGHBOURS, to be run as two
// processes on one device (and, with `both`, as two streams of one process).  Prepared at the end of round 5; not yet run.
//
//   victim      workgroups of 256 threads and 100 KB of LDS that repeat the probe kernel's accumulate pattern: a [16 x 772] fp32 image
//               in LDS, per-lane ds_read_b128 of the thread's own 4 columns, broadcast ds_read_b128 of 12 x 16 weights, 12 f32x4
//               accumulators per thread (packed FMAs).  Every workgroup computes the same numbers; the host compares all of them with
//               workgroup 0 of the same launch and prints which (workgroup, thread, accumulator) entries differ.
//   neighbour k workgroups of 256 threads and 41 KB of LDS (they fit beside a victim workgroup), looping one instruction mix:
//               0 = LDS row fragments -> v_mfma_f32_16x16x32_bf16 with AccVGPR accumulators (what phases B / C of the temporal
//                   attention backward do), 1 = the same MFMAs with arch-VGPR accumulators, 2 = the LDS reads alone,
//               3 = MFMAs on register operands (no LDS traffic), 4 = exp2 / FMA VALU work alone
//   both k      victim and neighbour k on two streams of THIS process (does the effect need two address spaces?)
//
//   hipcc --offload-arch=gfx950 -O3 tools/platform/neighbor_lab.hip -o tools/platform/neighbor_lab
//   tools/platform/neighbor_lab victim 20 & tools/platform/neighbor_lab neighbour 0 25 ; wait
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int V_ROWS = 16, V_D = 768, V_XP = V_D + 4, V_NH = 12;
constexpr int V_LDS = (V_ROWS * V_XP + 16 * V_ROWS) * 4 + 40 * 1024;      // image + weights + ballast up to ~100 KB like the probe kernel

__global__ __launch_bounds__(256) void victim_kernel(float* out, int trips) {
  extern __shared__ __attribute__((aligned(16))) float vs[];
  float* xs = vs;                       // [16][772]
  float* wts = vs + V_ROWS * V_XP;      // [16 heads][16 tokens]
  const int tid = threadIdx.x;
  const bool own = tid * 4 < V_D;
  if (own)
    for (int t = 0; t < V_ROWS; ++t) {
      f32x4_t v;
      for (int j = 0; j < 4; ++j) v[j] = (float)(((t * 131 + tid * 4 + j) * 2654435761u >> 8) & 0xffff) * (1.0f / 65536.0f) - 0.5f;
      *reinterpret_cast<f32x4_t*>(xs + t * V_XP + tid * 4) = v;
    }
  wts[tid] = (float)(((tid * 40503u) >> 4) & 0xfff) * (1.0f / 4096.0f);
  __syncthreads();
  f32x4_t acc[V_NH];
#pragma unroll
  for (int i = 0; i < V_NH; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  if (own)
    for (int trip = 0; trip < trips; ++trip) {
#pragma unroll
      for (int i = 0; i < V_NH; ++i) acc[i] *= 0.96875f;
      int off = tid * 4, woff = 0;
      asm volatile("" : "+v"(off), "+v"(woff));      // opaque per trip: the (loop-invariant) LDS reads stay inside the loop, as ds_read_b128
#pragma unroll
      for (int t4 = 0; t4 < V_ROWS; t4 += 4) {
        f32x4_t cx[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) cx[t] = *reinterpret_cast<const f32x4_t*>(xs + (t4 + t) * V_XP + off);
#pragma unroll
        for (int i = 0; i < V_NH; ++i) {
          const f32x4_t w = *reinterpret_cast<const f32x4_t*>(wts + i * V_ROWS + t4 + woff);
          acc[i] += w[0] * cx[0]; acc[i] += w[1] * cx[1]; acc[i] += w[2] * cx[2]; acc[i] += w[3] * cx[3];
        }
      }
    }
  if (own)
#pragma unroll
    for (int i = 0; i < V_NH; ++i) *reinterpret_cast<f32x4_t*>(out + (((size_t)blockIdx.x * 192 + tid) * V_NH + i) * 4) = acc[i];
}

// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int N_LDS = 41472;            // the temporal attention backward's allocation at 16 frames

template <int MODE>
__global__ __launch_bounds__(256) void neighbour_kernel(float* sink, int trips) {
  extern __shared__ __attribute__((aligned(16))) char ns[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* base = ns + wave * (N_LDS / 4);
  for (int i = lane; i < N_LDS / 4 / 16; i += 64) {
    const i32x4_t v = {0x3f803f80 + i, 0x3f003f00 + lane, 0x3e803e80, 0x3f803f00};
    *reinterpret_cast<i32x4_t*>(base + i * 16) = v;
  }
  __builtin_amdgcn_wave_barrier();
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  i32x4_t ra = {0x3f803f80, 0x3f003f00 + lane, 0x3e803e80, 0x3f803f00}, rb = {0x3f003f00, 0x3f803f80, 0x3f003e80 + lane, 0x3e803f00};
  float e = 0.001f * lane;
  for (int trip = 0; trip < trips; ++trip) {
    const int o = ((trip * 7 + lane) & 127) * 16 + ((lane >> 4) << 11);
    if (MODE == 0 || MODE == 1 || MODE == 2) {
      ra = *reinterpret_cast<const i32x4_t*>(base + (o % (N_LDS / 4 - 16) & ~15));
      rb = *reinterpret_cast<const i32x4_t*>(base + ((o + 4096) % (N_LDS / 4 - 16) & ~15));
    }
    if (MODE == 0) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %3, %2, %1" : "+a"(acc0), "+a"(acc1) : "v"(ra), "v"(rb));
    } else if (MODE == 1 || MODE == 3) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %3, %2, %1" : "+v"(acc0), "+v"(acc1) : "v"(ra), "v"(rb));
    } else if (MODE == 2) {
      e += __builtin_bit_cast(float, ra[0] ^ rb[1]) * 1e-30f;
    } else {
      e = __builtin_amdgcn_exp2f(e * 0.5f - 1.0f) + e * 0.25f;
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the inline MFMAs are invisible to the compiler's hazard recognizer
  float r = e;
  if (MODE == 0) {
    f32x4_t t0, t1;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t0[0]) : "a"(acc0[0]));
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t1[0]) : "a"(acc1[0]));
    r += t0[0] + t1[0];
  } else if (MODE == 1 || MODE == 3) {
    r += acc0[0] + acc1[0];
  }
  if (r == 123.456f) sink[tid] = r;     // keeps the work alive, never true in practice
}

static void launch_neighbour(int mode, float* sink, int trips, hipStream_t s) {
  const dim3 g(4096), b(256);
  switch (mode) {
    case 0: hipLaunchKernelGGL(neighbour_kernel<0>, g, b, N_LDS, s, sink, trips); break;
    case 1: hipLaunchKernelGGL(neighbour_kernel<1>, g, b, N_LDS, s, sink, trips); break;
    case 2: hipLaunchKernelGGL(neighbour_kernel<2>, g, b, N_LDS, s, sink, trips); break;
    case 3: hipLaunchKernelGGL(neighbour_kernel<3>, g, b, N_LDS, s, sink, trips); break;
    default: hipLaunchKernelGGL(neighbour_kernel<4>, g, b, N_LDS, s, sink, trips); break;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
struct Victim {
  float* out = nullptr;
  int wgs = 256;
  size_t n = 0;
  std::vector<float> host;
  long launches = 0, bad_launches = 0, bad_entries = 0;
  int shown = 0;
  void init() {
    n = (size_t)wgs * 192 * V_NH * 4;
    CHECK(hipMalloc(&out, n * sizeof(float)));
    host.resize(n);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&victim_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  void launch(hipStream_t s, int lds) { hipLaunchKernelGGL(victim_kernel, dim3(wgs), dim3(256), lds, s, out, 64); }
  void check() {
    CHECK(hipMemcpy(host.data(), out, n * sizeof(float), hipMemcpyDeviceToHost));
    const size_t per = (size_t)192 * V_NH * 4;
    long bad = 0;
    for (int w = 1; w < wgs; ++w)
      for (size_t k = 0; k < per; ++k)
        if (memcmp(&host[w * per + k], &host[k], 4) != 0) {
          ++bad;
          if (shown < 24) {
            ++shown;
            const int tid = (int)(k / (V_NH * 4)), i = (int)((k / 4) % V_NH), j = (int)(k % 4);
            printf("  launch %ld workgroup %d thread %d (wave %d lane %d) acc[%d][%d]: %.9g against %.9g\n", launches, w, tid, tid >> 6, tid & 63, i, j,
                   host[w * per + k], host[k]);
          }
        }
    ++launches;
    if (bad) { ++bad_launches; bad_entries += bad; }
  }
};

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: neighbor_lab victim [seconds] [lds_bytes] | neighbour <mode> [seconds] | both <mode> [seconds]\n"); return 1; }
  const std::string what = argv[1];
  const auto t0 = std::chrono::steady_clock::now();
  auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  if (what == "victim") {
    const double seconds = argc > 2 ? atof(argv[2]) : 20.0;
    const int lds = argc > 3 ? atoi(argv[3]) : V_LDS;
    Victim v; v.init();
    while (elapsed() < seconds) { v.launch(0, lds); CHECK(hipDeviceSynchronize()); v.check(); }
    printf("victim (%d bytes of LDS): %ld launches of %d workgroups, %ld launches with differing workgroups, %ld differing values\n", lds, v.launches, v.wgs,
           v.bad_launches, v.bad_entries);
  } else if (what == "neighbour") {
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const double seconds = argc > 3 ? atof(argv[3]) : 25.0;
    float* sink; CHECK(hipMalloc(&sink, 4096));
    long n = 0;
    while (elapsed() < seconds) { for (int i = 0; i < 8; ++i) launch_neighbour(mode, sink, 2000, 0); CHECK(hipDeviceSynchronize()); n += 8; }
    printf("neighbour mode %d: %ld launches\n", mode, n);
  } else if (what == "both") {
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const double seconds = argc > 3 ? atof(argv[3]) : 20.0;
    hipStream_t sv, sn; CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sn, hipStreamNonBlocking));
    float* sink; CHECK(hipMalloc(&sink, 4096));
    Victim v; v.init();
    while (elapsed() < seconds) {
      for (int i = 0; i < 4; ++i) launch_neighbour(mode, sink, 2000, sn);
      v.launch(sv, V_LDS);
      CHECK(hipStreamSynchronize(sv)); v.check();
      CHECK(hipStreamSynchronize(sn));
    }
    printf("both (one process, two streams), neighbour mode %d: %ld victim launches, %ld with differing workgroups, %ld differing values\n", mode, v.launches,
           v.bad_launches, v.bad_entries);
  }
  return 0;
}
