// A C++ host of the C ABI (include/streamformer_hip.h) with no Python and no torch in the process:
// reads a weight dump and a clip, runs the encoder forward on the GPU, writes last_hidden_state and
// pooler_output.  What a non-Python caller of the path would write; tests/test_c_host.py runs it against
// the Python mirror on the same files (bit-identical outputs).
//
//   hipcc --offload-arch=gfx950 examples/host_forward.cpp -Iinclude -Lstreamformer_amd -lstreamformer_hip \
//         -Wl,-rpath,'$ORIGIN/../streamformer_amd' -o examples/host_forward
//   examples/host_forward weights.bin clip.bin out.bin [bf16|fp32]
//
// File formats (little endian):
//   weights.bin: int32 config[12], float layer_norm_eps, int32 n_tensors, then per tensor:
//                int32 name_len, name bytes, int32 ndim, int64 shape[ndim], float data[numel]
//   clip.bin:    int32 B, T, H, W, then float pixels[B*T*3*H*W]
//   out.bin:     int32 B, T, N, D, float last_hidden_state[B*T*N*D], float pooler_output[B*T*D]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "streamformer_hip.h"

#define CHECK_SF(expr)                                                                   \
  do {                                                                                   \
    const int rc_ = (expr);                                                              \
    if (rc_ != SF_OK) {                                                                  \
      std::fprintf(stderr, "%s failed (%d): %s\n", #expr, rc_, sf_last_error());         \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)
#define CHECK_HIP(expr)                                                                  \
  do {                                                                                   \
    const hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                              \
      std::fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_));                    \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

template <typename T>
static bool rd(std::FILE* f, T* p, size_t n = 1) { return std::fread(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s weights.bin clip.bin out.bin [bf16|fp32]\n", argv[0]);
    return 2;
  }
  const int compute = (argc > 4 && std::strcmp(argv[4], "fp32") == 0) ? SF_COMPUTE_BF16X3 : SF_COMPUTE_BF16;

  std::FILE* fw = std::fopen(argv[1], "rb");
  if (!fw) { std::perror(argv[1]); return 1; }
  int32_t c[12];
  float eps;
  int32_t n_tensors;
  if (!rd(fw, c, 12) || !rd(fw, &eps) || !rd(fw, &n_tensors)) { std::fprintf(stderr, "short weight file\n"); return 1; }
  sf_config cfg;
  cfg.image_size = c[0]; cfg.patch_size = c[1]; cfg.num_channels = c[2]; cfg.num_frames = c[3];
  cfg.hidden_size = c[4]; cfg.num_hidden_layers = c[5]; cfg.num_attention_heads = c[6]; cfg.intermediate_size = c[7];
  cfg.hidden_act = c[8]; cfg.qkv_bias = c[9]; cfg.enable_causal_temporal = c[10]; cfg.add_lora_spatial = c[11];
  cfg.layer_norm_eps = eps;

  CHECK_HIP(hipSetDevice(0));
  sf_encoder* enc = nullptr;
  CHECK_SF(sf_create(&cfg, 0, &enc));
  for (int i = 0; i < n_tensors; ++i) {
    int32_t name_len, ndim;
    if (!rd(fw, &name_len)) return 1;
    std::string name((size_t)name_len, '\0');
    if (!rd(fw, &name[0], (size_t)name_len) || !rd(fw, &ndim)) return 1;
    std::vector<int64_t> shape((size_t)(ndim > 0 ? ndim : 1), 1);
    size_t numel = 1;
    for (int d = 0; d < ndim; ++d) { if (!rd(fw, &shape[d])) return 1; numel *= (size_t)shape[d]; }
    std::vector<float> data(numel);
    if (!rd(fw, data.data(), numel)) return 1;
    const int rc = sf_load_tensor(enc, name.c_str(), data.data(), SF_F32, shape.data(), ndim);
    if (rc != SF_OK && rc != SF_ERR_UNKNOWN_KEY) { std::fprintf(stderr, "%s: %s\n", name.c_str(), sf_last_error()); return 1; }
  }
  std::fclose(fw);
  CHECK_SF(sf_finalize_weights(enc, compute, /*merge_lora=*/1, /*fuse_temporal_proj=*/1));

  std::FILE* fc = std::fopen(argv[2], "rb");
  if (!fc) { std::perror(argv[2]); return 1; }
  int32_t g[4];
  if (!rd(fc, g, 4)) return 1;
  const int B = g[0], T = g[1], H = g[2], W = g[3];
  const size_t npx = (size_t)B * T * cfg.num_channels * H * W;
  std::vector<float> pixels(npx);
  if (!rd(fc, pixels.data(), npx)) return 1;
  std::fclose(fc);

  const int N = (H / cfg.patch_size) * (W / cfg.patch_size), D = cfg.hidden_size;
  size_t ws_bytes = 0;
  CHECK_SF(sf_workspace_bytes(enc, B, T, H, W, &ws_bytes));
  float *d_px = nullptr, *d_lhs = nullptr, *d_pool = nullptr;
  void* d_ws = nullptr;
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  CHECK_HIP(hipMalloc(&d_px, npx * sizeof(float)));
  CHECK_HIP(hipMalloc(&d_lhs, (size_t)B * T * N * D * sizeof(float)));
  CHECK_HIP(hipMalloc(&d_pool, (size_t)B * T * D * sizeof(float)));
  CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
  CHECK_HIP(hipMemcpyAsync(d_px, pixels.data(), npx * sizeof(float), hipMemcpyHostToDevice, stream));
  CHECK_SF(sf_forward(enc, d_px, SF_F32, B, T, H, W, d_lhs, d_pool, nullptr, nullptr, d_ws, ws_bytes, stream));
  std::vector<float> lhs((size_t)B * T * N * D), pool((size_t)B * T * D);
  CHECK_HIP(hipMemcpyAsync(lhs.data(), d_lhs, lhs.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
  CHECK_HIP(hipMemcpyAsync(pool.data(), d_pool, pool.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
  CHECK_HIP(hipStreamSynchronize(stream));

  std::FILE* fo = std::fopen(argv[3], "wb");
  if (!fo) { std::perror(argv[3]); return 1; }
  const int32_t dims[4] = {B, T, N, D};
  std::fwrite(dims, sizeof(int32_t), 4, fo);
  std::fwrite(lhs.data(), sizeof(float), lhs.size(), fo);
  std::fwrite(pool.data(), sizeof(float), pool.size(), fo);
  std::fclose(fo);
  double cs = 0.0;
  for (float v : lhs) cs += v;
  std::printf("forward ok: B=%d T=%d N=%d D=%d, workspace %.1f MB, sum(last_hidden_state) = %.6f\n", B, T, N, D, ws_bytes / 1e6, cs);
  (void)hipFree(d_px); (void)hipFree(d_lhs); (void)hipFree(d_pool); (void)hipFree(d_ws);
  sf_destroy(enc);
  return 0;
}
