// C-ABI implementation: handle, weight packing, and the launch schedule of the encoder forward
// (reference TimesformerMultiTaskingModelSigLIP.forward, modeling:1299-1354; layer body :934-1004;
// streaming copy vqa_enc:1316-1392).  See include/streamformer_hip.h for the contract.
#include "sf_internal.h"
#include "sf_common.h"
#include "sf_switches.h"
#include "sf_pool_head.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define SF_ABI_VERSION 5
static const int kLoraRank = 32;  // modeling:1280-1281

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
int sf_set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define set_err sf_set_err

// ------------------------------------------------------------------------------------------------
// host-side helpers
// ------------------------------------------------------------------------------------------------
static inline uint16_t h_f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float h_bf2f(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  size_t numel() const { return data.size(); }
};

struct DevLinear {          // y = x W^T + b ; W [N,K]
  bf16_t* w_hi = nullptr;
  bf16_t* w_lo = nullptr;
  float* bias = nullptr;
  float* ln_s = nullptr;    // LN-folded variants only: s_n = sum_k bf16(W'[n,k])
  bf16_t* w_frag = nullptr; // t_qkv_f, bf16 mode: w_hi in MFMA-fragment order for sf_stream_fused.hip (SfStreamQkvArgs::w_frag)
  int N = 0, K = 0;
};
struct DevLN { float* g = nullptr; float* b = nullptr; };
struct DevLayer {
  DevLN ln_t, ln_b, ln_a;
  DevLinear t_qkv, t_out, t_dense, t_fused, s_qkv, s_out, up, down;
  DevLinear t_qkv_f, s_qkv_f, up_f;   // the preceding LayerNorm folded in (W' = W*gamma, b' = b + W beta; hi + lo planes in the accurate mode)
  DevLinear t_qkv_fp;                 // t_qkv_f with its rows permuted for sf_gemm_qkv.hip's fused tile: [q | k | v] of two heads per 384 rows (bf16 mode)
  float gate_tanh = 0.f;
};

struct sf_encoder {
  sf_config cfg;
  int device = 0;
  int N = 0, D = 0, I = 0, L = 0, Kp = 0, hd = 64;      // I, Kp: padded to multiples of 64 (see sf_create)
  std::map<std::string, HostTensor> host;   // staged fp32 copies until finalize
  std::map<std::string, std::vector<int64_t>> expected;
  bool finalized = false;
  int compute = SF_COMPUTE_BF16;
  bool fused_temporal = false;
  std::vector<void*> allocs;
  // device weights
  DevLinear patch;
  float* pos = nullptr;       // [N,D]
  float* time_tab = nullptr;  // [num_frames,D]
  std::vector<DevLayer> layers;
  DevLN post_ln, head_ln;
  DevLinear head_out, head_fc1, head_fc2;
  // pooling head with the k / v projections of the tokens folded away (sf_pool_head.hip): U_h = Wk_h^T q_h as hi + lo bf16
  // planes [16, D] (q = the projected, scaled probe), the value projection and its bias in fp32
  bf16_t* head_u_hi = nullptr; bf16_t* head_u_lo = nullptr;
  float* head_u = nullptr;    // [16, D] fp32
  float* head_wv = nullptr;   // [D, D]
  float* head_bv = nullptr;   // [D]
  size_t weight_bytes = 0;
  uint64_t generation = 0;    // process-unique id of this handle's current weight packing (bumped by every finalize)
  SfPixelNorm pixel_norm = {{1.0f / 127.5f, 1.0f / 127.5f, 1.0f / 127.5f, 1.0f / 127.5f}, {-1.f, -1.f, -1.f, -1.f}};
  // sf_forward_profile: HIP events around the launches of four kernel classes INSIDE a real forward (0: N = 768 residual projection
  // at K = D, 1: the same at K = I, 2: spatial attention, 3: temporal attention)
  struct ProfSpan { int cls; hipEvent_t e0, e1; };
  bool prof_on = false;
  std::vector<ProfSpan> prof;
};
// run `launch` between two events on `s` when the encoder is in profiling mode
template <typename F>
static hipError_t prof_span(const sf_encoder* ce, int cls, hipStream_t s, F&& launch) {
  sf_encoder* e = const_cast<sf_encoder*>(ce);
  if (!e->prof_on) return launch();
  sf_encoder::ProfSpan sp;
  sp.cls = cls;
  hipError_t err = hipEventCreate(&sp.e0);
  if (err != hipSuccess) return err;
  err = hipEventCreate(&sp.e1);
  if (err != hipSuccess) return err;
  (void)hipEventRecord(sp.e0, s);
  err = launch();
  (void)hipEventRecord(sp.e1, s);
  e->prof.push_back(sp);
  return err;
}

struct sf_cache {
  sf_encoder* enc = nullptr;
  uint64_t enc_generation = 0;   // packing of the encoder this cache was sized for (element size, device, weights)
  int B = 0, cap = 0, H = 0, W = 0, N = 0, len = 0;
  bool warmed = false;      // one eager call has run on this cache (lazy kernel set-up done) -> captures may start
  std::vector<void*> qkv;   // per layer [B, cap, N, 3D] (bf16 or fp32 by compute mode)
  size_t bytes = 0;
  // hipGraph of the per-call launch sequence, one per (frames cached, frames added): everything between the patch
  // extraction (reads the caller's pixels) and the copy-out (writes the caller's outputs) is replayed as one graph launch
  struct GraphEntry {
    hipGraphExec_t exec = nullptr;
    void* ws = nullptr;
    const float* pos = nullptr;
    bool pooler = false;
    int pixel_kind = 0;
  };
  std::map<uint32_t, GraphEntry> graphs;
  hipStream_t cap_stream = nullptr;   // captures are recorded on a private stream (the caller's may be the null stream, which
                                      // cannot capture) and replayed on the caller's
  int policy = 0;                     // 0 = stop at capacity (the reference raises there), 1 = sliding window over the last `cap` frames
  SfStreamParams* dparams = nullptr;  // device block {pixels, outputs, position} of the single-frame graph (one graph for all positions)
};

// ------------------------------------------------------------------------------------------------
// expected weights
// ------------------------------------------------------------------------------------------------
static void build_expected(sf_encoder* e) {
  const sf_config& c = e->cfg;
  const int64_t D = c.hidden_size, I = c.intermediate_size, P = c.patch_size, C = c.num_channels;
  const int64_t N = e->N, T = c.num_frames;
  auto& x = e->expected;
  x["embeddings.position_embeddings"] = {1, N, D};
  x["embeddings.time_embeddings"] = {1, T, D};
  x["embeddings.patch_embeddings.projection.weight"] = {D, C, P, P};
  x["embeddings.patch_embeddings.projection.bias"] = {D};
  auto lin = [&](const std::string& p, int64_t out, int64_t in, bool bias) {
    x[p + ".weight"] = {out, in};
    if (bias) x[p + ".bias"] = {out};
  };
  auto ln = [&](const std::string& p) { x[p + ".weight"] = {D}; x[p + ".bias"] = {D}; };
  for (int i = 0; i < c.num_hidden_layers; ++i) {
    const std::string p = "encoder.layer." + std::to_string(i) + ".";
    x[p + "temporal_attention_gating"] = {};
    ln(p + "temporal_layernorm");
    lin(p + "temporal_attention.attention.qkv", 3 * D, D, c.qkv_bias);
    lin(p + "temporal_attention.output.dense", D, D, true);
    lin(p + "temporal_dense", D, D, true);
    ln(p + "layernorm_before");
    lin(p + "attention.attention.qkv", 3 * D, D, c.qkv_bias);
    lin(p + "attention.output.dense", D, D, true);
    if (c.add_lora_spatial) {
      x[p + "attention.attention.qkv_lora_a.weight"] = {kLoraRank, D};
      x[p + "attention.attention.qkv_lora_b.weight"] = {3 * D, kLoraRank};
      x[p + "attention.output.dense_lora_a.weight"] = {kLoraRank, D};
      x[p + "attention.output.dense_lora_b.weight"] = {D, kLoraRank};
    }
    ln(p + "layernorm_after");
    lin(p + "intermediate.dense", I, D, true);
    lin(p + "output.dense", D, I, true);
  }
  ln("post_layernorm");
  x["head.probe"] = {1, 1, D};
  x["head.attention.in_proj_weight"] = {3 * D, D};
  x["head.attention.in_proj_bias"] = {3 * D};
  lin("head.attention.out_proj", D, D, true);
  ln("head.layernorm");
  lin("head.mlp.fc1", I, D, true);
  lin("head.mlp.fc2", D, I, true);
}

// ------------------------------------------------------------------------------------------------
// lifetime
// ------------------------------------------------------------------------------------------------
extern "C" int sf_abi_version(void) { return SF_ABI_VERSION; }
extern "C" const char* sf_last_error(void) { return g_err; }

extern "C" int sf_create(const sf_config* cfg, int device, sf_encoder** out) {
  if (!cfg || !out) return set_err(SF_ERR_INVALID, "sf_create: null argument");
  const sf_config& c = *cfg;
  if (c.hidden_size <= 0 || c.num_attention_heads <= 0 || c.hidden_size % c.num_attention_heads)
    return set_err(SF_ERR_INVALID, "hidden_size %d not divisible by heads %d", c.hidden_size, c.num_attention_heads);
  {   // head_dim 64: the tuned attention kernels; any other multiple of 8 up to 128 (SigLIP-so400m: 72): sf_attention_generic.hip
    const int hd = c.hidden_size / c.num_attention_heads;
    if (hd < 8 || hd > 128 || hd % 8)
      return set_err(SF_ERR_INVALID, "head_dim %d unsupported: multiples of 8 from 8 to 128 (64 runs on the tuned kernels, the others on the generic fp32 attention kernel)", hd);
  }
  if (c.num_attention_heads > 16)
    return set_err(SF_ERR_INVALID, "%d attention heads unsupported: the pooling-head kernels hold at most 16 heads per MFMA tile", c.num_attention_heads);
  // intermediate_size and C*P*P need no alignment: the weights are zero-padded to multiples of 64 at upload (gelu(0) = 0 and zero
  // weight columns make the padding exact): SigLIP-so400m has I = 4304 and 14 x 14 patches (K = 588)
  if (c.hidden_size % 64 || c.intermediate_size <= 0 || c.patch_size <= 0 || c.num_channels <= 0)
    return set_err(SF_ERR_INVALID, "hidden_size must be a multiple of 64; intermediate_size, patch_size, num_channels positive");
  if (c.image_size % c.patch_size) return set_err(SF_ERR_INVALID, "image_size not a multiple of patch_size");
  if (c.hidden_act < 0 || c.hidden_act > 2) return set_err(SF_ERR_INVALID, "unsupported hidden_act code %d", c.hidden_act);
  if (c.num_frames <= 0 || c.num_frames > 256) return set_err(SF_ERR_INVALID, "num_frames must be in 1..256");
  sf_encoder* e = new sf_encoder();
  e->cfg = c;
  e->device = device;
  e->D = c.hidden_size;
  e->I = (c.intermediate_size + 63) / 64 * 64;          // padded: rows / columns past cfg.intermediate_size are zero weights
  e->L = c.num_hidden_layers;
  e->N = (c.image_size / c.patch_size) * (c.image_size / c.patch_size);
  e->Kp = (c.num_channels * c.patch_size * c.patch_size + 63) / 64 * 64;      // padded patch-vector length (zero columns)
  e->hd = c.hidden_size / c.num_attention_heads;
  build_expected(e);
  *out = e;
  return SF_OK;
}

static uint64_t next_generation() {
  static uint64_t g = 0;      // a handle is used from one thread at a time; creation from several threads is serialised by the caller
  return ++g;
}

static void free_device(sf_encoder* e) {
  for (void* p : e->allocs) (void)hipFree(p);
  e->allocs.clear();
}

extern "C" void sf_destroy(sf_encoder* e) {
  if (!e) return;
  free_device(e);
  delete e;
}

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
extern "C" int sf_load_tensor(sf_encoder* e, const char* key, const void* host_ptr, int dtype,
                              const int64_t* shape, int ndim) {
  if (!e || !key || !host_ptr || ndim < 0 || (ndim && !shape)) return set_err(SF_ERR_INVALID, "sf_load_tensor: null argument");
  std::string k(key);
  if (k.rfind("timesformer.", 0) == 0) k = k.substr(12);   // wrapper checkpoints (base_model_prefix, modeling:1073)
  auto it = e->expected.find(k);
  if (it == e->expected.end()) return set_err(SF_ERR_UNKNOWN_KEY, "'%s' is not a weight of this model", key);
  size_t n = 1, ne = 1;
  for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
  for (int64_t d : it->second) ne *= (size_t)d;
  bool same = n == ne;
  if (same && (int)it->second.size() == ndim)
    for (int i = 0; i < ndim; ++i) same = same && it->second[i] == shape[i];
  if (!same) return set_err(SF_ERR_INVALID, "'%s': shape mismatch (%zu elements given, %zu expected)", key, n, ne);
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.data.resize(n);
  switch (dtype) {
    case SF_F32: memcpy(t.data.data(), host_ptr, n * 4); break;
    case SF_F64: for (size_t i = 0; i < n; ++i) t.data[i] = (float)((const double*)host_ptr)[i]; break;
    case SF_BF16: for (size_t i = 0; i < n; ++i) t.data[i] = h_bf2f(((const uint16_t*)host_ptr)[i]); break;
    case SF_F16: {
      const uint16_t* p = (const uint16_t*)host_ptr;
      for (size_t i = 0; i < n; ++i) {
        const uint32_t s = (p[i] >> 15) & 1, ex = (p[i] >> 10) & 31, m = p[i] & 1023;
        float v;
        if (ex == 0) v = ldexpf((float)m, -24);
        else if (ex == 31) v = m ? NAN : INFINITY;
        else v = ldexpf((float)(m | 1024), (int)ex - 25);
        t.data[i] = s ? -v : v;
      }
      break;
    }
    default: return set_err(SF_ERR_INVALID, "unknown dtype %d", dtype);
  }
  e->host[k] = std::move(t);
  e->finalized = false;
  return SF_OK;
}

extern "C" int sf_set_pixel_normalization(sf_encoder* e, const float* mean, const float* std, int channels, float rescale) {
  if (!e || !mean || !std || channels < 1 || channels > 4) return set_err(SF_ERR_INVALID, "bad pixel normalisation");
  for (int c = 0; c < 4; ++c) {
    const int k = c < channels ? c : channels - 1;
    if (!(std[k] > 0.f)) return set_err(SF_ERR_INVALID, "image_std must be positive");
    e->pixel_norm.scale[c] = rescale / std[k];       // (x * rescale - mean) / std
    e->pixel_norm.shift[c] = -mean[k] / std[k];
  }
  return SF_OK;
}

extern "C" int sf_missing_weights(sf_encoder* e) {
  if (!e) return set_err(SF_ERR_INVALID, "null handle");
  int missing = 0;
  std::string names;
  for (auto& kv : e->expected)
    if (!e->host.count(kv.first)) {
      ++missing;
      if (names.size() < 800) names += kv.first + " ";
    }
  if (missing) set_err(SF_ERR_STATE, "missing %d weights: %s", missing, names.c_str());
  return missing;
}

template <typename T>
static int dev_upload(sf_encoder* e, const std::vector<T>& h, T** out) {
  void* p = nullptr;
  const size_t bytes = h.size() * sizeof(T);
  HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
  e->allocs.push_back(p);
  if (bytes) HIP_TRY(hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice));
  e->weight_bytes += bytes;
  *out = (T*)p;
  return SF_OK;
}

static int upload_linear(sf_encoder* e, const std::vector<float>& w, const std::vector<float>* bias, int N,
                         int K, DevLinear* out, bool force_split = false) {
  std::vector<uint16_t> hi(w.size()), lo;
  const bool split = force_split || e->compute == SF_COMPUTE_BF16X3;     // force_split: the pooling head's small Linears keep the lo plane in bf16 mode too
  if (split) lo.resize(w.size());
  for (size_t i = 0; i < w.size(); ++i) {
    hi[i] = h_f2bf(w[i]);
    if (split) lo[i] = h_f2bf(w[i] - h_bf2f(hi[i]));
  }
  int rc = dev_upload<uint16_t>(e, hi, &out->w_hi);
  if (rc) return rc;
  if (split && (rc = dev_upload<uint16_t>(e, lo, &out->w_lo))) return rc;
  if (bias && (rc = dev_upload<float>(e, *bias, &out->bias))) return rc;
  out->N = N;
  out->K = K;
  return SF_OK;
}

// LayerNorm(gamma, beta) followed by Linear(W, b)  ==  rstd * (x W'^T - mean * s) + b'
// with W' = W * gamma (per input column), b' = b + W beta, s_n = sum_k W'[n,k] (of the ROUNDED W').
static int upload_folded_linear(sf_encoder* e, const std::vector<float>& w, const std::vector<float>* bias,
                                const std::vector<float>& gamma, const std::vector<float>& beta, int N, int K,
                                DevLinear* out, bool frag_copy = false) {
  std::vector<float> wf(w.size()), bf(N), sn(N);
  const bool split = e->compute == SF_COMPUTE_BF16X3;      // the MFMAs then see hi + lo planes of W'
  for (int n = 0; n < N; ++n) {
    double bb = bias ? (double)(*bias)[n] : 0.0, ss = 0.0;
    for (int k = 0; k < K; ++k) {
      const float wv = w[(size_t)n * K + k];
      const float wg = (float)((double)wv * (double)gamma[k]);
      wf[(size_t)n * K + k] = wg;
      const float hi = h_bf2f(h_f2bf(wg));
      ss += (double)hi + (split ? (double)h_bf2f(h_f2bf(wg - hi)) : 0.0);
      bb += (double)wv * (double)beta[k];
    }
    bf[n] = (float)bb;
    sn[n] = (float)ss;
  }
  int rc = upload_linear(e, wf, &bf, N, K, out);
  if (rc) return rc;
  if (frag_copy && !split && N % 16 == 0 && K % 32 == 0) {
    // fragment-major copy: [n-tile t][k-step j][lane = 16 g + l15][8] = W'[16 t + l15][32 j + 8 g ..]: the 64 lanes of a wave read ONE
    // contiguous KiB per MFMA operand (the row-major matrix gives every 16-lane group 16 different lines of 16 bytes each)
    const int NK = K / 32;
    std::vector<uint16_t> fr((size_t)N * K);
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < K; ++k)
        fr[((((size_t)(n / 16) * NK + k / 32) * 64) + ((k % 32) / 8) * 16 + n % 16) * 8 + k % 8] = h_f2bf(wf[(size_t)n * K + k]);
    if ((rc = dev_upload<uint16_t>(e, fr, &out->w_frag))) return rc;
  }
  return dev_upload<float>(e, sn, &out->ln_s);
}

static int upload_ln(sf_encoder* e, const std::string& p, DevLN* out) {
  int rc = dev_upload<float>(e, e->host[p + ".weight"].data, &out->g);
  if (rc) return rc;
  return dev_upload<float>(e, e->host[p + ".bias"].data, &out->b);
}

// W += B A  (lora_b [out, r] x lora_a [r, in]); accumulate in double
static void merge_lora_into(std::vector<float>& w, const std::vector<float>& a, const std::vector<float>& b,
                            int out, int in, int r) {
  for (int o = 0; o < out; ++o)
    for (int i = 0; i < in; ++i) {
      double acc = 0.0;
      for (int k = 0; k < r; ++k) acc += (double)b[(size_t)o * r + k] * (double)a[(size_t)k * in + i];
      w[(size_t)o * in + i] = (float)((double)w[(size_t)o * in + i] + acc);
    }
}

extern "C" int sf_finalize_weights(sf_encoder* e, int compute, int merge_lora, int fuse_temporal_proj) {
  if (!e) return set_err(SF_ERR_INVALID, "null handle");
  if (compute != SF_COMPUTE_BF16 && compute != SF_COMPUTE_BF16X3) return set_err(SF_ERR_INVALID, "unknown compute mode %d", compute);
  if (sf_missing_weights(e)) return SF_ERR_STATE;
  if (e->cfg.add_lora_spatial && !merge_lora)
    return set_err(SF_ERR_INVALID, "un-merged LoRA execution is a training-time path; the inference forward needs merge_lora=1");
  HIP_TRY(hipSetDevice(e->device));
  free_device(e);
  e->weight_bytes = 0;
  e->compute = compute;
  e->fused_temporal = fuse_temporal_proj != 0;
  const int D = e->D, I = e->I, Ir = e->cfg.intermediate_size;
  auto H = [&](const std::string& k) -> std::vector<float>& { return e->host[k].data; };
  auto Hopt = [&](const std::string& k) -> std::vector<float>* { return e->host.count(k) ? &e->host[k].data : nullptr; };
  // zero padding of a [rows, cols] matrix to [rows_p, cols_p] (intermediate_size -> I, C*P*P -> Kp); a no-op copy when nothing changes
  auto pad2 = [](const std::vector<float>& w, int rows, int cols, int rows_p, int cols_p) {
    if (rows == rows_p && cols == cols_p) return w;
    std::vector<float> o((size_t)rows_p * cols_p, 0.f);
    for (int r = 0; r < rows; ++r) std::copy(w.begin() + (size_t)r * cols, w.begin() + (size_t)(r + 1) * cols, o.begin() + (size_t)r * cols_p);
    return o;
  };
  auto pad1 = [](const std::vector<float>* b, int n, int n_p) {
    std::vector<float> o((size_t)n_p, 0.f);
    if (b) std::copy(b->begin(), b->begin() + n, o.begin());
    return o;
  };
  int rc;
#define TRY(x) do { if ((rc = (x))) return rc; } while (0)
  {
    const int Kr = e->cfg.num_channels * e->cfg.patch_size * e->cfg.patch_size;
    const std::vector<float> wp = pad2(H("embeddings.patch_embeddings.projection.weight"), D, Kr, D, e->Kp);
    TRY(upload_linear(e, wp, Hopt("embeddings.patch_embeddings.projection.bias"), D, e->Kp, &e->patch));
  }
  TRY(dev_upload<float>(e, H("embeddings.position_embeddings"), &e->pos));
  TRY(dev_upload<float>(e, H("embeddings.time_embeddings"), &e->time_tab));
  e->layers.assign(e->L, DevLayer());
  for (int i = 0; i < e->L; ++i) {
    const std::string p = "encoder.layer." + std::to_string(i) + ".";
    DevLayer& l = e->layers[i];
    l.gate_tanh = std::tanh(H(p + "temporal_attention_gating")[0]);
    TRY(upload_ln(e, p + "temporal_layernorm", &l.ln_t));
    TRY(upload_ln(e, p + "layernorm_before", &l.ln_b));
    TRY(upload_ln(e, p + "layernorm_after", &l.ln_a));
    TRY(upload_linear(e, H(p + "temporal_attention.attention.qkv.weight"), Hopt(p + "temporal_attention.attention.qkv.bias"), 3 * D, D, &l.t_qkv));
    TRY(upload_folded_linear(e, H(p + "temporal_attention.attention.qkv.weight"), Hopt(p + "temporal_attention.attention.qkv.bias"),
                             H(p + "temporal_layernorm.weight"), H(p + "temporal_layernorm.bias"), 3 * D, D, &l.t_qkv_f,
                             SF_LAB_SWITCH("SF_STREAM_QKV_FUSE") && e->compute == SF_COMPUTE_BF16 && D % 128 == 0 && D <= 768));
#ifdef SF_LAB      // lab library only: the permuted copy for sf_gemm_qkv.hip's fused tile (profiles/r04_qkv_fused_ab.txt)
    if (e->compute == SF_COMPUTE_BF16 && D % 128 == 0 && SF_LAB_SWITCH("SF_QKV_FUSED")) {
      // row 384 j + 64 i + c of the permuted matrix = row (i / 2) D + (2 j + (i & 1)) 64 + c of the original (i = 0..5: q0 q1 k0 k1 v0 v1)
      const std::vector<float>& w0 = H(p + "temporal_attention.attention.qkv.weight");
      const std::vector<float>* b0 = Hopt(p + "temporal_attention.attention.qkv.bias");
      std::vector<float> wp(w0.size()), bp(3 * (size_t)D, 0.f);
      for (int r = 0; r < 3 * D; ++r) {
        const int j = r / 384, i = (r % 384) / 64, cc = r % 64;
        const int src = (i >> 1) * D + (2 * j + (i & 1)) * 64 + cc;
        std::copy(w0.begin() + (size_t)src * D, w0.begin() + (size_t)(src + 1) * D, wp.begin() + (size_t)r * D);
        if (b0) bp[r] = (*b0)[src];
      }
      TRY(upload_folded_linear(e, wp, b0 ? &bp : nullptr, H(p + "temporal_layernorm.weight"), H(p + "temporal_layernorm.bias"), 3 * D, D, &l.t_qkv_fp));
    }
#endif
    if (e->fused_temporal) {
      // temporal_dense(output.dense(x)) = (W2 W1) x + (W2 b1 + b2)      (modeling:947-954)
      const std::vector<float>& w1 = H(p + "temporal_attention.output.dense.weight");
      const std::vector<float>& b1 = H(p + "temporal_attention.output.dense.bias");
      const std::vector<float>& w2 = H(p + "temporal_dense.weight");
      const std::vector<float>& b2 = H(p + "temporal_dense.bias");
      std::vector<float> wf((size_t)D * D), bf(D);
      std::vector<double> row(D);
      for (int o = 0; o < D; ++o) {
        std::fill(row.begin(), row.end(), 0.0);
        double bb = b2[o];
        for (int k = 0; k < D; ++k) {
          const double a = w2[(size_t)o * D + k];
          bb += a * b1[k];
          const float* w1r = &w1[(size_t)k * D];
          for (int j = 0; j < D; ++j) row[j] += a * w1r[j];
        }
        for (int j = 0; j < D; ++j) wf[(size_t)o * D + j] = (float)row[j];
        bf[o] = (float)bb;
      }
      TRY(upload_linear(e, wf, &bf, D, D, &l.t_fused));
    } else {
      TRY(upload_linear(e, H(p + "temporal_attention.output.dense.weight"), Hopt(p + "temporal_attention.output.dense.bias"), D, D, &l.t_out));
      TRY(upload_linear(e, H(p + "temporal_dense.weight"), Hopt(p + "temporal_dense.bias"), D, D, &l.t_dense));
    }
    std::vector<float> wq = H(p + "attention.attention.qkv.weight");
    std::vector<float> wo = H(p + "attention.output.dense.weight");
    if (e->cfg.add_lora_spatial) {
      merge_lora_into(wq, H(p + "attention.attention.qkv_lora_a.weight"), H(p + "attention.attention.qkv_lora_b.weight"), 3 * D, D, kLoraRank);
      merge_lora_into(wo, H(p + "attention.output.dense_lora_a.weight"), H(p + "attention.output.dense_lora_b.weight"), D, D, kLoraRank);
    }
    TRY(upload_linear(e, wq, Hopt(p + "attention.attention.qkv.bias"), 3 * D, D, &l.s_qkv));
    TRY(upload_folded_linear(e, wq, Hopt(p + "attention.attention.qkv.bias"), H(p + "layernorm_before.weight"),
                             H(p + "layernorm_before.bias"), 3 * D, D, &l.s_qkv_f));
    {
      const std::vector<float> wu = pad2(H(p + "intermediate.dense.weight"), Ir, D, I, D);
      const std::vector<float> bu = pad1(Hopt(p + "intermediate.dense.bias"), Ir, I);
      const std::vector<float> wd = pad2(H(p + "output.dense.weight"), D, Ir, D, I);
      TRY(upload_folded_linear(e, wu, &bu, H(p + "layernorm_after.weight"), H(p + "layernorm_after.bias"), I, D, &l.up_f));
      TRY(upload_linear(e, wo, Hopt(p + "attention.output.dense.bias"), D, D, &l.s_out));
      TRY(upload_linear(e, wu, &bu, I, D, &l.up));
      TRY(upload_linear(e, wd, Hopt(p + "output.dense.bias"), D, I, &l.down));
    }
  }
  TRY(upload_ln(e, "post_layernorm", &e->post_ln));
  TRY(upload_ln(e, "head.layernorm", &e->head_ln));
  {
    // nn.MultiheadAttention packed in_proj rows are [q; k; v] (modeling:1135-1137).  The query is
    // the learned probe only, identical for every frame: project and scale it once, in double.
    const std::vector<float>& w = H("head.attention.in_proj_weight");
    const std::vector<float>& b = H("head.attention.in_proj_bias");
    const std::vector<float>& probe = H("head.probe");
    std::vector<double> q(D);
    const int hd = e->hd;
    const double sc = 1.0 / std::sqrt((double)hd);
    for (int o = 0; o < D; ++o) {
      double acc = b[o];
      for (int k = 0; k < D; ++k) acc += (double)w[(size_t)o * D + k] * probe[k];
      q[o] = acc * sc;
    }
    // The keys only ever meet this one query, so the key projection folds into it: score_hn = (Wk_h^T q_h) . x_n + q_h . bk_h, and
    // the second term is constant over n — the softmax drops it.  U_h = Wk_h^T q_h, in double; the values need no projection of
    // the tokens either: ctx_h = Wv_h (sum_n p_hn x_n) + bv_h.
    const int heads = e->cfg.num_attention_heads;
    std::vector<uint16_t> uh((size_t)16 * D, 0), ul((size_t)16 * D, 0);
    std::vector<float> uf((size_t)16 * D, 0.f);
    for (int h = 0; h < heads; ++h)
      for (int d = 0; d < D; ++d) {
        double acc = 0.0;
        for (int j = 0; j < hd; ++j) acc += (double)w[((size_t)D + h * hd + j) * D + d] * q[h * hd + j];
        const float v = (float)acc;
        uf[(size_t)h * D + d] = v;
        uh[(size_t)h * D + d] = h_f2bf(v);
        ul[(size_t)h * D + d] = h_f2bf(v - h_bf2f(uh[(size_t)h * D + d]));
      }
    TRY(dev_upload<uint16_t>(e, uh, &e->head_u_hi));
    TRY(dev_upload<uint16_t>(e, ul, &e->head_u_lo));
    TRY(dev_upload<float>(e, uf, &e->head_u));          // fp32 copy for the generic-width kernel (sf_launch_pool_generic)
    std::vector<float> wv(w.begin() + (size_t)2 * D * D, w.end());
    std::vector<float> bv(b.begin() + 2 * D, b.end());
    TRY(dev_upload<float>(e, wv, &e->head_wv));
    TRY(dev_upload<float>(e, bv, &e->head_bv));
  }
  TRY(upload_linear(e, H("head.attention.out_proj.weight"), Hopt("head.attention.out_proj.bias"), D, D, &e->head_out, true));
  {
    const std::vector<float> w1 = pad2(H("head.mlp.fc1.weight"), Ir, D, I, D);
    const std::vector<float> b1 = pad1(Hopt("head.mlp.fc1.bias"), Ir, I);
    const std::vector<float> w2 = pad2(H("head.mlp.fc2.weight"), D, Ir, D, I);
    TRY(upload_linear(e, w1, &b1, I, D, &e->head_fc1, true));
    TRY(upload_linear(e, w2, Hopt("head.mlp.fc2.bias"), D, I, &e->head_fc2, true));
  }
#undef TRY
  e->finalized = true;
  e->generation = next_generation();   // caches created against an earlier packing are refused from here on
  return SF_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace carving
// ------------------------------------------------------------------------------------------------
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base((char*)b) {}
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

struct Workspace {
  float* resid; float* te_rows; float* ln_stats;
  float* embed_tab; bf16_t* patch_buf;  // pos + time table [T*N, D]; patch matrix [M, Kp] when the embedding GEMM runs on the
                                        // panel kernel (its bf16 output goes to xn_hi, so the A operand needs its own buffer)
  bf16_t *xn_hi, *xn_lo, *ctx_hi, *ctx_lo, *tmp_hi, *tmp_lo, *mid_hi, *mid_lo;
  void* qkv;          // spatial qkv; fast: bf16 [M,3D], accurate: fp32 [M,3D]
  void* tqkv;         // temporal qkv of the current layer when no cache is used
  float* attn_out;    // head: [F, D]
  float *head_ctx, *head_mid;      // head tail on one to four rows (streamed frames): fp32 context [F, D] and MLP activation [F, I]
  float *pool_z, *pool_ml;         // head: weighted token sums [F, S, heads, D] and {max, sum} [F, S, heads, 2] per token split
  bf16_t *pc_hi, *pc_lo, *hn_hi, *hn_lo, *hm_hi, *hm_lo;
  float *lhs_stage, *pool_stage;   // streaming only: graph-owned outputs, copied to the caller's tensors after the replay
  bf16_t* res_bf;                  // small-M LayerNorm fold: bf16 copy of the residual stream (A operand of the folded Linears)
  bf16_t* res_lo;                  // BASELINE-sized M, bf16 mode: lo plane of the residual stream (hi plane = xn_hi), see SfGemmArgs::resid_hi
  bf16_t* res_lo2;                 // accurate mode: third plane of the residual stream (hi = xn_hi, lo = xn_lo), see SfGemmArgs::resid_lo2
  size_t bytes;
};

static Workspace carve(const sf_encoder* e, void* base, int B, int T, int N, bool need_tqkv) {
  Workspace w;
  Carver c(base);
  const size_t M = (size_t)B * T * N, F = (size_t)B * T;
  const size_t D = e->D, I = e->I;
  const bool acc = e->compute == SF_COMPUTE_BF16X3;
  const size_t wide = D > (size_t)e->Kp ? D : (size_t)e->Kp;
  w.resid = c.take<float>(M * D);
  w.te_rows = c.take<float>((size_t)T * D);
  w.ln_stats = c.take<float>(M * 8);                  // rows of 8 floats in both modes (SfGemmArgs::ln_stats_wide)
  w.embed_tab = (!acc && M >= 2048) ? c.take<float>((size_t)T * N * D) : nullptr;
  w.patch_buf = (!acc && M >= 2048) ? c.take<bf16_t>(M * (size_t)e->Kp) : nullptr;
  w.xn_hi = c.take<bf16_t>(M * wide);
  w.xn_lo = acc ? c.take<bf16_t>(M * wide) : nullptr;
  w.ctx_hi = c.take<bf16_t>(M * D);
  w.ctx_lo = acc ? c.take<bf16_t>(M * D) : nullptr;
  if (!e->fused_temporal) {
    w.tmp_hi = c.take<bf16_t>(M * D);
    w.tmp_lo = acc ? c.take<bf16_t>(M * D) : nullptr;
  } else {
    w.tmp_hi = w.tmp_lo = nullptr;
  }
  w.mid_hi = c.take<bf16_t>(M * I);
  w.mid_lo = acc ? c.take<bf16_t>(M * I) : nullptr;
  w.qkv = c.take<char>(M * 3 * D * (acc ? 4 : 2));
  w.tqkv = need_tqkv ? (void*)c.take<char>(M * 3 * D * (acc ? 4 : 2)) : nullptr;
  w.attn_out = c.take<float>(F * D);
  w.head_ctx = c.take<float>(F * D);
  w.head_mid = c.take<float>(F * I);
  {
    size_t pz = sf_pool_z_floats((int)F, N, e->cfg.num_attention_heads, (int)D);
    const size_t pgen = sf_pool_generic_scratch_floats((int)F, N, e->cfg.num_attention_heads, (int)D);      // generic widths: scores + z
    w.pool_z = c.take<float>(pz > pgen ? pz : pgen);
  }
  w.pool_ml = c.take<float>(sf_pool_ml_floats((int)F, N, e->cfg.num_attention_heads));
  w.pc_hi = c.take<bf16_t>(F * D);          // the pooling head's one-row-per-frame tensors keep hi + lo planes in both modes
  w.pc_lo = c.take<bf16_t>(F * D);
  w.hn_hi = c.take<bf16_t>(F * D);
  w.hn_lo = c.take<bf16_t>(F * D);
  w.hm_hi = c.take<bf16_t>(F * I);
  w.hm_lo = c.take<bf16_t>(F * I);
  w.res_bf = (M <= (size_t)sf_infold_max_rows()) ? c.take<bf16_t>(M * D) : nullptr;      // accurate mode (round 6): the hi plane; the lo plane travels in xn_lo
  // lo plane of the residual stream: bf16 mode at BASELINE-sized M (pm / gm); accurate mode at small M (round 6: in-kernel LayerNorm fold of streamed frames)
  w.res_lo = ((!acc && M > (size_t)sf_infold_max_rows()) || (acc && M <= (size_t)sf_infold_max_rows())) ? c.take<bf16_t>(M * D) : nullptr;
  w.res_lo2 = (acc && M >= 2048) ? c.take<bf16_t>(M * D) : nullptr;
  w.lhs_stage = !need_tqkv ? c.take<float>(M * D) : nullptr;       // streaming carve (the cache holds the temporal qkv)
  w.pool_stage = !need_tqkv ? c.take<float>(F * D) : nullptr;
  w.bytes = (c.off + 255) & ~(size_t)255;
  return w;
}

// ------------------------------------------------------------------------------------------------
// the launch schedule
// ------------------------------------------------------------------------------------------------
static hipError_t run_linear(const sf_encoder* e, const DevLinear& lin, const bf16_t* a_hi, const bf16_t* a_lo,
                             int M, int epi, hipStream_t s, float* out_f32, bf16_t* out_hi, bf16_t* out_lo,
                             const float* resid = nullptr, float alpha = 1.f, int ldc = 0, int grp_rows = 0,
                             int grp_stride = 0, int grp_off = 0, const float* ln_stats = nullptr,
                             float* ln_stats_out = nullptr, bool ln_inkernel = false, const int* grp_off_dev = nullptr,
                             int grp_off_scale = 0, const bf16_t* resid_hi = nullptr, const bf16_t* resid_lo = nullptr,
                             bf16_t* resid_lo2 = nullptr, bool force_split = false) {
  SfGemmArgs g;
  memset(&g, 0, sizeof(g));
  const bool split = force_split || e->compute == SF_COMPUTE_BF16X3;
  g.a_hi = a_hi; g.a_lo = split ? a_lo : nullptr;
  g.w_hi = lin.w_hi; g.w_lo = split ? lin.w_lo : nullptr;
  g.bias = lin.bias;
  g.M = M; g.N = lin.N; g.K = lin.K;
  g.epi = epi; g.act = e->cfg.hidden_act; g.alpha = alpha; g.resid = resid;
  g.out_f32 = out_f32; g.out_hi = out_hi; g.out_lo = (split || resid_hi) ? out_lo : nullptr;
  g.resid_hi = resid_hi; g.resid_lo = resid_lo; g.resid_lo2 = resid_lo2; g.out_lo2 = resid_lo2;     // third plane: in place
  g.ldc = ldc ? ldc : lin.N;
  g.grp_rows = grp_rows; g.grp_stride = grp_stride; g.grp_off = grp_off;
  g.grp_off_dev = grp_off_dev; g.grp_off_scale = grp_off_scale;
  if (grp_rows > 0 && grp_stride == grp_rows && grp_off == 0 && !grp_off_dev) g.grp_rows = 0;   // identity remap (full clip)
  g.ln_stats = ln_stats; g.ln_s = ln_stats ? lin.ln_s : nullptr; g.ln_eps = e->cfg.layer_norm_eps;
  g.ln_stats_out = ln_stats_out;
  g.ln_stats_wide = 1;
  if (ln_inkernel) { g.ln_inkernel = 1; g.ln_s = lin.ln_s; }
  if (epi == SF_EPI_RESID_F32) g.out_hi = out_hi;     // LN-fold producer: bf16 copy of the new residual rows
  return sf_launch_gemm(g, split, s);
}

static bool ln_fold_small_ok(const sf_encoder* e, int M);

// LayerNorm folding needs the panel kernel as every residual producer (it emits the row statistics)
// and the 256^2 kernel as every consumer (it applies them): true for the BASELINE shape.
// Two clips per call (sf_tile_fold_min_rows() <= M <= sf_tile_max_rows()): the narrow tile kernel stays the residual producer (it beats
// the panel kernel there) but emits the row statistics, so that the folded consumers can run on the 256^2 kernel (which beats the wide tiles there)
static bool ln_fold_tile_ok(const sf_encoder* e, int M) {
  if (e->compute != SF_COMPUTE_BF16 || sf_sw(SW_DISABLE_LN_FOLD)) return false;
  SfGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.ln_stats_wide = 1;
  g.M = M; g.N = e->D; g.ldc = e->D; g.epi = SF_EPI_RESID_F32; g.out_f32 = (float*)1; g.resid = (const float*)1; g.out_hi = (bf16_t*)1;
  g.ln_stats_out = (float*)1;
  g.K = e->D;
  if (!sf_gemm_tile_supported(g, false)) return false;
  g.K = e->I;
  if (!sf_gemm_tile_supported(g, false)) return false;
  memset(&g, 0, sizeof(g));
  g.ln_stats_wide = 1;
  g.M = M; g.epi = SF_EPI_BF16; g.K = e->D; g.N = 3 * e->D;
  if (!sf_gemm256_supported(g, false)) return false;
  g.epi = SF_EPI_ACT_BF16; g.N = e->I; g.act = e->cfg.hidden_act;
  return sf_gemm256_supported(g, false);
}

static bool ln_fold_ok(const sf_encoder* e, int M) {
  if (e->compute != SF_COMPUTE_BF16) return false;
  if (sf_sw(SW_DISABLE_LN_FOLD)) return false;        // A/B switch for measurements
  if (ln_fold_tile_ok(e, M)) return true;
  if (ln_fold_small_ok(e, M)) return false;              // the small-M fold (in-kernel statistics: skinny / 64 x 64 / tile kernels) takes these
  SfGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.M = M; g.N = e->D; g.epi = SF_EPI_RESID_F32;
  g.K = e->D;
  if (!sf_gemm_panel_supported(g, false)) return false;
  g.K = e->I;
  if (!sf_gemm_panel_supported(g, false)) return false;
  g.epi = SF_EPI_BF16; g.K = e->D; g.N = 3 * e->D;
  if (!sf_gemm256_supported(g, false)) return false;
  g.epi = SF_EPI_ACT_BF16; g.N = e->I; g.act = e->cfg.hidden_act;
  return sf_gemm256_supported(g, false);
}

// fp32-accurate mode: the fold on the bf16x3 256^2 kernel, with the residual stream itself carried as the hi + lo bf16 planes
// that are the folded Linears' operands (xn_hi / xn_lo): a residual producer reads and writes the planes and emits one
// {sum x, sum x^2} pair per 256-column tile; no fp32 residual, no LayerNorm launch.  The planes hold x to 2^-18 relative, the
// precision every bf16x3 operand has anyway.  SF_DISABLE_ACC_FOLD restores fp32 residual + standalone LayerNorm (A/B).
static bool ln_fold_acc_ok(const sf_encoder* e, int M) {
  if (e->compute != SF_COMPUTE_BF16X3 || (e->D % 256) || e->D > 1024) return false;      // one statistics pair per 256-column producer tile, four per row
  const bool off = sf_sw(SW_DISABLE_ACC_FOLD) != nullptr;
  if (off) return false;
  SfGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.a_lo = (const bf16_t*)1; g.w_lo = (const bf16_t*)1; g.ln_stats_wide = 1;
  g.M = M; g.N = e->D; g.K = e->D; g.epi = SF_EPI_RESID_F32;
  g.out_hi = (bf16_t*)1; g.out_lo = (bf16_t*)1; g.ln_stats_out = (float*)1; g.resid_hi = (const bf16_t*)1; g.resid_lo = (const bf16_t*)1;
  if (!sf_gemm256_supported(g, true)) return false;
  g.K = e->I;
  if (!sf_gemm256_supported(g, true)) return false;
  g.ln_stats_out = nullptr; g.out_hi = nullptr; g.resid_hi = g.resid_lo = nullptr; g.ln_stats = (const float*)1; g.ln_s = (const float*)1;
  g.epi = SF_EPI_BF16; g.K = e->D; g.N = 3 * e->D;
  if (!sf_gemm256_supported(g, true)) return false;
  g.epi = SF_EPI_ACT_BF16; g.N = e->I; g.act = e->cfg.hidden_act;
  return sf_gemm256_supported(g, true);
}

// bf16 mode at widths the panel kernel does not take (D = 1024: a ViT-L-shaped encoder): the same plane-form residual stream + fold with
// the 256^2 kernel as producer (planes in / out + one statistics pair per 256-column tile) and consumer (its bf16 LNF instance sums
// the four pairs).  Without it those widths pay three standalone LayerNorm launches per layer (9.7 % of the ViT-L forward).
static bool ln_fold_g256_ok(const sf_encoder* e, int M) {
  if (e->compute != SF_COMPUTE_BF16 || (e->D % 256) || e->D > 1024) return false;
  if (sf_sw(SW_DISABLE_LN_FOLD)) return false;
  SfGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.ln_stats_wide = 1;
  g.M = M; g.N = e->D; g.K = e->D; g.epi = SF_EPI_RESID_F32;
  g.out_hi = (bf16_t*)1; g.out_lo = (bf16_t*)1; g.ln_stats_out = (float*)1; g.resid_hi = (const bf16_t*)1; g.resid_lo = (const bf16_t*)1;
  if (!sf_gemm256_supported(g, false)) return false;
  g.K = e->I;
  if (!sf_gemm256_supported(g, false)) return false;
  g.ln_stats_out = nullptr; g.out_hi = (bf16_t*)1; g.out_lo = nullptr; g.resid_hi = g.resid_lo = nullptr; g.ln_stats = (const float*)1; g.ln_s = (const float*)1;
  g.epi = SF_EPI_BF16; g.K = e->D; g.N = 3 * e->D;
  if (!sf_gemm256_supported(g, false)) return false;
  g.epi = SF_EPI_ACT_BF16; g.N = e->I; g.act = e->cfg.hidden_act;
  return sf_gemm256_supported(g, false);
}

// Small-M variant (the per-frame streaming step): the skinny GEMM derives the row statistics itself from the A fragments
// it reads anyway, every residual producer only adds the bf16 copy of its output rows.
static bool ln_fold_small_ok(const sf_encoder* e, int M) {
  if (M > sf_infold_max_rows()) return false;
  const bool off = sf_sw(SW_DISABLE_STREAM_FOLD) != nullptr;
  if (off) return false;
  // accurate mode (round 6): the consumers take x as hi + lo planes and sum the statistics of hi + lo (32 x 32 skinny kernel only)
  const bool acc = e->compute == SF_COMPUTE_BF16X3;
  auto takes = [acc](const SfGemmArgs& g) { return sf_gemm_skinny_supported(g, acc) || (!acc && sf_gemm_tile_supported(g, false)); };
  SfGemmArgs g;
  memset(&g, 0, sizeof(g));
  if (acc) { g.a_lo = (const bf16_t*)1; g.w_lo = (const bf16_t*)1; g.out_lo = (bf16_t*)1; }
  g.M = M; g.K = e->D; g.ldc = 3 * e->D; g.ln_inkernel = 1; g.ln_s = (const float*)1; g.out_hi = (bf16_t*)1;
  g.epi = acc ? SF_EPI_F32 : SF_EPI_BF16; g.N = 3 * e->D;
  if (acc) g.out_f32 = (float*)1;
  if (!takes(g)) return false;
  g.epi = SF_EPI_ACT_BF16; g.N = e->I; g.ldc = e->I;
  if (!takes(g)) return false;
  g.ln_inkernel = 0; g.ln_s = nullptr; g.epi = SF_EPI_RESID_F32; g.N = e->D; g.ldc = e->D; g.out_hi = (bf16_t*)1;
  g.out_f32 = (float*)1; g.resid = (const float*)1;
  if (!takes(g)) return false;                                // producers (they add the bf16 copy): K = D ...
  g.K = e->I;
  return takes(g);                                            // ... and K = I
}

// the head's per-frame tail on one to four rows runs as row-vector launches (sf_launch_rowlin)
static bool sf_head_rows_ok(const sf_encoder* e, int F) {
  return e->head_out.w_lo && e->head_fc1.w_lo && e->head_fc2.w_lo && sf_rowlin_supported(F, e->D, e->D) && sf_rowlin_supported(F, e->I, e->D) &&
         sf_rowlin_supported(F, e->D, e->I);
}

static int time_rows(const sf_encoder* e, int t_past, int T, bool streaming, SfRowIndex* idx) {
  const int nf = e->cfg.num_frames;
  if (T > 256) return set_err(SF_ERR_INVALID, "at most 256 frames per call (got %d)", T);
  idx->n = T;
  if (streaming) {
    // vqa_enc:328-369: direct rows; the reference raises past num_frames rows
    if (t_past + T > nf)
      return set_err(SF_ERR_CAPACITY, "streaming needs time-embedding row %d but config.num_frames is %d", t_past + T - 1, nf);
    for (int t = 0; t < T; ++t) idx->idx[t] = t_past + t;
  } else if (T <= nf) {
    for (int t = 0; t < T; ++t) idx->idx[t] = t;          // modeling:436-439 slice
  } else {
    // F.interpolate(mode="nearest") (modeling:441-447): src = floor(dst * scale), scale = float(in)/out
    const float scale = (float)nf / (float)T;
    for (int t = 0; t < T; ++t) {
      int src = (int)floorf((float)t * scale);
      idx->idx[t] = src < nf - 1 ? src : nf - 1;
    }
  }
  return SF_OK;
}

// One call of the encoder over T new frames.  tq = per-layer temporal qkv buffers ([B,cap,N,3D]).
static int run_forward(sf_encoder* e, const void* pixels, int pixel_dtype, int B, int T, int H, int W,
                       float* last_hidden, float* pooler, float* hidden_states, const float* pos_dev,
                       const Workspace& ws, void* const* layer_tqkv, int cap, int t_past, bool streaming,
                       hipStream_t s, float* attentions = nullptr, int stages = 7, int la = 0, int lb = -1,
                       int ready = 0, const SfStreamParams* sp = nullptr, const int* pos3 = nullptr) {
  // pos3 (streaming, T == 1): {time-embedding row, cache slot, keys seen} when they differ from (t_past, t_past, t_past + 1):
  // the sliding-window policy of a full cache
  const int t_row = pos3 ? pos3[0] : t_past, slot = pos3 ? pos3[1] : t_past, tk = pos3 ? pos3[2] : t_past + T;
  // sp (streaming, T == 1, graph capture): the cache position is read on the DEVICE from sp->t_past by the kernels that need
  // it (time-embedding row, KV-cache append row, single-query attention); t_past here only selects kernel variants
  // ready: 1 = the patch matrix is already in the workspace (the streaming entry extracts it outside its graph),
  //        2 = ws.res_bf already holds bf16(residual) (a later layer range of the same call)
  const bool patches_ready = (ready & 1) != 0;
  // stages: 1 = embeddings -> ws.resid, 2 = layers [la, lb) on ws.resid, 4 = post-LayerNorm + pooling head,
  // 8 = pooling head alone on already-normalised tokens in ws.resid
  // (the sub-module entry points run them one at a time on a caller-owned residual stream)
  if (lb < 0) lb = e->L;
  const sf_config& c = e->cfg;
  const int P = c.patch_size, D = e->D, heads = c.num_attention_heads;
  const int N = (H / P) * (W / P);
  const int M = B * T * N, F = B * T;
  const bool acc = e->compute == SF_COMPUTE_BF16X3;
  const int qkv_epi = acc ? SF_EPI_F32 : SF_EPI_BF16;
  const size_t esz = acc ? 4 : 2;
  const float scale = 1.0f / sqrtf((float)e->hd);

  bool embed_emitted_stats = false;
  // statistics rows of the bf16 fold are 8 floats = four pairs; the 384-column panel kernel fills pairs 0 and 2 only
  if (e->compute == SF_COMPUTE_BF16 && !streaming && (stages & 3) && ln_fold_ok(e, B * T * ((H / e->cfg.patch_size) * (W / e->cfg.patch_size))))
    HIP_TRY(hipMemsetAsync(ws.ln_stats, 0, (size_t)B * T * ((H / e->cfg.patch_size) * (W / e->cfg.patch_size)) * 8 * sizeof(float), s));
  // pm: the residual stream of a whole bf16 forward at BASELINE-sized M travels as hi + lo bf16 planes (hi = xn_hi, the A operand
  // of the folded Linears; lo = res_lo) instead of fp32: every residual producer moves 154 MB instead of 192 (no separate bf16
  // copy).  Only for complete forwards without hidden_states (the fp32 tensor is the interface of the stage-wise entry points).
  const bool planes_off = sf_sw(SW_DISABLE_RESID_PLANES) != nullptr;
  bool pm = !planes_off && !acc && !streaming && ws.res_lo && !hidden_states && (stages & 7) == 7 && ln_fold_ok(e, M) && !ln_fold_tile_ok(e, M) && ws.embed_tab &&
            ws.patch_buf && e->Kp % 32 == 0 && e->Kp >= 128 && D == 768 && !sf_sw(SW_EMBED_VIA_GEMM128);
  if (stages & 1) {
  SfRowIndex idx;
  int rc = time_rows(e, t_row, T, streaming, &idx);
  if (rc) return rc;
  if (sp) { for (int t = 0; t < T; ++t) idx.idx[t] = t; }
  // streamed frame inside the position-free graph: the embedding GEMM reads its time-embedding row straight from the table
  // (row index from the device block) when the skinny kernel takes the shape — no gather launch
  bool time_folded = false;
  if (sp && T == 1) {
    SfGemmArgs gt;
    memset(&gt, 0, sizeof(gt));
    gt.a_hi = ws.xn_hi; gt.a_lo = acc ? ws.xn_lo : nullptr; gt.w_hi = e->patch.w_hi; gt.w_lo = acc ? e->patch.w_lo : nullptr;
    gt.M = M; gt.N = D; gt.K = e->Kp; gt.ldc = D; gt.epi = SF_EPI_EMBED_F32; gt.out_f32 = ws.resid;
    time_folded = sf_gemm_skinny_supported(gt, acc) && !sf_sw(SW_DISABLE_SKINNY) && !ln_fold_ok(e, M);      // exactly when sf_launch_gemm takes the skinny kernel
  }
  if (!time_folded) HIP_TRY(sf_launch_gather_rows(e->time_tab, ws.te_rows, idx, D, s, sp ? &sp->t_row : nullptr));
  // LN folding (decided here because the folded path lets the embedding GEMM emit bf16(x) + row statistics itself:
  // panel kernel, out = table[m % (T N)] + patches W^T + b with table = pos + time rows)
  bool embed_panel = false;
  if (ln_fold_ok(e, M) && !streaming && (stages & 2) && ws.embed_tab && ws.patch_buf && e->Kp % 32 == 0 && e->Kp >= 128 && D == 768 &&
      !sf_sw(SW_EMBED_VIA_GEMM128))
    embed_panel = true;
  bf16_t* patches = embed_panel ? ws.patch_buf : ws.xn_hi;
  if (!patches_ready)
    HIP_TRY(sf_launch_patchify(pixels, pixel_dtype == SF_U8 ? 2 : (pixel_dtype == SF_BF16 ? 1 : 0), patches, ws.xn_lo, F, c.num_channels, H, W, P, s,
                               &e->pixel_norm, nullptr, nullptr, nullptr, e->Kp));
  {
    SfGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a_hi = patches; g.a_lo = acc ? ws.xn_lo : nullptr;
    g.w_hi = e->patch.w_hi; g.w_lo = acc ? e->patch.w_lo : nullptr;
    g.bias = e->patch.bias;
    g.M = M; g.N = D; g.K = e->Kp;
    g.out_f32 = ws.resid; g.ldc = D;
    SfGemmArgs gp = g;
    gp.epi = SF_EPI_RESID_F32; gp.alpha = 1.f; gp.resid = ws.embed_tab; gp.resid_mod = T * N;
    gp.out_hi = ws.xn_hi; gp.ln_stats_out = ws.ln_stats; gp.ln_stats_wide = 1;
    if (pm) {
      SfGemmArgs gq = gp;
      gq.out_f32 = nullptr; gq.out_lo = ws.res_lo;
      if (embed_panel && sf_gemm_panel_supported(gq, false)) gp = gq; else pm = false;
    }
    // two clips per call (ln_fold_tile_ok): the embedding GEMM on the statistics-producing narrow tile (table add in its epilogue)
    SfGemmArgs gt = g;
    gt.epi = SF_EPI_EMBED_F32; gt.pos = pos_dev ? pos_dev : e->pos; gt.time_rows = ws.te_rows; gt.Np = N; gt.Tn = T;
    gt.out_hi = ws.xn_hi; gt.ln_stats_out = ws.ln_stats; gt.ln_stats_wide = 1;
    if (!acc && !streaming && (stages & 2) && patches != ws.xn_hi && ln_fold_tile_ok(e, M) && sf_gemm_tile_supported(gt, false)) {      // (A must not alias out_hi)
      HIP_TRY(sf_launch_gemm_tile(gt, s));
      embed_emitted_stats = true;
    } else
    if (embed_panel && sf_gemm_panel_supported(gp, false)) {
      HIP_TRY(sf_launch_pos_time_table(pos_dev ? pos_dev : e->pos, ws.te_rows, ws.embed_tab, T, N, D, s));
      HIP_TRY(sf_launch_gemm_panel(gp, s));
      embed_emitted_stats = true;
    } else {
      g.epi = SF_EPI_EMBED_F32;
      g.pos = pos_dev ? pos_dev : e->pos; g.time_rows = ws.te_rows; g.Np = N; g.Tn = T;
      if (time_folded) { g.time_rows = e->time_tab; g.time_base_dev = &sp->t_row; }
      if (ws.res_bf && (!acc || ws.res_lo) && ln_fold_small_ok(e, M) && (stages & 2)) { g.out_hi = ws.res_bf; if (acc) g.out_lo = ws.res_lo; }   // layer 0's folded qkv reads bf16(x) (accurate: hi + lo)
      HIP_TRY(sf_launch_gemm(g, acc, s));
    }
  }
  }
  const size_t hs_stride = (size_t)M * D;
  // LN folding (bf16 mode, BASELINE-sized M): xn_hi holds bf16(residual), ln_stats the row sums; the
  // three per-layer LayerNorm launches disappear into the neighbouring GEMM epilogues.
  const bool fold = ln_fold_ok(e, M) && !streaming;
  // sfold: the same algebra at small M (streamed frames), statistics computed inside the consumer GEMM
  const bool sfold = !fold && ws.res_bf && (!acc || ws.res_lo) && ln_fold_small_ok(e, M);
  // xm: the accurate mode's counterpart of fold + pm (ln_fold_acc_ok): whole clips on the plane-fed attention kernels
  const bool two_planes = sf_sw(SW_ACC_TWO_PLANES) != nullptr;      // A/B: drop the third plane (max-abs 1.4e-4 instead of 5e-5)
  bf16_t* plo2 = two_planes ? nullptr : ws.res_lo2;
  const bool xm = acc && !streaming && !hidden_states && (stages & 7) == 7 && !layer_tqkv && cap == T && t_past == 0 &&
                  sf_temporal_planes_ok(T, T) && sf_spatial_planes_ok(N, attentions != nullptr) && ln_fold_acc_ok(e, M);
  // gm: the bf16 mode's counterpart at widths without a panel kernel (ln_fold_g256_ok): lo plane = res_lo, two planes
  const bool gm = !acc && !fold && !sfold && !streaming && !hidden_states && (stages & 7) == 7 && ws.res_lo != nullptr && ln_fold_g256_ok(e, M);
  bf16_t* fold_hi = (fold || xm || gm) ? ws.xn_hi : (sfold ? ws.res_bf : nullptr);
  float* fold_st = (fold || xm || gm) ? ws.ln_stats : nullptr;
  const bf16_t* ln_in = sfold ? ws.res_bf : ws.xn_hi;       // A operand of the three LayerNorm'd Linears
  const bf16_t* ln_in_lo = (sfold && acc) ? ws.res_lo : ws.xn_lo;      // its lo plane (accurate mode)
  const bool anyfold = fold || sfold || xm || gm;
  const bool rplanes = pm || xm || gm;                       // residual stream as two bf16 planes: hi = xn_hi, lo = plo
  bf16_t* plo = (pm || gm) ? ws.res_lo : (xm ? ws.xn_lo : ((sfold && acc) ? ws.res_lo : nullptr));      // sfold in the accurate mode: lo plane of the residual for the folded consumers
  if (fold && (stages & 2) && !embed_emitted_stats) HIP_TRY(sf_launch_rowstats_cast(ws.resid, ws.xn_hi, ws.ln_stats, M, D, s, nullptr, nullptr, 1));
  if (!xm) plo2 = nullptr;
  if (xm) HIP_TRY(sf_launch_rowstats_cast(ws.resid, ws.xn_hi, ws.ln_stats, M, D, s, ws.xn_lo, plo2));      // embeddings -> planes + wide statistics
  if (gm) HIP_TRY(sf_launch_rowstats_cast(ws.resid, ws.xn_hi, ws.ln_stats, M, D, s, ws.res_lo, nullptr));
  if (sfold && (stages & 2) && !(stages & 1) && !(ready & 2)) HIP_TRY(sf_launch_split(ws.resid, ws.res_bf, acc ? ws.res_lo : nullptr, (size_t)M * D, s));   // sf_layers entry
  for (int li = la; li < lb && (stages & 2); ++li) {
    const DevLayer& l = e->layers[li];
    if (hidden_states)
      HIP_TRY(hipMemcpyAsync(hidden_states + li * hs_stride, ws.resid, hs_stride * 4, hipMemcpyDeviceToDevice, s));
    // ---- temporal attention (modeling:937-958) ---------------------------------------------------
    if (!anyfold) HIP_TRY(sf_launch_layernorm(ws.resid, l.ln_t.g, l.ln_t.b, nullptr, ws.xn_hi, ws.xn_lo, M, D, c.layer_norm_eps, s));
    void* tq = layer_tqkv ? layer_tqkv[li] : ws.tqkv;
    // accurate mode, whole short clip (no cache): hi + lo bf16 planes for the DMA kernel, like the spatial attention below
    const bool tplanes = acc && !layer_tqkv && cap == T && t_past == 0 && sf_temporal_planes_ok(T, T);
    const size_t tsz = tplanes ? 2 : esz;
    // lab library, SF_QKV_FUSED=1 (bf16 mode, whole 16-frame clips, no cache): qkv projection + temporal attention in ONE launch
    // (sf_gemm_qkv.hip; bit-identical, -0.27 ms of kernel time per forward under rocprof, +0.07 ms on the wall clock: docs/history.md H.4)
    bool t_fused_attn = false;
#ifdef SF_LAB
    if (fold && !acc && !layer_tqkv && cap == T && t_past == 0 && tk == T && !sp && l.t_qkv_fp.w_hi) {
      SfQkvArgs q;
      memset(&q, 0, sizeof(q));
      q.a = ln_in; q.w = l.t_qkv_fp.w_hi; q.bias = l.t_qkv_fp.bias; q.ln_s = l.t_qkv_fp.ln_s; q.ln_stats = fold_st; q.ln_eps = c.layer_norm_eps;
      q.M = M; q.K = D; q.D = D; q.B = B; q.T = T; q.NP = N; q.out = ws.ctx_hi; q.scale = scale; q.causal = c.enable_causal_temporal;
      if (sf_gemm_qkv_supported(q, true)) {
        HIP_TRY(prof_span(e, 3, s, [&]() { return sf_launch_gemm_qkv(q, true, s); }));
        t_fused_attn = true;
      }
    }
#endif
#ifdef SF_LAB
    // lab library, SF_STREAM_QKV_FUSE=1 (one streamed frame per stream, bf16 mode, cache of <= 64 frames): projection + cache append +
    // single-query attention in one launch (tools/lab/sf_stream_fused.hip; bit-identical, 15.0 against 6.1 + 7.4 us: a CU's ingest)
    if (!t_fused_attn && sfold && !acc && T == 1 && layer_tqkv && l.t_qkv_f.ln_s) {
      SfStreamQkvArgs q;
      memset(&q, 0, sizeof(q));
      q.a = ln_in; q.w_frag = l.t_qkv_f.w_frag; q.bias = l.t_qkv_f.bias; q.ln_s = l.t_qkv_f.ln_s; q.ln_eps = c.layer_norm_eps;
      q.M = M; q.K = l.t_qkv_f.K; q.D = D; q.heads = heads; q.N = N;
      q.cache = (bf16_t*)tq; q.cap = cap; q.slot = slot; q.Tk = tk; q.pos_dev = sp ? &sp->slot : nullptr;
      q.ctx = ws.ctx_hi; q.scale = scale;
      if (l.t_qkv_f.N == 3 * D && sf_stream_qkv_decode_supported(q)) {
        HIP_TRY(prof_span(e, 3, s, [&]() { return sf_launch_stream_qkv_decode(q, s); }));
        t_fused_attn = true;
      }
    }
#endif
    if (!t_fused_attn)
    HIP_TRY(run_linear(e, anyfold ? l.t_qkv_f : l.t_qkv, ln_in, ln_in_lo, M, tplanes ? (int)SF_EPI_BF16 : qkv_epi, s, (float*)tq, (bf16_t*)tq,
                       tplanes ? (bf16_t*)tq + (size_t)M * 3 * D : nullptr, nullptr, 1.f,
                       3 * D, T * N, cap * N, sp ? 0 : slot * N, fold_st, nullptr, sfold, sp ? &sp->slot : nullptr, N));
    if (!t_fused_attn) {
      SfAttnArgs a;
      memset(&a, 0, sizeof(a));
      a.q = tq; a.k = (char*)tq + (size_t)D * tsz; a.v = (char*)tq + (size_t)2 * D * tsz;
      a.in_is_f32 = acc && !tplanes; a.lo_plane_off = tplanes ? (long long)M * 3 * D : 0;
      a.row_pitch_q = 3 * D; a.row_pitch_kv = 3 * D; a.heads = heads; a.scale = scale; a.head_dim = e->hd;
      a.N = N; a.B = B; a.Tq = T; a.Tk = tk; a.Tcap = cap; a.t_past = tk - T;
      a.causal = c.enable_causal_temporal; a.Tq_cap = cap; a.q_t0 = slot;
      a.pos_dev = sp ? &sp->slot : nullptr;
      a.ctx_hi = ws.ctx_hi; a.ctx_lo = ws.ctx_lo; a.D = D;
      HIP_TRY(prof_span(e, 3, s, [&]() { return sf_launch_temporal_attention(a, acc, s); }));
    }
    if (e->fused_temporal) {
      HIP_TRY(prof_span(e, 0, s, [&]() {
        return run_linear(e, l.t_fused, ws.ctx_hi, ws.ctx_lo, M, SF_EPI_RESID_F32, s, rplanes ? nullptr : ws.resid, fold_hi, plo, rplanes ? nullptr : ws.resid, l.gate_tanh,
                          0, 0, 0, 0, nullptr, fold_st, false, nullptr, 0, rplanes ? fold_hi : nullptr, plo, plo2); }));
    } else {
      HIP_TRY(run_linear(e, l.t_out, ws.ctx_hi, ws.ctx_lo, M, SF_EPI_BF16, s, nullptr, ws.tmp_hi, ws.tmp_lo));
      HIP_TRY(run_linear(e, l.t_dense, ws.tmp_hi, ws.tmp_lo, M, SF_EPI_RESID_F32, s, rplanes ? nullptr : ws.resid, fold_hi, plo, rplanes ? nullptr : ws.resid, l.gate_tanh,
                         0, 0, 0, 0, nullptr, fold_st, false, nullptr, 0, rplanes ? fold_hi : nullptr, plo, plo2));
    }
    // ---- spatial attention (modeling:962-996) ------------------------------------------------------
    if (!anyfold) HIP_TRY(sf_launch_layernorm(ws.resid, l.ln_b.g, l.ln_b.b, nullptr, ws.xn_hi, ws.xn_lo, M, D, c.layer_norm_eps, s));
    // accurate mode: q / k / v leave the GEMM as hi + lo bf16 planes (the bytes of the fp32 tensor) when the DMA attention
    // kernel can take them; fp32 otherwise (probabilities requested, more than 224 tokens per frame)
    const bool planes = acc && sf_spatial_planes_ok(N, attentions != nullptr);
    const size_t sesz = planes ? 2 : esz;
    // lab library, SF_SQKV_PANEL=1: the folded spatial qkv projection on the panel tile (768 tiles of 196 x 384 = three whole rounds)
    bool s_qkv_panel = false;
#ifdef SF_LAB
    static const bool s_panel_on = SF_LAB_SWITCH("SF_SQKV_PANEL") != 0;
    if (s_panel_on && fold && !acc && qkv_epi == SF_EPI_BF16 && l.s_qkv_f.ln_s) {
      SfQkvArgs q;
      memset(&q, 0, sizeof(q));
      q.a = ln_in; q.w = l.s_qkv_f.w_hi; q.bias = l.s_qkv_f.bias; q.ln_s = l.s_qkv_f.ln_s; q.ln_stats = fold_st; q.ln_eps = c.layer_norm_eps;
      q.M = M; q.K = D; q.D = D; q.out = (bf16_t*)ws.qkv;
      if (sf_gemm_qkv_supported(q, false)) {
        HIP_TRY(sf_launch_gemm_qkv(q, false, s));
        s_qkv_panel = true;
      }
    }
#endif
    if (!s_qkv_panel)
    HIP_TRY(run_linear(e, anyfold ? l.s_qkv_f : l.s_qkv, ln_in, ln_in_lo, M, planes ? (int)SF_EPI_BF16 : qkv_epi, s, (float*)ws.qkv, (bf16_t*)ws.qkv,
                       planes ? (bf16_t*)ws.qkv + (size_t)M * 3 * D : nullptr, nullptr, 1.f, 0, 0, 0, 0, fold_st, nullptr, sfold));
    {
      SfAttnArgs a;
      memset(&a, 0, sizeof(a));
      a.q = ws.qkv; a.k = (char*)ws.qkv + (size_t)D * sesz; a.v = (char*)ws.qkv + (size_t)2 * D * sesz;
      a.in_is_f32 = acc && !planes; a.lo_plane_off = planes ? (long long)M * 3 * D : 0;
      a.row_pitch_q = 3 * D; a.row_pitch_kv = 3 * D; a.heads = heads; a.scale = scale; a.head_dim = e->hd;
      a.N = N; a.frames = F; a.ctx_hi = ws.ctx_hi; a.ctx_lo = ws.ctx_lo; a.D = D;
      a.probs = attentions ? attentions + (size_t)(li - la) * F * heads * N * N : nullptr;
      HIP_TRY(prof_span(e, 2, s, [&]() { return sf_launch_spatial_attention(a, acc, s); }));
    }
    HIP_TRY(prof_span(e, 0, s, [&]() {
      return run_linear(e, l.s_out, ws.ctx_hi, ws.ctx_lo, M, SF_EPI_RESID_F32, s, rplanes ? nullptr : ws.resid, fold_hi, plo, rplanes ? nullptr : ws.resid, 1.f,
                        0, 0, 0, 0, nullptr, fold_st, false, nullptr, 0, rplanes ? fold_hi : nullptr, plo, plo2); }));
    // ---- MLP (modeling:997-1000) ---------------------------------------------------------------------
    if (!anyfold) HIP_TRY(sf_launch_layernorm(ws.resid, l.ln_a.g, l.ln_a.b, nullptr, ws.xn_hi, ws.xn_lo, M, D, c.layer_norm_eps, s));
    HIP_TRY(run_linear(e, anyfold ? l.up_f : l.up, ln_in, ln_in_lo, M, SF_EPI_ACT_BF16, s, nullptr, ws.mid_hi, ws.mid_lo, nullptr, 1.f,
                       0, 0, 0, 0, fold_st, nullptr, sfold));
    HIP_TRY(prof_span(e, 1, s, [&]() {
      return run_linear(e, l.down, ws.mid_hi, ws.mid_lo, M, SF_EPI_RESID_F32, s, rplanes ? nullptr : ws.resid, fold_hi, plo, rplanes ? nullptr : ws.resid, 1.f,
                        0, 0, 0, 0, nullptr, fold_st, false, nullptr, 0, rplanes ? fold_hi : nullptr, plo, plo2); }));
  }
  if (hidden_states && (stages & 2) && lb == e->L)
    HIP_TRY(hipMemcpyAsync(hidden_states + (size_t)e->L * hs_stride, ws.resid, hs_stride * 4, hipMemcpyDeviceToDevice, s));
  if (!(stages & 12)) return SF_OK;
  // ---- post LayerNorm + pooling head (modeling:1330-1340, 1141-1154) -------------------------------
  // (the head reads the fp32 tokens: no bf16 copy of the normalised rows is written any more)
  if (stages & 4)       // pm: the rows arrive as the two planes of the residual stream
    HIP_TRY(sf_launch_layernorm(ws.resid, e->post_ln.g, e->post_ln.b, last_hidden, nullptr, nullptr, M, D, c.layer_norm_eps, s,
                                rplanes ? ws.xn_hi : nullptr, plo, plo2, sp ? &sp->lhs : nullptr));   // sp: straight into the caller's tensor
  // stage 8: the head alone on tokens the caller has already normalised (model.head(x)): they sit in ws.resid
  if (pooler) {
    // The probe attention without projecting the tokens (sf_pool_head.hip): scores = x . U, z_h = sum_n p_hn x_n on the fp32
    // tokens, ctx_h = Wv_h z_h + bv_h — bf16x3 / fp32 arithmetic in BOTH modes (the [M, 2D] k / v tensor and its 59 GFLOP are gone).
    const bool hacc = !acc;          // the head's one-row-per-frame tensors keep hi + lo planes in both modes (force_split in bf16 mode)
    {
      const float* tok = (stages & 4) ? last_hidden : ws.resid;
      const bool rows = sf_head_rows_ok(e, F);
      if (e->hd != 64 || D > 1024) {      // widths the MFMA probe kernels do not cover: the whole probe attention of a frame in one fp32 workgroup
        SfPoolGenArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.x = tok; ga.x_ind = (sp && (stages & 4)) ? reinterpret_cast<const float* const*>(&sp->lhs) : nullptr;
        ga.u = e->head_u; ga.wv = e->head_wv; ga.ldw = D; ga.bv = e->head_bv;
        ga.F = F; ga.N = N; ga.heads = heads; ga.hd = e->hd; ga.D = D;
        ga.scratch = ws.pool_z;
        if (rows) ga.ctx_f32 = ws.head_ctx; else { ga.ctx_hi = ws.pc_hi; ga.ctx_lo = ws.pc_lo; }
        HIP_TRY(sf_launch_pool_generic(ga, s));
      } else {
      SfPoolArgs pa;
      memset(&pa, 0, sizeof(pa));
      pa.x = tok; pa.x_ind = (sp && (stages & 4)) ? reinterpret_cast<const float* const*>(&sp->lhs) : nullptr;
      pa.u_hi = e->head_u_hi; pa.u_lo = e->head_u_lo; pa.zpart = ws.pool_z; pa.ml = ws.pool_ml;
      pa.F = F; pa.N = N; pa.heads = heads; pa.D = D; pa.S = sf_pool_splits(F, N, heads); pa.normalize = pa.S == 1;
      HIP_TRY(sf_launch_pool_probe(pa, s));
      SfPoolCtxArgs ca;
      memset(&ca, 0, sizeof(ca));
      ca.zpart = ws.pool_z; ca.ml = ws.pool_ml; ca.wv = e->head_wv; ca.ldw = D; ca.bv = e->head_bv;
      ca.ctx_hi = ws.pc_hi; ca.ctx_lo = ws.pc_lo; ca.F = F; ca.heads = heads; ca.D = D; ca.S = pa.S;
      if (rows) { ca.ctx_hi = ca.ctx_lo = nullptr; ca.ctx_f32 = ws.head_ctx; }
      HIP_TRY(sf_launch_pool_ctx(ca, s));
      }
      if (rows) {
        // One to four rows (streamed frames): the per-frame tail as three row-vector launches with fp32 activations — out_proj,
        // LayerNorm + fc1 + GELU, fc2 + residual (written straight into the caller's pooler_output inside the position-free graph)
        SfRowLinArgs r;
        memset(&r, 0, sizeof(r));
        r.F = F; r.act = -1;
        r.x = ws.head_ctx; r.ldx = D; r.K = D; r.N = D; r.w_hi = e->head_out.w_hi; r.w_lo = e->head_out.w_lo; r.bias = e->head_out.bias;
        r.out = ws.attn_out; r.ldo = D;
        HIP_TRY(sf_launch_rowlin(r, s));
        r.x = ws.attn_out; r.ln_g = e->head_ln.g; r.ln_b = e->head_ln.b; r.ln_eps = c.layer_norm_eps; r.N = e->I; r.act = c.hidden_act;
        r.w_hi = e->head_fc1.w_hi; r.w_lo = e->head_fc1.w_lo; r.bias = e->head_fc1.bias; r.out = ws.head_mid; r.ldo = e->I;
        HIP_TRY(sf_launch_rowlin(r, s));
        r.x = ws.head_mid; r.ldx = e->I; r.K = e->I; r.N = D; r.ln_g = r.ln_b = nullptr; r.act = -1;
        r.w_hi = e->head_fc2.w_hi; r.w_lo = e->head_fc2.w_lo; r.bias = e->head_fc2.bias; r.resid = ws.attn_out; r.ldr = D;
        r.out = pooler; r.ldo = D; r.out_ind = sp ? &sp->pooler : nullptr;
        HIP_TRY(sf_launch_rowlin(r, s));
        return SF_OK;
      }
    }
    // one row per frame from here on (F rows: 0.03 % of the forward's FLOPs): three bf16 products per operand pair in BOTH modes —
    // pooler_output is the product both loss heads and the feature dumps consume, and the bf16 mode's own head added 1.5e-2 of
    // max-abs error to it on top of what the encoder's tokens carry (tools/pool_err.py)
    HIP_TRY(run_linear(e, e->head_out, ws.pc_hi, ws.pc_lo, F, SF_EPI_F32, s, ws.attn_out, nullptr, nullptr, nullptr, 1.f, 0, 0, 0, 0, nullptr, nullptr,
                       false, nullptr, 0, nullptr, nullptr, nullptr, hacc));
    HIP_TRY(sf_launch_layernorm(ws.attn_out, e->head_ln.g, e->head_ln.b, nullptr, ws.hn_hi, (acc || hacc) ? ws.hn_lo : nullptr, F, D, c.layer_norm_eps, s));
    HIP_TRY(run_linear(e, e->head_fc1, ws.hn_hi, ws.hn_lo, F, SF_EPI_ACT_BF16, s, nullptr, ws.hm_hi, ws.hm_lo, nullptr, 1.f, 0, 0, 0, 0, nullptr, nullptr,
                       false, nullptr, 0, nullptr, nullptr, nullptr, hacc));
    HIP_TRY(run_linear(e, e->head_fc2, ws.hm_hi, ws.hm_lo, F, SF_EPI_RESID_F32, s, pooler, nullptr, nullptr, ws.attn_out, 1.f, 0, 0, 0, 0, nullptr, nullptr,
                       false, nullptr, 0, nullptr, nullptr, nullptr, hacc));
  }
  return SF_OK;
}

static int check_geometry(const sf_encoder* e, int B, int T, int H, int W, const float* pos_dev, int* N_out) {
  if (!e) return set_err(SF_ERR_INVALID, "null handle");
  const int P = e->cfg.patch_size;
  if (B <= 0 || T <= 0 || H < P || W < P) return set_err(SF_ERR_INVALID, "bad geometry B=%d T=%d H=%d W=%d", B, T, H, W);
  const int N = (H / P) * (W / P);
  if (N > 2048) return set_err(SF_ERR_INVALID, "%d patches per frame; the attention kernels handle <= 2048", N);
  if (!pos_dev && !(N == e->N && H == W))
    return set_err(SF_ERR_INVALID, "input %dx%d differs from image_size %d: pass a resized position table (pos_dev)", H, W, e->cfg.image_size);
  if ((size_t)B * T * N > (size_t)1 << 30) return set_err(SF_ERR_INVALID, "too many token rows");
  *N_out = N;
  return SF_OK;
}

extern "C" int sf_workspace_bytes(sf_encoder* e, int B, int T, int H, int W, size_t* out) {
  int N;
  const float* dummy = (const float*)1;
  int rc = check_geometry(e, B, T, H, W, dummy, &N);
  if (rc) return rc;
  if (!out) return set_err(SF_ERR_INVALID, "null out");
  if (!e->finalized) return set_err(SF_ERR_STATE, "sf_finalize_weights has not run");
  *out = carve(e, nullptr, B, T, N, true).bytes;
  return SF_OK;
}

static int forward_common(sf_encoder* e, const void* pixels, int pixel_dtype, int B, int T, int H, int W, float* last_hidden,
                          float* pooler, float* hidden_states, float* attentions, const float* pos_dev, void* workspace,
                          size_t workspace_bytes, sf_stream stream) {
  int N;
  int rc = check_geometry(e, B, T, H, W, pos_dev, &N);
  if (rc) return rc;
  if (!e->finalized) return set_err(SF_ERR_STATE, "sf_finalize_weights has not run");
  if (!pixels || !last_hidden || !workspace) return set_err(SF_ERR_INVALID, "null buffer");
  if (pixel_dtype != SF_F32 && pixel_dtype != SF_BF16 && pixel_dtype != SF_U8) return set_err(SF_ERR_INVALID, "pixels must be fp32, bf16 or uint8");
  if (T > 256) return set_err(SF_ERR_INVALID, "at most 256 frames per clip");
  Workspace ws = carve(e, workspace, B, T, N, true);
  if (ws.bytes > workspace_bytes) return set_err(SF_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, ws.bytes);
  return run_forward(e, pixels, pixel_dtype, B, T, H, W, last_hidden, pooler, hidden_states, pos_dev, ws, nullptr, T, 0,
                     false, (hipStream_t)stream, attentions);
}

extern "C" int sf_forward(sf_encoder* e, const void* pixels, int pixel_dtype, int B, int T, int H, int W,
                          float* last_hidden, float* pooler, float* hidden_states, const float* pos_dev,
                          void* workspace, size_t workspace_bytes, sf_stream stream) {
  return forward_common(e, pixels, pixel_dtype, B, T, H, W, last_hidden, pooler, hidden_states, nullptr, pos_dev, workspace,
                        workspace_bytes, stream);
}

// One forward with HIP events around the launches of four kernel classes (see sf_encoder::prof): the in-situ launch durations the
// bench's roofline reports — isolated back-to-back launches of the same kernel run on an Infinity-Cache-warm working set and read
// 4 % faster.  out_ms[2 c] = mean milliseconds per launch of class c, out_ms[2 c + 1] = launches of that class.
extern "C" int sf_forward_profile(sf_encoder* e, const void* pixels, int pixel_dtype, int B, int T, int H, int W, float* last_hidden,
                                  float* pooler, void* workspace, size_t workspace_bytes, sf_stream stream, float* out_ms) {
  if (!e || !out_ms) return set_err(SF_ERR_INVALID, "null argument");
  e->prof.clear();
  e->prof_on = true;
  int rc = forward_common(e, pixels, pixel_dtype, B, T, H, W, last_hidden, pooler, nullptr, nullptr, nullptr, workspace, workspace_bytes, stream);
  e->prof_on = false;
  hipError_t se = hipStreamSynchronize((hipStream_t)stream);
  double sum[4] = {0, 0, 0, 0};
  int cnt[4] = {0, 0, 0, 0};
  for (auto& sp : e->prof) {
    float ms = 0.f;
    if (rc == SF_OK && se == hipSuccess && hipEventElapsedTime(&ms, sp.e0, sp.e1) == hipSuccess && sp.cls >= 0 && sp.cls < 4) { sum[sp.cls] += ms; cnt[sp.cls]++; }
    (void)hipEventDestroy(sp.e0);
    (void)hipEventDestroy(sp.e1);
  }
  e->prof.clear();
  for (int c = 0; c < 4; ++c) { out_ms[2 * c] = cnt[c] ? (float)(sum[c] / cnt[c]) : 0.f; out_ms[2 * c + 1] = (float)cnt[c]; }
  if (rc != SF_OK) return rc;
  if (se != hipSuccess) return set_err(SF_ERR_HIP, "sf_forward_profile: %s", hipGetErrorString(se));
  return SF_OK;
}

extern "C" int sf_forward_attentions(sf_encoder* e, const void* pixels, int pixel_dtype, int B, int T, int H, int W,
                                     float* last_hidden, float* pooler, float* hidden_states, float* attentions,
                                     const float* pos_dev, void* workspace, size_t workspace_bytes, sf_stream stream) {
  return forward_common(e, pixels, pixel_dtype, B, T, H, W, last_hidden, pooler, hidden_states, attentions, pos_dev, workspace,
                        workspace_bytes, stream);
}

// ---- sub-module entry points: the reference's model.embeddings / model.encoder.layer[i] / post_layernorm + head
// called one at a time (adapter and classification users, modeling_timesformer_siglip_adapter.py:424-425,
// downstream/AR/models/modeling_timesformer_video_classification.py:121-134).  hidden is the caller's
// residual stream, fp32 frame-major [B,T,N,D], updated in place by sf_layers.
static int stage_common(sf_encoder* e, const void* pixels, int pixel_dtype, int B, int T, int H, int W, float* hidden,
                        float* last_hidden, float* pooler, float* attentions, const float* pos_dev, void* workspace,
                        size_t workspace_bytes, sf_stream stream, int stages, int la, int lb) {
  int N;
  int rc = check_geometry(e, B, T, H, W, pos_dev, &N);
  if (rc) return rc;
  if (!e->finalized) return set_err(SF_ERR_STATE, "sf_finalize_weights has not run");
  if (!hidden || !workspace) return set_err(SF_ERR_INVALID, "null buffer");
  if (T > 256) return set_err(SF_ERR_INVALID, "at most 256 frames per clip");
  if ((stages & 2) && (la < 0 || lb > e->L || la > lb)) return set_err(SF_ERR_INVALID, "layer range [%d, %d) outside [0, %d)", la, lb, e->L);
  Workspace ws = carve(e, workspace, B, T, N, true);
  if (ws.bytes > workspace_bytes) return set_err(SF_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, ws.bytes);
  ws.resid = hidden;                       // the caller's tensor IS the residual stream
  return run_forward(e, pixels, pixel_dtype, B, T, H, W, last_hidden, pooler, nullptr, pos_dev, ws, nullptr, T, 0, false,
                     (hipStream_t)stream, attentions, stages, la, lb);
}

extern "C" int sf_embed(sf_encoder* e, const void* pixels, int pixel_dtype, int B, int T, int H, int W, float* hidden_out,
                        const float* pos_dev, void* workspace, size_t workspace_bytes, sf_stream stream) {
  if (!pixels) return set_err(SF_ERR_INVALID, "null pixels");
  if (pixel_dtype != SF_F32 && pixel_dtype != SF_BF16 && pixel_dtype != SF_U8) return set_err(SF_ERR_INVALID, "pixels must be fp32, bf16 or uint8");
  return stage_common(e, pixels, pixel_dtype, B, T, H, W, hidden_out, nullptr, nullptr, nullptr, pos_dev, workspace, workspace_bytes,
                      stream, 1, 0, 0);
}

extern "C" int sf_layers(sf_encoder* e, float* hidden, int B, int T, int H, int W, int layer_begin, int layer_end,
                         float* attentions, void* workspace, size_t workspace_bytes, sf_stream stream) {
  // geometry only sizes the workspace here: a non-null table pointer passes the resolution check
  const float* pos_ok = e ? e->pos : nullptr;
  return stage_common(e, nullptr, SF_F32, B, T, H, W, hidden, nullptr, nullptr, attentions, pos_ok, workspace, workspace_bytes, stream,
                      2, layer_begin, layer_end);
}

extern "C" int sf_post_head(sf_encoder* e, float* hidden, int B, int T, int H, int W, float* last_hidden, float* pooler,
                            void* workspace, size_t workspace_bytes, sf_stream stream) {
  // last_hidden != NULL: post_layernorm(hidden) -> last_hidden, then the head on it -> pooler (may be NULL);
  // last_hidden == NULL: `hidden` is already normalised, the head alone -> pooler
  if (!last_hidden && !pooler) return set_err(SF_ERR_INVALID, "nothing to compute");
  return stage_common(e, nullptr, SF_F32, B, T, H, W, hidden, last_hidden, pooler, nullptr, e ? e->pos : nullptr, workspace,
                      workspace_bytes, stream, last_hidden ? 4 : 8, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// streaming
// ------------------------------------------------------------------------------------------------
extern "C" int sf_cache_create(sf_encoder* e, int B, int max_frames, int H, int W, sf_cache** out) {
  int N;
  const float* dummy = (const float*)1;
  int rc = check_geometry(e, B, 1, H, W, dummy, &N);
  if (rc) return rc;
  if (!out || max_frames <= 0 || max_frames > 256) return set_err(SF_ERR_INVALID, "max_frames must be in 1..256");
  if (!e->finalized) return set_err(SF_ERR_STATE, "sf_finalize_weights has not run");
  HIP_TRY(hipSetDevice(e->device));
  sf_cache* c = new sf_cache();
  c->enc = e; c->enc_generation = e->generation; c->B = B; c->cap = max_frames; c->H = H; c->W = W; c->N = N;
  const size_t per = (size_t)B * max_frames * N * 3 * e->D * (e->compute == SF_COMPUTE_BF16X3 ? 4 : 2);
  for (int i = 0; i < e->L; ++i) {
    void* p = nullptr;
    hipError_t err = hipMalloc(&p, per);
    if (err != hipSuccess) {
      for (void* q : c->qkv) (void)hipFree(q);
      delete c;
      return set_err(SF_ERR_HIP, "hipMalloc(%zu) for the KV-cache: %s", per, hipGetErrorString(err));
    }
    c->qkv.push_back(p);
  }
  c->bytes = per * e->L;
  if (hipMalloc((void**)&c->dparams, sizeof(SfStreamParams)) != hipSuccess) {
    for (void* q : c->qkv) (void)hipFree(q);
    delete c;
    return set_err(SF_ERR_HIP, "hipMalloc for the stream parameter block failed");
  }
  *out = c;
  return SF_OK;
}
extern "C" int sf_cache_reset(sf_cache* c) {
  if (!c) return set_err(SF_ERR_INVALID, "null cache");
  c->len = 0;
  return SF_OK;
}
extern "C" int sf_cache_length(const sf_cache* c) { return c ? c->len : 0; }
extern "C" int sf_cache_set_policy(sf_cache* c, int policy) {
  if (!c || (policy != 0 && policy != 1)) return set_err(SF_ERR_INVALID, "policy must be 0 (stop at capacity) or 1 (sliding window)");
  if (c->len != 0) return set_err(SF_ERR_STATE, "the policy of a cache is chosen while it is empty");
  c->policy = policy;
  return SF_OK;
}
extern "C" size_t sf_cache_bytes(const sf_cache* c) { return c ? c->bytes : 0; }
static void drop_graphs(sf_cache* c) {
  for (auto& kv : c->graphs)
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  c->graphs.clear();
}
extern "C" void sf_cache_destroy(sf_cache* c) {
  if (!c) return;
  drop_graphs(c);
  if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
  for (void* q : c->qkv) (void)hipFree(q);
  if (c->dparams) (void)hipFree(c->dparams);
  delete c;
}
extern "C" int sf_stream_workspace_bytes(sf_encoder* e, const sf_cache* c, int T_new, size_t* out) {
  if (!e || !c || !out || T_new <= 0) return set_err(SF_ERR_INVALID, "bad argument");
  if (c->enc != e || c->enc_generation != e->generation) return set_err(SF_ERR_STATE, "cache does not belong to this encoder's current weight packing");
  *out = carve(e, nullptr, c->B, T_new, c->N, false).bytes;
  return SF_OK;
}
static int forward_stream_impl(sf_encoder* e, sf_cache* c, const void* pixels, int pixel_dtype, int T_new,
                               float* last_hidden, float* pooler, float* hidden_states, float* attentions, const float* pos_dev,
                               void* workspace, size_t workspace_bytes, sf_stream stream) {
  if (!e || !c || c->enc != e) return set_err(SF_ERR_INVALID, "cache does not belong to this encoder");
  // a handle address can be reused after sf_destroy, and sf_finalize_weights may have changed the compute mode (cache element
  // size) or the device: the generation stamp tells a cache of an earlier packing from a current one
  if (c->enc_generation != e->generation)
    return set_err(SF_ERR_STATE, "cache was created for an earlier weight packing of this encoder (sf_finalize_weights ran since): create a new cache");
  int N;
  int rc = check_geometry(e, c->B, T_new, c->H, c->W, pos_dev, &N);
  if (rc) return rc;
  if (!pixels || !last_hidden || !workspace) return set_err(SF_ERR_INVALID, "null buffer");
  if (pixel_dtype != SF_F32 && pixel_dtype != SF_BF16 && pixel_dtype != SF_U8) return set_err(SF_ERR_INVALID, "pixels must be fp32, bf16 or uint8");
  // Positions of the call.  Sliding window (policy 1, beyond the reference, which raises at num_frames): once the cache is
  // full the new frame overwrites the oldest slot, its query sees the last `cap` frames, and frames past the time-embedding
  // table reuse its last row (the nearest-neighbour extension modeling:441-447 applies to long clips).
  const int nf = e->cfg.num_frames;
  const bool slide = c->policy == 1 && c->len + T_new > (c->cap < nf ? c->cap : nf);
  if (slide && T_new != 1)
    return set_err(SF_ERR_CAPACITY, "a sliding-window cache that is full (%d of %d frames) advances one frame per call, got %d", c->len, c->cap, T_new);
  if (!slide && c->len + T_new > c->cap)
    return set_err(SF_ERR_CAPACITY, "cache holds %d of %d frames; %d more do not fit", c->len, c->cap, T_new);
  if (!slide && c->len + T_new > nf)
    return set_err(SF_ERR_CAPACITY, "streaming needs time-embedding row %d but config.num_frames is %d", c->len + T_new - 1, nf);
  int pos3[3] = {c->len, c->len, c->len + T_new};
  if (c->policy == 1 && T_new == 1) {
    pos3[0] = c->len < nf ? c->len : nf - 1;
    pos3[1] = c->len % c->cap;
    pos3[2] = c->len + 1 < c->cap ? c->len + 1 : c->cap;
  }
  const int* pos = (c->policy == 1 && T_new == 1) ? pos3 : nullptr;
  Workspace ws = carve(e, workspace, c->B, T_new, N, false);
  if (ws.bytes > workspace_bytes) return set_err(SF_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, ws.bytes);
  hipStream_t s = (hipStream_t)stream;
  // ---- graph replay (the per-frame sequence is ~100 dependent launches of a few microseconds) -----------------
  // Not when the caller wants the per-layer hidden states (memcpy nodes to caller memory), is capturing the stream
  // itself, or asked for eager launches.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cap);
  const bool graphs_off = sf_sw(SW_DISABLE_STREAM_GRAPH) != nullptr;
  const bool use_graph = !graphs_off && !hidden_states && !attentions && cap == hipStreamCaptureStatusNone && T_new < 256 && c->len < 65536;
  if (!use_graph) {
    rc = run_forward(e, pixels, pixel_dtype, c->B, T_new, c->H, c->W, last_hidden, pooler, hidden_states, pos_dev, ws,
                     c->qkv.data(), c->cap, c->len, true, s, attentions, 7, 0, -1, 0, nullptr, pos);
    if (rc == SF_OK) c->len += T_new;
    return rc;
  }
  const int F = c->B * T_new, M = F * N;
  // One new frame per call (the streaming step proper): ONE graph per cache serves every position.  The caller's tensors
  // and the position reach the kernels through a small device block written by one tiny launch in front of the replay;
  // only the number of 64-key passes of the single-query attention is compiled in (1 / 2 / 4 -> at most three graphs).
  // Several frames per call keep one graph per (position, count): their attention kernels take the position by value.
  const bool posfree_off = sf_sw(SW_STREAM_GRAPH_PER_POSITION) != nullptr;
  const bool posfree = T_new == 1 && !posfree_off && c->dparams != nullptr;
  const int kp = (pos3[2] + 63) >> 6;
  const int kcls = kp <= 1 ? 1 : (kp <= 2 ? 2 : 4);
  const uint32_t key = posfree ? (0x80000000u | (uint32_t)kcls) : (((uint32_t)c->len << 8) | (uint32_t)T_new);
  if (!posfree && pos) return set_err(SF_ERR_STATE, "SF_STREAM_GRAPH_PER_POSITION cannot serve a sliding-window cache");
  sf_cache::GraphEntry& g = c->graphs[key];
  if (g.exec && (g.ws != workspace || g.pos != pos_dev || g.pooler != (pooler != nullptr))) {
    (void)hipGraphExecDestroy(g.exec);
    g.exec = nullptr;
  }
  const int pk = pixel_dtype == SF_U8 ? 2 : (pixel_dtype == SF_BF16 ? 1 : 0);
  if (posfree && g.exec && g.pixel_kind != pk) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
  if (!c->warmed) {
    // first call of this cache: run eagerly once so that every lazy per-kernel set-up (hipFuncSetAttribute, device
    // queries) has happened before a capture is opened
    rc = run_forward(e, pixels, pixel_dtype, c->B, T_new, c->H, c->W, last_hidden, pooler, nullptr, pos_dev, ws, c->qkv.data(),
                     c->cap, c->len, true, s, nullptr, 7, 0, -1, 0, nullptr, pos);
    if (rc == SF_OK) { c->len += T_new; c->warmed = true; }
    if (!g.exec) c->graphs.erase(key);
    return rc;
  }
  // patch extraction reads the caller's frames: outside the graphs.  For the position-free graph the same launch stores the call's
  // parameter block {pixels, outputs, position} that the graph's kernels read (it used to be a launch of its own)
  if (!posfree)
    HIP_TRY(sf_launch_patchify(pixels, pk, ws.xn_hi, ws.xn_lo, F, e->cfg.num_channels, c->H, c->W, e->cfg.patch_size, s, &e->pixel_norm, nullptr, nullptr, nullptr,
                               e->Kp));
  if (!g.exec) {
    hipGraph_t graph = nullptr;
    if (!c->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal));
    hipError_t pe = hipSuccess;
    if (posfree) {
      // representative position of the class: selects the 64-key pass count; the kernels read the live one from dparams.
      // The patch extraction stays OUTSIDE the graph (it also delivers the parameter block, below); last_hidden_state is written
      // straight into the caller's tensor through the block, pooler_output too when the head's row-vector tail runs (<= 4 streams):
      // no staging copy at the end of the graph.
      int rep_tk = kcls * 64;
      if (rep_tk > c->cap) rep_tk = c->cap;
      if (c->policy != 1 && rep_tk > nf) rep_tk = nf;
      const int rep3[3] = {0, 0, rep_tk};
      rc = run_forward(e, pixels, pixel_dtype, c->B, T_new, c->H, c->W, ws.lhs_stage, pooler ? ws.pool_stage : nullptr, nullptr, pos_dev,
                       ws, c->qkv.data(), c->cap, 0, true, c->cap_stream, nullptr, 7, 0, -1, 1, c->dparams, rep3);
      if (rc == SF_OK && pooler && !sf_head_rows_ok(e, F))
        pe = sf_launch_copy2(nullptr, nullptr, 0, ws.pool_stage, nullptr, (size_t)F * e->D, c->cap_stream, c->dparams);
    } else {
      rc = run_forward(e, pixels, pixel_dtype, c->B, T_new, c->H, c->W, ws.lhs_stage, pooler ? ws.pool_stage : nullptr, nullptr, pos_dev,
                       ws, c->qkv.data(), c->cap, c->len, true, c->cap_stream, nullptr, 7, 0, -1, 1);
    }
    hipError_t ce = hipStreamEndCapture(c->cap_stream, &graph);
    if (rc != SF_OK || pe != hipSuccess || ce != hipSuccess || !graph) {
      if (graph) (void)hipGraphDestroy(graph);
      c->graphs.erase(key);
      if (rc != SF_OK && rc != SF_ERR_HIP) return rc;
      return set_err(SF_ERR_HIP, "stream capture failed: %s", hipGetErrorString(ce != hipSuccess ? ce : pe));
    }
    hipError_t ie = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess) {
      c->graphs.erase(key);
      return set_err(SF_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ie));
    }
    g.ws = workspace; g.pos = pos_dev; g.pooler = pooler != nullptr; g.pixel_kind = pk;
  }
  if (posfree) {
    SfStreamParams v;
    v.pixels = pixels; v.lhs = last_hidden; v.pooler = pooler; v.t_row = pos3[0]; v.slot = pos3[1]; v.tk = pos3[2];
    HIP_TRY(sf_launch_patchify(pixels, pk, ws.xn_hi, ws.xn_lo, F, e->cfg.num_channels, c->H, c->W, e->cfg.patch_size, s, &e->pixel_norm, nullptr,
                               c->dparams, &v, e->Kp));
    HIP_TRY(hipGraphLaunch(g.exec, s));
  } else {
    // (Launching the embedding + first layers eagerly to cover the replay's host-side submit time measured no gain: 0.78 vs 0.77 ms.)
    HIP_TRY(hipGraphLaunch(g.exec, s));
    HIP_TRY(sf_launch_copy2(ws.lhs_stage, last_hidden, (size_t)M * e->D, pooler ? ws.pool_stage : nullptr, pooler, (size_t)F * e->D, s));
  }
  c->len += T_new;
  return SF_OK;
}

extern "C" int sf_forward_stream(sf_encoder* e, sf_cache* c, const void* pixels, int pixel_dtype, int T_new,
                                 float* last_hidden, float* pooler, float* hidden_states, const float* pos_dev, void* workspace,
                                 size_t workspace_bytes, sf_stream stream) {
  return forward_stream_impl(e, c, pixels, pixel_dtype, T_new, last_hidden, pooler, hidden_states, nullptr, pos_dev, workspace, workspace_bytes,
                             stream);
}
// output_attentions while streaming (timesformer_encoder.py:494, 557, 633, 659, 720-754): the spatial attention probabilities of
// the NEW frames, [L, B * T_new, heads, N, N] fp32, as sf_forward_attentions returns them for whole clips
extern "C" int sf_forward_stream_attentions(sf_encoder* e, sf_cache* c, const void* pixels, int pixel_dtype, int T_new,
                                            float* last_hidden, float* pooler, float* hidden_states, float* attentions,
                                            const float* pos_dev, void* workspace, size_t workspace_bytes, sf_stream stream) {
  if (!attentions) return set_err(SF_ERR_INVALID, "null attentions buffer");
  return forward_stream_impl(e, c, pixels, pixel_dtype, T_new, last_hidden, pooler, hidden_states, attentions, pos_dev, workspace,
                             workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// single operators (parity tests)
// ------------------------------------------------------------------------------------------------
__global__ void sf_combine_kernel(const bf16_t* hi, const bf16_t* lo, float* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = bf2f(hi[i]) + (lo ? bf2f(lo[i]) : 0.f);
}

extern "C" int sf_op_layernorm(const float* x, const float* gamma, const float* beta, float* y, int rows, int D,
                               float eps, sf_stream stream) {
  HIP_TRY(sf_launch_layernorm(x, gamma, beta, y, nullptr, nullptr, rows, D, eps, (hipStream_t)stream));
  return SF_OK;
}

extern "C" size_t sf_op_linear_workspace_bytes(int M, int N, int K) {
  return ((size_t)M * K * 2 + (size_t)N * K * 2 + (size_t)M * N * 2) * 2 + 4096;
}

extern "C" int sf_op_linear(const float* x, const float* w, const float* b, const float* resid, float alpha, int gelu,
                            float* y, int M, int N, int K, int compute, void* workspace, size_t workspace_bytes,
                            sf_stream stream) {
  if (workspace_bytes < sf_op_linear_workspace_bytes(M, N, K)) return set_err(SF_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const bool split = compute == SF_COMPUTE_BF16X3;
  Carver c(workspace);
  bf16_t* xh = c.take<bf16_t>((size_t)M * K); bf16_t* xl = c.take<bf16_t>((size_t)M * K);
  bf16_t* wh = c.take<bf16_t>((size_t)N * K); bf16_t* wl = c.take<bf16_t>((size_t)N * K);
  bf16_t* oh = c.take<bf16_t>((size_t)M * N); bf16_t* ol = c.take<bf16_t>((size_t)M * N);
  HIP_TRY(sf_launch_split(x, xh, split ? xl : nullptr, (size_t)M * K, s));
  HIP_TRY(sf_launch_split(w, wh, split ? wl : nullptr, (size_t)N * K, s));
  SfGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.a_hi = xh; g.a_lo = split ? xl : nullptr; g.w_hi = wh; g.w_lo = split ? wl : nullptr; g.bias = b;
  g.M = M; g.N = N; g.K = K; g.ldc = N; g.alpha = alpha; g.resid = resid; g.act = 0;
  if (gelu) {
    g.epi = SF_EPI_ACT_BF16; g.out_hi = oh; g.out_lo = split ? ol : nullptr;
    HIP_TRY(sf_launch_gemm(g, split, s));
    hipLaunchKernelGGL(sf_combine_kernel, dim3(1024), dim3(256), 0, s, oh, split ? ol : nullptr, y, (size_t)M * N);
    HIP_TRY(hipGetLastError());
  } else {
    g.epi = resid ? SF_EPI_RESID_F32 : SF_EPI_F32; g.out_f32 = y;
    HIP_TRY(sf_launch_gemm(g, split, s));
  }
  return SF_OK;
}

extern "C" size_t sf_op_attention_workspace_bytes(int groups, int L, int heads, int head_dim) {
  const size_t rows = (size_t)groups * L, D = (size_t)heads * head_dim;
  return rows * 3 * D * 2 * 2 + rows * D * 2 * 2 + 4096;     // qkv hi (+ lo) planes, ctx hi + lo
}

extern "C" int sf_op_attention(const float* qkv, float* ctx, int groups, int L, int heads, int head_dim, int causal,
                               int temporal_layout, int N_tokens, int compute, void* workspace, size_t workspace_bytes,
                               sf_stream stream) {
  if (head_dim < 8 || head_dim > 128 || head_dim % 8) return set_err(SF_ERR_INVALID, "head_dim must be a multiple of 8 in 8..128");
  if (workspace_bytes < sf_op_attention_workspace_bytes(groups, L, heads, head_dim)) return set_err(SF_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const bool acc = compute == SF_COMPUTE_BF16X3;
  const int D = heads * head_dim;
  const size_t rows = (size_t)groups * L;
  Carver c(workspace);
  bf16_t* ch = c.take<bf16_t>(rows * D);
  bf16_t* cl = c.take<bf16_t>(rows * D);
  bf16_t* qb = c.take<bf16_t>(rows * 3 * D);
  const void* base = qkv;
  size_t esz = 4;
  const bool planes = acc && head_dim == 64 && (temporal_layout ? sf_temporal_planes_ok(L, L) : sf_spatial_planes_ok(L, false));
  bf16_t* ql = nullptr;
  if (!acc || planes) {
    if (planes) ql = c.take<bf16_t>(rows * 3 * D);
    HIP_TRY(sf_launch_split(qkv, qb, ql, rows * 3 * D, s));
    base = qb;
    esz = 2;
  }
  SfAttnArgs a;
  memset(&a, 0, sizeof(a));
  a.q = base; a.k = (const char*)base + (size_t)D * esz; a.v = (const char*)base + (size_t)2 * D * esz;
  a.in_is_f32 = acc && !planes; a.lo_plane_off = planes ? (long long)(ql - qb) : 0;
  a.row_pitch_q = 3 * D; a.row_pitch_kv = 3 * D; a.heads = heads; a.scale = 1.0f / sqrtf((float)head_dim);
  a.ctx_hi = ch; a.ctx_lo = cl; a.D = D; a.head_dim = head_dim;
  if (temporal_layout) {
    // rows are [B, L, N_tokens, 3D] with groups = B * N_tokens sequences of length L
    if (N_tokens <= 0 || groups % N_tokens) return set_err(SF_ERR_INVALID, "groups must be a multiple of N_tokens");
    a.N = N_tokens; a.B = groups / N_tokens; a.Tq = L; a.Tk = L; a.Tcap = L; a.t_past = 0; a.causal = causal;
    a.Tq_cap = L; a.q_t0 = 0;
    HIP_TRY(sf_launch_temporal_attention(a, acc, s));
  } else {
    if (causal) return set_err(SF_ERR_INVALID, "the spatial layout has no causal variant");
    a.N = L; a.frames = groups;
    HIP_TRY(sf_launch_spatial_attention(a, acc, s));
  }
  hipLaunchKernelGGL(sf_combine_kernel, dim3(1024), dim3(256), 0, s, ch, acc ? cl : nullptr, ctx, rows * D);
  HIP_TRY(hipGetLastError());
  return SF_OK;
}

// ------------------------------------------------------------------------------------------------
// loss heads
// ------------------------------------------------------------------------------------------------
extern "C" size_t sf_loss_workspace_bytes(int B, int T) {
  return sf_loss_partial_bytes(B > 0 && T > 0 ? B * T : 1);
}
extern "C" int sf_retrieval_loss(const float* pooler, const float* text, int B, int T, int D, int Bt, int pos_offset,
                                 const float* logit_scale, const float* logit_bias, float* loss, float* grad_pooler,
                                 float* grad_scalars, void* workspace, size_t workspace_bytes, sf_stream stream) {
  if (!pooler || !text || !loss || !logit_scale || !logit_bias || !workspace) return set_err(SF_ERR_INVALID, "sf_retrieval_loss: null buffer");
  if (B <= 0 || Bt <= 0 || T <= 0 || D <= 0) return set_err(SF_ERR_INVALID, "sf_retrieval_loss: bad shape B=%d Bt=%d T=%d D=%d", B, Bt, T, D);
  if (D > 2048) return set_err(SF_ERR_CAPACITY, "sf_retrieval_loss: feature width %d > 2048 (256 threads x 8 features)", D);
  if (pos_offset >= 0 && pos_offset + B > Bt)
    return set_err(SF_ERR_INVALID, "sf_retrieval_loss: positives %d..%d fall outside the %d text rows", pos_offset, pos_offset + B - 1, Bt);
  if (workspace_bytes < sf_loss_partial_bytes(B)) return set_err(SF_ERR_WORKSPACE, "loss workspace %zu < %zu bytes", workspace_bytes, sf_loss_partial_bytes(B));
  HIP_TRY(sf_launch_retrieval_loss(pooler, text, B, T, D, Bt, pos_offset, logit_scale, logit_bias, loss, grad_pooler,
                                   grad_scalars, (float*)workspace, (hipStream_t)stream));
  return SF_OK;
}
extern "C" int sf_localization_loss(const float* pooler, const float* label_emb, const int32_t* labels, int B, int T,
                                    int D, int L, const float* logit_scale, const float* logit_bias, float* loss,
                                    float* grad_pooler, float* grad_scalars, void* workspace, size_t workspace_bytes,
                                    sf_stream stream) {
  if (!pooler || !label_emb || !labels || !loss || !logit_scale || !logit_bias || !workspace) return set_err(SF_ERR_INVALID, "sf_localization_loss: null buffer");
  if (B <= 0 || T <= 0 || D <= 0 || L <= 0) return set_err(SF_ERR_INVALID, "sf_localization_loss: bad shape B=%d T=%d D=%d L=%d", B, T, D, L);
  if (L > 4096) return set_err(SF_ERR_CAPACITY, "sf_localization_loss: %d label classes > 4096 (one LDS row of similarities per frame)", L);
  if (workspace_bytes < sf_loss_partial_bytes(B * T)) return set_err(SF_ERR_WORKSPACE, "loss workspace %zu < %zu bytes", workspace_bytes, sf_loss_partial_bytes(B * T));
  HIP_TRY(sf_launch_localization_loss(pooler, label_emb, labels, B, T, D, L, logit_scale, logit_bias, loss, grad_pooler,
                                      grad_scalars, (float*)workspace, (hipStream_t)stream));
  return SF_OK;
}

// ------------------------------------------------------------------------------------------------
// bench hook: time the dominant GEMM with HIP events on the caller's stream
// ------------------------------------------------------------------------------------------------
extern "C" int sf_bench_gemm(sf_encoder* e, int M, int which, int iters, void* workspace, size_t workspace_bytes,
                             sf_stream stream, float* mean_ms_out, double* flops_out) {
  if (!e || !e->finalized || e->layers.empty()) return set_err(SF_ERR_STATE, "encoder not finalized");
  if (iters <= 0 || M <= 0 || !workspace || !mean_ms_out) return set_err(SF_ERR_INVALID, "bad argument");
  const DevLayer& l = e->layers[0];
  // which 0..3 = MLP-up, MLP-down, qkv, attention out-proj with the epilogues the forward gives them; 4 / 5 = out-proj / MLP-down PLAIN (bf16
  // output, no residual, no row sums): the like-for-like partner of a vendor-library GEMM (bench.py roofline.yardstick_tflops)
  const bool plain = which == 4 || which == 5;
  if (plain) which = which == 4 ? 3 : 1;
  const DevLinear* lin = which == 0 ? &l.up : which == 1 ? &l.down : which == 2 ? &l.s_qkv : &l.s_out;
  const bool acc = e->compute == SF_COMPUTE_BF16X3;
  Carver c(workspace);
  bf16_t* ah = c.take<bf16_t>((size_t)M * lin->K);
  bf16_t* al = c.take<bf16_t>((size_t)M * lin->K);
  bf16_t* oh = c.take<bf16_t>((size_t)M * lin->N);
  bf16_t* ol = c.take<bf16_t>((size_t)M * lin->N);
  float* of = c.take<float>((size_t)M * lin->N);
  if (c.off > workspace_bytes) return set_err(SF_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, c.off);
  hipStream_t s = (hipStream_t)stream;
  const int epi = plain ? (acc ? SF_EPI_F32 : SF_EPI_BF16) : which == 0 ? SF_EPI_ACT_BF16 : which == 2 ? (acc ? SF_EPI_F32 : SF_EPI_BF16) : SF_EPI_RESID_F32;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  if (acc && epi == SF_EPI_RESID_F32) oh = nullptr;     // the accurate mode keeps no bf16 copy of the residual (no LayerNorm fold)
  // bf16 mode at BASELINE-sized M: the residual projections as the forward launches them — hi + lo planes in and out, LayerNorm
  // row sums of the next Linear (run_forward's `pm`); fp32 residual + bf16 copy otherwise
  const bool planes_off = sf_sw(SW_DISABLE_RESID_PLANES) != nullptr;
  const bool pm = !planes_off && !acc && epi == SF_EPI_RESID_F32 && ln_fold_ok(e, M) && !ln_fold_tile_ok(e, M);
  float* st = pm ? c.take<float>((size_t)M * 8) : nullptr;
  if (c.off > workspace_bytes) return set_err(SF_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, c.off);
  if (st) HIP_TRY(hipMemsetAsync(st, 0, (size_t)M * 8 * sizeof(float), (hipStream_t)stream));
  auto go = [&]() {
    return pm ? run_linear(e, *lin, ah, al, M, epi, s, nullptr, oh, ol, nullptr, 0.f, 0, 0, 0, 0, nullptr, st, false, nullptr, 0, oh, ol)
              : run_linear(e, *lin, ah, al, M, epi, s, of, oh, ol, of, 0.f);
  };
  HIP_TRY(go());   // warm
  HIP_TRY(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) HIP_TRY(go());
  HIP_TRY(hipEventRecord(e1, s));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *mean_ms_out = ms / iters;
  if (flops_out) *flops_out = 2.0 * M * (double)lin->N * (double)lin->K;
  return SF_OK;
}

extern "C" int sf_bench_attention(sf_encoder* e, int B, int T, int which, int iters, void* workspace,
                                  size_t workspace_bytes, sf_stream stream, float* mean_ms_out, double* bytes_out,
                                  double* flops_out) {
  if (!e || !e->finalized) return set_err(SF_ERR_STATE, "encoder not finalized");
  if (iters <= 0 || B <= 0 || T <= 0 || !workspace || !mean_ms_out) return set_err(SF_ERR_INVALID, "bad argument");
  const bool acc = e->compute == SF_COMPUTE_BF16X3;
  const int D = e->D, N = e->N, heads = e->cfg.num_attention_heads;
  const size_t M = (size_t)B * T * N, esz = acc ? 4 : 2;
  Carver c(workspace);
  char* qkv = c.take<char>(M * 3 * D * esz);
  bf16_t* ch = c.take<bf16_t>(M * D);
  bf16_t* cl = c.take<bf16_t>(M * D);
  if (c.off > workspace_bytes) return set_err(SF_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, c.off);
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(qkv, 0x3c, M * 3 * D * esz, s));   // finite, non-trivial bit pattern
  SfAttnArgs a;
  memset(&a, 0, sizeof(a));
  a.q = qkv; a.k = qkv + (size_t)D * esz; a.v = qkv + (size_t)2 * D * esz;
  a.in_is_f32 = acc; a.row_pitch_q = 3 * D; a.row_pitch_kv = 3 * D; a.heads = heads; a.scale = 0.125f;
  a.ctx_hi = ch; a.ctx_lo = cl; a.D = D; a.N = N;
  if (which == 0 && acc && sf_spatial_planes_ok(N, false)) {      // what the forward does: hi + lo planes in the fp32 tensor's bytes
    a.k = qkv + (size_t)D * 2; a.v = qkv + (size_t)2 * D * 2;
    a.in_is_f32 = 0; a.lo_plane_off = (long long)M * 3 * D;
  }
  if (which == 1 && acc && sf_temporal_planes_ok(T, T)) {
    a.k = qkv + (size_t)D * 2; a.v = qkv + (size_t)2 * D * 2;
    a.in_is_f32 = 0; a.lo_plane_off = (long long)M * 3 * D;
  }
  if (which == 0) {
    a.frames = B * T;
  } else {
    a.B = B; a.Tq = T; a.Tk = T; a.Tcap = T; a.t_past = 0; a.causal = e->cfg.enable_causal_temporal; a.Tq_cap = T; a.q_t0 = 0;
  }
  auto launch = [&]() { return which == 0 ? sf_launch_spatial_attention(a, acc, s) : sf_launch_temporal_attention(a, acc, s); };
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(launch());
  HIP_TRY(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) HIP_TRY(launch());
  HIP_TRY(hipEventRecord(e1, s));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *mean_ms_out = ms / iters;
  if (bytes_out) *bytes_out = (double)M * 3 * D * esz + (double)M * D * 2 * (acc ? 2 : 1);
  if (flops_out) {
    const double L = which == 0 ? N : T;        // 2 matmuls of L x L x 64 per (sequence, head), full (not causal-halved)
    const double seqs = which == 0 ? (double)B * T : (double)B * N;
    *flops_out = seqs * heads * 2.0 * (2.0 * L * L * 64.0);
  }
  return SF_OK;
}

// ------------------------------------------------------------------------------------------------
// launch floor (bench hook): a graph of dependent empty kernels
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sf_null_kernel(int* sink) {
  if (sink && threadIdx.x == 1024) *sink = 0;     // never true: the kernel has no memory traffic
}
extern "C" int sf_bench_launch_floor(int device, int launches, int iters, sf_stream stream, float* us_per_launch_out) {
  if (launches <= 0 || iters <= 0 || !us_per_launch_out) return set_err(SF_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  hipStream_t cap = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipError_t err = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
  if (err == hipSuccess) {
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(sf_null_kernel, dim3(256), dim3(256), 0, cap, (int*)nullptr);
    err = hipStreamEndCapture(cap, &graph);
  }
  if (err == hipSuccess) err = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  float ms = 0.f;
  if (err == hipSuccess) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipGraphLaunch(exec, s);
    (void)hipStreamSynchronize(s);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) (void)hipGraphLaunch(exec, s);
    (void)hipEventRecord(e1, s);
    err = hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  if (exec) (void)hipGraphExecDestroy(exec);
  if (graph) (void)hipGraphDestroy(graph);
  (void)hipStreamDestroy(cap);
  if (err != hipSuccess) return set_err(SF_ERR_HIP, "sf_bench_launch_floor: %s", hipGetErrorString(err));
  *us_per_launch_out = 1e3f * ms / ((float)iters * (float)launches);
  return SF_OK;
}
