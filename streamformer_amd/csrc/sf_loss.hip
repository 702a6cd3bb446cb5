// SigLIP sigmoid-loss heads of the multitask pre-training step (BASELINE config #3):
//   retrieval     TimesformerVideoRetrievalHead.forward + SigLipLoss._loss
//                 (reference modeling:2324-2351, 221-237)
//   localization  TimesformerUniversalLocalizationHead.forward, training branch (modeling:2238-2282)
// Both return the loss, d loss / d pooler_output and d loss / d (logit_scale, logit_bias).  The work
// is a few MFLOP, so each is ONE 256-thread workgroup with a fixed reduction order: deterministic
// (no float atomics) and launch-latency bound.
#include "sf_common.h"

SF_DEVICE float log_sigmoid(float x) {  // matches F.logsigmoid: min(x,0) - log1p(exp(-|x|))
  return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}
SF_DEVICE float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__device__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return r;
}

// logits z[i,j] = s * <img_i/|img_i|, txt_j/|txt_j|> + b ; label +1 iff j == pos_offset + i
__global__ __launch_bounds__(256) void sf_retrieval_loss_kernel(const float* __restrict__ pooler,
                                                                const float* __restrict__ text, int B, int T,
                                                                int D, int Bt, int pos_offset, float logit_scale,
                                                                float logit_bias, float* __restrict__ loss,
                                                                float* __restrict__ grad_pooler,
                                                                float* __restrict__ grad_scalars) {
  extern __shared__ float sm[];
  float* inorm = sm;               // [B]
  float* tnorm = inorm + B;        // [Bt]
  float* sim = tnorm + Bt;         // [B*Bt]  cosine similarity
  float* dz = sim + B * Bt;        // [B*Bt]  dL/dz
  float* red = dz + B * Bt;        // [4]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float s = expf(logit_scale);
  for (int r = wave; r < B + Bt; r += 4) {
    const float* x = r < B ? pooler + ((size_t)r * T + (T - 1)) * D : text + (size_t)(r - B) * D;
    float a = 0.f;
    for (int d = lane; d < D; d += 64) a = fmaf(x[d], x[d], a);
    a = wave_sum(a);
    if (lane == 0) (r < B ? inorm[r] : tnorm[r - B]) = sqrtf(a);
  }
  __syncthreads();
  float lsum = 0.f, gs = 0.f, gb = 0.f;
  for (int pr = wave; pr < B * Bt; pr += 4) {
    const int i = pr / Bt, j = pr % Bt;
    const float* x = pooler + ((size_t)i * T + (T - 1)) * D;
    const float* y = text + (size_t)j * D;
    float a = 0.f;
    for (int d = lane; d < D; d += 64) a = fmaf(x[d], y[d], a);
    a = wave_sum(a) / (inorm[i] * tnorm[j]);
    const float z = s * a + logit_bias;
    const float lab = (pos_offset >= 0 && j == pos_offset + i) ? 1.f : -1.f;
    const float g = -lab * sigmoidf(-lab * z) / (float)B;
    if (lane == 0) {
      sim[pr] = a;
      dz[pr] = g;
      lsum += -log_sigmoid(lab * z) / (float)B;
      gs += g * s * a;
      gb += g;
    }
  }
  lsum = block_sum(lsum, red);
  gs = block_sum(gs, red);
  gb = block_sum(gb, red);
  if (threadIdx.x == 0) {
    loss[0] = lsum;
    if (grad_scalars) { grad_scalars[0] = gs; grad_scalars[1] = gb; }
  }
  if (!grad_pooler) return;
  for (size_t i = threadIdx.x; i < (size_t)B * T * D; i += 256) grad_pooler[i] = 0.f;
  __syncthreads();
  // dL/dx_i = (g_i - ihat_i <ihat_i, g_i>) / |x_i|, g_i = sum_j dz_ij * s * that_j ; <ihat_i,g_i> = s*sum_j dz_ij*sim_ij
  for (int i = 0; i < B; ++i) {
    float dotg = 0.f;
    for (int j = 0; j < Bt; ++j) dotg += dz[i * Bt + j] * s * sim[i * Bt + j];
    const float* x = pooler + ((size_t)i * T + (T - 1)) * D;
    for (int d = threadIdx.x; d < D; d += 256) {
      float g = 0.f;
      for (int j = 0; j < Bt; ++j) g = fmaf(dz[i * Bt + j] * s / tnorm[j], text[(size_t)j * D + d], g);
      grad_pooler[((size_t)i * T + (T - 1)) * D + d] = (g - x[d] / inorm[i] * dotg) / inorm[i];
    }
  }
}

hipError_t sf_launch_retrieval_loss(const float* pooler, const float* text, int B, int T, int D, int Bt,
                                    int pos_offset, float logit_scale, float logit_bias, float* loss,
                                    float* grad_pooler, float* grad_scalars, hipStream_t s) {
  if (B <= 0 || Bt <= 0 || T <= 0 || D <= 0 || (size_t)B * Bt > 4096) return hipErrorInvalidValue;
  const size_t lds = (size_t)(B + Bt + 2 * B * Bt + 4) * sizeof(float);
  hipLaunchKernelGGL(sf_retrieval_loss_kernel, dim3(1), dim3(256), lds, s, pooler, text, B, T, D, Bt, pos_offset,
                     logit_scale, logit_bias, loss, grad_pooler, grad_scalars);
  return hipGetLastError();
}

// per frame (b,t): z[l] = s * <p/|p|, E_l> + bias ; target +1 at labels[b,t] (if >= 0) else -1
// loss = mean_b( -sum_{t,l} logsigmoid(y z) / T )
// One workgroup per frame row writes {loss, d scale, d bias} partials; a finish workgroup adds them in
// row order (deterministic).
__global__ __launch_bounds__(256) void sf_localization_loss_kernel(const float* __restrict__ pooler,
                                                                   const float* __restrict__ label_emb,
                                                                   const int* __restrict__ labels, int B, int T,
                                                                   int D, int L, float logit_scale, float logit_bias,
                                                                   float* __restrict__ partial,
                                                                   float* __restrict__ grad_pooler) {
  extern __shared__ float sm[];
  float* simr = sm;          // [L] similarity of the row
  float* dzr = simr + L;     // [L]
  float* red = dzr + L;      // [4]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float s = expf(logit_scale);
  const float wgt = 1.f / ((float)T * (float)B);
  float lsum = 0.f, gs = 0.f, gb = 0.f;
  const int row = blockIdx.x;
  const float* x = pooler + (size_t)row * D;
  float a = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) a = fmaf(x[d], x[d], a);
  a = block_sum(a, red);
  const float n = sqrtf(a);
  const int lab = labels[row];
  for (int l = wave; l < L; l += 4) {
    const float* e = label_emb + (size_t)l * D;
    float c = 0.f;
    for (int d = lane; d < D; d += 64) c = fmaf(x[d], e[d], c);
    c = wave_sum(c) / n;
    const float z = s * c + logit_bias;
    const float y = (lab >= 0 && lab == l) ? 1.f : -1.f;
    const float g = -y * sigmoidf(-y * z) * wgt;
    if (lane == 0) {
      simr[l] = c;
      dzr[l] = g;
      lsum += -log_sigmoid(y * z) * wgt;
      gs += g * s * c;
      gb += g;
    }
  }
  __syncthreads();
  if (grad_pooler) {
    float dotg = 0.f;
    for (int l = 0; l < L; ++l) dotg += dzr[l] * s * simr[l];
    for (int d = threadIdx.x; d < D; d += 256) {
      float g = 0.f;
      for (int l = 0; l < L; ++l) g = fmaf(dzr[l] * s, label_emb[(size_t)l * D + d], g);
      grad_pooler[(size_t)row * D + d] = (g - x[d] / n * dotg) / n;
    }
  }
  lsum = block_sum(lsum, red);
  gs = block_sum(gs, red);
  gb = block_sum(gb, red);
  if (threadIdx.x == 0) {
    partial[row * 3 + 0] = lsum;
    partial[row * 3 + 1] = gs;
    partial[row * 3 + 2] = gb;
  }
}

__global__ __launch_bounds__(64) void sf_localization_loss_finish_kernel(const float* __restrict__ partial, int rows,
                                                                         float* __restrict__ loss,
                                                                         float* __restrict__ grad_scalars) {
  if (threadIdx.x != 0) return;
  float l = 0.f, gs = 0.f, gb = 0.f;
  for (int r = 0; r < rows; ++r) {
    l += partial[r * 3 + 0];
    gs += partial[r * 3 + 1];
    gb += partial[r * 3 + 2];
  }
  loss[0] = l;
  if (grad_scalars) { grad_scalars[0] = gs; grad_scalars[1] = gb; }
}

// per-device scratch for the row partials (grown on demand; the loss heads take no workspace argument)
static float* loc_scratch(int rows) {
  static float* buf[64] = {nullptr};
  static int cap[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (cap[dev] < rows) {
    if (buf[dev]) (void)hipFree(buf[dev]);
    buf[dev] = nullptr;
    const int want = rows < 4096 ? 4096 : rows;
    if (hipMalloc(&buf[dev], (size_t)want * 3 * sizeof(float)) != hipSuccess) { cap[dev] = 0; return nullptr; }
    cap[dev] = want;
  }
  return buf[dev];
}

hipError_t sf_launch_localization_loss(const float* pooler, const float* label_emb, const int* labels,
                                       int B, int T, int D, int L, float logit_scale, float logit_bias,
                                       float* loss, float* grad_pooler, float* grad_scalars, hipStream_t s) {
  if (B <= 0 || T <= 0 || D <= 0 || L <= 0 || L > 4096) return hipErrorInvalidValue;
  float* partial = loc_scratch(B * T);
  if (!partial) return hipErrorOutOfMemory;
  const size_t lds = (size_t)(2 * L + 8) * sizeof(float);
  hipLaunchKernelGGL(sf_localization_loss_kernel, dim3(B * T), dim3(256), lds, s, pooler, label_emb, labels, B, T, D, L,
                     logit_scale, logit_bias, partial, grad_pooler);
  hipLaunchKernelGGL(sf_localization_loss_finish_kernel, dim3(1), dim3(64), 0, s, partial, B * T, loss, grad_scalars);
  return hipGetLastError();
}
