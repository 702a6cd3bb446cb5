// SigLIP sigmoid-loss heads of the multitask pre-training step (BASELINE config #3):
//   retrieval     TimesformerVideoRetrievalHead.forward + SigLipLoss._loss
//                 (reference modeling:2324-2351, 221-237)
//   localization  TimesformerUniversalLocalizationHead.forward, training branch (modeling:2238-2282)
// Both return the loss, d loss / d pooler_output and d loss / d (logit_scale, logit_bias).  One workgroup per
// image / frame row + a one-wave finish with a fixed reduction order: deterministic (no float atomics).  The two
// scalars are read from device memory and the row partials live in the caller's workspace, so a training step
// has no host synchronisation and no library-side allocation here.
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
// One workgroup per image row i; the text table is walked in chunks of SF_RET_CHUNK rows (two passes per chunk:
// similarities / dL/dz by wave-per-column, then the gradient row by thread-per-feature), so there is no limit
// on Bt = world * B (the one-workgroup version stopped at B * Bt = 4096).  logit_scale / logit_bias are
// read from DEVICE memory (the trainer's parameter buffer): no host round trip between forward and backward.
// Row partials {loss, d scale, d bias} go to the caller's workspace; the finish kernel adds them in row order.
#define SF_RET_CHUNK 1024
#define SF_RET_MAXD 8            // features per thread: D <= 256 * 8
__global__ __launch_bounds__(256) void sf_retrieval_loss_kernel(const float* __restrict__ pooler,
                                                                const float* __restrict__ text, int B, int T,
                                                                int D, int Bt, int pos_offset,
                                                                const float* __restrict__ logit_scale_p,
                                                                const float* __restrict__ logit_bias_p,
                                                                float* __restrict__ partial,
                                                                float* __restrict__ grad_pooler) {
  __shared__ float dzs[SF_RET_CHUNK];    // dL/dz_ij * s / |t_j| of the chunk's columns
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x;
  const float s = expf(logit_scale_p[0]), bias = logit_bias_p[0];
  const float* x = pooler + ((size_t)i * T + (T - 1)) * D;
  float a = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) a = fmaf(x[d], x[d], a);
  const float inorm = sqrtf(block_sum(a, red));
  float lsum = 0.f, gs = 0.f, gb = 0.f;
  float g[SF_RET_MAXD];
#pragma unroll
  for (int k = 0; k < SF_RET_MAXD; ++k) g[k] = 0.f;
  for (int j0 = 0; j0 < Bt; j0 += SF_RET_CHUNK) {
    const int nj = min(SF_RET_CHUNK, Bt - j0);
    for (int j = wave; j < nj; j += 4) {
      const float* y = text + (size_t)(j0 + j) * D;
      float dot = 0.f, yy = 0.f;
      for (int d = lane; d < D; d += 64) {
        const float yv = y[d];
        dot = fmaf(x[d], yv, dot);
        yy = fmaf(yv, yv, yy);
      }
      dot = wave_sum(dot);
      const float tnorm = sqrtf(wave_sum(yy));
      const float c = dot / (inorm * tnorm);
      const float z = s * c + bias;
      const float lab = (pos_offset >= 0 && j0 + j == pos_offset + i) ? 1.f : -1.f;
      const float gz = -lab * sigmoidf(-lab * z) / (float)B;
      if (lane == 0) {
        dzs[j] = gz * s / tnorm;
        lsum += -log_sigmoid(lab * z) / (float)B;
        gs += gz * s * c;
        gb += gz;
      }
    }
    __syncthreads();
    if (grad_pooler) {
      // g_i += sum_j dz_ij * s * that_j   (thread-per-feature; the text rows stream through L2)
      for (int j = 0; j < nj; ++j) {
        const float w = dzs[j];
        const float* y = text + (size_t)(j0 + j) * D;
#pragma unroll
        for (int k = 0; k < SF_RET_MAXD; ++k) {
          const int d = threadIdx.x + k * 256;
          if (d < D) g[k] = fmaf(w, y[d], g[k]);
        }
      }
    }
    __syncthreads();
  }
  if (grad_pooler) {
    // <ihat_i, g_i> as a block reduction over the finished gradient row
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < SF_RET_MAXD; ++k) {
      const int d = threadIdx.x + k * 256;
      if (d < D) part = fmaf(x[d], g[k], part);
    }
    const float ihat_dot_g = block_sum(part, red) / inorm;
    float* gp = grad_pooler + (size_t)i * T * D;
    for (size_t e = threadIdx.x; e < (size_t)(T - 1) * D; e += 256) gp[e] = 0.f;     // only the last frame feeds the loss
#pragma unroll
    for (int k = 0; k < SF_RET_MAXD; ++k) {
      const int d = threadIdx.x + k * 256;
      // dL/dx_i = (g_i - ihat_i <ihat_i, g_i>) / |x_i|
      if (d < D) gp[(size_t)(T - 1) * D + d] = (g[k] - x[d] / inorm * ihat_dot_g) / inorm;
    }
  }
  lsum = block_sum(lsum, red);
  gs = block_sum(gs, red);
  gb = block_sum(gb, red);
  if (threadIdx.x == 0) {
    partial[i * 3 + 0] = lsum;
    partial[i * 3 + 1] = gs;
    partial[i * 3 + 2] = gb;
  }
}

// per frame (b,t): z[l] = s * <p/|p|, E_l> + bias ; target +1 at labels[b,t] (if >= 0) else -1
// loss = mean_b( -sum_{t,l} logsigmoid(y z) / T )
// One workgroup per frame row writes {loss, d scale, d bias} partials; a finish workgroup adds them in
// row order (deterministic).
__global__ __launch_bounds__(256) void sf_localization_loss_kernel(const float* __restrict__ pooler,
                                                                   const float* __restrict__ label_emb,
                                                                   const int* __restrict__ labels, int B, int T,
                                                                   int D, int L,
                                                                   const float* __restrict__ logit_scale_p,
                                                                   const float* __restrict__ logit_bias_p,
                                                                   float* __restrict__ partial,
                                                                   float* __restrict__ grad_pooler) {
  extern __shared__ float sm[];
  float* simr = sm;          // [L] similarity of the row
  float* dzr = simr + L;     // [L]
  float* red = dzr + L;      // [4]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float s = expf(logit_scale_p[0]), logit_bias = logit_bias_p[0];
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

// one wave adds the row partials in a fixed order (lane-strided sums, then the DPP tree): deterministic
__global__ __launch_bounds__(64) void sf_loss_finish_kernel(const float* __restrict__ partial, int rows,
                                                            float* __restrict__ loss,
                                                            float* __restrict__ grad_scalars) {
  float l = 0.f, gs = 0.f, gb = 0.f;
  for (int r = threadIdx.x; r < rows; r += 64) {
    l += partial[r * 3 + 0];
    gs += partial[r * 3 + 1];
    gb += partial[r * 3 + 2];
  }
  l = wave_sum(l);
  gs = wave_sum(gs);
  gb = wave_sum(gb);
  if (threadIdx.x == 0) {
    loss[0] = l;
    if (grad_scalars) { grad_scalars[0] = gs; grad_scalars[1] = gb; }
  }
}

size_t sf_loss_partial_bytes(int rows) { return (size_t)(rows > 0 ? rows : 1) * 3 * sizeof(float); }

hipError_t sf_launch_retrieval_loss(const float* pooler, const float* text, int B, int T, int D, int Bt,
                                    int pos_offset, const float* logit_scale, const float* logit_bias, float* loss,
                                    float* grad_pooler, float* grad_scalars, float* partial, hipStream_t s) {
  hipLaunchKernelGGL(sf_retrieval_loss_kernel, dim3(B), dim3(256), 0, s, pooler, text, B, T, D, Bt, pos_offset,
                     logit_scale, logit_bias, partial, grad_pooler);
  hipLaunchKernelGGL(sf_loss_finish_kernel, dim3(1), dim3(64), 0, s, partial, B, loss, grad_scalars);
  return hipGetLastError();
}

hipError_t sf_launch_localization_loss(const float* pooler, const float* label_emb, const int* labels,
                                       int B, int T, int D, int L, const float* logit_scale, const float* logit_bias,
                                       float* loss, float* grad_pooler, float* grad_scalars, float* partial,
                                       hipStream_t s) {
  const size_t lds = (size_t)(2 * L + 8) * sizeof(float);
  hipLaunchKernelGGL(sf_localization_loss_kernel, dim3(B * T), dim3(256), lds, s, pooler, label_emb, labels, B, T, D, L,
                     logit_scale, logit_bias, partial, grad_pooler);
  hipLaunchKernelGGL(sf_loss_finish_kernel, dim3(1), dim3(64), 0, s, partial, B * T, loss, grad_scalars);
  return hipGetLastError();
}
