// Attention-pooling head with the key / value projections folded away (exact algebra).
//
// Reference: TimesformerSiglipMultiheadAttentionPoolingHead.forward (modeling:1141-1154) runs nn.MultiheadAttention with a
// CONSTANT query (the learned probe) against the N tokens x_n of a frame.  With q_h the projected, scaled query of head h:
//
//   score_hn = q_h . (Wk_h x_n + bk_h) = (Wk_h^T q_h) . x_n + const_h      -> softmax over n drops const_h
//   ctx_h    = sum_n p_hn (Wv_h x_n + bv_h) = Wv_h (sum_n p_hn x_n) + bv_h  (sum_n p_hn = 1)
//
// so with U_h = Wk_h^T q_h  ([heads, D], prepared once per weight update) a frame needs
//   scores  [N x D] . [D x heads]          (MFMA, bf16x3 on fp32 tokens split in registers)
//   z_h     = sum_n p_hn x_n                (fp32 VALU, one pass over the frame's tokens)
//   ctx_h   = Wv_h z_h + bv_h               (MFMA, bf16x3: 12 [64 x D] mat-vecs per frame, batched over 16 frames per tile)
// instead of projecting all M = B T N tokens to K and V (2 M D 2D FLOP: 59 GFLOP at B = 8, a 77 MB [M, 2D] tensor).
// The token rows are read as fp32 (post_layernorm's output), so the head adds no operand rounding of its own in either
// compute mode.  Token splits (S > 1) give a single streamed frame enough workgroups; their partial sums are combined
// flash-decoding style by the ctx kernel.
//
// Backward (training step): dz_h = Wv_h^T dctx_h, dWv_h += dctx_h z_h^T, dbv += dctx; dp_hn = dz_h . x_n,
// ds_hn = p_hn (dp_hn - dz_h . z_h), dx_n = sum_h (p_hn dz_h + ds_hn U_h), dU_h = sum_n ds_hn x_n (a [32 x D] weight-gradient
// GEMM over all token rows), dWk_h += q_h dU_h^T, dq_h = Wk_h dU_h; bk receives exactly zero (its term cancels in the softmax).
#include "sf_common.h"
#include "sf_pool_head.h"

typedef __attribute__((ext_vector_type(8))) __bf16 v8bf_t;
SF_DEVICE f32x4_t pmfma(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf_t, a), __builtin_bit_cast(v8bf_t, b), c, 0, 0, 0);
}
// result layout of pmfma(A, B): lane (l15 = lane & 15, g = lane >> 4) holds C[A-row 4g + j][B-row l15], j = 0..3;
// operand fragment: lane holds row l15, k = 8g .. 8g + 7 of the 32-deep k-step

SF_DEVICE void split8(f32x4_t a, f32x4_t b, bf16x8_t& hi, bf16x8_t& lo) {
  unsigned h[8], l[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { split_bf(a[i], h[i], l[i]); split_bf(b[i], h[4 + i], l[4 + i]); }
  const u32x4_t hv = {h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
  const u32x4_t lv = {l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
  hi = __builtin_bit_cast(bf16x8_t, hv);
  lo = __builtin_bit_cast(bf16x8_t, lv);
}
SF_DEVICE bf16x8_t cvt8(f32x4_t a, f32x4_t b) {
  const u32x4_t hv = {pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
  return __builtin_bit_cast(bf16x8_t, hv);
}
SF_DEVICE void split8p(const float* p, bf16x8_t& hi, bf16x8_t& lo) {
  split8(*reinterpret_cast<const f32x4_t*>(p), *reinterpret_cast<const f32x4_t*>(p + 4), hi, lo);
}

#define LOG2E 1.4426950408889634f

// ------------------------------------------------------------------------------------------------
// U_h = Wk_h^T q_h as fp32 and as hi + lo bf16 planes, rows >= heads zero   (grid: 16 x ceil(D / 256))
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sf_pool_u_kernel(const float* __restrict__ wk, const float* __restrict__ q, float* __restrict__ u,
                                                        bf16_t* __restrict__ u_hi, bf16_t* __restrict__ u_lo, int heads, int D) {
  const int h = blockIdx.x, d = blockIdx.y * 256 + threadIdx.x;
  if (d >= D) return;
  float t = 0.f;
  if (h < heads) {
#pragma unroll 8
    for (int j = 0; j < 64; ++j) t = fmaf(wk[(size_t)(h * 64 + j) * D + d], q[h * 64 + j], t);
  }
  unsigned hi, lo;
  split_bf(t, hi, lo);
  u[(size_t)h * D + d] = t;
  u_hi[(size_t)h * D + d] = (bf16_t)hi;
  u_lo[(size_t)h * D + d] = (bf16_t)lo;
}
hipError_t sf_launch_pool_u(const float* wk, const float* q, float* u, bf16_t* u_hi, bf16_t* u_lo, int heads, int D, hipStream_t s) {
  if (heads > 16 || D != heads * 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sf_pool_u_kernel, dim3(16, (D + 255) / 256), dim3(256), 0, s, wk, q, u, u_hi, u_lo, heads, D);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// forward 1: scores -> softmax (within the workgroup's token split) -> weighted token sums
// grid (F * S, heads / HP); block = 64 * ceil(D / 256) threads (a thread owns 4 columns of the weighted sum)
// ------------------------------------------------------------------------------------------------
template <int HP>
__global__ __launch_bounds__(256) void sf_pool_probe_kernel(SfPoolArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sc[];      // [16][pitch] scores -> exp weights
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int f = blockIdx.x / p.S, sp = blockIdx.x % p.S, hg = blockIdx.y;
  const int per = (((p.N + p.S - 1) / p.S) + 15) & ~15;
  const int pitch = per + 4;
  const int n0 = sp * per;
  const int n1 = n0 + per < p.N ? n0 + per : p.N;
  const int nt = n1 > n0 ? n1 - n0 : 0;
  const int tiles = (nt + 15) >> 4;
  const float* xf = p.x + (size_t)f * p.N * p.D;
  const int D = p.D;

  // ---- scores S[head][token] = U[head] . x[token], three bf16 products per operand pair ----------------------------
  for (int t = wave; t < tiles; t += nw) {
    const int tok = n0 + t * 16 + l15;
    const int tokc = tok < p.N ? tok : p.N - 1;
    const float* xr = xf + (size_t)tokc * D + g * 8;
    const bf16_t* uh = p.u_hi + (size_t)l15 * D + g * 8;
    const bf16_t* ul = p.u_lo + (size_t)l15 * D + g * 8;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < D; k += 64) {          // D = heads * 64: two k-steps per trip, all eight loads in front of the arithmetic
      const f32x4_t x0 = *reinterpret_cast<const f32x4_t*>(xr + k), x1 = *reinterpret_cast<const f32x4_t*>(xr + k + 4);
      const f32x4_t x2 = *reinterpret_cast<const f32x4_t*>(xr + k + 32), x3 = *reinterpret_cast<const f32x4_t*>(xr + k + 36);
      const bf16x8_t ah0 = *reinterpret_cast<const bf16x8_t*>(uh + k), al0 = *reinterpret_cast<const bf16x8_t*>(ul + k);
      const bf16x8_t ah1 = *reinterpret_cast<const bf16x8_t*>(uh + k + 32), al1 = *reinterpret_cast<const bf16x8_t*>(ul + k + 32);
      bf16x8_t xh, xl;
      split8(x0, x1, xh, xl);
      acc = pmfma(al0, xh, acc);
      acc = pmfma(ah0, xl, acc);
      acc = pmfma(ah0, xh, acc);
      split8(x2, x3, xh, xl);
      acc = pmfma(al1, xh, acc);
      acc = pmfma(ah1, xl, acc);
      acc = pmfma(ah1, xh, acc);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sc[(4 * g + j) * pitch + t * 16 + l15] = tok < n1 ? acc[j] : -INFINITY;
  }
  __syncthreads();
  // ---- softmax over this split's tokens, heads of this workgroup ---------------------------------------------------
  const int h0 = hg * HP;
  for (int hh = wave; hh < HP; hh += nw) {
    const int h = h0 + hh;
    float* row = sc + h * pitch;
    float m = -INFINITY;
    for (int n = lane; n < nt; n += 64) m = fmaxf(m, row[n]);
    m = wave_max(m);
    float l = 0.f;
    for (int n = lane; n < tiles * 16; n += 64) {
      const float e = n < nt ? __builtin_amdgcn_exp2f((row[n] - m) * LOG2E) : 0.f;
      row[n] = e;
      l += e;
    }
    l = wave_sum(l);
    if (p.normalize) {
      const float inv = nt > 0 ? 1.0f / l : 0.f;
      for (int n = lane; n < tiles * 16; n += 64) {
        const float pr = row[n] * inv;
        row[n] = pr;
        if (p.probs && n < nt) p.probs[((size_t)f * p.heads + h) * p.N + n0 + n] = pr;
      }
    }
    if (lane == 0 && p.ml) {
      float* o = p.ml + (((size_t)f * p.S + sp) * p.heads + h) * 2;
      o[0] = nt > 0 ? m : -INFINITY;
      o[1] = nt > 0 ? l : 0.f;
    }
  }
  __syncthreads();
  // ---- z[h][d] = sum_n w[h][n] x[n][d]  (fp32; a thread owns 4 columns, HP heads) ----------------------------------
  for (int c4 = tid; c4 * 4 < D; c4 += blockDim.x) {
    f32x4_t acc[HP];
#pragma unroll
    for (int i = 0; i < HP; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const float* xc = xf + (size_t)n0 * D + c4 * 4;
    const int nt4 = tiles * 16;          // weights of the padding tokens are zero; their rows are clamped, finite reads
    for (int n = 0; n < nt4; n += 4) {
      f32x4_t xv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tk = n0 + n + i < p.N ? n + i : p.N - 1 - n0;
        xv[i] = *reinterpret_cast<const f32x4_t*>(xc + (size_t)tk * D);
      }
#pragma unroll
      for (int i = 0; i < HP; ++i) {
        const f32x4_t w = *reinterpret_cast<const f32x4_t*>(sc + (h0 + i) * pitch + n);
        acc[i] += w[0] * xv[0];
        acc[i] += w[1] * xv[1];
        acc[i] += w[2] * xv[2];
        acc[i] += w[3] * xv[3];
      }
    }
#pragma unroll
    for (int i = 0; i < HP; ++i)
      *reinterpret_cast<f32x4_t*>(p.zpart + (((size_t)f * p.S + sp) * p.heads + h0 + i) * D + c4 * 4) = acc[i];
  }
}

static int pool_hp(int heads) {
  if (heads <= 8) return heads;
  for (int hp = 6; hp >= 2; --hp)
    if (heads % hp == 0) return hp;
  return 1;
}
int sf_pool_splits(int F, int N, int heads) {
  const int hg = heads / pool_hp(heads);
  int S = 1;
  while (S < 8 && F * hg * S < 256 && (N + 2 * S - 1) / (2 * S) >= 16) S *= 2;
  return S;
}
size_t sf_pool_z_floats(int F, int N, int heads, int D) { return (size_t)F * sf_pool_splits(F, N, heads) * heads * D; }
size_t sf_pool_ml_floats(int F, int N, int heads) { return (size_t)F * sf_pool_splits(F, N, heads) * heads * 2; }

hipError_t sf_launch_pool_probe(const SfPoolArgs& a, hipStream_t s) {
  if (a.heads > 16 || a.D != a.heads * 64 || a.F <= 0 || a.N <= 0 || a.S < 1) return hipErrorInvalidValue;
  if (a.normalize && a.S != 1) return hipErrorInvalidValue;
  const int per = (((a.N + a.S - 1) / a.S) + 15) & ~15;
  const size_t lds = (size_t)16 * (per + 4) * sizeof(float);
  if (lds > 150 * 1024) return hipErrorInvalidValue;
  const int hp = pool_hp(a.heads);
  const int threads = 64 * ((a.D / 4 + 63) / 64) > 256 ? 256 : 64 * ((a.D / 4 + 63) / 64);
  const dim3 grid(a.F * a.S, a.heads / hp);
#define SF_POOL_CASE(HP)                                                                                              \
  case HP: {                                                                                                          \
    static SfPerDeviceOnce once;                                                                                      \
    if (once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_pool_probe_kernel<HP>),             \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);              \
    hipLaunchKernelGGL(sf_pool_probe_kernel<HP>, grid, dim3(threads), lds, s, a);                                     \
    break;                                                                                                            \
  }
  switch (hp) {
    SF_POOL_CASE(1) SF_POOL_CASE(2) SF_POOL_CASE(3) SF_POOL_CASE(4) SF_POOL_CASE(5) SF_POOL_CASE(6) SF_POOL_CASE(7) SF_POOL_CASE(8)
    default: return hipErrorInvalidValue;
  }
#undef SF_POOL_CASE
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// forward 2: combine the token splits, ctx[f][h*64 + c] = Wv[h*64 + c] . z[f][h] + bv   (MFMA, bf16x3)
// grid (ceil(F / 16), heads); 4 waves = the four 16-column tiles of the head
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sf_pool_ctx_kernel(SfPoolCtxArgs p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int f = blockIdx.x * 16 + l15, h = blockIdx.y;
  const bool fv = f < p.F;
  const int fc = fv ? f : p.F - 1;
  const int D = p.D, S = p.S;
  float wgt[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) wgt[i] = 0.f;
  if (S > 1) {
    float m = -INFINITY;
    for (int sidx = 0; sidx < S; ++sidx) m = fmaxf(m, p.ml[(((size_t)fc * S + sidx) * p.heads + h) * 2]);
    float L = 0.f;
#pragma unroll
    for (int sidx = 0; sidx < 8; ++sidx)
      if (sidx < S) {
        const float* o = p.ml + (((size_t)fc * S + sidx) * p.heads + h) * 2;
        const float w = o[1] > 0.f ? __builtin_amdgcn_exp2f((o[0] - m) * LOG2E) : 0.f;
        wgt[sidx] = w;
        L += w * o[1];
      }
    const float inv = 1.0f / L;
#pragma unroll
    for (int sidx = 0; sidx < 8; ++sidx) wgt[sidx] *= inv;
  }
  const float* wr = p.wv + (size_t)(h * 64 + wave * 16 + l15) * p.ldw + g * 8;
  const float* zr = p.zpart + (((size_t)fc * S) * p.heads + h) * D + g * 8;
  const size_t zs = (size_t)p.heads * D;         // split stride
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < D; k += 32) {
    f32x4_t za = {0.f, 0.f, 0.f, 0.f}, zb = {0.f, 0.f, 0.f, 0.f};
    if (fv) {
      if (S == 1) {
        za = *reinterpret_cast<const f32x4_t*>(zr + k);
        zb = *reinterpret_cast<const f32x4_t*>(zr + k + 4);
      } else {
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx)
          if (sidx < S) {
            za += wgt[sidx] * *reinterpret_cast<const f32x4_t*>(zr + sidx * zs + k);
            zb += wgt[sidx] * *reinterpret_cast<const f32x4_t*>(zr + sidx * zs + k + 4);
          }
      }
    }
    bf16x8_t zh, zl, wh, wl;
    split8(za, zb, zh, zl);
    split8p(wr + k, wh, wl);
    acc = pmfma(wl, zh, acc);
    acc = pmfma(wh, zl, acc);
    acc = pmfma(wh, zh, acc);
    if (p.z_out && S > 1 && wave == 0 && fv) {       // the combined, normalised sums (kept for inspection / a later backward)
      *reinterpret_cast<f32x4_t*>(p.z_out + ((size_t)f * p.heads + h) * D + g * 8 + k) = za;
      *reinterpret_cast<f32x4_t*>(p.z_out + ((size_t)f * p.heads + h) * D + g * 8 + k + 4) = zb;
    }
  }
  // lane: ctx[frame l15][column h*64 + wave*16 + 4g + j]
  if (!fv) return;
  const int c = h * 64 + wave * 16 + 4 * g;
  const f32x4_t b = *reinterpret_cast<const f32x4_t*>(p.bv + c);
  acc += b;
  unsigned hi[4], lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split_bf(acc[j], hi[j], lo[j]);
  const size_t o = (size_t)f * D + c;
  if (p.ctx_f32) *reinterpret_cast<f32x4_t*>(p.ctx_f32 + o) = acc;
  if (p.ctx_hi) *reinterpret_cast<u32x2_t*>(p.ctx_hi + o) = (u32x2_t){hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16)};
  if (p.ctx_lo) *reinterpret_cast<u32x2_t*>(p.ctx_lo + o) = (u32x2_t){lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16)};
}
hipError_t sf_launch_pool_ctx(const SfPoolCtxArgs& a, hipStream_t s) {
  if (a.heads > 16 || a.D != a.heads * 64 || a.F <= 0 || a.S < 1 || a.S > 8 || (a.S > 1 && !a.ml)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sf_pool_ctx_kernel, dim3((a.F + 15) / 16, a.heads), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ================================================================================================
// backward
// ================================================================================================
// dz[f][h][d] = sum_j dctx[f][h*64 + j] Wv[h*64 + j][d]: A = Wv^T rows (the trainer's transposed bf16 working copy
// wT [D][ldt], value columns at col0), B = bf16(dctx) rows.  grid (ceil(F / 16), heads); waves walk the D / 16 column tiles
__global__ __launch_bounds__(256) void sf_pool_dz_kernel(const float* __restrict__ dctx, const bf16_t* __restrict__ wT, int ldt, int col0,
                                                         float* __restrict__ dz, int F, int heads, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int f = blockIdx.x * 16 + l15, h = blockIdx.y;
  const bool fv = f < F;
  const float* dr = dctx + (size_t)(fv ? f : F - 1) * D + h * 64 + g * 8;
  const bf16x8_t b0 = cvt8(*reinterpret_cast<const f32x4_t*>(dr), *reinterpret_cast<const f32x4_t*>(dr + 4));
  const bf16x8_t b1 = cvt8(*reinterpret_cast<const f32x4_t*>(dr + 32), *reinterpret_cast<const f32x4_t*>(dr + 36));
  for (int dt = wave; dt * 16 < D; dt += 4) {
    const bf16_t* ar = wT + (size_t)(dt * 16 + l15) * ldt + col0 + h * 64 + g * 8;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    acc = pmfma(*reinterpret_cast<const bf16x8_t*>(ar), b0, acc);
    acc = pmfma(*reinterpret_cast<const bf16x8_t*>(ar + 32), b1, acc);
    if (fv) *reinterpret_cast<f32x4_t*>(dz + ((size_t)f * heads + h) * D + dt * 16 + 4 * g) = acc;
  }
}
// dWv[h*64 + j][d] += sum_f dctx[f][h*64 + j] z[f][h][d];  dbv[h*64 + j] += sum_f dctx[f][h*64 + j]
// grid (heads, 8): eight value rows per workgroup; a thread owns 4 columns; frames in a fixed order (deterministic)
__global__ __launch_bounds__(256) void sf_pool_dwv_kernel(const float* __restrict__ dctx, const float* __restrict__ z, float* __restrict__ dwv,
                                                          int ldw, float* __restrict__ dbv, int F, int heads, int D) {
  __shared__ float dl[128][8];
  const int h = blockIdx.x, j0 = blockIdx.y * 8, tid = threadIdx.x;
  const int c4 = tid;
  const bool cv = c4 * 4 < D;
  f32x4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  for (int f0 = 0; f0 < F; f0 += 128) {
    const int nf = F - f0 < 128 ? F - f0 : 128;
    for (int i = tid; i < nf * 8; i += blockDim.x) dl[i >> 3][i & 7] = dctx[(size_t)(f0 + (i >> 3)) * D + h * 64 + j0 + (i & 7)];
    __syncthreads();
    if (cv) {
      const float* zc = z + ((size_t)f0 * heads + h) * D + c4 * 4;
#pragma unroll 4
      for (int ff = 0; ff < nf; ++ff) {
        const f32x4_t zv = *reinterpret_cast<const f32x4_t*>(zc + (size_t)ff * heads * D);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += dl[ff][i] * zv;
      }
    }
    if (tid < 8)
      for (int ff = 0; ff < nf; ++ff) bsum += dl[ff][tid];
    __syncthreads();
  }
  if (cv && dwv) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x4_t* o = reinterpret_cast<f32x4_t*>(dwv + (size_t)(h * 64 + j0 + i) * ldw + c4 * 4);
      *o = *o + acc[i];
    }
  }
  if (tid < 8 && dbv) dbv[h * 64 + j0 + tid] += bsum;
}
hipError_t sf_launch_pool_ctx_bwd(const float* dctx, const bf16_t* wT, int ldt, int col0, const float* z, float* dz, float* dwv, int ldw,
                                  float* dbv, int F, int heads, int D, hipStream_t s) {
  if (heads > 16 || D != heads * 64 || D > 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sf_pool_dz_kernel, dim3((F + 15) / 16, heads), dim3(256), 0, s, dctx, wT, ldt, col0, dz, F, heads, D);
  if (dwv || dbv) hipLaunchKernelGGL(sf_pool_dwv_kernel, dim3(heads, 8), dim3(256), 0, s, dctx, z, dwv, ldw, dbv, F, heads, D);
  return hipGetLastError();
}

// probe attention backward: grid (F, S2), 256 threads.  LDS: T = [dz ; U]^T as hi + lo bf16 [D][40] (k = 0..15 dz heads, 16..31 U heads),
// PD = [p | ds] of this workgroup's tokens [tokens][36] fp32, delta[16]
#define TP 40
#define PDP 36
__global__ __launch_bounds__(256) void sf_pool_probe_bwd_kernel(SfPoolBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int f = blockIdx.x, sp = blockIdx.y;
  const int D = p.D, N = p.N, heads = p.heads;
  const int per = (((N + gridDim.y - 1) / gridDim.y) + 15) & ~15;
  const int n0 = sp * per;
  const int n1 = n0 + per < N ? n0 + per : N;
  const int nt = n1 > n0 ? n1 - n0 : 0;
  const int tiles = (nt + 15) >> 4;
  bf16_t* T_hi = reinterpret_cast<bf16_t*>(smem);
  bf16_t* T_lo = T_hi + (size_t)D * TP;
  float* PD = reinterpret_cast<float*>(T_lo + (size_t)D * TP);
  float* delta = PD + (size_t)per * PDP;
  const float* dzf = p.dz + (size_t)f * heads * D;
  const float* zf = p.z + (size_t)f * heads * D;
  for (int i = tid; i < 32 * D; i += 256) {
    const int k = i / D, d = i - k * D;
    float v = 0.f;
    if (k < 16) { if (k < heads) v = dzf[(size_t)k * D + d]; }
    else if (k - 16 < heads) v = p.u[(size_t)(k - 16) * D + d];
    unsigned hi, lo;
    split_bf(v, hi, lo);
    T_hi[d * TP + k] = (bf16_t)hi;
    T_lo[d * TP + k] = (bf16_t)lo;
  }
  for (int h = wave; h < 16; h += 4) {
    float t = 0.f;
    if (h < heads)
      for (int d = lane; d < D; d += 64) t = fmaf(dzf[(size_t)h * D + d], zf[(size_t)h * D + d], t);
    t = wave_sum(t);
    if (lane == 0) delta[h] = t;
  }
  __syncthreads();
  for (int t = wave; t < tiles; t += 4) {
    // ---- dp[head][token] = dz[head] . x[token]  (x: the saved bf16 normalised tokens) -----------------------------
    const int tl = t * 16 + l15;                 // token inside this workgroup's range
    const int tok = n0 + tl;
    const bool tv = tok < n1;
    const size_t row = (size_t)f * N + (tv ? tok : N - 1);
    const bf16_t* xr = p.x_bf + row * D + g * 8;
    const float* ar = dzf + (size_t)(l15 < heads ? l15 : 0) * D + g * 8;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < D; k += 64) {
      const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(ar + k), a1 = *reinterpret_cast<const f32x4_t*>(ar + k + 4);
      const f32x4_t a2 = *reinterpret_cast<const f32x4_t*>(ar + k + 32), a3 = *reinterpret_cast<const f32x4_t*>(ar + k + 36);
      const bf16x8_t xb0 = *reinterpret_cast<const bf16x8_t*>(xr + k), xb1 = *reinterpret_cast<const bf16x8_t*>(xr + k + 32);
      bf16x8_t ah, al;
      split8(a0, a1, ah, al);
      acc = pmfma(al, xb0, acc);
      acc = pmfma(ah, xb0, acc);
      split8(a2, a3, ah, al);
      acc = pmfma(al, xb1, acc);
      acc = pmfma(ah, xb1, acc);
    }
    // lane: dp[head 4g + j][token l15]
    float pr[4], ds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int h = 4 * g + j;
      const bool ok = tv && h < heads;
      pr[j] = ok ? p.probs[((size_t)f * heads + h) * N + tok] : 0.f;
      ds[j] = ok ? pr[j] * (acc[j] - delta[h]) : 0.f;
    }
    *reinterpret_cast<f32x4_t*>(PD + (size_t)tl * PDP + 4 * g) = (f32x4_t){pr[0], pr[1], pr[2], pr[3]};
    *reinterpret_cast<f32x4_t*>(PD + (size_t)tl * PDP + 16 + 4 * g) = (f32x4_t){ds[0], ds[1], ds[2], ds[3]};
    if (tv) {      // ds as the dY operand of the dU weight-gradient GEMM: [M][32] bf16, columns >= 16 zero
      bf16_t* o = p.ds_bf + ((size_t)f * N + tok) * 32;
      *reinterpret_cast<u32x2_t*>(o + 4 * g) = (u32x2_t){pack_bf2(ds[0], ds[1]), pack_bf2(ds[2], ds[3])};
      *reinterpret_cast<u32x2_t*>(o + 16 + 4 * g) = (u32x2_t){0u, 0u};
    }
    // ---- dx[token][d] = sum_k [p | ds][token][k] T[d][k]: one 32-deep k-step per 16 x 16 tile ------------------------
    // (this wave wrote the PD rows it reads: LDS operations of a wave complete in order)
    const float* pdr = PD + (size_t)tl * PDP + 8 * g;
    bf16x8_t bh, bl;
    split8(*reinterpret_cast<const f32x4_t*>(pdr), *reinterpret_cast<const f32x4_t*>(pdr + 4), bh, bl);
    float* orow = p.dx + ((size_t)f * N + (tv ? tok : N - 1)) * D + 4 * g;
    const float* lrow = p.d_lhs ? p.d_lhs + ((size_t)f * N + (tv ? tok : N - 1)) * D + 4 * g : nullptr;
    for (int dt = 0; dt * 16 < D; ++dt) {
      const bf16x8_t th = *reinterpret_cast<const bf16x8_t*>(T_hi + (size_t)(dt * 16 + l15) * TP + 8 * g);
      const bf16x8_t tlv = *reinterpret_cast<const bf16x8_t*>(T_lo + (size_t)(dt * 16 + l15) * TP + 8 * g);
      f32x4_t a2 = {0.f, 0.f, 0.f, 0.f};
      a2 = pmfma(tlv, bh, a2);
      a2 = pmfma(th, bl, a2);
      a2 = pmfma(th, bh, a2);
      if (tv) {
        if (lrow) a2 += *reinterpret_cast<const f32x4_t*>(lrow + dt * 16);
        *reinterpret_cast<f32x4_t*>(orow + dt * 16) = a2;
      }
    }
  }
}
hipError_t sf_launch_pool_probe_bwd(const SfPoolBwdArgs& a, hipStream_t s) {
  if (a.heads > 16 || a.D != a.heads * 64 || a.D > 1024 || a.F <= 0 || a.N <= 0) return hipErrorInvalidValue;
  int S2 = 1;
  while (S2 < 4 && a.F * S2 < 256 && (a.N + 2 * S2 - 1) / (2 * S2) >= 16) S2 *= 2;
  const int per = (((a.N + S2 - 1) / S2) + 15) & ~15;
  const size_t lds = (size_t)2 * a.D * TP * sizeof(bf16_t) + (size_t)per * PDP * sizeof(float) + 16 * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static SfPerDeviceOnce once;
  if (once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_pool_probe_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(sf_pool_probe_bwd_kernel, dim3(a.F, S2), dim3(256), lds, s, a);
  return hipGetLastError();
}

// dWk[h*64 + j][d] += q[h*64 + j] dU[h][d];  dq[h*64 + j] = sum_d Wk[h*64 + j][d] dU[h][d]     (grid D rows, 256 threads)
__global__ __launch_bounds__(256) void sf_pool_u_bwd_kernel(const float* __restrict__ du, const float* __restrict__ wk, const float* __restrict__ q,
                                                            float* __restrict__ dwk, float* __restrict__ dq, int D) {
  __shared__ float red[4];
  const int c = blockIdx.x, h = c >> 6, tid = threadIdx.x;
  const float qc = q[c];
  float t = 0.f;
  for (int d = tid; d < D; d += 256) {
    const float g = du[(size_t)h * D + d];
    t = fmaf(wk[(size_t)c * D + d], g, t);
    if (dwk) dwk[(size_t)c * D + d] += qc * g;
  }
  t = wave_sum(t);
  if ((tid & 63) == 0) red[tid >> 6] = t;
  __syncthreads();
  if (tid == 0) dq[c] = (red[0] + red[1]) + (red[2] + red[3]);
}
hipError_t sf_launch_pool_u_bwd(const float* du, const float* wk, const float* q, float* dwk, float* dq, int D, hipStream_t s) {
  hipLaunchKernelGGL(sf_pool_u_bwd_kernel, dim3(D), dim3(256), 0, s, du, wk, q, dwk, dq, D);
  return hipGetLastError();
}
