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
#include "sf_switches.h"

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
// forward 1: one pass over the tokens of a (frame, token split) — chunks of 16 tokens staged in LDS (the next chunk's global
// loads in flight under the current chunk's arithmetic), scores by MFMA with the K range split over the four waves, online
// softmax (running max / sum per head, rescaled accumulators), weighted token sums in fp32 from the LDS image.
// grid F * S; 256 threads; a thread owns 4 columns x all heads of the weighted sums
// ------------------------------------------------------------------------------------------------
#define PCH 16                         // tokens per chunk
// all-reduce over the 16 lanes of a DPP row by rotations (row_ror:8 / 4 / 2 / 1): every lane ends with the row's result
template <int ROT>
SF_DEVICE float row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + ROT, 0xf, 0xf, false));
}
SF_DEVICE float row16_max(float v) {
  v = fmaxf(v, row_ror<8>(v)); v = fmaxf(v, row_ror<4>(v)); v = fmaxf(v, row_ror<2>(v)); return fmaxf(v, row_ror<1>(v));
}
SF_DEVICE float row16_sum(float v) {
  v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); return v + row_ror<1>(v);
}
// NH: heads rounded up to 4 / 8 / 12 / 16 (U rows past `heads` are zero); KI: k-steps per wave, ceil(D / 128) — a compile-time
// count keeps the score loop one straight-line region (run-time trip counts split it into load -> wait -> use blocks)
template <int NH, int KI>
__global__ __launch_bounds__(256) void sf_pool_probe_kernel(SfPoolArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int D = p.D, XP = D + 4, UP = D + 8;        // LDS row pitches (fp32 tokens, bf16 U rows): 4 banks of skew per row
  float* xs = smem_f;                               // [PCH][XP] the current chunk's tokens (MFMA operand image)
  float* red = xs + PCH * XP;                       // [4 waves][64 lanes][4] K-split partial scores
  float* wts = red + 4 * 64 * 4;                    // [16 heads][PCH] exp weights of the current chunk
  float* st = wts + 16 * PCH;                       // [2][48]: running max [16], running sum [16], this chunk's rescale factor [16]
  bf16_t* us_hi = reinterpret_cast<bf16_t*>(st + 96);   // [16][UP]
  bf16_t* us_lo = us_hi + 16 * UP;
  const int f = blockIdx.x / p.S, sp = blockIdx.x % p.S;
  const int per = (((p.N + p.S - 1) / p.S) + 15) & ~15;
  const int n0 = sp * per;
  const int n1 = n0 + per < p.N ? n0 + per : p.N;
  const int nt = n1 > n0 ? n1 - n0 : 0;
  const int nch = (nt + PCH - 1) / PCH;
  const bool own = tid * 4 < D;                     // this thread owns columns 4 tid .. 4 tid + 3 of every token row
  const float* xcol = (p.x_ind ? *p.x_ind : p.x) + ((size_t)f * p.N + n0) * D + (own ? tid * 4 : 0);

  if (tid < 96) st[tid] = (tid % 48) < 16 ? -INFINITY : 0.f;
  {   // U planes -> LDS, 16 bytes per access; all loads first (a fixed trip count; an index past the end repeats the last element)
    const int c8n = D >> 3, tot = 16 * c8n;
    u32x4_t th[KI], tl[KI];
    int dst[KI];
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int e = i * 256 + tid < tot ? i * 256 + tid : tot - 1;
      const int r = e / c8n, c8 = e - r * c8n;
      th[i] = *reinterpret_cast<const u32x4_t*>(p.u_hi + (size_t)r * D + c8 * 8);
      tl[i] = *reinterpret_cast<const u32x4_t*>(p.u_lo + (size_t)r * D + c8 * 8);
      dst[i] = r * UP + c8 * 8;
    }
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      *reinterpret_cast<u32x4_t*>(us_hi + dst[i]) = th[i];
      *reinterpret_cast<u32x4_t*>(us_lo + dst[i]) = tl[i];
    }
  }
  f32x4_t acc[NH];
#pragma unroll
  for (int i = 0; i < NH; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // A thread reads its own 4 columns of the chunk's 16 tokens: exactly the operands of ITS weighted sums, which therefore stay in
  // registers; the LDS image only feeds the score MFMAs.  The next chunk's 16 loads fly under the current chunk's arithmetic.
  f32x4_t cx[PCH], nx[PCH];
  auto issue = [&](int c) {
#pragma unroll
    for (int t = 0; t < PCH; ++t) {
      const int tok = c * PCH + t < nt ? c * PCH + t : nt - 1;          // rows past the split: a valid row, weight zero
      nx[t] = *reinterpret_cast<const f32x4_t*>(xcol + (size_t)tok * D);
    }
  };
  if (nch > 0 && own) issue(0);
  const int nks = D >> 5;
  for (int c = 0; c < nch; ++c) {
    const float* sto = st + (c & 1) * 48;
    float* stn = st + ((c + 1) & 1) * 48;
    if (own) {
#pragma unroll
      for (int t = 0; t < PCH; ++t) {
        cx[t] = nx[t];
        *reinterpret_cast<f32x4_t*>(xs + t * XP + tid * 4) = cx[t];
      }
      if (c + 1 < nch) issue(c + 1);
    }
    __syncthreads();
    // ---- scores of the chunk: S[head][token] = U[head] . x[token]; wave w takes k-steps w, w + 4, ... ------------------
    {
      f32x4_t a = {0.f, 0.f, 0.f, 0.f};
      const float* xr = xs + l15 * XP + g * 8;
      const bf16_t* uh = us_hi + l15 * UP + g * 8;
      const bf16_t* ul = us_lo + l15 * UP + g * 8;
      f32x4_t xa[KI], xb4[KI];
      bf16x8_t ah[KI], al[KI];
#pragma unroll
      for (int i = 0; i < KI; ++i) {                 // a k-step past the end (ragged D): a valid address, operand zeroed below
        const int ks = wave + 4 * i;
        const int kc = ks < nks ? ks : nks - 1;
        xa[i] = *reinterpret_cast<const f32x4_t*>(xr + kc * 32);
        xb4[i] = *reinterpret_cast<const f32x4_t*>(xr + kc * 32 + 4);
        ah[i] = *reinterpret_cast<const bf16x8_t*>(uh + kc * 32);
        al[i] = *reinterpret_cast<const bf16x8_t*>(ul + kc * 32);
      }
#pragma unroll
      for (int i = 0; i < KI; ++i) {
        const float keep = wave + 4 * i < nks ? 1.f : 0.f;
        bf16x8_t xh, xl;
        split8(xa[i] * keep, xb4[i] * keep, xh, xl);
        a = pmfma(al[i], xh, a);
        a = pmfma(ah[i], xl, a);
        a = pmfma(ah[i], xh, a);
      }
      *reinterpret_cast<f32x4_t*>(red + (wave * 64 + lane) * 4) = a;
    }
    __syncthreads();
    {   // thread (h = tid / 16, tok = tid % 16): the score, then the online-softmax update of head h over the 16 lanes of its DPP row
      const int h = tid >> 4, tk = tid & 15;
      const int src = ((h >> 2) * 16 + tk) * 4 + (h & 3);
      const float sraw = (red[src] + red[256 + src]) + (red[512 + src] + red[768 + src]);
      const bool ok = c * PCH + tk < nt;
      if (p.probs && ok && h < p.heads) p.probs[((size_t)f * p.heads + h) * p.N + n0 + c * PCH + tk] = sraw;      // raw scores, finished below
      const float sv = ok ? sraw : -INFINITY;
      const float cm = row16_max(sv);
      const float m_old = sto[h], l_old = sto[16 + h];
      const float m_new = fmaxf(m_old, cm);
      const float e = ok ? __builtin_amdgcn_exp2f((sv - m_new) * LOG2E) : 0.f;
      const float cl = row16_sum(e);
      const float alpha = m_old == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((m_old - m_new) * LOG2E);
      wts[h * PCH + tk] = e;
      if (tk == 0) { stn[h] = m_new; stn[16 + h] = l_old * alpha + cl; stn[32 + h] = alpha; }
    }
    __syncthreads();
    // ---- z[h][d] = alpha_h z[h][d] + sum_tok w[h][tok] x[tok][d], x from this thread's registers ----------------------
    if (own) {
#pragma unroll
      for (int i = 0; i < NH; ++i) {
        acc[i] *= stn[32 + i];
#pragma unroll
        for (int t4 = 0; t4 < PCH; t4 += 4) {
          const f32x4_t w = *reinterpret_cast<const f32x4_t*>(wts + i * PCH + t4);
          acc[i] += w[0] * cx[t4];
          acc[i] += w[1] * cx[t4 + 1];
          acc[i] += w[2] * cx[t4 + 2];
          acc[i] += w[3] * cx[t4 + 3];
        }
      }
    }
    // no barrier here: the next trip's first LDS writes go to xs (last read before the barrier above), and red / wts / st
    // are only rewritten behind the next trip's barriers
  }
  __syncthreads();
  const float* stf = st + (nch & 1) * 48;
  // ---- results: the weighted sums (normalised when this workgroup saw the whole frame), {max, sum} per head, probabilities ----
  if (own) {
#pragma unroll
    for (int i = 0; i < NH; ++i)
      if (i < p.heads) {
        const float l = stf[16 + i];
        const float sc = p.normalize ? (l > 0.f ? 1.0f / l : 0.f) : 1.0f;
        *reinterpret_cast<f32x4_t*>(p.zpart + (((size_t)f * p.S + sp) * p.heads + i) * D + tid * 4) = acc[i] * sc;
      }
  }
  if (p.ml && tid < p.heads) {
    float* o = p.ml + (((size_t)f * p.S + sp) * p.heads + tid) * 2;
    o[0] = stf[tid];
    o[1] = stf[16 + tid];
  }
  if (p.probs && !p.probs_raw) {      // raw scores (written by this workgroup above; S == 1) -> probabilities
    __threadfence_block();
    __syncthreads();
    for (int i = tid; i < p.heads * nt; i += 256) {
      const int h = i / nt, n = i - h * nt;
      float* o = p.probs + ((size_t)f * p.heads + h) * p.N + n0 + n;
      *o = __builtin_amdgcn_exp2f((*o - stf[h]) * LOG2E) / stf[16 + h];
    }
  }
}

// Both forward kernels request a CU's WHOLE LDS (160 KB), not the 17 - 102 KB they use: no other workgroup then shares their CU.
// Measured reason ("Device sharing"): with a second process on the device, workgroups of its sf_temporal_attn_bwd_kernel
// that landed on the same CU made these kernels' results differ in single registers of 16 lanes (about one forward in ten); with the
// whole-CU request 0 of 900 forwards differed.  One process per GPU never co-schedules another kernel with them (same stream), and
// the request costs the combine kernel a second round of workgroups (384 on 256 CUs) — SF_POOL_SHARE_CU=1 gives the exact sizes back.
static size_t pool_lds(size_t used) { return sf_sw(SW_POOL_SHARE_CU) ? used : (size_t)160 * 1024; }

int sf_pool_splits(int F, int N, int heads) {
  (void)heads;
  int S = 1;
  while (S < 8 && F * S < 256 && (N + 2 * S - 1) / (2 * S) >= 16) S *= 2;
  return S;
}
size_t sf_pool_z_floats(int F, int N, int heads, int D) { return (size_t)F * sf_pool_splits(F, N, heads) * heads * D; }
size_t sf_pool_ml_floats(int F, int N, int heads) { return (size_t)F * sf_pool_splits(F, N, heads) * heads * 2; }

hipError_t sf_launch_pool_probe(const SfPoolArgs& a, hipStream_t s) {
  if (a.heads > 16 || a.D != a.heads * 64 || a.D > 1024 || a.F <= 0 || a.N <= 0 || a.S < 1) return hipErrorInvalidValue;
  if ((a.normalize || (a.probs && !a.probs_raw)) && a.S != 1) return hipErrorInvalidValue;
  if (a.S > 1 && !a.ml) return hipErrorInvalidValue;
  const size_t lds = pool_lds(((size_t)PCH * (a.D + 4) + 4 * 64 * 4 + 16 * PCH + 96) * sizeof(float) + (size_t)2 * 16 * (a.D + 8) * sizeof(bf16_t));
  const dim3 grid(a.F * a.S);
#define SF_POOL_CASE(NH, KI)                                                                                          \
  {                                                                                                                   \
    static SfPerDeviceOnce once;                                                                                      \
    if (once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_pool_probe_kernel<NH, KI>),         \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);              \
    hipLaunchKernelGGL((sf_pool_probe_kernel<NH, KI>), grid, dim3(256), lds, s, a);                                   \
  }
  // D = 64 heads: k-steps per wave = ceil(2 heads / 4)
  if (a.heads <= 2) SF_POOL_CASE(4, 1)
  else if (a.heads <= 4) SF_POOL_CASE(4, 2)
  else if (a.heads <= 6) SF_POOL_CASE(8, 3)
  else if (a.heads <= 8) SF_POOL_CASE(8, 4)
  else if (a.heads <= 12) SF_POOL_CASE(12, 6)
  else SF_POOL_CASE(16, 8)
#undef SF_POOL_CASE
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// forward 2: combine the token splits, ctx[f][h*64 + c] = Wv[h*64 + c] . z[f][h] + bv   (MFMA, bf16x3)
// grid (ceil(F / 16), heads * 4): one 16-frame x 16-column tile per workgroup, its K range split over the four waves
// (every load of a wave issued in front of its arithmetic: the tile is one memory round trip deep)
// ------------------------------------------------------------------------------------------------
template <int CTX_KI>                  // k-steps per wave, ceil(D / 128): compile-time, so that every load of a wave is issued before its first use
__global__ __launch_bounds__(256) void sf_pool_ctx_kernel(SfPoolCtxArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int f0 = blockIdx.x * 16, h = blockIdx.y >> 2, ct = blockIdx.y & 3;
  const int D = p.D, S = p.S, ZP = D + 4;
  float* red = smem_f;                               // [4][64][4]
  float* zs = red + 4 * 64 * 4;                      // S > 1: combined z tile [16][ZP]
  const int nf = p.F - f0 < 16 ? p.F - f0 : 16;
  const bool fv = l15 < nf;
  const int nks = D >> 5;
  // this wave's weight fragments (k-steps wave, wave + 4, ...): issued first, they fly under the combine below
  const float* wr = p.wv + (size_t)(h * 64 + ct * 16 + l15) * p.ldw + g * 8;
  f32x4_t wa[CTX_KI], wb[CTX_KI];
#pragma unroll
  for (int i = 0; i < CTX_KI; ++i) {
    const int ks = wave + 4 * i;
    const int kc = ks < nks ? ks : nks - 1;          // ragged D: a valid address, the z fragment is zeroed below
    wa[i] = *reinterpret_cast<const f32x4_t*>(wr + kc * 32);
    wb[i] = *reinterpret_cast<const f32x4_t*>(wr + kc * 32 + 4);
  }
  if (S > 1) {
    // combine the splits of the valid frames, flash-decoding style: z = sum_s exp(m_s - m) z_s / sum_s exp(m_s - m) l_s
    float* wl = zs + 16 * ZP;                        // [16 frames][8 splits] weights
    const int nv4 = D >> 2;
    // the partial sums of this thread's elements first (independent of the weights), then the weights, then the combination
    f32x4_t part[8];
    const int e0 = tid;                              // one valid frame of <= 1024 columns (the streamed case); more frames loop below
    const bool one = nf == 1 && nv4 <= 256;
    if (one && e0 < nf * nv4) {
      const float* zr0 = p.zpart + (((size_t)f0 * S) * p.heads + h) * D + e0 * 4;
#pragma unroll
      for (int sidx = 0; sidx < 8; ++sidx)          // branch-free: a split past S reads split 0, its weight below is zero
        part[sidx] = *reinterpret_cast<const f32x4_t*>(zr0 + (size_t)(sidx < S ? sidx : 0) * p.heads * D);
    }
    if (tid < 16) {
      float w[8], mm[8], ll[8];
      const int fr = tid < nf ? f0 + tid : f0;
#pragma unroll
      for (int sidx = 0; sidx < 8; ++sidx) {
        const float* o = p.ml + (((size_t)fr * S + (sidx < S ? sidx : 0)) * p.heads + h) * 2;
        const float a0 = o[0], a1 = o[1];
        mm[sidx] = sidx < S ? a0 : -INFINITY;
        ll[sidx] = sidx < S ? a1 : 0.f;
      }
      float m = -INFINITY, L = 0.f;
#pragma unroll
      for (int sidx = 0; sidx < 8; ++sidx) m = fmaxf(m, mm[sidx]);
#pragma unroll
      for (int sidx = 0; sidx < 8; ++sidx) {
        w[sidx] = ll[sidx] > 0.f ? __builtin_amdgcn_exp2f((mm[sidx] - m) * LOG2E) : 0.f;
        L += w[sidx] * ll[sidx];
      }
      const float inv = L > 0.f ? 1.0f / L : 0.f;
#pragma unroll
      for (int sidx = 0; sidx < 8; ++sidx) wl[tid * 8 + sidx] = w[sidx] * inv;
    }
    __syncthreads();
    if (one) {
      if (e0 < nf * nv4) {
        f32x4_t v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) v += (sidx < S ? wl[sidx] : 0.f) * part[sidx];
        *reinterpret_cast<f32x4_t*>(zs + e0 * 4) = v;
        if (p.z_out && ct == 0) *reinterpret_cast<f32x4_t*>(p.z_out + ((size_t)f0 * p.heads + h) * D + e0 * 4) = v;
      }
    } else {
      const int tot = nf * nv4;
      for (int e0b = 0; e0b < tot; e0b += 3 * 256) {       // three elements per thread and trip: 24 independent loads in flight
        f32x4_t pv[3][8];
        int fr3[3], c43[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int e = e0b + u * 256 + tid < tot ? e0b + u * 256 + tid : tot - 1;        // past the end: the last element again
          fr3[u] = e / nv4; c43[u] = e - fr3[u] * nv4;
          const float* zr = p.zpart + (((size_t)(f0 + fr3[u]) * S) * p.heads + h) * D + c43[u] * 4;
#pragma unroll
          for (int sidx = 0; sidx < 8; ++sidx) pv[u][sidx] = *reinterpret_cast<const f32x4_t*>(zr + (size_t)(sidx < S ? sidx : 0) * p.heads * D);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          f32x4_t v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int sidx = 0; sidx < 8; ++sidx) v += (sidx < S ? wl[fr3[u] * 8 + sidx] : 0.f) * pv[u][sidx];
          *reinterpret_cast<f32x4_t*>(zs + fr3[u] * ZP + c43[u] * 4) = v;       // a repeated last element rewrites the same value
          if (p.z_out && ct == 0) *reinterpret_cast<f32x4_t*>(p.z_out + ((size_t)(f0 + fr3[u]) * p.heads + h) * D + c43[u] * 4) = v;
        }
      }
    }
    __syncthreads();
  }
  const float* zr = S > 1 ? zs + (fv ? l15 : 0) * ZP + g * 8 : p.zpart + ((size_t)(f0 + (fv ? l15 : 0)) * p.heads + h) * D + g * 8;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  {
    f32x4_t za[CTX_KI], zb[CTX_KI];
#pragma unroll
    for (int i = 0; i < CTX_KI; ++i) {
      const int ks = wave + 4 * i;
      const int kc = ks < nks ? ks : nks - 1;
      za[i] = *reinterpret_cast<const f32x4_t*>(zr + kc * 32);
      zb[i] = *reinterpret_cast<const f32x4_t*>(zr + kc * 32 + 4);
    }
#pragma unroll
    for (int i = 0; i < CTX_KI; ++i) {
      const float keep = wave + 4 * i < nks ? 1.f : 0.f;
      bf16x8_t zh, zl, wh, wl2;
      split8(za[i] * keep, zb[i] * keep, zh, zl);
      split8(wa[i], wb[i], wh, wl2);
      acc = pmfma(wl2, zh, acc);
      acc = pmfma(wh, zl, acc);
      acc = pmfma(wh, zh, acc);
    }
  }
  *reinterpret_cast<f32x4_t*>(red + (wave * 64 + lane) * 4) = acc;
  __syncthreads();
  if (wave != 0 || !fv) return;
  // lane: ctx[frame f0 + l15][column h*64 + ct*16 + 4g + j]
  const float* r0 = red + lane * 4;
  acc = (*reinterpret_cast<const f32x4_t*>(r0) + *reinterpret_cast<const f32x4_t*>(r0 + 256)) +
        (*reinterpret_cast<const f32x4_t*>(r0 + 512) + *reinterpret_cast<const f32x4_t*>(r0 + 768));
  const int c = h * 64 + ct * 16 + 4 * g;
  acc += *reinterpret_cast<const f32x4_t*>(p.bv + c);
  unsigned hi[4], lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split_bf(acc[j], hi[j], lo[j]);
  const size_t o = (size_t)(f0 + l15) * D + c;
  if (p.ctx_f32) *reinterpret_cast<f32x4_t*>(p.ctx_f32 + o) = acc;
  if (p.ctx_hi) *reinterpret_cast<u32x2_t*>(p.ctx_hi + o) = (u32x2_t){hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16)};
  if (p.ctx_lo) *reinterpret_cast<u32x2_t*>(p.ctx_lo + o) = (u32x2_t){lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16)};
}
hipError_t sf_launch_pool_ctx(const SfPoolCtxArgs& a, hipStream_t s) {
  if (a.heads > 16 || a.D != a.heads * 64 || a.D > 1024 || a.F <= 0 || a.S < 1 || a.S > 8 || (a.S > 1 && !a.ml)) return hipErrorInvalidValue;
  const size_t lds = pool_lds(((size_t)4 * 64 * 4 + (a.S > 1 ? (size_t)16 * (a.D + 4) + 16 * 8 : 0)) * sizeof(float));
  const dim3 grid((a.F + 15) / 16, a.heads * 4);
#define SF_CTX_CASE(KI)                                                                                               \
  {                                                                                                                   \
    static SfPerDeviceOnce once;                                                                                      \
    if (once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_pool_ctx_kernel<KI>),               \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);              \
    hipLaunchKernelGGL(sf_pool_ctx_kernel<KI>, grid, dim3(256), lds, s, a);                                           \
  }
  if (a.heads <= 2) SF_CTX_CASE(1)
  else if (a.heads <= 4) SF_CTX_CASE(2)
  else if (a.heads <= 6) SF_CTX_CASE(3)
  else if (a.heads <= 8) SF_CTX_CASE(4)
  else if (a.heads <= 12) SF_CTX_CASE(6)
  else SF_CTX_CASE(8)
#undef SF_CTX_CASE
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// head tail for one to four rows (the streamed frame): y[f][n] = act(sum_k LN?(x)[f][k] (w_hi + w_lo)[n][k] + b[n]) (+ resid[f][n])
// 4 waves per workgroup, a wave per output column at a time; a lane holds 8 consecutive k per 512-wide pass, KI passes (all
// loads of a column issued before its arithmetic); rows of x in LDS as fp32.
// ------------------------------------------------------------------------------------------------
template <int KI, int FR>
__global__ __launch_bounds__(256) void sf_rowlin_kernel(SfRowLinArgs p, int cpw) {
  extern __shared__ __attribute__((aligned(16))) float xs[];          // [FR][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.K;
  for (int i = tid; i < FR * (K >> 2); i += 256) {
    const int f = i / (K >> 2), c4 = i - f * (K >> 2);
    const int fr = f < p.F ? f : p.F - 1;
    *reinterpret_cast<f32x4_t*>(xs + f * K + c4 * 4) = *reinterpret_cast<const f32x4_t*>(p.x + (size_t)fr * p.ldx + c4 * 4);
  }
  __syncthreads();
  if (p.ln_g) {                      // LayerNorm of the rows, in place (wave f normalises row f; two-pass statistics)
    if (wave < FR) {
      float* r = xs + wave * K;
      float s1 = 0.f;
      for (int k = lane; k < K; k += 64) s1 += r[k];
      const float mean = wave_sum_dpp(s1) / (float)K;
      float s2 = 0.f;
      for (int k = lane; k < K; k += 64) { const float d = r[k] - mean; s2 += d * d; }
      const float rstd = rsqrtf(wave_sum_dpp(s2) / (float)K + p.ln_eps);
      for (int k = lane; k < K; k += 64) r[k] = (r[k] - mean) * rstd * p.ln_g[k] + p.ln_b[k];
    }
    __syncthreads();
  }
  const int n_base = (blockIdx.x * 4 + wave) * cpw;
  for (int ci = 0; ci < cpw; ++ci) {
    const int n = n_base + ci;
    if (n >= p.N) break;
    u32x4_t wh[KI], wl[KI];
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int k = i * 512 + lane * 8;
      const int kc = k + 8 <= K ? k : K - 8;            // past the end: a valid address, the products are masked below
      wh[i] = *reinterpret_cast<const u32x4_t*>(p.w_hi + (size_t)n * K + kc);
      wl[i] = *reinterpret_cast<const u32x4_t*>(p.w_lo + (size_t)n * K + kc);
    }
    float acc[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f) acc[f] = 0.f;
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int k = i * 512 + lane * 8;
      const bool ok = k + 8 <= K;
      const int kc = ok ? k : K - 8;
      float w[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float m = ok ? 1.f : 0.f;
        w[2 * j] = (bf2f(wh[i][j] & 0xffffu) + bf2f(wl[i][j] & 0xffffu)) * m;
        w[2 * j + 1] = (bf2f(wh[i][j] >> 16) + bf2f(wl[i][j] >> 16)) * m;
      }
#pragma unroll
      for (int f = 0; f < FR; ++f) {
        const f32x4_t x0 = *reinterpret_cast<const f32x4_t*>(xs + f * K + kc), x1 = *reinterpret_cast<const f32x4_t*>(xs + f * K + kc + 4);
        acc[f] = fmaf(w[0], x0[0], acc[f]); acc[f] = fmaf(w[1], x0[1], acc[f]); acc[f] = fmaf(w[2], x0[2], acc[f]); acc[f] = fmaf(w[3], x0[3], acc[f]);
        acc[f] = fmaf(w[4], x1[0], acc[f]); acc[f] = fmaf(w[5], x1[1], acc[f]); acc[f] = fmaf(w[6], x1[2], acc[f]); acc[f] = fmaf(w[7], x1[3], acc[f]);
      }
    }
    float* out = p.out_ind ? *p.out_ind : p.out;
#pragma unroll
    for (int f = 0; f < FR; ++f) {
      float v = wave_sum_dpp(acc[f]);
      if (lane == 0 && f < p.F) {
        if (p.bias) v += p.bias[n];
        if (p.act >= 0) v = apply_act(v, p.act);
        if (p.resid) v += p.resid[(size_t)f * p.ldr + n];
        out[(size_t)f * p.ldo + n] = v;
      }
    }
  }
}
bool sf_rowlin_supported(int F, int N, int K) {
  return F >= 1 && F <= 4 && N >= 1 && K >= 8 && K % 8 == 0 && K <= 4096 && (size_t)4 * K * sizeof(float) <= 64 * 1024;
}
hipError_t sf_launch_rowlin(const SfRowLinArgs& a, hipStream_t s) {
  if (!sf_rowlin_supported(a.F, a.N, a.K) || (a.ldx % 4) || !a.w_lo) return hipErrorInvalidValue;
  const int ki = (a.K + 511) / 512;
  const int fr = a.F <= 1 ? 1 : (a.F <= 2 ? 2 : 4);
  int cpw = (a.N + 1023) / 1024;                      // ~256 workgroups of four columns at a time
  const dim3 grid((a.N + 4 * cpw - 1) / (4 * cpw));
  const size_t lds = (size_t)fr * a.K * sizeof(float);
#define SF_RL(KI, FR)                                                                                                         \
  {                                                                                                                           \
    static SfPerDeviceOnce once;                                                                                              \
    if (once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_rowlin_kernel<KI, FR>),                     \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);                       \
    hipLaunchKernelGGL((sf_rowlin_kernel<KI, FR>), grid, dim3(256), lds, s, a, cpw);                                          \
  }
#define SF_RL_F(KI) { if (fr == 1) SF_RL(KI, 1) else if (fr == 2) SF_RL(KI, 2) else SF_RL(KI, 4) }
  if (ki <= 1) SF_RL_F(1)
  else if (ki <= 2) SF_RL_F(2)
  else if (ki <= 4) SF_RL_F(4)
  else if (ki <= 6) SF_RL_F(6)
  else SF_RL_F(8)
#undef SF_RL_F
#undef SF_RL
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
// grid (heads, 8): eight value rows per workgroup; a thread owns 4 columns; frames in a fixed order (deterministic).  The eight dctx values
// of a frame are the same for the whole workgroup: they arrive by scalar loads (no LDS image, no barrier).
__global__ __launch_bounds__(256) void sf_pool_dwv_kernel(const float* __restrict__ dctx, const float* __restrict__ z, float* __restrict__ dwv,
                                                          int ldw, float* __restrict__ dbv, int F, int heads, int D) {
  const int h = blockIdx.x, j0 = blockIdx.y * 8, tid = threadIdx.x;
  const int c4 = tid;
  const bool cv = c4 * 4 < D;
  f32x4_t acc[8];
  float bs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; bs[i] = 0.f; }
  const float* dr = dctx + h * 64 + j0;                       // + f * D: workgroup-uniform
  const float* zc = z + (size_t)h * D + (cv ? c4 * 4 : 0);    // + f * heads * D
#pragma unroll 2
  for (int f = 0; f < F; ++f) {
    const f32x4_t zv = *reinterpret_cast<const f32x4_t*>(zc + (size_t)f * heads * D);
    float d8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) d8[i] = dr[(size_t)f * D + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] += d8[i] * zv; bs[i] += d8[i]; }
  }
  if (cv && dwv) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x4_t* o = reinterpret_cast<f32x4_t*>(dwv + (size_t)(h * 64 + j0 + i) * ldw + c4 * 4);
      *o = *o + acc[i];
    }
  }
  if (tid == 0 && dbv) {
#pragma unroll
    for (int i = 0; i < 8; ++i) dbv[h * 64 + j0 + i] += bs[i];
  }
}
hipError_t sf_launch_pool_ctx_bwd(const float* dctx, const bf16_t* wT, int ldt, int col0, const float* z, float* dz, float* dwv, int ldw,
                                  float* dbv, int F, int heads, int D, hipStream_t s) {
  if (heads > 16 || D != heads * 64 || D > 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sf_pool_dz_kernel, dim3((F + 15) / 16, heads), dim3(256), 0, s, dctx, wT, ldt, col0, dz, F, heads, D);
  if (dwv || dbv) hipLaunchKernelGGL(sf_pool_dwv_kernel, dim3(heads, 8), dim3(256), 0, s, dctx, z, dwv, ldw, dbv, F, heads, D);
  return hipGetLastError();
}

// probe attention backward: grid (F, S2), 256 threads; bf16 operands like every backward GEMM.
// LDS: T = [dz ; U]^T [D][40] bf16 (k = 0..15 dz heads, 16..31 U heads: the A operand of dx), dz rows [16][D + 8] as hi + lo bf16 planes (the
// A operand of dp: two products per k-step), PD = [p | ds] of this workgroup's tokens [tokens][36] fp32, delta[16], {max, sum} of the forward's softmax [16][2].
// KS = k-steps of the dp product (D / 32, rounded up to an instantiated count): all of a token tile's x fragments are loaded before its
// first MFMA, the dz fragments come from LDS — one memory round trip per tile.
#define TP 40
#define PDP 36
template <int KS>
__global__ __launch_bounds__(256) void sf_pool_probe_bwd_kernel(SfPoolBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int f = blockIdx.x, sp = blockIdx.y;
  const int D = p.D, N = p.N, heads = p.heads, ZP = D + 8;
  const int per = (((N + gridDim.y - 1) / gridDim.y) + 15) & ~15;
  const int n0 = sp * per;
  const int n1 = n0 + per < N ? n0 + per : N;
  const int nt = n1 > n0 ? n1 - n0 : 0;
  const int tiles = (nt + 15) >> 4;
  bf16_t* T_hi = reinterpret_cast<bf16_t*>(smem);
  bf16_t* dzs = T_hi + (size_t)D * TP;              // hi plane [16][ZP], then the lo plane
  bf16_t* dzl = dzs + (size_t)16 * ZP;
  float* PD = reinterpret_cast<float*>(dzl + (size_t)16 * ZP);
  float* delta = PD + (size_t)per * PDP;
  float* mls = delta + 16;
  const float* dzf = p.dz + (size_t)f * heads * D;
  const float* zf = p.z + (size_t)f * heads * D;
  // a thread takes 4 columns: 16 + 16 row loads issued together (rows past `heads`: row 0 again, value zeroed), then the images
  for (int c4 = tid; c4 * 4 < D; c4 += 256) {
    f32x4_t vz[16], vu[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int kc = k < heads ? k : 0;
      vz[k] = *reinterpret_cast<const f32x4_t*>(dzf + (size_t)kc * D + c4 * 4);
      vu[k] = *reinterpret_cast<const f32x4_t*>(p.u + (size_t)kc * D + c4 * 4);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float keep = k < heads ? 1.f : 0.f;
      vz[k] *= keep; vu[k] *= keep;
      unsigned h4[4], l4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split_bf(vz[k][j], h4[j], l4[j]);
      *reinterpret_cast<u32x2_t*>(dzs + k * ZP + c4 * 4) = (u32x2_t){h4[0] | (h4[1] << 16), h4[2] | (h4[3] << 16)};
      *reinterpret_cast<u32x2_t*>(dzl + k * ZP + c4 * 4) = (u32x2_t){l4[0] | (l4[1] << 16), l4[2] | (l4[3] << 16)};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {          // T[d][k]: 32 consecutive k per column = four 16-byte stores
      bf16_t* row = T_hi + (size_t)(c4 * 4 + j) * TP;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        *reinterpret_cast<u32x4_t*>(row + q * 8) = (u32x4_t){pack_bf2(vz[q * 8][j], vz[q * 8 + 1][j]), pack_bf2(vz[q * 8 + 2][j], vz[q * 8 + 3][j]),
                                                              pack_bf2(vz[q * 8 + 4][j], vz[q * 8 + 5][j]), pack_bf2(vz[q * 8 + 6][j], vz[q * 8 + 7][j])};
        *reinterpret_cast<u32x4_t*>(row + 16 + q * 8) = (u32x4_t){pack_bf2(vu[q * 8][j], vu[q * 8 + 1][j]), pack_bf2(vu[q * 8 + 2][j], vu[q * 8 + 3][j]),
                                                                   pack_bf2(vu[q * 8 + 4][j], vu[q * 8 + 5][j]), pack_bf2(vu[q * 8 + 6][j], vu[q * 8 + 7][j])};
      }
    }
  }
  for (int h = wave; h < 16; h += 4) {
    float t = 0.f;
    if (h < heads)
      for (int d = lane; d < D; d += 64) t = fmaf(dzf[(size_t)h * D + d], zf[(size_t)h * D + d], t);
    t = wave_sum(t);
    if (lane == 0) delta[h] = t;
  }
  if (tid < 16) {      // {max, sum} of the forward's softmax for head tid, combined over the forward's token splits
    float m = 0.f, L = 1.f;
    if (p.ml && tid < heads) {
      const int S = p.ml_splits > 0 ? p.ml_splits : 1;
      m = -INFINITY;
      for (int sidx = 0; sidx < S; ++sidx) m = fmaxf(m, p.ml[(((size_t)f * S + sidx) * heads + tid) * 2]);
      L = 0.f;
      for (int sidx = 0; sidx < S; ++sidx) {
        const float* o = p.ml + (((size_t)f * S + sidx) * heads + tid) * 2;
        if (o[1] > 0.f) L += o[1] * __builtin_amdgcn_exp2f((o[0] - m) * LOG2E);
      }
    }
    mls[2 * tid] = m; mls[2 * tid + 1] = L;
  }
  __syncthreads();
  const int nks = D >> 5;
  for (int t = wave; t < tiles; t += 4) {
    // ---- dp[head][token] = dz[head] . x[token]  (x: the saved bf16 normalised tokens) -----------------------------
    const int tl = t * 16 + l15;                 // token inside this workgroup's range
    const int tok = n0 + tl;
    const bool tv = tok < n1;
    const size_t row = (size_t)f * N + (tv ? tok : N - 1);
    const bf16_t* xr = p.x_bf + row * D + g * 8;
    bf16x8_t xb[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) xb[i] = *reinterpret_cast<const bf16x8_t*>(xr + (i < nks ? i : nks - 1) * 32);
    const bf16_t* ar = dzs + l15 * ZP + g * 8;
    const bf16_t* arl = dzl + l15 * ZP + g * 8;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KS; ++i) {               // k-steps past the end (ragged instantiation) multiply a zeroed fragment
      const bf16x8_t z8 = {0, 0, 0, 0, 0, 0, 0, 0};
      const int kc = i < nks ? i : nks - 1;
      const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(ar + kc * 32);
      const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(arl + kc * 32);
      acc = pmfma(i < nks ? a0 : z8, xb[i], acc);            // dz = hi + lo: two products, x is stored in bf16
      acc2 = pmfma(i < nks ? a1 : z8, xb[i], acc2);
    }
    acc += acc2;
    // lane: dp[head 4g + j][token l15]
    float pr[4], ds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int h = 4 * g + j;
      const bool ok = tv && h < heads;
      float pv = ok ? p.probs[((size_t)f * heads + h) * N + tok] : 0.f;
      if (p.probs_raw) pv = ok ? __builtin_amdgcn_exp2f((pv - mls[2 * h]) * LOG2E) / mls[2 * h + 1] : 0.f;     // raw score -> probability
      pr[j] = pv;
      ds[j] = ok ? pv * (acc[j] - delta[h]) : 0.f;
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
    for (int dt = 0; dt * 16 < D; dt += 4) {           // D % 64 == 0: four column tiles per trip
      f32x4_t o4[4], l4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bf16x8_t th = *reinterpret_cast<const bf16x8_t*>(T_hi + (size_t)((dt + q) * 16 + l15) * TP + 8 * g);
        o4[q] = pmfma(th, bl, (f32x4_t){0.f, 0.f, 0.f, 0.f});
        o4[q] = pmfma(th, bh, o4[q]);
        if (lrow) l4[q] = *reinterpret_cast<const f32x4_t*>(lrow + (dt + q) * 16);
      }
      if (tv) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (lrow) o4[q] += l4[q];
          *reinterpret_cast<f32x4_t*>(orow + (dt + q) * 16) = o4[q];
        }
      }
    }
  }
}
hipError_t sf_launch_pool_probe_bwd(const SfPoolBwdArgs& a, hipStream_t s) {
  if (a.heads > 16 || a.D != a.heads * 64 || a.D > 1024 || a.F <= 0 || a.N <= 0) return hipErrorInvalidValue;
  if (a.probs_raw && !a.ml) return hipErrorInvalidValue;
  int S2 = 1;
  while (S2 < 4 && a.F * S2 < 256 && (a.N + 2 * S2 - 1) / (2 * S2) >= 16) S2 *= 2;
  auto lds_for = [&](int s2) {
    const int per = (((a.N + s2 - 1) / s2) + 15) & ~15;
    return ((size_t)a.D * TP + (size_t)2 * 16 * (a.D + 8)) * sizeof(bf16_t) + ((size_t)per * PDP + 16 + 32) * sizeof(float);
  };
  while (lds_for(S2) > 160 * 1024 && S2 < 16 && (a.N + S2 - 1) / S2 > 16) S2 *= 2;      // D = 1024: the images take 148 KB, more token splits shrink the rest
  const size_t lds = lds_for(S2);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const dim3 grid(a.F, S2);
#define SF_PB_CASE(KS)                                                                                                 \
  {                                                                                                                    \
    static SfPerDeviceOnce once;                                                                                       \
    if (once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_pool_probe_bwd_kernel<KS>),          \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);               \
    hipLaunchKernelGGL(sf_pool_probe_bwd_kernel<KS>, grid, dim3(256), lds, s, a);                                      \
  }
  const int nks = a.D / 32;
  if (nks <= 4) SF_PB_CASE(4)
  else if (nks <= 8) SF_PB_CASE(8)
  else if (nks <= 16) SF_PB_CASE(16)
  else if (nks <= 24) SF_PB_CASE(24)
  else SF_PB_CASE(32)
#undef SF_PB_CASE
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

// ------------------------------------------------------------------------------------------------------------------------------
// Generic widths (round 6): head_dim != 64 or D > 1024 (SigLIP-so400m: 1152 / 16 = 72; configuration_streamformer.py:90-135 takes any
// hidden_size / heads).  The same algebra as above — scores = x . U_h, z_h = sum_n p_hn x_n, ctx_h = Wv_h z_h + bv_h
// (modeling:1141-1154 with the key / value projections folded away) — as plain fp32 FMAs in three small launches, scratch in global memory:
//   1. scores[f][h][n] = x[f, n] . U_h             grid (token chunks, F): a wave per token, lanes over D, one wave reduction per head
//   2. z[f][h][d] = sum_n softmax(scores)[h][n] x[f, n][d]     grid (D / 256, F): the frame's softmax recomputed per workgroup (heads x N
//                                                   exps), a thread per d with every head's accumulator in registers
//   3. ctx[f][c] = bv[c] + Wv[c] . z[f][c / head_dim]          grid (D / 64, F): a wave per output column, lanes over D
// A fallback, not a tuned kernel (first version: one workgroup per frame for everything, 5.0 ms per call at 64 frames of 256 tokens).
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sf_pool_gen_scores_kernel(SfPoolGenArgs p) {
  const int f = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= p.N) return;
  const float* xr = (p.x_ind ? *p.x_ind : p.x) + ((size_t)f * p.N + n) * p.D;
  for (int h = 0; h < p.heads; ++h) {
    const float* ur = p.u + (size_t)h * p.D;
    float t = 0.f;
    for (int d = lane; d < p.D; d += 64) t = fmaf(xr[d], ur[d], t);
    t = wave_sum(t);
    if (lane == 0) p.scores[((size_t)f * p.heads + h) * p.N + n] = t;
  }
}

__global__ __launch_bounds__(256) void sf_pool_gen_z_kernel(SfPoolGenArgs p) {
  extern __shared__ __attribute__((aligned(16))) float pg_smem[];      // [heads][N] probabilities of the frame
  const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = p.N, H = p.heads, D = p.D;
  const float* scg = p.scores + (size_t)f * H * N;
  for (int h = wave; h < H; h += 4) {               // U already carries the 1 / sqrt(head_dim) of the query
    float m = -INFINITY;
    for (int n = lane; n < N; n += 64) m = fmaxf(m, scg[h * N + n]);
    m = wave_max(m);
    float su = 0.f;
    for (int n = lane; n < N; n += 64) { const float e = expf(scg[h * N + n] - m); pg_smem[h * N + n] = e; su += e; }
    su = wave_sum(su);
    const float inv = 1.0f / su;
    for (int n = lane; n < N; n += 64) pg_smem[h * N + n] *= inv;
  }
  __syncthreads();
  const int d = blockIdx.x * 256 + tid;
  if (d >= D) return;
  const float* x = (p.x_ind ? *p.x_ind : p.x) + (size_t)f * N * D;
  float acc[16];
#pragma unroll
  for (int h = 0; h < 16; ++h) acc[h] = 0.f;
  for (int n = 0; n < N; ++n) {
    const float xv = x[(size_t)n * D + d];
#pragma unroll
    for (int h = 0; h < 16; ++h)
      if (h < H) acc[h] = fmaf(pg_smem[h * N + n], xv, acc[h]);
  }
#pragma unroll
  for (int h = 0; h < 16; ++h)
    if (h < H) p.z[((size_t)f * H + h) * D + d] = acc[h];
}

__global__ __launch_bounds__(256) void sf_pool_gen_ctx_kernel(SfPoolGenArgs p) {
  const int f = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = p.D;
  for (int c = blockIdx.x * 64 + wave; c < min(D, (int)blockIdx.x * 64 + 64); c += 4) {
    const float* wr = p.wv + (size_t)c * p.ldw;
    const float* zr = p.z + ((size_t)f * p.heads + c / p.hd) * D;
    float t = 0.f;
    for (int d = lane; d < D; d += 64) t = fmaf(wr[d], zr[d], t);
    t = wave_sum(t);
    if (lane == 0) {
      t += p.bv[c];
      const size_t o = (size_t)f * D + c;
      if (p.ctx_f32) p.ctx_f32[o] = t;
      if (p.ctx_hi) {
        unsigned int hi, lo;
        split_bf(t, hi, lo);
        p.ctx_hi[o] = (bf16_t)hi;
        if (p.ctx_lo) p.ctx_lo[o] = (bf16_t)lo;
      }
    }
  }
}

bool sf_pool_generic_supported(int N, int heads, int D) {
  return heads >= 1 && heads <= 16 && N >= 1 && D >= 1 && (size_t)heads * N * sizeof(float) <= (size_t)160 * 1024;
}
size_t sf_pool_generic_scratch_floats(int F, int N, int heads, int D) { return (size_t)F * heads * ((size_t)N + D); }
hipError_t sf_launch_pool_generic(const SfPoolGenArgs& a_in, hipStream_t s) {
  SfPoolGenArgs a = a_in;
  if (!sf_pool_generic_supported(a.N, a.heads, a.D) || a.F <= 0 || a.hd <= 0 || a.D != a.heads * a.hd || !a.scratch) return hipErrorInvalidValue;
  a.scores = a.scratch;                                      // [F][heads][N]
  a.z = a.scratch + (size_t)a.F * a.heads * a.N;             // [F][heads][D]
  const size_t lds = (size_t)a.heads * a.N * sizeof(float);
  static SfPerDeviceOnce attr_set;
  if (attr_set.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_pool_gen_z_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(sf_pool_gen_scores_kernel, dim3((a.N + 3) / 4, a.F), dim3(256), 0, s, a);
  hipLaunchKernelGGL(sf_pool_gen_z_kernel, dim3((a.D + 255) / 256, a.F), dim3(256), lds, s, a);
  hipLaunchKernelGGL(sf_pool_gen_ctx_kernel, dim3((a.D + 63) / 64, a.F), dim3(256), 0, s, a);
  return hipGetLastError();
}
