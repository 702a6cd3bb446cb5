// Attention for head widths the tuned kernels of sf_attention.hip (head_dim 64) do not cover — the reference's config takes any
// hidden_size / num_attention_heads (models/configuration_streamformer.py:90-135; SigLIP-so400m: 1152 / 16 = 72).  Same contracts as
// sf_launch_spatial_attention / sf_launch_temporal_attention (modeling:688-717 / 575-615, cache + single query: vqa_enc:491-560), any
// head_dim that is a multiple of 8 up to 128, bf16 / fp32 / hi + lo plane inputs, both compute modes.
//
// One wave = one (sequence, head, 16-query tile), flash style over 16-key tiles, everything in fp32 on the f32 matrix pipe
// (v_mfma_f32_16x16x4_f32: exact fp32 products, 1/16 of the bf16 rate — attention is a few percent of the encoder's FLOPs, and one
// kernel then serves both compute modes with no operand rounding at all).  No LDS, no transposes: the two products are arranged so
// that every operand is something a lane can load straight from the token rows, and the query index of a lane is l15 in BOTH results:
//   S^T[key][query] = sum_d K[key][d] Q[query][d]          A = K: lane (key l15, slot g), B = Q: lane (query l15, slot g); the k-slot g of
//                                                          MFMA step s stands for d = g * (HD / 4) + s (any bijection does: it is a sum),
//                                                          so a lane's K / Q elements are HD / 4 CONTIGUOUS values of its row
//                                                          -> lane (query l15, g) holds the scores of keys 4 g .. 4 g + 3
//   O^T[d][query]   = sum_key V[key][d] P[query][key]      B = P^T: lane (query l15, g), step r: key 4 g + r — exactly the register r the
//                                                          lane holds; A = V: lane (dim l15 of the 16-dim tile, g): V[4 g + r][..]
//                                                          -> lane (query l15, g) holds dims 16 t + 4 g .. + 3 of output tile t
// Row maximum / sum of a query: its 16 scores of a tile sit in 4 registers x the 4 lanes {l15, l15 + 16, + 32, + 48}: two xor-shuffles.
#include "sf_common.h"

typedef __attribute__((ext_vector_type(4))) float gf4_t;

SF_DEVICE gf4_t ga_mfma(float a, float b, gf4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// element i of a row-major token buffer: KIND 0 = bf16, 1 = fp32, 2 = hi + lo bf16 planes (lo plane `lo_off` elements behind hi)
template <int KIND>
SF_DEVICE float ga_ld(const void* base, size_t i, long long lo_off) {
  if (KIND == 1) return reinterpret_cast<const float*>(base)[i];
  const bf16_t* b = reinterpret_cast<const bf16_t*>(base);
  float v = bf2f(b[i]);
  if (KIND == 2) v += bf2f(b[i + lo_off]);
  return v;
}

struct SfGenAttn {
  SfAttnArgs a;
  int temporal;          // 0: sequence = frame f, rows f * N + t;  1: sequence = (b, n), rows as in SfAttnArgs
  int hd;                // head_dim
  int nseq, Lq, qtiles;
};

template <int KIND, int HDQ>      // HDQ = head_dim / 4 <= 32
__global__ __launch_bounds__(256) void sf_attention_generic_kernel(SfGenAttn p) {
  const SfAttnArgs& a = p.a;
  const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int total = p.nseq * a.heads * p.qtiles;
  if (wid >= total) return;
  const int qt = wid % p.qtiles, h = (wid / p.qtiles) % a.heads, seq = wid / (p.qtiles * a.heads);
  constexpr int HD = HDQ * 4;
  constexpr int NT = (HD + 15) / 16;               // 16-dim output tiles
  int Lk = p.temporal ? a.Tk : a.N, q_t0 = a.q_t0, t_past = a.t_past;
  if (p.temporal && a.pos_dev) {                   // streamed frame inside the position-free graph: {slot, keys}, every key visible
    q_t0 = a.pos_dev[0];
    Lk = a.pos_dev[1];
    t_past = Lk - 1;
  }
  const int b = p.temporal ? seq / a.N : 0, n = p.temporal ? seq % a.N : 0;
  auto qrow = [&](int t) -> size_t { return p.temporal ? ((size_t)b * a.Tq_cap + q_t0 + t) * a.N + n : (size_t)seq * a.N + t; };
  auto krow = [&](int t) -> size_t { return p.temporal ? ((size_t)b * a.Tcap + t) * a.N + n : (size_t)seq * a.N + t; };
  auto orow = [&](int t) -> size_t { return p.temporal ? ((size_t)b * a.Tq + t) * a.N + n : (size_t)seq * a.N + t; };

  // this lane's query (l15 of the tile) and its HDQ contiguous dims
  const int qi = qt * 16 + l15;
  const int qic = qi < p.Lq ? qi : p.Lq - 1;
  float qreg[HDQ];
  {
    const size_t o = qrow(qic) * a.row_pitch_q + h * HD + g * HDQ;
#pragma unroll
    for (int s = 0; s < HDQ; ++s) qreg[s] = ga_ld<KIND>(a.q, o + s, a.lo_plane_off);
  }
  const float c2 = a.scale * 1.44269504088896340736f;
  gf4_t o_acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) o_acc[t] = (gf4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  // causal: query t sits at absolute frame t_past + t and sees keys <= that; tiles wholly behind the last visible key are skipped
  const int last_q = min(qt * 16 + 15, p.Lq - 1);
  const int k_end = (p.temporal && a.causal) ? min(Lk, t_past + last_q + 1) : Lk;
  for (int k0 = 0; k0 < k_end; k0 += 16) {
    // ---- S^T tile: keys k0 + l15 (A operand), HDQ steps --------------------------------------------------------------------------
    const int kj = min(k0 + l15, Lk - 1);
    gf4_t s4 = {0.f, 0.f, 0.f, 0.f};
    {
      const size_t o = krow(kj) * a.row_pitch_kv + h * HD + g * HDQ;
#pragma unroll
      for (int s = 0; s < HDQ; ++s) s4 = ga_mfma(ga_ld<KIND>(a.k, o + s, a.lo_plane_off), qreg[s], s4);
    }
    // lane (query l15, g): keys k0 + 4 g + r
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = k0 + 4 * g + r;
      const bool ok = key < Lk && (!(p.temporal && a.causal) || key <= t_past + qi);
      s4[r] = ok ? s4[r] : -INFINITY;
      mx = fmaxf(mx, s4[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    // a query with no visible key yet (m_new = -inf) keeps zeros: exp2(-inf - (-inf)) would be NaN
    const float corr = m_new == -INFINITY ? 1.f : __builtin_amdgcn_exp2f((m_run - m_new) * c2);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s4[r] = m_new == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((s4[r] - m_new) * c2);
      psum += s4[r];
    }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * corr + psum;
    m_run = m_new;
    // ---- O^T += V^T P^T: 4 steps per 16-dim tile --------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      o_acc[t] *= corr;
      const int d = t * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = min(k0 + 4 * g + r, Lk - 1);          // masked keys carry p = 0
        const float v = d < HD ? ga_ld<KIND>(a.v, krow(key) * a.row_pitch_kv + h * HD + d, a.lo_plane_off) : 0.f;
        o_acc[t] = ga_mfma(v, s4[r], o_acc[t]);
      }
    }
  }
  if (qi >= p.Lq) return;
  const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
  const size_t ob = orow(qi) * a.D + h * HD;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int d = t * 16 + 4 * g;
    if (d < HD) {                                   // HD % 4 == 0: the four dims of a lane are all inside or all outside
      unsigned int hb[4], lb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) split_bf(o_acc[t][r] * inv, hb[r], lb[r]);
      *reinterpret_cast<u32x2_t*>(a.ctx_hi + ob + d) = (u32x2_t){hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16)};
      if (a.ctx_lo) *reinterpret_cast<u32x2_t*>(a.ctx_lo + ob + d) = (u32x2_t){lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16)};
    }
  }
}

bool sf_attention_generic_supported(const SfAttnArgs& a, int head_dim) {
  return head_dim >= 8 && head_dim <= 128 && head_dim % 8 == 0 && a.D == a.heads * head_dim && !a.probs && !a.lse2_out && !a.drop.on;
}

template <int KIND>
static hipError_t ga_launch(const SfGenAttn& p, hipStream_t s) {
  const int total = p.nseq * p.a.heads * p.qtiles;
  const dim3 grid((total + 3) / 4), block(256);
  switch (p.hd / 8) {
#define GA_CASE(E) case E: hipLaunchKernelGGL((sf_attention_generic_kernel<KIND, 2 * E>), grid, block, 0, s, p); break;
    GA_CASE(1) GA_CASE(2) GA_CASE(3) GA_CASE(4) GA_CASE(5) GA_CASE(6) GA_CASE(7) GA_CASE(8)
    GA_CASE(9) GA_CASE(10) GA_CASE(11) GA_CASE(12) GA_CASE(13) GA_CASE(14) GA_CASE(15) GA_CASE(16)
#undef GA_CASE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t sf_launch_attention_generic(const SfAttnArgs& a, int head_dim, bool temporal, hipStream_t s) {
  if (!sf_attention_generic_supported(a, head_dim)) return hipErrorInvalidValue;
  SfGenAttn p;
  p.a = a; p.temporal = temporal ? 1 : 0; p.hd = head_dim;
  if (temporal) {
    if (a.B <= 0 || a.N <= 0 || a.Tq <= 0 || a.Tk <= 0) return hipErrorInvalidValue;
    p.nseq = a.B * a.N; p.Lq = a.Tq;
  } else {
    if (a.frames <= 0 || a.N <= 0) return hipErrorInvalidValue;
    p.nseq = a.frames; p.Lq = a.N;
  }
  p.qtiles = (p.Lq + 15) / 16;
  if (a.in_is_f32) return ga_launch<1>(p, s);
  if (a.lo_plane_off > 0) return ga_launch<2>(p, s);
  return ga_launch<0>(p, s);
}
