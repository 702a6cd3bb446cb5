// Attention for head widths the tuned kernels of sf_attention.hip (head_dim 64) do not cover — the reference's config takes any
// hidden_size / num_attention_heads (models/configuration_streamformer.py:90-135; SigLIP-so400m: 1152 / 16 = 72).  Same contracts as
// sf_launch_spatial_attention / sf_launch_temporal_attention (modeling:688-717 / 575-615, cache + single query: vqa_enc:491-560), any
// head_dim that is a multiple of 8 up to 128, bf16 / fp32 / hi + lo plane inputs, both compute modes.
//
// One wave = one (sequence, head, 16-query tile), flash style over 16-key tiles, everything in fp32 on the f32 matrix pipe
// (v_mfma_f32_16x16x4_f32: exact fp32 products, 1/16 of the bf16 rate — attention is a few percent of the encoder's FLOPs, and one
// kernel then serves both compute modes with no operand rounding at all).  No transposes and no workgroup barriers: the two products are
// arranged so that every operand is a single float at (row, slot) of the K / V tile — staged once per tile in a wave-private fp32 LDS
// image by 16-byte row loads — and the query index of a lane is l15 in BOTH results:
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

// eight consecutive elements starting at element i (i % 8 == 0: 16-byte aligned for bf16, 32 for fp32)
template <int KIND>
SF_DEVICE void ga_ld8(const void* base, size_t i, long long lo_off, float* out) {
  if (KIND == 1) {
    const gf4_t v0 = *reinterpret_cast<const gf4_t*>(reinterpret_cast<const float*>(base) + i);
    const gf4_t v1 = *reinterpret_cast<const gf4_t*>(reinterpret_cast<const float*>(base) + i + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { out[j] = v0[j]; out[4 + j] = v1[j]; }
  } else {
    const u32x4_t h = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const bf16_t*>(base) + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) { out[2 * j] = bf2f(h[j] & 0xffffu); out[2 * j + 1] = __uint_as_float(h[j] & 0xffff0000u); }
    if (KIND == 2) {
      const u32x4_t l = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const bf16_t*>(base) + i + lo_off);
#pragma unroll
      for (int j = 0; j < 4; ++j) { out[2 * j] += bf2f(l[j] & 0xffffu); out[2 * j + 1] += __uint_as_float(l[j] & 0xffff0000u); }
    }
  }
}

template <int KIND, int HDQ>      // HDQ = head_dim / 4 <= 32
__global__ __launch_bounds__(256) void sf_attention_generic_kernel(SfGenAttn p) {
  const SfAttnArgs& a = p.a;
  // wave-private fp32 images of the current K and V tiles ([16 keys][HD + 4]): the rows arrive by 16-byte loads (a key row of one head is
  // contiguous) and the MFMA operands — single floats at (row l15, slot) — are LDS reads.  (First version: every operand element
  // loaded from global memory by its lane, 2 bytes per lane and instruction: 520 us per so400m-shaped spatial launch, load-bound.)
  extern __shared__ __attribute__((aligned(16))) float ga_smem[];
  constexpr int GA_LD = HDQ * 4 + 4;
  float* kt = ga_smem + (threadIdx.x >> 6) * (2 * 16 * GA_LD);
  float* vt = kt + 16 * GA_LD;
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
    // ---- stage the tile's K and V rows (keys past the end: the last key again; they are masked below) ----------------------------
    {
      constexpr int CH = HD / 8;                     // 8-element chunks per row
#pragma unroll
      for (int it = 0; it < (16 * CH + 63) / 64; ++it) {
        const int c = it * 64 + lane;
        if (c < 16 * CH) {
          const int row = c / CH, cc = c % CH;
          const size_t o = krow(min(k0 + row, Lk - 1)) * a.row_pitch_kv + h * HD + cc * 8;
          float e[8];
          ga_ld8<KIND>(a.k, o, a.lo_plane_off, e);
          *reinterpret_cast<gf4_t*>(kt + row * GA_LD + cc * 8) = (gf4_t){e[0], e[1], e[2], e[3]};
          *reinterpret_cast<gf4_t*>(kt + row * GA_LD + cc * 8 + 4) = (gf4_t){e[4], e[5], e[6], e[7]};
          ga_ld8<KIND>(a.v, o, a.lo_plane_off, e);
          *reinterpret_cast<gf4_t*>(vt + row * GA_LD + cc * 8) = (gf4_t){e[0], e[1], e[2], e[3]};
          *reinterpret_cast<gf4_t*>(vt + row * GA_LD + cc * 8 + 4) = (gf4_t){e[4], e[5], e[6], e[7]};
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own LDS writes, read below by other lanes of the same wave
    }
    // ---- S^T tile: keys k0 + l15 (A operand), HDQ steps --------------------------------------------------------------------------
    gf4_t s4 = {0.f, 0.f, 0.f, 0.f};
    {
      const float* kr = kt + l15 * GA_LD + g * HDQ;
#pragma unroll
      for (int s = 0; s < HDQ; ++s) s4 = ga_mfma(kr[s], qreg[s], s4);
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
        const float v = d < HD ? vt[(4 * g + r) * GA_LD + d] : 0.f;          // masked keys carry p = 0
        o_acc[t] = ga_mfma(v, s4[r], o_acc[t]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // every read of the images retired before the next tile overwrites them
  }
  const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
  if (a.probs && !p.temporal) {
    // output_attentions (modeling:703-716): a second sweep over the key tiles with the final row statistics writes the fp32
    // probabilities [frame, head, query, key] (K tiles only; every lane of the wave takes part in the staging)
    for (int k0 = 0; k0 < Lk; k0 += 16) {
      constexpr int CH = HD / 8;
#pragma unroll
      for (int it = 0; it < (16 * CH + 63) / 64; ++it) {
        const int c = it * 64 + lane;
        if (c < 16 * CH) {
          const int row = c / CH, cc = c % CH;
          float e[8];
          ga_ld8<KIND>(a.k, krow(min(k0 + row, Lk - 1)) * a.row_pitch_kv + h * HD + cc * 8, a.lo_plane_off, e);
          *reinterpret_cast<gf4_t*>(kt + row * GA_LD + cc * 8) = (gf4_t){e[0], e[1], e[2], e[3]};
          *reinterpret_cast<gf4_t*>(kt + row * GA_LD + cc * 8 + 4) = (gf4_t){e[4], e[5], e[6], e[7]};
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      gf4_t s4 = {0.f, 0.f, 0.f, 0.f};
      const float* kr = kt + l15 * GA_LD + g * HDQ;
#pragma unroll
      for (int s = 0; s < HDQ; ++s) s4 = ga_mfma(kr[s], qreg[s], s4);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (qi < p.Lq) {
        float* prow = a.probs + (((size_t)seq * a.heads + h) * a.N + qi) * (size_t)a.N;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = k0 + 4 * g + r;
          if (key < Lk) prow[key] = __builtin_amdgcn_exp2f((s4[r] - m_run) * c2) * inv;
        }
      }
    }
  }
  if (qi >= p.Lq) return;
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
  return head_dim >= 8 && head_dim <= 128 && head_dim % 8 == 0 && a.D == a.heads * head_dim && !a.lse2_out && !a.drop.on;
}

template <int KIND>
static hipError_t ga_launch(const SfGenAttn& p, hipStream_t s) {
  const int total = p.nseq * p.a.heads * p.qtiles;
  const dim3 grid((total + 3) / 4), block(256);
  const size_t lds = (size_t)4 * 2 * 16 * (p.hd + 4) * sizeof(float);      // <= 67.6 KB at head_dim 128
  static SfPerDeviceOnce attr_set[3];
  if (attr_set[KIND].first()) {
#define GA_ATTR(E) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_attention_generic_kernel<KIND, 2 * E>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    GA_ATTR(13) GA_ATTR(14) GA_ATTR(15) GA_ATTR(16)
#undef GA_ATTR
  }
  switch (p.hd / 8) {
#define GA_CASE(E) case E: hipLaunchKernelGGL((sf_attention_generic_kernel<KIND, 2 * E>), grid, block, lds, s, p); break;
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
    if (a.B <= 0 || a.N <= 0 || a.Tq <= 0 || a.Tk <= 0 || a.probs) return hipErrorInvalidValue;
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
