// "Skinny" MFMA GEMM for few token rows (M <= 2560, sf_skinny_max_rows()): the per-frame streaming step (M = 196 rows per
// call, vqa_enc:1316-1392), the pooling-head MLP (M = frames) and small test shapes.
//   C[M,N] = A[M,K] * W[N,K]^T (+ the same fused epilogues as sf_gemm.hip)
//
// Why a separate kernel: at M = 196 the 128x128 tiling yields 12-48 workgroups on 256 CUs and every
// Linear of the streaming step costs 18-60 us although its weight matrix (1.2-4.7 MB) streams in ~1 us.
// These launches are latency-bound (Little's law on the bytes a CU keeps in flight), so the tile is
// made SMALL and the ring DEEP:
//   * a workgroup owns a [32 rows x 32 columns] output tile and the whole K range: N/32 x ceil(M/32)
//     workgroups (168 for N = 768 at one frame, 672 for N = 3072); no split-K, so the result is
//     deterministic and every epilogue stays fused;
//   * 4 waves, one 16x16 MFMA tile each;
//   * operands go HBM/L2 -> LDS by global_load_lds into an 8-stage ring of [32+32 rows x 64 k] tiles
//     (8 KB per stage), seven K-tiles in flight, ONE barrier per K-tile, counted vmcnt;
//   * A is re-read by the N/32 column workgroups from L2 (it is 0.3-1.2 MB), W by the M/32 row groups.
// SPLIT = the fp32-accurate bf16x3 mode (hi/lo planes of both operands, three MFMAs per fragment pair).
#include "sf_common.h"
#include "sf_switches.h"
#include <cstdlib>

#define SK_BM 32
#define SK_BN 32
#define SK_BK 64
#define SK_THREADS 256
#define SK_STAGES 8
#define SK_ROWS (SK_BM + SK_BN)               // A rows then W rows in one stage image
#define SK_PLANE (SK_ROWS * SK_BK * 2)        // 8 KB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

SF_DEVICE f32x4_t sk_mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}
SF_DEVICE bf16x8_t sk_frag(const char* img, int row, int kc) {
  return *reinterpret_cast<const bf16x8_t*>(img + row * (SK_BK * 2) + ((kc ^ ((row >> 1) & 7)) << 4));
}
template <int N>
SF_DEVICE void sk_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LayerNorm folded into the consumer at small M (LNF): sf_lnf_stats / sf_lnf_finish in sf_common.h (shared with sf_stream_fused.hip)
SF_DEVICE void sk_stats(const bf16x8_t& f, float& s1, float& s2) { sf_lnf_stats(f, s1, s2); }
SF_DEVICE void sk_ln_finish(float s1, float s2, int K, float eps, float& mean, float& rstd) { sf_lnf_finish(s1, s2, K, eps, mean, rstd); }

// residual producers of the small-M LayerNorm fold: the A operand of the folded Linear that follows — bf16(x) in bf16 mode, the hi + lo
// planes of x in the accurate mode (out_lo; round 6)
template <bool SPLIT>
SF_DEVICE void sk_fold_copy(const SfGemmArgs& p, size_t o, const f32x4_t& v) {
  if (!SPLIT) {
    *reinterpret_cast<u32x2_t*>(p.out_hi + o) = (u32x2_t){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
  } else {
    unsigned int h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_bf(v[j], h[j], l[j]);
    *reinterpret_cast<u32x2_t*>(p.out_hi + o) = (u32x2_t){h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
    if (p.out_lo) *reinterpret_cast<u32x2_t*>(p.out_lo + o) = (u32x2_t){l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
  }
}

// TPS = K-tiles consumed per barrier (1 or 2): the loop is a serial chain of wait -> barrier -> LDS reads -> MFMA, a few
// hundred cycles per step with two MFMAs of work in it, so halving the step count (12 -> 6 at K = 768) is worth more than
// the one stage of prefetch depth it costs (STAGES - TPS tiles in flight instead of STAGES - 1).
template <bool SPLIT, int EPI, bool LNF = false, int TPS = 1>
__global__ __launch_bounds__(SK_THREADS) void sf_gemm_skinny_kernel(SfGemmArgs p) {
  constexpr int STAGE = SK_PLANE * (SPLIT ? 2 : 1);     // hi plane (+ lo plane)
  constexpr int LOADS = SK_ROWS * 8 / SK_THREADS;       // 16-byte chunks per thread per plane = 2
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * SK_BN, m0 = blockIdx.y * SK_BM;

  // per-thread DMA sources: chunk c of the stage image = (row, 16-byte slot); rows 0..31 = A, 32..63 = W
  const bf16_t* src_hi[LOADS];
  const bf16_t* src_lo[LOADS];
#pragma unroll
  for (int i = 0; i < LOADS; ++i) {
    const int c = i * SK_THREADS + tid;
    const int row = c >> 3, slot = c & 7;
    const int kc = slot ^ ((row >> 1) & 7);
    if (row < SK_BM) {
      int gr = m0 + row;
      gr = gr < p.M ? gr : p.M - 1;
      src_hi[i] = p.a_hi + (size_t)gr * p.K + kc * 8;
      src_lo[i] = SPLIT ? p.a_lo + (size_t)gr * p.K + kc * 8 : nullptr;
    } else {
      int gr = n0 + row - SK_BM;
      gr = gr < p.N ? gr : p.N - 1;
      src_hi[i] = p.w_hi + (size_t)gr * p.K + kc * 8;
      src_lo[i] = SPLIT ? p.w_lo + (size_t)gr * p.K + kc * 8 : nullptr;
    }
  }
  const bool w_nt = p.w_nt != 0;
  auto issue = [&](int kt) {
    char* dst = smem + (kt % SK_STAGES) * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      if (i == 1 && w_nt) __builtin_amdgcn_global_load_lds((gptr_t)(src_hi[i] + kt * SK_BK), (lptr_t)(dst + i * 4096), 16, 0, 2);   // W rows, nt
      else __builtin_amdgcn_global_load_lds((gptr_t)(src_hi[i] + kt * SK_BK), (lptr_t)(dst + i * 4096), 16, 0, 0);
      if (SPLIT) __builtin_amdgcn_global_load_lds((gptr_t)(src_lo[i] + kt * SK_BK), (lptr_t)(dst + SK_PLANE + i * 4096), 16, 0, 0);
    }
  };

  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  float ln1 = 0.f, ln2 = 0.f;
  const int nkt = p.K / SK_BK;
  constexpr int PER = LOADS * (SPLIT ? 2 : 1);            // load instructions per stage per thread
  const int mt = wave & 1, nt = wave >> 1;                // this wave's 16x16 output tile

  for (int s = 0; s < SK_STAGES - TPS && s < nkt; ++s) issue(s);
  // epilogue operands (bias, folded-LayerNorm column sums, residual row) requested NOW, behind the first DMA stages: at one
  // frame the launch is a chain of memory latencies, and each of these loads used to add its own at the very end
  const int n_e = n0 + nt * 16 + g * 4, m_e = m0 + mt * 16 + l15;
  const bool live_e = n_e < p.N && m_e < p.M;
  f32x4_t pre_bias = {0.f, 0.f, 0.f, 0.f}, pre_lns = {0.f, 0.f, 0.f, 0.f}, pre_res = {0.f, 0.f, 0.f, 0.f};
  if (live_e) {
    if (p.bias) pre_bias = *reinterpret_cast<const f32x4_t*>(p.bias + n_e);
    if (LNF) pre_lns = *reinterpret_cast<const f32x4_t*>(p.ln_s + n_e);
    if (EPI == SF_EPI_RESID_F32 && p.grp_rows <= 0) pre_res = *reinterpret_cast<const f32x4_t*>(p.resid + (size_t)m_e * p.ldc + n_e);
  }
  for (int kt = 0; kt < nkt; kt += TPS) {
    // tiles kt .. kt+TPS-1 complete: at most the later in-flight tiles may remain outstanding
    switch (min(nkt - TPS - kt, SK_STAGES - 2 * TPS)) {
      case 6: sk_wait<6 * PER>(); break;
      case 5: sk_wait<5 * PER>(); break;
      case 4: sk_wait<4 * PER>(); break;
      case 3: sk_wait<3 * PER>(); break;
      case 2: sk_wait<2 * PER>(); break;
      case 1: sk_wait<PER>(); break;
      default: sk_wait<0>(); break;
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int u = 0; u < TPS; ++u)                                  // overwrites the stages read in the previous step
      if (kt + SK_STAGES - TPS + u < nkt) issue(kt + SK_STAGES - TPS + u);
#pragma unroll
    for (int u = 0; u < TPS; ++u) {
      const char* img = smem + ((kt + u) % SK_STAGES) * STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int kc = ks * 4 + g;
        const bf16x8_t wf = sk_frag(img, SK_BM + nt * 16 + l15, kc);
        const bf16x8_t af = sk_frag(img, mt * 16 + l15, kc);
        if (LNF && !SPLIT) sk_stats(af, ln1, ln2);
        if (SPLIT) {
          const char* lo = img + SK_PLANE;
          const bf16x8_t wl = sk_frag(lo, SK_BM + nt * 16 + l15, kc);
          const bf16x8_t al = sk_frag(lo, mt * 16 + l15, kc);
          // accurate mode (round 6): statistics of x = hi + lo.  The two waves that share these 16 rows (nt = 0 / 1) take one k-half of
          // every K-tile each and exchange their sums after the loop (20 v_dot2 per fragment pair: 3-4 us per launch when every wave did all)
          if (LNF && ks == nt) sf_lnf_stats_split(af, al, ln1, ln2);
          acc = sk_mfma(wl, af, acc);
          acc = sk_mfma(wf, al, acc);
        }
        acc = sk_mfma(wf, af, acc);
      }
    }
  }
  if (LNF) {       // the four k-groups of row l15 (all 64 lanes still active here)
    ln1 += __shfl_xor(ln1, 16, 64); ln1 += __shfl_xor(ln1, 32, 64);
    ln2 += __shfl_xor(ln2, 16, 64); ln2 += __shfl_xor(ln2, 32, 64);
    if (SPLIT) {   // the partner wave's k-half (same rows, other nt): through the now idle ring; a + b on both sides, bit-identical
      __syncthreads();
      float* xs = reinterpret_cast<float*>(smem);
      if (g == 0) { xs[(wave * 16 + l15) * 2] = ln1; xs[(wave * 16 + l15) * 2 + 1] = ln2; }
      __syncthreads();
      const int pw = wave ^ 2;
      const float o1 = xs[(pw * 16 + l15) * 2], o2 = xs[(pw * 16 + l15) * 2 + 1];
      ln1 = nt ? o1 + ln1 : ln1 + o1;          // same operand order in both waves
      ln2 = nt ? o2 + ln2 : ln2 + o2;
    }
  }

  // ---- epilogue: lane holds C[m = tile row l15][n = tile col 4g .. 4g+3] ----------------------------------
  const int n = n0 + nt * 16 + g * 4;
  const int m = m0 + mt * 16 + l15;
  if (n >= p.N || m >= p.M) return;
  const f32x4_t bias = pre_bias;
  {
    size_t orow = (size_t)m;
    if (p.grp_rows > 0) orow = sf_out_row(p, m);
    if (LNF) {
      float mean, rstd;
      sk_ln_finish(ln1, ln2, p.K, p.ln_eps, mean, rstd);
      acc = rstd * (acc - mean * pre_lns);
    }
    f32x4_t v = acc + bias;
    const size_t o = orow * (size_t)p.ldc + n;
    if (EPI == SF_EPI_F32) {
      *reinterpret_cast<f32x4_t*>(p.out_f32 + o) = v;
    } else if (EPI == SF_EPI_RESID_F32) {
      const f32x4_t r = p.grp_rows <= 0 ? pre_res : *reinterpret_cast<const f32x4_t*>(p.resid + o);
      v = r + p.alpha * v;
      *reinterpret_cast<f32x4_t*>(p.out_f32 + o) = v;
      if (p.out_hi) sk_fold_copy<SPLIT>(p, o, v);      // small-M LayerNorm fold producer: bf16 copy (accurate mode: hi + lo planes) of the new residual rows
    } else if (EPI == SF_EPI_EMBED_F32) {
      const int pn = m % p.Np, tt = (m / p.Np) % p.Tn + (p.time_base_dev ? *p.time_base_dev : 0);
      const f32x4_t pe = *reinterpret_cast<const f32x4_t*>(p.pos + (size_t)pn * p.N + n);
      const f32x4_t te = *reinterpret_cast<const f32x4_t*>(p.time_rows + (size_t)tt * p.N + n);
      v = v + pe + te;
      *reinterpret_cast<f32x4_t*>(p.out_f32 + o) = v;
      if (p.out_hi) sk_fold_copy<SPLIT>(p, o, v);      // small-M LayerNorm fold: the embedded rows for layer 0's folded qkv
    } else {
      if (EPI == SF_EPI_ACT_BF16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = SPLIT ? apply_act(v[j], p.act) : apply_act_bf16(v[j], p.act);
      }
      unsigned int h[4], l[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split_bf(v[j], h[j], l[j]);
      *reinterpret_cast<u32x2_t*>(p.out_hi + o) = (u32x2_t){h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
      if (p.out_lo) *reinterpret_cast<u32x2_t*>(p.out_lo + o) = (u32x2_t){l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
    }
  }
}

// The same kernel with NB 16-column MFMA tiles per wave (output tile 32 x 32 NB; NB = 2 / 3), for the bf16 LayerNorm-folded consumers of
// ONE streamed frame.  With 32 x 32 tiles the MLP up-projection is 672 workgroups = three on the critical CUs, and what bounds the launch
// is the L2 -> LDS ingest of a CU (~50 GB/s: 3 x 98 KB); a wider tile re-uses the A fragment for NB MFMAs and moves (32 + 32 NB) rows
// per 32 x 32 NB outputs: 7 x 32 = 224 workgroups of 32 x 96, one per CU with 196 KB each.  (Kept apart from the kernel above: folded
// into it as a template parameter, hipcc 7.2 put an s_waitcnt vmcnt(0) behind the epilogue-operand prefetch of the NB = 1 instances —
// a whole memory latency in the prologue, 6.4 -> 6.75 us on the streamed qkv projection.)
template <bool SPLIT, int EPI, bool LNF, int TPS, int NB>
__global__ __launch_bounds__(SK_THREADS) void sf_gemm_skinny_wide_kernel(SfGemmArgs p) {
  constexpr int ROSK_STAGES = SK_BM + 32 * NB;                  // A rows then W rows in one stage image
  constexpr int PLANE = ROSK_STAGES * SK_BK * 2;
  constexpr int STAGE = PLANE * (SPLIT ? 2 : 1);         // hi plane (+ lo plane)
  constexpr int LOADS = ROSK_STAGES * 8 / SK_THREADS;           // 16-byte chunks per thread per plane = 2 (NB = 1), 3, 4
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * (32 * NB), m0 = blockIdx.y * SK_BM;

  // per-thread DMA sources: chunk c of the stage image = (row, 16-byte slot); rows 0..31 = A, the rest = W
  const bf16_t* src_hi[LOADS];
  const bf16_t* src_lo[LOADS];
#pragma unroll
  for (int i = 0; i < LOADS; ++i) {
    const int c = i * SK_THREADS + tid;
    const int row = c >> 3, slot = c & 7;
    const int kc = slot ^ ((row >> 1) & 7);
    if (row < SK_BM) {
      int gr = m0 + row;
      gr = gr < p.M ? gr : p.M - 1;
      src_hi[i] = p.a_hi + (size_t)gr * p.K + kc * 8;
      src_lo[i] = SPLIT ? p.a_lo + (size_t)gr * p.K + kc * 8 : nullptr;
    } else {
      int gr = n0 + row - SK_BM;
      gr = gr < p.N ? gr : p.N - 1;
      src_hi[i] = p.w_hi + (size_t)gr * p.K + kc * 8;
      src_lo[i] = SPLIT ? p.w_lo + (size_t)gr * p.K + kc * 8 : nullptr;
    }
  }
  const bool w_nt = p.w_nt != 0;
  auto issue = [&](int kt) {
    char* dst = smem + (kt % SK_STAGES) * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      if (i >= 1 && w_nt) __builtin_amdgcn_global_load_lds((gptr_t)(src_hi[i] + kt * SK_BK), (lptr_t)(dst + i * 4096), 16, 0, 2);   // W rows, nt
      else __builtin_amdgcn_global_load_lds((gptr_t)(src_hi[i] + kt * SK_BK), (lptr_t)(dst + i * 4096), 16, 0, 0);
      if (SPLIT) __builtin_amdgcn_global_load_lds((gptr_t)(src_lo[i] + kt * SK_BK), (lptr_t)(dst + PLANE + i * 4096), 16, 0, 0);
    }
  };

  f32x4_t acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float ln1 = 0.f, ln2 = 0.f;
  const int nkt = p.K / SK_BK;
  constexpr int PER = LOADS * (SPLIT ? 2 : 1);            // load instructions per stage per thread
  const int mt = wave & 1, nt = wave >> 1;                // this wave's 16 x 16 NB output tile

  for (int s = 0; s < SK_STAGES - TPS && s < nkt; ++s) issue(s);
  // epilogue operands (bias, folded-LayerNorm column sums, residual row) requested NOW, behind the first DMA stages: at one
  // frame the launch is a chain of memory latencies, and each of these loads used to add its own at the very end
  const int m_e = m0 + mt * 16 + l15;
  f32x4_t pre_bias[NB], pre_lns[NB], pre_res = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n_e = n0 + (nt * NB + j) * 16 + g * 4;
    // (locals, then one assignment per array element: with the conditional loads written straight into the arrays hipcc 7.2 put an
    //  s_waitcnt vmcnt(0) behind the first of them — a whole memory latency in the prologue of every launch)
    f32x4_t b4 = {0.f, 0.f, 0.f, 0.f}, l4 = {0.f, 0.f, 0.f, 0.f};
    if (n_e < p.N && m_e < p.M) {
      if (p.bias) b4 = *reinterpret_cast<const f32x4_t*>(p.bias + n_e);
      if (LNF) l4 = *reinterpret_cast<const f32x4_t*>(p.ln_s + n_e);
      if (NB == 1 && EPI == SF_EPI_RESID_F32 && p.grp_rows <= 0) pre_res = *reinterpret_cast<const f32x4_t*>(p.resid + (size_t)m_e * p.ldc + n_e);
    }
    pre_bias[j] = b4;
    pre_lns[j] = l4;
  }
  for (int kt = 0; kt < nkt; kt += TPS) {
    // tiles kt .. kt+TPS-1 complete: at most the later in-flight tiles may remain outstanding
    switch (min(nkt - TPS - kt, SK_STAGES - 2 * TPS)) {
      case 6: sk_wait<6 * PER>(); break;
      case 5: sk_wait<5 * PER>(); break;
      case 4: sk_wait<4 * PER>(); break;
      case 3: sk_wait<3 * PER>(); break;
      case 2: sk_wait<2 * PER>(); break;
      case 1: sk_wait<PER>(); break;
      default: sk_wait<0>(); break;
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int u = 0; u < TPS; ++u)                                  // overwrites the stages read in the previous step
      if (kt + SK_STAGES - TPS + u < nkt) issue(kt + SK_STAGES - TPS + u);
#pragma unroll
    for (int u = 0; u < TPS; ++u) {
      const char* img = smem + ((kt + u) % SK_STAGES) * STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int kc = ks * 4 + g;
        bf16x8_t wf[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) wf[j] = sk_frag(img, SK_BM + (nt * NB + j) * 16 + l15, kc);      // every fragment read issued before the first use
        const bf16x8_t af = sk_frag(img, mt * 16 + l15, kc);
        if (LNF) sk_stats(af, ln1, ln2);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          if (SPLIT) {
            const char* lo = img + PLANE;
            const bf16x8_t wl = sk_frag(lo, SK_BM + (nt * NB + j) * 16 + l15, kc);
            const bf16x8_t al = sk_frag(lo, mt * 16 + l15, kc);
            acc[j] = sk_mfma(wl, af, acc[j]);
            acc[j] = sk_mfma(wf[j], al, acc[j]);
          }
          acc[j] = sk_mfma(wf[j], af, acc[j]);
        }
      }
    }
  }
  if (LNF) {       // the four k-groups of row l15 (all 64 lanes still active here)
    ln1 += __shfl_xor(ln1, 16, 64); ln1 += __shfl_xor(ln1, 32, 64);
    ln2 += __shfl_xor(ln2, 16, 64); ln2 += __shfl_xor(ln2, 32, 64);
  }

  // ---- epilogue: lane holds C[m = tile row l15][n = tile col 4g .. 4g+3] of each of its NB tiles --------------------
  const int m = m0 + mt * 16 + l15;
  if (m >= p.M) return;
  size_t orow = (size_t)m;
  if (p.grp_rows > 0) orow = sf_out_row(p, m);
  float mean = 0.f, rstd = 1.f;
  if (LNF) sk_ln_finish(ln1, ln2, p.K, p.ln_eps, mean, rstd);
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = n0 + (nt * NB + j) * 16 + g * 4;
    if (n >= p.N) continue;
    f32x4_t a4 = acc[j];
    if (LNF) a4 = rstd * (a4 - mean * pre_lns[j]);
    f32x4_t v = a4 + pre_bias[j];
    const size_t o = orow * (size_t)p.ldc + n;
    if (EPI == SF_EPI_F32) {
      *reinterpret_cast<f32x4_t*>(p.out_f32 + o) = v;
    } else if (EPI == SF_EPI_RESID_F32) {
      const f32x4_t r = (NB == 1 && p.grp_rows <= 0) ? pre_res : *reinterpret_cast<const f32x4_t*>(p.resid + o);
      v = r + p.alpha * v;
      *reinterpret_cast<f32x4_t*>(p.out_f32 + o) = v;
      if (!SPLIT && p.out_hi)    // small-M LayerNorm fold producer: bf16 copy of the new residual rows
        *reinterpret_cast<u32x2_t*>(p.out_hi + o) = (u32x2_t){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    } else if (EPI == SF_EPI_EMBED_F32) {
      const int pn = m % p.Np, tt = (m / p.Np) % p.Tn + (p.time_base_dev ? *p.time_base_dev : 0);
      const f32x4_t pe = *reinterpret_cast<const f32x4_t*>(p.pos + (size_t)pn * p.N + n);
      const f32x4_t te = *reinterpret_cast<const f32x4_t*>(p.time_rows + (size_t)tt * p.N + n);
      v = v + pe + te;
      *reinterpret_cast<f32x4_t*>(p.out_f32 + o) = v;
      if (!SPLIT && p.out_hi)      // small-M LayerNorm fold: bf16 copy of the embedded rows for layer 0's folded qkv
        *reinterpret_cast<u32x2_t*>(p.out_hi + o) = (u32x2_t){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    } else {
      if (EPI == SF_EPI_ACT_BF16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = SPLIT ? apply_act(v[e], p.act) : apply_act_bf16(v[e], p.act);
      }
      unsigned int h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split_bf(v[e], h[e], l[e]);
      *reinterpret_cast<u32x2_t*>(p.out_hi + o) = (u32x2_t){h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
      if (p.out_lo) *reinterpret_cast<u32x2_t*>(p.out_lo + o) = (u32x2_t){l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K-parallel variant for the N = 768 projections of the streaming step (168 workgroups at one frame: every
// workgroup has a CU to itself, and what bounds it is the number of SEQUENTIAL K-tiles — 12 at K = 768,
// 48 at K = 3072 for the MLP down-projection).  KG groups of 4 waves split the K range, each with its own
// 4-stage ring; the groups iterate in lock-step on one barrier per step (a quarter / half as many steps),
// then groups 1.. hand their partial 16x16 tiles to group 0 through LDS (fixed order: deterministic) and
// group 0 runs the usual fused epilogue.  KG = 4 (bf16, 1024 threads), KG = 2 (bf16x3, 512 threads).
// ------------------------------------------------------------------------------------------------
#define SKG_STAGES 4

// PG > 0: K-tiles per group as a compile-time constant (every group owns exactly PG <= SKG_STAGES tiles: K = 64 KG PG) — the issue and
// compute loops of the one-step path unroll into straight-line code (all fragment reads of the slice in flight before the first MFMA)
// instead of one [ds_read -> wait -> MFMA] block per tile behind a run-time trip count (round 4, same finding as the attention kernels)
template <bool SPLIT, int EPI, int KG, bool LNF = false, int PG = 0>
__global__ __launch_bounds__(SK_THREADS * KG) void sf_gemm_skinny_kg_kernel(SfGemmArgs p) {
  constexpr int STAGE = SK_PLANE * (SPLIT ? 2 : 1);
  constexpr int RING = SKG_STAGES * STAGE;
  constexpr int LOADS = SK_ROWS * 8 / SK_THREADS;       // per thread of a group = 2
  constexpr int PER = LOADS * (SPLIT ? 2 : 1);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kgrp = wave >> 2, w4 = wave & 3, tid_g = tid & (SK_THREADS - 1);
  const int l15 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * SK_BN, m0 = blockIdx.y * SK_BM;
  const int nkt = p.K / SK_BK;
  const int per_grp = PG ? PG : (nkt + KG - 1) / KG;    // K-tiles of a group (the last group may own fewer)
  const int kt_base = kgrp * per_grp;
  const int mine = PG ? PG : max(0, min(per_grp, nkt - kt_base));

  const bf16_t* src_hi[LOADS];
  const bf16_t* src_lo[LOADS];
#pragma unroll
  for (int i = 0; i < LOADS; ++i) {
    const int c = i * SK_THREADS + tid_g;
    const int row = c >> 3, slot = c & 7;
    const int kc = slot ^ ((row >> 1) & 7);
    if (row < SK_BM) {
      int gr = m0 + row;
      gr = gr < p.M ? gr : p.M - 1;
      src_hi[i] = p.a_hi + (size_t)gr * p.K + kc * 8;
      src_lo[i] = SPLIT ? p.a_lo + (size_t)gr * p.K + kc * 8 : nullptr;
    } else {
      int gr = n0 + row - SK_BM;
      gr = gr < p.N ? gr : p.N - 1;
      src_hi[i] = p.w_hi + (size_t)gr * p.K + kc * 8;
      src_lo[i] = SPLIT ? p.w_lo + (size_t)gr * p.K + kc * 8 : nullptr;
    }
  }
  char* ring = smem + kgrp * RING;
  const bool w_nt = p.w_nt != 0;
  auto issue = [&](int j) {                              // j = K-tile index inside the group's range
    char* dst = ring + (j % SKG_STAGES) * STAGE + w4 * 1024;
    const int kt = kt_base + j;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      if (i == 1 && w_nt) __builtin_amdgcn_global_load_lds((gptr_t)(src_hi[i] + kt * SK_BK), (lptr_t)(dst + i * 4096), 16, 0, 2);   // W rows, nt
      else __builtin_amdgcn_global_load_lds((gptr_t)(src_hi[i] + kt * SK_BK), (lptr_t)(dst + i * 4096), 16, 0, 0);
      if (SPLIT) __builtin_amdgcn_global_load_lds((gptr_t)(src_lo[i] + kt * SK_BK), (lptr_t)(dst + SK_PLANE + i * 4096), 16, 0, 0);
    }
  };

  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  float ln1 = 0.f, ln2 = 0.f;
  const int mt = w4 & 1, nt = w4 >> 1;
  // epilogue operands of group 0 (the group that finishes the tile), requested before the K loop: see sf_gemm_skinny_kernel
  const int n_e = n0 + nt * 16 + g * 4, m_e = m0 + mt * 16 + l15;
  const bool live_e = kgrp == 0 && n_e < p.N && m_e < p.M;
  f32x4_t pre_bias = {0.f, 0.f, 0.f, 0.f}, pre_lns = {0.f, 0.f, 0.f, 0.f}, pre_res = {0.f, 0.f, 0.f, 0.f};
  auto prefetch_epilogue = [&]() {
    if (live_e) {
      if (p.bias) pre_bias = *reinterpret_cast<const f32x4_t*>(p.bias + n_e);
      if (LNF) pre_lns = *reinterpret_cast<const f32x4_t*>(p.ln_s + n_e);
      if (EPI == SF_EPI_RESID_F32 && p.grp_rows <= 0) pre_res = *reinterpret_cast<const f32x4_t*>(p.resid + (size_t)m_e * p.ldc + n_e);
    }
  };
  auto compute = [&](int j) {
    const char* img = ring + (j % SKG_STAGES) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kc = ks * 4 + g;
      const bf16x8_t wf = sk_frag(img, SK_BM + nt * 16 + l15, kc);
      const bf16x8_t af = sk_frag(img, mt * 16 + l15, kc);
      if (LNF) sk_stats(af, ln1, ln2);
      if (SPLIT) {
        const char* lo = img + SK_PLANE;
        const bf16x8_t wl = sk_frag(lo, SK_BM + nt * 16 + l15, kc);
        const bf16x8_t al = sk_frag(lo, mt * 16 + l15, kc);
        acc = sk_mfma(wl, af, acc);
        acc = sk_mfma(wf, al, acc);
      }
      acc = sk_mfma(wf, af, acc);
    }
  };
  // every group runs the same number of steps (one barrier domain); a step is a serial wait -> barrier -> LDS read -> MFMA
  // chain of a few hundred cycles.  (Two tiles per step on the 4-stage ring measured slower at K = 3072: it gives up the
  // prefetch distance.)
  if (PG) {
#pragma unroll
    for (int j = 0; j < PG; ++j) issue(j);
    prefetch_epilogue();
    sk_wait<0>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < PG; ++j) compute(j);
  } else if (per_grp <= SKG_STAGES) {                    // the whole slice fits the ring (K = 768: 3 tiles): ONE step
    for (int j = 0; j < mine; ++j) issue(j);
    prefetch_epilogue();
    sk_wait<0>();
    __builtin_amdgcn_s_barrier();
    for (int j = 0; j < mine; ++j) compute(j);
  } else {
    for (int j = 0; j < SKG_STAGES - 1 && j < mine; ++j) issue(j);
    for (int j = 0; j < per_grp; ++j) {
      if (j == per_grp - 2 || per_grp == 1) prefetch_epilogue();      // late enough not to count against the vmcnt of the ring (vmcnt(0) tail)
      const int later = min(mine - 1 - j, SKG_STAGES - 2);
      if (later >= 2) sk_wait<2 * PER>(); else if (later == 1) sk_wait<PER>(); else sk_wait<0>();
      __builtin_amdgcn_s_barrier();
      if (j + SKG_STAGES - 1 < mine) issue(j + SKG_STAGES - 1);
      if (j < mine) compute(j);
    }
  }
  // ---- cross-group reduction through the (now idle) ring of group 0 ------------------------------------
  if (LNF) {
    ln1 += __shfl_xor(ln1, 16, 64); ln1 += __shfl_xor(ln1, 32, 64);
    ln2 += __shfl_xor(ln2, 16, 64); ln2 += __shfl_xor(ln2, 32, 64);
  }
  __builtin_amdgcn_s_barrier();
  f32x4_t* red = reinterpret_cast<f32x4_t*>(smem);
  float* reds = reinterpret_cast<float*>(smem + (KG - 1) * 4 * 64 * 16);       // [KG-1][4 waves][2][16 rows]
  if (kgrp > 0) {
    red[((kgrp - 1) * 4 + w4) * 64 + lane] = acc;
    if (LNF && g == 0) {
      reds[(((kgrp - 1) * 4 + w4) * 2 + 0) * 16 + l15] = ln1;
      reds[(((kgrp - 1) * 4 + w4) * 2 + 1) * 16 + l15] = ln2;
    }
  }
  __syncthreads();
  if (kgrp > 0) return;
#pragma unroll
  for (int k = 1; k < KG; ++k) {
    acc += red[((k - 1) * 4 + w4) * 64 + lane];
    if (LNF) {
      ln1 += reds[(((k - 1) * 4 + w4) * 2 + 0) * 16 + l15];
      ln2 += reds[(((k - 1) * 4 + w4) * 2 + 1) * 16 + l15];
    }
  }

  const int n = n0 + nt * 16 + g * 4;
  const int m = m0 + mt * 16 + l15;
  if (n >= p.N || m >= p.M) return;
  const f32x4_t bias = pre_bias;
  size_t orow = (size_t)m;
  if (p.grp_rows > 0) orow = sf_out_row(p, m);
  if (LNF) {
    float mean, rstd;
    sk_ln_finish(ln1, ln2, p.K, p.ln_eps, mean, rstd);
    acc = rstd * (acc - mean * pre_lns);
  }
  f32x4_t v = acc + bias;
  const size_t o = orow * (size_t)p.ldc + n;
  if (EPI == SF_EPI_F32) {
    *reinterpret_cast<f32x4_t*>(p.out_f32 + o) = v;
  } else if (EPI == SF_EPI_RESID_F32) {
    const f32x4_t r = p.grp_rows <= 0 ? pre_res : *reinterpret_cast<const f32x4_t*>(p.resid + o);
    v = r + p.alpha * v;
    *reinterpret_cast<f32x4_t*>(p.out_f32 + o) = v;
    if (p.out_hi) sk_fold_copy<SPLIT>(p, o, v);        // small-M LayerNorm fold producer
  } else {
    if (EPI == SF_EPI_ACT_BF16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = SPLIT ? apply_act(v[j], p.act) : apply_act_bf16(v[j], p.act);
    }
    unsigned int h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_bf(v[j], h[j], l[j]);
    *reinterpret_cast<u32x2_t*>(p.out_hi + o) = (u32x2_t){h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
    if (p.out_lo) *reinterpret_cast<u32x2_t*>(p.out_lo + o) = (u32x2_t){l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
  }
}


// ------------------------------------------------------------------------------------------------
// Mid-size variant (512 < M <= sf_skinny_max_rows(): several streams advancing one frame per call, bf16 mode).
// At M = 1568 the [32 x 32] tiles above move 346 MB from L2 to LDS for the qkv projection (every workgroup streams
// 2 x 32 x K operands for 32 x 32 outputs: 16 FLOP per byte) and the launch is bound by that traffic.  Here a workgroup
// owns [64 x 64], a wave a [32 x 32] quarter as 2 x 2 MFMA tiles: half the L2 -> LDS bytes and half the ds_reads per MFMA;
// 4-stage ring of 16 KB stages (two workgroups per CU), one barrier per K-tile, same fragment layout, same in-kernel
// LayerNorm statistics and epilogues.
// ------------------------------------------------------------------------------------------------
#define MD_B 64
#define MD_STAGES 4
#define MD_STAGE (2 * MD_B * SK_BK * 2)       // 16 KB: 64 A rows then 64 W rows

template <int EPI, bool LNF>
__global__ __launch_bounds__(SK_THREADS) void sf_gemm_mid_kernel(SfGemmArgs p) {
  constexpr int LOADS = 2 * MD_B * 8 / SK_THREADS;      // 4 chunks of 16 bytes per thread and stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * MD_B, m0 = blockIdx.y * MD_B;
  const bf16_t* src[LOADS];
#pragma unroll
  for (int i = 0; i < LOADS; ++i) {
    const int c = i * SK_THREADS + tid;
    const int row = c >> 3, slot = c & 7;
    const int kc = slot ^ ((row >> 1) & 7);
    if (row < MD_B) {
      int gr = m0 + row;
      gr = gr < p.M ? gr : p.M - 1;
      src[i] = p.a_hi + (size_t)gr * p.K + kc * 8;
    } else {
      int gr = n0 + row - MD_B;
      gr = gr < p.N ? gr : p.N - 1;
      src[i] = p.w_hi + (size_t)gr * p.K + kc * 8;
    }
  }
  auto issue = [&](int kt) {
    char* dst = smem + (kt % MD_STAGES) * MD_STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < LOADS; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + kt * SK_BK), (lptr_t)(dst + i * 4096), 16, 0, 0);
  };
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float ln1[2] = {0.f, 0.f}, ln2[2] = {0.f, 0.f};
  const int nkt = p.K / SK_BK;
  const int wm = wave & 1, wn = wave >> 1;
  for (int s = 0; s < MD_STAGES - 1 && s < nkt; ++s) issue(s);
  for (int kt = 0; kt < nkt; ++kt) {
    const int later = min(nkt - 1 - kt, MD_STAGES - 2);       // tiles that may stay in flight behind tile kt
    if (later >= 2) sk_wait<2 * LOADS>(); else if (later == 1) sk_wait<LOADS>(); else sk_wait<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + MD_STAGES - 1 < nkt) issue(kt + MD_STAGES - 1);   // into the stage every wave finished reading before this barrier
    const char* img = smem + (kt % MD_STAGES) * MD_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kc = ks * 4 + g;
      bf16x8_t af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = sk_frag(img, wm * 32 + i * 16 + l15, kc);
        wf[i] = sk_frag(img, MD_B + wn * 32 + i * 16 + l15, kc);
        if (LNF) sk_stats(af[i], ln1[i], ln2[i]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = sk_mfma(wf[j], af[i], acc[i][j]);
    }
  }
  if (LNF) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ln1[i] += __shfl_xor(ln1[i], 16, 64); ln1[i] += __shfl_xor(ln1[i], 32, 64);
      ln2[i] += __shfl_xor(ln2[i], 16, 64); ln2[i] += __shfl_xor(ln2[i], 32, 64);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 32 + i * 16 + l15;
    if (m >= p.M) continue;
    size_t orow = (size_t)m;
    if (p.grp_rows > 0) orow = sf_out_row(p, m);
    float mean = 0.f, rstd = 1.f;
    if (LNF) sk_ln_finish(ln1[i], ln2[i], p.K, p.ln_eps, mean, rstd);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 32 + j * 16 + g * 4;
      if (n >= p.N) continue;
      f32x4_t v = acc[i][j];
      if (LNF) v = rstd * (v - mean * *reinterpret_cast<const f32x4_t*>(p.ln_s + n));
      if (p.bias) v += *reinterpret_cast<const f32x4_t*>(p.bias + n);
      const size_t o = orow * (size_t)p.ldc + n;
      if (EPI == SF_EPI_F32) {
        *reinterpret_cast<f32x4_t*>(p.out_f32 + o) = v;
      } else if (EPI == SF_EPI_RESID_F32) {
        v = *reinterpret_cast<const f32x4_t*>(p.resid + o) + p.alpha * v;
        *reinterpret_cast<f32x4_t*>(p.out_f32 + o) = v;
        if (p.out_hi) *reinterpret_cast<u32x2_t*>(p.out_hi + o) = (u32x2_t){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
      } else {
        if (EPI == SF_EPI_ACT_BF16) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = apply_act_bf16(v[q], p.act);
        }
        *reinterpret_cast<u32x2_t*>(p.out_hi + o) = (u32x2_t){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
      }
    }
  }
}

static bool mid_ok(const SfGemmArgs& a, bool split) {
  const bool off = sf_sw(SW_DISABLE_GEMM_MID) != nullptr;
  const int min_m = sf_sw(SW_GEMM_MID_MIN_M) ? atoi(sf_sw(SW_GEMM_MID_MIN_M)) : 512;
  return !off && !split && a.M > min_m && a.N >= 64 && a.epi != SF_EPI_EMBED_F32 && !a.out_lo;
}
static hipError_t mid_launch(const SfGemmArgs& a, hipStream_t s) {
  const dim3 grid((a.N + MD_B - 1) / MD_B, (a.M + MD_B - 1) / MD_B);
  const size_t lds = (size_t)MD_STAGES * MD_STAGE;
  static SfPerDeviceOnce attr_set;
  if (attr_set.first()) {
#define MD_ATTR(E, L) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_mid_kernel<E, L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    MD_ATTR(SF_EPI_F32, false) MD_ATTR(SF_EPI_BF16, false) MD_ATTR(SF_EPI_ACT_BF16, false) MD_ATTR(SF_EPI_RESID_F32, false)
    MD_ATTR(SF_EPI_BF16, true) MD_ATTR(SF_EPI_ACT_BF16, true)
#undef MD_ATTR
  }
#define MD_GO(E, L) hipLaunchKernelGGL((sf_gemm_mid_kernel<E, L>), grid, dim3(SK_THREADS), lds, s, a)
  if (a.ln_inkernel) {
    if (a.epi == SF_EPI_BF16) MD_GO(SF_EPI_BF16, true);
    else if (a.epi == SF_EPI_ACT_BF16) MD_GO(SF_EPI_ACT_BF16, true);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  switch (a.epi) {
    case SF_EPI_F32: MD_GO(SF_EPI_F32, false); break;
    case SF_EPI_BF16: MD_GO(SF_EPI_BF16, false); break;
    case SF_EPI_ACT_BF16: MD_GO(SF_EPI_ACT_BF16, false); break;
    case SF_EPI_RESID_F32: MD_GO(SF_EPI_RESID_F32, false); break;
    default: return hipErrorInvalidValue;
  }
#undef MD_GO
  return hipGetLastError();
}

template <bool SPLIT, int KG>
static hipError_t skg_launch(const SfGemmArgs& a, dim3 grid, hipStream_t s) {
  const size_t lds = (size_t)KG * SKG_STAGES * SK_PLANE * (SPLIT ? 2 : 1);
  if (a.ln_inkernel) {       // LayerNorm-folded consumers (bf16 mode): qkv (BF16) and MLP-up (ACT_BF16) epilogues
    if (SPLIT) return hipErrorInvalidValue;
    static SfPerDeviceOnce attr_ln;
    if (attr_ln.first()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_skinny_kg_kernel<false, SF_EPI_BF16, KG, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_skinny_kg_kernel<false, SF_EPI_ACT_BF16, KG, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (a.epi == SF_EPI_BF16) hipLaunchKernelGGL((sf_gemm_skinny_kg_kernel<false, SF_EPI_BF16, KG, true>), grid, dim3(SK_THREADS * KG), lds, s, a);
    else if (a.epi == SF_EPI_ACT_BF16) hipLaunchKernelGGL((sf_gemm_skinny_kg_kernel<false, SF_EPI_ACT_BF16, KG, true>), grid, dim3(SK_THREADS * KG), lds, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  static SfPerDeviceOnce attr_set;
  if (attr_set.first()) {
#define SKG_ATTR(E) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_skinny_kg_kernel<SPLIT, E, KG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SKG_ATTR(SF_EPI_F32) SKG_ATTR(SF_EPI_BF16) SKG_ATTR(SF_EPI_ACT_BF16) SKG_ATTR(SF_EPI_RESID_F32)
#undef SKG_ATTR
  }
  if constexpr (!SPLIT && KG == 4) {        // the streamed K = 768 residual projections (24 of a frame's 108 launches): three K-tiles per group, unrolled
    const bool pg_off = sf_sw(SW_DISABLE_SKG_UNROLL) != nullptr;      // A/B switch
    if (!pg_off && a.epi == SF_EPI_RESID_F32 && a.K == 64 * KG * 3) {
      static SfPerDeviceOnce attr_pg;
      if (attr_pg.first())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_skinny_kg_kernel<false, SF_EPI_RESID_F32, 4, false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((sf_gemm_skinny_kg_kernel<false, SF_EPI_RESID_F32, 4, false, 3>), grid, dim3(SK_THREADS * KG), lds, s, a);
      return hipGetLastError();
    }
  }
#define SKG_CASE(E)                                                                                              \
  case E:                                                                                                        \
    hipLaunchKernelGGL((sf_gemm_skinny_kg_kernel<SPLIT, E, KG>), grid, dim3(SK_THREADS * KG), lds, s, a);        \
    break;
  switch (a.epi) {
    SKG_CASE(SF_EPI_F32)
    SKG_CASE(SF_EPI_BF16)
    SKG_CASE(SF_EPI_ACT_BF16)
    SKG_CASE(SF_EPI_RESID_F32)
    default:
      return hipErrorInvalidValue;
  }
#undef SKG_CASE
  return hipGetLastError();
}

int sf_skinny_max_rows() {
  static int m = 0;
  if (!m) {
    m = 2560;      // several streams per call (M = 196 per stream): 4 streams 2.51 -> 1.58 ms, 8 streams 2.79 -> 2.43 ms per step against
                   // the 128^2 kernel; from 16 streams (M = 3136) on the large-tile kernels win
    if (const char* e = sf_sw(SW_SKINNY_MAX_M)) m = atoi(e) > 0 ? atoi(e) : 2560;
  }
  return m;
}

bool sf_gemm_skinny_supported(const SfGemmArgs& a, bool split) {
  // few rows, or few output columns (the rank-32 LoRA projections of the training step: A streams once)
  const int max_rows = split ? (sf_skinny_max_rows() < 1024 ? sf_skinny_max_rows() : 1024) : sf_skinny_max_rows();   // bf16x3: the 128^2 kernel wins from 8 streams on
  if (a.M <= 0 || (a.M > max_rows && a.N > 64) || a.K < SK_BK || (a.K % SK_BK) || (a.N % 4) || (a.ldc % 4)) return false;
  if (a.ln_stats || a.ln_stats_out) return false;                  // the statistics-buffer fold: panel / 256^2 kernels
  if (a.ln_inkernel && (!a.ln_s || (a.epi != SF_EPI_BF16 && a.epi != SF_EPI_ACT_BF16 && !(split && a.epi == SF_EPI_F32)))) return false;
  if (a.ln_inkernel && split && ((a.N + SK_BN - 1) / SK_BN) * ((a.M + SK_BM - 1) / SK_BM) <= 256) return false;      // accurate fold: the 32 x 32 kernel only (no K-parallel instance)
  if (split && (!a.a_lo || !a.w_lo)) return false;
  return true;
}

template <bool SPLIT, int TPS>
static hipError_t sk_launch_epi(const SfGemmArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static SfPerDeviceOnce attr_set;
  if (attr_set.first()) {
#define SK_ATTR(E) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_skinny_kernel<SPLIT, E, false, TPS>), hipFuncAttributeMaxDynamicSharedMemorySize, SK_STAGES * SK_PLANE * 2);
    SK_ATTR(SF_EPI_F32) SK_ATTR(SF_EPI_BF16) SK_ATTR(SF_EPI_ACT_BF16) SK_ATTR(SF_EPI_RESID_F32) SK_ATTR(SF_EPI_EMBED_F32)
#undef SK_ATTR
  }
#define SK_CASE(E)                                                                                           \
  case E:                                                                                                    \
    hipLaunchKernelGGL((sf_gemm_skinny_kernel<SPLIT, E, false, TPS>), grid, dim3(SK_THREADS), lds, s, a);    \
    break;
  switch (a.epi) {
    SK_CASE(SF_EPI_F32)
    SK_CASE(SF_EPI_BF16)
    SK_CASE(SF_EPI_ACT_BF16)
    SK_CASE(SF_EPI_RESID_F32)
    SK_CASE(SF_EPI_EMBED_F32)
    default:
      return hipErrorInvalidValue;
  }
#undef SK_CASE
  return hipGetLastError();
}

hipError_t sf_launch_gemm_skinny(const SfGemmArgs& a_in, bool split, hipStream_t s) {
  if (!sf_gemm_skinny_supported(a_in, split)) return hipErrorInvalidValue;
  // non-temporal policy on the weight stream of a single streamed frame (each 32-row slab is read by the 7 row tiles of one
  // XCD, once): config #5 p50 0.773 -> 0.751 ms; neutral from 8 streams on (M = 1568), where the default policy stays
  const int nt_env = sf_sw(SW_SKINNY_NT) ? atoi(sf_sw(SW_SKINNY_NT)) : -1;
  SfGemmArgs a = a_in;
  a.w_nt = nt_env >= 0 ? nt_env : (a.M <= 512 ? 1 : 0);
  if (mid_ok(a, split)) return mid_launch(a, s);
  const dim3 grid((a.N + SK_BN - 1) / SK_BN, (a.M + SK_BM - 1) / SK_BM);
  // every workgroup gets its own CU and the K loop is long: the K-parallel variant
  if ((int)(grid.x * grid.y) <= 256 && a.K >= 512 && a.epi != SF_EPI_EMBED_F32 && !sf_sw(SW_SKINNY_NO_KG))
    return split ? skg_launch<true, 2>(a, grid, s) : skg_launch<false, 4>(a, grid, s);
  const size_t lds = (size_t)SK_STAGES * SK_PLANE * (split ? 2 : 1);
  if (a.ln_inkernel && split) {      // accurate mode (round 6): statistics of x = hi + lo inside the consumer, fp32 qkv / hi + lo activation outputs
    static SfPerDeviceOnce attr_lns;
    if (attr_lns.first()) {
#define SK_LNSATTR(E, T) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_skinny_kernel<true, E, true, T>), hipFuncAttributeMaxDynamicSharedMemorySize, SK_STAGES * SK_PLANE * 2);
      SK_LNSATTR(SF_EPI_F32, 4) SK_LNSATTR(SF_EPI_F32, 1) SK_LNSATTR(SF_EPI_BF16, 4) SK_LNSATTR(SF_EPI_BF16, 1) SK_LNSATTR(SF_EPI_ACT_BF16, 4) SK_LNSATTR(SF_EPI_ACT_BF16, 1)
#undef SK_LNSATTR
    }
    const bool four = ((a.K / SK_BK) % 4) == 0;
#define SK_LNSGO(E) do { if (four) hipLaunchKernelGGL((sf_gemm_skinny_kernel<true, E, true, 4>), grid, dim3(SK_THREADS), lds, s, a); \
                         else hipLaunchKernelGGL((sf_gemm_skinny_kernel<true, E, true, 1>), grid, dim3(SK_THREADS), lds, s, a); } while (0)
    if (a.epi == SF_EPI_F32) SK_LNSGO(SF_EPI_F32);
    else if (a.epi == SF_EPI_BF16) SK_LNSGO(SF_EPI_BF16);
    else SK_LNSGO(SF_EPI_ACT_BF16);
#undef SK_LNSGO
    return hipGetLastError();
  }
  if (a.ln_inkernel && !split && a.M <= 256 && ((a.K / SK_BK) % 4) == 0 && (int)(grid.x * grid.y) > 512) {
    // one streamed frame, more than two 32 x 32 tiles per CU (MLP-up: 672): the narrowest wider tile that gives every workgroup its
    // own CU (sf_gemm_skinny_wide_kernel) — 32 x 96 for MLP-up (224 workgroups): 10.2 -> 8.5 us per launch, p50 0.733 -> 0.716 ms
    // (profiles/r05_skinny_wide_ab.txt).  qkv (504 tiles) stays on 32 x 32: its 32 x 64 instance measured 6.9 against 6.4 us on 4
    // waves (too few waves to issue a CU's ingest) and level on 16 waves with every K-tile in flight.
    const int nb_env = sf_sw(SW_SKINNY_NB) ? atoi(sf_sw(SW_SKINNY_NB)) : 0;      // A/B: 1 keeps the 32 x 32 tiles, 2 / 3 force a width
    int nb = 1;
    for (int c = 2; c <= 3 && nb == 1; ++c)
      if (a.N % (32 * c) == 0 && (a.N / (32 * c)) * (int)grid.y <= 256) nb = c;
    if (nb_env >= 1 && nb_env <= 3 && a.N % (32 * nb_env) == 0) nb = nb_env;
    if (nb > 1) {
      const size_t ldw = (size_t)SK_STAGES * (SK_BM + 32 * nb) * SK_BK * 2;
      const dim3 gw(a.N / (32 * nb), grid.y);
      static SfPerDeviceOnce attr_w;
      if (attr_w.first()) {
#define SK_WATTR(E, B) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_skinny_wide_kernel<false, E, true, 4, B>), hipFuncAttributeMaxDynamicSharedMemorySize, SK_STAGES * (SK_BM + 32 * B) * SK_BK * 2);
        SK_WATTR(SF_EPI_BF16, 2) SK_WATTR(SF_EPI_BF16, 3) SK_WATTR(SF_EPI_ACT_BF16, 2) SK_WATTR(SF_EPI_ACT_BF16, 3)
#undef SK_WATTR
      }
      if (a.epi == SF_EPI_BF16) {
        if (nb == 2) hipLaunchKernelGGL((sf_gemm_skinny_wide_kernel<false, SF_EPI_BF16, true, 4, 2>), gw, dim3(SK_THREADS), ldw, s, a);
        else hipLaunchKernelGGL((sf_gemm_skinny_wide_kernel<false, SF_EPI_BF16, true, 4, 3>), gw, dim3(SK_THREADS), ldw, s, a);
      } else {
        if (nb == 2) hipLaunchKernelGGL((sf_gemm_skinny_wide_kernel<false, SF_EPI_ACT_BF16, true, 4, 2>), gw, dim3(SK_THREADS), ldw, s, a);
        else hipLaunchKernelGGL((sf_gemm_skinny_wide_kernel<false, SF_EPI_ACT_BF16, true, 4, 3>), gw, dim3(SK_THREADS), ldw, s, a);
      }
      return hipGetLastError();
    }
  }
  if (a.ln_inkernel) {
    static SfPerDeviceOnce attr_ln;
    if (attr_ln.first()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_skinny_kernel<false, SF_EPI_BF16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SK_STAGES * SK_PLANE * 2);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_skinny_kernel<false, SF_EPI_ACT_BF16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SK_STAGES * SK_PLANE * 2);
    }
    const int tps_env = sf_sw(SW_SKINNY_TPS) ? atoi(sf_sw(SW_SKINNY_TPS)) : 0;
    const int nkt = a.K / SK_BK;
    int tps = tps_env ? tps_env : 4;
    while (tps > 1 && (nkt % tps)) --tps;
    static SfPerDeviceOnce attr_ln2;
    if (attr_ln2.first()) {
#define SK_LNATTR(E, T) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_skinny_kernel<false, E, true, T>), hipFuncAttributeMaxDynamicSharedMemorySize, SK_STAGES * SK_PLANE * 2);
      SK_LNATTR(SF_EPI_BF16, 2) SK_LNATTR(SF_EPI_BF16, 3) SK_LNATTR(SF_EPI_BF16, 4)
      SK_LNATTR(SF_EPI_ACT_BF16, 2) SK_LNATTR(SF_EPI_ACT_BF16, 3) SK_LNATTR(SF_EPI_ACT_BF16, 4)
#undef SK_LNATTR
    }
#define SK_LNGO(E)                                                                                                               \
    switch (tps) {                                                                                                               \
      case 4: hipLaunchKernelGGL((sf_gemm_skinny_kernel<false, E, true, 4>), grid, dim3(SK_THREADS), lds, s, a); break;          \
      case 3: hipLaunchKernelGGL((sf_gemm_skinny_kernel<false, E, true, 3>), grid, dim3(SK_THREADS), lds, s, a); break;          \
      case 2: hipLaunchKernelGGL((sf_gemm_skinny_kernel<false, E, true, 2>), grid, dim3(SK_THREADS), lds, s, a); break;          \
      default: hipLaunchKernelGGL((sf_gemm_skinny_kernel<false, E, true>), grid, dim3(SK_THREADS), lds, s, a); break;            \
    }
    if (a.epi == SF_EPI_BF16) { SK_LNGO(SF_EPI_BF16) } else { SK_LNGO(SF_EPI_ACT_BF16) }
#undef SK_LNGO
    return hipGetLastError();
  }
  const bool four = ((a.K / SK_BK) % 4) == 0;         // four K-tiles per barrier when the tile count allows
  if (split) return four ? sk_launch_epi<true, 4>(a, grid, lds, s) : sk_launch_epi<true, 1>(a, grid, lds, s);
  return four ? sk_launch_epi<false, 4>(a, grid, lds, s) : sk_launch_epi<false, 1>(a, grid, lds, s);
}
