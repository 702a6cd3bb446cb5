// Attention kernels of the encoder path (gfx950, MFMA 16x16x32 bf16):
//   spatial  : softmax(q k^T d^-0.5) v over the N patches of one frame   (reference modeling:688-717)
//   temporal : the same over the frames of one patch, causal + KV-cache  (modeling:575-615;
//              streaming copy vqa_enc:491-560: cache append, offset causal mask)
//   pooling  : one learned query against the N tokens of a frame         (modeling:1141-1148)
//
// Shared structure ("swapped" QK^T): the wave computes S^T = K Q^T, so in the MFMA C layout a lane
// holds, for ONE query (column = lane&15), 4 keys per 16-key tile.  The softmax row reduction is
// then in-lane + two xor-shuffles (lanes 16/32 apart), and the exponentiated tile is already in the
// B-operand layout of the second MFMA, O^T = V^T P^T (no LDS round trip for P).  To make the 8
// k-elements a lane feeds to that MFMA contiguous keys, the K rows of each 32-key pair of tiles are
// read in a permuted order (MFMA row i of half hh <-> key 32*kt2 + 8*(i>>2) + 4*hh + (i&3)).
// V is transposed on its way into LDS (V^T[d][key]) so the A-operand is one ds_read_b128.
// ACC = the fp32-accurate mode: fp32 inputs are split into bf16 hi+lo and every product becomes
// hi*hi + hi*lo + lo*hi (same three-term scheme as the GEMM).
#include "sf_common.h"
#include "sf_switches.h"
#include <cstdlib>

#define HD 64  // head_dim supported by these kernels (SigLIP-base/large: 64)

SF_DEVICE f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}

// K-tile LDS swizzle: the 16 keys one ds_read_b128 lane group touches are
// {8a + 4hh + r : a in 0..3, r in 0..3}; (r&1) picks the 128-byte half of the 256-byte bank row, so
// the slot XOR must separate (r>>1, a): 8 values.
SF_DEVICE int kswz(int key) { return ((key >> 1) & 1) | (((key >> 3) & 3) << 1); }

// V^T LDS swizzle of the spatial kernel (4 bits: rows are padded to a multiple of 16 chunks)
SF_DEVICE int vswz(int d) { return ((d ^ (d >> 3)) & 7) | (((d >> 3) & 1) << 3); }

// load 8 consecutive elements (bf16 or fp32 storage) as floats
template <bool F32>
SF_DEVICE void load8(const void* base, size_t elem_off, float (&out)[8]) {
  if (F32) {
    const f32x4_t* p = reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(base) + elem_off);
    const f32x4_t a = p[0], b = p[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) { out[j] = a[j]; out[4 + j] = b[j]; }
  } else {
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const bf16_t*>(base) + elem_off);
#pragma unroll
    for (int j = 0; j < 4; ++j) { out[2 * j] = bf2f(v[j] & 0xffffu); out[2 * j + 1] = bf2f(v[j] >> 16); }
  }
}

// MFMA operand fragment straight from global memory: 8 consecutive elements -> bf16x8 (hi [, lo])
template <bool F32>
SF_DEVICE void load_frag(const void* base, size_t elem_off, bf16x8_t& hi, bf16x8_t& lo) {
  if (F32) {
    float f[8];
    load8<true>(base, elem_off, f);
    unsigned int h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split_bf(f[j], h[j], l[j]);
    u32x4_t hv = {h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
    u32x4_t lv = {l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
    hi = __builtin_bit_cast(bf16x8_t, hv);
    lo = __builtin_bit_cast(bf16x8_t, lv);
  } else {
    hi = *reinterpret_cast<const bf16x8_t*>(reinterpret_cast<const bf16_t*>(base) + elem_off);
    lo = hi;
  }
}

// pack 8 probabilities into the B-operand fragment (hi [, lo])
template <bool ACC>
SF_DEVICE void pack_p(const f32x4_t& a, const f32x4_t& b, bf16x8_t& hi, bf16x8_t& lo) {
  unsigned int h[8], l[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) { split_bf(a[j], h[j], l[j]); split_bf(b[j], h[4 + j], l[4 + j]); }
  u32x4_t hv = {h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
  hi = __builtin_bit_cast(bf16x8_t, hv);
  if (ACC) {
    u32x4_t lv = {l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
    lo = __builtin_bit_cast(bf16x8_t, lv);
  }
}


// Softmax numerator over the score tiles of one query column, in place.  `valid(kt2, key)` says whether
// a key takes part.  The scale (> 0) is folded into the exponent: p = 2^((s - max s) * scale * log2 e),
// one FMA + v_exp_f32 per element; tiles known to be fully valid skip the mask.  Returns sum(p).
template <bool ACC, int MAXNT2, typename Valid>
SF_DEVICE float softmax_tiles(f32x4_t (&s)[MAXNT2][2], int nt2, int g, float scale, int first_masked_tile, Valid valid,
                              float* max2_out = nullptr) {
  float mx = -INFINITY;
#pragma unroll
  for (int kt2 = 0; kt2 < MAXNT2; ++kt2) {
    if (kt2 < nt2) {
      if (kt2 >= first_masked_tile) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (!valid(kt2, kt2 * 32 + g * 8 + hh * 4 + r)) s[kt2][hh][r] = -INFINITY;
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
        mx = fmaxf(mx, fmaxf(fmaxf(s[kt2][hh][0], s[kt2][hh][1]), fmaxf(s[kt2][hh][2], s[kt2][hh][3])));
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float c = scale * 1.44269504088896340736f;
  const float mc = mx * c;
  if (max2_out) *max2_out = mc;         // row maximum in the base-2 exponent domain
  float sum = 0.f;
#pragma unroll
  for (int kt2 = 0; kt2 < MAXNT2; ++kt2) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e = 0.f;
        if (kt2 < nt2) {
          const float a = fmaf(s[kt2][hh][r], c, -mc);
          e = __builtin_amdgcn_exp2f(a);      // v_exp_f32 (1 ulp) in both modes: arguments are <= 0, results below 2^-126 may flush
        }
        s[kt2][hh][r] = e;
        sum += e;
      }
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  return sum;
}

template <bool ACC>
SF_DEVICE void store_ctx(bf16_t* ctx_hi, bf16_t* ctx_lo, size_t off, const f32x4_t& o, float inv) {
  unsigned int h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split_bf(o[j] * inv, h[j], l[j]);
  *reinterpret_cast<u32x2_t*>(ctx_hi + off) = (u32x2_t){h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
  if (ACC) *reinterpret_cast<u32x2_t*>(ctx_lo + off) = (u32x2_t){l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
}

// ================================================================================================
// spatial attention: block = (frame, head), 8 waves; K and V^T of the head live in LDS.
// Latency plan: every wave first issues the Q-fragment loads of its (up to two) query tiles, then
// the block stages K / V^T, so the Q latency hides under the staging; the context rows of a query
// tile are staged through a per-wave LDS patch and leave as whole 128-byte rows.
// ================================================================================================
#define SP_WAVES 8
#define SP_QT 2     // query tiles per wave: 16 * SP_WAVES * SP_QT = 256 >= 224 queries

template <bool ACC, int MAXNT2>
__global__ __launch_bounds__(SP_WAVES * 64) void sf_spatial_attn_kernel(SfAttnArgs p, int vpitch, int qsplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  // qsplit > 1 (few frames in flight, e.g. the per-frame streaming step: 12 (frame, head) problems on 256 CUs): qsplit
  // workgroups per (frame, head); each stages K / V^T (all 8 waves) and then waves 0 .. tpw-1 take one query tile each of
  // the workgroup's share, so qsplit times as many CUs work on the launch
  const int fhq = blockIdx.x / qsplit, qs = blockIdx.x % qsplit;
  const int frame = fhq / p.heads, h = fhq % p.heads;
  const int N = p.N;
  const int nkp = (N + 31) & ~31;
  const int nt2 = nkp >> 5;
  char* k_hi = smem;
  char* k_lo = k_hi + (ACC ? nkp * 128 : 0);
  char* v_hi = k_lo + nkp * 128;
  char* v_lo = v_hi + (ACC ? HD * vpitch : 0);
  char* o_st = v_lo + HD * vpitch + wave * (ACC ? 4096 : 2048);   // per-wave [16 rows][128 B] (hi [, lo])
  const size_t row0 = (size_t)frame * N;
  const int nqt = (N + 15) >> 4;
  const int tpw = (nqt + qsplit - 1) / qsplit;         // query tiles per workgroup when split (<= SP_WAVES)
  auto tile_of = [&](int u) -> int {                     // query tile of this wave in round u, or -1
    if (qsplit == 1) return wave + u * SP_WAVES;
    return (u == 0 && wave < tpw) ? qs * tpw + wave : -1;
  };

  // ---- Q fragments of this wave's query tiles (in flight during the staging below) ----------------
  bf16x8_t qh[SP_QT][2], ql[SP_QT][2];
#pragma unroll
  for (int u = 0; u < SP_QT; ++u) {
    const int qt_u = tile_of(u);
    if (qt_u < 0) continue;
    int qi = qt_u * 16 + l15;
    qi = qi < N ? qi : N - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      load_frag<ACC>(p.q, (row0 + qi) * p.row_pitch_q + h * HD + ks * 32 + g * 8, qh[u][ks], ql[u][ks]);
  }

  // ---- stage K (swizzled rows) -------------------------------------------------------------------
  for (int i = tid; i < nkp * 8; i += SP_WAVES * 64) {
    const int key = i >> 3, c = i & 7;
    u32x4_t hv = {0, 0, 0, 0}, lv = {0, 0, 0, 0};
    if (key < N) {
      bf16x8_t a, b;
      load_frag<ACC>(p.k, (row0 + key) * p.row_pitch_kv + h * HD + c * 8, a, b);
      hv = __builtin_bit_cast(u32x4_t, a);
      lv = __builtin_bit_cast(u32x4_t, b);
    }
    const int off = key * 128 + ((c ^ kswz(key)) << 4);
    *reinterpret_cast<u32x4_t*>(k_hi + off) = hv;
    if (ACC) *reinterpret_cast<u32x4_t*>(k_lo + off) = lv;
  }
  // ---- stage V^T: item = (key pair, 8-wide d chunk) -> 8 dword writes {V[2kp][d], V[2kp+1][d]};
  //      (bank-swizzled, see below) ------------------------------------------------------------------
  for (int i = tid; i < (nkp >> 1) * 8; i += SP_WAVES * 64) {
    const int kp = i >> 3, c = i & 7;
    float v0[8], v1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { v0[j] = 0.f; v1[j] = 0.f; }
    if (2 * kp < N) load8<ACC>(p.v, (row0 + 2 * kp) * p.row_pitch_kv + h * HD + c * 8, v0);
    if (2 * kp + 1 < N) load8<ACC>(p.v, (row0 + 2 * kp + 1) * p.row_pitch_kv + h * HD + c * 8, v1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      unsigned int h0, l0, h1, l1;
      split_bf(v0[j], h0, l0);
      split_bf(v1[j], h1, l1);
      // V^T row d = c*8+j, 16-byte key chunks XOR-swizzled by vswz(d): distinct for the 8 rows a wave
      // writes together (d = 8c+j, c = 0..7) AND for the 16 rows one ds_read_b128 group reads
      const int d = c * 8 + j;
      const int off = d * vpitch + ((((kp >> 2) ^ vswz(d)) << 2) + (kp & 3)) * 4;
      *reinterpret_cast<unsigned int*>(v_hi + off) = h0 | (h1 << 16);
      if (ACC) *reinterpret_cast<unsigned int*>(v_lo + off) = l0 | (l1 << 16);
    }
  }
  __syncthreads();

#pragma unroll
  for (int u = 0; u < SP_QT; ++u) {
    const int qt = tile_of(u);
    if (qt < 0 || qt >= nqt) continue;

    // ---- S^T = K Q^T ---------------------------------------------------------------------------
    f32x4_t s[MAXNT2][2];
#pragma unroll
    for (int kt2 = 0; kt2 < MAXNT2; ++kt2) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        if (kt2 < nt2) {
          const int key = kt2 * 32 + (l15 >> 2) * 8 + hh * 4 + (l15 & 3);
          const int sw = kswz(key);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int off = key * 128 + (((ks * 4 + g) ^ sw) << 4);
            const bf16x8_t kh = *reinterpret_cast<const bf16x8_t*>(k_hi + off);
            if (ACC) {
              const bf16x8_t kl = *reinterpret_cast<const bf16x8_t*>(k_lo + off);
              acc = mfma16(kl, qh[u][ks], acc);
              acc = mfma16(kh, ql[u][ks], acc);
            }
            acc = mfma16(kh, qh[u][ks], acc);
          }
        }
        s[kt2][hh] = acc;
      }
    }
    // ---- softmax over keys (lane: query l15; keys 32*kt2 + 8*g + 4*hh + r) ------------------------
    float max2;
    const float sum = softmax_tiles<ACC, MAXNT2>(s, nt2, g, p.scale, N >> 5, [&](int, int key) { return key < N; }, &max2);
    const float inv = 1.0f / sum;
    if (p.lse2_out && g == 0) {
      const int qi = qt * 16 + l15;
      if (qi < N) p.lse2_out[((size_t)frame * p.heads + h) * N + qi] = max2 + __log2f(sum);
    }
    if (p.probs) {          // output_attentions: the probabilities leave as fp32 rows (wave-uniform branch)
      const int qi = qt * 16 + l15;
      if (qi < N) {
        float* prow = p.probs + (((size_t)frame * p.heads + h) * N + qi) * N;
#pragma unroll
        for (int kt2 = 0; kt2 < MAXNT2; ++kt2)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = kt2 * 32 + g * 8 + hh * 4 + r;
              if (kt2 < nt2 && key < N) prow[key] = s[kt2][hh][r] * inv;
            }
      }
    }

    // ---- O^T = V^T P^T ----------------------------------------------------------------------------
    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt2 = 0; kt2 < MAXNT2; ++kt2) {
      if (kt2 < nt2) {
        bf16x8_t ph, pl;
        pack_p<ACC>(s[kt2][0], s[kt2][1], ph, pl);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int d = dt * 16 + l15;
          const int off = d * vpitch + (((kt2 * 4 + g) ^ vswz(d)) << 4);
          const bf16x8_t vh = *reinterpret_cast<const bf16x8_t*>(v_hi + off);
          if (ACC) {
            const bf16x8_t vl = *reinterpret_cast<const bf16x8_t*>(v_lo + off);
            o[dt] = mfma16(vl, ph, o[dt]);
            o[dt] = mfma16(vh, pl, o[dt]);
          }
          o[dt] = mfma16(vh, ph, o[dt]);
        }
      }
    }
    // ---- context rows: lane holds d = dt*16 + g*4 .. +4 of query l15 -> per-wave LDS patch -> rows --
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      unsigned int hb[4], lb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split_bf(o[dt][j] * inv, hb[j], lb[j]);
      const int off = l15 * 128 + (((dt * 2 + (g >> 1)) ^ (l15 & 7)) << 4) + (g & 1) * 8;
      *reinterpret_cast<u32x2_t*>(o_st + off) = (u32x2_t){hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16)};
      if (ACC) *reinterpret_cast<u32x2_t*>(o_st + 2048 + off) = (u32x2_t){lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16)};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = it * 64 + lane;           // 128 chunks: 16 rows x 8 chunks of 16 B
      const int r = idx >> 3, c = idx & 7;
      const int qi = qt * 16 + r;
      const int off = r * 128 + ((c ^ (r & 7)) << 4);
      if (qi < N) {
        const size_t o_off = (row0 + qi) * p.D + h * HD + c * 8;
        *reinterpret_cast<u32x4_t*>(p.ctx_hi + o_off) = *reinterpret_cast<const u32x4_t*>(o_st + off);
        if (ACC) *reinterpret_cast<u32x4_t*>(p.ctx_lo + o_off) = *reinterpret_cast<const u32x4_t*>(o_st + 2048 + off);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ================================================================================================
// spatial attention, bf16 throughput mode, round 2: the same (frame, head) problem with NO register staging.
//   * K and V go global -> LDS by LDS-DMA (global_load_lds, 16 B per lane, whole 128-byte rows) as plain row-major
//     [key][64] images, 16-byte chunks XOR-swizzled on the SOURCE side (a DMA writes lane-linearly); no VALU, no
//     VGPRs, and the loads of the next workgroup on the CU overlap the arithmetic of this one;
//   * V is never transposed in memory: the A operand of O^T = V^T P^T (lane = head-dim column, k = 8 keys) comes
//     from the row-major image by ds_read_b64_tr_b16 (a 16-lane group reads a [4 x 16] block and receives it transposed);
//     its k order — keys {4g..4g+3} of one 16-key tile and {4g..4g+3} of the next — is exactly how the S^T = K Q^T
//     result tiles sit in the lane (4 consecutive keys per 16-key tile), so K rows are read in NATURAL order and P never
//     touches LDS.  (Round 1 transposed V on its way into LDS with 8 dword writes per item and permuted the K rows:
//     19 of the kernel's 57 us at B = 8 were staging.)
// Used unless the probabilities (output_attentions) or the fp32-accurate mode are requested.
// ================================================================================================
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
typedef const __attribute__((address_space(1))) void* sp_gptr_t;
typedef __attribute__((address_space(3))) void* sp_lptr_t;

// chunk swizzle of the row-major images (same as sf_attention_bwd.hip): bijective in row bits 1..3 (row fragments of 16
// rows conflict-free), upper two bits bijective in row bits 1..2 (the 8 rows of a half-wave transposed read)
// (round 6) gfx950 serves a ds_read_b128 in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32), not in contiguous sixteens: a row
// fragment's group holds rows 0-3 and 12-15 at chunk c and rows 4-11 at chunk c ^ 1.  With 128-byte rows the even rows share one half of the
// 256-byte bank row, so {s(0), s(2), s(12), s(14)} and {s(4), s(6), s(8), s(10)} ^ 1 must partition the 8 slots: s = 2 * ((row >> 1) & 3)
// gives {0, 2, 4, 6} and {5, 7, 1, 3}.  Rounds 2-5 also folded row bit 3 in (| ((row >> 3) & 1)): bijective over 16 contiguous rows, 2-way
// conflicted on the real groups (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.24-0.28 on these kernels).  SF_ATTN_SWZ_LEGACY: the old function (A/B builds).
#ifdef SF_ATTN_SWZ_LEGACY
SF_DEVICE int sp_bswz(int row) { return (((row >> 1) & 3) << 1) | ((row >> 3) & 1); }
#else
SF_DEVICE int sp_bswz(int row) { return ((row >> 1) & 3) << 1; }
#endif
SF_DEVICE int sp_img_off(int row, int chunk) { return row * 128 + ((chunk ^ sp_bswz(row)) << 4); }
SF_DEVICE bf16x8_t sp_row_frag(const char* img, int row, int chunk) {
  return *reinterpret_cast<const bf16x8_t*>(img + sp_img_off(row, chunk));
}
// token-major fragment: lane (l15 = column e of head-dim tile et, g) gets rows {r0+4g..+3} and {r0+16+4g..+3}
SF_DEVICE bf16x8_t sp_tr_frag(const char* img, int r0, int et, int lane) {
  const int t16 = lane & 15, g = lane >> 4;
  const int row = r0 + 4 * g + (t16 >> 2);
  const int off = sp_img_off(row, 2 * et + ((t16 & 3) >> 1)) + ((t16 & 1) << 3);
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(img + off));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(img + off + 16 * 128));
  bf16x8_t f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

// ACC = the fp32-accurate mode on the same structure: q / k / v arrive as hi + lo bf16 planes (the qkv GEMM writes them
// instead of fp32, same bytes), four images (K hi, V hi, K lo, V lo) land by DMA, every product is three MFMAs
// (lo*hi + hi*lo + hi*hi) and the probabilities are split into hi + lo in registers.
// DROP = dropout on the probabilities (training forward with attention_probs_dropout_prob > 0): its own instance, so that the mask
// arithmetic costs the plain kernel no registers (117 VGPRs = two workgroups per CU; with the branch compiled in: 156, one workgroup)
// NTC = compile-time number of score tiles that hold real keys (13 for the 196 patches of a 224^2 frame), 0 = decided at run time.
// With the count known every `if (tile exists)` of the score / PV loops folds away: hipcc then schedules the 28 K-fragment reads and
// MFMAs of a query tile as ONE block (reads of the next tiles in flight under the MFMAs of this one) instead of fourteen basic blocks
// of [2 ds_read -> wait -> MFMA -> wait -> MFMA] — round 4, found in the ISA, not in a profile.
template <int MAXNT, bool ACC, bool DROP = false, int NTC = 0>       // 16-key tiles held in registers: 14 -> N <= 224
__global__ __launch_bounds__(SP_WAVES * 64) void sf_spatial_attn_dma_kernel(SfAttnArgs p, int qsplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int fhq = blockIdx.x / qsplit, qs = blockIdx.x % qsplit;
  const int frame = fhq / p.heads, h = fhq % p.heads;
  const int N = p.N;
  const int nkp = NTC ? ((NTC + 1) & ~1) * 16 : (N + 31) & ~31;      // keys padded to whole 32-key pairs of tiles
  const int nt = nkp >> 4;
  char* k_img = smem;
  char* v_img = k_img + nkp * 128;
  char* kl_img = v_img + nkp * 128;                      // ACC only
  char* vl_img = kl_img + nkp * 128;
  char* o_st = v_img + nkp * 128 * (ACC ? 3 : 1) + wave * (ACC ? 4096 : 2048);          // per-wave [16 rows][128 B] (hi [, lo])
  const size_t row0 = (size_t)frame * N;
  const int nqt = (N + 15) >> 4;
  const int tpw = (nqt + qsplit - 1) / qsplit;
  auto tile_of = [&](int u) -> int {
    if (qsplit == 1) return wave + u * SP_WAVES;
    return (u == 0 && wave < tpw) ? qs * tpw + wave : -1;
  };

  // ---- K / V images by LDS-DMA: instruction j covers rows 8j .. 8j+7 (lane -> row 8j + lane/8, slot lane%8) -------------
  const bf16_t* kbase = reinterpret_cast<const bf16_t*>(p.k) + h * HD;
  const bf16_t* vbase = reinterpret_cast<const bf16_t*>(p.v) + h * HD;
  for (int j = wave; j < (nkp >> 3); j += SP_WAVES) {
    const int row = j * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ sp_bswz(row);         // the slot this lane fills holds logical chunk `chunk`
    const int key = row < N ? row : N - 1;               // padding rows repeat the last key (masked in the softmax, P = 0 for V)
    const size_t src = (row0 + key) * (size_t)p.row_pitch_kv + chunk * 8;
    __builtin_amdgcn_global_load_lds((sp_gptr_t)(kbase + src), (sp_lptr_t)(k_img + j * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((sp_gptr_t)(vbase + src), (sp_lptr_t)(v_img + j * 1024), 16, 0, 0);
    if (ACC) {
      __builtin_amdgcn_global_load_lds((sp_gptr_t)(kbase + p.lo_plane_off + src), (sp_lptr_t)(kl_img + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((sp_gptr_t)(vbase + p.lo_plane_off + src), (sp_lptr_t)(vl_img + j * 1024), 16, 0, 0);
    }
  }
  // ---- Q fragments of this wave's query tiles (register loads, in flight with the DMA) -------------------------------
  bf16x8_t qh[SP_QT][2], ql[SP_QT][2];
#pragma unroll
  for (int u = 0; u < SP_QT; ++u) {
    const int qt_u = tile_of(u);
    if (qt_u < 0) continue;
    int qi = qt_u * 16 + l15;
    qi = qi < N ? qi : N - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16_t* qp = reinterpret_cast<const bf16_t*>(p.q) + (row0 + qi) * p.row_pitch_q + h * HD + ks * 32 + g * 8;
      qh[u][ks] = *reinterpret_cast<const bf16x8_t*>(qp);
      if (ACC) ql[u][ks] = *reinterpret_cast<const bf16x8_t*>(qp + p.lo_plane_off);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const float c2 = p.scale * 1.44269504088896340736f;
#pragma unroll
  for (int u = 0; u < SP_QT; ++u) {
    const int qt = tile_of(u);
    if (qt < 0 || qt >= nqt) continue;
    // ---- S^T = K Q^T, natural key order: lane (query l15, g) holds keys 16 jt + 4 g + r ---------------------------------
    f32x4_t s[MAXNT];
#pragma unroll
    for (int jt = 0; jt < MAXNT; ++jt) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      if (NTC ? jt < NTC : jt * 16 < N) {                  // tiles made of padding keys only are never computed
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8_t kh = sp_row_frag(k_img, jt * 16 + l15, ks * 4 + g);
          if (ACC) {
            acc = mfma16(sp_row_frag(kl_img, jt * 16 + l15, ks * 4 + g), qh[u][ks], acc);
            acc = mfma16(kh, ql[u][ks], acc);
          }
          acc = mfma16(kh, qh[u][ks], acc);
        }
      }
      s[jt] = acc;
    }
    // mask + row maximum in a SECOND pass over the finished tiles.  Taking the maximum of a tile right behind its MFMA
    // (v_max on the accumulator VGPRs two instructions after v_mfma_f32_16x16x32_bf16) gave run-to-run different maxima on
    // gfx950 / hipcc 7.2 — the VALU read of a still-in-flight MFMA result; results stayed within tolerance (softmax is
    // shift-invariant) but were not bit-reproducible.  Here every tile is read long after its last MFMA issued.
    __builtin_amdgcn_sched_barrier(0);
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < MAXNT; ++jt) {
      if (NTC ? jt < NTC : jt * 16 < N) {
        if (NTC ? jt == NTC - 1 : jt * 16 + 16 > N) {      // the tile that holds the first padding keys
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (jt * 16 + 4 * g + r >= N) s[jt][r] = -INFINITY;
        }
        mx = fmaxf(mx, fmaxf(fmaxf(s[jt][0], s[jt][1]), fmaxf(s[jt][2], s[jt][3])));
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * c2;
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < MAXNT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e = 0.f;
        if (NTC ? jt < NTC : jt * 16 < N) e = __builtin_amdgcn_exp2f(fmaf(s[jt][r], c2, -mc));
        s[jt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (p.lse2_out && g == 0) {        // training forward: base-2 log-sum-exp of the scaled scores, kept for the backward kernel
      const int qi = qt * 16 + l15;
      if (qi < N) p.lse2_out[((size_t)frame * p.heads + h) * N + qi] = mc + __log2f(sum);
    }
    if (DROP && p.drop.on) {           // training: dropout on the (normalised) probabilities (modeling:705) — `sum` above is unmasked
      const int qd = qt * 16 + l15 < N ? qt * 16 + l15 : N - 1;
      const unsigned dbase = (unsigned)((((size_t)frame * p.heads + h) * N + qd) * N);
#pragma unroll
      for (int jt = 0; jt < MAXNT; ++jt)
        if (NTC ? jt < NTC : jt * 16 < N) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = jt * 16 + 4 * g + r;
            s[jt][r] *= sf_drop_factor(p.drop, dbase + (unsigned)(key < N ? key : N - 1));
          }
        }
    }

    // ---- O^T = V^T P^T: 32 keys per step, V^T fragments by transposed reads of the row-major image ------------------------
    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j2 = 0; j2 < MAXNT / 2; ++j2) {
      if (NTC ? 2 * j2 < ((NTC + 1) & ~1) : 2 * j2 < nt) {
        const u32x4_t pu = {pack_bf2(s[2 * j2][0], s[2 * j2][1]), pack_bf2(s[2 * j2][2], s[2 * j2][3]),
                            pack_bf2(s[2 * j2 + 1][0], s[2 * j2 + 1][1]), pack_bf2(s[2 * j2 + 1][2], s[2 * j2 + 1][3])};
        const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pu);
        bf16x8_t pl;
        if (ACC) {
          u32x4_t lu;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const f32x4_t& sv = s[2 * j2 + (w >> 1)];
            const int r0 = (w & 1) * 2;
            lu[w] = pack_bf2(sv[r0] - bf2f(pu[w] & 0xffffu), sv[r0 + 1] - bf2f(pu[w] >> 16));
          }
          pl = __builtin_bit_cast(bf16x8_t, lu);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8_t vh = sp_tr_frag(v_img, j2 * 32, dt, lane);
          if (ACC) {
            o[dt] = mfma16(sp_tr_frag(vl_img, j2 * 32, dt, lane), pf, o[dt]);
            o[dt] = mfma16(vh, pl, o[dt]);
          }
          o[dt] = mfma16(vh, pf, o[dt]);
        }
      }
    }
    // ---- context rows: lane holds d = dt*16 + g*4 .. +4 of query l15 -> per-wave LDS patch -> whole 128-byte rows ---------
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int off = l15 * 128 + (((dt * 2 + (g >> 1)) ^ (l15 & 7)) << 4) + (g & 1) * 8;
      const f32x4_t ov = o[dt] * inv;
      const u32x2_t hv = {pack_bf2(ov[0], ov[1]), pack_bf2(ov[2], ov[3])};
      *reinterpret_cast<u32x2_t*>(o_st + off) = hv;
      if (ACC)
        *reinterpret_cast<u32x2_t*>(o_st + 2048 + off) = (u32x2_t){pack_bf2(ov[0] - bf2f(hv[0] & 0xffffu), ov[1] - bf2f(hv[0] >> 16)),
                                                                  pack_bf2(ov[2] - bf2f(hv[1] & 0xffffu), ov[3] - bf2f(hv[1] >> 16))};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = it * 64 + lane;
      const int r = idx >> 3, c = idx & 7;
      const int qi = qt * 16 + r;
      if (qi < N) {
        const size_t oo = (row0 + qi) * p.D + h * HD + c * 8;
        *reinterpret_cast<u32x4_t*>(p.ctx_hi + oo) = *reinterpret_cast<const u32x4_t*>(o_st + r * 128 + ((c ^ (r & 7)) << 4));
        if (ACC) *reinterpret_cast<u32x4_t*>(p.ctx_lo + oo) = *reinterpret_cast<const u32x4_t*>(o_st + 2048 + r * 128 + ((c ^ (r & 7)) << 4));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ================================================================================================
// Round 4: the same (frame, head) problem on a PERSISTENT workgroup with explicit double buffering.  The kernel above leaves
// the overlap of one problem's K / V load with another's arithmetic to two independent workgroups per CU that start in
// phase (0.43 of the HBM roofline at 8 clips: each problem pays its own load latency).  Here one workgroup of 16 waves walks
// its problems p = blockIdx.x, + gridDim.x, ... with TWO K / V image pairs in LDS: the LDS-DMA of problem i + 1 and the Q
// fragment loads of its tile are issued before the arithmetic of problem i starts, so a problem costs max(load, compute).
// A wave owns ONE 16-query tile (13 of 16 waves busy at 196 tokens; all 16 issue the DMA pieces).  Two barriers per problem:
// A = every wave's pieces of problem i have landed (counted vmcnt: the pieces of problem i + 1 stay in flight), B = every
// wave has finished reading image i & 1 before the pieces of problem i + 2 may overwrite it.
// bf16 mode, N <= 224, no probabilities; the tile arithmetic is the one of sf_spatial_attn_dma_kernel.
// ================================================================================================
#ifdef SF_LAB      // measured slower than the two-workgroups-per-CU kernel above (profiles/r04_spatial_pers_lab.txt): lab library only, source in tools/lab/
#include "../../tools/lab/sf_spatial_pers.inc"
#endif   // SF_LAB

// ================================================================================================
// spatial attention for N > 224 tokens per frame (higher-resolution inputs, modeling:380-411 resizes
// the position table): block = (frame, head, block of 128 queries), 8 waves = one 16-query tile each;
// keys stream through LDS in chunks of 128 (K rows + V^T, same images and swizzles as above) with the
// online-softmax recurrence (running max m, running sum l, accumulators rescaled by 2^((m - m')c)).
// The score layout keeps a lane on one query, so the rescale is lane-local.
// ================================================================================================
#define SL_KC 128                 // keys per chunk (= 4 tiles of 32; V^T rows are 256 B = 16 chunks, as vswz wants)

template <bool ACC>
__global__ __launch_bounds__(SP_WAVES * 64) void sf_spatial_attn_large_kernel(SfAttnArgs p, int qblocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int VP = 2 * SL_KC;   // V^T row pitch in bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int qb = blockIdx.x % qblocks;
  const int fh = blockIdx.x / qblocks;
  const int frame = fh / p.heads, h = fh % p.heads;
  const int N = p.N;
  char* k_hi = smem;
  char* k_lo = k_hi + (ACC ? SL_KC * 128 : 0);
  char* v_hi = k_lo + SL_KC * 128;
  char* v_lo = v_hi + (ACC ? HD * VP : 0);
  char* o_st = v_lo + HD * VP + wave * (ACC ? 4096 : 2048);
  const size_t row0 = (size_t)frame * N;
  const int qt = qb * SP_WAVES + wave;                 // this wave's query tile
  const bool active = qt * 16 < N;

  bf16x8_t qh[2], ql[2];
  {
    int qi = qt * 16 + l15;
    qi = qi < N ? qi : N - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) load_frag<ACC>(p.q, (row0 + qi) * p.row_pitch_q + h * HD + ks * 32 + g * 8, qh[ks], ql[ks]);
  }
  const float c = p.scale * 1.44269504088896340736f;
  float m = -INFINITY, l = 0.f;                        // running max (raw scores) and sum, per query (lane l15, partial over g)
  f32x4_t o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  for (int key0 = 0; key0 < N; key0 += SL_KC) {
    __syncthreads();                                   // previous chunk fully consumed
    for (int i = tid; i < SL_KC * 8; i += SP_WAVES * 64) {
      const int key = i >> 3, ch = i & 7;
      u32x4_t hv = {0, 0, 0, 0}, lv = {0, 0, 0, 0};
      if (key0 + key < N) {
        bf16x8_t a, b;
        load_frag<ACC>(p.k, (row0 + key0 + key) * p.row_pitch_kv + h * HD + ch * 8, a, b);
        hv = __builtin_bit_cast(u32x4_t, a);
        lv = __builtin_bit_cast(u32x4_t, b);
      }
      const int off = key * 128 + ((ch ^ kswz(key)) << 4);
      *reinterpret_cast<u32x4_t*>(k_hi + off) = hv;
      if (ACC) *reinterpret_cast<u32x4_t*>(k_lo + off) = lv;
    }
    for (int i = tid; i < (SL_KC >> 1) * 8; i += SP_WAVES * 64) {
      const int kp = i >> 3, ch = i & 7;
      float v0[8], v1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { v0[j] = 0.f; v1[j] = 0.f; }
      if (key0 + 2 * kp < N) load8<ACC>(p.v, (row0 + key0 + 2 * kp) * p.row_pitch_kv + h * HD + ch * 8, v0);
      if (key0 + 2 * kp + 1 < N) load8<ACC>(p.v, (row0 + key0 + 2 * kp + 1) * p.row_pitch_kv + h * HD + ch * 8, v1);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned int h0, l0, h1, l1;
        split_bf(v0[j], h0, l0);
        split_bf(v1[j], h1, l1);
        const int d = ch * 8 + j;
        const int off = d * VP + ((((kp >> 2) ^ vswz(d)) << 2) + (kp & 3)) * 4;
        *reinterpret_cast<unsigned int*>(v_hi + off) = h0 | (h1 << 16);
        if (ACC) *reinterpret_cast<unsigned int*>(v_lo + off) = l0 | (l1 << 16);
      }
    }
    __syncthreads();
    if (!active) continue;                             // wave-uniform; barriers stay outside

    f32x4_t s[4][2];
#pragma unroll
    for (int kt2 = 0; kt2 < 4; ++kt2)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        const int key = kt2 * 32 + (l15 >> 2) * 8 + hh * 4 + (l15 & 3);
        const int sw = kswz(key);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int off = key * 128 + (((ks * 4 + g) ^ sw) << 4);
          const bf16x8_t kh = *reinterpret_cast<const bf16x8_t*>(k_hi + off);
          if (ACC) {
            const bf16x8_t kl = *reinterpret_cast<const bf16x8_t*>(k_lo + off);
            acc = mfma16(kl, qh[ks], acc);
            acc = mfma16(kh, ql[ks], acc);
          }
          acc = mfma16(kh, qh[ks], acc);
        }
        s[kt2][hh] = acc;
      }
    // lane: query l15; keys key0 + 32*kt2 + 8*g + 4*hh + r
    float cm = -INFINITY;
#pragma unroll
    for (int kt2 = 0; kt2 < 4; ++kt2)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (key0 + kt2 * 32 + g * 8 + hh * 4 + r >= N) s[kt2][hh][r] = -INFINITY;
          cm = fmaxf(cm, s[kt2][hh][r]);
        }
    cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
    cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
    const float mn = fmaxf(m, cm);                     // finite: every chunk holds at least one valid key
    const float alpha = __builtin_amdgcn_exp2f((m - mn) * c);
    const float mc = mn * c;
    float add = 0.f;
#pragma unroll
    for (int kt2 = 0; kt2 < 4; ++kt2)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = fmaf(s[kt2][hh][r], c, -mc);
          const float e = __builtin_amdgcn_exp2f(a);
          s[kt2][hh][r] = e;
          add += e;
        }
    l = l * alpha + add;
    m = mn;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
#pragma unroll
    for (int kt2 = 0; kt2 < 4; ++kt2) {
      bf16x8_t ph, pl;
      pack_p<ACC>(s[kt2][0], s[kt2][1], ph, pl);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int d = dt * 16 + l15;
        const int off = d * VP + (((kt2 * 4 + g) ^ vswz(d)) << 4);
        const bf16x8_t vh = *reinterpret_cast<const bf16x8_t*>(v_hi + off);
        if (ACC) {
          const bf16x8_t vl = *reinterpret_cast<const bf16x8_t*>(v_lo + off);
          o[dt] = mfma16(vl, ph, o[dt]);
          o[dt] = mfma16(vh, pl, o[dt]);
        }
        o[dt] = mfma16(vh, ph, o[dt]);
      }
    }
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = active ? 1.0f / l : 0.f;
  if (p.probs) {
    // output_attentions above 224 patches (round 4): a second sweep over the keys with the FINAL row maximum and sum — the scores of a
    // chunk are recomputed from the re-staged K rows and leave as probabilities, [frames, heads, N, N] fp32 (modeling:703-716)
    const float mc = m * c;
    for (int key0 = 0; key0 < N; key0 += SL_KC) {
      __syncthreads();
      for (int i = tid; i < SL_KC * 8; i += SP_WAVES * 64) {
        const int key = i >> 3, ch = i & 7;
        u32x4_t hv = {0, 0, 0, 0}, lv = {0, 0, 0, 0};
        if (key0 + key < N) {
          bf16x8_t a, b;
          load_frag<ACC>(p.k, (row0 + key0 + key) * p.row_pitch_kv + h * HD + ch * 8, a, b);
          hv = __builtin_bit_cast(u32x4_t, a);
          lv = __builtin_bit_cast(u32x4_t, b);
        }
        const int off = key * 128 + ((ch ^ kswz(key)) << 4);
        *reinterpret_cast<u32x4_t*>(k_hi + off) = hv;
        if (ACC) *reinterpret_cast<u32x4_t*>(k_lo + off) = lv;
      }
      __syncthreads();
      if (!active) continue;
      const int qi = qt * 16 + l15;
#pragma unroll
      for (int kt2 = 0; kt2 < 4; ++kt2)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
          const int key = kt2 * 32 + (l15 >> 2) * 8 + hh * 4 + (l15 & 3);
          const int sw = kswz(key);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int off = key * 128 + (((ks * 4 + g) ^ sw) << 4);
            const bf16x8_t kh = *reinterpret_cast<const bf16x8_t*>(k_hi + off);
            if (ACC) {
              const bf16x8_t kl = *reinterpret_cast<const bf16x8_t*>(k_lo + off);
              acc = mfma16(kl, qh[ks], acc);
              acc = mfma16(kh, ql[ks], acc);
            }
            acc = mfma16(kh, qh[ks], acc);
          }
          // lane: query l15; keys key0 + 32 kt2 + 8 g + 4 hh + r
          const int kbase = key0 + kt2 * 32 + g * 8 + hh * 4;
          if (qi < N) {
            float* dst = p.probs + (((size_t)frame * p.heads + h) * N + qi) * (size_t)N + kbase;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kbase + r < N) dst[r] = __builtin_amdgcn_exp2f(fmaf(acc[r], c, -mc)) * inv;
          }
        }
    }
  }
  if (!active) return;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    unsigned int hb[4], lb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_bf(o[dt][j] * inv, hb[j], lb[j]);
    const int off = l15 * 128 + (((dt * 2 + (g >> 1)) ^ (l15 & 7)) << 4) + (g & 1) * 8;
    *reinterpret_cast<u32x2_t*>(o_st + off) = (u32x2_t){hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16)};
    if (ACC) *reinterpret_cast<u32x2_t*>(o_st + 2048 + off) = (u32x2_t){lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16)};
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = it * 64 + lane;
    const int r = idx >> 3, ch = idx & 7;
    const int qi = qt * 16 + r;
    const int off = r * 128 + ((ch ^ (r & 7)) << 4);
    if (qi < N) {
      const size_t o_off = (row0 + qi) * p.D + h * HD + ch * 8;
      *reinterpret_cast<u32x4_t*>(p.ctx_hi + o_off) = *reinterpret_cast<const u32x4_t*>(o_st + off);
      if (ACC) *reinterpret_cast<u32x4_t*>(p.ctx_lo + o_off) = *reinterpret_cast<const u32x4_t*>(o_st + 2048 + off);
    }
  }
}

static int vt_pitch(int nkp) {
  // bytes per V^T row: 2*nkp + pad with pitch % 256 in {32, 224}: the 16 d-rows of a ds_read_b128
  // lane group then fall on 16 distinct 16-byte slots ("LDS layouts").
  for (int pad = 0; pad < 256; pad += 32) {
    const int m = (2 * nkp + pad) % 256;
    if (m == 32 || m == 224) return 2 * nkp + pad;
  }
  return 2 * nkp + 32;
}

// accurate mode: may the caller hand q / k / v as hi + lo bf16 planes (DMA kernel) instead of fp32?
bool sf_spatial_planes_ok(int N, bool probs) {
  const bool off = sf_sw(SW_DISABLE_SPATIAL_DMA_ACC) != nullptr;
  return !off && !probs && N > 0 && ((N + 31) & ~31) <= 32 * 7;
}

hipError_t sf_launch_spatial_attention(const SfAttnArgs& a, bool accurate, hipStream_t s) {
  if (a.head_dim && a.head_dim != HD) return sf_launch_attention_generic(a, a.head_dim, false, s);      // sf_attention_generic.hip
  if (a.D != a.heads * HD || a.N <= 0 || a.frames <= 0) return hipErrorInvalidValue;
  if (a.drop.on && (accurate || a.probs || a.N > 224 || sf_sw(SW_DISABLE_SPATIAL_DMA) || (a.row_pitch_kv % 8))) return hipErrorInvalidValue;   // dropout: DMA kernel only
  const int nkp = (a.N + 31) & ~31;
  if (nkp > 32 * 7) {                               // more than 224 tokens per frame: streaming-key kernel
    const int qblocks = (a.N + 127) / 128;
    const size_t lds = (size_t)(SL_KC * 128 + HD * 2 * SL_KC + SP_WAVES * 2048) * (accurate ? 2 : 1);
    static SfPerDeviceOnce attr_l;
    if (attr_l.first()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_spatial_attn_large_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    }
    const dim3 grid(a.frames * a.heads * qblocks), block(SP_WAVES * 64);
    if (accurate) hipLaunchKernelGGL((sf_spatial_attn_large_kernel<true>), grid, block, lds, s, a, qblocks);
    else hipLaunchKernelGGL((sf_spatial_attn_large_kernel<false>), grid, block, lds, s, a, qblocks);
    return hipGetLastError();
  }
  const int vp = (2 * nkp + 255) & ~255;          // V^T row pitch: whole groups of 16 chunks (vswz is 4-bit)
  const size_t lds = (size_t)(nkp * 128 + HD * vp + SP_WAVES * 2048) * (accurate ? 2 : 1);
  // few (frame, head) problems: split the query tiles of each over several workgroups (each re-stages K / V^T: 50 KB)
  const int nqt = (a.N + 15) >> 4, fh = a.frames * a.heads;
  int qsplit = 1;
  if (!a.probs && a.N > 32) {
    if (fh <= 32) qsplit = (nqt + 1) / 2;           // two query tiles per workgroup: 84 workgroups for one 196-patch frame
    else if (fh <= 64) qsplit = (nqt + 3) / 4;
    else if (fh <= 128) qsplit = (nqt + 7) / 8;
    const char* env = sf_sw(SW_SPATIAL_TPW);   // tuning: query tiles per workgroup
    if (env) qsplit = (nqt + atoi(env) - 1) / atoi(env);
    if (qsplit < 1) qsplit = 1;
  }
  const dim3 grid(a.frames * a.heads * qsplit), block(SP_WAVES * 64);
  static SfPerDeviceOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_spatial_attn_kernel<false, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_spatial_attn_kernel<true, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const bool dma_off = sf_sw(SW_DISABLE_SPATIAL_DMA) != nullptr;
  static SfPerDeviceOnce attr2;
  if (attr2.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_spatial_attn_dma_kernel<14, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_spatial_attn_dma_kernel<14, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  if (!accurate && !a.probs && !dma_off && (a.row_pitch_kv % 8) == 0) {
#ifdef SF_LAB
    // many (frame, head) problems: the persistent double-buffered kernel (>= 4 problems per CU, so that the pipeline has something to overlap)
    const bool pers_off = sf_sw(SW_SPATIAL_PERS) == nullptr;
    static int cus = 0;
    if (!cus) {
      int dev = 0;
      hipDeviceProp_t prop;
      cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    if (!pers_off && qsplit == 1 && fh >= 4 * cus && a.N >= 64) {
      const size_t lds3 = (size_t)nkp * 512 + SPP_WAVES * 2048;
      static SfPerDeviceOnce attr3;
      if (attr3.first())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_spatial_attn_pers_kernel<14>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((sf_spatial_attn_pers_kernel<14>), dim3(cus), dim3(SPP_WAVES * 64), lds3, s, a, fh);
      return hipGetLastError();
    }
#endif
    const size_t lds2 = (size_t)nkp * 256 + SP_WAVES * 2048;
    if (a.drop.on) {
      static SfPerDeviceOnce attr4;
      if (attr4.first())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_spatial_attn_dma_kernel<14, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((sf_spatial_attn_dma_kernel<14, false, true>), grid, block, lds2, s, a, qsplit);
      return hipGetLastError();
    }
    const bool ntc_off = sf_sw(SW_DISABLE_SPATIAL_NTC) != nullptr;      // A/B switch
    if (!ntc_off && ((a.N + 15) >> 4) == 13) {       // 193 .. 208 tokens per frame (224^2 inputs): the tile count as a compile-time constant
      static SfPerDeviceOnce attr5;
      if (attr5.first())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_spatial_attn_dma_kernel<14, false, false, 13>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((sf_spatial_attn_dma_kernel<14, false, false, 13>), grid, block, lds2, s, a, qsplit);
      return hipGetLastError();
    }
    hipLaunchKernelGGL((sf_spatial_attn_dma_kernel<14, false>), grid, block, lds2, s, a, qsplit);
    return hipGetLastError();
  }
  if (accurate && !a.in_is_f32) {       // hi + lo bf16 planes (sf_spatial_planes_ok): the DMA kernel with three products
    if (a.probs || a.lo_plane_off <= 0 || (a.row_pitch_kv % 8) || (a.lo_plane_off % 8)) return hipErrorInvalidValue;
    const size_t lds2 = (size_t)nkp * 512 + SP_WAVES * 4096;
    const bool ntc_acc_off = sf_sw(SW_DISABLE_SPATIAL_NTC) != nullptr;      // A/B switch (shared with the bf16 instance)
    if (!ntc_acc_off && ((a.N + 15) >> 4) == 13) {       // compile-time tile count: 129.9 -> 118.2 us per launch, bit-identical (profiles/r04_spatial_ntc_acc_ab.txt)
      static SfPerDeviceOnce attr6;
      if (attr6.first())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_spatial_attn_dma_kernel<14, true, false, 13>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((sf_spatial_attn_dma_kernel<14, true, false, 13>), grid, block, lds2, s, a, qsplit);
      return hipGetLastError();
    }
    hipLaunchKernelGGL((sf_spatial_attn_dma_kernel<14, true>), grid, block, lds2, s, a, qsplit);
    return hipGetLastError();
  }
  if (accurate) hipLaunchKernelGGL((sf_spatial_attn_kernel<true, 7>), grid, block, lds, s, a, vp, qsplit);
  else hipLaunchKernelGGL((sf_spatial_attn_kernel<false, 7>), grid, block, lds, s, a, vp, qsplit);
  return hipGetLastError();
}

// ================================================================================================
// temporal attention: ONE WAVE per (b, patch, head) task, four consecutive tasks (adjacent heads =
// adjacent 128-byte lines) per workgroup, no workgroup barrier.  Every global load of the task
// (Q, K fragments, V rows) is issued up front so a task costs one memory latency; V^T goes through
// a wave-private LDS patch, the context rows leave as whole 128-byte segments.
// Row addressing: see SfAttnArgs.
// ================================================================================================
template <bool ACC, int MAXNT2>
__global__ __launch_bounds__(256) void sf_temporal_attn_kernel(SfAttnArgs p, int vpitch, int ntasks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool PRELOAD_K = MAXNT2 <= 2;          // K fragments of the whole key range live in registers
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int task = blockIdx.x * (blockDim.x >> 6) + wave;
  if (task >= ntasks) return;
  const int h = task % p.heads, bn = task / p.heads;
  const int b = bn / p.N, n = bn % p.N;
  const int Tk = p.Tk, Tq = p.Tq;
  const int tkp = (Tk + 31) & ~31;
  const int nt2 = tkp >> 5;
  const int patch = HD * vpitch * (ACC ? 2 : 1) + (ACC ? 4096 : 2048);
  char* v_hi = smem + (size_t)wave * patch;
  char* v_lo = v_hi + (ACC ? HD * vpitch : 0);
  char* o_st = v_lo + HD * vpitch;

  // ---- issue every load of the task ------------------------------------------------------------------
  bf16x8_t qh[2], ql[2];
  {
    int t = l15 < Tq ? l15 : Tq - 1;
    const size_t qrow = ((size_t)b * p.Tq_cap + p.q_t0 + t) * p.N + n;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) load_frag<ACC>(p.q, qrow * p.row_pitch_q + h * HD + ks * 32 + g * 8, qh[ks], ql[ks]);
  }
  bf16x8_t kfh[PRELOAD_K ? MAXNT2 : 1][2][2], kfl[PRELOAD_K ? MAXNT2 : 1][2][2];
  if (PRELOAD_K) {
#pragma unroll
    for (int kt2 = 0; kt2 < MAXNT2; ++kt2)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        int key = kt2 * 32 + (l15 >> 2) * 8 + hh * 4 + (l15 & 3);
        key = key < Tk ? key : Tk - 1;                     // clamped rows are masked below
        const size_t krow = ((size_t)b * p.Tcap + key) * p.N + n;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          load_frag<ACC>(p.k, krow * p.row_pitch_kv + h * HD + ks * 32 + g * 8, kfh[kt2][hh][ks], kfl[kt2][hh][ks]);
      }
  }
  // V rows -> V^T patch: item = (key pair, 8-wide d chunk)
  for (int i = lane; i < (tkp >> 1) * 8; i += 64) {
    const int kp = i >> 3, c = i & 7;
    float v0[8], v1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { v0[j] = 0.f; v1[j] = 0.f; }
    if (2 * kp < Tk) load8<ACC>(p.v, (((size_t)b * p.Tcap + 2 * kp) * p.N + n) * p.row_pitch_kv + h * HD + c * 8, v0);
    if (2 * kp + 1 < Tk) load8<ACC>(p.v, (((size_t)b * p.Tcap + 2 * kp + 1) * p.N + n) * p.row_pitch_kv + h * HD + c * 8, v1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      unsigned int h0, l0, h1, l1;
      split_bf(v0[j], h0, l0);
      split_bf(v1[j], h1, l1);
      const int off = (c * 8 + j) * vpitch + kp * 4;
      *reinterpret_cast<unsigned int*>(v_hi + off) = h0 | (h1 << 16);
      if (ACC) *reinterpret_cast<unsigned int*>(v_lo + off) = l0 | (l1 << 16);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const int nqt = (Tq + 15) >> 4;
  for (int qt = 0; qt < nqt; ++qt) {
    int t = qt * 16 + l15;
    const bool qvalid = t < Tq;
    if (!qvalid) t = Tq - 1;
    if (qt > 0) {      // further query tiles (T_new > 16): reload Q
      const size_t qrow = ((size_t)b * p.Tq_cap + p.q_t0 + t) * p.N + n;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) load_frag<ACC>(p.q, qrow * p.row_pitch_q + h * HD + ks * 32 + g * 8, qh[ks], ql[ks]);
    }
    f32x4_t s[MAXNT2][2];
#pragma unroll
    for (int kt2 = 0; kt2 < MAXNT2; ++kt2) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        if (kt2 < nt2) {
          int key = kt2 * 32 + (l15 >> 2) * 8 + hh * 4 + (l15 & 3);
          key = key < Tk ? key : Tk - 1;
          const size_t krow = ((size_t)b * p.Tcap + key) * p.N + n;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t kh, kl;
            if (PRELOAD_K) { kh = kfh[kt2][hh][ks]; kl = kfl[kt2][hh][ks]; }
            else load_frag<ACC>(p.k, krow * p.row_pitch_kv + h * HD + ks * 32 + g * 8, kh, kl);
            if (ACC) {
              acc = mfma16(kl, qh[ks], acc);
              acc = mfma16(kh, ql[ks], acc);
            }
            acc = mfma16(kh, qh[ks], acc);
          }
        }
        s[kt2][hh] = acc;
      }
    }
    const int qpos = p.t_past + t;   // absolute frame index of this lane's query
    const int causal = p.causal;
    const float sum = softmax_tiles<ACC, MAXNT2>(s, nt2, g, p.scale, 0,
                                                 [&](int, int key) { return key < Tk && (!causal || key <= qpos); });
    const float inv = 1.0f / sum;

    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt2 = 0; kt2 < MAXNT2; ++kt2) {
      if (kt2 < nt2) {
        bf16x8_t ph, pl;
        pack_p<ACC>(s[kt2][0], s[kt2][1], ph, pl);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int off = (dt * 16 + l15) * vpitch + (kt2 * 32 + g * 8) * 2;
          const bf16x8_t vh = *reinterpret_cast<const bf16x8_t*>(v_hi + off);
          if (ACC) {
            const bf16x8_t vl = *reinterpret_cast<const bf16x8_t*>(v_lo + off);
            o[dt] = mfma16(vl, ph, o[dt]);
            o[dt] = mfma16(vh, pl, o[dt]);
          }
          o[dt] = mfma16(vh, ph, o[dt]);
        }
      }
    }
    // ---- context: lane holds d = dt*16 + g*4..+4 of query l15 -> wave patch -> 128-byte row segments
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      unsigned int hb[4], lb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split_bf(o[dt][j] * inv, hb[j], lb[j]);
      const int off = l15 * 128 + (((dt * 2 + (g >> 1)) ^ (l15 & 7)) << 4) + (g & 1) * 8;
      *reinterpret_cast<u32x2_t*>(o_st + off) = (u32x2_t){hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16)};
      if (ACC) *reinterpret_cast<u32x2_t*>(o_st + 2048 + off) = (u32x2_t){lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16)};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = it * 64 + lane;
      const int r = idx >> 3, c = idx & 7;
      const int tq = qt * 16 + r;
      if (tq < Tq) {
        const size_t o_off = (((size_t)b * Tq + tq) * p.N + n) * p.D + h * HD + c * 8;
        const int off = r * 128 + ((c ^ (r & 7)) << 4);
        *reinterpret_cast<u32x4_t*>(p.ctx_hi + o_off) = *reinterpret_cast<const u32x4_t*>(o_st + off);
        if (ACC) *reinterpret_cast<u32x4_t*>(p.ctx_lo + o_off) = *reinterpret_cast<const u32x4_t*>(o_st + 2048 + off);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ================================================================================================
// temporal attention, bf16 mode, short sequences (Tq <= 16 queries against Tk <= 32 cached frames: every full 16-frame
// clip): the round-2 staging of the spatial kernel applied to the per-(patch, head) problem.  One wave per task; K and
// V rows (one 128-byte line per frame) go global -> LDS by LDS-DMA as row-major images, K fragments are plain row reads,
// V^T fragments transposed reads (ds_read_b64_tr_b16), natural key order, P stays in registers.  The MFMA kernel above
// keeps the long caches, the accurate mode and T_new > 16.
// ================================================================================================
// ACC: hi + lo bf16 planes of q / k / v (the accurate mode's full-clip forward writes them instead of fp32), four images per
// wave, three MFMAs per product, probabilities and context split into hi + lo in registers.
template <bool ACC>
__global__ __launch_bounds__(256) void sf_temporal_attn_dma_kernel(SfAttnArgs p, int ntasks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int task = blockIdx.x * 4 + wave;
  if (task >= ntasks) return;
  const int h = task % p.heads, bn = task / p.heads;
  const int b = bn / p.N, n = bn % p.N;
  const int Tk = p.Tk, Tq = p.Tq;
  char* k_img = smem + wave * (ACC ? 20480 : 10240);          // [32 rows][128 B]
  char* v_img = k_img + 4096;                 // [32 rows][128 B]
  char* o_st = v_img + 4096;                  // [16 rows][128 B] (hi; ACC: lo at + 2048 ... see below)
  char* kl_img = o_st + (ACC ? 4096 : 2048);  // ACC only
  char* vl_img = kl_img + 4096;
  const bf16_t* kbase = reinterpret_cast<const bf16_t*>(p.k) + h * HD;
  const bf16_t* vbase = reinterpret_cast<const bf16_t*>(p.v) + h * HD;
#pragma unroll
  for (int j = 0; j < 4; ++j) {               // instruction j covers frames 8j .. 8j+7
    const int row = j * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ sp_bswz(row);
    const int key = row < Tk ? row : Tk - 1;  // padding rows repeat the last frame (masked below)
    const size_t src = (((size_t)b * p.Tcap + key) * p.N + n) * (size_t)p.row_pitch_kv + chunk * 8;
    __builtin_amdgcn_global_load_lds((sp_gptr_t)(kbase + src), (sp_lptr_t)(k_img + j * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((sp_gptr_t)(vbase + src), (sp_lptr_t)(v_img + j * 1024), 16, 0, 0);
    if (ACC) {
      __builtin_amdgcn_global_load_lds((sp_gptr_t)(kbase + p.lo_plane_off + src), (sp_lptr_t)(kl_img + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((sp_gptr_t)(vbase + p.lo_plane_off + src), (sp_lptr_t)(vl_img + j * 1024), 16, 0, 0);
    }
  }
  bf16x8_t qf[2], ql[2];
  {
    const int t = l15 < Tq ? l15 : Tq - 1;
    const size_t qrow = ((size_t)b * p.Tq_cap + p.q_t0 + t) * p.N + n;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16_t* qp = reinterpret_cast<const bf16_t*>(p.q) + qrow * p.row_pitch_q + h * HD + ks * 32 + g * 8;
      qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp);
      if (ACC) ql[ks] = *reinterpret_cast<const bf16x8_t*>(qp + p.lo_plane_off);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();            // the images are wave-private: no workgroup barrier
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // S^T = K Q^T: lane (query l15, g) holds keys 16 jt + 4 g + r
  f32x4_t s[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t kh = sp_row_frag(k_img, jt * 16 + l15, ks * 4 + g);
      if (ACC) {
        acc = mfma16(sp_row_frag(kl_img, jt * 16 + l15, ks * 4 + g), qf[ks], acc);
        acc = mfma16(kh, ql[ks], acc);
      }
      acc = mfma16(kh, qf[ks], acc);
    }
    s[jt] = acc;
  }
  __builtin_amdgcn_sched_barrier(0);          // mask / maximum only after both tiles' MFMAs (see the spatial kernel)
  const int qpos = p.t_past + l15;            // absolute frame of this lane's query
  const int causal = p.causal;
  float mx = -INFINITY;
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = jt * 16 + 4 * g + r;
      const bool ok = key < Tk && (!causal || key <= qpos);
      s[jt][r] = ok ? s[jt][r] : -INFINITY;
      mx = fmaxf(mx, s[jt][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float c2 = p.scale * 1.44269504088896340736f;
  const float mc = mx * c2;
  float sum = 0.f;
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __builtin_amdgcn_exp2f(fmaf(s[jt][r], c2, -mc));
      s[jt][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  if (p.drop.on) {      // training: dropout on the (normalised) probabilities (modeling:603) — the denominator above is unmasked
    const unsigned dbase = (unsigned)((((size_t)bn * p.heads + h) * Tq + (l15 < Tq ? l15 : Tq - 1)) * Tk);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[jt][r] *= sf_drop_factor(p.drop, dbase + (unsigned)(jt * 16 + 4 * g + r));
  }
  const u32x4_t pu = {pack_bf2(s[0][0], s[0][1]), pack_bf2(s[0][2], s[0][3]), pack_bf2(s[1][0], s[1][1]), pack_bf2(s[1][2], s[1][3])};
  const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pu);
  bf16x8_t pl;
  if (ACC) {
    u32x4_t lu;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const f32x4_t& sv = s[w >> 1];
      const int r0 = (w & 1) * 2;
      lu[w] = pack_bf2(sv[r0] - bf2f(pu[w] & 0xffffu), sv[r0 + 1] - bf2f(pu[w] >> 16));
    }
    pl = __builtin_bit_cast(bf16x8_t, lu);
  }
  f32x4_t o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const bf16x8_t vh = sp_tr_frag(v_img, 0, dt, lane);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if (ACC) {
      acc = mfma16(sp_tr_frag(vl_img, 0, dt, lane), pf, acc);
      acc = mfma16(vh, pl, acc);
    }
    o[dt] = mfma16(vh, pf, acc);
  }
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const int off = l15 * 128 + (((dt * 2 + (g >> 1)) ^ (l15 & 7)) << 4) + (g & 1) * 8;
    const f32x4_t ov = o[dt] * inv;
    const u32x2_t hv = {pack_bf2(ov[0], ov[1]), pack_bf2(ov[2], ov[3])};
    *reinterpret_cast<u32x2_t*>(o_st + off) = hv;
    if (ACC)
      *reinterpret_cast<u32x2_t*>(o_st + 2048 + off) = (u32x2_t){pack_bf2(ov[0] - bf2f(hv[0] & 0xffffu), ov[1] - bf2f(hv[0] >> 16)),
                                                                pack_bf2(ov[2] - bf2f(hv[1] & 0xffffu), ov[3] - bf2f(hv[1] >> 16))};
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = it * 64 + lane;
    const int r = idx >> 3, c = idx & 7;
    if (r < Tq) {
      const size_t oo = (((size_t)b * Tq + r) * p.N + n) * p.D + h * HD + c * 8;
      *reinterpret_cast<u32x4_t*>(p.ctx_hi + oo) = *reinterpret_cast<const u32x4_t*>(o_st + r * 128 + ((c ^ (r & 7)) << 4));
      if (ACC) *reinterpret_cast<u32x4_t*>(p.ctx_lo + oo) = *reinterpret_cast<const u32x4_t*>(o_st + 2048 + r * 128 + ((c ^ (r & 7)) << 4));
    }
  }
}

// ================================================================================================
// temporal attention of ONE new frame per stream (Tq = 1: the streaming step, vqa_enc:1316-1392 with a
// single-frame call): a matrix-vector problem per (b, patch, head) — q (64) against Tk cached keys — so there
// is nothing for the MFMA to do.  One wave per task, plain VALU:
//   scores : lane = key (KP passes of 64 keys); the lane reads its key row (128 B bf16 / 256 B fp32) and dots it
//            with q in fp32 (exact products of the stored operands; the fp32-accurate mode needs no bf16x3 here);
//   softmax: two wave reductions (max, sum) on the DPP path;
//   P V    : lane = (key mod 8, 8-wide d chunk): eight keys x 128 B per load instruction (whole lines), the lane's
//            probability comes by ds_bpermute, partial rows meet by three xor-shuffles per value.
// Every load of the task is issued before the first use (one memory latency per task), no LDS, no barrier.
// The MFMA kernel above pays ~10-17 us per launch at one frame (V^T staging through LDS, 16-query tiles with one
// live query); this one is bound by the cache read: Tk x 196 x 768 x 2 x 2 B.
// ================================================================================================
SF_DEVICE float wave_max_dpp(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

template <bool F32, int KP>
__global__ __launch_bounds__(256) void sf_temporal_decode_kernel(SfAttnArgs p, int ntasks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int task = blockIdx.x * 4 + wave;
  if (task >= ntasks) return;
  const int h = task % p.heads, bn = task / p.heads;
  const int b = bn / p.N, n = bn % p.N;
  // the cache position by value, or (streamed frame replayed from the position-free graph) from device memory
  const int q_t0 = p.pos_dev ? p.pos_dev[0] : p.q_t0;
  const int Tk = p.pos_dev ? min(p.pos_dev[1], KP * 64) : p.Tk;
  const int t_past = p.pos_dev ? Tk - 1 : p.t_past;         // absolute index of the query among the keys (causal: keys <= it)
  constexpr int ESZ = F32 ? 4 : 2;
  constexpr int QL = F32 ? 16 : 8;                 // 16-byte loads per 64-dim row
  const char* qb = reinterpret_cast<const char*>(p.q);
  const char* kb = reinterpret_cast<const char*>(p.k);
  const char* vb = reinterpret_cast<const char*>(p.v);
  const size_t qoff = ((((size_t)b * p.Tq_cap + q_t0) * p.N + n) * p.row_pitch_q + h * HD) * ESZ;

  // ---- issue: q (same 128 / 256 bytes for every lane), this lane's key rows, this lane's V chunks -------------------
  u32x4_t qv[QL], kv[KP][QL], vv[KP][8][F32 ? 2 : 1];
#pragma unroll
  for (int c = 0; c < QL; ++c) qv[c] = *reinterpret_cast<const u32x4_t*>(qb + qoff + c * 16);
#pragma unroll
  for (int kp = 0; kp < KP; ++kp) {
    int key = kp * 64 + lane;
    key = key < Tk ? key : Tk - 1;                 // clamped rows are masked below
    const size_t off = ((((size_t)b * p.Tcap + key) * p.N + n) * p.row_pitch_kv + h * HD) * ESZ;
#pragma unroll
    for (int c = 0; c < QL; ++c) kv[kp][c] = *reinterpret_cast<const u32x4_t*>(kb + off + c * 16);
  }
  const int tsub = lane >> 3, ch = lane & 7;       // P V layout: key = 8 i + tsub, d = 8 ch .. 8 ch + 7
  // fp32 rows with four key passes (the pooling head of the accurate mode: 196 keys): K alone is 256 VGPRs per lane, so the
  // V chunks are requested after the scores have consumed the K rows (two latencies per task instead of 100 spilled VGPRs)
  constexpr bool V_LATE = F32 && KP >= 4;
  auto load_v = [&]() {
#pragma unroll
    for (int kp = 0; kp < KP; ++kp)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int key = kp * 64 + i * 8 + tsub;
        key = key < Tk ? key : Tk - 1;
        const size_t off = ((((size_t)b * p.Tcap + key) * p.N + n) * p.row_pitch_kv + h * HD + ch * 8) * ESZ;
        vv[kp][i][0] = *reinterpret_cast<const u32x4_t*>(vb + off);
        if (F32) vv[kp][i][F32 ? 1 : 0] = *reinterpret_cast<const u32x4_t*>(vb + off + 16);
      }
  };
  if (!V_LATE) load_v();
  __builtin_amdgcn_sched_barrier(0);     // keep every load of the task ahead of the arithmetic (one latency per task)

  // ---- scores -----------------------------------------------------------------------------------------------------
  const int qpos = t_past;                         // absolute frame index of the one query
  float sc[KP];
  float mx = -INFINITY;
#pragma unroll
  for (int kp = 0; kp < KP; ++kp) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int c = 0; c < QL; ++c) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (F32) {
          a0 = fmaf(__uint_as_float(qv[c][j]), __uint_as_float(kv[kp][c][j]), a0);
        } else {
          a0 = fmaf(bf2f(qv[c][j] & 0xffffu), bf2f(kv[kp][c][j] & 0xffffu), a0);
          a1 = fmaf(__uint_as_float(qv[c][j] & 0xffff0000u), __uint_as_float(kv[kp][c][j] & 0xffff0000u), a1);
        }
      }
    }
    const int key = kp * 64 + lane;
    const bool ok = key < Tk && (!p.causal || key <= qpos);
    sc[kp] = ok ? (a0 + a1) : -INFINITY;
    mx = fmaxf(mx, sc[kp]);
  }
  if (V_LATE) {
    __builtin_amdgcn_sched_barrier(0);
    load_v();
    __builtin_amdgcn_sched_barrier(0);
  }
  mx = wave_max_dpp(mx);
  const float c2 = p.scale * 1.44269504088896340736f;
  float pr[KP], sum = 0.f;
#pragma unroll
  for (int kp = 0; kp < KP; ++kp) {
    const float a = (sc[kp] - mx) * c2;
    pr[kp] = __builtin_amdgcn_exp2f(a);       // masked: 2^-inf = 0
    sum += pr[kp];
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;

  // ---- P V ----------------------------------------------------------------------------------------------------------
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
  for (int kp = 0; kp < KP; ++kp)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float pt = __shfl(pr[kp], i * 8 + tsub, 64);          // probability of key kp*64 + 8 i + tsub
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (F32) {
          o[j] = fmaf(pt, __uint_as_float(vv[kp][i][0][j]), o[j]);
          o[4 + j] = fmaf(pt, __uint_as_float(vv[kp][i][F32 ? 1 : 0][j]), o[4 + j]);
        } else {
          o[2 * j] = fmaf(pt, bf2f(vv[kp][i][0][j] & 0xffffu), o[2 * j]);
          o[2 * j + 1] = fmaf(pt, __uint_as_float(vv[kp][i][0][j] & 0xffff0000u), o[2 * j + 1]);
        }
      }
    }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    o[j] += __shfl_xor(o[j], 8, 64);
    o[j] += __shfl_xor(o[j], 16, 64);
    o[j] += __shfl_xor(o[j], 32, 64);
  }
  if (tsub == 0) {
    const size_t o_off = (((size_t)b * p.Tq) * p.N + n) * p.D + h * HD + ch * 8;
    unsigned int hb[8], lb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split_bf(o[j] * inv, hb[j], lb[j]);
    *reinterpret_cast<u32x4_t*>(p.ctx_hi + o_off) = (u32x4_t){hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16)};
    if (F32 || p.ctx_lo) *reinterpret_cast<u32x4_t*>(p.ctx_lo + o_off) = (u32x4_t){lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16), lb[4] | (lb[5] << 16), lb[6] | (lb[7] << 16)};
  }
}

// The same problem with whole cache lines per load instruction (bf16 rows, <= 128 keys).  A key row of one head is ONE 128-byte
// line; in the kernel above a lane owns a key, so every K load instruction touches 64 lines for 16 bytes each (and is issued
// eight times over the same lines): the vector-memory path serves 8x the lines the data needs.  Here lane = (key mod 8, 16-byte
// chunk) for K exactly as for V: an instruction reads 8 keys x 128 B; the lane dots its 8 dims with its 16 bytes of q (one
// load instead of eight), the 8 chunks of a key meet by three DPP adds (quad_perm x 2, row_half_mirror: bit-identical in all 8
// lanes), so the lane that holds V[key][chunk] already holds p[key] (no ds_bpermute for the probabilities); max / sum / output
// cross the 8 key groups by one DPP row rotate + two xor-shuffles instead of six dependent shuffles each.
template <int CTRL>
SF_DEVICE float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int KP>
__global__ __launch_bounds__(256) void sf_temporal_decode_lines_kernel(SfAttnArgs p, int ntasks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int task = blockIdx.x * 4 + wave;
  if (task >= ntasks) return;
  const int h = task % p.heads, bn = task / p.heads;
  const int b = bn / p.N, n = bn % p.N;
  int q_t0 = p.q_t0, Tk = p.Tk, t_past = p.t_past;
  if (p.pos_dev) {                                 // streamed frame replayed from the position-free graph: {slot, keys} in one scalar load
    struct __attribute__((aligned(4))) SlotKeys { int slot, tk; };
    const SlotKeys sk = *reinterpret_cast<const SlotKeys*>(p.pos_dev);
    q_t0 = sk.slot;
    Tk = min(sk.tk, KP * 64);
    t_past = Tk - 1;
  }
  const int tsub = lane >> 3, ch = lane & 7;       // key = 64 kp + 8 i + tsub, dims 8 ch .. 8 ch + 7
  const char* kb = reinterpret_cast<const char*>(p.k);
  const char* vb = reinterpret_cast<const char*>(p.v);
  const size_t qoff = ((((size_t)b * p.Tq_cap + q_t0) * p.N + n) * p.row_pitch_q + h * HD + ch * 8) * 2;
  const u32x4_t qv = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(p.q) + qoff);
  u32x4_t kv[KP][8], vv[KP][8];
#pragma unroll
  for (int kp = 0; kp < KP; ++kp)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int key = kp * 64 + i * 8 + tsub;
      key = key < Tk ? key : Tk - 1;               // clamped rows (one line, already in the cache) are masked below
      const size_t off = ((((size_t)b * p.Tcap + key) * p.N + n) * p.row_pitch_kv + h * HD + ch * 8) * 2;
      kv[kp][i] = *reinterpret_cast<const u32x4_t*>(kb + off);
      vv[kp][i] = *reinterpret_cast<const u32x4_t*>(vb + off);
    }
  __builtin_amdgcn_sched_barrier(0);     // every load of the task ahead of the arithmetic (one latency per task)

  float qf[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    qf[2 * j] = bf2f(qv[j] & 0xffffu);
    qf[2 * j + 1] = __uint_as_float(qv[j] & 0xffff0000u);
  }
  float sc[KP][8];
  float mx = -INFINITY;
#pragma unroll
  for (int kp = 0; kp < KP; ++kp)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0 = fmaf(qf[2 * j], bf2f(kv[kp][i][j] & 0xffffu), a0);
        a1 = fmaf(qf[2 * j + 1], __uint_as_float(kv[kp][i][j] & 0xffff0000u), a1);
      }
      float a = a0 + a1;
      a += dpp_move<0xB1>(a);                      // quad_perm [1,0,3,2]
      a += dpp_move<0x4E>(a);                      // quad_perm [2,3,0,1]
      a += dpp_move<0x141>(a);                     // row_half_mirror: the other quad of the 8 chunk lanes
      const int key = kp * 64 + i * 8 + tsub;
      const bool ok = key < Tk && (!p.causal || key <= t_past);
      sc[kp][i] = ok ? a : -INFINITY;
      mx = fmaxf(mx, sc[kp][i]);
    }
  mx = fmaxf(mx, dpp_move<0x128>(mx));             // row_ror:8 : key group tsub ^ 1
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float c2 = p.scale * 1.44269504088896340736f;
  float sum = 0.f;
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
  for (int kp = 0; kp < KP; ++kp)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float pt = __builtin_amdgcn_exp2f((sc[kp][i] - mx) * c2);        // masked: 2^-inf = 0
      sum += pt;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[2 * j] = fmaf(pt, bf2f(vv[kp][i][j] & 0xffffu), o[2 * j]);
        o[2 * j + 1] = fmaf(pt, __uint_as_float(vv[kp][i][j] & 0xffff0000u), o[2 * j + 1]);
      }
    }
  sum += dpp_move<0x128>(sum);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] += dpp_move<0x128>(o[j]);
  sum += __shfl_xor(sum, 16, 64);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] += __shfl_xor(o[j], 16, 64);
  sum += __shfl_xor(sum, 32, 64);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] += __shfl_xor(o[j], 32, 64);
  if (tsub == 0) {
    const float inv = 1.0f / sum;
    const size_t o_off = (((size_t)b * p.Tq) * p.N + n) * p.D + h * HD + ch * 8;
    unsigned int hb[8], lb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split_bf(o[j] * inv, hb[j], lb[j]);
    *reinterpret_cast<u32x4_t*>(p.ctx_hi + o_off) = (u32x4_t){hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16)};
    if (p.ctx_lo) *reinterpret_cast<u32x4_t*>(p.ctx_lo + o_off) = (u32x4_t){lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16), lb[4] | (lb[5] << 16), lb[6] | (lb[7] << 16)};
  }
}

// accurate mode: may the caller hand q / k / v of the temporal attention as hi + lo bf16 planes (short clips, no cache)?
bool sf_temporal_planes_ok(int Tq, int Tk) {
  const bool off = sf_sw(SW_DISABLE_TEMPORAL_DMA_ACC) != nullptr || sf_sw(SW_DISABLE_TEMPORAL_DMA) != nullptr;
  return !off && Tq > 1 && Tq <= 16 && Tk <= 32;
}

hipError_t sf_launch_temporal_attention(const SfAttnArgs& a, bool accurate, hipStream_t s) {
  if (a.head_dim && a.head_dim != HD) return sf_launch_attention_generic(a, a.head_dim, true, s);       // sf_attention_generic.hip
  if (a.D != a.heads * HD || a.Tq <= 0 || a.Tk <= 0 || a.B <= 0 || a.N <= 0) return hipErrorInvalidValue;
  // dropout on the probabilities (training forward): the DMA-staged whole-clip kernel only
  if (a.drop.on && (accurate || a.Tq == 1 || a.Tq > 16 || a.Tk > 32 || (a.row_pitch_kv % 8) || sf_sw(SW_DISABLE_TEMPORAL_DMA))) return hipErrorInvalidValue;
  const bool decode_off = sf_sw(SW_DISABLE_TEMPORAL_DECODE) != nullptr;
  if (a.Tq == 1 && a.Tk <= 256 && !decode_off) {        // one new frame per stream: the matrix-vector kernel
    const int ntasks = a.B * a.N * a.heads;
    const dim3 grid((ntasks + 3) / 4), block(256);
    const int kp = (a.Tk + 63) >> 6;
#define SF_TD(F, KPV) hipLaunchKernelGGL((sf_temporal_decode_kernel<F, KPV>), grid, block, 0, s, a, ntasks)
    if (accurate) { if (kp <= 1) SF_TD(true, 1); else if (kp <= 2) SF_TD(true, 2); else SF_TD(true, 4); }
    else if (kp <= 2 && (a.row_pitch_kv % 8) == 0 && (a.row_pitch_q % 8) == 0 && !sf_sw(SW_TEMPORAL_DECODE_LANE_KEY)) {
      // bf16 rows, <= 128 keys: whole cache lines per load instruction
      if (kp <= 1) hipLaunchKernelGGL(sf_temporal_decode_lines_kernel<1>, grid, block, 0, s, a, ntasks);
      else hipLaunchKernelGGL(sf_temporal_decode_lines_kernel<2>, grid, block, 0, s, a, ntasks);
    }
    else { if (kp <= 1) SF_TD(false, 1); else if (kp <= 2) SF_TD(false, 2); else SF_TD(false, 4); }
#undef SF_TD
    return hipGetLastError();
  }
  const bool tdma_off = sf_sw(SW_DISABLE_TEMPORAL_DMA) != nullptr;
  if (!accurate && !tdma_off && a.Tq <= 16 && a.Tk <= 32 && (a.row_pitch_kv % 8) == 0) {     // every full 16-frame clip
    const int ntasks = a.B * a.N * a.heads;
    hipLaunchKernelGGL(sf_temporal_attn_dma_kernel<false>, dim3((ntasks + 3) / 4), dim3(256), 4 * 10240, s, a, ntasks);
    return hipGetLastError();
  }
  if (accurate && !a.in_is_f32) {        // hi + lo planes (sf_temporal_planes_ok)
    if (a.Tq > 16 || a.Tk > 32 || a.lo_plane_off <= 0 || (a.row_pitch_kv % 8) || (a.lo_plane_off % 8)) return hipErrorInvalidValue;
    const int ntasks = a.B * a.N * a.heads;
    static SfPerDeviceOnce attr_t;
    if (attr_t.first())
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_temporal_attn_dma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 20480);
    hipLaunchKernelGGL(sf_temporal_attn_dma_kernel<true>, dim3((ntasks + 3) / 4), dim3(256), 4 * 20480, s, a, ntasks);
    return hipGetLastError();
  }
  const int tkp = (a.Tk + 31) & ~31;
  if (tkp > 32 * 8) return hipErrorInvalidValue;   // <= 256 cached frames per stream
  const int vp = vt_pitch(tkp);
  const size_t patch = (size_t)HD * vp * (accurate ? 2 : 1) + (accurate ? 4096 : 2048);
  int waves = 4;
  while (waves > 1 && patch * waves > 150 * 1024) waves >>= 1;
  const size_t lds = patch * waves;
  const int ntasks = a.B * a.N * a.heads;
  const dim3 grid((ntasks + waves - 1) / waves), block(waves * 64);
#define SF_TL(ACCV, NT)                                                                              \
  do {                                                                                               \
    static SfPerDeviceOnce attr;                                                                        \
    if (attr.first()) {                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_temporal_attn_kernel<ACCV, NT>),   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);              \
    }                                                                                                \
    hipLaunchKernelGGL((sf_temporal_attn_kernel<ACCV, NT>), grid, block, lds, s, a, vp, ntasks);     \
  } while (0)
  const int nt2 = tkp >> 5;
  if (accurate) {
    if (nt2 <= 1) SF_TL(true, 1); else if (nt2 <= 2) SF_TL(true, 2); else if (nt2 <= 4) SF_TL(true, 4); else SF_TL(true, 8);
  } else {
    if (nt2 <= 1) SF_TL(false, 1); else if (nt2 <= 2) SF_TL(false, 2); else if (nt2 <= 4) SF_TL(false, 4); else SF_TL(false, 8);
  }
#undef SF_TL
  return hipGetLastError();
}
