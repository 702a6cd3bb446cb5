// Attention backward for the training step (torch autograd through modeling_timesformer_siglip.py:590-612
// temporal causal attention, :703-714 spatial attention, :1145-1149 pooling-head attention).
//
//   S = scale * Q K^T (+ causal mask),  P = softmax_row(S),  O = P V
//   dV = P^T dO,  dP = dO V^T,  Delta_i = sum_e dO_ie O_ie,  dS = P o (dP - Delta) * scale
//   dQ = dS K,  dK = dS^T Q
//
// gfx950 design: the four row-major [tokens][64] bf16 images (Q, K, V, dO) of one (sequence, head)
// problem live in LDS (XOR-swizzled 16-byte chunks).  Products that contract over the head dim read
// row fragments (ds_read_b128); products that contract over tokens (dV, dK, dQ) need a token-major
// B operand, which ds_read_b64_tr_b16 delivers straight from the same row-major images — no
// transposed copies.  The probabilities never touch LDS: the 16x16 MFMA result layout (lane = one
// column, 4 consecutive rows) of two neighbouring tiles is exactly the 8-value A operand of the next
// product, in a k order that the transposed reads reproduce.
//   phase A: row statistics (log-sum-exp in base 2) and Delta             — waves own query tiles
//   phase B: dK, dV                                                        — waves own 32-key blocks
//   phase C: dQ (scores recomputed in the swapped orientation)             — waves own 32-query blocks
// No atomics: every output element has one owner, results are bit-reproducible.
// Spatial: one 8-wave workgroup per (frame, head), L <= 224.  Temporal: one WAVE per (batch, patch,
// head) sequence with wave-private images, L <= 32.
#include "sf_train.h"
#include "sf_switches.h"
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

#define LOG2E 1.4426950408889634f
#define NEG_BIG (-1.0e30f)

SF_DEVICE f32x4_t ab_mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}
// chunk swizzle: bijective in row bits 1..3 (row fragments of 16 rows conflict-free), and its upper
// two bits bijective in row bits 1..2 (the 8 rows of a half-wave transposed read conflict-free)
// (round 6) gfx950 serves a ds_read_b128 in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32), not in contiguous sixteens: a row
// fragment's group holds rows 0-3 and 12-15 at chunk c and rows 4-11 at chunk c ^ 1.  With 128-byte rows the even rows share one half of the
// 256-byte bank row, so {s(0), s(2), s(12), s(14)} and {s(4), s(6), s(8), s(10)} ^ 1 must partition the 8 slots: s = 2 * ((row >> 1) & 3)
// gives {0, 2, 4, 6} and {5, 7, 1, 3}.  Rounds 2-5 also folded row bit 3 in (| ((row >> 3) & 1)): bijective over 16 contiguous rows, 2-way
// conflicted on the real groups (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.24-0.28 on these kernels).  SF_ATTN_SWZ_LEGACY: the old function (A/B builds).
#ifdef SF_ATTN_SWZ_LEGACY
SF_DEVICE int bswz(int row) { return (((row >> 1) & 3) << 1) | ((row >> 3) & 1); }
#else
SF_DEVICE int bswz(int row) { return ((row >> 1) & 3) << 1; }
#endif
SF_DEVICE int img_off(int row, int chunk) { return row * 128 + ((chunk ^ bswz(row)) << 4); }

SF_DEVICE bf16x8_t row_frag(const char* img, int row, int chunk) {
  return *reinterpret_cast<const bf16x8_t*>(img + img_off(row, chunk));
}
#ifdef SF_LAB
__device__ int g_tbwd_plain_reads;       // lab (SF_TBWD_LAB=6): plain 8-byte reads in place of the transposed ones (results invalid)
#endif
// token-major fragment: lane (l15 = column e of tile et, g) gets rows {r0+4g..+3} and {r0+16+4g..+3}
template <bool TWO>
SF_DEVICE bf16x8_t tr_frag(const char* img, int r0, int et, int lane) {
  const int t16 = lane & 15, g = lane >> 4;
  const int row = r0 + 4 * g + (t16 >> 2);
  const int off = img_off(row, 2 * et + ((t16 & 3) >> 1)) + ((t16 & 1) << 3);
#ifdef SF_LAB
  if (g_tbwd_plain_reads) {
    const s16x4_t lo = *reinterpret_cast<const s16x4_t*>(img + off);
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    if (TWO) {
      const s16x4_t hi = *reinterpret_cast<const s16x4_t*>(img + off + 16 * 128);
      f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    } else {
      f[4] = f[5] = f[6] = f[7] = 0;
    }
    return f;
  }
#endif
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(img + off));
  bf16x8_t f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  if (TWO) {
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(img + off + 16 * 128));
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  } else {
    f[4] = f[5] = f[6] = f[7] = 0;
  }
  return f;
}
SF_DEVICE bf16x8_t pack_a(const f32x4_t lo, const f32x4_t hi) {
  u32x4_t u = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(hi[0], hi[1]), pack_bf2(hi[2], hi[3])};
  return __builtin_bit_cast(bf16x8_t, u);
}

struct BwdView {
  const char *q, *k, *v, *d_o;     // LDS images
  float* lse2;                     // [rows] base-2 log-sum-exp of the scaled scores
  float* delta;                    // [rows]
  char* patch;                     // wave-private [32][64] bf16 output staging
  int L;
  int causal;
  float sl2;                       // scale * log2(e)
  float scale;
  SfDrop drop;                     // attention-probability dropout of the forward (modeling:556, 603, 669, 705): factor of element
  unsigned drop_base;              //   (query qi, key kj) = sf_drop_factor(drop, drop_base + qi * L + kj); O = (m o P) V, so dV sees m o P
};                                 //   and dS = P o (m o dP - Delta) with Delta = rowsum(dO o O) unchanged
SF_DEVICE float bwd_drop(const BwdView& w, int qi, int kj) { return w.drop.on ? sf_drop_factor(w.drop, w.drop_base + (unsigned)(qi * w.L + kj)) : 1.f; }
// compile-time variant for the register-bound spatial kernel (the plain instance must not pay the mask's registers: +21 spilled VGPRs, 165 -> 206 us)
template <bool DROP>
SF_DEVICE float bwd_drop_t(const BwdView& w, int qi, int kj) { return DROP ? sf_drop_factor(w.drop, w.drop_base + (unsigned)(qi * w.L + kj)) : 1.f; }

// ---- phase A: statistics of query tile `it` (16 queries), swapped scores: lane = query l15 ----------
template <bool TWO>
SF_DEVICE void phase_a_tile(const BwdView& w, int it, int nt, int lane) {
  const int l15 = lane & 15, g = lane >> 4;
  const int qi = it * 16 + l15;
  bf16x8_t qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) qf[ks] = row_frag(w.q, it * 16 + l15, ks * 4 + g);
  float m = NEG_BIG, l = 0.f;
  for (int jt = 0; jt < nt; ++jt) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) acc = ab_mfma(row_frag(w.k, jt * 16 + l15, ks * 4 + g), qf[ks], acc);
    float s[4];
    float tm = NEG_BIG;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kj = jt * 16 + 4 * g + r;
      const bool ok = kj < w.L && !(w.causal && kj > qi);
      s[r] = ok ? acc[r] * w.sl2 : NEG_BIG;
      tm = fmaxf(tm, s[r]);
    }
    const float mn = fmaxf(m, tm);
    float add = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) add += s[r] > 0.5f * NEG_BIG ? __builtin_amdgcn_exp2f(s[r] - mn) : 0.f;
    l = l * __builtin_amdgcn_exp2f(m - mn) + add;
    m = mn;
  }
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {
    const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
    const float mn = fmaxf(m, m2);
    l = l * __builtin_amdgcn_exp2f(m - mn) + l2 * __builtin_amdgcn_exp2f(m2 - mn);
    m = mn;
  }
  if (g == 0) w.lse2[qi] = (qi < w.L && l > 0.f) ? m + __log2f(l) : -NEG_BIG;   // padding queries: p = 0
}

SF_DEVICE void store_patch(char* patch, int row, int col, float v) {
  *reinterpret_cast<bf16_t*>(patch + img_off(row, col >> 3) + ((col & 7) << 1)) = (bf16_t)f2bf(v);
}

// copy the wave's 32-row patch to global rows tok0 + r (r < nrows valid), 64 columns at `dst_col`
SF_DEVICE void patch_to_global(const char* patch, bf16_t* dst, size_t ld, long row_base, long row_step, int tok0, int L,
                               int rows, int lane) {
  for (int c = lane; c < rows * 8; c += 64) {
    const int r = c >> 3, ch = c & 7;
    const int tok = tok0 + r;
    if (tok < L) {
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(patch + img_off(r, ch));
      *reinterpret_cast<u32x4_t*>(dst + (size_t)(row_base + tok * row_step) * ld + ch * 8) = v;
    }
  }
}

// ---- phase B: dK, dV of key block jb (32 keys) -----------------------------------------------------
template <bool TWO>
SF_DEVICE void phase_b_block(const BwdView& w, int jb, int nb, f32x4_t (&dk)[2][4], f32x4_t (&dv)[2][4], int lane) {
  constexpr int NT2 = TWO ? 2 : 1;
  const int l15 = lane & 15, g = lane >> 4;
  bf16x8_t kf[2][2], vf[2][2];
#pragma unroll
  for (int jt2 = 0; jt2 < NT2; ++jt2)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[jt2][ks] = row_frag(w.k, jb * 32 + jt2 * 16 + l15, ks * 4 + g);
      vf[jt2][ks] = row_frag(w.v, jb * 32 + jt2 * 16 + l15, ks * 4 + g);
    }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) dk[a][b] = dv[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int ib0 = w.causal ? jb : 0;       // queries before the key block see none of its keys
  for (int ib = ib0; ib < nb; ++ib) {
    f32x4_t p[2][2], ds[2][2];             // [query tile][key tile]
#pragma unroll
    for (int it2 = 0; it2 < 2; ++it2) {
      if (it2 < NT2) {
        const int q0 = ib * 32 + it2 * 16;
        bf16x8_t qf[2], gf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          qf[ks] = row_frag(w.q, q0 + l15, ks * 4 + g);
          gf[ks] = row_frag(w.d_o, q0 + l15, ks * 4 + g);
        }
        const f32x4_t lse = *reinterpret_cast<const f32x4_t*>(w.lse2 + q0 + 4 * g);
        const f32x4_t dl = *reinterpret_cast<const f32x4_t*>(w.delta + q0 + 4 * g);
#pragma unroll
        for (int jt2 = 0; jt2 < 2; ++jt2) {
          if (jt2 < NT2) {
            f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              s = ab_mfma(qf[ks], kf[jt2][ks], s);
              dp = ab_mfma(gf[ks], vf[jt2][ks], dp);
            }
            const int kj = jb * 32 + jt2 * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int qi = q0 + 4 * g + r;
              const bool ok = kj < w.L && !(w.causal && kj > qi);
              const float pv = __builtin_amdgcn_exp2f(ok ? (s[r] * w.sl2 - lse[r]) : -INFINITY);   // mask the ARGUMENT: `ok ? exp2f() : 0` compiles to one exec-masked branch per element
              const float fd = bwd_drop(w, qi, kj);
              p[it2][jt2][r] = pv * fd;
              ds[it2][jt2][r] = pv * (fd * dp[r] - dl[r]) * w.scale;
            }
          } else {
            p[it2][jt2] = ds[it2][jt2] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          }
        }
      } else {
        p[it2][0] = p[it2][1] = ds[it2][0] = ds[it2][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int et = 0; et < 4; ++et) {
      const bf16x8_t gt = tr_frag<TWO>(w.d_o, ib * 32, et, lane);
      const bf16x8_t qt = tr_frag<TWO>(w.q, ib * 32, et, lane);
#pragma unroll
      for (int jt2 = 0; jt2 < NT2; ++jt2) {
        dv[jt2][et] = ab_mfma(pack_a(p[0][jt2], p[1][jt2]), gt, dv[jt2][et]);
        dk[jt2][et] = ab_mfma(pack_a(ds[0][jt2], ds[1][jt2]), qt, dk[jt2][et]);
      }
    }
  }
}

// ---- phase C: dQ of query block ib (32 queries), swapped scores --------------------------------------
template <bool TWO>
SF_DEVICE void phase_c_block(const BwdView& w, int ib, int nb, f32x4_t (&dq)[2][4], int lane) {
  constexpr int NT2 = TWO ? 2 : 1;
  const int l15 = lane & 15, g = lane >> 4;
  bf16x8_t qf[2][2], gf[2][2];
  float lse[2], dl[2];
#pragma unroll
  for (int it2 = 0; it2 < NT2; ++it2) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[it2][ks] = row_frag(w.q, ib * 32 + it2 * 16 + l15, ks * 4 + g);
      gf[it2][ks] = row_frag(w.d_o, ib * 32 + it2 * 16 + l15, ks * 4 + g);
    }
    lse[it2] = w.lse2[ib * 32 + it2 * 16 + l15];
    dl[it2] = w.delta[ib * 32 + it2 * 16 + l15];
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) dq[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int jb1 = w.causal ? ib + 1 : nb;    // keys after the query block are masked for all its queries
  for (int jb = 0; jb < jb1; ++jb) {
    f32x4_t ds[2][2];                         // [key tile][query tile]
#pragma unroll
    for (int jt2 = 0; jt2 < 2; ++jt2) {
      if (jt2 < NT2) {
        const int k0 = jb * 32 + jt2 * 16;
        bf16x8_t kf[2], vf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          kf[ks] = row_frag(w.k, k0 + l15, ks * 4 + g);
          vf[ks] = row_frag(w.v, k0 + l15, ks * 4 + g);
        }
#pragma unroll
        for (int it2 = 0; it2 < 2; ++it2) {
          if (it2 < NT2) {
            f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              s = ab_mfma(kf[ks], qf[it2][ks], s);
              dp = ab_mfma(vf[ks], gf[it2][ks], dp);
            }
            const int qi = ib * 32 + it2 * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kj = k0 + 4 * g + r;
              const bool ok = kj < w.L && !(w.causal && kj > qi);
              const float pv = __builtin_amdgcn_exp2f(ok ? (s[r] * w.sl2 - lse[it2]) : -INFINITY);
              ds[jt2][it2][r] = pv * (bwd_drop(w, qi, kj) * dp[r] - dl[it2]) * w.scale;
            }
          } else {
            ds[jt2][it2] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          }
        }
      } else {
        ds[jt2][0] = ds[jt2][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int et = 0; et < 4; ++et) {
      const bf16x8_t kt = tr_frag<TWO>(w.k, jb * 32, et, lane);
#pragma unroll
      for (int it2 = 0; it2 < NT2; ++it2) dq[it2][et] = ab_mfma(pack_a(ds[0][it2], ds[1][it2]), kt, dq[it2][et]);
    }
  }
}

template <bool TWO>
SF_DEVICE void tiles_to_patch(char* patch, const f32x4_t (&t)[2][4], int lane) {
  constexpr int NT2 = TWO ? 2 : 1;
  const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int a = 0; a < NT2; ++a)
#pragma unroll
    for (int et = 0; et < 4; ++et)
#pragma unroll
      for (int r = 0; r < 4; ++r) store_patch(patch, a * 16 + 4 * g + r, et * 16 + l15, t[a][et][r]);
}

// The same patch by 32-bit stores: the two lanes of a column pair exchange the rows the other one writes (even lane: rows 0 and 2 of
// its four, odd lane: rows 1 and 3), so a lane stores two {column, column + 1} words per tile instead of four 16-bit values — half
// the LDS store instructions, no sub-dword LDS writes.  Same conversion (v_cvt_pk_bf16_f32), bit-identical patch.
template <bool TWO>
SF_DEVICE void tiles_to_patch_pairs(char* patch, const f32x4_t (&t)[2][4], int lane) {
  constexpr int NT2 = TWO ? 2 : 1;
  const int l15 = lane & 15, g = lane >> 4;
  const bool odd = l15 & 1;
#pragma unroll
  for (int a = 0; a < NT2; ++a)
#pragma unroll
    for (int et = 0; et < 4; ++et) {
      const f32x4_t v = t[a][et];
      const float r0 = __shfl_xor(odd ? v[0] : v[1], 1, 64), r1 = __shfl_xor(odd ? v[2] : v[3], 1, 64);
      const int col = et * 16 + (l15 & ~1);
      const int row = a * 16 + 4 * g + (odd ? 1 : 0);
      const unsigned w0 = odd ? pack_bf2(r0, v[1]) : pack_bf2(v[0], r0);
      const unsigned w1 = odd ? pack_bf2(r1, v[3]) : pack_bf2(v[2], r1);
      *reinterpret_cast<unsigned*>(patch + img_off(row, col >> 3) + ((col & 7) << 1)) = w0;
      *reinterpret_cast<unsigned*>(patch + img_off(row + 2, col >> 3) + ((col & 7) << 1)) = w1;
    }
}

// All five operand images of a problem in one pass: every global load of the pass is issued before the first LDS write, so a
// wave pays one memory round trip instead of ten (round 3 walked image by image and hipcc kept each loop's load -> wait -> ds_write
// order: the temporal kernel, one wave per sequence with nothing to switch to, measured 78 us for 306 MB).  Token rows t < L from
// global, zeros beyond (clamped for the load, zeroed for the write); Delta_i = sum_e dO_ie * O_ie by 8 lanes per row, xor-shuffle reduce.
template <int IT>
SF_DEVICE void stage_problem(char* iq, char* ik, char* iv, char* ig, float* delta, const bf16_t* qkv, int D, const bf16_t* d_o,
                             const bf16_t* o, size_t ld_qkv, size_t ld_o, long row_base, long row_step, int L, int rows_pad,
                             int tid, int nthreads) {
  u32x4_t q[IT], k[IT], v[IT], g[IT], oo[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c = it * nthreads + tid;
    const int r = c >> 3, ch = c & 7;
    const int rr = r < L ? r : L - 1;
    const size_t row = (size_t)(row_base + rr * row_step);
    const bf16_t* pq = qkv + row * ld_qkv + ch * 8;
    q[it] = *reinterpret_cast<const u32x4_t*>(pq);
    k[it] = *reinterpret_cast<const u32x4_t*>(pq + D);
    v[it] = *reinterpret_cast<const u32x4_t*>(pq + 2 * D);
    g[it] = *reinterpret_cast<const u32x4_t*>(d_o + row * ld_o + ch * 8);
    oo[it] = *reinterpret_cast<const u32x4_t*>(o + row * ld_o + ch * 8);
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c = it * nthreads + tid;
    const int r = c >> 3, ch = c & 7;
    const bool in = c < rows_pad * 8;
    const u32x4_t z = {0u, 0u, 0u, 0u};
    if (r >= L) q[it] = k[it] = v[it] = g[it] = oo[it] = z;
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dot += bf2f(g[it][i] & 0xffffu) * bf2f(oo[it][i] & 0xffffu);
      dot += bf2f(g[it][i] >> 16) * bf2f(oo[it][i] >> 16);
    }
    if (in) {
      const int off = img_off(r, ch);
      *reinterpret_cast<u32x4_t*>(iq + off) = q[it];
      *reinterpret_cast<u32x4_t*>(ik + off) = k[it];
      *reinterpret_cast<u32x4_t*>(iv + off) = v[it];
      *reinterpret_cast<u32x4_t*>(ig + off) = g[it];
    }
    dot += __shfl_xor(dot, 1, 64);
    dot += __shfl_xor(dot, 2, 64);
    dot += __shfl_xor(dot, 4, 64);
    if (in && ch == 0) delta[r] = dot;
  }
}

// ---- 16-row owners (spatial kernel): the same products with one 16-key tile (phase B) or one 16-query tile (phase C) per
// wave, so that 13 of 16 waves (L = 196) work instead of 7 of 8 and every SIMD has four waves to hide the
// LDS -> MFMA -> exp2 -> MFMA chain behind.  The contraction over tokens still runs 32 rows per MFMA.
// NB > 0: the number of 32-row blocks as a compile-time constant (non-causal only) — the block loop unrolls into one
// schedulable region, so the next block's row fragments are in flight under this block's MFMAs (same finding as the
// forward kernel's NTC instance)
template <bool DROP, int NB = 0>
SF_DEVICE void phase_b_tile16(const BwdView& w, int jt, int nb_rt, f32x4_t (&dk)[4], f32x4_t (&dv)[4], int lane) {
  constexpr int kUnroll = NB ? NB : 1;
  const int nb = NB ? NB : nb_rt;
  const bool causal = NB ? false : (w.causal != 0);
  const int l15 = lane & 15, g = lane >> 4;
  bf16x8_t kf[2], vf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    kf[ks] = row_frag(w.k, jt * 16 + l15, ks * 4 + g);
    vf[ks] = row_frag(w.v, jt * 16 + l15, ks * 4 + g);
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) dk[b] = dv[b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int kj = jt * 16 + l15;
  const int ib0 = causal ? (jt >> 1) : 0;
#pragma unroll kUnroll
  for (int ib = ib0; ib < nb; ++ib) {
    f32x4_t p[2], ds[2];                   // [query tile]
#pragma unroll
    for (int it2 = 0; it2 < 2; ++it2) {
      const int q0 = ib * 32 + it2 * 16;
      f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        sc = ab_mfma(row_frag(w.q, q0 + l15, ks * 4 + g), kf[ks], sc);
        dp = ab_mfma(row_frag(w.d_o, q0 + l15, ks * 4 + g), vf[ks], dp);
      }
      const f32x4_t lse = *reinterpret_cast<const f32x4_t*>(w.lse2 + q0 + 4 * g);
      const f32x4_t dl = *reinterpret_cast<const f32x4_t*>(w.delta + q0 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = q0 + 4 * g + r;
        const bool ok = kj < w.L && !(causal && kj > qi);
        const float pv = __builtin_amdgcn_exp2f(ok ? (sc[r] * w.sl2 - lse[r]) : -INFINITY);
        const float fd = bwd_drop_t<DROP>(w, qi, kj);
        p[it2][r] = pv * fd;
        ds[it2][r] = pv * (fd * dp[r] - dl[r]) * w.scale;
      }
    }
    const bf16x8_t pa = pack_a(p[0], p[1]), dsa = pack_a(ds[0], ds[1]);
#pragma unroll
    for (int et = 0; et < 4; ++et) {
      dv[et] = ab_mfma(pa, tr_frag<true>(w.d_o, ib * 32, et, lane), dv[et]);
      dk[et] = ab_mfma(dsa, tr_frag<true>(w.q, ib * 32, et, lane), dk[et]);
    }
  }
}

template <bool DROP, int NB = 0>
SF_DEVICE void phase_c_tile16(const BwdView& w, int it, int nb_rt, f32x4_t (&dq)[4], int lane) {
  constexpr int kUnroll = NB ? NB : 1;
  const int nb = NB ? NB : nb_rt;
  const bool causal = NB ? false : (w.causal != 0);
  const int l15 = lane & 15, g = lane >> 4;
  bf16x8_t qf[2], gf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    qf[ks] = row_frag(w.q, it * 16 + l15, ks * 4 + g);
    gf[ks] = row_frag(w.d_o, it * 16 + l15, ks * 4 + g);
  }
  const int qi = it * 16 + l15;
  const float lse = w.lse2[qi], dl = w.delta[qi];
#pragma unroll
  for (int b = 0; b < 4; ++b) dq[b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int jb1 = causal ? (it >> 1) + 1 : nb;
#pragma unroll kUnroll
  for (int jb = 0; jb < jb1; ++jb) {
    f32x4_t ds[2];                          // [key tile]
#pragma unroll
    for (int jt2 = 0; jt2 < 2; ++jt2) {
      const int k0 = jb * 32 + jt2 * 16;
      f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        sc = ab_mfma(row_frag(w.k, k0 + l15, ks * 4 + g), qf[ks], sc);
        dp = ab_mfma(row_frag(w.v, k0 + l15, ks * 4 + g), gf[ks], dp);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kj = k0 + 4 * g + r;
        const bool ok = kj < w.L && !(causal && kj > qi);
        const float pv = __builtin_amdgcn_exp2f(ok ? (sc[r] * w.sl2 - lse) : -INFINITY);
        ds[jt2][r] = pv * (bwd_drop_t<DROP>(w, qi, kj) * dp[r] - dl) * w.scale;
      }
    }
    const bf16x8_t dsa = pack_a(ds[0], ds[1]);
#pragma unroll
    for (int et = 0; et < 4; ++et) dq[et] = ab_mfma(dsa, tr_frag<true>(w.k, jb * 32, et, lane), dq[et]);
  }
}

SF_DEVICE void tile16_to_patch(char* patch, const f32x4_t (&t)[4], int lane) {
  const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int et = 0; et < 4; ++et)
#pragma unroll
    for (int r = 0; r < 4; ++r) store_patch(patch, 4 * g + r, et * 16 + l15, t[et][r]);
}

// ------------------------------------------------------------------------------------------------
// spatial: one workgroup (16 waves) per (frame, head)
// ------------------------------------------------------------------------------------------------
#define SB_THREADS 1024
#define SB_WAVES 16
#define SB_ROWS 224
#define SB_IMG (SB_ROWS * 128)
#define SB_PATCH 2048

template <bool DROP, int NTC = 0>
__global__ __launch_bounds__(SB_THREADS) void sf_spatial_attn_bwd_kernel(SfAttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
  const int L = a.L;
  constexpr int NB = (NTC + 1) / 2;
  const int nt = NTC ? NTC : (L + 15) >> 4, nb = NTC ? NB : (L + 31) >> 5;
  const int rows_pad = nb * 32;
  char* iq = smem;
  char* ik = smem + SB_IMG;
  char* iv = smem + 2 * SB_IMG;
  char* ig = smem + 3 * SB_IMG;
  float* lse2 = reinterpret_cast<float*>(smem + 4 * SB_IMG);
  float* delta = lse2 + SB_ROWS;
  char* patch = smem + 4 * SB_IMG + 2 * SB_ROWS * 4 + wave * SB_PATCH;

  const long row_base = (long)f * L;
  const bf16_t* qkv = a.qkv + h * 64;
  static_assert(SB_ROWS * 8 <= 2 * SB_THREADS, "two staging passes cover the padded image");
  // the forward's saved row statistics ride with the staging pass (one memory round trip per workgroup, not two); padding queries get +big so that p = 0
  static_assert(SB_ROWS <= SB_THREADS, "one statistics row per thread");
  float lse_saved = -NEG_BIG;
  if (a.lse2 && tid < L) lse_saved = a.lse2[((size_t)f * a.heads + h) * L + tid];
  stage_problem<2>(iq, ik, iv, ig, delta, qkv, a.D, a.d_o + h * 64, a.o + h * 64, a.ld_qkv, a.ld_o, row_base, 1, L, rows_pad, tid, SB_THREADS);
  if (a.lse2 && tid < rows_pad) lse2[tid] = lse_saved;
  __syncthreads();

  BwdView w;
  w.q = iq; w.k = ik; w.v = iv; w.d_o = ig; w.lse2 = lse2; w.delta = delta; w.patch = patch;
  w.L = L; w.causal = a.causal; w.scale = a.scale; w.sl2 = a.scale * LOG2E;
  w.drop = a.drop; w.drop_base = (unsigned)(((size_t)f * a.heads + h) * (size_t)L * L);

  if (!a.lse2) {      // no saved statistics (the op entry): phase A recomputes them
    for (int it = wave; it < 2 * nb; it += SB_WAVES) phase_a_tile<true>(w, it, nt, lane);
    __syncthreads();
  }

  bf16_t* dqkv = a.d_qkv + h * 64;
  for (int jt = wave; jt < nt && !(a.lab & 1); jt += SB_WAVES) {
    f32x4_t dk[4], dv[4];
    phase_b_tile16<DROP, NB>(w, jt, nb, dk, dv, lane);
    if ((a.lab & 4) && dk[0][0] + dv[0][0] != 12345.f) continue;
    tile16_to_patch(patch, dk, lane);
    patch_to_global(patch, dqkv + a.D, a.ld_qkv, row_base, 1, jt * 16, L, 16, lane);
    tile16_to_patch(patch, dv, lane);
    patch_to_global(patch, dqkv + 2 * a.D, a.ld_qkv, row_base, 1, jt * 16, L, 16, lane);
  }
  for (int it = wave; it < nt && !(a.lab & 2); it += SB_WAVES) {
    f32x4_t dq[4];
    phase_c_tile16<DROP, NB>(w, it, nb, dq, lane);
    if ((a.lab & 4) && dq[0][0] != 12345.f) continue;
    tile16_to_patch(patch, dq, lane);
    patch_to_global(patch, dqkv, a.ld_qkv, row_base, 1, it * 16, L, 16, lane);
  }
}

hipError_t sf_launch_spatial_attention_bwd(const SfAttnBwdArgs& a, hipStream_t s) {
  if (a.L <= 0 || a.L > SB_ROWS || a.nseq <= 0 || a.D != a.heads * 64) return hipErrorInvalidValue;
  if ((a.ld_qkv % 8) || (a.ld_o % 8)) return hipErrorInvalidValue;
  const size_t lds = 4 * SB_IMG + 2 * SB_ROWS * 4 + SB_WAVES * SB_PATCH;
  static SfPerDeviceOnce attr_set;
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_spatial_attn_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_spatial_attn_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_spatial_attn_bwd_kernel<false, 13>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  static const int lab = SF_LAB_SWITCH("SF_ATTN_BWD_LAB");      // timing lab: phases off, results invalid (lab builds only)
  const bool ntc_off = sf_sw(SW_DISABLE_SPATIAL_NTC) != nullptr;      // A/B switch (same one as the forward kernel)
  SfAttnBwdArgs b = a;
  b.lab = lab;
  const int nt = (a.L + 15) >> 4;
  if (b.drop.on) hipLaunchKernelGGL(sf_spatial_attn_bwd_kernel<true>, dim3(a.nseq * a.heads), dim3(SB_THREADS), lds, s, b);
  else if (nt == 13 && !a.causal && !ntc_off)      // 224^2 frames: 196 patches = 13 tiles, block loops unrolled
    hipLaunchKernelGGL((sf_spatial_attn_bwd_kernel<false, 13>), dim3(a.nseq * a.heads), dim3(SB_THREADS), lds, s, b);
  else hipLaunchKernelGGL(sf_spatial_attn_bwd_kernel<false>, dim3(a.nseq * a.heads), dim3(SB_THREADS), lds, s, b);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// temporal: one wave per (batch, patch, head) sequence; wave-private images of NP rows
// ------------------------------------------------------------------------------------------------
// `lab` (lab builds only, SF_TBWD_LAB; always 0 in the product library): the device-sharing experiment of DESIGN.md 4 —
// 1 = no patch writes, 2 = return behind the operand staging, 3 = 64 KB of LDS, 5 = the patch by 32-bit pair stores,
// 6 = plain reads in place of ds_read_b64_tr_b16, 7 = return behind phase A
// WPB = waves (= sequences) per workgroup: 4 with the exact LDS (41 KB at 16 frames, three workgroups per CU) in the product launch.
// SF_TBWD_OWN_CU=1 takes 12-wave workgroups that ask for a CU's whole LDS instead: the same 12 waves per CU, but no workgroup of
// another kernel beside them — this kernel's phases B / C are what disturbed a neighbouring workgroup of another process (DESIGN.md 4,
// "Device sharing").  Bit-identical results, 65.3 against 51.0 us per launch at the training step's shape (tools/tbwd_ab.py: a
// 12-wave workgroup holds its CU until its last wave ends), so it is the opt-in for jobs that share a device, not the default.
template <int NP, int WPB>
__global__ __launch_bounds__(64 * WPB) void sf_temporal_attn_bwd_kernel(SfAttnBwdArgs a, int nprob, int lab) {
  constexpr bool TWO = NP > 16;
  constexpr int IMG = NP * 128;
  constexpr int PER_WAVE = 4 * IMG + 2 * NP * 4 + IMG;      // images, lse2/delta, patch
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int prob = blockIdx.x * WPB + wave;
  if (prob >= nprob) return;
  const int h = prob % a.heads;
  const int bn = prob / a.heads;
  const int n = bn % a.seq_rows, b = bn / a.seq_rows;
  const int L = a.L;
  char* base = smem + wave * PER_WAVE;
  char* iq = base;
  char* ik = base + IMG;
  char* iv = base + 2 * IMG;
  char* ig = base + 3 * IMG;
  float* lse2 = reinterpret_cast<float*>(base + 4 * IMG);
  float* delta = lse2 + NP;
  char* patch = base + 4 * IMG + 2 * NP * 4;

  const long row_base = (long)b * L * a.seq_rows + n, row_step = a.seq_rows;
  const bf16_t* qkv = a.qkv + h * 64;
  stage_problem<NP / 8>(iq, ik, iv, ig, delta, qkv, a.D, a.d_o + h * 64, a.o + h * 64, a.ld_qkv, a.ld_o, row_base, row_step, L, NP, lane, 64);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef SF_LAB
  if (lab == 2) return;
#endif

  BwdView w;
  w.q = iq; w.k = ik; w.v = iv; w.d_o = ig; w.lse2 = lse2; w.delta = delta; w.patch = patch;
  w.L = L; w.causal = a.causal; w.scale = a.scale; w.sl2 = a.scale * LOG2E;
  w.drop = a.drop; w.drop_base = (unsigned)(((size_t)bn * a.heads + h) * (size_t)L * L);
  constexpr int NT = NP / 16;
  for (int it = 0; it < NT; ++it) phase_a_tile<TWO>(w, it, NT, lane);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef SF_LAB
  if (lab == 7) return;               // phase A only (row fragments, MFMA, exp2, shuffles; no transposed reads)
#endif

  bf16_t* dqkv = a.d_qkv + h * 64;
#ifdef SF_LAB
#define TBWD_PATCH(t) do { if (lab == 5) tiles_to_patch_pairs<TWO>(patch, t, lane); else if (lab != 1) tiles_to_patch<TWO>(patch, t, lane); } while (0)
#else
#define TBWD_PATCH(t) tiles_to_patch<TWO>(patch, t, lane)
#endif
  {
    f32x4_t dk[2][4], dv[2][4];
    phase_b_block<TWO>(w, 0, 1, dk, dv, lane);
    TBWD_PATCH(dk);
    patch_to_global(patch, dqkv + a.D, a.ld_qkv, row_base, row_step, 0, L, NP, lane);
    TBWD_PATCH(dv);
    patch_to_global(patch, dqkv + 2 * a.D, a.ld_qkv, row_base, row_step, 0, L, NP, lane);
  }
  {
    f32x4_t dq[2][4];
    phase_c_block<TWO>(w, 0, 1, dq, lane);
    TBWD_PATCH(dq);
    patch_to_global(patch, dqkv, a.ld_qkv, row_base, row_step, 0, L, NP, lane);
  }
#undef TBWD_PATCH
}

hipError_t sf_launch_temporal_attention_bwd(const SfAttnBwdArgs& a, hipStream_t s) {
  if (a.L <= 0 || a.L > 32 || a.nseq <= 0 || a.seq_rows <= 0 || a.D != a.heads * 64) return hipErrorInvalidValue;
  if ((a.ld_qkv % 8) || (a.ld_o % 8)) return hipErrorInvalidValue;
  const int nprob = a.nseq * a.heads;
  const int lab = SF_LAB_SWITCH("SF_TBWD_LAB");       // 0 in the product library
#ifdef SF_LAB
  { const int plain = lab == 6; (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_tbwd_plain_reads), &plain, sizeof(int), 0, hipMemcpyHostToDevice, s); }
#endif
  const char* own_sw = sf_sw(SW_TBWD_OWN_CU);
  const bool share = own_sw == nullptr || own_sw[0] == '0';      // default: share the CU (exact LDS sizes); the trainer sets 1 when world > 1
  const size_t whole_cu = (size_t)160 * 1024;
  static SfPerDeviceOnce attr_set;
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_temporal_attn_bwd_kernel<16, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)whole_cu);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_temporal_attn_bwd_kernel<32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)whole_cu);
  }
  if (a.L <= 16) {
    if (share) {
      size_t lds = 4 * (size_t)(5 * 16 * 128 + 2 * 16 * 4);
      if (lab == 3) lds = 64 * 1024;      // lab: an allocation that cannot share a CU with the 102 KB probe kernel
      hipLaunchKernelGGL((sf_temporal_attn_bwd_kernel<16, 4>), dim3((nprob + 3) / 4), dim3(256), lds, s, a, nprob, lab);
    } else {
      hipLaunchKernelGGL((sf_temporal_attn_bwd_kernel<16, 12>), dim3((nprob + 11) / 12), dim3(768), whole_cu, s, a, nprob, lab);
    }
  } else {
    // 32-row images: 4 waves take 83 KB, a second workgroup of this kernel never fitted beside them; the whole-CU request keeps others out too
    const size_t lds = share ? 4 * (size_t)(5 * 32 * 128 + 2 * 32 * 4) : whole_cu;
    hipLaunchKernelGGL((sf_temporal_attn_bwd_kernel<32, 4>), dim3((nprob + 3) / 4), dim3(256), lds, s, a, nprob, lab);
  }
  return hipGetLastError();
}

