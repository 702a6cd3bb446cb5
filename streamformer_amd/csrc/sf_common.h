// Shared device helpers and kernel launch declarations (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // 16x16 MFMA C/D fragment
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define SF_DEVICE __device__ __forceinline__

// Lab switches that DISCARD results (stores off, phases off) exist only in builds made with -DSF_LAB
// (`python streamformer_amd/build.py --lab` -> libstreamformer_hip_lab.so, used by tools/ alone).  In the product library the
// macro is the constant 0 and the variable name is not even compiled in (tests/test_abi.py checks the .so for such names).
#ifdef SF_LAB
#define SF_LAB_SWITCH(name) (getenv(name) ? atoi(getenv(name)) : 0)
#else
#define SF_LAB_SWITCH(name) 0
#endif

// 16-byte slot XOR of an LDS image with 64-BYTE rows (32 bf16 of K) whose MFMA fragments are read with ds_read_b128 (lane -> row
// base + (lane & 15), slot lane >> 4).  gfx950 serves one ds_read_b128 in four groups of 16 lanes that are NOT contiguous —
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) — so a group holds rows 0-3 and 12-15 at
// slot g and rows 4-11 at slot g ^ 1.  Rows r, r+4, r+8, r+12 share their 64 bytes of the 256-byte bank row and must land on four
// distinct slots: with s(q) = -q & 3 (q = row >> 2) the group reads slots {g, g^1^3, g^1^2, g^1} = all four.  Rounds 1-5 used s(q) = q,
// correct for contiguous 16-lane groups and 2-way conflicted on the real ones (r05 counters: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
// 0.78 on the panel kernel against 0.077 on the 128-byte-row images, whose (row >> 1) & 7 is conflict-free under the same table).
// tools/lds_swizzle_lab.hip times every s: {0..3} -> {0..3} on the device.  SF_SWZ64_LEGACY (lab builds): the old function, for A/B.
SF_DEVICE int sf_swz64(int row) {
#ifdef SF_SWZ64_LEGACY
  return (row >> 2) & 3;
#else
  return (0 - (row >> 2)) & 3;
#endif
}

// round-to-nearest-even fp32 -> bf16: the casts lower to gfx950's v_cvt_pk_bf16_f32 (one VALU op per
// pair instead of the 4-op integer sequence)
SF_DEVICE unsigned int f2bf(float f) {
  const __bf16 b = (__bf16)f;
  return (unsigned int)__builtin_bit_cast(unsigned short, b);
}
SF_DEVICE float bf2f(unsigned int b) { return __uint_as_float(b << 16); }
SF_DEVICE unsigned int pack_bf2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): the operand split of SF_COMPUTE_BF16X3
SF_DEVICE void split_bf(float x, unsigned int& hi, unsigned int& lo) {
  hi = f2bf(x);
  lo = f2bf(x - bf2f(hi));
}

// LayerNorm folded into the consumer at small M (LNF): A = bf16(x) of the residual stream, W' = W * gamma.  The wave adds
// up sum x and sum x^2 of its 16 rows from the very A fragments it feeds to the MFMAs (v_dot2c_f32_bf16: two VALU ops
// per 4 products; lane (l15, g) sees the k-chunks g, g+4 of row l15), the four k-groups meet by two xor-shuffles, and the
// epilogue finishes y = rstd (acc - mean s_n) + b'.  No statistics buffer, no extra pass over the rows: the 36 LayerNorm
// launches of a streamed frame disappear.  The statistics are those of the bf16-rounded rows (what the products see).
SF_DEVICE void sf_lnf_stats(const bf16x8_t& f, float& s1, float& s2) {
  typedef __attribute__((ext_vector_type(2))) __bf16 v2bf;
  typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
  const v8bf h = __builtin_bit_cast(v8bf, f);
  const v2bf one = {(__bf16)1.0f, (__bf16)1.0f};
  // pairs taken with shufflevector: indexing a bit-cast u32x4 copy of the fragment (u[j]) made hipcc 7.2 feed dword 0 to
  // all four dot products (seen in the ISA as four v_dot2c on the same VGPR)
  const v2bf x0 = __builtin_shufflevector(h, h, 0, 1), x1 = __builtin_shufflevector(h, h, 2, 3);
  const v2bf x2 = __builtin_shufflevector(h, h, 4, 5), x3 = __builtin_shufflevector(h, h, 6, 7);
  s1 = __builtin_amdgcn_fdot2_f32_bf16(x0, one, s1, false);
  s2 = __builtin_amdgcn_fdot2_f32_bf16(x0, x0, s2, false);
  s1 = __builtin_amdgcn_fdot2_f32_bf16(x1, one, s1, false);
  s2 = __builtin_amdgcn_fdot2_f32_bf16(x1, x1, s2, false);
  s1 = __builtin_amdgcn_fdot2_f32_bf16(x2, one, s1, false);
  s2 = __builtin_amdgcn_fdot2_f32_bf16(x2, x2, s2, false);
  s1 = __builtin_amdgcn_fdot2_f32_bf16(x3, one, s1, false);
  s2 = __builtin_amdgcn_fdot2_f32_bf16(x3, x3, s2, false);
}
// the same for a row that arrives as hi + lo bf16 planes (fp32-accurate mode): sums of x = h + l and of x^2 = h^2 + 2 h l + l^2
SF_DEVICE void sf_lnf_stats_split(const bf16x8_t& fh, const bf16x8_t& fl, float& s1, float& s2) {
  typedef __attribute__((ext_vector_type(2))) __bf16 v2bf;
  typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
  const v8bf h = __builtin_bit_cast(v8bf, fh), l = __builtin_bit_cast(v8bf, fl);
  const v2bf one = {(__bf16)1.0f, (__bf16)1.0f};
#define SF_LNF_PAIR(A, B)                                                   \
  {                                                                         \
    const v2bf hh = __builtin_shufflevector(h, h, A, B), ll = __builtin_shufflevector(l, l, A, B); \
    s1 = __builtin_amdgcn_fdot2_f32_bf16(hh, one, s1, false);               \
    s1 = __builtin_amdgcn_fdot2_f32_bf16(ll, one, s1, false);               \
    s2 = __builtin_amdgcn_fdot2_f32_bf16(hh, hh, s2, false);                \
    float c = __builtin_amdgcn_fdot2_f32_bf16(hh, ll, 0.f, false);          \
    s2 += 2.0f * c;                                                         \
    s2 = __builtin_amdgcn_fdot2_f32_bf16(ll, ll, s2, false);                \
  }
  SF_LNF_PAIR(0, 1) SF_LNF_PAIR(2, 3) SF_LNF_PAIR(4, 5) SF_LNF_PAIR(6, 7)
#undef SF_LNF_PAIR
}
SF_DEVICE void sf_lnf_finish(float s1, float s2, int K, float eps, float& mean, float& rstd) {
  const float inv_k = 1.0f / (float)K;
  mean = s1 * inv_k;
  rstd = __builtin_amdgcn_rsqf(fmaxf(s2 * inv_k - mean * mean, 0.f) + eps);
}

// hipFuncSetAttribute (the > 64 KB dynamic-LDS opt-in) is a per-DEVICE setting: run the set-up once for every device a
// process launches on, not once per process.
struct SfPerDeviceOnce {
  bool done[64] = {};
  bool first() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
    if (done[d]) return false;
    done[d] = true;
    return true;
  }
};

SF_DEVICE float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Same sum on the DPP path (six VALU adds with row_shr / row_bcast modifiers + one v_readlane instead
// of six dependent ds_bpermute round trips).  All 64 lanes must be active; the result is wave-uniform.
template <int CTRL, int ROW_MASK>
SF_DEVICE float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, moved);
}
SF_DEVICE float wave_sum_dpp(float v) {
  v = dpp_add<0x111, 0xf>(v);   // row_shr:1   inclusive scan inside each row of 16 lanes
  v = dpp_add<0x112, 0xf>(v);   // row_shr:2
  v = dpp_add<0x114, 0xf>(v);   // row_shr:4
  v = dpp_add<0x118, 0xf>(v);   // row_shr:8   lane 15 of each row = row total
  v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3: lane 63 = wave total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
SF_DEVICE float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

SF_DEVICE float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
SF_DEVICE float gelu_tanh(float x) {
  const float k = 0.7978845608028654f;
  return 0.5f * x * (1.0f + tanhf(k * (x + 0.044715f * x * x * x)));
}
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7) on v_rcp/v_exp: ~12 VALU instead of the
// branchy ocml erff.  Used where the result is rounded to bf16 anyway (throughput mode).
SF_DEVICE float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float r = 1.0f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
// erf-GELU with the constants folded: gelu(x) = x * Phi(x), Phi(x) = x >= 0 ? 1 - h : h,
// h = 0.5 * poly(t) * exp(-x^2/2), t = 1 / (1 + p/sqrt(2) * |x|)   (same A&S 7.1.26 approximation)
SF_DEVICE float gelu_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.2316418882f, ax, 1.0f));           // 0.3275911 / sqrt(2)
  float poly = fmaf(t, 0.5307027145f, -0.7265760135f);                            // 0.5 * a5, 0.5 * a4
  poly = fmaf(poly, t, 0.7107068705f);
  poly = fmaf(poly, t, -0.142248368f);
  poly = fmaf(poly, t, 0.127414796f);
  const float h = poly * t * __builtin_amdgcn_exp2f(x * x * -0.72134752044f);     // exp(-x^2/2) = 2^(-x^2 * log2(e)/2)
  return fmaf(-ax, h, fmaxf(x, 0.f));     // x >= 0: x (1 - h);  x < 0: x h = -|x| h   (h = Phi(-|x|))
}
// erf-GELU for results that are rounded to bf16 straight away (the bf16 mode's MLP up-projection epilogues):
// gelu(x) = max(x, 0) - |x| * Phi(-|x|),  Phi(-a) = 0.5 - a * g(a^2) with g a degree-7 polynomial fitted on a <= 4 under
// g(16) = 1/8, i.e. Phi(-a) = 0 for a >= 4 (exact saturation on both sides).  |error| <= 1.4e-4 absolute against the exact
// erf form (peak at |x| = 4: 0.4 % of a bf16 ulp there), 12 VALU ops without a transcendental one and all of them
// packed-fp32 material, against 2 quarter-rate + ~10 full-rate ops of gelu_fast: the up-projection epilogue was spending
// ~28 us per launch at M = 25 088 on the activation (docs/history.md A.4.2a).
SF_DEVICE float gelu_bf16(float x) {
  const float a = fminf(fabsf(x), 4.0f);
  const float u = a * a;
  float r = fmaf(u, -1.300031527e-09f, 1.057452934e-07f);
  r = fmaf(r, u, -3.740224429e-06f);
  r = fmaf(r, u, 7.655379159e-05f);
  r = fmaf(r, u, -1.023336896e-03f);
  r = fmaf(r, u, 9.588709101e-03f);
  r = fmaf(r, u, -6.607423723e-02f);
  r = fmaf(r, u, 3.988095224e-01f);
  const float h = fmaf(-a, r, 0.5f);
  return fmaf(-fabsf(x), h, fmaxf(x, 0.f));
}
// d/dx [x * Phi(x)] = Phi(x) + x * phi(x), Phi through the same A&S erf
SF_DEVICE float gelu_grad_fast(float x) {
  const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
  const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(x * x * -0.72134752044f);
  return cdf + x * pdf;
}
SF_DEVICE float apply_act_fast(float x, int act) {
  if (act == 0) return gelu_fast(x);
  return act == 1 ? gelu_tanh(x) : fmaxf(x, 0.0f);
}
// bf16-rounded outputs only (see gelu_bf16)
SF_DEVICE float apply_act_bf16(float x, int act) {
  if (act == 0) return gelu_bf16(x);
  return act == 1 ? gelu_tanh(x) : fmaxf(x, 0.0f);
}
SF_DEVICE float apply_act(float x, int act) {
  return act == 0 ? gelu_erf(x) : (act == 1 ? gelu_tanh(x) : fmaxf(x, 0.0f));
}

// ------------------------------------------------------------------------------------------------
// GEMM:  C[M,N] = A[M,K] * W[N,K]^T   (both operands K-contiguous bf16; optional lo halves)
// ------------------------------------------------------------------------------------------------
enum SfEpilogue {
  SF_EPI_F32 = 0,        // out_f32 = acc + bias
  SF_EPI_BF16 = 1,       // out_bf16 (+ out_lo) = acc + bias
  SF_EPI_ACT_BF16 = 2,   // out_bf16 (+ out_lo) = act(acc + bias)
  SF_EPI_RESID_F32 = 3,  // out_f32 = resid + alpha * (acc + bias)      (out may alias resid)
  SF_EPI_EMBED_F32 = 4,  // out_f32 = acc + bias + pos[row % Np] + time[(row / Np) % Tn]
};

struct SfGemmArgs {
  const bf16_t* a_hi; const bf16_t* a_lo;   // [M,K]   (a_lo == nullptr unless split)
  const bf16_t* w_hi; const bf16_t* w_lo;   // [N,K]
  const float* bias;                        // [N] or nullptr
  int M, N, K;
  int epi;
  int act;                                  // SF_EPI_ACT_BF16: 0 erf-gelu, 1 tanh-gelu, 2 relu
  float alpha;                              // SF_EPI_RESID_F32
  const float* resid;                       // [M,N] fp32 ([resid_mod,N] when resid_mod > 0: row m reads resid[m % resid_mod],
  int resid_mod;                            //  the position + time embedding table of the patch-embedding GEMM; panel kernel)
  // panel kernel, bf16 mode at BASELINE-sized M: the residual stream as TWO bf16 planes (hi = bf16(x), lo = bf16(x - hi)) instead of
  // fp32 — the hi plane IS the A operand of the LayerNorm-folded Linear that follows, so the separate bf16 copy disappears
  // (192 -> 154 MB per launch).  resid_hi / resid_lo: the incoming residual (instead of `resid`); out_hi / out_lo: the new one.
  const bf16_t* resid_hi; const bf16_t* resid_lo;
  // fp32-accurate mode: a third plane lo2 = bf16(x - hi - lo) that only the residual producers (and the last LayerNorm) read and
  // write — hi + lo + lo2 hold x to 2^-27, so the stream loses nothing to the 36 re-splits of a forward
  const bf16_t* resid_lo2; bf16_t* out_lo2;
  const float* pos; const float* time_rows; // SF_EPI_EMBED_F32: [Np,N], [Tn,N]
  int Np, Tn;
  const int* time_base_dev;                 // skinny kernel only: time row = (row / Np) % Tn + *time_base_dev, i.e. time_rows is the whole
                                            // time-embedding table and the streamed frame's row comes from device memory (no gather launch)
  float* out_f32;                           // [*,ldc]
  bf16_t* out_hi; bf16_t* out_lo;           // [*,ldc]
  int ldc;                                  // output row pitch in elements
  // output row remap (KV-cache appends): out_row = (m / grp_rows) * grp_stride + grp_off + m % grp_rows
  int grp_rows, grp_stride, grp_off;
  // streaming: the cache position comes from DEVICE memory (one hipGraph serves every position):
  //   grp_off += *grp_off_dev * grp_off_scale
  const int* grp_off_dev; int grp_off_scale;
  int w_nt;                                 // skinny kernels: non-temporal policy on the weight loads (lab switch SF_SKINNY_NT)
  // LayerNorm folded into the NEXT Linear (bf16 mode): a residual producer also emits bf16(x) in out_hi
  // and per-row partial sums {sum x, sum x^2} per 384-column half in ln_stats_out [M][4]; the consumer
  // (A = bf16(x), W' = W * gamma) finishes y = rstd * (acc - mean * ln_s[n]) + bias' in its epilogue.
  float* ln_stats_out;
  const float* ln_stats; const float* ln_s; float ln_eps;
  // ln_stats_wide = 1: rows of 8 floats — ln_stats and ln_stats_out alike.  fp32-accurate mode: one {sum x, sum x^2} pair per
  // 256-column tile of the bf16x3 256^2 kernel that produced the residual row (3 pairs + 2 pad).  bf16 mode (round 4): four pairs,
  // one per 192-column quarter (sf_gemm_pp.hip); the 384-column panel kernel fills pairs 0 and 2 and zeroes 1 and 3.
  int ln_stats_wide;
  // small-M variant of the fold (sf_gemm_skinny.hip): ln_inkernel = 1 -> the consumer derives mean / rstd of its rows from the
  // A fragments it streams anyway (A = bf16(x), ln_s as above); no statistics buffer exists
  int ln_inkernel;
  // training-step fusions on a bf16 output (256^2 kernel only; sf_gemm256_aux_supported):
  //   aux_mode 1: also write aux[row, col] = gelu(out)            (forward: pre-activation + activation)
  //   aux_mode 2: out *= gelu'(aux[row, col])                     (backward: d pre = d act * gelu'(pre))
  int aux_mode;
  bf16_t* aux;                              // [*, ldc], same row remap as the output
};
SF_DEVICE size_t sf_out_row(const SfGemmArgs& p, int m) {
  if (p.grp_rows <= 0) return (size_t)m;
  const int off = p.grp_off + (p.grp_off_dev ? *p.grp_off_dev * p.grp_off_scale : 0);
  return (size_t)(m / p.grp_rows) * p.grp_stride + off + (m % p.grp_rows);
}
// nanoseconds -> ticks of wall_clock64() on the current device (s_memrealtime: 100 MHz on MI355X; queried once)
int sf_wall_clock_ticks(int ns);
hipError_t sf_launch_gemm(const SfGemmArgs& a, bool split, hipStream_t s);      // dispatches skinny / panel / 256^2 / 128^2
hipError_t sf_launch_gemm128(const SfGemmArgs& a, bool split, hipStream_t s);   // sf_gemm.hip
bool sf_gemm256_supported(const SfGemmArgs& a, bool split);                      // sf_gemm256.hip
bool sf_gemm256_aux_supported(const SfGemmArgs& a);                              // aux_mode epilogues
hipError_t sf_launch_gemm256(const SfGemmArgs& a, hipStream_t s);
int sf_skinny_max_rows();                                                        // sf_gemm_skinny.hip: largest M the skinny kernels take (several streams per call)
bool sf_gemm_skinny_supported(const SfGemmArgs& a, bool split);                  // sf_gemm_skinny.hip (M <= 512)
hipError_t sf_launch_gemm_skinny(const SfGemmArgs& a, bool split, hipStream_t s);
bool sf_gemm_tile_supported(const SfGemmArgs& a, bool split);                    // sf_gemm_tile.hip: one to a few clips (2560 < M <= sf_tile_max_rows())
hipError_t sf_launch_gemm_tile(const SfGemmArgs& a, hipStream_t s);
int sf_tile_max_rows();
int sf_tile_fold_min_rows();                                                     // tile producers emit row statistics from here up (sf_gemm_tile.hip)
int sf_infold_max_rows();                                                        // largest M of the in-kernel-statistics LayerNorm fold (skinny + tile kernels)
// LayerNorm-folded qkv projection on the panel tile (sf_gemm_qkv.hip): plain = the [M, 3D] bf16 tensor (spatial attention's input),
// fused = the temporal attention of a full 16-frame clip computed in the epilogue, ctx [M, D] written instead of qkv
struct SfQkvArgs {
  const bf16_t* a;                          // [M, K] bf16(x): the hi plane of the residual stream, rows in (clip, frame, patch) order
  const bf16_t* w;                          // [3D, K] W' = W * gamma; fused: rows permuted so that 384-column tile j = [q | k | v] of heads 2j, 2j+1
  const float* bias; const float* ln_s;     // [3D] b' = b + W beta and s_n = sum_k bf16(W')[n, k], permuted like w
  const float* ln_stats; float ln_eps;      // [M][8] row sums of the producer (SfGemmArgs::ln_stats_wide layout)
  int M, K, D;                              // D = heads * 64, N = 3 D
  int B, T, NP;                             // fused: clips, frames per clip (16), patches per frame; M = B T NP
  bf16_t* out;                              // plain: qkv [M, 3D]; fused: ctx [M, D]
  float scale; int causal;                  // fused: softmax scale, causal mask over frames
};
bool sf_gemm_qkv_supported(const SfQkvArgs& a, bool fused);                        // sf_gemm_qkv.hip
hipError_t sf_launch_gemm_qkv(const SfQkvArgs& a, bool fused, hipStream_t s);
// Streamed frame, bf16 mode: temporal qkv projection (LayerNorm folded, statistics in the kernel) + cache append + single-query temporal
// attention in one launch (tools/lab/sf_stream_fused.hip, lab library only; vqa_enc:491-560).  Rows m = b * N + n of ONE new frame per stream.
struct SfStreamQkvArgs {
  const bf16_t* a;                          // [M, K] bf16(x) of the residual stream
  const bf16_t* w_frag;                     // W' = W * gamma [3D, K] in MFMA-fragment order: [n-tile][k-step][lane 16 g + l15][8] = W'[16 t + l15][32 j + 8 g ..]
  const float* bias; const float* ln_s; float ln_eps;      // [3D] b' = b + W beta, s_n = sum_k bf16(W')[n, k]
  int M, K, D, heads, N;                    // N = patches per frame, M = streams * N, D = heads * 64
  bf16_t* cache;                            // temporal qkv rows [(b * cap + t) * N + n][3D]: the frame's row is written, rows t < keys are read
  int cap, slot, Tk;                        // slot = the frame's row, Tk = keys visible (<= 64) when pos_dev == nullptr
  const int* pos_dev;                       // {slot, tk} in device memory (position-free graph)
  bf16_t* ctx;                              // [M, D]
  float scale;
};
bool sf_stream_qkv_decode_supported(const SfStreamQkvArgs& a);
hipError_t sf_launch_stream_qkv_decode(const SfStreamQkvArgs& a, hipStream_t s);
bool sf_gemm_panel_supported(const SfGemmArgs& a, bool split);                   // sf_gemm_panel.hip
hipError_t sf_launch_gemm_panel(const SfGemmArgs& a, hipStream_t s);
bool sf_gemm_pp_supported(const SfGemmArgs& a, bool split);                      // sf_gemm_pp.hip: two workgroups per CU, epilogue beside main loop
hipError_t sf_launch_gemm_pp(const SfGemmArgs& a, hipStream_t s);
bool sf_gemm_pipe_supported(const SfGemmArgs& a, bool split);                    // sf_gemm_pipe.hip: persistent, role-split waves, epilogue of tile k under tile k + 1
hipError_t sf_launch_gemm_pipe(const SfGemmArgs& a, hipStream_t s);
int sf_gemm_pipe_failed();                                                       // 1 after a bounded spin of that kernel gave up (never expected)

// ------------------------------------------------------------------------------------------------
// row-wise / elementwise kernels
// ------------------------------------------------------------------------------------------------
// LayerNorm over D: x fp32 [rows,D] -> any of {y_f32, y_hi, y_lo} (nullptr = skip)
hipError_t sf_launch_layernorm(const float* x, const float* gamma, const float* beta, float* y_f32,
                               bf16_t* y_hi, bf16_t* y_lo, int rows, int D, float eps, hipStream_t s, const bf16_t* xp_hi = nullptr,
                               const bf16_t* xp_lo = nullptr, const bf16_t* xp_lo2 = nullptr, float* const* y_f32_ind = nullptr);
                               // y_f32_ind: the fp32 destination is read from device memory (the streamed frame's caller tensor)
// pixels [F,C,H,W] -> patch matrix [F*N, C*P*P] bf16 (+lo), columns (c,ph,pw).
// pixel_kind 0 fp32, 1 bf16, 2 uint8 raw frames normalised on the fly: y = x * scale[c] + shift[c]
struct SfPixelNorm { float scale[4]; float shift[4]; };
// Per-call parameters of a streamed frame, in DEVICE memory (written by one tiny launch in front of the graph replay): the
// caller's input / output tensors and the cache position.  Kernels of the captured sequence read them through this block, so
// ONE graph serves every call (no per-position capture, no staging copy of the outputs).
// t_row = time-embedding row of the new frame, slot = its row in the KV-cache, tk = keys its query sees (cached + itself).
// Plain streaming: t_row = slot = frames cached, tk = slot + 1.  Sliding window (cache full): slot = t mod capacity,
// tk = capacity, t_row = min(t, num_frames - 1).
struct SfStreamParams { const void* pixels; float* lhs; float* pooler; int t_row; int slot; int tk; };
hipError_t sf_launch_stream_params(SfStreamParams* dst, const SfStreamParams& v, hipStream_t s);
hipError_t sf_launch_patchify(const void* pixels, int pixel_kind, bf16_t* out_hi, bf16_t* out_lo,
                              int F, int C, int H, int W, int P, hipStream_t s, const SfPixelNorm* norm = nullptr,
                              const SfStreamParams* sp = nullptr,      // sp != nullptr: pixels = sp->pixels (device read)
                              SfStreamParams* sp_write = nullptr, const SfStreamParams* sp_value = nullptr,
                              int Kpad = 0);       // > C * P * P: row pitch of the patch matrix, zero-filled past the patch vector
                              // sp_write: the launch also stores *sp_value there (the streamed frame's parameter block rides on the
                              // patch extraction instead of a launch of its own)
// fp32 [n] -> bf16 hi (+lo)
hipError_t sf_launch_split(const float* x, bf16_t* hi, bf16_t* lo, size_t n, hipStream_t s);
// two fp32 copies in one launch (b may be null): the streaming path's hand-over of graph-owned outputs to the caller's tensors
hipError_t sf_launch_copy2(const float* a_src, float* a_dst, size_t na, const float* b_src, float* b_dst, size_t nb, hipStream_t s,
                           const SfStreamParams* sp = nullptr);        // sp != nullptr: a_dst = sp->lhs, b_dst = sp->pooler (device reads)
// fp32 rows -> bf16 copy + LayerNorm partial statistics {sum x, sum x^2, 0, 0} per row (stats [rows][4])
hipError_t sf_launch_rowstats_cast(const float* x, bf16_t* xb, float* stats, int rows, int D, hipStream_t s, bf16_t* xlo = nullptr,
                                   bf16_t* xlo2 = nullptr, int wide = 0);      // wide (or xlo): stats rows of 8 floats
// out[t*N + n, :] = pos[n, :] + time_rows[t, :]   (the additive table of the embeddings, modeling:413-457)
hipError_t sf_launch_pos_time_table(const float* pos, const float* time_rows, float* out, int T, int N, int D, hipStream_t s);
// gather rows: out[t,:] = table[idx[t],:]   (idx passed by value, T <= 256)
struct SfRowIndex { int n; int idx[256]; };
hipError_t sf_launch_gather_rows(const float* table, float* out, const SfRowIndex& idx, int D, hipStream_t s,
                                 const int* base_dev = nullptr);       // row = idx[t] + *base_dev

// ------------------------------------------------------------------------------------------------
// attention
// ------------------------------------------------------------------------------------------------
// Dropout (hidden_dropout_prob / attention_probs_dropout_prob of the config; reference sites modeling:374, 378, 556, 603, 669, 705, 752,
// 761, 822, 835) with COUNTER-BASED masks: element `idx` of site `key` is kept iff sf_drop_hash(idx, key) < thresh and then scaled by
// 1 / keep.  No mask tensor exists: forward, backward and the CPU oracle (oracle/train_oracle.py) evaluate the same integer hash.
// idx = flat index of the element in its frame-major tensor ([M, width] rows (b, t, n); attention: [(sequence, head), query, key]).
struct SfDrop { unsigned on, thresh, key; float scale; };
__host__ __device__ inline unsigned sf_drop_hash(unsigned idx, unsigned key) {
  unsigned x = idx * 0x9E3779B1u + key;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__host__ __device__ inline float sf_drop_factor(const SfDrop& d, unsigned idx) { return sf_drop_hash(idx, d.key) < d.thresh ? d.scale : 0.f; }
// site keys: layer * 8 + {0 temporal SelfOutput, 1 spatial SelfOutput, 2 MLP activation, 3 MLP output, 4 temporal probabilities,
// 5 spatial probabilities}; embeddings: L * 8 + {0 position, 1 time}
inline SfDrop sf_drop_make(float p, unsigned seed, unsigned site) {
  SfDrop d = {0u, 0u, 0u, 1.f};
  if (p > 0.f) {
    const double keep = 1.0 - (double)p;
    const double th = keep * 4294967296.0;
    d.on = 1u; d.thresh = th >= 4294967295.0 ? 4294967295u : (unsigned)th; d.key = sf_drop_hash(site, seed); d.scale = (float)(1.0 / keep);
  }
  return d;
}
struct SfAttnArgs {
  // q/k/v element (seq position p of sequence g, head h, dim e) lives at
  //   base[(g_off(g) + p * pos_stride) * row_pitch + col0 + h*64 + e]
  // with col0 = 0 / D / 2D for q / k / v.  Either bf16 (fast) or fp32 (accurate) storage.
  const void* q; const void* k; const void* v;
  int in_is_f32;
  long long lo_plane_off;             // accurate mode with bf16 storage: elements from a hi value to its lo value (spatial DMA kernel)
  int row_pitch_q, row_pitch_kv;     // elements per token row of the q / kv buffers
  int heads;
  float scale;
  // spatial: one sequence per frame: rows [f*N, f*N+N)
  int N;                              // keys == queries == N tokens
  int frames;
  // temporal: sequence over frames for fixed (b, n):
  //   q rows ((b*Tq_cap + q_t0 + t)*N + n), t < Tq ; kv rows ((b*Tcap + t)*N + n), t < Tk ;
  //   query t sits at absolute frame t_past + t (causal: keys <= that); ctx rows ((b*Tq + t)*N + n)
  int B, Tq, Tk, Tcap, t_past, causal, Tq_cap, q_t0;
  const int* pos_dev;                 // single-query decode kernel only: {slot, tk} from device memory (position-free graph): q row = slot,
                                      // tk keys, all of them visible
  bf16_t* ctx_hi; bf16_t* ctx_lo;     // [rows, D] output (lo only in accurate mode)
  int D;
  int head_dim;                       // 0 or 64: the tuned kernels of sf_attention.hip; any other multiple of 8 up to 128: sf_attention_generic.hip
  float* lse2_out;                    // spatial only, optional: base-2 log-sum-exp of the scaled scores per query,
                                      // [frames, heads, N] fp32 (kept by the training forward for the backward kernel)
  float* probs;                       // spatial only, optional: softmax probabilities [frames, heads, N, N] fp32
                                      // (output_attentions=True, modeling:703-716); N <= 224
  SfDrop drop;                        // training forward: dropout on the probabilities (on = 0: none); element ((seq * heads + h) * Lq + q) * Lk + k,
                                      // seq = frame (spatial) / b * N + n (temporal).  DMA-staged bf16 kernels only.
};
hipError_t sf_launch_spatial_attention(const SfAttnArgs& a, bool accurate, hipStream_t s);
bool sf_spatial_planes_ok(int N, bool probs);
bool sf_temporal_planes_ok(int Tq, int Tk);
hipError_t sf_launch_temporal_attention(const SfAttnArgs& a, bool accurate, hipStream_t s);
bool sf_attention_generic_supported(const SfAttnArgs& a, int head_dim);            // sf_attention_generic.hip: head_dim % 8 == 0, <= 128
hipError_t sf_launch_attention_generic(const SfAttnArgs& a, int head_dim, bool temporal, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// loss heads
// ------------------------------------------------------------------------------------------------
// logit_scale / logit_bias: device pointers (one float each); partial: caller scratch of sf_loss_partial_bytes(rows)
size_t sf_loss_partial_bytes(int rows);
hipError_t sf_launch_retrieval_loss(const float* pooler, const float* text, int B, int T, int D, int Bt,
                                    int pos_offset, const float* logit_scale, const float* logit_bias,
                                    float* loss, float* grad_pooler, float* grad_scalars, float* partial, hipStream_t s);
hipError_t sf_launch_localization_loss(const float* pooler, const float* label_emb, const int* labels,
                                       int B, int T, int D, int L, const float* logit_scale, const float* logit_bias,
                                       float* loss, float* grad_pooler, float* grad_scalars, float* partial, hipStream_t s);
