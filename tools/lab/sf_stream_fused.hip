// Streamed frame (config #5, vqa_enc:491-560): the temporal qkv projection, the KV-cache append and the single-query temporal
// attention in ONE launch (round 6).  Per layer the streamed frame ran  [LayerNorm-folded qkv GEMM -> cache row]  and
// [single-query attention over the cached rows]  as two dependent launches of 6.5 + 7.4 us that are both latency chains; a patch's
// q, k, v feed only that patch's attention (timesformer_encoder.py:491-560: the new frame's key / value are appended, the one
// query attends to every cached frame of its own patch), so the pair is row-local per (patch, head).
//
// Workgroup = (12 patches, one head), 12 waves:
//   * GEMM: the tile is [16 rows (12 owned + 4 of the next tile, discarded)] x [192 columns = q | k | v of the head]; wave w owns
//     the 16-column MFMA tile w and keeps its whole W' slice in REGISTERS ([16 rows][K] bf16 = K/32 x 16 bytes per lane, 96
//     VGPRs at K = 768), fetched from a fragment-major copy of W' by K/32 contiguous-KiB loads that are all in flight at once — no LDS ring, no K loop of wait -> barrier -> read
//     steps: one memory latency for the whole operand.  A (16 rows of bf16(x)) goes through LDS once (LDS-DMA, XOR-swizzled
//     16-byte slots, conflict-free for the real ds_read_b128 lane groups of gfx950, see sf_swz64 in sf_common.h for the table);
//     LayerNorm statistics from the A fragments as in the skinny kernels (same instruction order: the q / k / v values are
//     bit-identical to the unfused path's).
//   * q | k | v of the 16 x 192 tile go to LDS as bf16 and the 12 owned rows to the cache row of the frame's slot.
//   * attention: wave w = patch w of the tile, the arithmetic of sf_temporal_decode_lines_kernel<1> (<= 64 keys: 8 keys x 128 B
//     per load instruction, DPP reductions); the key / value of the frame itself come from LDS, not from the row just stored.
// 17 x 12 = 204 workgroups at one stream: every workgroup has a CU, one round.
//
// RESULT (profiles/r06_streaming_fused_qkv_ab.txt): bit-identical to the two launches, 31 streaming tests green — and 14.95 us per launch
// against 6.1 + 7.4 us, p50 0.717-0.724 against 0.708-0.711 ms.  Three versions: W' from the row-major matrix 19.3 us (16 lines x 16 B per
// 16-lane group), fragment-major W' + key prefetch 15.0, + scalar position load / no vmcnt(0) between operand and key loads 14.95: the
// launch is bound by what ONE CU can ingest — 320 KB per workgroup arrive at ~27 GB/s (11 B/clk) whatever the order of the requests —
// where the unfused projection spreads the same bytes over 504 workgroups on all CUs.  Same finding as round 3's ring-buffered form
// (docs/history.md A.4.1a).  Lab library only (SF_LIB=lab SF_STREAM_QKV_FUSE=1); the product path keeps the two launches.
#include "sf_common.h"
#include "sf_switches.h"

#define SQ_ROWS 12
#define SQ_WAVES 12
#define SQ_THREADS (64 * SQ_WAVES)
#define SQ_HD 64

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

SF_DEVICE f32x4_t sq_mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}
template <int CTRL>
SF_DEVICE float sq_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

template <int NK>      // K = 32 NK, NK % 4 == 0
__global__ __launch_bounds__(SQ_THREADS) void sf_stream_qkv_decode_kernel(SfStreamQkvArgs p) {
  constexpr int CPR = NK * 4;                      // 16-byte chunks per A row
  constexpr int ROWB = NK * 64;                    // bytes per A row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* a_img = smem;                              // [16][ROWB], slot = chunk ^ row
  unsigned short* qkv_l = reinterpret_cast<unsigned short*>(smem + 16 * ROWB);      // [16][192] bf16
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * SQ_ROWS, h = blockIdx.y;
  const int K = NK * 32;

  // cache position of the frame: from device memory inside the position-free graph.  A SCALAR load issued first: as a vector load
  // behind the operand loads (what hipcc makes of a plain read in a kernel that also stores) its vmcnt(0) put the whole W' latency in
  // front of the key prefetch below
  int slot = p.slot, Tk = p.Tk;
  if (p.pos_dev) {
    typedef __attribute__((ext_vector_type(2))) int i32x2_t;
    i32x2_t sk;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sk) : "s"(p.pos_dev) : "memory");
    slot = sk[0];
    Tk = min(sk[1], 64);
  }
  // ---- A rows -> LDS (LDS-DMA: the image is lane-linear, the swizzle sits on the source side) ----------------------------------
  constexpr int A_CHUNKS = 16 * CPR;
#pragma unroll
  for (int i = 0; i < (A_CHUNKS + SQ_THREADS - 1) / SQ_THREADS; ++i) {
    const int c = i * SQ_THREADS + tid;
    if (c < A_CHUNKS) {
      const int row = c / CPR, pos = c % CPR;
      const int gr = min(m0 + row, p.M - 1);
      const bf16_t* src = p.a + (size_t)gr * K + (pos ^ row) * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(a_img + (i * SQ_THREADS + wave * 64) * 16), 16, 0, 0);
    }
  }
  // ---- this wave's W' slice -> registers: rows col0 .. col0 + 15 of [3D, K], every load in flight at once -------------------------
  const int part = wave >> 2, sub = wave & 3;      // 0 = q, 1 = k, 2 = v; 16-column block of the head
  const int col0 = part * p.D + h * SQ_HD + sub * 16;
  bf16x8_t wreg[NK];
  {
    // fragment-major copy of W' (sf_encoder.hip, upload_folded_linear): one contiguous KiB per load instruction.  (Read from the
    // row-major matrix — 16 lines x 16 bytes per 16-lane group — the same launch took 19.3 us instead of the 13.6 us of the two
    // launches it replaces: profiles/r06_streaming_fused_qkv_ab.txt.)
    const bf16_t* wp = p.w_frag + ((size_t)(col0 >> 4) * NK * 64 + lane) * 8;
#pragma unroll
    for (int j = 0; j < NK; ++j) wreg[j] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)j * 512);
  }
  // epilogue operands behind them
  const int n_e = col0 + g * 4;
  const f32x4_t bias4 = *reinterpret_cast<const f32x4_t*>(p.bias + n_e);      // unconditional (b' = b + W beta always exists): a load under a branch costs a vmcnt(0) at the join
  const f32x4_t lns4 = *reinterpret_cast<const f32x4_t*>(p.ln_s + n_e);
  // the cached keys of this wave's attention task (patch m0 + wave) do not depend on the projection: their lines are requested NOW, so
  // that the cache's HBM latency runs beside the operand fetch instead of behind the GEMM (the values follow after the MFMAs, when
  // the W' registers are free)
  const int m_t = min(m0 + wave, p.M - 1);
  const int b_t = m_t / p.N, n_t = m_t % p.N;
  const int tsub = lane >> 3, ch = lane & 7;       // key = 8 i + tsub, dims 8 ch .. 8 ch + 7
  const char* kb = reinterpret_cast<const char*>(p.cache + p.D);
  const char* vb = reinterpret_cast<const char*>(p.cache + 2 * p.D);
  unsigned koff[8];                               // byte offsets inside one layer's cache (< 4 GiB: checked by the launcher)
  u32x4_t kv[8], vv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int key = i * 8 + tsub;
    key = key < Tk ? key : Tk - 1;
    koff[i] = (unsigned)(((((size_t)b_t * p.cap + key) * p.N + n_t) * (size_t)(3 * p.D) + h * SQ_HD + ch * 8) * 2);
    kv[i] = *reinterpret_cast<const u32x4_t*>(kb + koff[i]);
  }
  // one wait for everything: the A image is complete once every wave's DMA pieces have landed, and the first MFMA needs wreg[0] anyway
  // (all loads of the wave were issued together; the 24 MFMAs behind the wait are ~0.2 us — nothing to pipeline)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  float ln1 = 0.f, ln2 = 0.f;
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(a_img + l15 * ROWB + (((j * 4 + g) ^ l15) << 4));
    sf_lnf_stats(af, ln1, ln2);
    acc = sq_mfma(wreg[j], af, acc);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) vv[i] = *reinterpret_cast<const u32x4_t*>(vb + koff[i]);
  ln1 += __shfl_xor(ln1, 16, 64); ln1 += __shfl_xor(ln1, 32, 64);
  ln2 += __shfl_xor(ln2, 16, 64); ln2 += __shfl_xor(ln2, 32, 64);
  float mean, rstd;
  sf_lnf_finish(ln1, ln2, K, p.ln_eps, mean, rstd);
  const f32x4_t y = rstd * (acc - mean * lns4) + bias4;      // lane: row l15, columns col0 + 4 g .. + 3
  const u32x2_t yb = {pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
  *reinterpret_cast<u32x2_t*>(qkv_l + l15 * 192 + part * 64 + sub * 16 + g * 4) = yb;
  {
    const int m = m0 + l15;
    if (l15 < SQ_ROWS && m < p.M) {                 // the frame's row of the cache: q | k | v like the unfused projection writes it
      const int b = m / p.N, n = m % p.N;
      const size_t row = ((size_t)b * p.cap + slot) * p.N + n;
      *reinterpret_cast<u32x2_t*>(p.cache + row * (size_t)(3 * p.D) + col0 + g * 4) = yb;
    }
  }
  __syncthreads();

  // ---- single-query attention of patch m0 + wave, head h (sf_temporal_decode_lines_kernel<1>) ----------------------------------
  const int m = m0 + wave;
  if (m >= p.M) return;
  const u32x4_t qv = *reinterpret_cast<const u32x4_t*>(qkv_l + wave * 192 + ch * 8);
  const u32x4_t k_new = *reinterpret_cast<const u32x4_t*>(qkv_l + wave * 192 + 64 + ch * 8);
  const u32x4_t v_new = *reinterpret_cast<const u32x4_t*>(qkv_l + wave * 192 + 128 + ch * 8);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 8; ++i)      // the frame's own key: from the tile (its cache row was requested before it was written); so are the clamped
    if (i * 8 + tsub == slot || i * 8 + tsub >= Tk) { kv[i] = k_new; vv[i] = v_new; }      // rows past the last key (masked: weight 0 x a finite value)

  float qf[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    qf[2 * j] = bf2f(qv[j] & 0xffffu);
    qf[2 * j + 1] = __uint_as_float(qv[j] & 0xffff0000u);
  }
  float sc[8];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a0 = fmaf(qf[2 * j], bf2f(kv[i][j] & 0xffffu), a0);
      a1 = fmaf(qf[2 * j + 1], __uint_as_float(kv[i][j] & 0xffff0000u), a1);
    }
    float a = a0 + a1;
    a += sq_dpp<0xB1>(a);                          // quad_perm [1,0,3,2]
    a += sq_dpp<0x4E>(a);                          // quad_perm [2,3,0,1]
    a += sq_dpp<0x141>(a);                         // row_half_mirror
    const int key = i * 8 + tsub;
    const bool ok = key < Tk;                      // one new frame: every cached key is at or before the query (causal or not)
    sc[i] = ok ? a : -INFINITY;
    mx = fmaxf(mx, sc[i]);
  }
  mx = fmaxf(mx, sq_dpp<0x128>(mx));               // row_ror:8
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float c2 = p.scale * 1.44269504088896340736f;
  float sum = 0.f;
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float pt = __builtin_amdgcn_exp2f((sc[i] - mx) * c2);
    sum += pt;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[2 * j] = fmaf(pt, bf2f(vv[i][j] & 0xffffu), o[2 * j]);
      o[2 * j + 1] = fmaf(pt, __uint_as_float(vv[i][j] & 0xffff0000u), o[2 * j + 1]);
    }
  }
  sum += sq_dpp<0x128>(sum);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] += sq_dpp<0x128>(o[j]);
  sum += __shfl_xor(sum, 16, 64);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] += __shfl_xor(o[j], 16, 64);
  sum += __shfl_xor(sum, 32, 64);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] += __shfl_xor(o[j], 32, 64);
  if (tsub == 0) {
    const float inv = 1.0f / sum;
    unsigned int hb[8], lb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split_bf(o[j] * inv, hb[j], lb[j]);
    *reinterpret_cast<u32x4_t*>(p.ctx + (size_t)m * p.D + h * SQ_HD + ch * 8) =
        (u32x4_t){hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16)};
  }
}

bool sf_stream_qkv_decode_supported(const SfStreamQkvArgs& a) {
  if (!SF_LAB_SWITCH("SF_STREAM_QKV_FUSE")) return false;      // lab library only, opt-in
  if (!a.a || !a.w_frag || !a.bias || !a.ln_s || !a.cache || !a.ctx) return false;
  if (a.M <= 0 || a.M > 512 || a.N <= 0 || a.D != a.heads * SQ_HD) return false;      // every 12-row tile re-reads its head's W' slice: one or two streams per call
  if (a.K != 128 && a.K != 256 && a.K != 512 && a.K != 768) return false;      // W' slice in registers: K / 32 x 4 VGPRs, K / 32 % 4 == 0
  if (a.cap < 1 || a.cap > 64 || (size_t)((a.M + a.N - 1) / a.N) * a.cap * a.N * 3 * a.D * 2 >= ((size_t)1 << 32)) return false;       // one 64-key pass (KP = 1 of the decode kernel); longer caches take the two launches
  if (!a.pos_dev && (a.Tk < 1 || a.Tk > 64 || a.slot < 0 || a.slot >= a.cap)) return false;
  return true;
}

hipError_t sf_launch_stream_qkv_decode(const SfStreamQkvArgs& a, hipStream_t s) {
  if (!sf_stream_qkv_decode_supported(a)) return hipErrorInvalidValue;
  const dim3 grid((a.M + SQ_ROWS - 1) / SQ_ROWS, a.heads), block(SQ_THREADS);
  const size_t lds = (size_t)16 * a.K * 2 + 16 * 192 * 2;
  switch (a.K / 32) {
    case 4: hipLaunchKernelGGL(sf_stream_qkv_decode_kernel<4>, grid, block, lds, s, a); break;
    case 8: hipLaunchKernelGGL(sf_stream_qkv_decode_kernel<8>, grid, block, lds, s, a); break;
    case 16: hipLaunchKernelGGL(sf_stream_qkv_decode_kernel<16>, grid, block, lds, s, a); break;
    default: hipLaunchKernelGGL(sf_stream_qkv_decode_kernel<24>, grid, block, lds, s, a); break;
  }
  return hipGetLastError();
}
