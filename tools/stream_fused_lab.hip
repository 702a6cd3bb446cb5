// LAB, NOT PART OF THE LIBRARY (negative result of round 3; kept for the record — it was built as
// streamformer_amd/csrc/sf_stream_fused.hip with `struct SfQkvDecodeArgs` in sf_common.h and passed the streaming parity tests):
// Streamed frame, temporal branch in ONE launch per layer:
//   temporal qkv projection (LayerNorm folded, in-kernel statistics) -> KV-cache append -> single-query causal attention
// replacing sf_gemm_skinny_kernel<BF16, LNF> + sf_temporal_decode_kernel of the streaming step
// (reference: downstream/VideoQA/llava/model/multimodal_encoder/timesformer_encoder.py:491-560, one new frame per call).
//
// Why it fuses: the workgroup that produces (q_h, k_h, v_h) of 16 token rows holds everything head h's single query of
// those 16 (stream, patch) tasks needs from THIS frame; everything else is the cache of earlier frames, which does not depend
// on the projection at all.  So: grid = (16-row tiles) x heads, 16 waves per workgroup,
//   * every wave first requests the cached K / V rows of ITS task (row tile row = wave; lane = key, then lane = (key mod 8,
//     16-byte chunk), exactly the loads of sf_temporal_decode_kernel<false, 1>) — they fly while the projection runs;
//   * waves 4..7 stream A [16 x 768] and the head's three weight blocks W'[q_h | k_h | v_h] (192 rows) through a 4-stage
//     LDS-DMA ring (counted vmcnt), waves 0..3 run the MFMAs (wave w owns dims 16w..16w+15 of q, k and v), take the
//     LayerNorm statistics from the A fragments, finish y = rstd (acc - mean s) + b', write k / v of the new frame into the
//     cache and the (q, k, v) tile into LDS;
//   * after ONE barrier all 16 waves run the single-query attention of their task, the new key / value coming from LDS.
// bf16 mode, at most 64 cached frames (one 64-key pass), head_dim 64; other cases keep the two-kernel path.
#include "sf_common.h"
#include <cstdlib>

#define QD_THREADS 1024
#define QD_ROWS 16
#define QD_BK 64
#define QD_STAGES 4
#define QD_NI 7                                   // 256 loader threads x 16 B = 32 stage rows per instruction; 208 rows -> 7
#define QD_STAGE_BYTES (QD_NI * 4096)
#define QD_TILE_OFF (QD_STAGES * QD_STAGE_BYTES)   // [16 rows][192] bf16 = 6 KB behind the ring
#define QD_LDS (QD_TILE_OFF + QD_ROWS * 192 * 2)

typedef __attribute__((address_space(3))) void* qd_lptr_t;

SF_DEVICE f32x4_t qd_mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}
SF_DEVICE bf16x8_t qd_frag(const char* img, int row, int kc) {      // 128-byte rows, 16-byte slot XOR (row >> 1) & 7 (sk_frag)
  return *reinterpret_cast<const bf16x8_t*>(img + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
}
template <int N>
SF_DEVICE void qd_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
SF_DEVICE void qd_stats(const bf16x8_t& f, float& s1, float& s2) {   // see sk_stats (sf_gemm_skinny.hip)
  typedef __attribute__((ext_vector_type(2))) __bf16 v2bf;
  typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
  const v8bf h = __builtin_bit_cast(v8bf, f);
  const v2bf one = {(__bf16)1.0f, (__bf16)1.0f};
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

__global__ __launch_bounds__(QD_THREADS) void sf_qkv_decode_kernel(SfQkvDecodeArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int h = blockIdx.x % p.heads, rt = blockIdx.x / p.heads;
  const int m0 = rt * QD_ROWS;
  const int D = p.heads * 64;
  const int t_past = p.t_past_dev ? *p.t_past_dev : p.t_past;     // frames already cached = index of the new frame
  const int Tk = t_past + 1;

  // ---- this wave's attention task: cached K / V rows requested up front (they do not depend on the projection) --------
  const int row = m0 + wave;                       // token row of the task = (stream b, patch n)
  const bool live = row < p.M;
  const int rowc = live ? row : p.M - 1;
  const int b = rowc / p.N, n = rowc % p.N;
  const char* cache = reinterpret_cast<const char*>(p.cache);
  u32x4_t kv[8], vv[8];
  {
    int key = lane < Tk ? lane : Tk - 1;           // key t_past itself is this frame: its registers are replaced from LDS below
    const size_t offk = ((((size_t)b * p.cap + key) * p.N + n) * (size_t)(3 * D) + D + h * 64) * 2;
#pragma unroll
    for (int c = 0; c < 8; ++c) kv[c] = *reinterpret_cast<const u32x4_t*>(cache + offk + c * 16);
    const int tsub = lane >> 3, ch = lane & 7;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int kk = i * 8 + tsub;
      kk = kk < Tk ? kk : Tk - 1;
      const size_t offv = ((((size_t)b * p.cap + kk) * p.N + n) * (size_t)(3 * D) + 2 * D + h * 64 + ch * 8) * 2;
      vv[i] = *reinterpret_cast<const u32x4_t*>(cache + offv);
    }
  }

  // ---- projection: [16 rows] x [q_h | k_h | v_h] over K, LayerNorm folded ------------------------------------------------
  const int K = p.K, nkt = K / QD_BK;
  f32x4_t acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float ln1 = 0.f, ln2 = 0.f;
  if (wave >= 4 && wave < 8) {
    // loader waves: stage image rows 0..15 = A rows, 16 + 64 c + r = weight row c * D + h * 64 + r (c = q, k, v)
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a_hi, 0, (unsigned)p.M * (unsigned)K * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_hi, 0, 3u * (unsigned)D * (unsigned)K * 2u, 0x00020000);
    const int ltid = tid - 256;
    int off[QD_NI];
    bool is_a[QD_NI];
#pragma unroll
    for (int i = 0; i < QD_NI; ++i) {
      const int c = i * 256 + ltid;
      int r = c >> 3;
      const int slot = c & 7;
      const int kc = slot ^ ((r >> 1) & 7);
      if (r >= 208) r = 207;                       // padding rows of the last instruction re-read a valid row
      is_a[i] = r < 16;
      int gr;
      if (r < 16) {
        gr = m0 + r;
        gr = gr < p.M ? gr : p.M - 1;
      } else {
        const int wr = r - 16;
        gr = (wr >> 6) * D + h * 64 + (wr & 63);
      }
      off[i] = (int)(((unsigned)gr * (unsigned)K + kc * 8) * 2u);
    }
    const int dma_lds = (wave - 4) * 1024;
    auto issue = [&](int kt) {
      char* dst = smem + (kt % QD_STAGES) * QD_STAGE_BYTES + dma_lds;
      const int kof = kt * QD_BK * 2;
#pragma unroll
      for (int i = 0; i < QD_NI; ++i) {
        // instruction 0 covers stage rows 0..31: loader waves 0 / 1 hold the 16 A rows, waves 2 / 3 weight rows — wave-uniform
        if (i == 0) {
          if (is_a[0]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (qd_lptr_t)(dst), 16, (int)off[0], kof, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (qd_lptr_t)(dst), 16, (int)off[0], kof, 0, 2);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (qd_lptr_t)(dst + i * 4096), 16, (int)off[i], kof, 0, 2);
        }
      }
    };
    // QD_NI load instructions per stage and wave, on top of the 16 cache loads above (older: they retire first, a counted
    // wait only becomes stricter while they are in flight)
    constexpr int PER = QD_NI;
    for (int s = 0; s < QD_STAGES - 1 && s < nkt; ++s) issue(s);
    for (int kt = 0; kt < nkt; ++kt) {
      const int later = min(nkt - 1 - kt, QD_STAGES - 2);
      if (later >= 2) qd_wait<2 * PER>(); else if (later == 1) qd_wait<PER>(); else qd_wait<0>();
      __builtin_amdgcn_s_barrier();
      if (kt + QD_STAGES - 1 < nkt) issue(kt + QD_STAGES - 1);
    }
  } else if (wave < 4) {
    for (int kt = 0; kt < nkt; ++kt) {
      __builtin_amdgcn_s_barrier();
      const char* img = smem + (kt % QD_STAGES) * QD_STAGE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int kc = ks * 4 + g;
        const bf16x8_t af = qd_frag(img, l15, kc);
        qd_stats(af, ln1, ln2);
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = qd_mfma(qd_frag(img, 16 + j * 64 + wave * 16 + l15, kc), af, acc[j]);
      }
    }
    // lane (l15, g) holds row l15, dims wave * 16 + 4 g .. + 3 of q (j = 0), k (1), v (2)
    ln1 += __shfl_xor(ln1, 16, 64); ln1 += __shfl_xor(ln1, 32, 64);
    ln2 += __shfl_xor(ln2, 16, 64); ln2 += __shfl_xor(ln2, 32, 64);
    const float mean = ln1 / (float)K;
    const float rstd = __builtin_amdgcn_rsqf(fmaxf(ln2 / (float)K - mean * mean, 0.f) + p.ln_eps);
    const int mrow = m0 + l15;
    const int mb = (mrow < p.M ? mrow : p.M - 1) / p.N, mn = (mrow < p.M ? mrow : p.M - 1) % p.N;
    bf16_t* crow = p.cache + (((size_t)mb * p.cap + t_past) * p.N + mn) * (size_t)(3 * D);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = j * D + h * 64 + wave * 16 + g * 4;
      f32x4_t v = rstd * (acc[j] - mean * *reinterpret_cast<const f32x4_t*>(p.ln_s + col));
      if (p.bias) v += *reinterpret_cast<const f32x4_t*>(p.bias + col);
      const u32x2_t pk = (u32x2_t){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
      *reinterpret_cast<u32x2_t*>(smem + QD_TILE_OFF + l15 * 384 + (j * 64 + wave * 16 + g * 4) * 2) = pk;
      if (mrow < p.M) *reinterpret_cast<u32x2_t*>(crow + col) = pk;     // the cache row of the new frame (q slot included: same layout as the GEMM epilogue wrote)
    }
  } else {
    for (int kt = 0; kt < nkt; ++kt) __builtin_amdgcn_s_barrier();
  }
  __syncthreads();
  if (!live) return;

  // ---- single-query attention of this wave's task (sf_temporal_decode_kernel<false, 1>), new key / value from LDS ---------
  const char* tile = smem + QD_TILE_OFF + wave * 384;
  u32x4_t qv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) qv[c] = *reinterpret_cast<const u32x4_t*>(tile + c * 16);
  if (lane >= t_past) {          // the new key (and the masked lanes behind it: the cache row they re-read may be uninitialised)
#pragma unroll
    for (int c = 0; c < 8; ++c) kv[c] = *reinterpret_cast<const u32x4_t*>(tile + 128 + c * 16);
  }
  const int tsub = lane >> 3, ch = lane & 7;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i * 8 + tsub >= t_past) vv[i] = *reinterpret_cast<const u32x4_t*>(tile + 256 + ch * 16);      // keys past t_past carry probability 0
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a0 = fmaf(bf2f(qv[c][j] & 0xffffu), bf2f(kv[c][j] & 0xffffu), a0);
      a1 = fmaf(__uint_as_float(qv[c][j] & 0xffff0000u), __uint_as_float(kv[c][j] & 0xffff0000u), a1);
    }
  const float sc = lane < Tk ? (a0 + a1) : -INFINITY;
  float mx = sc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  const float pr = __builtin_amdgcn_exp2f((sc - mx) * (p.scale * 1.44269504088896340736f));
  const float inv = 1.0f / wave_sum(pr);
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float pt = __shfl(pr, i * 8 + tsub, 64);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[2 * j] = fmaf(pt, bf2f(vv[i][j] & 0xffffu), o[2 * j]);
      o[2 * j + 1] = fmaf(pt, __uint_as_float(vv[i][j] & 0xffff0000u), o[2 * j + 1]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    o[j] += __shfl_xor(o[j], 8, 64);
    o[j] += __shfl_xor(o[j], 16, 64);
    o[j] += __shfl_xor(o[j], 32, 64);
  }
  if (tsub == 0) {
    unsigned int hb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) hb[j] = f2bf(o[j] * inv);
    *reinterpret_cast<u32x4_t*>(p.ctx_hi + (size_t)row * D + h * 64 + ch * 8) =
        (u32x4_t){hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16)};
  }
}

bool sf_qkv_decode_supported(const SfQkvDecodeArgs& a) {
  static const bool off = getenv("SF_DISABLE_STREAM_FUSED") != nullptr;
  if (off) return false;
  if (a.heads <= 0 || a.K % QD_BK || a.K < QD_BK * 2 || a.M <= 0 || a.N <= 0 || a.cap <= 0 || a.cap > 64) return false;
  if (!a.t_past_dev && (a.t_past < 0 || a.t_past >= a.cap)) return false;
  if ((size_t)a.M * a.K * 2 >= ((size_t)1 << 32)) return false;
  return a.a_hi && a.w_hi && a.ln_s && a.cache && a.ctx_hi;
}

hipError_t sf_launch_qkv_decode(const SfQkvDecodeArgs& a, hipStream_t s) {
  if (!sf_qkv_decode_supported(a)) return hipErrorInvalidValue;
  static SfPerDeviceOnce attr_set;
  if (attr_set.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_qkv_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, QD_LDS);
  const int row_tiles = (a.M + QD_ROWS - 1) / QD_ROWS;
  hipLaunchKernelGGL(sf_qkv_decode_kernel, dim3(row_tiles * a.heads), dim3(QD_THREADS), QD_LDS, s, a);
  return hipGetLastError();
}
