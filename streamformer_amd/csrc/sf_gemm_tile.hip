// "Tile" MFMA GEMM for ONE TO FOUR CLIPS per call (2560 < M <= ~12.5k token rows; README.md:55-71 is B = 1, M = 3136):
//   C[M,N] = A[M,K] * W[N,K]^T with the fused epilogues of the forward (bf16 mode)
//
// Why another GEMM: at M = 3136 the 256^2 kernel fills 117-156 of 256 CUs and the N = 768 residual projections fell onto
// 32-row panel tiles that stream a [384 x K] weight slab per 25 token rows (76 us for the MLP down-projection, 49 % of
// the forward).  A launch at these sizes is ONE ROUND of the chip and is bound by the bytes a CU can pull from L2 into
// LDS (~95 GB/s per CU, tools/fill_lab), so the tile is picked per shape to (a) give >= 196 workgroups and (b) minimise
// (BM + BN) per MFMA:   128 x 96 / 256 x 96 / 256 x 192 for N = 768,   128 x 256 / 128 x 384 / 256 x 256 for the
// LayerNorm-folded consumers (qkv, MLP-up).
//
// Structure (gfx950): 8 CONSUMER waves = WM x WN, a wave owns MT x NT MFMA 16x16x32 tiles, + 4 LOADER waves (one per
// SIMD) that do nothing but move operands L2 -> LDS by buffer_load ... lds (32-bit lane offsets + scalar K offset) into
// a ring of [BM + BN rows][BK] stage images with the XOR-swizzled 16-byte slots of the skinny / panel kernels; one
// barrier per K-tile, counted vmcnt in the loaders only.  Why loader waves: with the DMA instructions inside the
// consumer waves (first version) every shape ran at the same ~50 GB/s of ingest per CU — an LDS-DMA instruction (1 KB
// per wave) costs 100-185 issue cycles between ds_reads and MFMAs but ~20 in a wave that issues nothing else
// (MI355X_MICROARCH.md, ldsdma-fill), and at these tile sizes the launch is bound by exactly that ingest.
// LayerNorm: the small-M fold (sf_gemm_skinny.hip) — consumers derive sum x / sum x^2 of their rows from the A
// fragments they feed to the MFMAs anyway (the WN waves that share a row block split its m-tiles), producers add the
// bf16 copy of the new residual rows; no statistics buffer.
// Epilogue: the C tile leaves through LDS, 16 WM rows at a time, as whole rows with 16-byte accesses: fp32 residual
// read-modify-write + bf16 copy, bf16 (+ LayerNorm finish, bias, erf-GELU), embedding table add.
#include "sf_common.h"
#include "sf_switches.h"
#include <cstdlib>

#define TL_CONSUMERS 512          // 8 MFMA waves
#define TL_LOADERS 256            // 4 LDS-DMA waves
#define TL_THREADS (TL_CONSUMERS + TL_LOADERS)

typedef __attribute__((address_space(3))) void* tl_lptr_t;

SF_DEVICE f32x4_t tl_mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}
// Barrier between the staging passes of the epilogue: orders LDS traffic only.  __syncthreads() also waits for every
// outstanding GLOBAL store of the wave (vmcnt(0)): with one pass per 16 WM rows that put a store round trip into each pass
// (MLP-up at M = 3136: 14 of 34 us were epilogue).
SF_DEVICE void tl_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N>
SF_DEVICE void tl_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// stage image row = BK bf16; 16-byte slot XOR: BK = 64 (8 slots): (row >> 1) & 7, BK = 32 (4 slots): sf_swz64(row) —
// conflict-free for the 16-lane groups of ds_read_b128 (same images as sk_frag / rd32)
template <int BK>
SF_DEVICE int tl_swz(int row) { return BK == 64 ? ((row >> 1) & 7) : sf_swz64(row); }
template <int BK>
SF_DEVICE bf16x8_t tl_frag(const char* img, int row, int kc) {
  return *reinterpret_cast<const bf16x8_t*>(img + row * (BK * 2) + ((kc ^ tl_swz<BK>(row)) << 4));
}
SF_DEVICE void tl_stats(const bf16x8_t& f, float& s1, float& s2) {      // see sk_stats (sf_gemm_skinny.hip)
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

template <int MT, int NT, int WM, int WN, int BK, int STAGES>
struct TlCfg {
  static constexpr int BM = 16 * MT * WM, BN = 16 * NT * WN;
  static constexpr int ROWS_PER_INST = TL_LOADERS * 16 / (BK * 2);            // stage rows one DMA instruction of the 4 loader waves covers
  static constexpr int NI_A = BM / ROWS_PER_INST, NI_W = (BN + ROWS_PER_INST - 1) / ROWS_PER_INST;
  static constexpr int NI = NI_A + NI_W;
  static constexpr int STAGE_BYTES = NI * TL_LOADERS * 16;
  static constexpr int STATS_OFF = STAGES * STAGE_BYTES;                      // the epilogue stages the C tile inside the idle ring
  static constexpr int LDS_BYTES = STATS_OFF + BM * 8;
  static constexpr int PART_OFF = LDS_BYTES;                                  // statistics producers: [BM rows][BN / 8] partial pairs behind it
  static constexpr int PART_BYTES = BM * (BN / 8) * 8;
  static_assert(BM % ROWS_PER_INST == 0, "A rows must fill whole DMA instructions");
  static_assert(WM * WN * 64 == TL_CONSUMERS, "8 consumer waves");
  static_assert(NI * (STAGES - 2) <= 63, "vmcnt is 6 bits");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(BN % 8 == 0, "copy-out handles 8 columns per thread");
};

template <int MT, int NT, int WM, int WN, int BK, int STAGES, int EPI, bool LNF>
__global__ __launch_bounds__(TL_THREADS) void sf_gemm_tile_kernel(SfGemmArgs p, int tiles_n, int ntiles, int per_xcd, int lab) {
  // lab (SF_TILE_LAB, measurements only; 0 in the product): 1 loader waves at raised priority, 2 consumers read their fragments but issue
  // no MFMA, 4 consumers only take the barriers (pure ingest), 8 no epilogue
  using C = TlCfg<MT, NT, WM, WN, BK, STAGES>;
  constexpr int BM = C::BM, BN = C::BN, NI = C::NI, NI_A = C::NI_A;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const bool loader = wave >= TL_CONSUMERS / 64;
  const int wm = (wave & 7) / WN, wn = (wave & 7) % WN;
  // XCD-aware tile order: block b runs on XCD b % 8 (observed placement, speed only): XCD x takes the contiguous logical
  // tiles [x * per_xcd, (x + 1) * per_xcd) — the column tiles of a row panel share an L2, the A panel is fetched once
  const int t = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (t >= ntiles) return;
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
  const int K = p.K;
  const int nkt = K / BK;

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  constexpr int NST = (MT + WN - 1) / WN;          // m-tiles whose LayerNorm sums this wave accumulates (i = wn, wn + WN, ...)
  float ln1[NST], ln2[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) ln1[i] = ln2[i] = 0.f;

  if (loader) {
    // ---- loader wave: stage kt + STAGES - 1 goes out right behind barrier kt (every consumer has finished stage kt - 1 by
    // then); the wave arrives at barrier kt + 1 once its share of stage kt + 1 has landed ---------------------------------
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a_hi, 0, (unsigned)p.M * (unsigned)K * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_hi, 0, (unsigned)p.N * (unsigned)K * 2u, 0x00020000);
    const int ltid = tid - TL_CONSUMERS;
    int off[NI];             // (the explicit (int) casts below are load-bearing: hipcc 7.2 silently drops the kernel host stub when
                             //  an element of an array of dependent size is passed to the builtin as it is)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      constexpr int SLOTS = BK / 8;
      const int c = i * TL_LOADERS + ltid;
      const int row = c / SLOTS, slot = c % SLOTS;               // row of the stage image (A rows, then W rows)
      const int kc = slot ^ tl_swz<BK>(row);
      int gr;
      if (i < NI_A) {
        gr = m0 + row;
        gr = gr < p.M ? gr : p.M - 1;
      } else {
        gr = n0 + row - BM;
        gr = gr < p.N ? gr : p.N - 1;                            // rows of the padded last instruction re-read a valid row
      }
      off[i] = (int)(((unsigned)gr * (unsigned)K + kc * 8) * 2u);
    }
    const int dma_lds = (wave - TL_CONSUMERS / 64) * 1024;
    auto issue = [&](int kt) {
      char* dst = smem + (kt % STAGES) * C::STAGE_BYTES + dma_lds;
      const int kof = kt * BK * 2;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (i < NI_A) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (tl_lptr_t)(dst + i * (TL_LOADERS * 16)), 16, (int)off[i], kof, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (tl_lptr_t)(dst + i * (TL_LOADERS * 16)), 16, (int)off[i], kof, 0, 0);
      }
    };
    if (lab & 1) __builtin_amdgcn_s_setprio(3);
    for (int s = 0; s < STAGES - 1 && s < nkt; ++s) issue(s);
    for (int kt = 0; kt < nkt; ++kt) {
      const int later = min(nkt - 1 - kt, STAGES - 2);            // stages that may stay in flight behind stage kt
      if (STAGES >= 5 && later >= 3) tl_wait<(STAGES >= 5 ? 3 : 0) * NI>();
      else if (later >= 2) tl_wait<2 * NI>(); else if (later == 1) tl_wait<NI>(); else tl_wait<0>();
      __builtin_amdgcn_s_barrier();
      if (kt + STAGES - 1 < nkt) issue(kt + STAGES - 1);
    }
  } else {
    for (int kt = 0; kt < nkt; ++kt) {
      __builtin_amdgcn_s_barrier();
      if (lab & 4) continue;
      const char* img = smem + (kt % STAGES) * C::STAGE_BYTES;
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        const int kc = ks * 4 + g;
        bf16x8_t af[MT], wf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = tl_frag<BK>(img, BM + (wn * NT + j) * 16 + l15, kc);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          af[i] = tl_frag<BK>(img, (wm * MT + i) * 16 + l15, kc);
          if (LNF && (i % WN) == wn) tl_stats(af[i], ln1[i / WN], ln2[i / WN]);   // wave wn of a row block owns m-tiles i = wn (mod WN)
        }
        if (lab & 2) {                                           // keep the reads alive without the matrix pipe
#pragma unroll
          for (int i = 0; i < MT; ++i) acc[i][0][0] += __builtin_bit_cast(float, (int)af[i][0] ^ (int)wf[i % NT][1]);
          continue;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = tl_mfma(wf[j], af[i], acc[i][j]);
      }
    }
  }
  if (lab & 8) {
    if (!loader && acc[0][0][0] == 123.456f) p.out_hi[tid] = 1;   // keeps the accumulators live
    return;
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------------
  __syncthreads();                                               // every wave is done with the ring
  float2* st = reinterpret_cast<float2*>(smem + C::STATS_OFF);   // [BM] {sum x, sum x^2} of the tile's rows
  if (LNF) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (!loader && (i % WN) == wn) {
        float s1 = ln1[i / WN], s2 = ln2[i / WN];
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        if (g == 0) st[(wm * MT + i) * 16 + l15] = make_float2(s1, s2);
      }
    }
    __syncthreads();
  }
  f32x4_t bias4[NT], lns4[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + (wn * NT + j) * 16 + g * 4;
    bias4[j] = p.bias ? *reinterpret_cast<const f32x4_t*>(p.bias + n) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    lns4[j] = LNF ? *reinterpret_cast<const f32x4_t*>(p.ln_s + n) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  // Staging: as many m-tiles of every wave per pass as the (idle) ring holds — the whole tile for every shipped shape but
  // 256 x 192 fp32 — as fp32 for the residual / embedding epilogues and as bf16 (half the LDS bytes) for the bf16 outputs.
  // One pass = staging writes, ONE barrier, whole-row copy-out; with one 16-row m-tile per pass (first version) the MLP-up
  // epilogue cost 14 of 34 us at M = 3136 (SF_TILE_LAB=8).
  constexpr bool F32OUT = (EPI == SF_EPI_RESID_F32 || EPI == SF_EPI_EMBED_F32 || EPI == SF_EPI_F32);
  constexpr int ES = F32OUT ? 4 : 2;
  constexpr int PITCH = BN * ES + 16;
  constexpr int RING = STAGES * C::STAGE_BYTES;
  constexpr int MTP_FIT = RING / (16 * WM * PITCH);
  constexpr int MTP = MTP_FIT >= MT ? MT : (MTP_FIT >= 1 ? MTP_FIT : 1);      // m-tiles per pass
  static_assert(16 * WM * PITCH * MTP <= C::STATS_OFF, "staging must stay below the statistics block");
  constexpr int CH = BN / 8;                                     // 8-column chunks per row
  float2* part = reinterpret_cast<float2*>(smem + C::PART_OFF);  // statistics producers only (the launch adds PART_BYTES of LDS)
  const float inv_k = 1.0f / (float)K;
#pragma unroll
  for (int q0 = 0; q0 < MT; q0 += MTP) {
    if (!loader) {
#pragma unroll
      for (int qq = 0; qq < MTP; ++qq) {
        const int q = q0 + qq;
        if (q >= MT) break;
        float mean = 0.f, rstd = 1.f;
        if (LNF) {
          const float2 sv = st[(wm * MT + q) * 16 + l15];
          mean = sv.x * inv_k;
          rstd = __builtin_amdgcn_rsqf(fmaxf(sv.y * inv_k - mean * mean, 0.f) + p.ln_eps);
        }
        char* srow = smem + ((wm * MTP + qq) * 16 + l15) * PITCH;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          f32x4_t v = acc[q][j];
          if (LNF) v = rstd * (v - mean * lns4[j]);
          v += bias4[j];
          if (EPI == SF_EPI_ACT_BF16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act_bf16(v[e], p.act);
          }
          const int col = (wn * NT + j) * 16 + g * 4;
          if (F32OUT) *reinterpret_cast<f32x4_t*>(srow + col * 4) = v;
          else *reinterpret_cast<u32x2_t*>(srow + col * 2) = (u32x2_t){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        }
      }
    }
    tl_lds_barrier();
    const int mtp = (MT - q0) < MTP ? (MT - q0) : MTP;            // m-tiles staged in this pass
    const int items = 16 * WM * mtp * CH;
    for (int it = tid; it < items; it += TL_THREADS) {
      const int rr = it / CH, c8 = it % CH;                      // rr = (wm, qq, l) packed as ((wm * mtp + qq) * 16 + l) over the staged rows
      const int w_ = rr / (16 * mtp), qq = (rr >> 4) % mtp, l = rr & 15;
      const int m = m0 + (w_ * MT + q0 + qq) * 16 + l;
      if (m >= p.M) continue;
      const int n = n0 + c8 * 8;
      const char* srow = smem + ((w_ * MTP + qq) * 16 + l) * PITCH;
      size_t orow = (size_t)m;
      if (p.grp_rows > 0) orow = sf_out_row(p, m);
      const size_t o = orow * (size_t)p.ldc + n;
      if (F32OUT) {
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(srow + c8 * 32);
        const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(srow + c8 * 32 + 16);
        f32x4_t x0 = v0, x1 = v1;
        if (EPI == SF_EPI_RESID_F32) {
          const float* rp = p.resid + o;
          x0 = *reinterpret_cast<const f32x4_t*>(rp) + p.alpha * v0;
          x1 = *reinterpret_cast<const f32x4_t*>(rp + 4) + p.alpha * v1;
        } else if (EPI == SF_EPI_EMBED_F32) {
          const float* pe = p.pos + (size_t)(m % p.Np) * p.N + n;
          const float* te = p.time_rows + (size_t)((m / p.Np) % p.Tn) * p.N + n;
          x0 = v0 + *reinterpret_cast<const f32x4_t*>(pe) + *reinterpret_cast<const f32x4_t*>(te);
          x1 = v1 + *reinterpret_cast<const f32x4_t*>(pe + 4) + *reinterpret_cast<const f32x4_t*>(te + 4);
        }
        *reinterpret_cast<f32x4_t*>(p.out_f32 + o) = x0;
        *reinterpret_cast<f32x4_t*>(p.out_f32 + o + 4) = x1;
        if ((EPI == SF_EPI_RESID_F32 || EPI == SF_EPI_EMBED_F32) && p.ln_stats_out)      // row-statistics producer: this chunk's share of {sum x, sum x^2}
          part[rr * CH + c8] = make_float2(((x0[0] + x0[1]) + (x0[2] + x0[3])) + ((x1[0] + x1[1]) + (x1[2] + x1[3])),
                                           ((x0[0] * x0[0] + x0[1] * x0[1]) + (x0[2] * x0[2] + x0[3] * x0[3])) +
                                               ((x1[0] * x1[0] + x1[1] * x1[1]) + (x1[2] * x1[2] + x1[3] * x1[3])));
        if (EPI != SF_EPI_F32 && p.out_hi)      // LayerNorm-fold producer: bf16 copy of the new residual rows
          *reinterpret_cast<u32x4_t*>(p.out_hi + o) =
              (u32x4_t){pack_bf2(x0[0], x0[1]), pack_bf2(x0[2], x0[3]), pack_bf2(x1[0], x1[1]), pack_bf2(x1[2], x1[3])};
      } else {
        *reinterpret_cast<u32x4_t*>(p.out_hi + o) = *reinterpret_cast<const u32x4_t*>(srow + c8 * 16);
      }
    }
    if ((EPI == SF_EPI_RESID_F32 || EPI == SF_EPI_EMBED_F32) && p.ln_stats_out) {
      // LayerNorm fold with a statistics buffer (the 256^2 kernel as consumer, M = 6272): one {sum x, sum x^2} pair per row and
      // column tile (N / BN <= 4 pairs per row: SfGemmArgs::ln_stats_wide), the CH chunk partials of a row summed in a fixed order
      tl_lds_barrier();
      for (int rr = tid; rr < 16 * WM * mtp; rr += TL_THREADS) {
        const int w_ = rr / (16 * mtp), qq = (rr >> 4) % mtp, l = rr & 15;
        const int m = m0 + (w_ * MT + q0 + qq) * 16 + l;
        if (m >= p.M) continue;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
        for (int c = 0; c < CH; ++c) {
          const float2 v = part[rr * CH + c];
          s1 += v.x; s2 += v.y;
        }
        *reinterpret_cast<float2*>(p.ln_stats_out + (size_t)m * 8 + (size_t)(n0 / BN) * 2) = make_float2(s1, s2);
      }
    }
    if (q0 + MTP < MT) tl_lds_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// host side: configuration table and dispatch
// ------------------------------------------------------------------------------------------------
static int tile_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus < 8) cus = 256;
  }
  return cus;
}
int sf_tile_min_rows() {
  const int m = sf_sw(SW_TILE_MIN_M) ? atoi(sf_sw(SW_TILE_MIN_M)) : 2560;      // below: skinny / 64 x 64 kernels
  return m;
}
int sf_tile_max_rows() {
  // above: panel / 256^2 kernels.  Measured (us per launch, tile vs panel / 256^2; tools/tile_lab.py): M = 3136 qkv 17 / 24,
  // MLP-up 34 / 29, out-proj 12 / 18, MLP-down 26 / 57 -> 1.83 against 2.64 ms per clip; M = 6272: 31 / 24, 63 / 55, 19 / 23,
  // 39 / 64 -> 3.02 against 3.36 ms for two clips (the narrow producers carry it; the two LayerNorm folds — in-kernel
  // statistics here, statistics buffer on the 256^2 kernel — do not mix, so the consumers come along); M = 12544: level or
  // slower everywhere.
  const int m = sf_sw(SW_TILE_MAX_M) ? atoi(sf_sw(SW_TILE_MAX_M)) : 6272;
  return m;
}

int sf_tile_fold_min_rows() {
  // from here up to sf_tile_max_rows() the LayerNorm-folded consumers (qkv, MLP-up) run on the 256^2 kernel with a statistics buffer that
  // the tile producers fill (two clips: 24 / 55 us against 31 / 63 us on the in-kernel-statistics tiles); below, the tiles win (M = 3136:
  // 17 / 34 against 24 / 29).  Whole forward, one box: M = 3920 2.21 (tiles) / 2.30 ms, 4312 2.60 / 2.41, 4704 2.57 / 2.46, 5488 2.89 / 2.69,
  // 6272 3.03 / 2.80 (profiles/r05_two_clips_ab.txt)
  return sf_sw(SW_TILE_FOLD_MIN_M) ? atoi(sf_sw(SW_TILE_FOLD_MIN_M)) : 4312;
}

struct TlShape { int bm, bn, id; };
// candidates in order of preference at equal cost; id selects the instantiation
static const TlShape kShapes[] = {
    {128, 96, 0}, {256, 96, 1}, {256, 192, 2},      // N = 768 residual producers / embedding
    {128, 256, 3}, {128, 384, 4}, {256, 256, 5},    // wide consumers
    {128, 192, 6},                                  // N = 768 residual producers that also emit the row statistics (four pairs per row)
};
static bool shape_takes(const TlShape& sh, const SfGemmArgs& a) {
  if (a.N % sh.bn) return false;
  // statistics producers: the consumer sums four pairs per row, and the partial sums need BM x BN bytes of LDS behind the ring
  if ((a.ln_stats_out != nullptr) != (sh.id == 6)) return false;
  if (sh.id == 6 && ((a.epi != SF_EPI_RESID_F32 && a.epi != SF_EPI_EMBED_F32) || a.N / sh.bn > 4)) return false;
  const bool lnf = a.ln_inkernel != 0;
  const bool wide = sh.id >= 3 && sh.id != 6;
  if (lnf && (!wide || sh.id == 5)) return false;                   // LayerNorm-folded epilogues: the 128-row wide tiles (256 x 256 would spill)
  if ((a.epi == SF_EPI_RESID_F32 || a.epi == SF_EPI_EMBED_F32 || a.epi == SF_EPI_F32) && wide) return false;
  return true;
}
// cycles per 32-deep K step of one workgroup: MFMA issue on 4 SIMDs vs L2 -> LDS ingest at ~47 B / cycle / CU
static double shape_cost(const TlShape& sh, const SfGemmArgs& a) {
  const int tiles = ((a.M + sh.bm - 1) / sh.bm) * (a.N / sh.bn);
  const int rounds = (tiles + tile_cus() - 1) / tile_cus();
  const double mfma = (double)sh.bm * sh.bn / 64.0;
  const double ingest = (double)(sh.bm + sh.bn) * 64.0 / 47.0;
  return rounds * (mfma > ingest ? mfma : ingest) + 40.0;
}
static const TlShape* pick_shape(const SfGemmArgs& a) {
  const TlShape* best = nullptr;
  double bc = 0;
  for (const TlShape& sh : kShapes) {
    if (!shape_takes(sh, a)) continue;
    const double c = shape_cost(sh, a);
    if (!best || c < bc) { best = &sh; bc = c; }
  }
  return best;
}

bool sf_gemm_tile_supported(const SfGemmArgs& a, bool split) {
  const bool off = sf_sw(SW_DISABLE_GEMM_TILE) != nullptr;
  if (off || split || a.a_lo || a.out_lo || a.aux_mode || a.ln_stats || a.resid_mod > 0) return false;
  if (a.ln_stats_out && (!a.ln_stats_wide || (a.epi != SF_EPI_RESID_F32 && a.epi != SF_EPI_EMBED_F32) || !a.out_hi || a.M < sf_tile_fold_min_rows())) return false;
  if (a.M <= sf_tile_min_rows() || a.M > sf_tile_max_rows()) return false;
  if (a.K < 128 || (a.K % 64) || (a.ldc % 8) || (a.N % 8)) return false;
  if ((size_t)a.M * a.K * 2 >= ((size_t)1 << 32) || (size_t)a.N * a.K * 2 >= ((size_t)1 << 32)) return false;
  if (a.ln_inkernel && (!a.ln_s || (a.epi != SF_EPI_BF16 && a.epi != SF_EPI_ACT_BF16))) return false;
  switch (a.epi) {
    case SF_EPI_F32: if (!a.out_f32 || a.grp_rows > 0) return false; break;
    case SF_EPI_RESID_F32: if (!a.out_f32 || !a.resid || a.grp_rows > 0) return false; break;
    case SF_EPI_EMBED_F32: if (!a.out_f32 || !a.pos || !a.time_rows || a.grp_rows > 0) return false; break;
    case SF_EPI_BF16: case SF_EPI_ACT_BF16: if (!a.out_hi) return false; break;
    default: return false;
  }
  return pick_shape(a) != nullptr;
}

template <int MT, int NT, int WM, int WN, int BK, int STAGES, int EPI, bool LNF>
static hipError_t tl_go(const SfGemmArgs& a, hipStream_t s) {
  using C = TlCfg<MT, NT, WM, WN, BK, STAGES>;
  constexpr bool PRODUCER = EPI == SF_EPI_RESID_F32 || EPI == SF_EPI_EMBED_F32;
  const bool stats = PRODUCER && a.ln_stats_out != nullptr;
  constexpr int LDS_MAX = (PRODUCER && C::LDS_BYTES + C::PART_BYTES <= 160 * 1024) ? C::LDS_BYTES + C::PART_BYTES : C::LDS_BYTES;
  const int lds = C::LDS_BYTES + (stats ? C::PART_BYTES : 0);
  if (lds > LDS_MAX) return hipErrorInvalidValue;
  static SfPerDeviceOnce attr_set;
  if (attr_set.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_tile_kernel<MT, NT, WM, WN, BK, STAGES, EPI, LNF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
  const int tiles_n = a.N / C::BN, tiles_m = (a.M + C::BM - 1) / C::BM;
  const int ntiles = tiles_n * tiles_m;
  const int per_xcd = (ntiles + 7) / 8;
  static const int lab_env = SF_LAB_SWITCH("SF_TILE_LAB");      // lab builds only
  hipLaunchKernelGGL((sf_gemm_tile_kernel<MT, NT, WM, WN, BK, STAGES, EPI, LNF>), dim3(per_xcd * 8), dim3(TL_THREADS), lds, s,
                     a, tiles_n, ntiles, per_xcd, lab_env);
  return hipGetLastError();
}

// producers (narrow tiles): RESID / EMBED / F32;  consumers (wide tiles): BF16 / ACT_BF16 with or without the LayerNorm fold
template <int MT, int NT, int WM, int WN, int BK, int STAGES>
static hipError_t tl_narrow(const SfGemmArgs& a, hipStream_t s) {
  switch (a.epi) {
    case SF_EPI_RESID_F32: return tl_go<MT, NT, WM, WN, BK, STAGES, SF_EPI_RESID_F32, false>(a, s);
    case SF_EPI_EMBED_F32: return tl_go<MT, NT, WM, WN, BK, STAGES, SF_EPI_EMBED_F32, false>(a, s);
    case SF_EPI_F32: return tl_go<MT, NT, WM, WN, BK, STAGES, SF_EPI_F32, false>(a, s);
    case SF_EPI_BF16: return tl_go<MT, NT, WM, WN, BK, STAGES, SF_EPI_BF16, false>(a, s);
    case SF_EPI_ACT_BF16: return tl_go<MT, NT, WM, WN, BK, STAGES, SF_EPI_ACT_BF16, false>(a, s);
    default: return hipErrorInvalidValue;
  }
}
template <int MT, int NT, int WM, int WN, int BK, int STAGES>
static hipError_t tl_wide_plain(const SfGemmArgs& a, hipStream_t s) {
  if (a.epi == SF_EPI_BF16) return tl_go<MT, NT, WM, WN, BK, STAGES, SF_EPI_BF16, false>(a, s);
  if (a.epi == SF_EPI_ACT_BF16) return tl_go<MT, NT, WM, WN, BK, STAGES, SF_EPI_ACT_BF16, false>(a, s);
  return hipErrorInvalidValue;
}
template <int MT, int NT, int WM, int WN, int BK, int STAGES>
static hipError_t tl_wide(const SfGemmArgs& a, hipStream_t s) {
  const bool lnf = a.ln_inkernel != 0;
  if (a.epi == SF_EPI_BF16) return lnf ? tl_go<MT, NT, WM, WN, BK, STAGES, SF_EPI_BF16, true>(a, s) : tl_go<MT, NT, WM, WN, BK, STAGES, SF_EPI_BF16, false>(a, s);
  if (a.epi == SF_EPI_ACT_BF16) return lnf ? tl_go<MT, NT, WM, WN, BK, STAGES, SF_EPI_ACT_BF16, true>(a, s) : tl_go<MT, NT, WM, WN, BK, STAGES, SF_EPI_ACT_BF16, false>(a, s);
  return hipErrorInvalidValue;
}

hipError_t sf_launch_gemm_tile(const SfGemmArgs& a, hipStream_t s) {
  if (!sf_gemm_tile_supported(a, false)) return hipErrorInvalidValue;
  const TlShape* sh = pick_shape(a);
  const int force = sf_sw(SW_TILE_SHAPE) ? atoi(sf_sw(SW_TILE_SHAPE)) : -1;      // lab switch: force a candidate id
  int id = sh->id;
  if (force >= 0 && force < 7 && shape_takes(kShapes[force], a)) id = force;
  switch (id) {
    //                      MT NT WM WN BK STAGES          stage image                  LDS
    case 0: return tl_narrow<2, 3, 4, 2, 64, 5>(a, s);   // 128 x  96: 224 rows x 128 B = 28 KB  140 KB
    case 1: return tl_narrow<4, 3, 4, 2, 64, 3>(a, s);   // 256 x  96: 352 rows         = 44 KB  132 KB
    case 2: return tl_narrow<4, 6, 4, 2, 32, 5>(a, s);   // 256 x 192: 448 rows x 64 B  = 28 KB  140 KB
    case 3: return tl_wide<4, 4, 2, 4, 64, 3>(a, s);     // 128 x 256: 384 rows x 128 B = 48 KB  144 KB
    case 4: return tl_wide<4, 6, 2, 4, 32, 4>(a, s);     // 128 x 384: 512 rows x 64 B  = 32 KB  128 KB
    case 6:                                              // 128 x 192: 320 rows x 128 B = 40 KB  120 KB + 24 KB of partial sums
      return a.epi == SF_EPI_EMBED_F32 ? tl_go<2, 6, 4, 2, 64, 3, SF_EPI_EMBED_F32, false>(a, s) : tl_go<2, 6, 4, 2, 64, 3, SF_EPI_RESID_F32, false>(a, s);
    default: return tl_wide_plain<8, 4, 2, 4, 32, 4>(a, s);   // 256 x 256: 512 rows x 64 B = 32 KB  128 KB
  }
}
