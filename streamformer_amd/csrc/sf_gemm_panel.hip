// "Panel" MFMA GEMM for the N = 768 projections with a residual epilogue (attention output
// projections and MLP down-projection):   out = resid + alpha * (A[M,K] * W[N,K]^T + bias)
//
// Why a third GEMM kernel: with N = 768 a 256x256 tiling gives 98 x 3 = 294 tiles on 256 CUs (two
// rounds, the second 15 % full) and 128x128 tiles run the 2-barrier schedule.  Here the tile is
// (rows/128) x 384: for the BASELINE shape M = 25088 that is 196 x 384 -> exactly 128 x 2 = 256 tiles,
// one per CU, one round, with fewer operand bytes per FLOP than a 256^2 tile.
//
// gfx950 structure: 8 waves = 1(M) x 8(N); a wave owns all 13 m-tiles x 3 n-tiles (208 x 48) = 39
// MFMA 16x16x32 per 32-deep K-tile.  A K-tile is one A piece (256 rows x 64 B, rows past the tile are
// clamped re-reads) + one W piece (384 x 64 B) = 40 KB, in a 4-slot LDS ring (160 KB: the whole CU).
// One phase per K-tile: [refill issue -> ds_reads -> counted vmcnt] -> barrier -> 39 MFMA -> barrier, with the two
// halves of the workgroup staggered by one barrier (one half in its MFMA segment while the other refills and reads).
// Hazards by count and distance, see the main loop: a slot is refilled only when every read of it has been consumed by
// an MFMA segment that has since completed, and read one phase after the vmcnt that retires it.  (Round 1 issued the
// refill in front of the wave's own MFMAs: 144-146 us per K = 3072 launch against 138-140 us now, same box.)
#include "sf_common.h"
#include "sf_switches.h"
#include <cstdlib>

#define P_THREADS 512
#define P_NT 3
#define P_SLOT_BYTES 40960
#define P_A_BYTES 16384

typedef __attribute__((address_space(3))) void* lptr_t;

SF_DEVICE f32x4_t mfma16p(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}
template <int N>
SF_DEVICE void wait_vmp() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
SF_DEVICE bf16x8_t rd32(const char* piece, int row, int kc) {
  return *reinterpret_cast<const bf16x8_t*>(piece + row * 64 + ((kc ^ sf_swz64(row)) << 4));
}

// SF_PANEL_TRACE (tools/panel_trace_lab.hip only): shader-clock stamps around the segments of the epilogue, summed over waves into
// panel_trace[]; bit 1 of the value drops the residual loads, bit 2 the plane stores, bit 3 the row arithmetic (what is left is latency)
#ifdef SF_PANEL_TRACE
__device__ unsigned long long panel_trace[16];
#define PT_DECL unsigned long long pt_[6] = {0, 0, 0, 0, 0, 0}, pt_t = 0, pt_0 = 0
#define PT_START() do { pt_t = pt_0 = __builtin_amdgcn_s_memtime(); } while (0)
#define PT_MARK(i) do { const unsigned long long pt_n = __builtin_amdgcn_s_memtime(); pt_[i] += pt_n - pt_t; pt_t = pt_n; } while (0)
#define PT_FLUSH() do { if (lane == 0) { for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&panel_trace[i_], pt_[i_]); \
    atomicAdd(&panel_trace[6], __builtin_amdgcn_s_memtime() - pt_0); atomicAdd(&panel_trace[7], 1ull); } } while (0)
#define PT_OFF(bit) ((SF_PANEL_TRACE) & (bit))
#else
#define PT_DECL
#define PT_START()
#define PT_MARK(i)
#define PT_FLUSH()
#define PT_OFF(bit) 0
#endif

template <int P_MT>
__global__ __launch_bounds__(P_THREADS) void sf_gemm_panel_kernel(SfGemmArgs p, int rows_per_tile, int ntiles, int stagger_ticks, int pad_clamp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2;
  const int l15 = lane & 15, g = lane >> 4;
  const int K = p.K;
  const int nkt = K >> 5;
  if (stagger_ticks > 0) {      // phase stagger, see sf_gemm256.hip
    const int grp = (blockIdx.x >> 4) % 3;
    if (grp) {
      const unsigned long long t0 = wall_clock64();
      while (wall_clock64() - t0 < (unsigned long long)grp * (unsigned long long)stagger_ticks) __builtin_amdgcn_s_sleep(8);
    }
  }
  // tile = (row panel, column half); the two halves of a row panel sit on the same XCD (b, b+8)
  for (int bid = blockIdx.x; bid < ntiles; bid += gridDim.x) {
  const int panel = (bid >> 4) * 8 + (bid & 7), nh = (bid >> 3) & 1;
  const int m0 = panel * rows_per_tile;
  const int m_end = min(m0 + rows_per_tile, p.M);
  const int n0 = nh * 384;
  if (m0 >= p.M) continue;

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a_hi, 0, (unsigned)p.M * (unsigned)K * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_hi, 0, (unsigned)p.N * (unsigned)K * 2u, 0x00020000);
  unsigned offA[2], offW[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * P_THREADS + tid;
    const int row = c >> 2, kc = (c & 3) ^ sf_swz64(row);
    if (i < 2) {
      // rows past the tile: an offset beyond the buffer's num_records — the DMA returns zeros without touching memory, and the MFMAs of the
      // padding rows (12 of 208 at 196-row panels) run on zero operands (SF_PANEL_PAD_CLAMP=1: round 1-3 behaviour, re-reads of the last row)
      const int ar = m0 + row;
      offA[i] = (ar < m_end || pad_clamp) ? ((unsigned)(ar < m_end ? ar : m_end - 1) * (unsigned)K + kc * 8) * 2u : 0xffff0000u;
    }
    offW[i] = ((unsigned)(n0 + row) * (unsigned)K + kc * 8) * 2u;
  }
  const int dma_lds = wave * 1024;
  auto issue = [&](int t) {
    char* dst = smem + (t & 3) * P_SLOT_BYTES + dma_lds;
    const int kof = t * 64;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)dst, 16, offA[0], kof, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(dst + 8192), 16, offA[1], kof, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(dst + P_A_BYTES), 16, offW[0], kof, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(dst + P_A_BYTES + 8192), 16, offW[1], kof, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(dst + P_A_BYTES + 16384), 16, offW[2], kof, 0, 0);
  };

  PT_DECL;
  PT_START();
  f32x4_t acc[P_MT][P_NT];
#pragma unroll
  for (int i = 0; i < P_MT; ++i)
#pragma unroll
    for (int j = 0; j < P_NT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t af[P_MT], wf[P_NT];

  auto reads = [&](int t) {
    const char* pa = smem + (t & 3) * P_SLOT_BYTES;
    const char* pw = pa + P_A_BYTES;
#pragma unroll
    for (int nt = 0; nt < P_NT; ++nt) wf[nt] = rd32(pw, wave * 48 + nt * 16 + l15, g);
#pragma unroll
    for (int mt = 0; mt < P_MT; ++mt) af[mt] = rd32(pa, mt * 16 + l15, g);
  };
  auto mma = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mt = 0; mt < P_MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < P_NT; ++nt) acc[mt][nt] = mfma16p(wf[nt], af[mt], acc[mt][nt]);
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- main loop ---------------------------------------------------------------------------------------------------
  // Two barriers per K-tile; the halves of the workgroup run one barrier apart (half 1 takes an extra barrier first), so
  // that on every SIMD one wave is in its MFMA segment (39 MFMAs, s_setprio 1) while the other is in its READ segment:
  // LDS-DMA refill + fragment reads + vmcnt wait.  Round 2: the refill is issued in the read segment, not in front of the
  // wave's own MFMAs (five DMA instructions cost 300-500 issue cycles that the SIMD's matrix pipe sat out: -4 % per launch;
  // without the s_setprio the loop is 15 % slower).  A slot may only be refilled once NO ds_read of it can be outstanding
  // anywhere in the workgroup; with the refill inside a read segment that is guaranteed by the distance, per half:
  //   half 1 (runs one segment late) refills K-tile t+3 = the slot of K-tile t-1: half 0 read it three segments earlier, half 1
  //          itself two segments earlier and has since gone through the MFMA segment that consumed those reads;
  //   half 0 refills K-tile t+2 = the slot of K-tile t-2, read by both halves at least three segments earlier.
  // (Half 0 refilling K-tile t+3 here could overtake half 1's reads of K-tile t-1 issued just before the barrier: the
  // timing margin is hundreds of cycles, but it is a margin, not an order.)
  int t = 0;
  if (half == 1) {
    issue(0); issue(1); issue(2);
    wait_vmp<10>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    for (; t + 3 < nkt; ++t) {
      issue(t + 3);
      reads(t);
      wait_vmp<10>();                   // own pieces of K-tile t+1 landed (t+2, t+3 may be in flight)
      __builtin_amdgcn_s_barrier();
      mma();
      __builtin_amdgcn_s_barrier();
    }
    reads(t); wait_vmp<5>(); __builtin_amdgcn_s_barrier(); mma(); __builtin_amdgcn_s_barrier(); ++t;
    reads(t); wait_vmp<0>(); __builtin_amdgcn_s_barrier(); mma(); __builtin_amdgcn_s_barrier(); ++t;
    reads(t); __builtin_amdgcn_s_barrier(); mma(); __builtin_amdgcn_s_barrier();
  } else {
    issue(0); issue(1);
    wait_vmp<5>();
    __builtin_amdgcn_s_barrier();
    for (; t + 2 < nkt; ++t) {
      issue(t + 2);
      reads(t);
      wait_vmp<5>();                    // own pieces of K-tile t+1 landed (t+2 may be in flight)
      __builtin_amdgcn_s_barrier();
      mma();
      __builtin_amdgcn_s_barrier();
    }
    reads(t); wait_vmp<0>(); __builtin_amdgcn_s_barrier(); mma(); __builtin_amdgcn_s_barrier(); ++t;
    reads(t); __builtin_amdgcn_s_barrier(); mma(); __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
  }

#ifdef SF_LAB      // lab library only (SF_PANEL_LAB_EPI): 1 = odd tiles, 3 = three of four tiles, 2 = all tiles leave without their epilogue —
  // is the epilogue bound by the chip's HBM rate (time falls with the number of CUs in it) or per CU?  Results are invalid.
  if (p.w_nt >= 78) {
    const int lm = p.w_nt - 78;
    if (lm == 2 || (lm == 1 && (blockIdx.x & 1)) || (lm == 3 && (blockIdx.x & 3))) {
      if (acc[0][0][0] == 12345.678f) p.out_hi[0] = 0;
      continue;
    }
  }
#endif
  // ---- epilogue: stage 64-row groups in LDS as fp32 rows of 384, then whole-row 16-byte I/O --------
  PT_MARK(0);                                      // [0] main loop
  int tid_e = threadIdx.x;
  asm volatile("" : "+v"(tid_e));
  const int el15 = tid_e & 15, eg = (tid_e >> 4) & 3;
  f32x4_t bias4[P_NT];
#pragma unroll
  for (int nt = 0; nt < P_NT; ++nt)
    bias4[nt] = p.bias ? *reinterpret_cast<const f32x4_t*>(p.bias + n0 + wave * 48 + nt * 16 + eg * 4)
                       : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // staging image: [64 rows][96 chunks of 16 B], chunk XOR (row & 31) inside its 32-chunk group.
  // Copy-out: wave w owns rows w, w+8, ... of the group; a row is 96 chunks = lanes 0..63 + lanes 0..31,
  // so the LayerNorm partial statistics of the row half (sum x, sum x^2 over these 384 columns) are
  // two deterministic wave reductions.
  const int elane = tid_e & 63;
#pragma unroll
  for (int grp = 0; grp < (P_MT + 3) / 4; ++grp) {
    constexpr int kRows = 8;                         // rows per wave in a 64-row group
    const int rows_w = (grp + 1) * 4 <= P_MT ? kRows : 2 * (P_MT - grp * 4);   // a partial last group: 16 rows per m-tile
    // residual rows of this group: all loads in flight before the staging pass (latency overlap)
    f32x4_t res[kRows][2];
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
      const int m = m0 + grp * 64 + wave + 8 * j;
      if (p.resid_hi) {            // residual stream as hi + lo bf16 planes: lane e < 48 takes columns 8 e .. 8 e + 7 (16 bytes of each plane)
        u32x4_t h = {0u, 0u, 0u, 0u}, l = {0u, 0u, 0u, 0u};
        if (j < rows_w && m < m_end && elane < 48 && !PT_OFF(2)) {
          const size_t ro = (size_t)m * (size_t)p.ldc + n0 + elane * 8;
          h = *reinterpret_cast<const u32x4_t*>(p.resid_hi + ro);
          l = *reinterpret_cast<const u32x4_t*>(p.resid_lo + ro);
        }
        res[j][0] = __builtin_bit_cast(f32x4_t, h);
        res[j][1] = __builtin_bit_cast(f32x4_t, l);
        continue;
      }
      const bool ok = j < rows_w && m < m_end && p.resid != nullptr;     // no residual: plain F32 / BF16 output
      const int mr = p.resid_mod > 0 ? m % p.resid_mod : m;          // embedding table: one row per (frame slot, patch)
      const float* rp = p.resid + (size_t)mr * (size_t)p.ldc + n0;
      res[j][0] = ok ? *reinterpret_cast<const f32x4_t*>(rp + elane * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
      res[j][1] = (ok && elane < 32) ? *reinterpret_cast<const f32x4_t*>(rp + 256 + elane * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mt = grp * 4 + q;
      if (mt < P_MT) {
        const int r = q * 16 + el15;
#pragma unroll
        for (int nt = 0; nt < P_NT; ++nt) {
          const int chunk = wave * 12 + nt * 4 + eg;
          const f32x4_t v = acc[mt][nt] + bias4[nt];
          *reinterpret_cast<f32x4_t*>(smem + r * 1536 + (((chunk & ~31) | ((chunk ^ r) & 31)) << 4)) = v;
        }
      }
    }
    PT_MARK(1);                                    // [1] residual loads issued + accumulators staged
    __syncthreads();
    PT_MARK(2);                                    // [2] barrier
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
      const int r = wave + 8 * j;
      const int m = m0 + grp * 64 + r;
      if (j < rows_w && m < m_end) {                 // wave-uniform
        float s1 = 0.f, s2 = 0.f;
        if (p.resid_hi) {          // plane form: 8 columns per lane, 16-byte accesses on both planes
          if (elane < 48) {
            const u32x4_t h = __builtin_bit_cast(u32x4_t, res[j][0]), l = __builtin_bit_cast(u32x4_t, res[j][1]);
            u32x4_t ho, lo;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
              const int c = 2 * elane + hf;
              const f32x4_t v = *reinterpret_cast<const f32x4_t*>(smem + r * 1536 + (((c & ~31) | ((c ^ r) & 31)) << 4));
              f32x4_t x;
              x[0] = bf2f(h[2 * hf] & 0xffffu) + bf2f(l[2 * hf] & 0xffffu) + p.alpha * v[0];
              x[1] = bf2f(h[2 * hf] >> 16) + bf2f(l[2 * hf] >> 16) + p.alpha * v[1];
              x[2] = bf2f(h[2 * hf + 1] & 0xffffu) + bf2f(l[2 * hf + 1] & 0xffffu) + p.alpha * v[2];
              x[3] = bf2f(h[2 * hf + 1] >> 16) + bf2f(l[2 * hf + 1] >> 16) + p.alpha * v[3];
              ho[2 * hf] = pack_bf2(x[0], x[1]); ho[2 * hf + 1] = pack_bf2(x[2], x[3]);
              lo[2 * hf] = pack_bf2(x[0] - bf2f(ho[2 * hf] & 0xffffu), x[1] - bf2f(ho[2 * hf] >> 16));
              lo[2 * hf + 1] = pack_bf2(x[2] - bf2f(ho[2 * hf + 1] & 0xffffu), x[3] - bf2f(ho[2 * hf + 1] >> 16));
              s1 += (x[0] + x[1]) + (x[2] + x[3]);
              s2 += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
            }
            const size_t o = (size_t)m * (size_t)p.ldc + n0 + elane * 8;
            if (!PT_OFF(4) || s1 == 12345.678f) {
              *reinterpret_cast<u32x4_t*>(p.out_hi + o) = ho;
              *reinterpret_cast<u32x4_t*>(p.out_lo + o) = lo;
            }
          }
        } else
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
          const int c = hp * 64 + elane;
          if (hp == 0 || elane < 32) {
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(smem + r * 1536 + (((c & ~31) | ((c ^ r) & 31)) << 4));
            const f32x4_t x = res[j][hp] + p.alpha * v;
            const size_t o = (size_t)m * (size_t)p.ldc + n0 + c * 4;
            if (p.out_f32) *reinterpret_cast<f32x4_t*>(p.out_f32 + o) = x;
            if (p.out_hi) {
              const u32x2_t hv = {pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3])};
              *reinterpret_cast<u32x2_t*>(p.out_hi + o) = hv;
              if (p.out_lo)
                *reinterpret_cast<u32x2_t*>(p.out_lo + o) = (u32x2_t){pack_bf2(x[0] - bf2f(hv[0] & 0xffffu), x[1] - bf2f(hv[0] >> 16)),
                                                                    pack_bf2(x[2] - bf2f(hv[1] & 0xffffu), x[3] - bf2f(hv[1] >> 16))};
              s1 += (x[0] + x[1]) + (x[2] + x[3]);
              s2 += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
            }
          }
        }
        if (p.ln_stats_out) {
          s1 = wave_sum_dpp(s1);
          s2 = wave_sum_dpp(s2);
          if (elane == 0) {
            // rows of 8 floats = four pairs (the lab kernels sf_gemm_pp / sf_gemm_pipe write one per 192-column quarter): this half's
            // pair; pairs 1 and 3 stay at the zeros run_forward put there once per forward
            p.ln_stats_out[(size_t)m * 8 + nh * 4 + 0] = s1;
            p.ln_stats_out[(size_t)m * 8 + nh * 4 + 1] = s2;
          }
        }
      }
    }
    PT_MARK(3);                                    // [3] row loop
    __syncthreads();
    PT_MARK(4);                                    // [4] closing barrier (its fence waits for the plane stores)
  }
  PT_FLUSH();
  }   // tiles
}

static int panel_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (const char* e = sf_sw(SW_ASSUME_CUS)) cus = atoi(e);      // experiment: kernels sized for a CU-masked stream
    if (cus < 16) cus = 256;
    cus &= ~15;
  }
  return cus;
}

// Tiling plan: P row panels (a multiple of CUs/2, so that the 2P tiles fill whole rounds), rows = ceil(M/P)
// per panel, MT = the smallest instantiated m-tile count covering `rows`.
struct PanelPlan { int panels, rows, mt, ok; };
static PanelPlan panel_plan(int M) {
  PanelPlan pl = {0, 0, 0, 0};
  const int half = panel_cus() / 2;
  int panels = ((M + 207) / 208 + half - 1) / half * half;
  if (panels <= 0) return pl;
  const int rows = (M + panels - 1) / panels;
  static const int kMt[4] = {2, 4, 7, 13};
  for (int i = 0; i < 4; ++i)
    if (rows <= 16 * kMt[i]) { pl.mt = kMt[i]; break; }
  if (!pl.mt) return pl;
  // >= 75 % of the MFMA rows real; from five clips on 55 % is enough — the folded schedule (and the plane-form residual that comes
  // with it) beats LayerNorm launches + the 256^2 / 128^2 residual kernels there (tools/bsweep.py: 5 clips 6.44 -> 6.09 ms, 6 clips
  // 6.95 -> 6.64; three clips at 66 % lose 6 %).  SF_PANEL_MIN_FILL_PCT forces one threshold (lab).
  const int forced = sf_sw(SW_PANEL_MIN_FILL_PCT) ? atoi(sf_sw(SW_PANEL_MIN_FILL_PCT)) : 0;
  const int min_fill = forced ? forced : (M >= 14000 ? 55 : 75);
  pl.panels = panels; pl.rows = rows; pl.ok = rows * 100 >= pl.mt * 16 * min_fill;
  return pl;
}

bool sf_gemm_panel_supported(const SfGemmArgs& a, bool split) {
  if (split || a.N != 768 || a.grp_rows > 0) return false;
  if (a.out_lo && (a.epi != SF_EPI_RESID_F32 || !a.out_hi)) return false;        // lo plane only next to the hi plane of a residual producer
  if (a.resid_hi && (!a.resid_lo || !a.out_lo || a.epi != SF_EPI_RESID_F32 || a.resid_mod > 0)) return false;
  if (a.epi != SF_EPI_RESID_F32 && a.epi != SF_EPI_F32 && a.epi != SF_EPI_BF16) return false;
  if (a.ln_stats) return false;                     // LN-folded consumers run on the 256^2 kernel
  if (a.ln_stats_out && !a.ln_stats_wide) return false;   // statistics rows are 8 floats wide (SfGemmArgs::ln_stats_wide)
  if (a.K % 32 || a.K < 128) return false;
  if ((size_t)a.M * a.K * 2 >= ((size_t)1 << 32)) return false;
  return panel_plan(a.M).ok != 0;
}

hipError_t sf_launch_gemm_panel(const SfGemmArgs& a_in, hipStream_t s) {
  SfGemmArgs a = a_in;
  if (a.epi == SF_EPI_F32) { a.resid = nullptr; a.alpha = 1.f; a.out_hi = nullptr; }
  if (a.epi == SF_EPI_BF16) { a.resid = nullptr; a.alpha = 1.f; a.out_f32 = nullptr; }
  const PanelPlan pl = panel_plan(a.M);
  if (!pl.ok) return hipErrorInvalidValue;
  const int cus = panel_cus();
  const int ntiles = pl.panels * 2;
  static SfPerDeviceOnce attr_set;
  if (attr_set.first()) {
#define SF_PATTR(MT) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_panel_kernel<MT>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * P_SLOT_BYTES);
    SF_PATTR(2) SF_PATTR(4) SF_PATTR(7) SF_PATTR(13)
#undef SF_PATTR
  }
  // Phase stagger (see sf_gemm256.hip): three groups 3.5 us apart spread the read-heavy main loops and the
  // residual read + store bursts of the epilogues.  Box-dependent: -3.3 % on the whole forward on one MI355X,
  // neutral on another; never slower in the sweeps (tools/stagger_sweep.py).  SF_PANEL_STAGGER_NS overrides.
  int stagger = (ntiles >= cus && pl.mt == 13) ? sf_wall_clock_ticks(3500) : 0;      // full-height tiles only: small tiles finish before a step elapses
  if (const char* e = sf_sw(SW_PANEL_STAGGER_NS)) stagger = sf_wall_clock_ticks(atoi(e));
  if (const int lm = SF_LAB_SWITCH("SF_PANEL_LAB_EPI")) a.w_nt = 78 + lm;      // lab builds only
  const int pad_clamp = sf_sw(SW_PANEL_PAD_CLAMP) ? 1 : 0;      // A/B switch
  const dim3 grid(ntiles < cus ? ntiles : cus), block(P_THREADS);
  const size_t lds = 4 * P_SLOT_BYTES;
  switch (pl.mt) {
    case 2: hipLaunchKernelGGL(sf_gemm_panel_kernel<2>, grid, block, lds, s, a, pl.rows, ntiles, stagger, pad_clamp); break;
    case 4: hipLaunchKernelGGL(sf_gemm_panel_kernel<4>, grid, block, lds, s, a, pl.rows, ntiles, stagger, pad_clamp); break;
    case 7: hipLaunchKernelGGL(sf_gemm_panel_kernel<7>, grid, block, lds, s, a, pl.rows, ntiles, stagger, pad_clamp); break;
    default: hipLaunchKernelGGL(sf_gemm_panel_kernel<13>, grid, block, lds, s, a, pl.rows, ntiles, stagger, pad_clamp); break;
  }
  return hipGetLastError();
}
