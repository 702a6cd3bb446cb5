// "Ping-pong" panel GEMM for the N = 768 residual projections at BASELINE-sized M (round 4):
//     out = resid + alpha * (A[M,K] * W[N,K]^T + bias)        residual stream as hi + lo bf16 planes
//
// Why a second panel kernel.  sf_gemm_panel.hip gives every CU exactly ONE 196 x 384 tile: 23 us of main loop, then 36 us of
// residual read-modify-write during which no MFMA runs anywhere on the chip (every CU is in the same phase).
// Here the tile is 98 x 192 (7 m-tiles x 12 n-tiles), a workgroup is FOUR waves (one per SIMD, 84 accumulator VGPRs) with a
// 76 KB LDS ring, so TWO workgroups are resident per CU and the hardware runs one workgroup's main loop (MFMA + L2 -> LDS
// traffic) beside the other's epilogue (HBM read-modify-write): 1024 tiles = four per CU, the second resident workgroup of a
// CU starts late by about one main loop so that the pair stays out of phase for the whole launch.
//
// gfx950 structure: wave w of a workgroup owns all 7 m-tiles x 3 n-tiles (112 x 48) = 21 MFMA 16x16x32 per 32-deep K-tile.
// A K-tile is one A piece (112 rows x 64 B, rows past the tile are clamped re-reads) + one W piece (192 x 64 B) = 19 KB in a
// 4-slot ring, staged by LDS-DMA (five 1 KB pieces per wave and K-tile).  One barrier per K-tile:
//     issue K-tile t+2  ->  ds_reads of K-tile t  ->  vmcnt: own pieces of t+1 landed  ->  barrier  ->  21 MFMA
// Hazards by count and distance: K-tile t+2 goes into the slot of K-tile t-2, whose reads every wave issued before barrier
// t-2 and consumed (MFMA operands) before barrier t-1, which the issuing wave has passed; K-tile t+1 is read after the
// barrier that follows every wave's vmcnt wait for its own pieces of it.
// The residual rows of the first 64-row group are requested BEFORE the first K-tile (64 VGPRs: there is room beside 84
// accumulators): they are the oldest loads of the wave, so the first counted wait covers them and they cost no latency later.
// Epilogue: C leaves through the (now idle) ring as fp32 rows of 192 columns, 64 rows at a time; the copy-out gives 24 lanes
// one row (8 columns = 16 bytes of each plane per lane, whole 128-byte lines), adds the residual planes, re-splits, stores
// both planes and reduces the row's {sum x, sum x^2} over its 192 columns for the LayerNorm fold of the next Linear
// (ln_stats_out rows of 8 floats: one pair per 192-column quarter, fixed order -> deterministic).
#include "sf_common.h"
#include <cstdlib>

#define Q_THREADS 256
#define Q_MT 7
#define Q_NT 3
#define Q_BN 192
#define Q_A_BYTES 7168
#define Q_SLOT_BYTES 19456
#define Q_LDS_BYTES (4 * Q_SLOT_BYTES)

typedef __attribute__((address_space(3))) void* lptr_t;

namespace {
SF_DEVICE f32x4_t mfma16q(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}
template <int N>
SF_DEVICE void wait_vmq() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
SF_DEVICE bf16x8_t rd32q(const char* piece, int row, int kc) {
  return *reinterpret_cast<const bf16x8_t*>(piece + row * 64 + ((kc ^ sf_swz64(row)) << 4));
}
// sum over each half-wave (lanes 0..31 / 32..63) on the DPP path; the totals sit in lanes 31 and 63.  All lanes active.
SF_DEVICE float half_sum_dpp(float v) {
  v = dpp_add<0x111, 0xf>(v);
  v = dpp_add<0x112, 0xf>(v);
  v = dpp_add<0x114, 0xf>(v);
  v = dpp_add<0x118, 0xf>(v);
  v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  return v;
}
// staging image of a 64-row group: [64 rows][48 chunks of 16 B], chunk XOR (row & 7) inside its 8-chunk group
SF_DEVICE int stage_off(int r, int chunk) { return r * 768 + (((chunk & ~7) | ((chunk ^ r) & 7)) << 4); }
}  // namespace

__global__ __launch_bounds__(Q_THREADS, 2) void sf_gemm_pp_kernel(SfGemmArgs p, int rows_per_tile, int panels, int stagger_ticks,
                                                                 int stagger_mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int K = p.K;
  const int nkt = K >> 5;
  // block -> tile: the four column quarters of a row panel are four consecutive blocks of ONE XCD (block b runs on XCD b % 8),
  // so the panel's A rows reach that XCD's L2 once
  const int b = blockIdx.x;
  const int xcd = b & 7, j = b >> 3;
  const int q = j & 3, panel = (j >> 2) * 8 + xcd;
  const int m0 = panel * rows_per_tile;
  if (panel >= panels || m0 >= p.M) return;
  const int m_end = min(m0 + rows_per_tile, p.M);
  const int n0 = q * Q_BN;
  if (stagger_ticks > 0) {
    // second resident workgroup of a CU: start about one main loop late (the pair then stays out of phase).  Which blocks share a
    // CU is the dispatcher's business; mode 1 assumes round-robin over the 32 CUs of an XCD, mode 2 pairs of consecutive blocks
    const int late = stagger_mode == 2 ? (j & 1) : ((j >> 5) & 1);
    if (late && j < 64) {
      const unsigned long long t0 = wall_clock64();
      while (wall_clock64() - t0 < (unsigned long long)stagger_ticks) __builtin_amdgcn_s_sleep(8);
    }
  }

  // ---- residual rows of group 0 (rows 0..63 of the tile): 8 passes of 8 rows, 24 lanes per row -------------------------
  const int sub = tid & 31, rgrp = tid >> 5;
  const bool col_ok = sub < 24;
  u32x4_t rh[8], rl[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const int m = m0 + jj * 8 + rgrp;
    rh[jj] = (u32x4_t){0u, 0u, 0u, 0u};
    rl[jj] = (u32x4_t){0u, 0u, 0u, 0u};
    if (col_ok && m < m_end) {
      const size_t ro = (size_t)m * (size_t)p.ldc + n0 + sub * 8;
      rh[jj] = *reinterpret_cast<const u32x4_t*>(p.resid_hi + ro);
      rl[jj] = *reinterpret_cast<const u32x4_t*>(p.resid_lo + ro);
    }
  }

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a_hi, 0, (unsigned)p.M * (unsigned)K * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_hi, 0, (unsigned)p.N * (unsigned)K * 2u, 0x00020000);
  // DMA pieces of 1 KB (16 rows x 64 B): A = 7 pieces, W = 12.  Wave w issues A pieces {w, w + 4} and W pieces {w, w + 4, w + 8};
  // A piece 7 does not exist, wave 3 re-issues its piece 3 there (same bytes to the same place) so that every wave has five
  // loads per K-tile in its vmcnt queue.
  unsigned offA[2], offW[3];
  int dstA[2], dstW[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * Q_THREADS + tid;
    const int row = c >> 2, kc = (c & 3) ^ sf_swz64(row);
    offW[i] = ((unsigned)(n0 + row) * (unsigned)K + kc * 8) * 2u;
    dstW[i] = Q_A_BYTES + i * 4096 + wave * 1024;
    if (i < 2) {
      const int ci = (i == 1 && wave == 3) ? tid : c;
      const int rowa = ci >> 2, kca = (ci & 3) ^ sf_swz64(rowa);
      int ar = m0 + rowa;
      ar = ar < m_end ? ar : m_end - 1;
      offA[i] = ((unsigned)ar * (unsigned)K + kca * 8) * 2u;
      dstA[i] = ((i == 1 && wave == 3) ? 0 : i * 4096) + wave * 1024;
    }
  }
  auto issue = [&](int t) {
    char* dst = smem + (t & 3) * Q_SLOT_BYTES;
    const int kof = t * 64;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(dst + dstA[0]), 16, (int)offA[0], kof, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(dst + dstA[1]), 16, (int)offA[1], kof, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(dst + dstW[0]), 16, (int)offW[0], kof, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(dst + dstW[1]), 16, (int)offW[1], kof, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(dst + dstW[2]), 16, (int)offW[2], kof, 0, 0);
  };

  f32x4_t acc[Q_MT][Q_NT];
#pragma unroll
  for (int i = 0; i < Q_MT; ++i)
#pragma unroll
    for (int jn = 0; jn < Q_NT; ++jn) acc[i][jn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t af[Q_MT], wf[Q_NT];
  auto reads = [&](int t) {
    const char* pa = smem + (t & 3) * Q_SLOT_BYTES;
    const char* pw = pa + Q_A_BYTES;
#pragma unroll
    for (int nt = 0; nt < Q_NT; ++nt) wf[nt] = rd32q(pw, wave * 48 + nt * 16 + l15, g);
#pragma unroll
    for (int mt = 0; mt < Q_MT; ++mt) af[mt] = rd32q(pa, mt * 16 + l15, g);
  };
  auto mma = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mt = 0; mt < Q_MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < Q_NT; ++nt) acc[mt][nt] = mfma16q(wf[nt], af[mt], acc[mt][nt]);
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- main loop ---------------------------------------------------------------------------------------------------
  issue(0);
  issue(1);
  wait_vmq<5>();                         // K-tile 0 (and the residual rows requested above) landed
  __builtin_amdgcn_s_barrier();
  int t = 0;
  for (; t + 2 < nkt; ++t) {
    issue(t + 2);
    reads(t);
    wait_vmq<5>();                       // own pieces of K-tile t+1 landed (t+2 in flight)
    __builtin_amdgcn_s_barrier();
    mma();
  }
  reads(t); wait_vmq<0>(); __builtin_amdgcn_s_barrier(); mma(); ++t;
  reads(t); mma();
  __syncthreads();                       // every wave's fragment reads retired: the ring becomes the staging area

  // ---- epilogue ------------------------------------------------------------------------------------------------------
  int tid_e = threadIdx.x;
  asm volatile("" : "+v"(tid_e));
  const int el15 = tid_e & 15, eg = (tid_e >> 4) & 3;
  f32x4_t bias4[Q_NT];
#pragma unroll
  for (int nt = 0; nt < Q_NT; ++nt)
    bias4[nt] = p.bias ? *reinterpret_cast<const f32x4_t*>(p.bias + n0 + wave * 48 + nt * 16 + eg * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // residual rows of group 1 (rows 64..111 of the tile; 6 passes): in flight across the staging of group 0
  u32x4_t rh1[6], rl1[6];
#pragma unroll
  for (int jj = 0; jj < 6; ++jj) {
    const int m = m0 + 64 + jj * 8 + rgrp;
    rh1[jj] = (u32x4_t){0u, 0u, 0u, 0u};
    rl1[jj] = (u32x4_t){0u, 0u, 0u, 0u};
    if (col_ok && m < m_end) {
      const size_t ro = (size_t)m * (size_t)p.ldc + n0 + sub * 8;
      rh1[jj] = *reinterpret_cast<const u32x4_t*>(p.resid_hi + ro);
      rl1[jj] = *reinterpret_cast<const u32x4_t*>(p.resid_lo + ro);
    }
  }
  auto copy_row = [&](int grp, int jj, const u32x4_t& h, const u32x4_t& l) {
    const int r = jj * 8 + rgrp;
    const int m = m0 + grp * 64 + r;
    const bool ok = col_ok && m < m_end;
    float s1 = 0.f, s2 = 0.f;
    if (ok) {
      u32x4_t ho, lo;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(smem + stage_off(r, 2 * sub + hf));
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
      const size_t o = (size_t)m * (size_t)p.ldc + n0 + sub * 8;
      *reinterpret_cast<u32x4_t*>(p.out_hi + o) = ho;
      *reinterpret_cast<u32x4_t*>(p.out_lo + o) = lo;
    }
    if (p.ln_stats_out) {
      s1 = half_sum_dpp(s1);
      s2 = half_sum_dpp(s2);
      if (sub == 31 && m < m_end)
        *reinterpret_cast<u32x2_t*>(p.ln_stats_out + (size_t)m * 8 + q * 2) = (u32x2_t){__float_as_uint(s1), __float_as_uint(s2)};
    }
  };
#pragma unroll
  for (int grp = 0; grp < 2; ++grp) {
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int mt = grp * 4 + qq;
      if (mt < Q_MT) {
        const int r = qq * 16 + el15;
#pragma unroll
        for (int nt = 0; nt < Q_NT; ++nt)
          *reinterpret_cast<f32x4_t*>(smem + stage_off(r, wave * 12 + nt * 4 + eg)) = acc[mt][nt] + bias4[nt];
      }
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) copy_row(0, jj, rh[jj], rl[jj]);
    } else {
#pragma unroll
      for (int jj = 0; jj < 6; ++jj) copy_row(1, jj, rh1[jj], rl1[jj]);
    }
    if (grp == 0) __syncthreads();
  }
}

static int pp_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus < 16) cus = 256;
    cus &= ~15;
  }
  return cus;
}

// Tiling plan: row panels of <= 112 rows, their count a multiple of 8 (XCD walk) that gives every CU a whole number of the
// 4 x panels tiles when it can (M = 25 088: 256 panels of 98 rows, 1024 tiles, four per CU).
static bool pp_plan(int M, int* panels_out, int* rows_out) {
  const int cus = pp_cus();
  const int unit = cus / 4 > 8 ? cus / 4 : 8;                  // panels per "one tile per CU"
  int panels = (M + Q_MT * 16 - 1) / (Q_MT * 16);
  panels = (panels + unit - 1) / unit * unit;
  const int rows = (M + panels - 1) / panels;
  if (rows > Q_MT * 16 || rows * 100 < Q_MT * 16 * 70) return false;     // >= 70 % of the MFMA rows real
  if (panels * 4 < 3 * cus) return false;                      // fewer than three tiles per CU: nothing to overlap with
  *panels_out = panels; *rows_out = rows;
  return true;
}

bool sf_gemm_pp_supported(const SfGemmArgs& a, bool split) {
  static const bool on = getenv("SF_PANEL_PP") != nullptr && atoi(getenv("SF_PANEL_PP")) != 0;
  if (!on) return false;
  if (split || a.N != 768 || a.grp_rows > 0 || a.epi != SF_EPI_RESID_F32) return false;
  if (!a.resid_hi || !a.resid_lo || !a.out_hi || !a.out_lo || a.resid_mod > 0 || a.out_f32) return false;
  if (a.ln_stats || (a.ln_stats_out && !a.ln_stats_wide)) return false;
  if (a.K % 32 || a.K < 128 || a.ldc % 8) return false;
  if ((size_t)a.M * a.K * 2 >= ((size_t)1 << 32)) return false;
  static const int max_k = getenv("SF_PANEL_PP_MAX_K") ? atoi(getenv("SF_PANEL_PP_MAX_K")) : 768;
  if (a.K > max_k) return false;
  int panels, rows;
  return pp_plan(a.M, &panels, &rows);
}

hipError_t sf_launch_gemm_pp(const SfGemmArgs& a, hipStream_t s) {
  int panels = 0, rows = 0;
  if (!pp_plan(a.M, &panels, &rows)) return hipErrorInvalidValue;
  static SfPerDeviceOnce attr_set;
  if (attr_set.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_pp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS_BYTES);
  // stagger = about one main loop of a tile running alone on its SIMDs (K-tiles x ~0.33 us), SF_PANEL_PP_STAGGER_NS overrides
  static const int forced_ns = getenv("SF_PANEL_PP_STAGGER_NS") ? atoi(getenv("SF_PANEL_PP_STAGGER_NS")) : -1;
  static const int mode = getenv("SF_PANEL_PP_STAGGER_MODE") ? atoi(getenv("SF_PANEL_PP_STAGGER_MODE")) : 1;
  const int ns = forced_ns >= 0 ? forced_ns : (a.K >> 5) * 330;
  const int stagger = ns > 0 ? sf_wall_clock_ticks(ns) : 0;
  hipLaunchKernelGGL(sf_gemm_pp_kernel, dim3(panels * 4), dim3(Q_THREADS), Q_LDS_BYTES, s, a, rows, panels, stagger, mode);
  return hipGetLastError();
}
