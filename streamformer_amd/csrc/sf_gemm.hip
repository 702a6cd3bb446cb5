// MFMA GEMM for every nn.Linear on the encoder path (reference modeling:513,629,728,811,830,895,
// 1118-1119, conv-as-GEMM :329-334):   C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogue)
//
// gfx950 design (v1):
//   * 128x128 block tile, 4 waves (2x2), each wave a 64x64 sub-tile = 4x4 MFMA 16x16x32 bf16 tiles.
//   * operands go HBM -> LDS with global_load_lds (16 B / lane, no VGPR round trip), LDS is
//     double-buffered, one barrier per K-tile.
//   * LDS image is lane-linear (a global_load_lds constraint), so the bank-conflict swizzle is put
//     on the per-lane SOURCE address and mirrored on the ds_read_b128 address (same involution).
//   * the MFMA is issued "swapped" (A-operand = weight rows, B-operand = activation rows) so each
//     lane ends up with 4 consecutive output columns of one row: 8/16-byte vector stores and
//     float4 bias / residual loads in the epilogue.
//   * SPLIT = the fp32-accurate mode: both operands carry a bf16 lo half, three MFMAs per
//     fragment pair (hi*hi + hi*lo + lo*hi), BK halves so the LDS footprint stays 64 KB.
//   * block id -> tile map is XCD-aware: the 8 XCDs each walk a contiguous band of row panels.
#include "sf_common.h"
#include "sf_switches.h"
#include <cstdlib>

#define BM 128
#define BN 128
#define NTHREADS 256

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BK>
SF_DEVICE int swz(int row) {
  // 16-byte slot XOR so the 16 rows one ds_read_b128 lane-group touches land on 16 distinct
  // slots of the 256-byte bank row.
  if (BK == 64) return (row >> 1) & 7;
  return sf_swz64(row);
}

// stage a [128 x BK] bf16 tile of X (row-major, leading dim K) into linear LDS at `lds`
template <int BK>
SF_DEVICE void stage_tile(const bf16_t* __restrict__ X, int row0, int rows_total, int K, int k0,
                          char* lds, int tid) {
  constexpr int CPR = BK / 8;                 // 16-byte chunks per row
  constexpr int ROUNDS = BM * CPR / NTHREADS;
  const int wave = tid >> 6;
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int c = r * NTHREADS + tid;
    const int row = c / CPR;
    const int slot = c % CPR;
    const int kc = slot ^ swz<BK>(row);
    int grow = row0 + row;
    grow = grow < rows_total ? grow : rows_total - 1;
    const bf16_t* src = X + (size_t)grow * K + k0 + kc * 8;
    char* dst = lds + (r * NTHREADS + wave * 64) * 16;   // wave-uniform base; lane*16 is implicit
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
  }
}

template <int BK>
SF_DEVICE bf16x8_t read_frag(const char* tile, int row, int kc) {
  constexpr int RB = BK * 2;
  const int off = row * RB + ((kc ^ swz<BK>(row)) << 4);
  return *reinterpret_cast<const bf16x8_t*>(tile + off);
}

SF_DEVICE f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}

template <bool SPLIT, int EPI>
__global__ __launch_bounds__(NTHREADS) void sf_gemm_kernel(SfGemmArgs p) {
  constexpr int BK = SPLIT ? 32 : 64;
  constexpr int TILE_BYTES = BM * BK * 2;                 // one [128 x BK] bf16 tile
  constexpr int STAGE_BYTES = TILE_BYTES * (SPLIT ? 4 : 2);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;

  // ---- XCD-aware tile mapping (bijective for any grid size) -----------------------------------
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nkt = p.K / BK;
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE_BYTES;
    stage_tile<BK>(p.a_hi, m0, p.M, p.K, kt * BK, base, tid);
    if (SPLIT) {
      stage_tile<BK>(p.a_lo, m0, p.M, p.K, kt * BK, base + TILE_BYTES, tid);
      stage_tile<BK>(p.w_hi, n0, p.N, p.K, kt * BK, base + 2 * TILE_BYTES, tid);
      stage_tile<BK>(p.w_lo, n0, p.N, p.K, kt * BK, base + 3 * TILE_BYTES, tid);
    } else {
      stage_tile<BK>(p.w_hi, n0, p.N, p.K, kt * BK, base + TILE_BYTES, tid);
    }
  };

  stage(0, 0);
  __syncthreads();   // hipcc drains the outstanding LDS-DMA (vmcnt(0)) in front of the barrier

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
    const char* base = smem + cur * STAGE_BYTES;
    const char* tA = base;
    const char* tAl = base + TILE_BYTES;
    const char* tW = base + (SPLIT ? 2 : 1) * TILE_BYTES;
    const char* tWl = base + 3 * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      const int kc = ks * 4 + g;
      bf16x8_t xa[4], wa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xa[i] = read_frag<BK>(tA, wr * 64 + i * 16 + l15, kc);
        wa[i] = read_frag<BK>(tW, wc * 64 + i * 16 + l15, kc);
      }
      if (SPLIT) {
        bf16x8_t xl[4], wl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xl[i] = read_frag<BK>(tAl, wr * 64 + i * 16 + l15, kc);
          wl[i] = read_frag<BK>(tWl, wc * 64 + i * 16 + l15, kc);
        }
        // small terms first, the hi*hi product last
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            acc[nt][mt] = mfma16(wl[nt], xa[mt], acc[nt][mt]);
            acc[nt][mt] = mfma16(wa[nt], xl[mt], acc[nt][mt]);
            acc[nt][mt] = mfma16(wa[nt], xa[mt], acc[nt][mt]);
          }
      } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = mfma16(wa[nt], xa[mt], acc[nt][mt]);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m][n..n+3], m = tile row (l15), n = 4 consecutive columns ---------
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wr * 64 + mt * 16 + l15;
    if (m >= p.M) continue;
    size_t orow = (size_t)m;
    if (p.grp_rows > 0) orow = sf_out_row(p, m);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wc * 64 + nt * 16 + g * 4;
      if (n >= p.N) continue;
      f32x4_t v = acc[nt][mt];
      if (p.bias) {
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(p.bias + n);
        v += b;
      }
      const size_t o = orow * (size_t)p.ldc + n;
      if (EPI == SF_EPI_F32) {
        *reinterpret_cast<f32x4_t*>(p.out_f32 + o) = v;
      } else if (EPI == SF_EPI_RESID_F32) {
        const f32x4_t r = *reinterpret_cast<const f32x4_t*>(p.resid + o);
        v = r + p.alpha * v;
        *reinterpret_cast<f32x4_t*>(p.out_f32 + o) = v;
      } else if (EPI == SF_EPI_EMBED_F32) {
        const int pn = m % p.Np, tt = (m / p.Np) % p.Tn;
        const f32x4_t pe = *reinterpret_cast<const f32x4_t*>(p.pos + (size_t)pn * p.N + n);
        const f32x4_t te = *reinterpret_cast<const f32x4_t*>(p.time_rows + (size_t)tt * p.N + n);
        v = v + pe + te;
        *reinterpret_cast<f32x4_t*>(p.out_f32 + o) = v;
      } else {
        if (EPI == SF_EPI_ACT_BF16) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = SPLIT ? apply_act(v[j], p.act) : apply_act_bf16(v[j], p.act);
        }
        unsigned int h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split_bf(v[j], h[j], l[j]);
        u32x2_t hv = {h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        *reinterpret_cast<u32x2_t*>(p.out_hi + o) = hv;
        if (p.out_lo) {
          u32x2_t lv = {l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
          *reinterpret_cast<u32x2_t*>(p.out_lo + o) = lv;
        }
      }
    }
  }
}

template <bool SPLIT>
static hipError_t launch_epi(const SfGemmArgs& a, dim3 grid, size_t lds, hipStream_t s) {
#define SF_CASE(E)                                                                       \
  case E:                                                                                \
    hipLaunchKernelGGL((sf_gemm_kernel<SPLIT, E>), grid, dim3(NTHREADS), lds, s, a);     \
    break;
  switch (a.epi) {
    SF_CASE(SF_EPI_F32)
    SF_CASE(SF_EPI_BF16)
    SF_CASE(SF_EPI_ACT_BF16)
    SF_CASE(SF_EPI_RESID_F32)
    SF_CASE(SF_EPI_EMBED_F32)
    default:
      return hipErrorInvalidValue;
  }
#undef SF_CASE
  return hipGetLastError();
}

int sf_wall_clock_ticks(int ns) {
  static int khz = 0;
  if (!khz) {
    int dev = 0, rate = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, dev) == hipSuccess && rate > 0) khz = rate;
    else khz = 100000;
  }
  return (int)((long long)ns * khz / 1000000);
}

int sf_infold_max_rows() {
  const int a = sf_skinny_max_rows(), b = sf_tile_max_rows();
  return a > b ? a : b;
}

hipError_t sf_launch_gemm(const SfGemmArgs& a, bool split, hipStream_t s) {
  if (a.aux_mode) return sf_gemm256_aux_supported(a) && !split ? sf_launch_gemm256(a, s) : hipErrorInvalidValue;
  if (a.ln_inkernel) {
    if (sf_gemm_skinny_supported(a, split)) return sf_launch_gemm_skinny(a, split, s);
    return sf_gemm_tile_supported(a, split) ? sf_launch_gemm_tile(a, s) : hipErrorInvalidValue;
  }
  if (sf_gemm_skinny_supported(a, split) && !sf_sw(SW_DISABLE_SKINNY)) return sf_launch_gemm_skinny(a, split, s);
  if (sf_gemm_tile_supported(a, split)) return sf_launch_gemm_tile(a, s);
#ifdef SF_LAB      // round-4 epilogue-overlap experiments (profiles/r04_panel_overlap_lab.txt): lab library only, behind SF_PANEL_PIPE / SF_PANEL_PP
  if (sf_gemm_pipe_supported(a, split)) return sf_launch_gemm_pipe(a, s);
  if (sf_gemm_pp_supported(a, split)) return sf_launch_gemm_pp(a, s);
#endif
  if (sf_gemm_panel_supported(a, split)) return sf_launch_gemm_panel(a, s);
  if (sf_gemm256_supported(a, split)) return sf_launch_gemm256(a, s);
  if (a.resid_hi) return hipErrorInvalidValue;      // plane-form residual: panel kernel only
  // N = 256 j + 128 (SigLIP-so400m: 1152, 3456): the first 256 j columns on the 256-column kernel, the last 128 on the 128^2 kernel — two
  // launches on disjoint column ranges of the same rows (every epilogue here is column-local), instead of the whole problem on 128^2 tiles
  if (a.N % 256 == 128 && a.N > 256 && !a.ln_stats && !a.ln_stats_out && a.epi != SF_EPI_EMBED_F32 && !sf_sw(SW_DISABLE_GEMM_COLSPLIT)) {
    SfGemmArgs m = a;
    m.N = a.N - 128;
    if (sf_gemm256_supported(m, split)) {
      SfGemmArgs t = a;
      const size_t n0 = (size_t)m.N;
      t.N = 128;
      t.w_hi = a.w_hi + n0 * a.K; if (a.w_lo) t.w_lo = a.w_lo + n0 * a.K;
      if (a.bias) t.bias = a.bias + n0;
      if (a.resid) t.resid = a.resid + n0;
      if (a.out_f32) t.out_f32 = a.out_f32 + n0;
      if (a.out_hi) t.out_hi = a.out_hi + n0;
      if (a.out_lo) t.out_lo = a.out_lo + n0;
      if (a.aux) t.aux = a.aux + n0;
      const hipError_t e = sf_launch_gemm256(m, s);
      if (e != hipSuccess) return e;
      return sf_launch_gemm128(t, split, s);
    }
  }
  return sf_launch_gemm128(a, split, s);
}

hipError_t sf_launch_gemm128(const SfGemmArgs& a, bool split, hipStream_t s) {
  const int bk = split ? 32 : 64;
  if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % bk) || (a.N % 4) || (a.ldc % 4)) return hipErrorInvalidValue;
  if (split && (!a.a_lo || !a.w_lo)) return hipErrorInvalidValue;
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  const size_t lds = 2 * (size_t)BM * 64 * 2 * 2;   // 64 KB: two stages of 32 KB in both modes
  static SfPerDeviceOnce attr_set;
  if (attr_set.first()) {
    // > 48 KB of dynamic LDS needs the opt-in attribute once per kernel
#define SF_ATTR(S, E) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_gemm_kernel<S, E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SF_ATTR(false, SF_EPI_F32) SF_ATTR(false, SF_EPI_BF16) SF_ATTR(false, SF_EPI_ACT_BF16)
    SF_ATTR(false, SF_EPI_RESID_F32) SF_ATTR(false, SF_EPI_EMBED_F32)
    SF_ATTR(true, SF_EPI_F32) SF_ATTR(true, SF_EPI_BF16) SF_ATTR(true, SF_EPI_ACT_BF16)
    SF_ATTR(true, SF_EPI_RESID_F32) SF_ATTR(true, SF_EPI_EMBED_F32)
#undef SF_ATTR
  }
  return split ? launch_epi<true>(a, dim3(tiles), lds, s) : launch_epi<false>(a, dim3(tiles), lds, s);
}
