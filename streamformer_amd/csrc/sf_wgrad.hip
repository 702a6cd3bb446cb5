// Weight-gradient GEMM of the training step:  C[N1,N2] = sum_m dY[m,N1] * X[m,N2]
// (torch autograd's grad_weight = grad_output^T @ input for every nn.Linear of
// modeling_timesformer_siglip.py:513, 629, 728, 811, 830, 895, 1118-1119 and the patch conv :329-334).
//
// gfx950 design: both operands are TOKEN-major (the contraction index m is the row index), which is
// exactly how the forward/backward kernels leave activations and their gradients in HBM — no
// transposed copies are made.  A [64 m x 128 col] bf16 tile of each operand goes HBM -> LDS by
// buffer_load ... lds (rows past M read as zero through the buffer descriptor), row-major, and the
// MFMA fragments (8 consecutive-k values per lane for one column) come out of it with
// ds_read_b64_tr_b16: one 16-lane group reads a [4 rows x 16 cols] block and receives it transposed.
// A lane's 8 k-values are rows {4g..4g+3} and {16+4g..16+4g+3} of the 32-row k-step — a permutation
// of k that both operands share, so the product is unchanged.  32-byte column blocks are XOR-swizzled
// with (row & 7) on the DMA source address, so the 8 rows a half-wave reads hit all 64 banks.
// The M range is split across workgroups (fp32 partial tiles) and reduced in a fixed order.
#include "sf_train.h"
#include <cstdlib>

#define WG_T 128          // tile edge (both N1 and N2)
#define WG_KM 64          // token rows per K-step
#define WG_THREADS 256
#define WG_TILE_BYTES (WG_KM * WG_T * 2)   // 16 KB

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

SF_DEVICE f32x4_t wg_mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}

// fragment for column block cb (16 columns) of a row-major [64][128] tile, k-step ks (32 rows)
SF_DEVICE bf16x8_t wg_frag(const char* tile, int ks, int cb, int lane) {
  const int t16 = lane & 15, g = lane >> 4;
  const int row = ks * 32 + 4 * g + (t16 >> 2);
  const int rsw = row & 7;
  const int off = row * (WG_T * 2) + ((((cb ^ rsw) << 1) + ((t16 & 3) >> 1)) << 4) + ((t16 & 1) << 3);
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + off));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + off + 16 * WG_T * 2));
  bf16x8_t f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

__global__ __launch_bounds__(WG_THREADS) void sf_wgrad_kernel(SfWgradArgs p, int tiles2, int ntiles, int kt_per, int kt_total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (A tile | B tile)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tile = blockIdx.x % ntiles, split = blockIdx.x / ntiles;
  const int n1_0 = (tile / tiles2) * WG_T, n2_0 = (tile % tiles2) * WG_T;
  const int kt0 = split * kt_per;
  const int kt1 = min(kt_total, kt0 + kt_per);

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (unsigned)p.M * (unsigned)p.ldy * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)p.M * (unsigned)p.ldx * 2u, 0x00020000);

  // per-lane DMA source offsets: 4 rounds x 16 rows; LDS position p of row r holds source chunk
  // p ^ ((r & 7) << 1) (16-byte chunks; the XOR moves whole 32-byte column blocks)
  unsigned offa[4], offb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = r * WG_THREADS + tid;
    const int row = c >> 4, pos = c & 15;
    const int src = pos ^ ((row & 7) << 1);
    offa[r] = ((unsigned)row * (unsigned)p.ldy + (unsigned)(n1_0 + src * 8)) * 2u;
    offb[r] = ((unsigned)row * (unsigned)p.ldx + (unsigned)(n2_0 + src * 8)) * 2u;
  }
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * 2 * WG_TILE_BYTES + wave * 1024;
    const unsigned soa = (unsigned)kt * WG_KM * (unsigned)p.ldy * 2u;
    const unsigned sob = (unsigned)kt * WG_KM * (unsigned)p.ldx * 2u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(base + r * 4096), 16, offa[r], soa, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(base + WG_TILE_BYTES + r * 4096), 16, offb[r], sob, 0, 0);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  if (kt0 < kt1) {
    stage(0, kt0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const int cur = (kt - kt0) & 1;
      if (kt + 1 < kt1) stage(cur ^ 1, kt + 1);
      const char* ta = smem + cur * 2 * WG_TILE_BYTES;
      const char* tb = ta + WG_TILE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fa[i] = wg_frag(ta, ks, wr * 4 + i, lane);
          fb[i] = wg_frag(tb, ks, wc * 4 + i, lane);
        }
        // swapped issue: lane ends up with C[n1 = .. + l15][n2 = .. + 4g .. 4g+3]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = wg_mfma(fb[j], fa[i], acc[i][j]);
      }
      __syncthreads();
    }
  }

  float* part = p.partial + (size_t)split * p.N1 * p.N2;
  const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n1 = n1_0 + wr * 64 + i * 16 + l15;
    if (n1 >= p.N1) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n2 = n2_0 + wc * 64 + j * 16 + g * 4;
      if (n2 >= p.N2) continue;            // N2 % 4 == 0 (launcher)
      *reinterpret_cast<f32x4_t*>(part + (size_t)n1 * p.N2 + n2) = acc[i][j];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 tile variant for the big projections (N1, N2 multiples of 256): 8 waves (2 x 4), a wave owns
// 128 x 64 of C = 8 x 4 MFMA tiles (24 transposed reads per 32 MFMAs), [64 m x 256] operand tiles,
// double-buffered (128 KB of LDS), one workgroup per CU and the M range split so that tiles x splits
// fills the chip once.  The bias gradient rides along for free: workgroups of the first tile column
// multiply the dY fragments by a ones fragment (column sums over m) — two extra MFMAs per k-step and wave.
// ------------------------------------------------------------------------------------------------
#define WB_T 256
#define WB_THREADS 512
#define WB_TILE_BYTES (WG_KM * WB_T * 2)   // 32 KB

SF_DEVICE bf16x8_t wb_frag(const char* tile, int ks, int cb, int lane) {
  const int t16 = lane & 15, g = lane >> 4;
  const int row = ks * 32 + 4 * g + (t16 >> 2);
  const int off = row * (WB_T * 2) + ((((cb ^ (row & 7)) << 1) + ((t16 & 3) >> 1)) << 4) + ((t16 & 1) << 3);
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + off));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + off + 16 * WB_T * 2));
  bf16x8_t f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

__global__ __launch_bounds__(WB_THREADS) void sf_wgrad256_kernel(SfWgradArgs p, int tiles2, int ntiles, int kt_per, int kt_total,
                                                                 float* bias_partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (A tile | B tile)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int tile = blockIdx.x % ntiles, split = blockIdx.x / ntiles;
  const int t1 = tile / tiles2, t2 = tile % tiles2;
  const int n1_0 = t1 * WB_T, n2_0 = t2 * WB_T;
  const int kt0 = split * kt_per;
  const int kt1 = min(kt_total, kt0 + kt_per);
  const bool do_bias = bias_partial != nullptr && t2 == 0;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (unsigned)p.M * (unsigned)p.ldy * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)p.M * (unsigned)p.ldx * 2u, 0x00020000);
  // 4 rounds x 16 rows per operand; LDS position p of row r holds source chunk p ^ ((r & 7) << 1)
  unsigned offa[4], offb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = r * WB_THREADS + tid;
    const int row = c >> 5, pos = c & 31;
    const int src = pos ^ ((row & 7) << 1);
    offa[r] = ((unsigned)row * (unsigned)p.ldy + (unsigned)(n1_0 + src * 8)) * 2u;
    offb[r] = ((unsigned)row * (unsigned)p.ldx + (unsigned)(n2_0 + src * 8)) * 2u;
  }
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * 2 * WB_TILE_BYTES + wave * 1024;
    const unsigned soa = (unsigned)kt * WG_KM * (unsigned)p.ldy * 2u;
    const unsigned sob = (unsigned)kt * WG_KM * (unsigned)p.ldx * 2u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(base + r * 8192), 16, offa[r], soa, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(base + WB_TILE_BYTES + r * 8192), 16, offb[r], sob, 0, 0);
    }
  };

  f32x4_t acc[8][4];
  f32x4_t accb[2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  accb[0] = accb[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (short)0x3f80;

  if (kt0 < kt1) {
    stage(0, kt0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const int cur = (kt - kt0) & 1;
      if (kt + 1 < kt1) stage(cur ^ 1, kt + 1);
      const char* ta = smem + cur * 2 * WB_TILE_BYTES;
      const char* tb = ta + WB_TILE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = wb_frag(tb, ks, wc * 4 + j, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bf16x8_t fa = wb_frag(ta, ks, wr * 8 + i, lane);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = wg_mfma(fb[j], fa, acc[i][j]);
        }
        if (do_bias) {          // column sums of dY: this wave's two of the eight 16-column blocks
#pragma unroll
          for (int q = 0; q < 2; ++q) accb[q] = wg_mfma(ones, wb_frag(ta, ks, wr * 8 + wc * 2 + q, lane), accb[q]);
        }
      }
      __syncthreads();
    }
  }

  float* part = p.partial + (size_t)split * p.N1 * p.N2;
  const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n1 = n1_0 + wr * 128 + i * 16 + l15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n2 = n2_0 + wc * 64 + j * 16 + g * 4;
      *reinterpret_cast<f32x4_t*>(part + (size_t)n1 * p.N2 + n2) = acc[i][j];
    }
  }
  if (do_bias && g == 0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) bias_partial[(size_t)split * p.N1 + n1_0 + wr * 128 + (wc * 2 + q) * 16 + l15] = accb[q][0];
  }
}

// out (+)= alpha * sum_s partial[s]; the trailing workgroups of the same launch reduce the bias partials
// (dbias[n1] += alpha * sum_s bias_partial[s][n1]) when the tile kernel produced them
__global__ __launch_bounds__(256) void sf_wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, size_t n12, int N2,
                                                              float alpha, float* out, int ldo, int accumulate, int main_blocks,
                                                              const float* __restrict__ bias_partial, int N1, float* dbias) {
  if ((int)blockIdx.x >= main_blocks) {
    const int n = ((int)blockIdx.x - main_blocks) * 256 + threadIdx.x;
    if (n >= N1) return;
    float t = 0.f;
    for (int s = 0; s < nsplit; ++s) t += bias_partial[(size_t)s * N1 + n];
    dbias[n] += alpha * t;
    return;
  }
  const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i4 * 4 >= n12) return;
  f32x4_t t = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < nsplit; ++s) t += *reinterpret_cast<const f32x4_t*>(partial + (size_t)s * n12 + i4 * 4);
  const size_t n1 = (i4 * 4) / N2, n2 = (i4 * 4) % N2;
  f32x4_t* dst = reinterpret_cast<f32x4_t*>(out + n1 * ldo + n2);
  f32x4_t o = t * alpha;
  if (accumulate) o += *dst;
  *dst = o;
}

struct WgPlan { int big, tiles1, tiles2, ntiles, kt_total, kt_per, nsplit; };
static int wg_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus < 16) cus = 256;
  }
  return cus;
}
static WgPlan wg_plan(int M, int N1, int N2) {
  WgPlan pl;
  pl.kt_total = (M + WG_KM - 1) / WG_KM;
  pl.big = (N1 % WB_T == 0) && (N2 % WB_T == 0) && pl.kt_total >= 32 && !getenv("SF_WGRAD_SMALL_TILES");
  const int T = pl.big ? WB_T : WG_T;
  pl.tiles1 = (N1 + T - 1) / T;
  pl.tiles2 = (N2 + T - 1) / T;
  pl.ntiles = pl.tiles1 * pl.tiles2;
  // 128^2: ~2 rounds of 2 workgroups per CU; 256^2: one workgroup per CU, one round
  int s = pl.big ? wg_cus() / pl.ntiles : (1024 + pl.ntiles - 1) / pl.ntiles;
  if (s > pl.kt_total / 4) s = pl.kt_total / 4;        // at least 4 K-steps per workgroup
  if (s < 1) s = 1;
  pl.kt_per = (pl.kt_total + s - 1) / s;
  pl.nsplit = (pl.kt_total + pl.kt_per - 1) / pl.kt_per;
  return pl;
}

size_t sf_wgrad_partial_floats(int M, int N1, int N2) {
  const WgPlan pl = wg_plan(M, N1, N2);
  return (size_t)pl.nsplit * N1 * N2 + (size_t)pl.nsplit * N1;     // + bias partials
}

hipError_t sf_launch_wgrad(const SfWgradArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N1 <= 0 || a.N2 <= 0) return hipErrorInvalidValue;
  if ((a.ldy % 8) || (a.ldx % 8) || (a.N2 % 4) || (a.ldo % 4)) return hipErrorInvalidValue;
  if ((size_t)a.M * a.ldy * 2 >= ((size_t)1 << 32) || (size_t)a.M * a.ldx * 2 >= ((size_t)1 << 32)) return hipErrorInvalidValue;
  const WgPlan pl = wg_plan(a.M, a.N1, a.N2);
  const size_t n12 = (size_t)a.N1 * a.N2;
  float* bias_partial = a.partial + (size_t)pl.nsplit * n12;
  static SfPerDeviceOnce attr_set;
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WG_TILE_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_wgrad256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WB_TILE_BYTES);
  }
  bool bias_done = false;
  if (pl.big) {
    hipLaunchKernelGGL(sf_wgrad256_kernel, dim3(pl.ntiles * pl.nsplit), dim3(WB_THREADS), 4 * WB_TILE_BYTES, s, a, pl.tiles2, pl.ntiles,
                       pl.kt_per, pl.kt_total, a.dbias ? bias_partial : nullptr);
    bias_done = a.dbias != nullptr;
  } else {
    hipLaunchKernelGGL(sf_wgrad_kernel, dim3(pl.ntiles * pl.nsplit), dim3(WG_THREADS), 4 * WG_TILE_BYTES, s, a, pl.tiles2, pl.ntiles,
                       pl.kt_per, pl.kt_total);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  {
    const int main_blocks = a.out ? (int)((n12 / 4 + 255) / 256) : 0;
    const int bias_blocks = bias_done ? (a.N1 + 255) / 256 : 0;
    if (main_blocks + bias_blocks > 0)
      hipLaunchKernelGGL(sf_wgrad_reduce_kernel, dim3((unsigned)(main_blocks + bias_blocks)), dim3(256), 0, s, a.partial, pl.nsplit, n12, a.N2,
                         a.alpha, a.out, a.ldo, a.accumulate, main_blocks, bias_partial, a.N1, a.dbias);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (a.dbias && !bias_done)      // small-tile path: separate column-sum kernel (scratch after the tile partials)
    e = sf_launch_colsum_bf16(a.dy, a.M, a.N1, a.ldy, a.alpha, a.dbias, 1, a.dbias_scratch, s);
  return e;
}
