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
#include "sf_switches.h"
#include <cstdlib>
#include <cstring>

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
#define WB_KM 32                                  // token rows per stage
#define WB_STAGE_BYTES (2 * WB_KM * WB_T * 2)     // dY tile | X tile: 32 KB
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

SF_DEVICE bf16x8_t wb_join(s16x4_t lo, s16x4_t hi) {
  bf16x8_t f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

__global__ __launch_bounds__(WB_THREADS) void sf_wgrad256_kernel(SfWgradGroup G, int ntiles, int nsplit, int per_xcd, int kt_per,
                                                                 int kt_total, unsigned part_stride, unsigned bias_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 4 stages x (dY tile | X tile)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  // work item = (split, job, t1, t2), t2 fastest; an XCD (blockIdx % 8) takes a run of consecutive items, so the tiles that
  // share its L2 read the same token rows of the same dY / X column panels
  const int item = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (((int)blockIdx.x >> 3) >= per_xcd || item >= ntiles * nsplit) return;
  const int tile = item % ntiles, split = item / ntiles;
  int ji = 0;
#pragma unroll
  for (int j = 1; j < SF_WG_MAX_JOBS; ++j)
    if (j < G.njobs && tile >= G.job[j].tile0) ji = j;
  const SfWgradJob& J = G.job[ji];
  const int lt = tile - J.tile0;
  const int t1 = lt / J.tiles2, t2 = lt % J.tiles2;
  const int n1_0 = t1 * WB_T, n2_0 = t2 * WB_T;
  const int kt0 = split * kt_per;
  const int kt1 = min(kt_total, kt0 + kt_per);
  const bool do_bias = J.dbias != nullptr && t2 == 0;
  const int ldy = J.ldy, ldx = J.ldx;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)J.dy, 0, (unsigned)G.M * (unsigned)ldy * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)J.x, 0, (unsigned)G.M * (unsigned)ldx * 2u, 0x00020000);
  // A stage = 32 token rows of both operands ([32][256] bf16 each, 2 x 16 KB); four stages in a ring, three in flight.
  // 2 rounds x 16 rows per operand; LDS position p of row r holds source chunk p ^ ((r & 7) << 1)
  unsigned offa[2], offb[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int c = r * WB_THREADS + tid;
    const int row = c >> 5, pos = c & 31;
    const int src = pos ^ ((row & 7) << 1);
    offa[r] = ((unsigned)row * (unsigned)ldy + (unsigned)(n1_0 + src * 8)) * 2u;
    offb[r] = ((unsigned)row * (unsigned)ldx + (unsigned)(n2_0 + src * 8)) * 2u;
  }
  auto stage = [&](int slot, int ht) {
    char* base = smem + slot * WB_STAGE_BYTES + wave * 1024;
    const unsigned soa = (unsigned)ht * WB_KM * (unsigned)ldy * 2u;
    const unsigned sob = (unsigned)ht * WB_KM * (unsigned)ldx * 2u;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(base + r * 8192), 16, (int)offa[r], soa, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(base + WB_STAGE_BYTES / 2 + r * 8192), 16, (int)offb[r], sob, 0, 0);
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

  // One barrier per 32-row step.  Step h: wait for stage h alone (the two younger stages stay in flight: counted vmcnt, a
  // lane has 4 DMA loads per stage), barrier, refill the slot that step h-1 read (every wave is past it now) with stage
  // h+3, then 12 transposed fragment reads + 32 MFMAs.  Round 2's two-stage loop waited for ALL loads at every 64-row step.
  const int h0 = kt0 * 2, h1 = kt1 * 2;
  // fragment addresses (wb_frag's arithmetic, k-step 0): row = 4 g + (t16 >> 2), 32-byte column block cb ^ (row & 7)
  const int t16f = lane & 15, gf = lane >> 4;
  const int frow = 4 * gf + (t16f >> 2);
  const int rsw = frow & 7;
  const unsigned lane_off = (unsigned)(frow * (WB_T * 2) + (((t16f & 3) >> 1) << 4) + ((t16f & 1) << 3));
  const unsigned frag_a = lane_off + (unsigned)(wr * 8 << 5);       // A block wr*8 + i -> (wr*8 + (i ^ rsw)) << 5
  const int rsw_b = rsw & 3;                                           // B block wc*4 + j: the XOR splits into (j ^ (rsw & 3)) and bit 2
  const unsigned frag_b = lane_off + (unsigned)(((wc * 4) ^ (rsw & 4)) << 5);
  const unsigned lds0 = (unsigned)(unsigned long)(lptr_t)smem;
  auto step = [&](int ht) {          // after the wait + barrier of step ht
    const int slot = (ht - h0) & 3;
      // the 24 transposed reads of the step go out as inline asm: hipcc 7.2 puts an s_waitcnt vmcnt(0) in front of every
      // ds_read_b64_tr_b16 it can see while a buffer_load ... lds is outstanding (it cannot tell the slots apart), which
      // would drain the two stages in flight at every step.  The fragments pass through the lgkmcnt wait as operands so
      // that no MFMA is scheduled above it.
      const unsigned sa = lds0 + (unsigned)(slot * WB_STAGE_BYTES), sb = sa + WB_STAGE_BYTES / 2;
      s16x4_t fbl[4], fbh[4], fal[8], fah[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned ad = sb + frag_b + (unsigned)((j ^ rsw_b) << 5);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fbl[j]) : "v"(ad));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(fbh[j]) : "v"(ad));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned ad = sa + frag_a + (unsigned)((i ^ rsw) << 5);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fal[i]) : "v"(ad));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(fah[i]) : "v"(ad));
      }
      if (ht + 3 < h1) stage((slot + 3) & 3, ht + 3);
      asm volatile("s_waitcnt lgkmcnt(8)"
                   : "+v"(fbl[0]), "+v"(fbh[0]), "+v"(fbl[1]), "+v"(fbh[1]), "+v"(fbl[2]), "+v"(fbh[2]), "+v"(fbl[3]), "+v"(fbh[3]),
                     "+v"(fal[0]), "+v"(fah[0]), "+v"(fal[1]), "+v"(fah[1]), "+v"(fal[2]), "+v"(fah[2]), "+v"(fal[3]), "+v"(fah[3]));
      bf16x8_t fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = wb_join(fbl[j], fbh[j]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8_t fa = wb_join(fal[i], fah[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = wg_mfma(fb[j], fa, acc[i][j]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fal[4]), "+v"(fah[4]), "+v"(fal[5]), "+v"(fah[5]), "+v"(fal[6]), "+v"(fah[6]), "+v"(fal[7]), "+v"(fah[7]));
#pragma unroll
      for (int i = 4; i < 8; ++i) {
        const bf16x8_t fa = wb_join(fal[i], fah[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = wg_mfma(fb[j], fa, acc[i][j]);
      }
      if (do_bias) {          // column sums of dY: this wave's two of the eight 16-column blocks (read again: a register index
                              // that depends on the wave would turn into a select chain over the eight fragments)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const unsigned ad = sa + frag_a + (unsigned)(((wc * 2 + q) ^ rsw) << 5);
          s16x4_t bl, bh;
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bl) : "v"(ad));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(bh) : "v"(ad));
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bl), "+v"(bh));
          accb[q] = wg_mfma(ones, wb_join(bl, bh), accb[q]);
        }
      }
  };
  if (h0 < h1) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (h0 + i < h1) stage(i, h0 + i);
    int ht = h0;
    for (; ht < h1 - 2; ++ht) {       // two younger stages stay in flight
      asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      step(ht);
    }
    if (ht < h1 - 1) {                // one younger stage
      asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      step(ht);
      ++ht;
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    step(ht);
  }

  const int l15 = lane & 15, g = lane >> 4;
  if (nsplit == 1) {            // the workgroup saw every token row: write the gradient itself
    const float alpha = J.alpha;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n1 = n1_0 + wr * 128 + i * 16 + l15;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n2 = n2_0 + wc * 64 + j * 16 + g * 4;
        f32x4_t* dst = reinterpret_cast<f32x4_t*>(J.out + (size_t)n1 * J.ldo + n2);
        f32x4_t o = acc[i][j] * alpha;
        if (J.accumulate) o += *dst;
        *dst = o;
      }
    }
    if (do_bias && g == 0) {
#pragma unroll
      for (int q = 0; q < 2; ++q) J.dbias[n1_0 + wr * 128 + (wc * 2 + q) * 16 + l15] += alpha * accb[q][0];
    }
    return;
  }
  float* part = G.partial + (size_t)split * part_stride + J.part_off;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n1 = n1_0 + wr * 128 + i * 16 + l15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n2 = n2_0 + wc * 64 + j * 16 + g * 4;
      *reinterpret_cast<f32x4_t*>(part + (size_t)n1 * J.N2 + n2) = acc[i][j];
    }
  }
  if (do_bias && g == 0) {
    float* bp = G.partial + (size_t)nsplit * part_stride + (size_t)split * bias_stride + J.bias_off;
#pragma unroll
    for (int q = 0; q < 2; ++q) bp[n1_0 + wr * 128 + (wc * 2 + q) * 16 + l15] = accb[q][0];
  }
}

// out (+)= alpha * sum_s partial[s] for every job of the group, in split order (deterministic); one block = 1024 consecutive
// floats of one job (N1 * N2 is a multiple of 65 536).  Trailing blocks reduce the bias partials, 256 entries each.
__global__ __launch_bounds__(256) void sf_wgrad_group_reduce_kernel(SfWgradGroup G, int nsplit, unsigned part_stride, unsigned bias_stride,
                                                                    int main_blocks) {
  if ((int)blockIdx.x >= main_blocks) {
    const unsigned n = ((unsigned)blockIdx.x - (unsigned)main_blocks) * 256u + threadIdx.x;      // position in the bias block
    if (n >= bias_stride) return;
    int ji = 0;
#pragma unroll
    for (int j = 1; j < SF_WG_MAX_JOBS; ++j)
      if (j < G.njobs && n >= G.job[j].bias_off) ji = j;
    const SfWgradJob& J = G.job[ji];
    if (!J.dbias) return;
    const float* bp = G.partial + (size_t)nsplit * part_stride + n;
    float t = 0.f;
    for (int s = 0; s < nsplit; ++s) t += bp[(size_t)s * bias_stride];
    J.dbias[n - J.bias_off] += J.alpha * t;
    return;
  }
  const unsigned e = ((unsigned)blockIdx.x * 256u + threadIdx.x) * 4u;       // float position in one split's block
  int ji = 0;
#pragma unroll
  for (int j = 1; j < SF_WG_MAX_JOBS; ++j)
    if (j < G.njobs && e >= G.job[j].part_off) ji = j;
  const SfWgradJob& J = G.job[ji];
  const float* pp = G.partial + e;
  f32x4_t t = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < nsplit; ++s) t += *reinterpret_cast<const f32x4_t*>(pp + (size_t)s * part_stride);
  const unsigned le = e - J.part_off;
  const unsigned n1 = le / (unsigned)J.N2, n2 = le % (unsigned)J.N2;
  f32x4_t* dst = reinterpret_cast<f32x4_t*>(J.out + (size_t)n1 * J.ldo + n2);
  f32x4_t o = t * J.alpha;
  if (J.accumulate) o += *dst;
  *dst = o;
}

// out (+)= alpha * sum_s partial[s] (128^2 path).  The outputs of this path are small (LoRA factors, head projections) and the
// split count large (up to 98), so a block covers 64 float4 positions x 4 split lanes: lane q adds splits q, q+4, ... and the
// four lane sums are combined in a fixed order through LDS — 4x the workgroups and a quarter of the dependent loads.
__global__ __launch_bounds__(256) void sf_wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, size_t n12, int N2,
                                                              float alpha, float* out, int ldo, int accumulate) {
  __shared__ f32x4_t red[4][64];
  const int pos = threadIdx.x & 63, q = threadIdx.x >> 6;
  const size_t i4 = (size_t)blockIdx.x * 64 + pos;
  const bool in = i4 * 4 < n12;
  f32x4_t t = {0.f, 0.f, 0.f, 0.f};
  if (in) {
    const float* pp = partial + i4 * 4;
    int sidx = q;
    for (; sidx + 12 < nsplit; sidx += 16) {       // four independent loads in flight
      const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(pp + (size_t)sidx * n12);
      const f32x4_t a1 = *reinterpret_cast<const f32x4_t*>(pp + (size_t)(sidx + 4) * n12);
      const f32x4_t a2 = *reinterpret_cast<const f32x4_t*>(pp + (size_t)(sidx + 8) * n12);
      const f32x4_t a3 = *reinterpret_cast<const f32x4_t*>(pp + (size_t)(sidx + 12) * n12);
      t += a0; t += a1; t += a2; t += a3;
    }
    for (; sidx < nsplit; sidx += 4) t += *reinterpret_cast<const f32x4_t*>(pp + (size_t)sidx * n12);
  }
  red[q][pos] = t;
  __syncthreads();
  if (q != 0 || !in) return;
  t = ((red[0][pos] + red[1][pos]) + red[2][pos]) + red[3][pos];
  const size_t n1 = (i4 * 4) / N2, n2 = (i4 * 4) % N2;
  f32x4_t* dst = reinterpret_cast<f32x4_t*>(out + n1 * ldo + n2);
  f32x4_t o = t * alpha;
  if (accumulate) o += *dst;
  *dst = o;
}

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

// ---- grouped 256^2 launches ----------------------------------------------------------------------------------------------
// split count of a group of `ntiles` tiles over kt K-steps: rounds x (K-steps per workgroup + the tile write) + the reduce
// pass, in K-step units (a K-step is ~2.4 us, a 256 KB partial write ~6 of them, reducing one split of one tile ~0.03)
static int wgg_nsplit(int ntiles, int kt) {
  const int forced = sf_sw(SW_WGRAD_NSPLIT) ? atoi(sf_sw(SW_WGRAD_NSPLIT)) : 0;     // lab: tools/wgrad_lab.py
  if (forced > 0) return forced;
  const int cus = wg_cus();
  int best = 1;
  double best_cost = 1e30;
  const int smax = kt / 4 < 1 ? 1 : (kt / 4 > 64 ? 64 : kt / 4);
  for (int s = 1; s <= smax; ++s) {
    const int per = (kt + s - 1) / s;
    if ((kt + per - 1) / per != s) continue;          // same split count after rounding
    const int rounds = (ntiles * s + cus - 1) / cus;
    const double cost = (double)rounds * (per + 6.0) + (s > 1 ? 0.03 * s * ntiles : 0.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
  }
  return best;
}
bool sf_wgrad_groupable(int M, int N1, int N2) {
  const bool small_only = sf_sw(SW_WGRAD_SMALL_TILES) != nullptr;
  return N1 > 0 && N2 > 0 && (N1 % WB_T == 0) && (N2 % WB_T == 0) && (M + WG_KM - 1) / WG_KM >= 32 && !small_only;
}
size_t sf_wgrad_group_partial_floats(int M, int ntiles, int sum_n1) {
  const int kt = (M + WG_KM - 1) / WG_KM;
  return (size_t)wgg_nsplit(ntiles, kt) * ((size_t)ntiles * WB_T * WB_T + (size_t)sum_n1);
}

hipError_t sf_launch_wgrad_group(SfWgradGroup& g, hipStream_t s) {
  if (g.njobs <= 0 || g.njobs > SF_WG_MAX_JOBS || g.M <= 0 || !g.partial) return hipErrorInvalidValue;
  int ntiles = 0;
  size_t part = 0, bias = 0;
  for (int j = 0; j < g.njobs; ++j) {
    SfWgradJob& J = g.job[j];
    if (!sf_wgrad_groupable(g.M, J.N1, J.N2) || !J.dy || !J.x || !J.out) return hipErrorInvalidValue;
    if ((J.ldy % 8) || (J.ldx % 8) || (J.ldo % 4)) return hipErrorInvalidValue;
    if ((size_t)g.M * J.ldy * 2 >= ((size_t)1 << 32) || (size_t)g.M * J.ldx * 2 >= ((size_t)1 << 32)) return hipErrorInvalidValue;
    J.tile0 = ntiles; J.tiles2 = J.N2 / WB_T;
    J.part_off = (unsigned)part; J.bias_off = (unsigned)bias;
    ntiles += (J.N1 / WB_T) * J.tiles2;
    part += (size_t)J.N1 * J.N2; bias += (size_t)J.N1;
  }
  if (part >= ((size_t)1 << 31)) return hipErrorInvalidValue;
  const int kt_total = (g.M + WG_KM - 1) / WG_KM;
  int nsplit = wgg_nsplit(ntiles, kt_total);
  const int kt_per = (kt_total + nsplit - 1) / nsplit;
  nsplit = (kt_total + kt_per - 1) / kt_per;
  const int items = ntiles * nsplit;
  const int per_xcd = (items + 7) / 8;
  static SfPerDeviceOnce attr_set;
  if (attr_set.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_wgrad256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WB_TILE_BYTES);
  hipLaunchKernelGGL(sf_wgrad256_kernel, dim3(per_xcd * 8), dim3(WB_THREADS), 4 * WB_TILE_BYTES, s, g, ntiles, nsplit, per_xcd, kt_per, kt_total,
                     (unsigned)part, (unsigned)bias);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || nsplit == 1) return e;
  const int main_blocks = (int)(part / 1024);
  const int bias_blocks = (int)((bias + 255) / 256);
  hipLaunchKernelGGL(sf_wgrad_group_reduce_kernel, dim3(main_blocks + bias_blocks), dim3(256), 0, s, g, nsplit, (unsigned)part, (unsigned)bias,
                     main_blocks);
  return hipGetLastError();
}

struct WgPlan { int tiles1, tiles2, ntiles, kt_total, kt_per, nsplit; };
static WgPlan wg_plan(int M, int N1, int N2) {       // 128^2 tiles: ~2 rounds of 2 workgroups per CU
  WgPlan pl;
  pl.kt_total = (M + WG_KM - 1) / WG_KM;
  pl.tiles1 = (N1 + WG_T - 1) / WG_T;
  pl.tiles2 = (N2 + WG_T - 1) / WG_T;
  pl.ntiles = pl.tiles1 * pl.tiles2;
  int s = (1024 + pl.ntiles - 1) / pl.ntiles;
  if (s > pl.kt_total / 4) s = pl.kt_total / 4;        // at least 4 K-steps per workgroup
  if (s < 1) s = 1;
  pl.kt_per = (pl.kt_total + s - 1) / s;
  pl.nsplit = (pl.kt_total + pl.kt_per - 1) / pl.kt_per;
  return pl;
}

size_t sf_wgrad_partial_floats(int M, int N1, int N2) {
  if (sf_wgrad_groupable(M, N1, N2)) return sf_wgrad_group_partial_floats(M, (N1 / WB_T) * (N2 / WB_T), N1);
  const WgPlan pl = wg_plan(M, N1, N2);
  return (size_t)pl.nsplit * N1 * N2;
}

hipError_t sf_launch_wgrad(const SfWgradArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N1 <= 0 || a.N2 <= 0) return hipErrorInvalidValue;
  if ((a.ldy % 8) || (a.ldx % 8) || (a.N2 % 4) || (a.ldo % 4)) return hipErrorInvalidValue;
  if ((size_t)a.M * a.ldy * 2 >= ((size_t)1 << 32) || (size_t)a.M * a.ldx * 2 >= ((size_t)1 << 32)) return hipErrorInvalidValue;
  if (a.out && sf_wgrad_groupable(a.M, a.N1, a.N2)) {      // a group of one
    SfWgradGroup g;
    memset(&g, 0, sizeof(g));
    g.njobs = 1; g.M = a.M; g.partial = a.partial;
    SfWgradJob& J = g.job[0];
    J.dy = a.dy; J.x = a.x; J.out = a.out; J.dbias = a.dbias; J.ldy = a.ldy; J.ldx = a.ldx; J.N1 = a.N1; J.N2 = a.N2; J.ldo = a.ldo;
    J.accumulate = a.accumulate; J.alpha = a.alpha;
    return sf_launch_wgrad_group(g, s);
  }
  const WgPlan pl = wg_plan(a.M, a.N1, a.N2);
  const size_t n12 = (size_t)a.N1 * a.N2;
  static SfPerDeviceOnce attr_set;
  if (attr_set.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sf_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WG_TILE_BYTES);
  hipLaunchKernelGGL(sf_wgrad_kernel, dim3(pl.ntiles * pl.nsplit), dim3(WG_THREADS), 4 * WG_TILE_BYTES, s, a, pl.tiles2, pl.ntiles,
                     pl.kt_per, pl.kt_total);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (a.out) {
    hipLaunchKernelGGL(sf_wgrad_reduce_kernel, dim3((unsigned)((n12 / 4 + 63) / 64)), dim3(256), 0, s, a.partial, pl.nsplit, n12, a.N2,
                       a.alpha, a.out, a.ldo, a.accumulate);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (a.dbias)      // small-tile path: separate column-sum kernel
    e = sf_launch_colsum_bf16(a.dy, a.M, a.N1, a.ldy, a.alpha, a.dbias, 1, a.dbias_scratch, s);
  return e;
}
